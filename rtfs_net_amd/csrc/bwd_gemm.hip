// Backward GEMMs of the training step.
//
//   rtfs_wgrad               dW[n][k] += sum_rows dY[row][n] * X[row][k]   (weight gradient of every 1x1 conv / linear map;
//                            X may be re-derived on load: gateway, PReLU, ReLU(gLN); rows may be segmented with an offset,
//                            which turns the unfold / conv-transpose Toeplitz structure into 8 plain calls)
//   rtfs_fold_gemm_bwd       d(LN4D output) of the dual-path layer-0 GEMM:  dxn[pos][c] = sum_kk dU0[pos-kk] . W0[kk*64+c]
//   rtfs_convt_bwd_input     dH3[l][j] = sum_{k,c} dG[l+k][c] * Wct[j][c][k]   (adjoint of rtfs_dp_convt_fwd)
//   (input gradients of the 1x1 convs are plain row GEMMs: rtfs_gemm_rows_fwd with the transposed weight)
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) with the ROW index as the contraction dimension: both operands sit in LDS
// row-major as they are in HBM and are read with ds_read_b32 (lanes = consecutive columns, conflict-free).
#include "common.h"

namespace rtfs {

// acc[a][b] += sum_k A[k][arow] * B[k][brow]; A, B stored [k][row] in LDS (row contiguous).  kdepth multiple of 8.
template <int TA, int TB>
__device__ __forceinline__ void mma_block_kk(floatx16 (&acc)[TA][TB], const float* As, int lda, const float* Bs, int ldb, int kdepth) {
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const float* ap = As + (kh * 4) * lda + i;
    const float* bp = Bs + (kh * 4) * ldb + i;
#pragma unroll 2
    for (int q = 0; q < kdepth; q += 8) {
        float a[TA][4], b[TB][4];
#pragma unroll
        for (int m = 0; m < TA; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[m][r] = ap[(q + r) * lda + m * 32];
#pragma unroll
        for (int n = 0; n < TB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) b[n][r] = bp[(q + r) * ldb + n * 32];
#pragma unroll
        for (int m = 0; m < TA; ++m)
#pragma unroll
            for (int n = 0; n < TB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][r], b[n][r], acc[m][n], 0, 0, 0);
    }
}

struct WgradArgs {
    const float* dY;
    int ldy;
    const float* X;
    int ldx;
    float* dW;
    int ldw;
    float* dbias;      // optional: dbias[n] += sum_r dY[r][n] (the conv bias gradient rides along; tile column 0, shift 0 only)
    int M;             // rows of dY
    int seg_len;       // rows per segment of dY (0: one segment)
    int x_seg;         // rows per segment of X
    int x_off;         // X row inside the segment = l + x_off + shift (invalid -> zero row)
    int nshift;        // number of shifts (Toeplitz taps) computed by this launch; shift z writes dW columns z*KIN..
    int NOUT, KIN;     // multiples of 32
    int rows_per_wg, ngroups;
    // prologue parameters
    const float *p0, *p1;  // gateway: gw, gb;  relu(gLN): gamma, beta
    float slope;
    const double* slot;
    double inv_n;
    int rows_per_b;
};

// PRO: 0 plain, 1 gateway prelu(x*gw+gb), 2 prelu(x), 3 relu(gLN(x))
// One workgroup = a (32*NT) x (32*KT) tile of dW over rows_per_wg rows of one shift.  The 4 waves split the ROWS of every
// 32-row LDS stage (8 each) and all hold the whole tile (no zero-padded MFMA work for the 64-wide maps); they are summed
// through LDS at the end and the tile leaves with one coalesced fp32 atomic per element.  Global loads of stage i+1 are in
// flight during the MFMAs of stage i (register prefetch, two LDS stages, one barrier per stage).  1-D grid decoded so that
// all (shift, tile) workgroups of one row range run back-to-back on ONE XCD: the 8 taps re-read the same rows from that L2.
// P: precision of the contraction (common.h NT: 0 fp32, 1 bf16, 3 split-bf16).  P != 0: the K = 16 instruction is fed the wave's 8 rows by
// lanes 0-31 (packed in registers from the scalar LDS reads) and zeros by lanes 32-63 - half of its K is padding, still 8/3 of the fp32 rate.
template <int NT, int KT, int PRO, int P = 0>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
    constexpr int NW = NT * 32, KW = KT * 32, LDY = NW + 4, LDX = KW + 4, CH = 32, STAGE = CH * (LDY + LDX);
    static_assert(NW * LDX <= 2 * STAGE, "reduction buffer aliases the stages");
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int kblocks = (a.KIN + KW - 1) / KW, nblocks = (a.NOUT + NW - 1) / NW;
    const int per_group = a.nshift * kblocks * nblocks;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int group = (local / per_group) * 8 + xcd, inner = local % per_group;
    if (group >= a.ngroups) return;
    const int z = inner % a.nshift, blk = inner / a.nshift;
    const int n0 = (blk / kblocks) * NW, k0 = (blk % kblocks) * KW;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = group * a.rows_per_wg;
    const int r1 = r0 + a.rows_per_wg < a.M ? r0 + a.rows_per_wg : a.M;
#ifdef WG_TIMING
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    const int xoff = a.x_off + z;

    // prefetch registers of the next stage (a second set, i.e. two stages of lead, measured no faster and spills the bf16 forms)
    struct Pref {
        float4 yr[NT], xr[KT];
        unsigned xvalid;
        int xb[KT];
        bool lean;  // the stage came through the lean path
    };
    Pref pa;
    float4 bs[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) bs[it] = f4(0, 0, 0, 0);
    const bool want_bias = a.dbias != nullptr && z == 0 && (blk % kblocks) == 0;  // uniform over the workgroup
    // Plain maps (one segment, no shift, whole tiles) take a lean path for every stage that lies inside the workgroup's row range: running row
    // pointers, no clamps, no validity selects.  The general path spent ~140 VALU instructions per 32-row stage on address arithmetic (8 v_mul_lo_u32,
    // 6 v_mad_i64, clamps) and zero-selects against 32 MFMAs - and an fp32 MFMA shares the SIMD's vector lanes with them.
    const bool plain = P == 0 && a.seg_len == 0 && xoff == 0 && a.x_seg >= r1 && n0 + NW <= a.NOUT && k0 + KW <= a.KIN;
    const float* yptr[NT];
    const float* xptr[KT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        const int idx = threadIdx.x + it * 256, lr = idx / (NW / 4), q4 = (idx % (NW / 4)) * 4;
        yptr[it] = a.dY + (size_t)(r0 + lr) * a.ldy + n0 + q4;
    }
#pragma unroll
    for (int it = 0; it < KT; ++it) {
        const int idx = threadIdx.x + it * 256, lr = idx / (KW / 4), q4 = (idx % (KW / 4)) * 4;
        xptr[it] = a.X + (size_t)(r0 + lr) * a.ldx + k0 + q4;
    }
    auto load = [&](Pref& pf, int rb) {
        float4(&yr)[NT] = pf.yr;
        float4(&xr)[KT] = pf.xr;
        int(&xb)[KT] = pf.xb;
        pf.lean = plain && rb + CH <= r1;
        if (pf.lean) {
            const size_t oy = (size_t)(rb - r0) * a.ldy, ox = (size_t)(rb - r0) * a.ldx;
#pragma unroll
            for (int it = 0; it < NT; ++it) yr[it] = ld4(yptr[it] + oy);
#pragma unroll
            for (int it = 0; it < KT; ++it) {
                xr[it] = ld4(xptr[it] + ox);
                if (PRO == 3) xb[it] = (rb + (int)((threadIdx.x + it * 256) / (KW / 4))) / a.rows_per_b;
            }
            return;
        }
        unsigned xvalid = 0;
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int idx = threadIdx.x + it * 256, lr = idx / (NW / 4), q4 = (idx % (NW / 4)) * 4;
            const int r = rb + lr;
            const bool ok = r < r1 && n0 + q4 < a.NOUT;
            const int rc = r < r1 ? r : r1 - 1, qc = n0 + q4 < a.NOUT ? n0 + q4 : 0;
            yr[it] = ld4(a.dY + (size_t)rc * a.ldy + qc);
            if (!ok) yr[it] = f4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const int idx = threadIdx.x + it * 256, lr = idx / (KW / 4), q4 = (idx % (KW / 4)) * 4;
            const int r = rb + lr, rc = r < r1 ? r : r1 - 1;
            int seq = 0, l = rc;
            if (a.seg_len) seq = (int)((unsigned)rc / (unsigned)a.seg_len), l = rc - seq * a.seg_len;
            const int xl = l + xoff;
            const bool ok = r < r1 && k0 + q4 < a.KIN && xl >= 0 && xl < a.x_seg;
            const int xlc = xl < 0 ? 0 : (xl < a.x_seg ? xl : a.x_seg - 1), qc = k0 + q4 < a.KIN ? k0 + q4 : 0;
            xr[it] = ld4(a.X + ((size_t)seq * a.x_seg + xlc) * a.ldx + qc);
            xvalid |= (ok ? 1u : 0u) << it;
            if (PRO == 3) xb[it] = rc / a.rows_per_b;
        }
        pf.xvalid = xvalid;
    };
    auto store = [&](const Pref& pf, float* st) {
        const float4(&yr)[NT] = pf.yr;
        const float4(&xr)[KT] = pf.xr;
        const int(&xb)[KT] = pf.xb;
        const unsigned xvalid = pf.xvalid;
        const bool lean = pf.lean;
        float* Ys = st;
        float* Xs = st + CH * LDY;
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int idx = threadIdx.x + it * 256, lr = idx / (NW / 4), q4 = (idx % (NW / 4)) * 4;
            st4(Ys + lr * LDY + q4, yr[it]);
            bs[it] = bs[it] + yr[it];  // this thread's column quad is the same in every stage
        }
#pragma unroll
        for (int it = 0; it < KT; ++it) {
            const int idx = threadIdx.x + it * 256, lr = idx / (KW / 4), q4 = (idx % (KW / 4)) * 4;
            float4 x = xr[it];
            const int kc = k0 + q4 < a.KIN ? k0 + q4 : 0;
            if (PRO == 1) x = prelu4(fma4(x, ld4(a.p0 + kc), ld4(a.p1 + kc)), a.slope);
            if (PRO == 2) x = prelu4(x, a.slope);
            if (PRO == 3) {
                float mean, rstd;
                stats_finalize(a.slot, xb[it], a.inv_n, mean, rstd);
                x = relu4(norm4(x, mean, rstd, ld4(a.p0 + kc), ld4(a.p1 + kc)));
            }
            if (!lean && !((xvalid >> it) & 1u)) x = f4(0, 0, 0, 0);
            st4(Xs + lr * LDX + q4, x);
        }
    };

    floatx16 acc[NT][KT];
    acc_zero(acc);
    load(pa, r0);
    store(pa, lds);
    __syncthreads();
    const int i = lane & 31, kh = lane >> 5;
    int cur = 0;
#ifdef WG_TIMING
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define WT_MARK(j) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[j] += t_ - tlast; tlast = t_; } while (0)
#else
#define WT_MARK(j) do { } while (0)
#endif
    auto stage = [&](int rb, Pref& pnext) {
        if (rb + CH < r1) load(pnext, rb + CH);  // in flight during this stage's MFMAs
        if constexpr (P == 0) {
            const float* yp = lds + cur * STAGE + (w * 8 + kh * 4) * LDY + i;
            const float* xp = lds + cur * STAGE + CH * LDY + (w * 8 + kh * 4) * LDX + i;
            float av[NT][4], bv[KT][4];
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) av[m][r] = yp[r * LDY + m * 32];
#pragma unroll
            for (int n = 0; n < KT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[n][r] = xp[r * LDX + n * 32];
#ifdef WG_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WT_MARK(0);
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < NT; ++m)
#pragma unroll
                    for (int n = 0; n < KT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][r], bv[n][r], acc[m][n], 0, 0, 0);
#ifdef WG_TIMING
            WT_MARK(1);
#endif
        } else {
            const float* yp = lds + cur * STAGE + (w * 8) * LDY + i;
            const float* xp = lds + cur * STAGE + CH * LDY + (w * 8) * LDX + i;
            const float keep = kh == 0 ? 1.f : 0.f;
            Frag fa[NT], fb[KT];
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = yp[r * LDY + m * 32] * keep;
                fa[m] = frag_f32<P>(f4(v[0], v[1], v[2], v[3]), f4(v[4], v[5], v[6], v[7]));
            }
#pragma unroll
            for (int n = 0; n < KT; ++n) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = xp[r * LDX + n * 32] * keep;
                fb[n] = frag_f32<P>(f4(v[0], v[1], v[2], v[3]), f4(v[4], v[5], v[6], v[7]));
            }
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int n = 0; n < KT; ++n) mma32<P>(acc[m][n], fa[m], fb[n]);
        }
        if (rb + CH < r1) store(pnext, lds + (cur ^ 1) * STAGE);
        WT_MARK(2);
        __syncthreads();
        WT_MARK(4);
        cur ^= 1;
    };
#pragma unroll 1
    for (int rb = r0; rb < r1; rb += CH) stage(rb, pa);
    if (want_bias) {
        constexpr int Q = NW / 4;  // column quads; thread t and stage slot `it` always hold quad t % Q
#pragma unroll
        for (int it = 0; it < NT; ++it) st4(lds + (it * 256 + threadIdx.x) * 4, bs[it]);
        __syncthreads();
        if (threadIdx.x < Q && n0 + threadIdx.x * 4 < a.NOUT) {
            float4 t = f4(0, 0, 0, 0);
            for (int j = 0; j < NT * 256 / Q; ++j) t = t + ld4(lds + (j * Q + threadIdx.x) * 4);
            float* o = a.dbias + n0 + threadIdx.x * 4;
            atomicAdd(o, t.x), atomicAdd(o + 1, t.y), atomicAdd(o + 2, t.z), atomicAdd(o + 3, t.w);
        }
        __syncthreads();
    }
    // cross-wave sum of the 4 row-partials, then one atomic per element.  Round 4: the partials travel through LDS as register dumps ([tile][register
    // quad][lane], one conflict-free ds_write_b128 / ds_read_b128 per quad) in a tree - wave 1 -> 0, wave 3 -> 2, wave 2 -> 0 - instead of four serial
    // passes of 128 dependent LDS read-modify-writes per lane (~65k cycles per workgroup: as long as its 32 MFMA stages, s_memtime build).
    constexpr int DUMP = NT * KT * 1024;
    static_assert(DUMP <= 2 * STAGE, "register dump aliases the stages");
    float* red = lds;
    auto dump = [&]() {
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int n = 0; n < KT; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) st4(red + (((m * KT + n) * 4 + g) * 64 + lane) * 4, acc_group(acc[m][n], g));
    };
    auto add = [&]() {
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int n = 0; n < KT; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = ld4(red + (((m * KT + n) * 4 + g) * 64 + lane) * 4);
                    acc[m][n][4 * g] += v.x, acc[m][n][4 * g + 1] += v.y, acc[m][n][4 * g + 2] += v.z, acc[m][n][4 * g + 3] += v.w;
                }
    };
    if (w == 1) dump();
    __syncthreads();
    if (w == 0) add();
    __syncthreads();
    if (w == 3) dump();
    __syncthreads();
    if (w == 2) add();
    __syncthreads();
    if (w == 2) dump();
    __syncthreads();
    if (w == 0) {
        add();
        dump();
    }
    __syncthreads();
    // dump element (tile t = m KT + n, register r, lane l) is dW[n0 + 32 m + (r & 3) + 8 (r >> 2) + 4 (l >> 5)][k0 + 32 n + (l & 31)]: consecutive threads
    // take consecutive lanes of one register - 128-byte lines per half wave
    float* dWz = a.dW + (size_t)z * a.KIN;
    for (int idx = threadIdx.x; idx < DUMP; idx += 256) {
        const int l = idx & 63, q = idx >> 6, comp = q & 3, g = (q >> 2) & 3, t = q >> 4;  // float4 slot (t, g) of lane l, component comp  ->  register 4 g + comp
        const int r = 4 * g + comp, m = t / KT, n = t % KT;
        const int nn = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), kk = 32 * n + (l & 31);
        if (n0 + nn < a.NOUT && k0 + kk < a.KIN) atomicAdd(dWz + (size_t)(n0 + nn) * a.ldw + k0 + kk, red[((t * 4 + g) * 64 + l) * 4 + comp]);
    }
#ifdef WG_TIMING
    if (lane == 0 && a.dbias) {  // (timing build: the segment sums of wave w of workgroup blockIdx.x go to dbias - repurposed as a dump buffer)
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.dbias) + (size_t)(blockIdx.x * 4 + w) * 8;
        for (int j = 0; j < 5; ++j) o[j] = tacc[j];
        o[5] = (unsigned long long)((r1 - r0 + CH - 1) / CH);
        o[6] = __builtin_amdgcn_s_memtime() - t_entry;
        o[7] = t_entry;
    }
#endif
#undef WT_MARK
}

// Toeplitz weight gradient, all 8 taps in one workgroup:  dW[n][z*64 + k] += sum_r dY[r][n] * X[xrow(r) + z][k],  z = 0..7  (KIN = 64).
// wgrad_kernel runs the taps as separate workgroups that each re-stage the same dY / X rows (24 KB of loads per 128 MFMAs);
// here one 32-row stage = dY chunk [32][64] + the X slab of those rows (32 + 7 rows per segment touched, at most two segments)
// feeds 512 MFMAs: wave w owns taps 2w, 2w+1 (2 x 2 x 2 accumulator tiles), so there is no cross-wave sum either.
// The row index is the MFMA contraction index, hence uniform per k-step: the slab row of chunk row i is i (+7 once the chunk
// has crossed into its second segment) + tap.  x_off shifts the window (conv-transpose: -7); rows outside [0, x_seg) are zero.
template <int P = 0>
__global__ __launch_bounds__(256, 2) void toeplitz_wgrad_kernel(WgradArgs a) {
    constexpr int LD = 68, CH = 32, XR = CH + 14, STAGE = (CH + XR) * LD;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    __shared__ int s_nA[2];
    const int nblocks = a.NOUT / 64;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int group = (local / nblocks) * 8 + xcd, nb = local % nblocks;
    if (group >= a.ngroups) return;
    const int n0 = nb * 64;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int r0 = group * a.rows_per_wg;
    const int r1 = r0 + a.rows_per_wg < a.M ? r0 + a.rows_per_wg : a.M;

    // staging: thread -> (row, quad).  dY: 32 rows x 16 quads = 2 per thread; X slab: 46 rows x 16 quads = 736 -> 3 per thread
    float4 yr[2], xr[3], bs[2];
    bs[0] = bs[1] = f4(0, 0, 0, 0);
    const bool want_bias = a.dbias != nullptr;
    auto load = [&](int rb) {
        // segment split of this chunk: rows [rb, rb+nA) lie in the first segment, the rest in the next one
        const int seq0 = rb / a.seg_len, l0 = rb - seq0 * a.seg_len;
        const int nA = min(CH, a.seg_len - l0);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = threadIdx.x + it * 256, lr = idx >> 4, q4 = (idx & 15) * 4;
            const int r = rb + lr;
            yr[it] = ld4(a.dY + (size_t)min(r, r1 - 1) * a.ldy + n0 + q4);
            if (r >= r1) yr[it] = f4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = threadIdx.x + it * 256, sr = idx >> 4, q4 = (idx & 15) * 4;  // slab row sr < 46
            // piece A: slab rows [0, nA+7) <-> (seq0, l0 + x_off + sr); piece B: slab rows [nA+7, ..) <-> (seq0+1, x_off + sr - nA - 7)
            const bool inB = sr >= nA + 7;
            const int seq = seq0 + (inB ? 1 : 0);
            const int xl = (inB ? sr - nA - 7 : l0 + sr) + a.x_off;
            const bool ok = sr < XR && xl >= 0 && xl < a.x_seg && (size_t)seq * a.seg_len < (size_t)a.M;
            const int xlc = min(max(xl, 0), a.x_seg - 1);
            const long long srow = (long long)min(seq, (a.M - 1) / a.seg_len) * a.x_seg + xlc;
            xr[it] = ld4(a.X + (size_t)srow * a.ldx + q4);
            if (!ok) xr[it] = f4(0, 0, 0, 0);
        }
        return nA;
    };
    auto store = [&](float* st, int nA, int slot) {
        float* Ys = st;
        float* Xs = st + CH * LD;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = threadIdx.x + it * 256, lr = idx >> 4, q4 = (idx & 15) * 4;
            st4(Ys + lr * LD + q4, yr[it]);
            bs[it] = bs[it] + yr[it];
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = threadIdx.x + it * 256, sr = idx >> 4, q4 = (idx & 15) * 4;
            if (sr < XR) st4(Xs + sr * LD + q4, xr[it]);
        }
        if (threadIdx.x == 0) s_nA[slot] = nA;
    };

    floatx16 acc[2][2][2];  // [tap of this wave][n tile][k tile]
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[z][m][n][r] = 0.f;
    int nA = load(r0);
    store(lds, nA, 0);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int rb = r0; rb < r1; rb += CH) {
        const bool more = rb + CH < r1;
        int nAn = 0;
        if (more) nAn = load(rb + CH);
        const float* Ys = lds + cur * STAGE;
        const float* Xs = Ys + CH * LD;
        const int nAc = s_nA[cur];
        if constexpr (P != 0) {
#pragma unroll
            for (int q = 0; q < CH; q += 16) {  // K = 16 per instruction: lane half kh supplies chunk rows q + 8 kh .. + 7
                float ya[2][8], xb[2][2][8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = q + kh * 8 + r;
                    const int srow = row + (row >= nAc ? 7 : 0) + 2 * w;
#pragma unroll
                    for (int m = 0; m < 2; ++m) ya[m][r] = Ys[row * LD + m * 32 + i];
#pragma unroll
                    for (int z = 0; z < 2; ++z)
#pragma unroll
                        for (int n = 0; n < 2; ++n) xb[z][n][r] = Xs[(srow + z) * LD + n * 32 + i];
                }
                Frag fa[2], fb[2][2];
#pragma unroll
                for (int m = 0; m < 2; ++m) fa[m] = frag_f32<P>(f4(ya[m][0], ya[m][1], ya[m][2], ya[m][3]), f4(ya[m][4], ya[m][5], ya[m][6], ya[m][7]));
#pragma unroll
                for (int z = 0; z < 2; ++z)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        fb[z][n] = frag_f32<P>(f4(xb[z][n][0], xb[z][n][1], xb[z][n][2], xb[z][n][3]), f4(xb[z][n][4], xb[z][n][5], xb[z][n][6], xb[z][n][7]));
#pragma unroll
                for (int z = 0; z < 2; ++z)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) mma32<P>(acc[z][m][n], fa[m], fb[z][n]);
            }
        } else
#pragma unroll
        for (int q = 0; q < CH; q += 8) {
            float av[2][4], bv[2][2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = q + kh * 4 + r;                    // chunk row = contraction index of this lane half
                const int srow = row + (row >= nAc ? 7 : 0) + 2 * w;  // slab row of tap 2w
#pragma unroll
                for (int m = 0; m < 2; ++m) av[m][r] = Ys[row * LD + m * 32 + i];
#pragma unroll
                for (int z = 0; z < 2; ++z)
#pragma unroll
                    for (int n = 0; n < 2; ++n) bv[z][n][r] = Xs[(srow + z) * LD + n * 32 + i];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int z = 0; z < 2; ++z)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[z][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][r], bv[z][n][r], acc[z][m][n], 0, 0, 0);
        }
        if (more) store(lds + (cur ^ 1) * STAGE, nAn, cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // each wave owns its tiles: straight to the accumulator (one request per 128-byte line per tile row)
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    atomicAdd(a.dW + (size_t)(n0 + m * 32 + acc_row(r)) * a.ldw + (2 * w + z) * 64 + n * 32 + i, acc[z][m][n][r]);
    if (want_bias) {
        __syncthreads();
        st4(lds + threadIdx.x * 4, bs[0]);
        st4(lds + 1024 + threadIdx.x * 4, bs[1]);
        __syncthreads();
        if (threadIdx.x < 64) {  // column c = quad (tid & 15) * 4 + e: thread t sums the slots whose quad == t / 4
            const int quad = threadIdx.x >> 2, e = threadIdx.x & 3;
            float t = 0.f;
            for (int j = 0; j < 32; ++j) t += lds[((j >> 4) * 256 + (j & 15) * 16 + quad) * 4 + e];
            atomicAdd(a.dbias + n0 + threadIdx.x, t);
        }
    }
}

// Prologue x tile forms that exist (round 6: the others were instantiated and never launched - tools/kernel_coverage.py): the gateway prologue (pro 1) feeds the
// projection's weight gradient (64 x 256: <2, 4>), PReLU (2) and ReLU(gLN) (3) the 256 x 256 maps (<4, 2>); everything else runs the plain prologue.
template <int NT, int KT>
static constexpr bool wgrad_has(int pro) { return pro == 0 || (NT == 2 && KT == 4 && pro == 1) || (NT == 4 && KT == 2 && (pro == 2 || pro == 3)); }

template <int NT, int KT, int P = 0>
static int wgrad_launch(const WgradArgs& a0, int pro, hipStream_t st) {
    if (!wgrad_has<NT, KT>(pro)) return RTFS_EINVAL;
    WgradArgs a = a0;
    const int nblk = ((a.NOUT + NT * 32 - 1) / (NT * 32)) * ((a.KIN + KT * 32 - 1) / (KT * 32)) * a.nshift;
    // ~2048 workgroups, but at least 1024 rows each: every workgroup ends with a cross-wave LDS sum and one atomic request per
    // line of its tile, and requests to one line are served serially (~27 ns) - a few hundred row groups keep that tail short
    // (round 4: ~512 workgroups = one resident round for the full-resolution maps - a workgroup's prologue + cross-wave sum + atomics cost as much as
    // ~9 of its 32-row stages, s_memtime build - the short SRU maps keep ~2048)
    const long long wg_target = a.M >= 500000 ? 512 : 2048;  // (round 6, in the step: 256 / 384 / 768 for the large maps measured the same within 0.2 ms)
    long long rpw = ((long long)a.M * nblk / wg_target + 31) / 32 * 32;
    a.rows_per_wg = (int)(rpw < 1024 ? 1024 : rpw);
    a.ngroups = (a.M + a.rows_per_wg - 1) / a.rows_per_wg;
    const dim3 grid((unsigned)((a.ngroups + 7) / 8 * 8 * nblk));
    if (pro == 0) hipLaunchKernelGGL((wgrad_kernel<NT, KT, 0, P>), grid, dim3(256), 0, st, a);
    if constexpr (wgrad_has<NT, KT>(1)) { if (pro == 1) hipLaunchKernelGGL((wgrad_kernel<NT, KT, 1, P>), grid, dim3(256), 0, st, a); }
    if constexpr (wgrad_has<NT, KT>(2)) { if (pro == 2) hipLaunchKernelGGL((wgrad_kernel<NT, KT, 2, P>), grid, dim3(256), 0, st, a); }
    if constexpr (wgrad_has<NT, KT>(3)) { if (pro == 3) hipLaunchKernelGGL((wgrad_kernel<NT, KT, 3, P>), grid, dim3(256), 0, st, a); }
    return RTFS_OK;
}

// ---- projection input gradient + gateway adjoint, fused ------------------------------------------------------------------------
// forward: G = prelu(u), u = s*gw + gb;  y0 = Wp . G + bp;  block output = ... + G.   Given dy0 [rows][64] and dx [rows][256] (the
// gradient that reaches G through the residual), this kernel forms dG = dx + dy0 . Wp in registers (same 64 -> 256 GEMM shape as
// resid_kernel: weights resident in VGPRs, 64-pixel tiles, accumulators transposed through LDS one 32-pixel half at a time) and
// applies the gateway adjoint on the way out:  ds (= or +=) dG*prelu'(u)*gw, the running d(a0) sum, and the three parameter
// reductions.  dG never exists in HBM: 2 GB less traffic per block than rtfs_gemm_rows(accumulate) + rtfs_gateway_bwd.
// ACCM: 0 none, 1 acc = ds, 2 acc += ds.   Wt: [256][64] (output channel major, k contiguous).
// NEXT (round 6, fp32): ds is the gradient of the PREVIOUS block's output, and the first thing that block's adjoint does with it is the residual conv's input
// gradient dE = ds . Wr^T (256 -> 64: rows_ws64_kernel<256>, a 1.06 GB read of ds at the start of every block's backward).  Here the finished ds rows of a 32-pixel
// half go back into the LDS tile they came through, and the four waves form dE for those rows - wave w its 16 output columns, the weight WrT [64][256] in 64 registers
// per lane, v_mfma_f32_16x16x4_f32 with the products in rows_ws64_kernel's order (same bits) - before the next half overwrites the tile: ds is not read again.
template <bool ACCUM, int ACCM, int P = 0, bool NEXT = false>  // P != 0: Wt host-PACKED (common.h)
__global__ __launch_bounds__(256, 2) void proj_gateway_bwd_kernel(const float* __restrict__ dy0, const float* __restrict__ Wt, const float* __restrict__ dx,
                                                                  const float* __restrict__ s_in, const float* __restrict__ gw,
                                                                  const float* __restrict__ gb, float slope, float* __restrict__ ds,
                                                                  float* __restrict__ acc_out, float* __restrict__ scr, int M, int tiles_per_wg,
                                                                  const float* __restrict__ WrT = nullptr, float* __restrict__ dE = nullptr) {
    static_assert(!NEXT || (P == 0 && !ACCUM && ACCM == 0), "the next block's input-gradient GEMM rides in the plain fp32 form only");
    constexpr int LDE = 68, LDO = 260;
    __shared__ __attribute__((aligned(16))) float Es[64 * LDE];
    __shared__ __attribute__((aligned(16))) float Ot[32 * LDO];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int cq = (threadIdx.x & 63) * 4, c4 = (threadIdx.x & 15) * 4;
    const float4 cgw = ld4(gw + cq), cgb = ld4(gb + cq);
    float4 wf[2][8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 8; ++q) wf[nt][q] = ld4(Wt + (size_t)(64 * w + 32 * nt + i) * 64 + 8 * q + 4 * kh);
    float4 aw = f4(0, 0, 0, 0), ab = f4(0, 0, 0, 0);
    float asl = 0.f;
    // NEXT: lane (j, kg) of wave w holds WrT[16 w + j][16 q + 4 kg .. + 3] (rows_ws64_kernel's weight layout)
    const int nj = lane & 15, nkg = lane >> 4;
    float4 wn[NEXT ? 16 : 1];
    if constexpr (NEXT) {
#pragma unroll
        for (int q = 0; q < 16; ++q) wn[q] = ld4(WrT + (size_t)(16 * w + nj) * kC + 16 * q + 4 * nkg);
    }
    const int tile0 = blockIdx.x * tiles_per_wg;
#pragma unroll 1
    for (int tl = 0; tl < tiles_per_wg; ++tl) {
        const int m0 = (tile0 + tl) * 64;
        if (m0 >= M) break;
        {
            float4 xa[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) xa[it] = ld4_off(dy0, ((unsigned)min(m0 + (int)(threadIdx.x >> 4) + it * 16, M - 1) * kH + c4) * 4u);
#pragma unroll
            for (int it = 0; it < 4; ++it) st4(Es + ((threadIdx.x >> 4) + it * 16) * LDE + c4, pack4<P>(xa[it]));
        }
        __syncthreads();
        floatx16 acc[2][2];
        acc_zero(acc);
        if constexpr (P != 0) {
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const Frag e0 = frag_lds<P>(ld4(Es + i * LDE + 16 * q2 + 4 * kh), ld4(Es + i * LDE + 16 * q2 + 8 + 4 * kh));
                const Frag e1 = frag_lds<P>(ld4(Es + (32 + i) * LDE + 16 * q2 + 4 * kh), ld4(Es + (32 + i) * LDE + 16 * q2 + 8 + 4 * kh));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const Frag wq = frag_lds<P>(wf[nt][2 * q2], wf[nt][2 * q2 + 1]);
                    mma32<P>(acc[nt][0], wq, e0);
                    mma32<P>(acc[nt][1], wq, e1);
                }
            }
        } else
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 e0 = ld4(Es + i * LDE + 8 * q + 4 * kh);
            const float4 e1 = ld4(Es + (32 + i) * LDE + 8 * q + 4 * kh);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].x, e0.x, acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].x, e1.x, acc[nt][1], 0, 0, 0);
                acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].y, e0.y, acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].y, e1.y, acc[nt][1], 0, 0, 0);
                acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].z, e0.z, acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].z, e1.z, acc[nt][1], 0, 0, 0);
                acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].w, e0.w, acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][q].w, e1.w, acc[nt][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            // two sub-passes of 4 pixel rows per thread keep the loaded operands at 16 float4
            const int prow = m0 + pt * 32 + (threadIdx.x >> 6);
            __syncthreads();  // pt = 0: every wave is done with Es / pt = 1: Ot of the previous half has been read
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    st4(Ot + i * LDO + 64 * w + 32 * nt + 8 * g + 4 * kh,
                        f4(acc[nt][pt][4 * g], acc[nt][pt][4 * g + 1], acc[nt][pt][4 * g + 2], acc[nt][pt][4 * g + 3]));
            __syncthreads();
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 sv[4], xv[4], old[4], ao[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned o = ((unsigned)min(prow + 4 * (4 * half + k), M - 1) * kC + cq) * 4u;
                    sv[k] = ld4_off(s_in, o), xv[k] = ld4_off(dx, o);
                    if (ACCUM) old[k] = ld4_off(ds, o);
                    if (ACCM == 2) ao[k] = ld4_off(acc_out, o);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = (threadIdx.x >> 6) + 4 * (4 * half + k), row = prow + 4 * (4 * half + k);
                    if (row < M) {
                        const float4 g = ld4(Ot + r * LDO + cq) + xv[k];
                        const float4 u = fma4(sv[k], cgw, cgb);
                        asl += (u.x > 0.f ? 0.f : g.x * u.x) + (u.y > 0.f ? 0.f : g.y * u.y) + (u.z > 0.f ? 0.f : g.z * u.z) + (u.w > 0.f ? 0.f : g.w * u.w);
                        const float4 du = f4(u.x > 0.f ? g.x : g.x * slope, u.y > 0.f ? g.y : g.y * slope, u.z > 0.f ? g.z : g.z * slope,
                                             u.w > 0.f ? g.w : g.w * slope);
                        aw = fma4(du, sv[k], aw);
                        ab = ab + du;
                        float4 d = du * cgw;
                        const unsigned o = ((unsigned)row * kC + cq) * 4u;
                        if (ACCM == 1) st4_off(acc_out, o, d);
                        if (ACCM == 2) st4_off(acc_out, o, d + ao[k]);
                        if (ACCUM) d = d + old[k];
                        st4_off(ds, o, d);
                        if constexpr (NEXT) st4(Ot + r * LDO + cq, d);  // (this thread's own slot of the tile: it read g from there above)
                    }
                }
            }
            if constexpr (NEXT) {
                __syncthreads();  // the half's 32 ds rows are in Ot (rows past M keep stale values: their outputs are not stored)
                const float* xp = Ot + nj * LDO + 4 * nkg;
                floatx4 na[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 e0 = ld4(xp + 16 * q), e1 = ld4(xp + 16 * LDO + 16 * q);
                    na[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[q].x, e0.x, na[0], 0, 0, 0);
                    na[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[q].x, e1.x, na[1], 0, 0, 0);
                    na[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[q].y, e0.y, na[0], 0, 0, 0);
                    na[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[q].y, e1.y, na[1], 0, 0, 0);
                    na[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[q].z, e0.z, na[0], 0, 0, 0);
                    na[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[q].z, e1.z, na[1], 0, 0, 0);
                    na[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[q].w, e0.w, na[0], 0, 0, 0);
                    na[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[q].w, e1.w, na[1], 0, 0, 0);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int row = m0 + pt * 32 + rt * 16 + nj;
                    if (row < M) st4(dE + (size_t)row * kH + 16 * w + 4 * nkg, f4(na[rt][0], na[rt][1], na[rt][2], na[rt][3]));
                }
            }
        }
    }
    // parameter gradients: threads (wave, cq) -> per channel over the 4 waves, one coalesced request per line into this
    // workgroup's copy of the spread scratch [dgw 256 | dgb 256 | dslope]
    __syncthreads();
    float* red = Ot;  // 2 x 1024 floats + 4
    st4(red + threadIdx.x * 4, aw);
    st4(red + 1024 + threadIdx.x * 4, ab);
    asl = wave_sum(asl);
    if (lane == 0) red[2048 + w] = asl;
    __syncthreads();
    float* mine = spread_copy(scr, blockIdx.x);
    const int c = threadIdx.x;
    atomicAdd(mine + c, red[c] + red[256 + c] + red[512 + c] + red[768 + c]);
    atomicAdd(mine + kC + c, red[1024 + c] + red[1280 + c] + red[1536 + c] + red[1792 + c]);
    if (c == 0) atomicAdd(mine + 2 * kC, red[2048] + red[2049] + red[2050] + red[2051]);
}

// ---- Toeplitz input gradients ---------------------------------------------------------------------------------------------
struct SeqMapB {
    int seq_div;
    long long stride_hi, stride_lo, pos_stride;
    int npos, L;
    __device__ __forceinline__ size_t base(int s) const { return (size_t)(s / seq_div) * stride_hi + (size_t)(s % seq_div) * stride_lo; }
};

// MODE 2 (conv-transpose input gradient): slab = dG rows m0..m0+70 of sequence s (G layout), K = 8*64, out dH[s][l][64]
// MODE 3 (fold, unfold-GEMM input gradient): slab = zero-padded dU0 rows (m0-7)..(m0+63), width 256, K = 8*256,
//         out dxn in G layout (plain store).
// Weights Wt: [64][K] k-contiguous.  64-row tile, 4 waves as 2x2 of 32x32, weights first (lanes = positions).
template <int MODE, int P = 0>  // P != 0: Wt host-PACKED, slab packed on store
__global__ __launch_bounds__(256) void toeplitz_bwd_kernel(SeqMapB map, const float* __restrict__ src, const float* __restrict__ Wt, float* __restrict__ dst) {
    constexpr int SW = MODE == 2 ? 64 : 256, LDSL = SW + 4, K = 8 * SW, BK = 64, LDB = BK + 4;
    constexpr int ROWS = 71;
    __shared__ __attribute__((aligned(16))) float slab[ROWS * LDSL];
    __shared__ __attribute__((aligned(16))) float Bs[2][64 * LDB];
    const int s = blockIdx.y, m0 = blockIdx.x * 64;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = w >> 1, wn = w & 1;
    const size_t sbase = map.base(s);
    ChunkRegs<64, BK> breg;
    breg.load(Wt, K, 0);
    for (int idx = threadIdx.x; idx < ROWS * (SW / 4); idx += 256) {
        const int row = idx / (SW / 4), c4 = (idx % (SW / 4)) * 4;
        float4 v = f4(0, 0, 0, 0);
        if (MODE == 2) {
            const int pos = m0 + row;
            v = ld4(src + sbase + (size_t)min(pos, map.npos - 1) * map.pos_stride + c4);
            if (pos >= map.npos) v = f4(0, 0, 0, 0);
        } else {
            const int l = m0 + row - 7;
            if (l >= 0 && l < map.L) v = ld4(src + ((size_t)s * map.L + l) * 256 + c4);
        }
        st4(slab + row * LDSL + c4, pack4<P>(v));
    }
    breg.store(Bs[0], LDB);
    __syncthreads();
    floatx16 acc[1][1];
    acc_zero(acc);
    constexpr int NK = K / BK;
#pragma unroll 1
    for (int kc = 0; kc < NK; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < NK) breg.load(Wt, K, (kc + 1) * BK);
        const int k0 = kc * BK, kk = k0 / SW, c0 = k0 % SW;
        mma_block_nt<P, 1, 1>(acc, Bs[cur] + wn * 32 * LDB, LDB, slab + (wm * 32 + kk) * LDSL + c0, LDSL, BK);
        if (kc + 1 < NK) breg.store(Bs[cur ^ 1], LDB);
        __syncthreads();
    }
    const int row = m0 + wm * 32 + (lane & 31);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = wn * 32 + 8 * g + 4 * (lane >> 5);
        const float4 v = acc_group(acc[0][0], g);
        if (MODE == 2) {
            if (row < map.L) st4(dst + ((size_t)s * map.L + row) * 64 + col, v);
        } else {
            if (row < map.npos) st4(dst + sbase + (size_t)row * map.pos_stride + col, v);
        }
    }
}

// Fold (input gradient of the unfold + layer-0 GEMM), K-chunked: dxn[pos][c] = sum_{k', n} dU0[pos + k' - 7][n] * Wt[c][k'*256 + n].
// toeplitz_bwd_kernel<3> keeps a 71-row x 256-wide slab of dU0 in LDS (107 KB with the weight stages -> ONE workgroup per CU, the
// MFMA pipe idles through every slab load).  Here the 256 columns are walked in four 64-wide chunks: slab chunk (64*NP+7) x 64 and the
// 64x64 weight stages fit 2-3 workgroups per CU, the next slab chunk / weight stage travel through registers under the MFMAs.
// Workgroup tile = 64*NP positions x 64 channels, 4 waves as (channel half) x (position half), weights first (lanes = positions).
template <int NP, int P = 0>  // P != 0: Wt host-PACKED, slab packed on store
__global__ __launch_bounds__(256, NP == 1 ? 3 : 2) void fold_gemm_bwd_kernel(SeqMapB map, const float* __restrict__ dU0, const float* __restrict__ Wt,
                                                               float* __restrict__ dst) {
    constexpr int TP = 64 * NP, ROWS = TP + 7, LDSL = 68, LDB = 68, SQ = ROWS * 16, SPT = (SQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) float slab[ROWS * LDSL];
    __shared__ __attribute__((aligned(16))) float Bs[2][64 * LDB];
    const int s = blockIdx.y, m0 = blockIdx.x * TP;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = w >> 1, wn = w & 1;
    const float* src = dU0 + (size_t)s * map.L * 256;
    float4 sreg[SPT];
    auto load_slab = [&](int nc) {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int idx = threadIdx.x + i * 256, row = idx >> 4, c4 = (idx & 15) * 4;
            const int l = m0 + row - 7;
            const bool ok = idx < SQ && l >= 0 && l < map.L;
            sreg[i] = ld4(src + (size_t)min(max(l, 0), map.L - 1) * 256 + nc * 64 + c4);
            if (!ok) sreg[i] = f4(0, 0, 0, 0);
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int idx = threadIdx.x + i * 256, row = idx >> 4, c4 = (idx & 15) * 4;
            if (idx < SQ) st4(slab + row * LDSL + c4, pack4<P>(sreg[i]));
        }
    };
    ChunkRegs<64, 64> breg;
    load_slab(0);
    breg.load(Wt, 2048, 0);
    store_slab();
    breg.store(Bs[0], LDB);
    __syncthreads();
    floatx16 acc[1][NP];
    acc_zero(acc);
    int cur = 0;
#pragma unroll 1
    for (int nc = 0; nc < 4; ++nc) {
        if (nc + 1 < 4) load_slab(nc + 1);  // in flight under the 8 taps of this chunk
#pragma unroll 1
        for (int kp = 0; kp < 8; ++kp) {
            const bool last = nc == 3 && kp == 7;
            if (!last) breg.load(Wt, 2048, (kp == 7 ? 0 : kp + 1) * 256 + (kp == 7 ? nc + 1 : nc) * 64);
            mma_block_nt<P, 1, NP>(acc, Bs[cur] + wm * 32 * LDB, LDB, slab + (wn * 32 * NP + kp) * LDSL, LDSL, 64);
            if (kp == 7 && nc + 1 < 4) {
                __syncthreads();  // every wave is done with this slab chunk
                store_slab();
            }
            if (!last) breg.store(Bs[cur ^ 1], LDB);
            __syncthreads();
            cur ^= 1;
        }
    }
    const size_t sbase = map.base(s);
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const int row = m0 + wn * 32 * NP + n * 32 + (lane & 31);
        if (row < map.npos) {
#pragma unroll
            for (int g = 0; g < 4; ++g) st4(dst + sbase + (size_t)row * map.pos_stride + wm * 32 + 8 * g + 4 * (lane >> 5), acc_group(acc[0][n], g));
        }
    }
}

}  // namespace rtfs

using namespace rtfs;

template <int P>
static int wgrad_impl(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, float* dbias, long long M, int seg_len, int x_seg, int x_off,
                      int nshift, int NOUT, int KIN, int pro, const float* p0, const float* p1, float slope, const double* stats, int rows_per_b,
                      void* stream) {
    if (M <= 0 || M >= (1ll << 31) || (NOUT & 31) || (KIN & 31) || pro < 0 || pro > 3 || nshift < 1) return RTFS_EINVAL;
    WgradArgs a;
    a.dY = dY, a.ldy = ldy, a.X = X, a.ldx = ldx, a.dW = dW, a.ldw = ldw, a.dbias = dbias, a.M = (int)M;
    a.seg_len = seg_len > 0 && seg_len < M ? seg_len : 0, a.x_seg = x_seg > 0 ? x_seg : (int)M, a.x_off = x_off, a.nshift = nshift;
    a.NOUT = NOUT, a.KIN = KIN;
    a.p0 = p0, a.p1 = p1, a.slope = slope, a.slot = stats, a.rows_per_b = rows_per_b > 0 ? rows_per_b : 1;
    a.inv_n = 1.0 / ((double)a.rows_per_b * KIN);
    hipStream_t st = (hipStream_t)stream;
    if (nshift == 8 && KIN == 64 && NOUT % 64 == 0 && pro == 0 && a.seg_len >= 32 && a.seg_len) {  // unfold / conv-transpose weights
        const int nblk = NOUT / 64;
        // ~384 workgroups (round 4: 1024 -> 512, -3 % alone).  Round 6: these launches run on the weight-gradient side stream under the bandwidth-bound stretch of the
        // adjoint chain, two 42 KB / 112-register workgroups per CU at 512; at 384 half the CUs carry one and the chain's kernels keep more of their waves - training step,
        // same box: 88.04 / 88.24 ms at 512, 87.39 / 87.41 at 384, 87.8-87.9 at 256 / 352, 88.0-88.2 at 416 / 448, 88.9 at 320
        long long rpw = ((long long)a.M * nblk / 384 + 31) / 32 * 32;
        a.rows_per_wg = (int)(rpw < 512 ? 512 : rpw);
        a.ngroups = (a.M + a.rows_per_wg - 1) / a.rows_per_wg;
        hipLaunchKernelGGL(toeplitz_wgrad_kernel<P>, dim3((unsigned)((a.ngroups + 7) / 8 * 8 * nblk)), dim3(256), 0, st, a);
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    }
    // the plain-shape kernel reads its operands as scalars from LDS and would split each of them in registers for the six-term mode: measured
    // slower than the fp32 MFMAs it replaces (825 vs 480 us on the projection weight gradient) - bf16x6 keeps it on the fp32 pipe
    constexpr int PP = P == 6 ? 0 : P;
    int rc;
    if (NOUT >= 128 && NOUT % 128 == 0) rc = wgrad_launch<4, 2, PP>(a, pro, st);
    else if (KIN >= 128) rc = wgrad_launch<2, 4, PP>(a, pro, st);
    else rc = wgrad_launch<2, 2, PP>(a, pro, st);
    if (rc != RTFS_OK) return rc;
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

template <int P>
static int proj_gateway_bwd_impl(const float* dy0, const float* WpT, const float* dx, const float* s, const float* gw, const float* gb, float slope, float* ds,
                                 int accumulate, float* acc, int acc_mode, float* dgw, float* dgb, float* dslope, long long rows, void* stream,
                                 const float* next_WrT = nullptr, float* next_dE = nullptr) {
    if (rows <= 0 || rows * 1024 >= (1ll << 32) || acc_mode < 0 || acc_mode > 2 || (acc_mode && (!acc || accumulate))) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    const int M = (int)rows, tiles = (M + 63) / 64, per = 16;
    const dim3 grid((tiles + per - 1) / per);
    hipStream_t st = (hipStream_t)stream;
#define PGB(A, MODE) hipLaunchKernelGGL((proj_gateway_bwd_kernel<A, MODE, P>), grid, dim3(256), 0, st, dy0, WpT, dx, s, gw, gb, slope, ds, acc, scr, M, per, nullptr, nullptr)
    if constexpr (P == 0) {
        if (next_WrT) {
            if (accumulate || acc_mode || !next_dE) return RTFS_EINVAL;
            hipLaunchKernelGGL((proj_gateway_bwd_kernel<false, 0, 0, true>), grid, dim3(256), 0, st, dy0, WpT, dx, s, gw, gb, slope, ds, acc, scr, M, per, next_WrT, next_dE);
            RTFS_LAUNCH_CHECK();
            return spread_finish(scr, SpreadOut{{dgw, dgb, dslope}, {kC, kC, 1}}, st);
        }
    }
    if (accumulate) PGB(true, 0);
    else if (acc_mode == 0) PGB(false, 0);
    else if (acc_mode == 1) PGB(false, 1);
    else PGB(false, 2);
#undef PGB
    RTFS_LAUNCH_CHECK();
    return spread_finish(scr, SpreadOut{{dgw, dgb, dslope}, {kC, kC, 1}}, st);
}

static SeqMapB make_map(int dim, int T2) {
    SeqMapB m;
    if (dim == 4) {
        m.seq_div = 1, m.stride_hi = (long long)kF2 * kH, m.stride_lo = 0, m.pos_stride = kH, m.npos = kF2;
    } else {
        m.seq_div = kF2, m.stride_hi = (long long)T2 * kF2 * kH, m.stride_lo = kH, m.pos_stride = (long long)kF2 * kH, m.npos = T2;
    }
    m.L = m.npos - 7;
    return m;
}

template <int P>
static int fold_impl(const float* dU0, const float* Wt, float* dxn, int B, int T2, int dim, void* stream) {
    if ((dim != 3 && dim != 4) || B <= 0 || T2 < 8) return RTFS_EINVAL;
    SeqMapB m = make_map(dim, T2);
    const int S = dim == 4 ? B * T2 : B * kF2;
    if (m.npos > 64)
        hipLaunchKernelGGL((fold_gemm_bwd_kernel<2, P>), dim3((m.npos + 127) / 128, S), dim3(256), 0, (hipStream_t)stream, m, dU0, Wt, dxn);
    else
        hipLaunchKernelGGL((fold_gemm_bwd_kernel<1, P>), dim3((m.npos + 63) / 64, S), dim3(256), 0, (hipStream_t)stream, m, dU0, Wt, dxn);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

template <int P>
static int convt_bwd_impl(const float* dG, const float* Wt, float* dH3, int B, int T2, int dim, void* stream, int variant = 0) {
    if ((dim != 3 && dim != 4) || B <= 0 || T2 < 8) return RTFS_EINVAL;
    SeqMapB m = make_map(dim, T2);
    const int S = dim == 4 ? B * T2 : B * kF2;
    if (P == 0 && variant == 0) {  // fp32, large batch: the weight-stationary fast-FIR kernel (dualpath.hip)
        const int rc = convt_bwd_input_ffa(dG, Wt, dH3, B, T2, dim, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    hipLaunchKernelGGL((toeplitz_bwd_kernel<2, P>), dim3((m.L + 63) / 64, S), dim3(256), 0, (hipStream_t)stream, m, dG, Wt, dH3);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

extern "C" {

// dW[n][z*KIN + k] (row stride ldw) += sum_r dY[r][n] * X'[xrow(r, z)][k] for the shifts z = 0..nshift-1;
// r = seq*seg_len + l, xrow = seq*x_seg + l + x_off + z (zero row if outside its segment).  nshift = 1 for the plain maps; the
// unfold / conv-transpose Toeplitz weight gradients are ONE launch with nshift = 8.
// dbias (optional): dbias[n] += sum_r dY[r][n].
// pro: 0 plain; 1 X' = prelu(X*p0+p1, slope) (gateway); 2 X' = prelu(X, slope); 3 X' = relu(gLN(X)) with stats slot / p0=gamma,p1=beta,
// rows_per_b rows per utterance (inv_n = 1/(rows_per_b*KIN)).  NOUT, KIN multiples of 32.
int rtfs_wgrad(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, float* dbias, long long M, int seg_len, int x_seg, int x_off, int nshift,
               int NOUT, int KIN, int pro, const float* p0, const float* p1, float slope, const double* stats, int rows_per_b, void* stream) {
    return wgrad_impl<0>(dY, ldy, X, ldx, dW, ldw, dbias, M, seg_len, x_seg, x_off, nshift, NOUT, KIN, pro, p0, p1, slope, stats, rows_per_b, stream);
}
// bf16 (terms 1) / split-bf16 (terms 3) products with fp32 accumulation; both operands are activations / gradients, packed inside the kernel
int rtfs_wgrad_bf16(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, float* dbias, long long M, int seg_len, int x_seg, int x_off,
                    int nshift, int NOUT, int KIN, int pro, const float* p0, const float* p1, float slope, const double* stats, int rows_per_b, int terms,
                    void* stream) {
    RTFS_TERMS_DISPATCH(terms, wgrad_impl<1>(dY, ldy, X, ldx, dW, ldw, dbias, M, seg_len, x_seg, x_off, nshift, NOUT, KIN, pro, p0, p1, slope, stats, rows_per_b, stream),
                        wgrad_impl<3>(dY, ldy, X, ldx, dW, ldw, dbias, M, seg_len, x_seg, x_off, nshift, NOUT, KIN, pro, p0, p1, slope, stats, rows_per_b, stream),
                        wgrad_impl<6>(dY, ldy, X, ldx, dW, ldw, dbias, M, seg_len, x_seg, x_off, nshift, NOUT, KIN, pro, p0, p1, slope, stats, rows_per_b, stream));
}

// ds (= or +=) gateway adjoint of (dx + dy0 . Wp); see proj_gateway_bwd_kernel.  rows * 1 KB must stay below 4 GB (32-bit byte offsets).
int rtfs_proj_gateway_bwd(const float* dy0, const float* WpT, const float* dx, const float* s, const float* gw, const float* gb, float slope, float* ds,
                          int accumulate, float* acc, int acc_mode, float* dgw, float* dgb, float* dslope, long long rows, void* stream) {
    return proj_gateway_bwd_impl<0>(dy0, WpT, dx, s, gw, gb, slope, ds, accumulate, acc, acc_mode, dgw, dgb, dslope, rows, stream);
}
// ... and the residual conv's input gradient of the block whose output gradient ds is (the next block of the backward pass): dE = ds . WrT^T, [rows][64]
// (= rtfs_gemm_rows(ds, WrT, .., 256, 64) on the stored ds, same bits); plain form only (no accumulate, no acc)
int rtfs_proj_gateway_bwd_next(const float* dy0, const float* WpT, const float* dx, const float* s, const float* gw, const float* gb, float slope, float* ds,
                               float* dgw, float* dgb, float* dslope, const float* next_WrT, float* next_dE, long long rows, void* stream) {
    if (!next_WrT || !next_dE) return RTFS_EINVAL;
    return proj_gateway_bwd_impl<0>(dy0, WpT, dx, s, gw, gb, slope, ds, 0, nullptr, 0, dgw, dgb, dslope, rows, stream, next_WrT, next_dE);
}
int rtfs_proj_gateway_bwd_bf16(const float* dy0, const void* WpT_pk, const float* dx, const float* s, const float* gw, const float* gb, float slope,
                               float* ds, int accumulate, float* acc, int acc_mode, float* dgw, float* dgb, float* dslope, long long rows, int terms,
                               void* stream) {
    const float* W = (const float*)WpT_pk;
    RTFS_TERMS_DISPATCH(terms, proj_gateway_bwd_impl<1>(dy0, W, dx, s, gw, gb, slope, ds, accumulate, acc, acc_mode, dgw, dgb, dslope, rows, stream),
                        proj_gateway_bwd_impl<3>(dy0, W, dx, s, gw, gb, slope, ds, accumulate, acc, acc_mode, dgw, dgb, dslope, rows, stream),
                        proj_gateway_bwd_impl<0>(dy0, W, dx, s, gw, gb, slope, ds, accumulate, acc, acc_mode, dgw, dgb, dslope, rows, stream));  // (terms 6: HBM-bound, the fp32 kernel - 1048 against 1140 us)
}

// dU0: [S][L][256] -> dxn in G layout [B][T2][F2][64] (plain store).  Wt: [64][2048], Wt[c][k'*256+n] = W0t[n][(7-k')*64+c]
int rtfs_fold_gemm_bwd(const float* dU0, const float* Wt, float* dxn, int B, int T2, int dim, void* stream) {
    return fold_impl<0>(dU0, Wt, dxn, B, T2, dim, stream);
}
int rtfs_fold_gemm_bwd_bf16(const float* dU0, const void* Wpk, float* dxn, int B, int T2, int dim, int terms, void* stream) {
    const float* W = (const float*)Wpk;
    // terms 6 (operands are plain fp32 in this mode): the fp32 kernel - the six-term form of this LDS-staged kernel splits every fragment it reads in registers and
    // measured 783-891 us against 558-585 (round 5, profiles/r05f_bf16x6_train_kernel_stats.txt); fp32-equivalent either way
    RTFS_TERMS_DISPATCH(terms, fold_impl<1>(dU0, W, dxn, B, T2, dim, stream), fold_impl<3>(dU0, W, dxn, B, T2, dim, stream), fold_impl<0>(dU0, W, dxn, B, T2, dim, stream));
}

// dG: G layout -> dH3 [S][L][64].  Wt: [64 j][512], Wt[j][k*64+c] = Wct[j][c][k]
int rtfs_convt_bwd_input(const float* dG, const float* Wt, float* dH3, int B, int T2, int dim, void* stream) {
    return convt_bwd_impl<0>(dG, Wt, dH3, B, T2, dim, stream);
}
// variant: 0 = the library's choice (the fast-FIR kernel at large batch), 1 = the direct 8-tap kernel
int rtfs_convt_bwd_input_form(const float* dG, const float* Wt, float* dH3, int B, int T2, int dim, int variant, void* stream) {
    if (variant < 0 || variant > 1) return RTFS_EINVAL;
    return convt_bwd_impl<0>(dG, Wt, dH3, B, T2, dim, stream, variant);
}
int rtfs_convt_bwd_input_bf16(const float* dG, const void* Wpk, float* dH3, int B, int T2, int dim, int terms, void* stream) {
    const float* W = (const float*)Wpk;
    // terms 6: the fp32 path (fast-FIR kernel at large batch: 121 us against 244 for the six-term LDS-staged form)
    RTFS_TERMS_DISPATCH(terms, convt_bwd_impl<1>(dG, W, dH3, B, T2, dim, stream), convt_bwd_impl<3>(dG, W, dH3, B, T2, dim, stream), convt_bwd_impl<0>(dG, W, dH3, B, T2, dim, stream));
}

}  // extern "C"
