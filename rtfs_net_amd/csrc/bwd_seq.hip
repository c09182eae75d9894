// Backward of the dual-path SRU stage (training step).  Forward lives in dualpath.hip; the recurrence and its adjoint
// follow oracle/sru_ref.py / SURVEY.md §8a ("Backward" paragraph).
//
//   rtfs_sru_scan_train_fwd   forward recurrence that also stores the cell state C[s][l][64] (needed by the adjoint)
//   rtfs_sru_scan_bwd         reverse-time adjoint: dH -> dU (same layout as U), dX (skip input, layers 1-3),
//                             weight_c / bias gradients accumulated with one atomic per lane per sequence
//   rtfs_sru_layer_bwd        layers 1-3: the recurrence adjoint with the weight and input gradients of the projection in the same launch (dU stays in LDS)
//   rtfs_ln4d_c_bwd           adjoint of LayerNormalization4D over the 64 channels of each position (normalizations.py:33-37)
//   rtfs_seq_gather           G layout -> sequence-major [S][npos][64] copy, raw or LN4D-normalised (operands of rtfs_wgrad)
#include "common.h"

namespace rtfs {

struct SeqMapS {
    int seq_div;
    long long stride_hi, stride_lo, pos_stride;
    int npos, L;
    __device__ __forceinline__ size_t base(int s) const { return (size_t)(s / seq_div) * stride_hi + (size_t)(s % seq_div) * stride_lo; }
};

// same lane mapping as sru_scan_kernel (dualpath.hip): one wave per sequence, lane = dir*32 + j
template <int KM>
__global__ __launch_bounds__(256) void sru_scan_train_kernel(const float* __restrict__ U, const float* __restrict__ X, const float* __restrict__ wc,
                                                             const float* __restrict__ bias, float scale_x, float* __restrict__ Hout,
                                                             float* __restrict__ Cout, int S, int L) {
    constexpr int UNR = 8;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int lane = threadIdx.x & 63;
    const bool rev = lane >= 32;
    const float wf = wc[lane], wr = wc[64 + lane], bf = bias[lane], br = bias[64 + lane];
    const float* u = U + (size_t)s * L * 64 * KM + (KM == 4 ? lane * 4 : lane);
    const float* x = X + (size_t)s * L * 64 + lane;
    float* h = Hout + (size_t)s * L * 64 + lane;
    float* cs = Cout + (size_t)s * L * 64 + lane;
    float c = 0.f;
    // two register sets: batch i + 1 is fetched before the recurrence of batch i (as in sru_scan_kernel, dualpath.hip)
    float va[UNR][4], vb[UNR][4];
    auto load = [&](int t0, float (&v)[UNR][4]) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const int t = min(t0 + i, L - 1);
            const int l = rev ? L - 1 - t : t;
            if (KM == 4) {
                const float4 q = ld4(u + (size_t)l * 256);
                v[i][0] = q.x, v[i][1] = q.y, v[i][2] = q.z, v[i][3] = q.w;
            } else {
                const float* p = u + (size_t)l * 192;
                v[i][0] = p[0], v[i][1] = p[64], v[i][2] = p[128];
                v[i][3] = x[(size_t)l * 64] * scale_x;
            }
        }
    };
    auto run = [&](int t0, const float (&v)[UNR][4]) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            const int t = t0 + i;
            if (t < L) {
                const int l = rev ? L - 1 - t : t;
                const float f = sigmoidf_fast(v[i][1] + bf + wf * c);
                const float r = sigmoidf_fast(v[i][2] + br + wr * c);
                c = v[i][0] + (c - v[i][0]) * f;
                cs[(size_t)l * 64] = c;
                h[(size_t)l * 64] = v[i][3] + (c - v[i][3]) * r;
            }
        }
    };
    load(0, va);
#pragma unroll 1
    for (int t0 = 0; t0 < L; t0 += 2 * UNR) {
        load(t0 + UNR, vb);
        run(t0, va);
        load(t0 + 2 * UNR, va);
        run(t0 + UNR, vb);
    }
}

// adjoint, walking the forward processing order backwards and carrying dc:
//   gh = dH_t; dx' = gh(1-r); dr = gh(c_t - x'); du2 = dr r(1-r); dct = gh r + dc; du0 = dct(1-f); df = dct(c_prev - u0);
//   du1 = df f(1-f); dc <- dct f + du1 wf + du2 wr; dwf += du1 c_prev; dwr += du2 c_prev; dbf += du1; dbr += du2
template <int KM>
__global__ __launch_bounds__(256) void sru_scan_bwd_kernel(const float* __restrict__ U, const float* __restrict__ X, const float* __restrict__ Cst,
                                                           const float* __restrict__ wc, const float* __restrict__ bias, float scale_x,
                                                           const float* __restrict__ dH, const float* __restrict__ dH2, float* __restrict__ dU,
                                                           float* __restrict__ dX, float* __restrict__ scr, int S, int L) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int lane = threadIdx.x & 63;
    const bool rev = lane >= 32;
    const float wf = wc[lane], wr = wc[64 + lane], bf = bias[lane], br = bias[64 + lane];
    const float* u = U + (size_t)s * L * 64 * KM + (KM == 4 ? lane * 4 : lane);
    float* du = dU + (size_t)s * L * 64 * KM + (KM == 4 ? lane * 4 : lane);
    const size_t o64 = (size_t)s * L * 64 + lane;
    float dc = 0.f, awf = 0.f, awr = 0.f, abf = 0.f, abr = 0.f;
    // the loads of a step do not depend on the carried state: 8 steps are fetched (clamped, unconditional) ahead of the dependent
    // chain, as in the forward scan (a per-step load pays one memory latency per step)
    constexpr int UNR = 8;
#pragma unroll 1
    for (int i0 = 0; i0 < L; i0 += UNR) {
        float cpv[UNR], cv[UNR], ghv[UNR], u0v[UNR], u1v[UNR], u2v[UNR], xpv[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            // forward processed l_k = rev ? L-1-k : k for k = 0..L-1; the adjoint visits k = L-1-i
            const int k = max(L - 1 - (i0 + j), 0);
            const int l = rev ? L - 1 - k : k;
            const int lp = min(max(rev ? l + 1 : l - 1, 0), L - 1);  // position processed just before l (masked below when k == 0)
            cpv[j] = Cst[o64 + (size_t)lp * 64];
            cv[j] = Cst[o64 + (size_t)l * 64];
            ghv[j] = dH[o64 + (size_t)l * 64];
            if (dH2) ghv[j] += dH2[o64 + (size_t)l * 64];  // the gradient arrives as the two per-direction parts of rtfs_sru_layer_bwd
            if (KM == 4) {
                const float4 v = ld4(u + (size_t)l * 256);
                u0v[j] = v.x, u1v[j] = v.y, u2v[j] = v.z, xpv[j] = v.w;
            } else {
                const float* p = u + (size_t)l * 192;
                u0v[j] = p[0], u1v[j] = p[64], u2v[j] = p[128];
                xpv[j] = X[o64 + (size_t)l * 64] * scale_x;
            }
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int i = i0 + j;
            if (i < L) {  // wave-uniform
                const int k = L - 1 - i;
                const int l = rev ? L - 1 - k : k;
                const float cprev = k > 0 ? cpv[j] : 0.f;
                const float c = cv[j], gh = ghv[j], u0 = u0v[j], u1 = u1v[j], u2 = u2v[j], xp = xpv[j];
                const float f = sigmoidf_fast(u1 + bf + wf * cprev);
                const float r = sigmoidf_fast(u2 + br + wr * cprev);
                const float dxp = gh * (1.f - r);
                const float dr = gh * (c - xp);
                const float du2 = dr * r * (1.f - r);
                const float dct = gh * r + dc;
                const float du0 = dct * (1.f - f);
                const float df = dct * (cprev - u0);
                const float du1 = df * f * (1.f - f);
                dc = dct * f + du1 * wf + du2 * wr;
                awf = fmaf(du1, cprev, awf);
                awr = fmaf(du2, cprev, awr);
                abf += du1;
                abr += du2;
                if (KM == 4) {
                    st4(du + (size_t)l * 256, f4(du0, du1, du2, dxp));
                } else {
                    float* q = du + (size_t)l * 192;
                    q[0] = du0, q[64] = du1, q[128] = du2;
                    dX[o64 + (size_t)l * 64] = dxp * scale_x;
                }
            }
        }
    }
    float* mine = spread_copy(scr, blockIdx.x);  // [dwc 128 | dbias 128]
    atomicAdd(mine + lane, awf);
    atomicAdd(mine + 64 + lane, awr);
    atomicAdd(mine + 128 + lane, abf);
    atomicAdd(mine + 192 + lane, abr);
}

// ---- adjoint of one SRU layer 1-3 in ONE launch: recurrence adjoint + weight gradient + input gradient ----
// The three-launch form (sru_scan_bwd_kernel<3>, wgrad_kernel, rows_ws64_kernel) writes dU [S L][192] once and reads it twice: 642 + 233 + 233 MB
// per layer at 32 utterances.  Here dU lives 8 steps at a time in LDS.  A wave owns ONE DIRECTION OF TWO SEQUENCES (lane = 32 * which sequence +
// hidden unit), runs 8 steps of the adjoint recurrence of both, parks dU (transposed: [sequence][step][gate * 32 + unit]) and the highway term in its
// LDS slab, then on the fp32 MFMA (16 x 16 x 4):
//   dW[n][k]      += sum_(seq, step) dU[seq, step][n] X[seq, t(step)][k]    24 tiles (this direction's 96 rows n), K = 16 (sequence, step) pairs
//   dXd[seq, t][k] = sum_n dU[seq, t][n] W[n][k] (+ highway on the direction's own 32 columns k)        4 tiles of 16 (sequence, step) rows, K = 96
// The two directions reach a time t in different iterations, so each writes its own buffer (dX0: forward direction, dX1: backward direction) and the
// consumer adds them - the next layer down takes the pair as dH, dH2.  (A one-buffer form - one wave per sequence, the later direction adding to the
// earlier one's rows behind a fence - measured 250 us per layer: one wave per SIMD, every load and LDS round trip bare.)  Two waves per SIMD here: one's
// recurrence and memory waits run under the other's MFMAs.  dW stays in 96 accumulator registers over all pairs of the wave, is summed over the four
// waves of a direction in LDS and leaves as the workgroup's partial in `work` (sru_layer_bwd_reduce_kernel adds the partials); dwc, dbias: spread scratch.
// HBM per layer: U, X, C, dH (x 2) in, dX0 + dX1 out = 523 MB (three launches: 1108 MB); 11.2 GFLOP.
constexpr int kLbTS = 97;                         // floats per step row of a sequence's dU slab
constexpr int kLbTD = 8 * kLbTS + 24;             // floats per sequence (32 mod 64: the two sequences' scan writes fall 32 banks apart)
constexpr int kLbPS = 36;                         // floats per row of the highway slab [16 (sequence, step)][32]
constexpr int kLbWave = 2 * kLbTD + 16 * kLbPS;   // 2176 floats per wave
constexpr int kLbOS = 68;                         // floats per row of the output tile staged in the dU slab [16][64]
static_assert(16 * kLbOS <= 2 * kLbTD, "the output tile is staged in the dU slab");
#ifndef LB_ABL
#define LB_ABL 0  // timing-only builds (tools/ffa_ablate.sh lb <mask>): 1 no dW MFMAs, 2 no dX MFMAs, 4 no recurrence arithmetic, 8 no dX stores, 16 no operand loads
#endif

// dU column (gate * 32 + unit) of k-slice g of MFMA q of the dX tiles: the 16 rows x 4 slices of one A read fall on 64 different banks
__device__ __forceinline__ int lb_col(int q, int g) { return 32 * (q >> 3) + (q & 7) + 8 * g; }

__global__ __launch_bounds__(512) void sru_layer_bwd_kernel(const float* __restrict__ U, const float* __restrict__ X, const float* __restrict__ Cst,
                                                            const float* __restrict__ W, const float* __restrict__ wc, const float* __restrict__ bias,
                                                            float scale_x, const float* __restrict__ dH, const float* __restrict__ dH2,
                                                            float* __restrict__ dX0, float* __restrict__ dX1, float* __restrict__ work,
                                                            float* __restrict__ scr, int S, int L) {
#ifdef LB_TIMING  // timing-only build (tools/ffa_ablate.sh lb with ABL_FLAGS=-DLB_TIMING, tools/sru_bwd_bench.py): s_memtime ticks per phase, per wave, into dX0
    unsigned long long lb_tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lb_last = __builtin_amdgcn_s_memtime();
#define LB_TICK(k)                                                      \
    {                                                                   \
        const unsigned long long lb_now = __builtin_amdgcn_s_memtime(); \
        lb_tm[k] += lb_now - lb_last;                                   \
        lb_last = lb_now;                                               \
    }
#else
#define LB_TICK(k)
#endif
    __shared__ float WsB[2 * 24 * 4 * 64];  // the B operands of the dX tiles in issue order: [direction][q][column tile][lane]
    __shared__ float Tw[8][kLbWave];
    for (int e = threadIdx.x; e < 2 * 24 * 4 * 64; e += 512) {
        const int ln = e & 63, ct = (e >> 6) & 3, q = (e >> 8) % 24, d = e / (24 * 256);
        const int col = lb_col(q, ln >> 4), n = (col >> 5) * 64 + d * 32 + (col & 31);
        WsB[e] = W[n * 64 + 16 * ct + (ln & 15)];
    }
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d = wv & 1;  // the wave's direction: iteration i of the adjoint visits position t = d ? i : L - 1 - i
    const int lane = threadIdx.x & 63, g = lane >> 4, j16 = lane & 15, hs = lane >> 5, un = lane & 31;
    const float wf = wc[d * 32 + un], wr = wc[64 + d * 32 + un], bf = bias[d * 32 + un], br = bias[64 + d * 32 + un];
    float* Ts = Tw[wv];
    float* Ps = Ts + 2 * kLbTD;
    float* tw = Ts + hs * kLbTD + un;  // this lane's column of its sequence's slab
    float* dXd = d ? dX1 : dX0;
    floatx4 accW[6][4];
#pragma unroll
    for (int rt = 0; rt < 6; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) accW[rt][ct] = floatx4{0.f, 0.f, 0.f, 0.f};
    float awf = 0.f, awr = 0.f, abf = 0.f, abr = 0.f;
    const int npair = (S + 1) >> 1, pstep = gridDim.x * 4;
    int p = blockIdx.x * 4 + (wv >> 1);
    if (p < npair) {
        // scan operands of 8 adjoint iterations i0 .. i0 + 7 of the pair pn (clamped, unconditional)
        float pu0[8], pu1[8], pu2[8], pc[8], pgh[8], pg2[8], pxp[8], pcn;
        // operands come through buffer loads: descriptor (scalar) + per-lane byte offset of the sequence and column (one register for the whole pair) +
        // scalar byte offset of the row of iteration i - no vector address arithmetic per load (the wave issues 57 - 73 loads per chunk)
        const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, (int)((long long)S * L * 768), 0x00020000);
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)((long long)S * L * 256), 0x00020000);
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Cst), 0, (int)((long long)S * L * 256), 0x00020000);
        const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dH), 0, (int)((long long)S * L * 256), 0x00020000);
        const __amdgpu_buffer_rsrc_t rG2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dH2 ? dH2 : dH), 0, (int)((long long)S * L * 256), 0x00020000);
        auto bld = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff) {
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
        };
        auto load = [&](int pn, int i0) {
            const unsigned sq = min(2 * pn + hs, S - 1), v64 = (sq * (unsigned)L * 64u + d * 32 + un) * 4u, v192 = (sq * (unsigned)L * 192u + d * 32 + un) * 4u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = min(i0 + j, L - 1), t = d ? i : L - 1 - i;  // uniform
                pc[j] = bld(rC, v64, t * 256);
                pgh[j] = bld(rG, v64, t * 256);
                pg2[j] = dH2 ? bld(rG2, v64, t * 256) : 0.f;
                pxp[j] = bld(rX, v64, t * 256);
                pu0[j] = bld(rU, v192, t * 768), pu1[j] = bld(rU, v192 + 256, t * 768), pu2[j] = bld(rU, v192 + 512, t * 768);
            }
            const int i = min(i0 + 8, L - 1);
            pcn = bld(rC, v64, (d ? i : L - 1 - i) * 256);
        };
        if (!(LB_ABL & 16)) load(p, 0);
        float dc = 0.f;
#pragma unroll 1
        for (int i0 = 0;;) {
            LB_TICK(0);
            const bool live = 2 * p + hs < S;  // the second sequence of the last pair of an odd S is a clamped copy: nothing of it is kept
            // rows of X of this chunk as B operands of the dW tiles: k = 4 q + g = 8 (which sequence) + step
            float xa[4][4];
            {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sq = min(2 * p + (q >> 1), S - 1);  // uniform
                    const int i = min(i0 + 4 * (q & 1) + g, L - 1);
                    const unsigned vo = ((d ? i : L - 1 - i) * 64u + j16) * 4u;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) xa[q][ct] = bld(rX, vo + 64 * ct, sq * L * 256);
                }
            }
            if (i0 == 0) dc = 0.f;
            // ---- 8 steps of the adjoint recurrence (sru_scan_bwd_kernel's arithmetic); no branch per step: a step past the end of the sequence (clamped
            // operands, only behind the last real step) runs with gh = 0 and dct = 0, all its outputs are zero ----
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j;
                float du0 = 0.f, du1 = 0.f, du2 = 0.f, dxp = 0.f;
                if (!(LB_ABL & 4)) {
                    const bool ok = live && i < L;
                    const float cprev = i < L - 1 ? (j < 7 ? pc[j + 1] : pcn) : 0.f;
                    const float c = pc[j], gh = ok ? pgh[j] + pg2[j] : 0.f, u0 = pu0[j], u1 = pu1[j], u2 = pu2[j], xp = pxp[j] * scale_x;
                    const float f = sigmoidf_fast(u1 + bf + wf * cprev);
                    const float r = sigmoidf_fast(u2 + br + wr * cprev);
                    dxp = gh * (1.f - r);
                    const float dr = gh * (c - xp);
                    du2 = dr * r * (1.f - r);
                    const float dct = ok ? gh * r + dc : 0.f;
                    du0 = dct * (1.f - f);
                    const float df = dct * (cprev - u0);
                    du1 = df * f * (1.f - f);
                    dc = dct * f + du1 * wf + du2 * wr;
                    awf = fmaf(du1, cprev, awf);
                    awr = fmaf(du2, cprev, awr);
                    abf += du1;
                    abr += du2;
                }
                tw[j * kLbTS] = du0, tw[j * kLbTS + 32] = du1, tw[j * kLbTS + 64] = du2;
                Ps[(8 * hs + j) * kLbPS + un] = dxp * scale_x;
            }
            LB_TICK(1);
            // ---- operands of the next chunk (of this pair, or the first of the wave's next pair): in flight under the MFMAs ----
            const bool last = i0 + 8 >= L;
            const int pn = last ? p + pstep : p, in = last ? 0 : i0 + 8;
            if (!(LB_ABL & 16)) load(pn < npair ? pn : p, in);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            LB_TICK(2);
            // ---- dW: 4 groups (4 (sequence, step) pairs) of 24 MFMAs; the six A operands of a group are read from LDS one group ahead ----
            {
                float aw[2][6];
                const float* tr = Ts + g * kLbTS + j16;
#pragma unroll
                for (int rt = 0; rt < 6; ++rt) aw[0][rt] = tr[16 * rt];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q + 1 < 4) {
#pragma unroll
                        for (int rt = 0; rt < 6; ++rt) aw[(q + 1) & 1][rt] = tr[((q + 1) >> 1) * kLbTD + 4 * ((q + 1) & 1) * kLbTS + 16 * rt];
                    }
                    if (!(LB_ABL & 1)) {
#pragma unroll
                        for (int rt = 0; rt < 6; ++rt)
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) accW[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[q & 1][rt], xa[q][ct], accW[rt][ct], 0, 0, 0);
                    }
                }
            }
            LB_TICK(3);
            // ---- dX: 12 groups of two k-steps (2 A + 8 B operands, 8 MFMAs), operands read one group ahead ----
            {
                floatx4 ax[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) ax[ct] = floatx4{0.f, 0.f, 0.f, 0.f};
                float oa[2][2], ob[2][2][4];
                const float* ta = Ts + (j16 >> 3) * kLbTD + (j16 & 7) * kLbTS + 8 * g;
                const float* wb = WsB + d * 24 * 256 + lane;
                auto fetch = [&](int gx, int slot) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int q = 2 * gx + h;
                        oa[slot][h] = ta[32 * (q >> 3) + (q & 7)];
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) ob[slot][h][ct] = wb[(q * 4 + ct) * 64];
                    }
                };
                fetch(0, 0);
#pragma unroll
                for (int gx = 0; gx < 12; ++gx) {
                    if (gx + 1 < 12) fetch(gx + 1, (gx + 1) & 1);
                    if (LB_ABL & 2) continue;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) ax[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[gx & 1][h], ob[gx & 1][h][ct], ax[ct], 0, 0, 0);
                }
                LB_TICK(4);
                // the tile (rows 4 g + r = sequence g >> 1 of the pair, iteration i0 + 4 (g & 1) + r; + highway on the direction's own columns) goes
                // through the dU slab, which the MFMAs above have finished reading, and leaves as whole 256-byte rows: 4 x 16-byte stores per lane
                // instead of 16 x 4-byte stores on 64-byte pieces
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        float v = ax[ct][r];
                        if ((ct >> 1) == d) v += Ps[(4 * g + r) * kLbPS + 16 * (ct & 1) + j16];
                        Ts[(4 * g + r) * kLbOS + 16 * ct + j16] = v;
                    }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = 4 * k + g;  // (sequence row >> 3, step row & 7)
                    const unsigned sq = 2 * p + (row >> 3);
                    const int i = i0 + (row & 7);
                    const bool ok = sq < (unsigned)S && i < L && !(LB_ABL & 8);
                    const float4 v = ld4(Ts + row * kLbOS + 4 * j16);
                    if (ok) st4(dXd + ((size_t)sq * L + (d ? i : L - 1 - i)) * 64 + 4 * j16, v);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            LB_TICK(5);
            if (last) {
                p += pstep;
                i0 = 0;
                if (p >= npair) break;
            } else {
                i0 += 8;
            }
        }
    }
    LB_TICK(6);
    // dW of the four waves of a direction is summed in LDS (the slabs are free now; [direction][register][lane], one wave after the other) and leaves as
    // this workgroup's partial in `work` [workgroup][direction][register][lane] - plain stores; sru_layer_bwd_reduce_kernel adds the partials into dW.
    // (Through atomics into the spread scratch - 24 per thread - this tail was 54 k of the wave's 387 k cycles.)
    float* mine = spread_copy(scr, blockIdx.x);  // [dwc 128 | dbias 128]
    float* red = &Tw[0][0] + d * 96 * 64 + lane;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        __syncthreads();
        if ((wv >> 1) == k) {
#pragma unroll
            for (int rt = 0; rt < 6; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* q = red + ((rt * 4 + ct) * 4 + r) * 64;
                        *q = k ? *q + accW[rt][ct][r] : accW[rt][ct][r];
                    }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x * 4; e < 2 * 96 * 64; e += 2048) st4(work + (size_t)blockIdx.x * 12288 + e, ld4(&Tw[0][0] + e));
#ifdef LB_TIMING
    lb_tm[7] += __builtin_amdgcn_s_memtime() - lb_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(dX0) + (blockIdx.x * 8 + wv) * 8;
        for (int k = 0; k < 8; ++k) o[k] = lb_tm[k];
    }
#endif
    atomicAdd(mine + d * 32 + un, awf);
    atomicAdd(mine + 64 + d * 32 + un, awr);
    atomicAdd(mine + 128 + d * 32 + un, abf);
    atomicAdd(mine + 192 + d * 32 + un, abr);
}

// dW[n][k] += sum over the workgroups' partials (work [nwg][direction][register (row tile, column tile, r)][lane]); grid (12, 32), a thread owns four
// consecutive elements: blockIdx.y takes every 32nd partial (8 independent 16-byte loads), 32 atomics per element
__global__ __launch_bounds__(256) void sru_layer_bwd_reduce_kernel(const float* __restrict__ work, float* __restrict__ dW, int nwg) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    float4 t = f4(0, 0, 0, 0);
#pragma unroll 8
    for (int w = blockIdx.y; w < nwg; w += 32) t = t + ld4(work + (size_t)w * 12288 + e);
    const int ln = e & 63, reg = (e >> 6) % 96, dd = e / (96 * 64);
    const int r = reg & 3, ct = (reg >> 2) & 3, rt = reg >> 4;
    const int col = 16 * rt + 4 * (ln >> 4) + r, n = (col >> 5) * 64 + dd * 32 + (col & 31);
    float* o = dW + n * 64 + 16 * ct + (ln & 15);
    atomicAdd(o, t.x), atomicAdd(o + 1, t.y), atomicAdd(o + 2, t.z), atomicAdd(o + 3, t.w);
}

// LN4D over channels, adjoint.  dxn, G, dG in G layout [rows][64]; dG += dx; dgamma/dbeta += per-channel sums.
__global__ __launch_bounds__(256) void ln4d_c_bwd_kernel(const float* __restrict__ dxn, const float* __restrict__ G, const float* __restrict__ gamma,
                                                         float* __restrict__ dG, float* __restrict__ scr, long long rows, int rows_per_wg) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    const int c4 = (threadIdx.x & 15) * 4, rsub = threadIdx.x >> 4;
    const float4 g4 = ld4(gamma + c4);
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    float4 ag = f4(0, 0, 0, 0), ab = f4(0, 0, 0, 0);
    for (long long rb = r0; rb < r1; rb += 16) {
        const long long r = rb + rsub;
        const bool ok = r < r1;
        const size_t o = (size_t)(ok ? r : r0) * 64 + c4;
        const float4 x = ld4(G + o);
        float4 g = ld4(dxn + o);
        if (!ok) g = f4(0, 0, 0, 0);
        float sum = x.x + x.y + x.z + x.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) sum += __shfl_xor(sum, m, 64);
        const float mean = sum * (1.f / 64.f);
        const float4 d = f4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
        float sq = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) sq += __shfl_xor(sq, m, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.f / 64.f) + kEps);
        const float4 xh = d * rstd;
        const float4 a = g * g4;
        float s1 = a.x + a.y + a.z + a.w, s2 = a.x * xh.x + a.y * xh.y + a.z * xh.z + a.w * xh.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
        }
        s1 *= (1.f / 64.f), s2 *= (1.f / 64.f);
        if (ok) {
            const float4 dx = f4((a.x - s1 - xh.x * s2) * rstd, (a.y - s1 - xh.y * s2) * rstd, (a.z - s1 - xh.z * s2) * rstd, (a.w - s1 - xh.w * s2) * rstd);
            st4(dG + o, dx + ld4(dG + o));
        }
        ag = fma4(g, xh, ag);
        ab = ab + g;
    }
    // threads (rsub, quad) -> per channel, one coalesced atomic request per array into this workgroup's scratch copy [dgamma 64 | dbeta 64]
    float* mine = spread_copy(scr, blockIdx.x);
    st4(lds + threadIdx.x * 4, ag);
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) t += lds[r * 64 + threadIdx.x];
        atomicAdd(mine + threadIdx.x, t);
    }
    __syncthreads();
    st4(lds + threadIdx.x * 4, ab);
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) t += lds[r * 64 + threadIdx.x];
        atomicAdd(mine + 64 + threadIdx.x, t);
    }
}

// out[s][pos][64] = (LN ? LN4D_c(G[map(s,pos)]) : G[map(s,pos)])
template <bool LN>
__global__ __launch_bounds__(256) void seq_gather_kernel(SeqMapS map, const float* __restrict__ G, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ out, int S) {
    const long long idx = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c4 = (threadIdx.x & 15) * 4;
    const bool ok = idx < (long long)S * map.npos;
    const int s = ok ? (int)(idx / map.npos) : 0, pos = ok ? (int)(idx % map.npos) : 0;
    float4 v = ld4(G + map.base(s) + (size_t)pos * map.pos_stride + c4);
    if (LN) {
        float sum = v.x + v.y + v.z + v.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) sum += __shfl_xor(sum, m, 64);
        const float mean = sum * (1.f / 64.f);
        const float4 d = f4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
        float sq = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) sq += __shfl_xor(sq, m, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.f / 64.f) + kEps);
        v = fma4(d * rstd, ld4(gamma + c4), ld4(beta + c4));
    }
    if (ok) st4(out + (size_t)idx * 64 + c4, v);
}

}  // namespace rtfs

using namespace rtfs;

static SeqMapS make_map(int dim, int T2) {
    SeqMapS m;
    if (dim == 4) {
        m.seq_div = 1, m.stride_hi = (long long)kF2 * kH, m.stride_lo = 0, m.pos_stride = kH, m.npos = kF2;
    } else {
        m.seq_div = kF2, m.stride_hi = (long long)T2 * kF2 * kH, m.stride_lo = kH, m.pos_stride = (long long)kF2 * kH, m.npos = T2;
    }
    m.L = m.npos - 7;
    return m;
}

extern "C" {

int rtfs_sru_scan_train_fwd(const float* U, const float* X, const float* wc, const float* bias, float scale_x, float* H, float* C, int S, int L,
                            int km, void* stream) {
    if (S <= 0 || L <= 0) return RTFS_EINVAL;
    dim3 grid((S + 3) / 4);
    if (km == 4)
        hipLaunchKernelGGL((sru_scan_train_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, U, X, wc, bias, scale_x, H, C, S, L);
    else if (km == 3)
        hipLaunchKernelGGL((sru_scan_train_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, U, X, wc, bias, scale_x, H, C, S, L);
    else
        return RTFS_EINVAL;
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// dwc, dbias: [2][64] accumulated into.  km = 3 also writes dX [S][L][64] (gradient w.r.t. the skip input, already * scale_x).
// dH2 (nullable): a second part of the incoming gradient, added to dH on load.
int rtfs_sru_scan_bwd2(const float* U, const float* X, const float* C, const float* wc, const float* bias, float scale_x, const float* dH,
                       const float* dH2, float* dU, float* dX, float* dwc, float* dbias, int S, int L, int km, void* stream) {
    if (S <= 0 || L <= 0) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    dim3 grid((S + 3) / 4);
    if (km == 4)
        hipLaunchKernelGGL((sru_scan_bwd_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, U, X, C, wc, bias, scale_x, dH, dH2, dU, dX, scr, S, L);
    else if (km == 3)
        hipLaunchKernelGGL((sru_scan_bwd_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, U, X, C, wc, bias, scale_x, dH, dH2, dU, dX, scr, S, L);
    else
        return RTFS_EINVAL;
    RTFS_LAUNCH_CHECK();
    return spread_finish(scr, SpreadOut{{dwc, dbias}, {128, 128}}, (hipStream_t)stream);
}

int rtfs_sru_scan_bwd(const float* U, const float* X, const float* C, const float* wc, const float* bias, float scale_x, const float* dH, float* dU,
                      float* dX, float* dwc, float* dbias, int S, int L, int km, void* stream) {
    return rtfs_sru_scan_bwd2(U, X, C, wc, bias, scale_x, dH, nullptr, dU, dX, dwc, dbias, S, L, km, stream);
}

// The adjoint of one fused SRU layer (rtfs_sru_layer_fwd in training mode) in one launch: W [192][64] as the forward takes it.  The incoming gradient
// is dH (+ dH2 when not null); the gradient w.r.t. the layer input (recurrence skip term + dU . W) leaves as TWO parts, dX0 + dX1 [S][L][64] each (the
// forward and the backward direction's share; both fully written) - hand them to the next layer down as dH, dH2.  dW [192][64], dwc, dbias [2][64]
// are accumulated into.  work: rtfs_sru_layer_bwd_work_floats(S) floats of device scratch (the workgroups' partial dW).
static int sru_layer_bwd_nwg(int S) { return min(((S + 1) / 2 + 3) / 4, 256); }

int rtfs_sru_layer_bwd_work_floats(int S) { return S > 0 ? sru_layer_bwd_nwg(S) * 12288 : 0; }

int rtfs_sru_layer_bwd(const float* U, const float* X, const float* C, const float* W, const float* wc, const float* bias, float scale_x, const float* dH,
                       const float* dH2, float* dX0, float* dX1, float* work, float* dW, float* dwc, float* dbias, int S, int L, void* stream) {
    if (S <= 0 || L <= 0 || (long long)S * L * 768 >= (1ll << 31)) return RTFS_EINVAL;  // buffer descriptors: byte sizes in an int
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    const int nwg = sru_layer_bwd_nwg(S);
    hipLaunchKernelGGL(sru_layer_bwd_kernel, dim3(nwg), dim3(512), 0, (hipStream_t)stream, U, X, C, W, wc, bias, scale_x, dH, dH2, dX0, dX1, work, scr, S,
                       L);
    RTFS_LAUNCH_CHECK();
    hipLaunchKernelGGL(sru_layer_bwd_reduce_kernel, dim3(12, 32), dim3(256), 0, (hipStream_t)stream, work, dW, nwg);
    RTFS_LAUNCH_CHECK();
    return spread_finish(scr, SpreadOut{{dwc, dbias}, {128, 128}}, (hipStream_t)stream);
}

int rtfs_ln4d_c_bwd(const float* dxn, const float* G, const float* gamma, float* dG, float* dgamma, float* dbeta, long long rows, void* stream) {
    if (rows <= 0) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    const int per = 256;
    hipLaunchKernelGGL(ln4d_c_bwd_kernel, dim3((unsigned)((rows + per - 1) / per)), dim3(256), 0, (hipStream_t)stream, dxn, G, gamma, dG, scr, rows, per);
    RTFS_LAUNCH_CHECK();
    return spread_finish(scr, SpreadOut{{dgamma, dbeta}, {64, 64}}, (hipStream_t)stream);
}

// ln != 0: LN4D-normalise (gamma, beta) while gathering.  out: [S][npos][64]
int rtfs_seq_gather(const float* G, const float* gamma, const float* beta, int ln, float* out, int B, int T2, int dim, void* stream) {
    if ((dim != 3 && dim != 4) || B <= 0) return RTFS_EINVAL;
    SeqMapS m = make_map(dim, T2);
    const int S = dim == 4 ? B * T2 : B * kF2;
    const long long n = (long long)S * m.npos;
    dim3 grid((unsigned)((n + 15) / 16));
    if (ln)
        hipLaunchKernelGGL(seq_gather_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, out, S);
    else
        hipLaunchKernelGGL(seq_gather_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, out, S);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
