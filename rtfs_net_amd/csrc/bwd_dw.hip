// rtfs_dw_adjoint: the WHOLE adjoint of up to four stride-1 depth-wise 4x4 convolutions that share one input, in ONE pass (training step, round 6).
//
// Forward (conv_layers.py:65-129 with groups = channels; tdanet.py:61-76, fusion.py:25-52):  y_k = gLN_k(conv_k(in'))  with  in' = in | gLN(in) | PReLU(gLN(in)),
//     conv_k(in')[t][f] = bias_k + sum_{dt,df} w_k[dt*4+df] * in'[t-1+dt][f-1+df]          ('same' padding of an even kernel: 1 before, 2 after).
// Until round 6 the adjoint of one such convolution was three launches - rtfs_gln_bwd_apply (dN, X -> dX), rtfs_dwconv_bwd_weight (dX, in -> dW),
// rtfs_dwconv_bwd_input (dX -> dIn, read-modify-write when several convolutions share the input) - and dX went through HBM three times.  With
//     W[r][c] = dX[ti-2+r][fi-2+c]   (the 4x4 neighbourhood of dX around an INPUT pixel (ti, fi))
// both gradients are sums over the same sixteen values:
//     dIn'[ti][fi]        = sum_{r,c} w[(3-r)*4 + (3-c)] * W[r][c]
//     dW[(3-r)*4 + (3-c)] += in'[ti][fi] * W[r][c]            (summed over every input pixel; dX = 0 outside the tensor)
//     dbias               += W[2][2]
// so one kernel stages the dX tile of convolution k in LDS - the gLN adjoint  dX = rstd (gamma dN - S1/N - xhat S2/N)  applied on the way in, i.e. dX never
// exists in HBM - walks it once for both sums, moves on to convolution k+1 with the dIn' accumulators still in registers, and stores dIn' once.
// HBM traffic in tensor units, group of the four global convolutions of the TFAR fusion layers (input G3): 31 -> 10; concat layer's two: 15 -> 6; single
// convolutions 7 -> 4 (gLN'd) and 4-5 -> 3-4 (direct dOut).
//
// Workgroup = 8 input rows x 8 columns per step x 64 channels, 256 threads.  Staging: thread = (pixel, channel quad), 16-byte loads, 11 x 11 pixel tile
// (two halo rows / columns before, one after).  Window pass: thread = (row, channel PAIR) - with two channels per lane the 16 weight-gradient partials of
// one convolution are 32 registers, so four convolutions' partials (128) stay resident next to the accumulators; taps are read from LDS (broadcast).
// Consecutive time tiles of an utterance run on the same XCD (the 1-D grid is re-mapped), so the halo rows a neighbour has just fetched are L2 hits.
#include "common.h"
#include "intdiv.h"

namespace rtfs {

constexpr int kAdjMax = 4;
struct DwAdjArgs {
    int T, F, B, nt, nseg, fseg;
    const float* dy[kAdjMax];     // GLN: gradient w.r.t. the NORMALISED output of convolution k; else w.r.t. its output itself
    const float* x[kAdjMax];      // GLN: the convolution's pre-norm output
    const double* slot[kAdjMax];  // GLN: forward (sum, sum of squares) of x per utterance
    const double* red[kAdjMax];   // GLN: adjoint sums (S1, S2) per utterance (rtfs_gln_bwd_reduce / rtfs_mix_gln_bwd / rtfs_d0_tail_bwd)
    const float* gamma[kAdjMax];  // GLN
    const float* w[kAdjMax];      // taps [16][64], tap = dt*4 + df
    const float* in;              // the common input [B][T][F][64]
    const double* in_slot;        // mode >= 1
    const float *in_gamma, *in_beta;
    float in_slope;               // mode 2
    int mode, accumulate, bias;
    double inv_n;                 // 1 / (T F 64): every tensor here has the convolution's size
    float* dIn;
    float* scr;                   // spread scratch: per convolution [dW 1024 | dbias 64 when bias]
    // MIX (one convolution, GLN): the convolution is the LOCAL branch of an InjectionMultiSum (fusion.py:54-69: out = gLN(loc) * sigmoid(gLN(gate))^ + gLN(glob)^),
    // dy[0] is the gradient w.r.t. the mix's OUTPUT and the gradient w.r.t. the normalised local branch, dy * sigmoid(gLN(gate))[nearest(t), nearest(f)], is formed
    // on load (what rtfs_mix_gln_bwd's apply pass wrote to HBM as dLoc until round 6); the sigmoid itself comes from the reduce pass of rtfs_mix_gln_bwd
    const float* gate_s;          // [B][Tg][Fg][64]: sigmoid(gLN(gate)), written by rtfs_mix_gln_bwd's reduce pass (which forms it anyway)
    int Tg, Fg;
    // mode 3: the common input is itself a TFAR mix  gLN(in) * sigmoid(gLN(mgate))^ + gLN(mglob)^  of three tensors (the concat layer reads the fusion layers'
    // outputs F0 / F1): re-formed per pixel as the forward's rtfs_dwconv_mix_fwd does - F0 / F1 are never materialised, not even for the step
    const float *mgate, *mglob;   // [B][mTg][mFg][64], pre-norm
    const double *mgate_slot, *mglob_slot;
    const float *mgate_gamma, *mgate_beta, *mglob_gamma, *mglob_beta;
    double m_inv_n;
    int mTg, mFg;
    unsigned mt, mf;              // ceil(2^32 / T), ceil(2^32 / F): nearest source index floor(i * in / out) without a division
};



typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v ld2v(const float* p) { return *reinterpret_cast<const float2v*>(p); }

template <int NCONV, bool GLN, bool MIX = false>
__global__ __launch_bounds__(256, (NCONV == 1 ? 4 : 2)) void dw_adjoint_kernel(DwAdjArgs a) {
    static_assert(!MIX || (NCONV == 1 && GLN), "the mix prologue belongs to one gLN'd convolution");
    constexpr int TR = 8, TC = 8, R = TR + 3, CB = TC + 3, RS = CB * 64;  // tile rows t0-2 .. t0+8, columns fb-2 .. fb+8
    constexpr int NIT = (R * CB * 16 + 255) / 256;                          // staging items (pixel, quad) per thread
    __shared__ __attribute__((aligned(16))) float tile[R * RS];
    __shared__ __attribute__((aligned(16))) float ws[NCONV][16 * 64];
    __shared__ __attribute__((aligned(16))) float coefA[GLN ? NCONV : 1][64];
    __shared__ int nearest[MIX ? 2 : 1][R > CB ? R : CB];  // nearest source row / column (in units of 64-float pixels) of the tile's rows / columns
    __shared__ int mcol[TC];                               // mode 3: nearest source column of the block's 8 input columns
    __shared__ __attribute__((aligned(16))) float mixc[4][64];  // mode 3: gLN of the mix's gate / global tensors folded (scale, shift)
    // ---- tile of this workgroup: consecutive time tiles of an utterance on one XCD (workgroups go to XCDs round-robin by linear index) ----
    const int ntiles = a.nt * a.B * a.nseg, per_xcd = (ntiles + 7) / 8;
    const int vid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (vid >= ntiles) return;
    const int tt = vid % a.nt, b = (vid / a.nt) % a.B, seg = vid / (a.nt * a.B);
    const int T = a.T, F = a.F;
    const int t0 = tt * TR, f0 = seg * a.fseg, f1 = min(F, f0 + a.fseg);
    for (int i = threadIdx.x; i < NCONV * 256; i += 256) st4(&ws[i >> 8][(i & 255) * 4], ld4(a.w[i >> 8] + (i & 255) * 4));
    float Bc[NCONV], Cc[NCONV];
    if (GLN) {
#pragma unroll
        for (int k = 0; k < NCONV; ++k) {
            float mean, rstd;
            stats_finalize(a.slot[k], b, a.inv_n, mean, rstd);
            const float m1 = (float)(a.red[k][kStatStride * b] * a.inv_n), m2 = (float)(a.red[k][kStatStride * b + 1] * a.inv_n);
            Bc[k] = m2 * rstd * rstd;
            Cc[k] = Bc[k] * mean - m1 * rstd;
            if (threadIdx.x < 64) coefA[k][threadIdx.x] = a.gamma[k][threadIdx.x] * rstd;
        }
    }
    if (MIX && threadIdx.x < R) nearest[0][threadIdx.x] = div_magic((unsigned)min(max(t0 - 2 + (int)threadIdx.x, 0), T - 1) * a.Tg, T, a.mt) * a.Fg;
    // ---- window-pass identity: (row, channel pair) ----
    const int r = threadIdx.x >> 5, ch = (threadIdx.x & 31) * 2;
    const int ti = t0 + r;
    const bool tvalid = ti < T;
    float2v isc = float2v{1.f, 1.f}, ish = float2v{0.f, 0.f};
    if (a.mode >= 1) {
        float mean, rstd;
        stats_finalize(a.in_slot, b, a.inv_n, mean, rstd);
        const float2v g = ld2v(a.in_gamma + ch), be = ld2v(a.in_beta + ch);
        isc = g * rstd;
        ish = be - isc * mean;
    }
    const float am1 = a.in_slope - 1.0f;
    const float *mgrow = nullptr, *merow = nullptr;
    if (NCONV <= 2 && a.mode == 3) {  // (the four-convolution form has no registers to spare and no caller with a mixed input)
        float gm, gr, em, er;
        stats_finalize(a.mgate_slot, b, a.m_inv_n, gm, gr);
        stats_finalize(a.mglob_slot, b, a.m_inv_n, em, er);
        if (threadIdx.x < 64) {
            const float gs = a.mgate_gamma[threadIdx.x] * gr, es = a.mglob_gamma[threadIdx.x] * er;
            mixc[0][threadIdx.x] = gs, mixc[1][threadIdx.x] = a.mgate_beta[threadIdx.x] - gm * gs;
            mixc[2][threadIdx.x] = es, mixc[3][threadIdx.x] = a.mglob_beta[threadIdx.x] - em * es;
        }
        const size_t lrow = ((size_t)b * a.mTg + div_magic((unsigned)(tvalid ? ti : 0) * a.mTg, T, a.mt)) * a.mFg * kH + ch;
        mgrow = a.mgate + lrow, merow = a.mglob + lrow;
    }
    const size_t ubase = (size_t)b * T * F * kH;
    const float* inrow = a.in + ubase + (size_t)(tvalid ? ti : 0) * F * kH + ch;
    float* outrow = a.dIn + ubase + (size_t)(tvalid ? ti : 0) * F * kH + ch;
    // ---- staging identity: (pixel, channel quad) ----
    const int q4 = (threadIdx.x & 15) * 4;
    float2v part[NCONV][16], pb[NCONV];
#pragma unroll
    for (int k = 0; k < NCONV; ++k) {
        pb[k] = float2v{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) part[k][i] = float2v{0.f, 0.f};
    }
#pragma unroll 1
    for (int fb = f0; fb < f1; fb += TC) {
        // the input pixels of this thread's row (transformed as the forward convolution saw them; zero outside the tensor)
        float2v inv[TC], acc[TC];
#pragma unroll
        for (int j = 0; j < TC; ++j) inv[j] = ld2v(inrow + (size_t)min(fb + j, F - 1) * kH);
        if (NCONV <= 2 && a.mode == 3) {  // (workgroup-uniform)
            __syncthreads();  // the previous block's readers of mcol are done
            if (threadIdx.x < TC) mcol[threadIdx.x] = div_magic((unsigned)min(fb + (int)threadIdx.x, F - 1) * a.mFg, F, a.mf) * kH;
            __syncthreads();
            int moff = ch;
            asm volatile("" : "+v"(moff));  // (the six per-channel constants are re-read from LDS per block instead of living in 12 registers)
            const float2v gs = ld2v(&mixc[0][moff]), gh = ld2v(&mixc[1][moff]), es = ld2v(&mixc[2][moff]), eh = ld2v(&mixc[3][moff]);
#pragma unroll
            for (int j0 = 0; j0 < TC; j0 += 4) {
                float2v vg[4], ve[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) vg[j] = ld2v(mgrow + mcol[j0 + j]), ve[j] = ld2v(merow + mcol[j0 + j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2v g = vg[j] * gs + gh;
                    const float2v sg = float2v{sigmoidf_fast(g.x), sigmoidf_fast(g.y)};
                    inv[j0 + j] = (inv[j0 + j] * isc + ish) * sg + (ve[j] * es + eh);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            float2v v = inv[j];
            if (a.mode == 1 || a.mode == 2) v = v * isc + ish;
            if (a.mode == 2) v = __builtin_elementwise_min(v, float2v{0.f, 0.f}) * am1 + v;  // prelu(x) = x + (slope - 1) min(x, 0)
            inv[j] = (tvalid && fb + j < f1) ? v : float2v{0.f, 0.f};
            acc[j] = float2v{0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < NCONV; ++k) {
            __syncthreads();  // the previous tile's window reads are done (and ws / coefA are written)
            // ---- stage dX_k: rows t0-2 .. t0+8, columns fb-2 .. fb+8; the gLN adjoint on the way in; zero outside the tensor ----
            const float* dyb = a.dy[k] + ubase;
            const float* xb = GLN ? a.x[k] + ubase : nullptr;
            const float* gb = MIX ? a.gate_s + (size_t)b * a.Tg * a.Fg * kH : nullptr;
            if (MIX) {  // (the previous block's staging reads of the column table are behind the barrier above)
                if (threadIdx.x < CB) nearest[1][threadIdx.x] = div_magic((unsigned)min(max(fb - 2 + (int)threadIdx.x, 0), F - 1) * a.Fg, F, a.mf);
                __syncthreads();
            }
            float4 A4 = f4(1, 1, 1, 1);
            if (GLN) A4 = ld4(&coefA[k][q4]);
            constexpr int NG = (NCONV >= 4 || MIX) ? 2 : 4;  // loads in flight per thread and tensor (four convolutions' partial sums leave room for two; the mix
                                                            // form reads three tensors and stays at 128 registers = four workgroups per CU with two)
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));  // opaque per stage: the items' tile coordinates are recomputed here (a dozen integer instructions each) instead of
                                           // living in ~30 registers across the window passes
#pragma unroll
            for (int h = 0; h < NIT; h += NG) {
                float4 vd[NG], vx[NG], vg[MIX ? NG : 1];
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const int item = tid + (h + i) * 256, px = min(item >> 4, R * CB - 1), pr = px / CB, pc = px - pr * CB;
                    const int tq = min(max(t0 - 2 + pr, 0), T - 1), fq = min(max(fb - 2 + pc, 0), F - 1);
                    const unsigned off = (((unsigned)tq * F + fq) * kH + q4) * 4u;
                    vd[i] = ld4_off(dyb, off);
                    if (GLN) vx[i] = ld4_off(xb, off);
                    if (MIX) vg[i] = ld4_off(gb, ((unsigned)(nearest[0][pr] + nearest[1][pc]) * kH + q4) * 4u);
                }
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const int item = tid + (h + i) * 256, px = item >> 4, pr = px / CB, pc = px - pr * CB;
                    const int tq = t0 - 2 + pr, fq = fb - 2 + pc;
                    float4 d = vd[i];
                    if (MIX) d = d * vg[i];
                    if (GLN)
                        d = f4(A4.x * d.x - Bc[k] * vx[i].x + Cc[k], A4.y * d.y - Bc[k] * vx[i].y + Cc[k], A4.z * d.z - Bc[k] * vx[i].z + Cc[k],
                               A4.w * d.w - Bc[k] * vx[i].w + Cc[k]);
                    if (!(tq >= 0 && tq < T && fq >= 0 && fq < F)) d = f4(0, 0, 0, 0);
                    if (px < R * CB) st4(tile + px * 64 + q4, d);
                }
            }
            __syncthreads();
            // ---- window pass: 8 columns in two groups of four; per tap row the 7 window columns and the 4 taps are read once.  Every tap row sits in its OWN basic
            // block (a branch on a scalar the compiler cannot see through, always taken): as one block, hipcc schedules the 88 window / tap reads of a convolution
            // first and the arithmetic behind them, and spills; a rolled loop would make the partial sums' register indices dynamic.
            const float* trow0 = tile + r * RS + ch;
            int always = 1;
            asm volatile("" : "+s"(always));
#pragma unroll
            for (int jb = 0; jb < TC; jb += 4) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) if (always) {
                    const float* trow = trow0 + dt * RS;
                    float2v wr[7];
#pragma unroll
                    for (int c = 0; c < 7; ++c) wr[c] = ld2v(trow + (jb + c) * 64);
                    if (dt == 2) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)  // dX at the thread's own pixels (zero where they lie outside the tensor; columns past f1 are the next segment's)
                            if (fb + jb + jj < f1) pb[k] += wr[jj + 2];
                    }
#pragma unroll
                    for (int dc = 0; dc < 4; ++dc) {
                        const int tap = (3 - dt) * 4 + (3 - dc);
                        const float2v w = ld2v(&ws[k][tap * 64 + ch]);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            acc[jb + jj] = w * wr[jj + dc] + acc[jb + jj];
                            part[k][tap] = inv[jb + jj] * wr[jj + dc] + part[k][tap];
                        }
                    }
                    asm volatile("" : "+s"(always));
                }
            }
        }
        if (tvalid) {
#pragma unroll
            for (int j = 0; j < TC; ++j)
                if (fb + j < f1) {
                    float* o = outrow + (size_t)(fb + j) * kH;
                    float2v v = acc[j];
                    if (a.accumulate) v += ld2v(o);
                    *reinterpret_cast<float2v*>(o) = v;
                }
        }
    }
    // ---- tap / bias gradients: the two rows of a wave by one xor shuffle, the four waves through LDS, one coalesced atomic per (tap, channel) ----
    float* mine = spread_copy(a.scr, blockIdx.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* redl = tile;  // [4 waves][17][64]
#pragma unroll
    for (int k = 0; k < NCONV; ++k) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            float2v v = i < 16 ? part[k][i] : pb[k];
            v.x += __shfl_xor(v.x, 32, 64);
            v.y += __shfl_xor(v.y, 32, 64);
            if (lane < 32) *reinterpret_cast<float2v*>(redl + (wave * 17 + i) * 64 + ch) = v;
        }
        __syncthreads();
        const int per = 1024 + (a.bias ? 64 : 0);
        for (int idx = threadIdx.x; idx < per; idx += 256)
            atomicAdd(mine + k * per + idx, redl[idx] + redl[17 * 64 + idx] + redl[2 * 17 * 64 + idx] + redl[3 * 17 * 64 + idx]);
    }
}


// Two or four gLN'd convolutions with a fresh dIn (the concat layer's global embedding + gate, input = the TFAR mix F1; the four global convolutions of the fusion
// layers, input G3 = the attention's output) - ONE channel per lane.  In the two-channels-per-lane kernel above four convolutions' partial sums are 128 registers:
// 256 registers with spills, two workgroups per CU, 2.6 TB/s of algorithmic bytes against 4.3 for one convolution at four workgroups per CU.  With one channel per
// lane (a wave = one tile row x 64 channels, 512 threads = 8 rows) they are 64, the kernel fits 128 registers and sixteen waves per CU: the staging latency of one
// workgroup hides under the window passes of the others again.  Scalar v_fma_f32 / ds_read_b32 instead of packed / 64-bit ones: twice the instructions for the
// same arithmetic, which this HBM-bound kernel has room for.  Input raw (mode 0) or re-formed from a TFAR mix (mode 3).
template <int NCONV, bool MIX3>
__global__ __launch_bounds__(512, 2) void dw_adjoint1c_kernel(DwAdjArgs a) {
    constexpr int TR = 8, TC = 8, R = TR + 3, CB = TC + 3, RS = CB * 64, NT = 512;
    constexpr int NIT = (R * CB * 16 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float tile[R * RS];
    __shared__ __attribute__((aligned(16))) float ws[NCONV][16 * 64];
    __shared__ __attribute__((aligned(16))) float coefA[NCONV][64];
    __shared__ int mcol[TC];
    __shared__ float mixc[6][64];  // mode 3: gLN of the mix's local / gate / global tensors folded (scale, shift each)
    const int ntiles = a.nt * a.B * a.nseg, per_xcd = (ntiles + 7) / 8;
    const int vid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (vid >= ntiles) return;
    const int tt = vid % a.nt, b = (vid / a.nt) % a.B, seg = vid / (a.nt * a.B);
    const int T = a.T, F = a.F;
    const int t0 = tt * TR, f0 = seg * a.fseg, f1 = min(F, f0 + a.fseg);
    for (int i = threadIdx.x; i < NCONV * 256; i += NT) st4(&ws[i >> 8][(i & 255) * 4], ld4(a.w[i >> 8] + (i & 255) * 4));
    float Bc[NCONV], Cc[NCONV];
#pragma unroll
    for (int k = 0; k < NCONV; ++k) {
        float mean, rstd;
        stats_finalize(a.slot[k], b, a.inv_n, mean, rstd);
        const float m1 = (float)(a.red[k][kStatStride * b] * a.inv_n), m2 = (float)(a.red[k][kStatStride * b + 1] * a.inv_n);
        // (workgroup-uniform: kept in scalar registers - the kernel has to fit 128 vector registers)
        Bc[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m2 * rstd * rstd)));
        Cc[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, Bc[k] * mean - m1 * rstd)));
        if (threadIdx.x < 64) coefA[k][threadIdx.x] = a.gamma[k][threadIdx.x] * rstd;
    }
    const int r = threadIdx.x >> 6, ch = threadIdx.x & 63;  // window pass: wave = tile row, lane = channel
    const int ti = t0 + r;
    const bool tvalid = ti < T;
    const size_t ubase = (size_t)b * T * F * kH;
    const float* inrow = a.in + ubase + (size_t)(tvalid ? ti : 0) * F * kH + ch;
    float* outrow = a.dIn + ubase + (size_t)(tvalid ? ti : 0) * F * kH + ch;
    const float *mgrow = nullptr, *merow = nullptr;
    if (MIX3) {
        float lm, lr, gm, gr, em, er;
        stats_finalize(a.in_slot, b, a.inv_n, lm, lr);
        stats_finalize(a.mgate_slot, b, a.m_inv_n, gm, gr);
        stats_finalize(a.mglob_slot, b, a.m_inv_n, em, er);
        if (threadIdx.x < 64) {
            const float ls = a.in_gamma[threadIdx.x] * lr, gs = a.mgate_gamma[threadIdx.x] * gr, es = a.mglob_gamma[threadIdx.x] * er;
            mixc[0][threadIdx.x] = ls, mixc[1][threadIdx.x] = a.in_beta[threadIdx.x] - lm * ls;
            mixc[2][threadIdx.x] = gs, mixc[3][threadIdx.x] = a.mgate_beta[threadIdx.x] - gm * gs;
            mixc[4][threadIdx.x] = es, mixc[5][threadIdx.x] = a.mglob_beta[threadIdx.x] - em * es;
        }
        const size_t lrow = ((size_t)b * a.mTg + div_magic((unsigned)(tvalid ? ti : 0) * a.mTg, T, a.mt)) * a.mFg * kH + ch;
        mgrow = a.mgate + lrow, merow = a.mglob + lrow;
    }
    const int q4 = (threadIdx.x & 15) * 4;  // staging: (pixel, channel quad)
    float part[NCONV][16];
#pragma unroll
    for (int k = 0; k < NCONV; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) part[k][i] = 0.f;
#pragma unroll 1
    for (int fb = f0; fb < f1; fb += TC) {
        float inv[TC], acc[TC];
#pragma unroll
        for (int j = 0; j < TC; ++j) inv[j] = inrow[(size_t)min(fb + j, F - 1) * kH];
        if (MIX3) {  // the input pixel = gLN(l) * sigmoid(gLN(gate))^ + gLN(glob)^, as rtfs_dwconv_mix_fwd forms it
            __syncthreads();
            if (threadIdx.x < TC) mcol[threadIdx.x] = div_magic((unsigned)min(fb + (int)threadIdx.x, F - 1) * a.mFg, F, a.mf) * kH;
            __syncthreads();
            int moff = ch;
            asm volatile("" : "+v"(moff));
            const float ls = mixc[0][moff], lh = mixc[1][moff], gs = mixc[2][moff], gh = mixc[3][moff], es = mixc[4][moff], eh = mixc[5][moff];
#pragma unroll
            for (int j0 = 0; j0 < TC; j0 += 4) {
                float vg[4], ve[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) vg[j] = mgrow[mcol[j0 + j]], ve[j] = merow[mcol[j0 + j]];
#pragma unroll
                for (int j = 0; j < 4; ++j) inv[j0 + j] = fmaf(fmaf(inv[j0 + j], ls, lh), sigmoidf_fast(fmaf(vg[j], gs, gh)), fmaf(ve[j], es, eh));
            }
        }
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            inv[j] = (tvalid && fb + j < f1) ? inv[j] : 0.f;
            acc[j] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < NCONV; ++k) {
            __syncthreads();
            const float* dyb = a.dy[k] + ubase;
            const float* xb = a.x[k] + ubase;
            const float4 A4 = ld4(&coefA[k][q4]);
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));  // (the items' tile coordinates are recomputed per stage, as in the kernel above)
#pragma unroll
            for (int h = 0; h < NIT; h += 2) {  // two items (four 16-byte loads) in flight per thread: 128 registers = sixteen waves per CU
                float4 vd[2], vx[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int item = tid + (h + i) * NT, px = min(item >> 4, R * CB - 1), pr = px / CB, pc = px - pr * CB;
                    const int tq = min(max(t0 - 2 + pr, 0), T - 1), fq = min(max(fb - 2 + pc, 0), F - 1);
                    const unsigned off = (((unsigned)tq * F + fq) * kH + q4) * 4u;
                    vd[i] = ld4_off(dyb, off);
                    vx[i] = ld4_off(xb, off);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int item = tid + (h + i) * NT, px = item >> 4, pr = px / CB, pc = px - pr * CB;
                    const int tq = t0 - 2 + pr, fq = fb - 2 + pc;
                    float4 d = f4(A4.x * vd[i].x - Bc[k] * vx[i].x + Cc[k], A4.y * vd[i].y - Bc[k] * vx[i].y + Cc[k], A4.z * vd[i].z - Bc[k] * vx[i].z + Cc[k],
                                  A4.w * vd[i].w - Bc[k] * vx[i].w + Cc[k]);
                    if (!(tq >= 0 && tq < T && fq >= 0 && fq < F)) d = f4(0, 0, 0, 0);
                    if (px < R * CB) st4(tile + px * 64 + q4, d);
                }
            }
            __syncthreads();
            const float* trow0 = tile + r * RS + ch;
            int always = 1;
            asm volatile("" : "+s"(always));
#pragma unroll
            for (int jb = 0; jb < TC; jb += 4) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) if (always) {  // (one basic block per tap row: see the kernel above)
                    const float* trow = trow0 + dt * RS;
                    float wr[7];
#pragma unroll
                    for (int c = 0; c < 7; ++c) wr[c] = trow[(jb + c) * 64];
#pragma unroll
                    for (int dc = 0; dc < 4; ++dc) {
                        const int tap = (3 - dt) * 4 + (3 - dc);
                        const float w = ws[k][tap * 64 + ch];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            acc[jb + jj] = fmaf(w, wr[jj + dc], acc[jb + jj]);
                            part[k][tap] = fmaf(inv[jb + jj], wr[jj + dc], part[k][tap]);
                        }
                    }
                    asm volatile("" : "+s"(always));
                }
            }
        }
        if (tvalid) {
#pragma unroll
            for (int j = 0; j < TC; ++j)
                if (fb + j < f1) outrow[(size_t)(fb + j) * kH] = acc[j];
        }
    }
    // tap gradients: a wave holds one row's partial sums; waves 0-3 park theirs in LDS, waves 4-7 add theirs, one coalesced atomic per (tap, channel)
    float* mine = spread_copy(a.scr, blockIdx.x);
    float* redl = tile;  // [4][16][64]
#pragma unroll
    for (int k = 0; k < NCONV; ++k) {
        __syncthreads();
        if (r < 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) redl[(r * 16 + i) * 64 + ch] = part[k][i];
        }
        __syncthreads();
        if (r >= 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) redl[((r - 4) * 16 + i) * 64 + ch] += part[k][i];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 1024; idx += NT)
            atomicAdd(mine + k * 1024 + idx, redl[idx] + redl[1024 + idx] + redl[2048 + idx] + redl[3072 + idx]);
    }
}

}  // namespace rtfs

using namespace rtfs;

static int dw_adjoint_launch(int nconv, const float* const* dy, const float* const* x, const double* const* x_stats, const double* const* red,
                             const float* const* gamma, const float* const* w, const float* in, const double* in_stats, const float* in_gamma,
                             const float* in_beta, float in_slope, int mode, float* dIn, int accumulate, float* const* dW, float* const* dbias, int B, int T,
                             int F, const float* gate_s, int Tg, int Fg, const void* const* in_mix, int in_Tg, int in_Fg, void* stream) {
    if (B <= 0 || T <= 0 || F <= 0 || (nconv != 1 && nconv != 2 && nconv != 4) || mode < 0 || mode > 3 || !dy || !w || !dW || !in || !dIn) return RTFS_EINVAL;
    if (mode >= 1 && (!in_stats || !in_gamma || !in_beta)) return RTFS_EINVAL;
    if ((size_t)T * F * kH * 4 >= (1ull << 32)) return RTFS_EINVAL;  // 32-bit byte offsets inside an utterance
    const bool gln = x != nullptr && x[0] != nullptr;
    const bool bias = dbias != nullptr && dbias[0] != nullptr;
    DwAdjArgs a{};
    a.T = T, a.F = F, a.B = B;
    a.nt = (T + 7) / 8;
    // f segments: enough workgroups to fill 256 CUs twice over, segments a multiple of the 8-column step
    a.nseg = 1;
    while (a.nt * B * a.nseg < 1024 && F / (a.nseg + 1) >= 16) ++a.nseg;
    a.fseg = (((F + a.nseg - 1) / a.nseg) + 7) / 8 * 8;
    a.nseg = (F + a.fseg - 1) / a.fseg;
    SpreadOut so{};
    for (int k = 0; k < nconv; ++k) {
        if (!dy[k] || !w[k] || !dW[k]) return RTFS_EINVAL;
        if (gln && (!x[k] || !x_stats || !x_stats[k] || !red || !red[k] || !gamma || !gamma[k])) return RTFS_EINVAL;
        if (bias && !dbias[k]) return RTFS_EINVAL;
        a.dy[k] = dy[k], a.w[k] = w[k];
        if (gln) a.x[k] = x[k], a.slot[k] = x_stats[k], a.red[k] = red[k], a.gamma[k] = gamma[k];
        if (bias) {
            so.dst[2 * k] = dW[k], so.n[2 * k] = 1024, so.dst[2 * k + 1] = dbias[k], so.n[2 * k + 1] = 64;
        } else {
            so.dst[k] = dW[k], so.n[k] = 1024;
        }
    }
    a.in = in, a.in_slot = in_stats, a.in_gamma = in_gamma, a.in_beta = in_beta, a.in_slope = in_slope;
    a.mode = mode, a.accumulate = accumulate ? 1 : 0, a.bias = bias ? 1 : 0;
    a.inv_n = 1.0 / ((double)T * F * kH);
    a.dIn = dIn;
    a.mt = div_magic_of(T), a.mf = div_magic_of(F);
    if (mode == 3) {
        if (!in_mix || in_Tg <= 0 || in_Fg <= 0 || in_Tg > T || in_Fg > F) return RTFS_EINVAL;
        for (int i = 0; i < 8; ++i)
            if (!in_mix[i]) return RTFS_EINVAL;
        a.mgate = (const float*)in_mix[0], a.mgate_slot = (const double*)in_mix[1], a.mgate_gamma = (const float*)in_mix[2], a.mgate_beta = (const float*)in_mix[3];
        a.mglob = (const float*)in_mix[4], a.mglob_slot = (const double*)in_mix[5], a.mglob_gamma = (const float*)in_mix[6], a.mglob_beta = (const float*)in_mix[7];
        a.mTg = in_Tg, a.mFg = in_Fg, a.m_inv_n = 1.0 / ((double)in_Tg * in_Fg * kH);
    }
    if (mode == 3 && nconv > 2 && !(x != nullptr && !accumulate)) return RTFS_EINVAL;  // (four convolutions with a mixed input: the one-channel-per-lane kernel only)
    const bool mix = gate_s != nullptr;
    if (mix) {
        if (nconv != 1 || !gln || Tg <= 0 || Fg <= 0 || Tg > T || Fg > F) return RTFS_EINVAL;
        a.gate_s = gate_s, a.Tg = Tg, a.Fg = Fg;
    }
    a.scr = spread_scratch();
    if (!a.scr) return RTFS_ELAUNCH;
    const int ntiles = a.nt * B * a.nseg;
    const dim3 grid((unsigned)(((ntiles + 7) / 8) * 8));
#define DWADJ(N, G) hipLaunchKernelGGL((dw_adjoint_kernel<N, G>), grid, dim3(256), 0, (hipStream_t)stream, a)
    if (mix) {
        hipLaunchKernelGGL((dw_adjoint_kernel<1, true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    } else if (gln && nconv >= 2 && !bias && (mode == 0 || mode == 3) && !accumulate) {  // the step's two- and four-convolution groups: one channel per lane
        if (nconv == 2 && mode == 3) hipLaunchKernelGGL((dw_adjoint1c_kernel<2, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
        else if (nconv == 2) hipLaunchKernelGGL((dw_adjoint1c_kernel<2, false>), grid, dim3(512), 0, (hipStream_t)stream, a);
        else if (mode == 3) hipLaunchKernelGGL((dw_adjoint1c_kernel<4, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((dw_adjoint1c_kernel<4, false>), grid, dim3(512), 0, (hipStream_t)stream, a);
    } else if (gln) {
        if (nconv == 1) DWADJ(1, true); else if (nconv == 2) DWADJ(2, true); else DWADJ(4, true);
    } else {
        if (nconv == 1) DWADJ(1, false); else if (nconv == 2) DWADJ(2, false); else DWADJ(4, false);
    }
#undef DWADJ
    RTFS_LAUNCH_CHECK();
    return spread_finish(a.scr, so, (hipStream_t)stream);
}

extern "C" {

int rtfs_dw_adjoint(int nconv, const float* const* dy, const float* const* x, const double* const* x_stats, const double* const* red,
                    const float* const* gamma, const float* const* w, const float* in, const double* in_stats, const float* in_gamma, const float* in_beta,
                    float in_slope, int mode, const void* const* in_mix, int in_Tg, int in_Fg, float* dIn, int accumulate, float* const* dW,
                    float* const* dbias, int B, int T, int F, void* stream) {
    return dw_adjoint_launch(nconv, dy, x, x_stats, red, gamma, w, in, in_stats, in_gamma, in_beta, in_slope, mode, dIn, accumulate, dW, dbias, B, T, F, nullptr,
                             0, 0, in_mix, in_Tg, in_Fg, stream);
}

int rtfs_dw_adjoint_mix(const float* dOut, const float* loc, const double* loc_stats, const double* loc_red, const float* loc_gamma, const float* gate_sig,
                        int Tg, int Fg, const float* w, const float* in, const double* in_stats, const float* in_gamma, const float* in_beta, float in_slope,
                        int mode, const void* const* in_mix, int in_Tg, int in_Fg, float* dIn, int accumulate, float* dW, int B, int T, int F, void* stream) {
    if (!dOut || !loc || !loc_stats || !loc_red || !loc_gamma || !gate_sig || !w || !dW) return RTFS_EINVAL;
    const float* dy[1] = {dOut};
    const float* x[1] = {loc};
    const double* xs[1] = {loc_stats};
    const double* red[1] = {loc_red};
    const float* gm[1] = {loc_gamma};
    const float* ww[1] = {w};
    float* dw[1] = {dW};
    return dw_adjoint_launch(1, dy, x, xs, red, gm, ww, in, in_stats, in_gamma, in_beta, in_slope, mode, dIn, accumulate, dw, nullptr, B, T, F, gate_sig, Tg, Fg,
                             in_mix, in_Tg, in_Fg, stream);
}

}  // extern "C"
