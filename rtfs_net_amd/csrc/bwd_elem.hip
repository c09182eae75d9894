// Backward primitives of the bandwidth-bound stages (training step, BASELINE configs 3-5).  The reference has no
// backward code: it is torch autograd over the forward modules, so every kernel here is the hand-derived adjoint of
// the forward kernel it names; parity is checked against autograd of the oracle (tests/test_hip_backward.py).
//
//   rtfs_colsum_add           bias gradients: out[n] += sum_rows X[row][n]
//   rtfs_gln_bwd_reduce/apply GroupNorm(1,C) backward (adjoint of "normalise on read"), optional PReLU after the norm
//   rtfs_dwconv_bwd_input     transposed depth-wise 4x4 convolution (stride 1 / 2)
//   rtfs_dwconv_bwd_weight    tap / bias gradients of a depth-wise convolution, input re-normalised on read
//   rtfs_pool_bwd             adjoint of adaptive_avg_pool2d + add            (tdanet.py:117-118)
//   rtfs_d0_tail_bwd          stride-2 conv input gradient + rtfs_pool_bwd + the reduce pass of D0's gLN adjoint in one pass over d(gLN(D0))
//   rtfs_mix_bwd              adjoint of InjectionMultiSum's gate/upsample mix (fusion.py:59-67)
//   rtfs_mix_gln_bwd          the same fused with the gLN adjoint of the local branch (no dNloc tensor) and the gate / global branches' reduce passes
//   rtfs_expand_fwd           materialise `expanded` (TFAR tail) for the residual_conv weight gradient
//   rtfs_axpy                 y += a * x
#include "common.h"
#include "intdiv.h"

namespace rtfs {

struct NormArg {
    const float* x;      // pre-norm tensor [B][rows][C]
    const double* slot;  // forward (sum, sumsq) per utterance
    double inv_n;
    const float *gamma, *beta;
};

// Reduce per-thread float4 partials that share a channel quad (thread = (row = tid / QUADS, quad = tid % QUADS)) and add
// the workgroup total to out[0 .. 4*QUADS) with ONE coalesced atomic request per 128-byte line.  lds: 1024 floats.
template <int QUADS>
__device__ __forceinline__ void quad_reduce_atomic(float4 v, float* lds, float* out) {
    constexpr int NC = QUADS * 4, ROWS = 256 / QUADS;
    st4(lds + threadIdx.x * 4, v);  // element (row, channel) sits at row*NC + channel
    __syncthreads();
    if (threadIdx.x < NC) {
        float s = 0.f;
#pragma unroll 4
        for (int r = 0; r < ROWS; ++r) s += lds[r * NC + threadIdx.x];
        atomicAdd(out + threadIdx.x, s);
    }
    __syncthreads();
}

// block-wide sum of a scalar -> one atomicAdd (float) by thread 0.  lds: >= 4 floats
__device__ __forceinline__ void scalar_reduce_atomic(float v, float* lds, float* out) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, lds[0] + lds[1] + lds[2] + lds[3]);
    __syncthreads();
}

// ---- colsum ---------------------------------------------------------------------------------------------------------
template <int QUADS>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, float* __restrict__ scr, long long M, int rows_per_wg) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    constexpr int N = QUADS * 4;
    const int quad = threadIdx.x % QUADS, rsub = threadIdx.x / QUADS;
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < M ? r0 + rows_per_wg : M;
    float4 s = f4(0, 0, 0, 0);
    for (long long r = r0 + rsub; r < r1; r += 256 / QUADS) s = s + ld4(X + r * N + quad * 4);
    quad_reduce_atomic<QUADS>(s, lds, spread_copy(scr, blockIdx.x));
}

// ---- gLN backward ---------------------------------------------------------------------------------------------------
// y = xhat*gamma + beta, xhat = (x-mean)*rstd over one utterance (N elements).  g = dL/dy; with ACT == 1 the forward
// applied PReLU after the norm: the incoming gradient is w.r.t. prelu(y) and g = dY * prelu'(y).
//   a = g*gamma;  S1 = sum a;  S2 = sum a*xhat;   dgamma_c = sum g*xhat;  dbeta_c = sum g;
//   dx = rstd * (a - S1/N - xhat*S2/N)
template <int C, int ACT>
__global__ __launch_bounds__(256) void gln_bwd_reduce_kernel(const float* __restrict__ dY, NormArg n, float slope, double* __restrict__ red,
                                                             float* __restrict__ scr, int rows, int rows_per_wg) {
    constexpr int QUADS = C / 4;
    __shared__ __attribute__((aligned(16))) float lds[1024];
    __shared__ float redl[8];
    const int b = blockIdx.y;
    const int c4 = (threadIdx.x % QUADS) * 4, rsub = threadIdx.x / QUADS;
    float mean, rstd;
    stats_finalize(n.slot, b, n.inv_n, mean, rstd);
    const float4 g4 = ld4(n.gamma + c4), be4 = ld4(n.beta + c4);
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    float4 dg = f4(0, 0, 0, 0), db = f4(0, 0, 0, 0);
    float s1 = 0.f, s2 = 0.f, dsl = 0.f;
    for (int r = r0 + rsub; r < r1; r += 256 / QUADS) {
        const size_t o = ((size_t)b * rows + r) * C + c4;
        const float4 xh = sub4(ld4(n.x + o), mean) * rstd;
        float4 g = ld4(dY + o);
        if (ACT == 1) {
            const float4 y = fma4(xh, g4, be4);
            dsl += (y.x > 0.f ? 0.f : g.x * y.x) + (y.y > 0.f ? 0.f : g.y * y.y) + (y.z > 0.f ? 0.f : g.z * y.z) + (y.w > 0.f ? 0.f : g.w * y.w);
            g = f4(y.x > 0.f ? g.x : g.x * slope, y.y > 0.f ? g.y : g.y * slope, y.z > 0.f ? g.z : g.z * slope, y.w > 0.f ? g.w : g.w * slope);
        }
        if (ACT == 2) {
            const float4 y = fma4(xh, g4, be4);
            g = f4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
        }
        db = db + g;
        dg = fma4(g, xh, dg);
        const float4 a = g * g4;
        s1 += hsum4(a);
        s2 += dot4(a, xh);
    }
    float* mine = spread_copy(scr, blockIdx.x + blockIdx.y);  // [dgamma C | dbeta C | dslope]
    quad_reduce_atomic<QUADS>(dg, lds, mine);
    quad_reduce_atomic<QUADS>(db, lds, mine + C);
    if (ACT == 1) scalar_reduce_atomic(dsl, lds, mine + 2 * C);
    block_stats_commit(s1, s2, redl, red, b);
}

// dX (= or +=) rstd * (a - S1/N - xhat*S2/N)
template <int C, int ACT, bool ACCUM>
__global__ __launch_bounds__(256) void gln_bwd_apply_kernel(const float* __restrict__ dY, NormArg n, float slope, const double* __restrict__ red,
                                                            float* __restrict__ dX, int rows) {
    constexpr int QUADS = C / 4;
    const int b = blockIdx.y;
    const int r = blockIdx.x * (256 / QUADS) + threadIdx.x / QUADS;
    if (r >= rows) return;
    const int c4 = (threadIdx.x % QUADS) * 4;
    float mean, rstd;
    stats_finalize(n.slot, b, n.inv_n, mean, rstd);
    const float m1 = (float)(red[kStatStride * b] * n.inv_n), m2 = (float)(red[kStatStride * b + 1] * n.inv_n);
    const float4 g4 = ld4(n.gamma + c4);
    const size_t o = ((size_t)b * rows + r) * C + c4;
    const float4 xh = sub4(ld4(n.x + o), mean) * rstd;
    float4 g = ld4(dY + o);
    if (ACT == 1) {
        const float4 y = fma4(xh, g4, ld4(n.beta + c4));
        g = f4(y.x > 0.f ? g.x : g.x * slope, y.y > 0.f ? g.y : g.y * slope, y.z > 0.f ? g.z : g.z * slope, y.w > 0.f ? g.w : g.w * slope);
    }
    if (ACT == 2) {
        const float4 y = fma4(xh, g4, ld4(n.beta + c4));
        g = f4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
    }
    const float4 a = g * g4;
    float4 d = f4((a.x - m1 - xh.x * m2) * rstd, (a.y - m1 - xh.y * m2) * rstd, (a.z - m1 - xh.z * m2) * rstd, (a.w - m1 - xh.w * m2) * rstd);
    if (ACCUM) d = d + ld4(dX + o);
    st4(dX + o, d);
}

// ---- depth-wise conv backward -----------------------------------------------------------------------------------------
// forward: out[to][fo] = bias + sum_{dt,df} w[dt*4+df] * in[to*S-1+dt][fo*S-1+df]  (zero outside), per channel.
// dIn[ti][fi] (= or +=) sum_{dt,df : (ti+1-dt) % S == 0, (fi+1-df) % S == 0} w[dt*4+df] * dOut[(ti+1-dt)/S][(fi+1-df)/S]
template <int STRIDE, bool ACCUM>
__global__ __launch_bounds__(256) void dwconv_bwd_input_kernel(const float* __restrict__ dOut, const float* __restrict__ w, float* __restrict__ dIn,
                                                               int Tin, int Fin, int Tout, int Fout) {
    __shared__ __attribute__((aligned(16))) float ws[16 * 64];
    for (int i = threadIdx.x; i < 256; i += 256) st4(&ws[i * 4], ld4(w + i * 4));
    __syncthreads();
    const int b = blockIdx.y;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= Tin * Fin) return;
    const int c4 = (threadIdx.x & 15) * 4;
    const int ti = p / Fin, fi = p - ti * Fin;
    float4 acc = f4(0, 0, 0, 0);
    const float* ob = dOut + (size_t)b * Tout * Fout * kH + c4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const int tn = ti + 1 - dt;
        if (tn < 0 || (tn % STRIDE) != 0) continue;
        const int to = tn / STRIDE;
        if (to >= Tout) continue;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const int fn = fi + 1 - df;
            if (fn < 0 || (fn % STRIDE) != 0) continue;
            const int fo = fn / STRIDE;
            if (fo >= Fout) continue;
            acc = fma4(ld4(&ws[(dt * 4 + df) * 64 + c4]), ld4(ob + ((size_t)to * Fout + fo) * kH), acc);
        }
    }
    float* o = dIn + ((size_t)b * Tin * Fin + p) * kH + c4;
    if (ACCUM) acc = acc + ld4(o);
    st4(o, acc);
}

// Stride-1 input gradient in sliding-window form (the transposed convolution is itself a 4x4 depth-wise convolution of dOut with
// the flipped taps): dIn[ti][fi] = sum_{r,cc} w[(3-r)*4 + (3-cc)] * dOut[ti-2+r][fi-2+cc].  Thread = (input time row, channel
// quad) walking a frequency segment with a 4x4 register window of dOut (column c in slot (c+2)&3) and the 16 taps in registers:
// dOut is loaded once per overlapping time row instead of 16x.  grid (ceil(T/16), B, nseg), fseg % 4 == 0.
template <bool ACCUM>
__global__ __launch_bounds__(256, 2) void dwconv_bwd_input_s1_kernel(const float* __restrict__ dOut, const float* __restrict__ w, float* __restrict__ dIn,
                                                                     int T, int F, int fseg) {
    const int b = blockIdx.y;
    const int c4 = (threadIdx.x & 15) * 4;
    const int ti = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int f0 = blockIdx.z * fseg, f1 = min(F, f0 + fseg);
    const bool tvalid = ti < T;
    const float* rowp[4];
    float rmask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int to = ti - 2 + r;
        rmask[r] = (tvalid && to >= 0 && to < T) ? 1.f : 0.f;
        rowp[r] = dOut + (((size_t)b * T + min(max(to, 0), T - 1)) * F) * kH + c4;
    }
    auto load_col = [&](int c, float4v(&col)[4]) {  // (native 4-vectors: packed FMAs, see dwconv_bwd_weight_kernel)
        const float cm = (c >= 0 && c < F) ? 1.f : 0.f;
        const size_t off = (size_t)min(max(c, 0), F - 1) * kH;
        float4v x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = ld4v(rowp[r] + off);
#pragma unroll
        for (int r = 0; r < 4; ++r) col[r] = x[r] * (cm * rmask[r]);
    };
    float4v wreg[16];  // wreg[r*4+cc] = w[(3-r)*4 + (3-cc)]
#pragma unroll
    for (int i = 0; i < 16; ++i) wreg[i] = ld4v(w + (15 - i) * 64 + c4);
    float4v win[4][4];
    load_col(f0 - 2, win[0]);
    load_col(f0 - 1, win[1]);
    load_col(f0, win[2]);
    float* orow = dIn + (((size_t)b * T + (tvalid ? ti : 0)) * F) * kH + c4;
#pragma unroll 1
    for (int f = f0; f < f1; f += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int fi = f + j;
            load_col(fi + 1, win[(j + 3) & 3]);
            float4v old = float4v{0.f, 0.f, 0.f, 0.f};
            if (ACCUM) old = ld4v(orow + (size_t)min(fi, F - 1) * kH);
            float4v acc = old;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) acc = wreg[r * 4 + cc] * win[(j + cc) & 3][r] + acc;
            if (tvalid && fi < f1) st4(orow + (size_t)fi * kH, to_f4(acc));
        }
    }
}

// dW[tap][c] += sum_{b,to,fo} dOut[to][fo][c] * xin[to*S-1+dt][fo*S-1+df][c];  dbias[c] += sum dOut.
// xin = transformed forward input (MODE 0 raw, 1 gLN, 2 PReLU(gLN)).  Same sliding-window walk as the forward kernel
// (tfar.hip dwconv_kernel): thread = (output time row, channel quad) keeps a 4-column x 4-row register window of xin while
// it walks a frequency segment, so xin is loaded once per overlapping time row instead of 16x; the 16 tap partials + the
// bias partial stay in registers.  The 4 rows of a wave are summed with two xor shuffles (lanes 16 and 32 apart),
// the 4 waves through LDS, and the workgroup leaves with one coalesced fp32 atomic per (tap, channel).

template <int STRIDE, int MODE>
__global__ __launch_bounds__(256, 2) void dwconv_bwd_weight_kernel(const float* __restrict__ dOut, NormArg n, float slope, float* __restrict__ scr,
                                                                   int Tin, int Fin, int Tout, int Fout, int fseg) {
    __shared__ __attribute__((aligned(16))) float red[4][17][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane = (row of the wave) * 16 + channel quad: 16 consecutive lanes read one row's 256 contiguous bytes.  (Until round 4 the row sat in the
    // low two lane bits so that quad_perm DPP adds could sum the rows: every group of four lanes then touched four different cache lines, and
    // the vector L1 - one line per cycle - made each load instruction cost 64 cycles instead of 16: the kernel ran at 3 TB/s on L1 issue.)
    const int c4 = (lane & 15) * 4;
    const int to = blockIdx.x * 16 + wave * 4 + (lane >> 4);
    const int f0 = blockIdx.z * fseg, f1 = min(Fout, f0 + fseg);
    float4 sc = f4(1, 1, 1, 1), sh = f4(0, 0, 0, 0);
    if (MODE >= 1) {
        float mean, rstd;
        stats_finalize(n.slot, b, n.inv_n, mean, rstd);
        const float4 g = ld4(n.gamma + c4), be = ld4(n.beta + c4);
        sc = g * rstd;
        sh = f4(be.x - mean * sc.x, be.y - mean * sc.y, be.z - mean * sc.z, be.w - mean * sc.w);
    }
    const bool tvalid = to < Tout;
    const float* rowp[4];
    float rmask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ti = to * STRIDE - 1 + r;
        const bool ok = tvalid && ti >= 0 && ti < Tin;
        rmask[r] = ok ? 1.f : 0.f;
        rowp[r] = n.x + (((size_t)b * Tin + min(max(ti, 0), Tin - 1)) * Fin) * kH + c4;
    }
    auto load_col = [&](int c, float4(&col)[4]) {
        const float cm = (c >= 0 && c < Fin) ? 1.f : 0.f;
        const size_t off = (size_t)min(max(c, 0), Fin - 1) * kH;
        float4 x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = ld4(rowp[r] + off);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 v = x[r];
            if (MODE >= 1) v = fma4(v, sc, sh);
            if (MODE == 2) v = prelu4(v, slope);
            col[r] = v * (cm * rmask[r]);
        }
    };
    float4 win[4][4];  // [column slot][time row]; column c of the input lives in slot (c+1)&3
    if (STRIDE == 1) {
        load_col(f0 - 1, win[0]);
        load_col(f0, win[1]);
        load_col(f0 + 1, win[2]);
    } else {
        load_col(2 * f0 - 1, win[0]);
        load_col(2 * f0, win[1]);
    }
    float4 part[16], pb = f4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) part[i] = f4(0, 0, 0, 0);
    const float* grow = dOut + (((size_t)b * Tout + (tvalid ? to : 0)) * Fout) * kH + c4;
    constexpr int STEPS = 4 / STRIDE;
#pragma unroll 1
    for (int f = f0; f < f1; f += STEPS) {
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const int fo = f + j;
            float4 g = ld4(grow + (size_t)min(fo, Fout - 1) * kH);
            int base;
            if (STRIDE == 1) {
                load_col(fo + 2, win[(j + 3) & 3]);
                base = j;
            } else {
                load_col(2 * fo + 1, win[(2 * j + 2) & 3]);
                load_col(2 * fo + 2, win[(2 * j + 3) & 3]);
                base = 2 * j;
            }
            g = g * ((tvalid && fo < f1) ? 1.f : 0.f);
            pb = pb + g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int df = 0; df < 4; ++df) part[dt * 4 + df] = fma4(g, win[(base + df) & 3][dt], part[dt * 4 + df]);
        }
    }
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        float4 v = i < 16 ? part[i] : pb;
        v = f4(v.x + __shfl_xor(v.x, 16, 64), v.y + __shfl_xor(v.y, 16, 64), v.z + __shfl_xor(v.z, 16, 64), v.w + __shfl_xor(v.w, 16, 64));
        v = f4(v.x + __shfl_xor(v.x, 32, 64), v.y + __shfl_xor(v.y, 32, 64), v.z + __shfl_xor(v.z, 32, 64), v.w + __shfl_xor(v.w, 32, 64));
        if (lane < 16) st4(&red[wave][i][c4], v);
    }
    __syncthreads();
    float* mine = spread_copy(scr, blockIdx.x + blockIdx.y + blockIdx.z);  // [dW 16*64 | dbias 64]
    for (int idx = threadIdx.x; idx < 17 * 64; idx += 256) {
        const int i = idx >> 6, c = idx & 63;
        atomicAdd(mine + idx, red[0][i][c] + red[1][i][c] + red[2][i][c] + red[3][i][c]);
    }
}

// ---- pool backward ----------------------------------------------------------------------------------------------------
// forward: G[t2][f2] = mean_{window(t2) x window(f2)} n(D0) + n(D1);  window(i) = [floor(i*in/out), ceil((i+1)*in/out)).
// dN_D0[t][f] += sum over windows containing (t,f) of dG / |window|.   (dN_D1 += dG is an rtfs_axpy.)
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dG, float* __restrict__ dN0, int T, int T2) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= T * kF) return;
    const int c4 = (threadIdx.x & 15) * 4;
    const int t = p / kF, f = p - t * kF;
    float4 acc = f4(0, 0, 0, 0);
    const int tc = (t * T2) / T, fc = (f * kF2) / kF;
    for (int t2 = max(tc - 1, 0); t2 <= min(tc + 1, T2 - 1); ++t2) {
        const int ts = (t2 * T) / T2, te = ((t2 + 1) * T + T2 - 1) / T2;
        if (t < ts || t >= te) continue;
        for (int f2 = max(fc - 1, 0); f2 <= min(fc + 1, kF2 - 1); ++f2) {
            const int fs = (f2 * kF) / kF2, fe = ((f2 + 1) * kF + kF2 - 1) / kF2;
            if (f < fs || f >= fe) continue;
            const float inv = 1.0f / (float)((te - ts) * (fe - fs));
            acc = fma4(ld4(dG + (((size_t)b * T2 + t2) * kF2 + f2) * kH + c4), f4(inv, inv, inv, inv), acc);
        }
    }
    float* o = dN0 + ((size_t)b * T * kF + p) * kH + c4;
    st4(o, acc + ld4(o));
}

// ---- the last two contributions to d(gLN(D0)) + the reduce pass of D0's gLN adjoint, one pass over dN0 ------------------------
// dN0[p] += (transposed stride-2 convolution of dD1: downsample_layers[1])[p] + (pooling adjoint of dG)[p]   -- dwconv_bwd_input_kernel<2, true>
// and pool_bwd_kernel in one read-modify-write instead of two; the finished value v feeds D0's gLN reduce pass in registers
// (dgamma += v*xhat, dbeta += v, S1 += v*gamma, S2 += v*gamma*xhat): rtfs_gln_bwd_reduce's two full-resolution reads are one read of D0.
// `ppw` groups of 16 positions per workgroup share one reduction epilogue (mix_gln_bwd_reduce_kernel's).
__global__ __launch_bounds__(256) void d0_tail_bwd_kernel(const float* __restrict__ dD1, const float* __restrict__ w, const float* __restrict__ dG,
                                                          float* __restrict__ dN0, NormArg n, double* __restrict__ red, float* __restrict__ scr, int T,
                                                          int T2, int ppw, unsigned mT, unsigned mT2) {
    __shared__ __attribute__((aligned(16))) float ws[16 * 64];
    __shared__ __attribute__((aligned(16))) float lds[4][2][64];
    __shared__ float redl[4][2];
    st4(&ws[threadIdx.x * 4], ld4(w + threadIdx.x * 4));
    __syncthreads();
    const int b = blockIdx.y;
    const int c4 = (threadIdx.x & 15) * 4;
    float mean, rstd;
    stats_finalize(n.slot, b, n.inv_n, mean, rstd);
    const float4 g4 = ld4(n.gamma + c4);
    float4 dg = f4(0, 0, 0, 0), db = f4(0, 0, 0, 0);
    float s1 = 0.f, s2 = 0.f;
    const float* ob = dD1 + (size_t)b * T2 * kF2 * kH + c4;
    const float* gb = dG + (size_t)b * T2 * kF2 * kH + c4;
    for (int it = 0; it < ppw; ++it) {
        const int p = (blockIdx.x * ppw + it) * 16 + (threadIdx.x >> 4);
        if (p >= T * kF) break;
        const int t = p / kF, f = p - t * kF;
        float4 acc = f4(0, 0, 0, 0);
        // Round 6: branch-free.  A load under a divergent branch is waited for before the next one issues (DESIGN.md rule 1), and this loop nest was 16 + 9 of
        // them per position (283 us against 186 us of HBM time).  (a) Stride-2 transposed convolution (padding 1): to = (t + 1 - dt) / 2 is an integer for exactly
        // two dt (dt = pt, pt + 2 with pt = (t + 1) & 1), likewise df: four candidate taps, loaded from clamped addresses in one batch and masked.  (b) Pooling
        // adjoint: the windows [floor(i in / out), ceil((i + 1) in / out)) that contain (t, f) are among t2 = tc - 1 .. tc + 1, f2 = fc - 1 .. fc + 1: nine
        // masked loads of dG (neighbouring positions share them in L1).  The divisions by the kernel arguments T / T2 are multiply-high + correction (intdiv.h).
        {
            const int pt = (t + 1) & 1, pf = (f + 1) & 1;
            float4 tv[4], tw[4];
            float tm[4];
#pragma unroll
            for (int ia = 0; ia < 2; ++ia) {
                const int dt = pt + 2 * ia, tn = t + 1 - dt, to = tn >> 1;
                const bool okt = tn >= 0 && to < T2;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int df = pf + 2 * ib, fn = f + 1 - df, fo = fn >> 1;
                    const bool ok = okt && fn >= 0 && fo < kF2;
                    tv[ia * 2 + ib] = ld4(ob + ((size_t)min(max(to, 0), T2 - 1) * kF2 + min(max(fo, 0), kF2 - 1)) * kH);
                    tw[ia * 2 + ib] = ld4(&ws[(dt * 4 + df) * 64 + c4]);
                    tm[ia * 2 + ib] = ok ? 1.f : 0.f;
                }
            }
            const int tc = div_magic((unsigned)(t * T2), T, mT), fc = (f * kF2) / kF;
            float4 pv[9];
            float pm[9];
#pragma unroll
            for (int ia = 0; ia < 3; ++ia) {
                const int t2 = tc - 1 + ia, t2c = min(max(t2, 0), T2 - 1);
                const int ts = div_magic((unsigned)(t2c * T), T2, mT2), te = div_magic((unsigned)((t2c + 1) * T + T2 - 1), T2, mT2);
                const bool okt = t2 >= 0 && t2 < T2 && t >= ts && t < te;
#pragma unroll
                for (int ib = 0; ib < 3; ++ib) {
                    const int f2 = fc - 1 + ib, f2c = min(max(f2, 0), kF2 - 1);
                    const int fs = (f2c * kF) / kF2, fe = ((f2c + 1) * kF + kF2 - 1) / kF2;
                    const bool ok = okt && f2 >= 0 && f2 < kF2 && f >= fs && f < fe;
                    pv[ia * 3 + ib] = ld4(gb + ((size_t)t2c * kF2 + f2c) * kH);
                    pm[ia * 3 + ib] = ok ? __builtin_amdgcn_rcpf((float)((te - ts) * (fe - fs))) : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fma4(tw[k] * tm[k], tv[k], acc);
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fma4(pv[k], f4(pm[k], pm[k], pm[k], pm[k]), acc);
        }
        const size_t o = ((size_t)b * T * kF + p) * kH + c4;
        const float4 v = acc + ld4(dN0 + o);
        st4(dN0 + o, v);
        const float4 xh = sub4(ld4(n.x + o), mean) * rstd;
        db = db + v;
        dg = fma4(v, xh, dg);
        const float4 a = v * g4;
        s1 += hsum4(a);
        s2 += dot4(a, xh);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float4 v = i ? db : dg;
        v = f4(v.x + __shfl_xor(v.x, 16, 64), v.y + __shfl_xor(v.y, 16, 64), v.z + __shfl_xor(v.z, 16, 64), v.w + __shfl_xor(v.w, 16, 64));
        v = f4(v.x + __shfl_xor(v.x, 32, 64), v.y + __shfl_xor(v.y, 32, 64), v.z + __shfl_xor(v.z, 32, 64), v.w + __shfl_xor(v.w, 32, 64));
        if (lane < 16) st4(&lds[wave][i][c4], v);
        const float tsum = wave_sum(i ? s2 : s1);
        if (lane == 0) redl[wave][i] = tsum;
    }
    __syncthreads();
    float* mine = spread_copy(scr, blockIdx.x + blockIdx.y);  // [dgamma 64 | dbeta 64]
    if (threadIdx.x < 128) {
        const int i = threadIdx.x >> 6, c = threadIdx.x & 63;
        atomicAdd(mine + threadIdx.x, lds[0][i][c] + lds[1][i][c] + lds[2][i][c] + lds[3][i][c]);
    }
    if (threadIdx.x < 2) {
        const int i = threadIdx.x;
        atomicAdd(red + kStatStride * b + i, (double)redl[0][i] + (double)redl[1][i] + (double)redl[2][i] + (double)redl[3][i]);
    }
}

// ---- mix backward -----------------------------------------------------------------------------------------------------
// forward: out[p] = n(loc)[p] * sigmoid(n(gate)[up(p)]) + n(glob)[up(p)],  up = nearest (floor(dst*in/out)).
// full-resolution part: dNloc[p] = dOut[p] * sigmoid(n(gate)[up(p)])
__global__ __launch_bounds__(256) void mix_bwd_loc_kernel(const float* __restrict__ dOut, NormArg gate, float* __restrict__ dNloc, int T, int F, int Tg,
                                                          int Fg) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= T * F) return;
    const int c4 = (threadIdx.x & 15) * 4;
    const int t = p / F, f = p - t * F;
    const int tg = nearest_src(t, Tg, T), fg = nearest_src(f, Fg, F);
    float mean, rstd;
    stats_finalize(gate.slot, b, gate.inv_n, mean, rstd);
    const float4 g = ld4(gate.gamma + c4), be = ld4(gate.beta + c4);
    const float4 s = sigmoid4(fma4(sub4(ld4(gate.x + (((size_t)b * Tg + tg) * Fg + fg) * kH + c4), mean) * rstd, g, be));
    const size_t o = ((size_t)b * T * F + p) * kH + c4;
    st4(dNloc + o, ld4(dOut + o) * s);
}

// low-resolution part: over the footprint {p : up(p) = q}:  A = sum dOut*n(loc), Bs = sum dOut
//   dNgate[q] = A * s(1-s), s = sigmoid(n(gate)[q]);   dNglob[q] = Bs
__global__ __launch_bounds__(256) void mix_bwd_glob_kernel(const float* __restrict__ dOut, NormArg loc, NormArg gate, float* __restrict__ dNgate,
                                                           float* __restrict__ dNglob, int T, int F, int Tg, int Fg) {
    const int b = blockIdx.y;
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (q >= Tg * Fg) return;
    const int c4 = (threadIdx.x & 15) * 4;
    const int tg = q / Fg, fg = q - tg * Fg;
    // footprint of nearest up-sampling: t with floor(t*Tg/T) == tg  <=>  ceil(tg*T/Tg) <= t < ceil((tg+1)*T/Tg)
    const int t0 = (tg * T + Tg - 1) / Tg, t1 = min(T, ((tg + 1) * T + Tg - 1) / Tg);
    const int f0 = (fg * F + Fg - 1) / Fg, f1 = min(F, ((fg + 1) * F + Fg - 1) / Fg);
    float lm, lr, gm, gr;
    stats_finalize(loc.slot, b, loc.inv_n, lm, lr);
    stats_finalize(gate.slot, b, gate.inv_n, gm, gr);
    const float4 lg = ld4(loc.gamma + c4), lb = ld4(loc.beta + c4);
    float4 A = f4(0, 0, 0, 0), Bs = f4(0, 0, 0, 0);
    for (int t = t0; t < t1; ++t)
        for (int f = f0; f < f1; ++f) {
            const size_t o = (((size_t)b * T + t) * F + f) * kH + c4;
            const float4 d = ld4(dOut + o);
            A = fma4(d, fma4(sub4(ld4(loc.x + o), lm) * lr, lg, lb), A);
            Bs = Bs + d;
        }
    const size_t o = ((size_t)b * Tg * Fg + q) * kH + c4;
    const float4 s = sigmoid4(fma4(sub4(ld4(gate.x + o), gm) * gr, ld4(gate.gamma + c4), ld4(gate.beta + c4)));
    st4(dNgate + o, f4(A.x * s.x * (1.f - s.x), A.y * s.y * (1.f - s.y), A.z * s.z * (1.f - s.z), A.w * s.w * (1.f - s.w)));
    st4(dNglob + o, Bs);
}

// ---- mix backward fused with the gLN adjoint reductions of its three branches --------------------------------------------
// The local branch's incoming gradient is g[p] = dOut[p] * s[up(p)], s = sigmoid(n(gate)): CONSTANT over the footprint of a low-resolution
// position q.  With Dx = sum_footprint dOut*xhat(loc) and Bs = sum_footprint dOut (what mix_bwd_glob_kernel forms anyway):
//   A = gamma*Dx + beta*Bs;  dgamma_loc += s*Dx;  dbeta_loc += s*Bs;  S1 += sum_c gamma*s*Bs;  S2 += sum_c gamma*s*Dx
// so the reduce pass of the local branch's gLN adjoint costs no extra read, and dNloc is never written: the apply pass re-forms it.
// The gate / global branches' incoming gradients dNgate = A s (1-s) and dNglob = Bs are in registers next to xhat(gate) (needed for s) and one
// extra low-resolution read of glob: their gLN reduce passes ride along too (rtfs_gln_bwd_apply then runs on dNgate / dNglob as before).
// Epilogue: the 4 position rows of a wave are summed with two xor shuffles, the 4 waves through LDS; 6 x 64 per-channel sums leave as one
// coalesced atomic each, the 6 per-utterance sums as fp64 atomics.   red: [3][B][kStatStride] (loc, gate, glob)
__global__ __launch_bounds__(256) void mix_gln_bwd_reduce_kernel(const float* __restrict__ dOut, NormArg loc, NormArg gate, NormArg glob,
                                                                 float* __restrict__ dNgate, float* __restrict__ dNglob, float* __restrict__ sig, double* __restrict__ red,
                                                                 float* __restrict__ scr, int T, int F, int Tg, int Fg, int B, int qpt, unsigned mTg, unsigned mFg) {
    __shared__ __attribute__((aligned(16))) float lds[4][6][64];
    __shared__ float redl[4][6];
    const int b = blockIdx.y;
    const int c4 = (threadIdx.x & 15) * 4;
    float4 ch[6];  // dgamma_loc, dbeta_loc, dgamma_gate, dbeta_gate, dgamma_glob, dbeta_glob
    float sm[6];   // (S1, S2) of loc, gate, glob
#pragma unroll
    for (int i = 0; i < 6; ++i) ch[i] = f4(0, 0, 0, 0), sm[i] = 0.f;
    float lm, lr, gm, gr, em, er;
    stats_finalize(loc.slot, b, loc.inv_n, lm, lr);
    stats_finalize(gate.slot, b, gate.inv_n, gm, gr);
    stats_finalize(glob.slot, b, glob.inv_n, em, er);
    const float4 lg = ld4(loc.gamma + c4), lb = ld4(loc.beta + c4);
    for (int it = 0; it < qpt; ++it) {  // qpt groups of 16 low-resolution positions per workgroup: one epilogue for all of them
        const int q = (blockIdx.x * qpt + it) * 16 + (threadIdx.x >> 4);
        if (q >= Tg * Fg) break;
        // (round 6: five divisions by the kernel arguments Tg / Fg per position as multiply-high + correction, csrc/intdiv.h)
        const int tg = div_magic((unsigned)q, Fg, mFg), fg = q - tg * Fg;
        const int t0 = div_magic((unsigned)(tg * T + Tg - 1), Tg, mTg), t1 = min(T, div_magic((unsigned)((tg + 1) * T + Tg - 1), Tg, mTg));
        const int f0 = div_magic((unsigned)(fg * F + Fg - 1), Fg, mFg), f1 = min(F, div_magic((unsigned)((fg + 1) * F + Fg - 1), Fg, mFg));
        const size_t o = ((size_t)b * Tg * Fg + q) * kH + c4;
        const float4 xg = sub4(ld4(gate.x + o), gm) * gr, xe = sub4(ld4(glob.x + o), em) * er;
        float4 Dx = f4(0, 0, 0, 0), Bs = f4(0, 0, 0, 0);
        for (int t = t0; t < t1; ++t)
            for (int f = f0; f < f1; ++f) {
                const size_t oo = (((size_t)b * T + t) * F + f) * kH + c4;
                const float4 d = ld4(dOut + oo);
                Dx = fma4(d, sub4(ld4(loc.x + oo), lm) * lr, Dx);
                Bs = Bs + d;
            }
        const float4 gg = ld4(gate.gamma + c4), eg = ld4(glob.gamma + c4);
        const float4 s = sigmoid4(fma4(xg, gg, ld4(gate.beta + c4)));
        const float4 A = fma4(lg, Dx, lb * Bs);
        const float4 dNg = f4(A.x * s.x * (1.f - s.x), A.y * s.y * (1.f - s.y), A.z * s.z * (1.f - s.z), A.w * s.w * (1.f - s.w));
        st4(dNgate + o, dNg);
        st4(dNglob + o, Bs);
        if (sig) st4(sig + o, s);  // (for a consumer that applies the local branch's adjoint on load: rtfs_dw_adjoint_mix)
        const float4 c0 = s * Dx, c1 = s * Bs, c2 = dNg * xg, c4v = Bs * xe;
        ch[0] = ch[0] + c0, ch[1] = ch[1] + c1;
        ch[2] = ch[2] + c2, ch[3] = ch[3] + dNg;
        ch[4] = ch[4] + c4v, ch[5] = ch[5] + Bs;
        sm[0] += dot4(lg, c1), sm[1] += dot4(lg, c0);
        sm[2] += dot4(gg, dNg), sm[3] += dot4(gg, c2);
        sm[4] += dot4(eg, Bs), sm[5] += dot4(eg, c4v);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float4 v = ch[i];
        v = f4(v.x + __shfl_xor(v.x, 16, 64), v.y + __shfl_xor(v.y, 16, 64), v.z + __shfl_xor(v.z, 16, 64), v.w + __shfl_xor(v.w, 16, 64));
        v = f4(v.x + __shfl_xor(v.x, 32, 64), v.y + __shfl_xor(v.y, 32, 64), v.z + __shfl_xor(v.z, 32, 64), v.w + __shfl_xor(v.w, 32, 64));
        if (lane < 16) st4(&lds[wave][i][c4], v);
        const float t = wave_sum(sm[i]);
        if (lane == 0) redl[wave][i] = t;
    }
    __syncthreads();
    float* mine = spread_copy(scr, blockIdx.x + blockIdx.y);  // [6][64]
    for (int idx = threadIdx.x; idx < 6 * 64; idx += 256) {
        const int i = idx >> 6, c = idx & 63;
        atomicAdd(mine + idx, lds[0][i][c] + lds[1][i][c] + lds[2][i][c] + lds[3][i][c]);
    }
    if (threadIdx.x < 6) {
        const int i = threadIdx.x;
        const double S = (double)redl[0][i] + (double)redl[1][i] + (double)redl[2][i] + (double)redl[3][i];
        atomicAdd(red + ((size_t)(i >> 1) * B + b) * kStatStride + (i & 1), S);
    }
}

// dLoc[p] = rstd * (a - S1/N - xhat*S2/N),  a = dOut[p] * s[up(p)] * gamma
__global__ __launch_bounds__(256) void mix_gln_bwd_apply_kernel(const float* __restrict__ dOut, NormArg loc, NormArg gate, const double* __restrict__ red,
                                                                float* __restrict__ dLoc, int T, int F, int Tg, int Fg) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= T * F) return;
    const int c4 = (threadIdx.x & 15) * 4;
    const int t = p / F, f = p - t * F;
    const int tg = nearest_src(t, Tg, T), fg = nearest_src(f, Fg, F);
    float lm, lr, gm, gr;
    stats_finalize(loc.slot, b, loc.inv_n, lm, lr);
    stats_finalize(gate.slot, b, gate.inv_n, gm, gr);
    const float m1 = (float)(red[kStatStride * b] * loc.inv_n), m2 = (float)(red[kStatStride * b + 1] * loc.inv_n);
    const float4 s = sigmoid4(fma4(sub4(ld4(gate.x + (((size_t)b * Tg + tg) * Fg + fg) * kH + c4), gm) * gr, ld4(gate.gamma + c4), ld4(gate.beta + c4)));
    const size_t o = ((size_t)b * T * F + p) * kH + c4;
    const float4 xh = sub4(ld4(loc.x + o), lm) * lr;
    const float4 a = ld4(dOut + o) * s * ld4(loc.gamma + c4);
    st4(dLoc + o, f4((a.x - m1 - xh.x * m2) * lr, (a.y - m1 - xh.y * m2) * lr, (a.z - m1 - xh.z * m2) * lr, (a.w - m1 - xh.w * m2) * lr));
}

// ---- expanded (TFAR tail), materialised for the residual_conv weight gradient ---------------------------------------
__global__ __launch_bounds__(256) void expand_kernel(NormArg cl, NormArg d0, NormArg cg, NormArg cgate, float* __restrict__ E, int T, int T2) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= T * kF) return;
    const int c4 = (threadIdx.x & 15) * 4;
    const int t = p / kF, f = p - t * kF;
    const int t2 = nearest_src(t, T2, T), f2 = nearest_src(f, kF2, kF);
    const size_t hi = ((size_t)b * T * kF + p) * kH + c4, lo = (((size_t)b * T2 + t2) * kF2 + f2) * kH + c4;
    auto nrm = [&](const NormArg& r, size_t o) {
        float m, rs;
        stats_finalize(r.slot, b, r.inv_n, m, rs);
        return fma4(sub4(ld4(r.x + o), m) * rs, ld4(r.gamma + c4), ld4(r.beta + c4));
    };
    st4(E + hi, fma4(nrm(cl, hi), sigmoid4(nrm(cgate, lo)), nrm(cg, lo)) + nrm(d0, hi));
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, float a, float* __restrict__ y, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) st4(y + i * 4, fma4(ld4(x + i * 4), f4(a, a, a, a), ld4(y + i * 4)));
}

}  // namespace rtfs

using namespace rtfs;

#define LAUNCH(kernel, grid, ...)                                                           \
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);      \
    RTFS_LAUNCH_CHECK();

extern "C" {

int rtfs_colsum_add(const float* X, float* out, long long M, int N, void* stream) {
    if (M <= 0) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    const int per = 512;
    dim3 grid((unsigned)((M + per - 1) / per));
    switch (N) {
        case 32: LAUNCH((colsum_kernel<8>), grid, X, scr, M, per); break;
        case 64: LAUNCH((colsum_kernel<16>), grid, X, scr, M, per); break;
        case 256: LAUNCH((colsum_kernel<64>), grid, X, scr, M, per); break;
        default: return RTFS_EINVAL;
    }
    return spread_finish(scr, SpreadOut{{out}, {N}}, (hipStream_t)stream);
}

// red: double[B][2], zeroed by the caller.  act: 0 none, 1 PReLU(slope) after the norm (C = 64; dslope accumulates), 2 ReLU after the norm (C = 256).
int rtfs_gln_bwd_reduce(const float* dY, const float* X, const double* stats, const float* gamma, const float* beta, int act, float slope, double* red,
                        float* dgamma, float* dbeta, float* dslope, int B, int rows, int C, void* stream) {
    if (B <= 0 || rows <= 0 || (C != 64 && C != 256)) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    NormArg n{X, stats, 1.0 / ((double)rows * C), gamma, beta};
    const int per = 256;
    dim3 grid((rows + per - 1) / per, B);
    if (C == 64) {
        if (act == 1) { LAUNCH((gln_bwd_reduce_kernel<64, 1>), grid, dY, n, slope, red, scr, rows, per); }
        else if (act == 0) { LAUNCH((gln_bwd_reduce_kernel<64, 0>), grid, dY, n, slope, red, scr, rows, per); }
        else return RTFS_EINVAL;
    } else {
        if (act == 2) { LAUNCH((gln_bwd_reduce_kernel<256, 2>), grid, dY, n, slope, red, scr, rows, per); }
        else if (act == 0) { LAUNCH((gln_bwd_reduce_kernel<256, 0>), grid, dY, n, slope, red, scr, rows, per); }
        else return RTFS_EINVAL;
    }
    return spread_finish(scr, SpreadOut{{dgamma, dbeta, act == 1 ? dslope : nullptr}, {C, C, act == 1 ? 1 : 0}}, (hipStream_t)stream);
}

int rtfs_gln_bwd_apply(const float* dY, const float* X, const double* stats, const float* gamma, const float* beta, int act, float slope,
                       const double* red, float* dX, int accumulate, int B, int rows, int C, void* stream) {
    if (B <= 0 || rows <= 0 || (C != 64 && C != 256)) return RTFS_EINVAL;
    NormArg n{X, stats, 1.0 / ((double)rows * C), gamma, beta};
    dim3 grid((rows + (1024 / C) - 1) / (1024 / C), B);
#define GLN_APPLY(CC, AA, AC) LAUNCH((gln_bwd_apply_kernel<CC, AA, AC>), grid, dY, n, slope, red, dX, rows)
    if (C == 64) {
        if (act == 1) { if (accumulate) { GLN_APPLY(64, 1, true); } else { GLN_APPLY(64, 1, false); } }
        else if (act == 0) { if (accumulate) { GLN_APPLY(64, 0, true); } else { GLN_APPLY(64, 0, false); } }
        else return RTFS_EINVAL;
    } else {
        if (act == 2) { if (accumulate) { GLN_APPLY(256, 2, true); } else { GLN_APPLY(256, 2, false); } }
        else if (act == 0) { if (accumulate) { GLN_APPLY(256, 0, true); } else { GLN_APPLY(256, 0, false); } }
        else return RTFS_EINVAL;
    }
#undef GLN_APPLY
    return RTFS_OK;
}

int rtfs_dwconv_bwd_input(const float* dOut, const float* w, float* dIn, int accumulate, int stride, int B, int Tin, int Fin, void* stream) {
    if (B <= 0 || (stride != 1 && stride != 2)) return RTFS_EINVAL;
    const int Tout = stride == 1 ? Tin : (Tin - 2) / 2 + 1, Fout = stride == 1 ? Fin : (Fin - 2) / 2 + 1;
    dim3 grid((Tin * Fin + 15) / 16, B);
    if (stride == 1) {
        const int nseg = Fin >= 96 ? 3 : 2, fseg = (((Fin + nseg - 1) / nseg) + 3) / 4 * 4;
        dim3 g1((Tin + 15) / 16, B, (Fin + fseg - 1) / fseg);
        if (accumulate) { LAUNCH((dwconv_bwd_input_s1_kernel<true>), g1, dOut, w, dIn, Tin, Fin, fseg); }
        else { LAUNCH((dwconv_bwd_input_s1_kernel<false>), g1, dOut, w, dIn, Tin, Fin, fseg); }
    } else {
        if (accumulate) { LAUNCH((dwconv_bwd_input_kernel<2, true>), grid, dOut, w, dIn, Tin, Fin, Tout, Fout); }
        else { LAUNCH((dwconv_bwd_input_kernel<2, false>), grid, dOut, w, dIn, Tin, Fin, Tout, Fout); }
    }
    return RTFS_OK;
}

// mode: transform of the forward input (0 raw, 1 gLN, 2 PReLU(gLN)); dbias may be NULL.
int rtfs_dwconv_bwd_weight(const float* dOut, const float* in, const double* stats_in, const float* gamma, const float* beta, float slope, int mode,
                           int stride, float* dW, float* dbias, int B, int Tin, int Fin, void* stream) {
    if (B <= 0 || (stride != 1 && stride != 2) || mode < 0 || mode > 2) return RTFS_EINVAL;
    const int Tout = stride == 1 ? Tin : (Tin - 2) / 2 + 1, Fout = stride == 1 ? Fin : (Fin - 2) / 2 + 1;
    NormArg n{in, stats_in, 1.0 / ((double)Tin * Fin * kH), gamma, beta};
    const int nseg = 2, fseg = (((Fout + nseg - 1) / nseg) + 3) / 4 * 4;
    dim3 grid((Tout + 15) / 16, B, (Fout + fseg - 1) / fseg);
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
#define DWW(S, M) LAUNCH((dwconv_bwd_weight_kernel<S, M>), grid, dOut, n, slope, scr, Tin, Fin, Tout, Fout, fseg)
    // stride 1, gLN input: the PReLU instantiation with slope 1 (x >= 0 ? x : 1 * x - the same bits): hipcc schedules the <1, 1> instantiation's
    // walk worse (same 247 VGPRs, 159 us against 123 at the headline shape)
    if (stride == 1 && mode == 1) { mode = 2; slope = 1.0f; }
    if (stride == 1) { if (mode == 0) { DWW(1, 0); } else { DWW(1, 2); } }
    else { if (mode == 0) { DWW(2, 0); } else if (mode == 1) { DWW(2, 1); } else { DWW(2, 2); } }
#undef DWW
    return spread_finish(scr, SpreadOut{{dW, dbias}, {16 * 64, 64}}, (hipStream_t)stream);
}

int rtfs_pool_bwd(const float* dG, float* dN0, int B, int T, int T2, void* stream) {
    if (B <= 0) return RTFS_EINVAL;
    LAUNCH(pool_bwd_kernel, dim3((T * kF + 15) / 16, B), dG, dN0, T, T2);
    return RTFS_OK;
}

// dN0 += stride-2 transposed conv of dD1 (taps w: downsample_layers[1]) + pooling adjoint of dG, then the reduce pass of D0's gLN adjoint on the
// finished dN0 (red: double[B][kStatStride] zeroed by the caller; dgamma / dbeta accumulate): replaces rtfs_dwconv_bwd_input(stride 2,
// accumulate) + rtfs_pool_bwd + rtfs_gln_bwd_reduce(act 0) of the block's tail.  T2 must be (T - 2) / 2 + 1.
int rtfs_d0_tail_bwd(const float* dD1, const float* w, const float* dG, float* dN0, const float* D0, const double* d0_stats, const float* gamma,
                     const float* beta, double* red, float* dgamma, float* dbeta, int B, int T, int T2, void* stream) {
    if (B <= 0 || T2 != (T - 2) / 2 + 1) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    NormArg n{D0, d0_stats, 1.0 / ((double)T * kF * kH), gamma, beta};
    const int ppw = 8;
    LAUNCH(d0_tail_bwd_kernel, dim3((T * kF + 16 * ppw - 1) / (16 * ppw), B), dD1, w, dG, dN0, n, red, scr, T, T2, ppw, div_magic_of(T), div_magic_of(T2));
    return spread_finish(scr, SpreadOut{{dgamma, dbeta}, {kH, kH}}, (hipStream_t)stream);
}

// loc at (T,F) with stats/gamma/beta; gate/glob at (Tg,Fg).  Outputs: dNloc [B][T][F][64], dNgate/dNglob [B][Tg][Fg][64].
int rtfs_mix_bwd(const float* dOut, const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                 const double* gate_stats, const float* gate_g, const float* gate_b, float* dNloc, float* dNgate, float* dNglob, int B, int T, int F, int Tg,
                 int Fg, void* stream) {
    if (B <= 0) return RTFS_EINVAL;
    NormArg l{loc, loc_stats, 1.0 / ((double)T * F * kH), loc_g, loc_b}, g{gate, gate_stats, 1.0 / ((double)Tg * Fg * kH), gate_g, gate_b};
    LAUNCH(mix_bwd_glob_kernel, dim3((Tg * Fg + 15) / 16, B), dOut, l, g, dNgate, dNglob, T, F, Tg, Fg);
    LAUNCH(mix_bwd_loc_kernel, dim3((T * F + 15) / 16, B), dOut, g, dNloc, T, F, Tg, Fg);
    return RTFS_OK;
}

// rtfs_mix_bwd followed by the gLN adjoint of the local branch (rtfs_gln_bwd_reduce + rtfs_gln_bwd_apply on dNloc) without dNloc - dLoc is the
// gradient w.r.t. the local conv's OUTPUT (pre-norm) - and by the REDUCE passes of the gate / global branches' gLN adjoints (the caller runs
// rtfs_gln_bwd_apply on dNgate / dNglob with red + B*16 / red + 2*B*16).  red: double[3][B][kStatStride] zeroed by the caller;
// dgb: six [64] accumulators (dgamma, dbeta of loc, gate, glob), all accumulate.
int rtfs_mix_gln_bwd(const float* dOut, const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                     const double* gate_stats, const float* gate_g, const float* gate_b, const float* glob, const double* glob_stats, const float* glob_g,
                     const float* glob_b, float* dLoc, float* dNgate, float* dNglob, double* red, float* const* dgb, int B, int T, int F, int Tg, int Fg,
                     void* stream) {
    return rtfs_mix_gln_bwd_sig(dOut, loc, loc_stats, loc_g, loc_b, gate, gate_stats, gate_g, gate_b, glob, glob_stats, glob_g, glob_b, dLoc, dNgate, dNglob, nullptr, red,
                                dgb, B, T, F, Tg, Fg, stream);
}

int rtfs_mix_gln_bwd_sig(const float* dOut, const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, const float* gate,
                         const double* gate_stats, const float* gate_g, const float* gate_b, const float* glob, const double* glob_stats, const float* glob_g,
                         const float* glob_b, float* dLoc, float* dNgate, float* dNglob, float* sig, double* red, float* const* dgb, int B, int T, int F, int Tg,
                         int Fg, void* stream) {
    if (B <= 0 || !dgb || (!dLoc && !sig)) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    NormArg l{loc, loc_stats, 1.0 / ((double)T * F * kH), loc_g, loc_b}, g{gate, gate_stats, 1.0 / ((double)Tg * Fg * kH), gate_g, gate_b},
        e{glob, glob_stats, 1.0 / ((double)Tg * Fg * kH), glob_g, glob_b};
    const int qpt = 8;  // 128 low-resolution positions per workgroup (measured 1 ... 16: the epilogue's 384 atomics stop showing from 4 up)
    LAUNCH(mix_gln_bwd_reduce_kernel, dim3((Tg * Fg + 16 * qpt - 1) / (16 * qpt), B), dOut, l, g, e, dNgate, dNglob, sig, red, scr, T, F, Tg, Fg, B, qpt, div_magic_of(Tg),
           div_magic_of(Fg));
    const int rc = spread_finish(scr, SpreadOut{{dgb[0], dgb[1], dgb[2], dgb[3], dgb[4], dgb[5]}, {kH, kH, kH, kH, kH, kH}}, (hipStream_t)stream);
    if (rc != RTFS_OK) return rc;
    if (dLoc) LAUNCH(mix_gln_bwd_apply_kernel, dim3((T * F + 15) / 16, B), dOut, l, g, red, dLoc, T, F, Tg, Fg);  // (NULL: the consumer applies it on load, rtfs_dw_adjoint_mix, from `sig`)
    return RTFS_OK;
}

int rtfs_expand_fwd(const float* cl, const double* cl_stats, const float* cl_g, const float* cl_b, const float* d0, const double* d0_stats,
                    const float* d0_g, const float* d0_b, const float* cg, const double* cg_stats, const float* cg_g, const float* cg_b, const float* cgate,
                    const double* cgate_stats, const float* cgate_g, const float* cgate_b, float* E, int B, int T, int T2, void* stream) {
    if (B <= 0) return RTFS_EINVAL;
    const double nf = 1.0 / ((double)T * kF * kH), nl = 1.0 / ((double)T2 * kF2 * kH);
    NormArg a{cl, cl_stats, nf, cl_g, cl_b}, d{d0, d0_stats, nf, d0_g, d0_b}, g{cg, cg_stats, nl, cg_g, cg_b}, s{cgate, cgate_stats, nl, cgate_g, cgate_b};
    LAUNCH(expand_kernel, dim3((T * kF + 15) / 16, B), a, d, g, s, E, T, T2);
    return RTFS_OK;
}

int rtfs_axpy(const float* x, float a, float* y, long long n, void* stream) {
    if (n <= 0 || (n & 3)) return RTFS_EINVAL;
    LAUNCH(axpy_kernel, dim3((unsigned)((n / 4 + 255) / 256)), x, a, y, n / 4);
    return RTFS_OK;
}

}  // extern "C"
