// VP (video processing) block, TRAINING step: the convolution / BatchNorm1d chain of the 1-D TDANetBlock
// (separators/tdanet.py:106-133 with is2d = False, config yaml:74-92) as HIP kernels, forward with batch statistics and the
// hand-derived adjoints.  The 13-token GlobalAttention in the middle of the block (layers/attention.py:28-73,192-220: LayerNorm,
// nn.MultiheadAttention, dropout / DropPath) stays PyTorch glue between the two halves (models/vp_train.py).
//
// BatchNorm1d in training mode couples ALL utterances of the batch (and, under SyncBatchNorm, of all ranks) at every one of the 26
// normalisations, so the chain is cut into kernels at exactly those points: a producer kernel writes the pre-norm ("raw") tensor and
// accumulates its per-channel (sum, sum of squares) into a statistics slot; consumers normalise ON READ from the slot.  Between the
// two the host may all-reduce the slot (SyncBatchNorm, train.py:145) - nothing else crosses ranks.  The adjoint mirrors it: a kernel
// leaves the gradient w.r.t. a BatchNorm OUTPUT ("dyhat"), rtfs_vp_bn_bwd_reduce forms (sum dyhat, sum dyhat * xhat) - which are
// also dbeta / dgamma - and the convolution adjoint applies  dx = gamma rstd (dyhat - S1/N - xhat S2/N)  on read.
//
// Tensors are [B][64][T] fp32 (the reference's NCT layout; 512 channels for the block input / output).  They are tiny (<= 100 tokens):
// one workgroup per utterance, thread = (channel c = tid & 63, time phase tid >> 6); parameter-gradient partials are reduced in the
// workgroup and leave with one atomic per element.
#include "common.h"

namespace rtfs {

constexpr int TVH = 64, TVIN = 512;

struct VBn {              // a BatchNorm read "on read": statistics slot [2][64] = (sum x, sum x^2) over n positions, float64: the sums of the
                          // <= B x 100 fp32 values are exact to 1e-16 whatever the order of the atomics, and E[x^2] - mean^2 is differenced in float64
    const double* stats;  // nullptr: no normalisation (identity)
    const float* gamma;
    const float* beta;
    float inv_n;          // 1 / (B * T) of the normalised tensor (all ranks under SyncBatchNorm)
};

__device__ __forceinline__ void vbn_coef(const VBn& r, int c, float& mean, float& rstd, float& sc, float& sh) {
    if (!r.stats) {
        mean = 0.f, rstd = 1.f, sc = 1.f, sh = 0.f;
        return;
    }
    const double m = r.stats[c] * (double)r.inv_n;
    const double var = fmax(r.stats[TVH + c] * (double)r.inv_n - m * m, 0.0);
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)kEps));
    sc = r.gamma[c] * rstd;
    sh = r.beta[c] - mean * sc;
}

// sum over the 4 time phases of a per-thread partial that belongs to channel c = tid & 63, then ONE atomic per channel
template <class Out>  // float: parameter gradients; double: BatchNorm statistics / adjoint sums (order-independent to 1e-16)
__device__ __forceinline__ void phase_reduce_atomic(float v, float* lds /*256*/, Out* out) {
    lds[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < 64)
        atomicAdd(out + threadIdx.x, (Out)lds[threadIdx.x] + (Out)lds[64 + threadIdx.x] + (Out)lds[128 + threadIdx.x] + (Out)lds[192 + threadIdx.x]);
    __syncthreads();
}

// ---- forward -------------------------------------------------------------------------------------------------------------------
// gateway (dw 1x1 + bias + PReLU) and projection conv:  r = prelu(x*gw + gb),  y = Wp . r + bp  (pre-BatchNorm) + statistics of y
__global__ __launch_bounds__(256) void vp_gate_proj_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gw, const float* __restrict__ gb,
                                                               float gslope, const float* __restrict__ Wp, const float* __restrict__ bp,
                                                               float* __restrict__ r, float* __restrict__ y, double* __restrict__ stats, int T) {
    // grid (B, ceil(T / 4)): one 4-step tile per workgroup - 9 KB of LDS, so that these side-stream kernels fit next to the resident workgroups of
    // the audio branch (a 32 KB tile made every launch wait for a CU to drain: the video chain then ran SLOWER than the PyTorch glue it replaces)
    __shared__ float red[256];
    __shared__ float rt[TVIN * 4];  // r tile [512][4 time steps]
    const int b = blockIdx.x, c = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const float* xb = x + (size_t)b * TVIN * T;
    float* rb = r + (size_t)b * TVIN * T;
    const int t0 = blockIdx.y * 4, nt = min(4, T - t0);
    for (int idx = threadIdx.x; idx < TVIN * 4; idx += 256) {
        const int k = idx >> 2, tt = idx & 3;
        float v = 0.f;
        if (tt < nt) {
            v = prelu(fmaf(xb[(size_t)k * T + t0 + tt], gw[k], gb[k]), gslope);
            rb[(size_t)k * T + t0 + tt] = v;
        }
        rt[idx] = v;
    }
    __syncthreads();
    float s = 0.f, q = 0.f;
    if (ph < nt) {  // thread = (output channel c, time step ph of the tile)
        float acc = bp[c];
        const float* wr = Wp + (size_t)c * TVIN;
#pragma unroll 8
        for (int k = 0; k < TVIN; ++k) acc = fmaf(wr[k], rt[k * 4 + ph], acc);
        y[((size_t)b * TVH + c) * T + t0 + ph] = acc;
        s = acc, q = acc * acc;
    }
    phase_reduce_atomic(s, red, stats);
    phase_reduce_atomic(q, red, stats + TVH);
}

// depth-wise k = 3 convolution (+ bias) of a tensor that is normalised (and optionally PReLU'd) on read; stride 1 ('same': pad 1, 1) or 2 (pad 1).
// in_act: 0 none, 1 PReLU(in_slope) after the normalisation.  nconv in {1, 2}: two convolutions of the same input (IMS global embedding + gate).
__global__ __launch_bounds__(256) void vp_dwconv_fwd_kernel(const float* __restrict__ src, VBn in, int in_act, float in_slope, const float* __restrict__ w0,
                                                            const float* __restrict__ b0, float* __restrict__ o0, double* __restrict__ st0,
                                                            const float* __restrict__ w1, float* __restrict__ o1, double* __restrict__ st1, int Tin,
                                                            int Tout, int stride) {
    __shared__ float red[256];
    const int b = blockIdx.x, c = threadIdx.x & 63, ph = threadIdx.x >> 6;
    float mean, rstd, sc, sh;
    vbn_coef(in, c, mean, rstd, sc, sh);
    const float* sb = src + ((size_t)b * TVH + c) * Tin;
    auto ld = [&](int p) {
        if (p < 0 || p >= Tin) return 0.f;  // zero padding applies to the transformed input
        float v = fmaf(sb[p], sc, sh);
        if (in_act == 1) v = prelu(v, in_slope);
        return v;
    };
    const float a0 = w0[c * 3], a1 = w0[c * 3 + 1], a2 = w0[c * 3 + 2], bias = b0 ? b0[c] : 0.f;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (w1) c0 = w1[c * 3], c1 = w1[c * 3 + 1], c2 = w1[c * 3 + 2];
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    for (int t = ph; t < Tout; t += 4) {
        const int p = t * stride - 1;
        const float x0 = ld(p), x1 = ld(p + 1), x2 = ld(p + 2);
        const float v = fmaf(a0, x0, fmaf(a1, x1, fmaf(a2, x2, bias)));
        o0[((size_t)b * TVH + c) * Tout + t] = v;
        s0 += v, q0 = fmaf(v, v, q0);
        if (w1) {
            const float u = fmaf(c0, x0, fmaf(c1, x1, c2 * x2));
            o1[((size_t)b * TVH + c) * Tout + t] = u;
            s1 += u, q1 = fmaf(u, u, q1);
        }
    }
    phase_reduce_atomic(s0, red, st0);
    phase_reduce_atomic(q0, red, st0 + TVH);
    if (w1) {
        phase_reduce_atomic(s1, red, st1);
        phase_reduce_atomic(q1, red, st1 + TVH);
    }
}

struct VPool {  // the four down-sampled tensors (raw, normalised on read) pooled to Tg and summed (tdanet.py:117-118)
    const float* raw[4];
    VBn bn[4];
    int T[4];
};
__global__ __launch_bounds__(256) void vp_pool_fwd_kernel(VPool p, float* __restrict__ g, int Tg) {
    const int b = blockIdx.x, c = threadIdx.x & 63, ph = threadIdx.x >> 6;
    for (int j = ph; j < Tg; j += 4) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float mean, rstd, sc, sh;
            vbn_coef(p.bn[i], c, mean, rstd, sc, sh);
            const int Ti = p.T[i], st = (j * Ti) / Tg, en = ((j + 1) * Ti + Tg - 1) / Tg;
            float a = 0.f;
            for (int q = st; q < en; ++q) a += fmaf(p.raw[i][((size_t)b * TVH + c) * Ti + q], sc, sh);
            acc += a / (float)(en - st);
        }
        g[((size_t)b * TVH + c) * Tg + j] = acc;
    }
}

// InjectionMultiSum mix (layers/fusion.py:59-67):  out = BN(loc) * sigmoid(BN(gate))^ + BN(emb)^ (+ BN(res));  ^ = nearest up-sampling To -> Tn
__global__ __launch_bounds__(256) void vp_mix_fwd_kernel(const float* __restrict__ loc, VBn lb, const float* __restrict__ gate, VBn gbn,
                                                         const float* __restrict__ emb, VBn eb, const float* __restrict__ res, VBn rb,
                                                         float* __restrict__ out, int Tn, int To) {
    const int b = blockIdx.x, c = threadIdx.x & 63, ph = threadIdx.x >> 6;
    float m, r, lsc, lsh, gsc, gsh, esc, esh, rsc = 1.f, rsh = 0.f;
    vbn_coef(lb, c, m, r, lsc, lsh);
    vbn_coef(gbn, c, m, r, gsc, gsh);
    vbn_coef(eb, c, m, r, esc, esh);
    if (res) vbn_coef(rb, c, m, r, rsc, rsh);
    for (int t = ph; t < Tn; t += 4) {
        const int src = nearest_src(t, To, Tn);
        const size_t on = ((size_t)b * TVH + c) * Tn + t, oo = ((size_t)b * TVH + c) * To + src;
        float v = fmaf(fmaf(loc[on], lsc, lsh), sigmoidf_fast(fmaf(gate[oo], gsc, gsh)), fmaf(emb[oo], esc, esh));
        if (res) v += fmaf(res[on], rsc, rsh);
        out[on] = v;
    }
}

// residual conv 64 -> 512 + bias + gateway residual:  out = Wr . e + br + r
__global__ __launch_bounds__(256) void vp_resid_fwd_kernel(const float* __restrict__ e, const float* __restrict__ Wr, const float* __restrict__ br,
                                                           const float* __restrict__ r, float* __restrict__ out, int T) {
    // grid (B, 8): output channels 64 y .. 64 y + 63; no LDS tile (see vp_gate_proj_fwd_kernel): e is read from L1 / L2
    const int b = blockIdx.x;
    const float* eb = e + (size_t)b * TVH * T;
    for (int il = threadIdx.x; il < 64 * T; il += 256) {
        const int idx = blockIdx.y * 64 * T + il;
        const int co = idx / T, t = idx - co * T;
        float acc = br[co];
        const float* wr = Wr + (size_t)co * TVH;
#pragma unroll 8
        for (int k = 0; k < TVH; ++k) acc = fmaf(wr[k], eb[k * T + t], acc);
        out[(size_t)b * TVIN * T + idx] = acc + r[(size_t)b * TVIN * T + idx];
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------------------
// adjoint of vp_resid_fwd w.r.t. e and the weights:  de = Wr^T . dout,  dWr += dout . e^T,  dbr += sum_t dout   (dr = dout is read by vp_gate_proj_bwd)
__global__ __launch_bounds__(256) void vp_resid_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ e, const float* __restrict__ Wr,
                                                           float* __restrict__ de, float* __restrict__ dWr, float* __restrict__ dbr, int T) {
    // grid (B, 8): workgroup (b, y) owns the 64 output channels co = 64 y .. 64 y + 63 of utterance b; de (zeroed by the caller) collects the 8 partial
    // sums over co with atomics.  Operands are read from L1 / L2 (no LDS tiles: see vp_gate_proj_fwd_kernel).
    const int b = blockIdx.x, co0 = blockIdx.y * 64;
    const float* db = dout + ((size_t)b * TVIN + co0) * T;
    const float* eb = e + (size_t)b * TVH * T;
    {  // dWr[co][k] += sum_t dout[co][t] e[k][t]: thread = (k = tid & 63, co phase)
        const int k = threadIdx.x & 63;
        for (int cl = threadIdx.x >> 6; cl < 64; cl += 4) {
            float acc = 0.f;
            for (int t = 0; t < T; ++t) acc = fmaf(db[cl * T + t], eb[k * T + t], acc);
            atomicAdd(dWr + (size_t)(co0 + cl) * TVH + k, acc);
        }
    }
    if (threadIdx.x < 64) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += db[threadIdx.x * T + t];
        atomicAdd(dbr + co0 + threadIdx.x, acc);
    }
    // de[k][t] += sum_{co in group} Wr[co][k] dout[co][t]
    for (int idx = threadIdx.x; idx < TVH * T; idx += 256) {
        const int k = idx / T, t = idx - k * T;
        float acc = 0.f;
        for (int cl = 0; cl < 64; ++cl) acc = fmaf(Wr[(size_t)(co0 + cl) * TVH + k], db[cl * T + t], acc);
        atomicAdd(de + (size_t)b * TVH * T + idx, acc);
    }
}

// adjoint of vp_mix_fwd: gradients w.r.t. the three BatchNorm OUTPUTS (written) and, accumulated, w.r.t. the residual's BatchNorm output
__global__ __launch_bounds__(256) void vp_mix_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ loc, VBn lb,
                                                         const float* __restrict__ gate, VBn gbn, float* __restrict__ dloc, float* __restrict__ dgate,
                                                         float* __restrict__ demb, float* __restrict__ dres_acc, int Tn, int To) {
    const int b = blockIdx.x, c = threadIdx.x & 63, ph = threadIdx.x >> 6;
    float m, r, lsc, lsh, gsc, gsh;
    vbn_coef(lb, c, m, r, lsc, lsh);
    vbn_coef(gbn, c, m, r, gsc, gsh);
    const size_t bn = ((size_t)b * TVH + c) * Tn, bo = ((size_t)b * TVH + c) * To;
    for (int t = ph; t < Tn; t += 4) {
        const int src = nearest_src(t, To, Tn);
        const float g = sigmoidf_fast(fmaf(gate[bo + src], gsc, gsh));
        dloc[bn + t] = dout[bn + t] * g;
        if (dres_acc) dres_acc[bn + t] += dout[bn + t];
    }
    for (int j = ph; j < To; j += 4) {  // children of low-resolution position j: the t with floor(t * To / Tn) == j
        const int t0 = (j * Tn + To - 1) / To, t1 = ((j + 1) * Tn + To - 1) / To;
        float sg = 0.f, se = 0.f;
        for (int t = t0; t < t1 && t < Tn; ++t) {
            const float d = dout[bn + t];
            sg = fmaf(d, fmaf(loc[bn + t], lsc, lsh), sg);
            se += d;
        }
        const float g = sigmoidf_fast(fmaf(gate[bo + j], gsc, gsh));
        dgate[bo + j] = sg * g * (1.f - g);
        demb[bo + j] = se;
    }
}

// (sum dyhat, sum dyhat * xhat) per channel of one BatchNorm = (dbeta, dgamma); sums [2][64] accumulated with atomics
__global__ __launch_bounds__(256) void vp_bn_bwd_reduce_kernel(const float* __restrict__ dyhat, const float* __restrict__ raw, VBn bn,
                                                               double* __restrict__ sums, int T) {
    __shared__ float red[256];
    const int b = blockIdx.x, c = threadIdx.x & 63, ph = threadIdx.x >> 6;
    float mean, rstd, sc, sh;
    vbn_coef(bn, c, mean, rstd, sc, sh);
    float s1 = 0.f, s2 = 0.f;
    const size_t o = ((size_t)b * TVH + c) * T;
    for (int t = ph; t < T; t += 4) {
        const float d = dyhat[o + t];
        s1 += d;
        s2 = fmaf(d, (raw[o + t] - mean) * rstd, s2);
    }
    phase_reduce_atomic(s1, red, sums);
    phase_reduce_atomic(s2, red, sums + TVH);
}

// adjoint of vp_dwconv_fwd for ONE convolution: dyhat = gradient w.r.t. the BatchNorm output of this convolution's result `raw`;
// dx = gamma rstd (dyhat - S1/N - xhat S2/N) (batch statistics; eval mode: gamma rstd dyhat) is formed on read, then
//   dW[c][k] += sum dx[t] u[t*s - 1 + k],  dbias[c] += sum dx,  du[p] = sum_k W[c][k] dx[(p + 1 - k) / s]
// where u = the transformed input (BatchNorm [+ PReLU] of `src`).  du is the gradient w.r.t. that transformed input; with in_act == 1 it is
// taken through the PReLU (dslope accumulated) so that what is stored is always the gradient w.r.t. the input's BatchNorm output.
__global__ __launch_bounds__(256) void vp_dwconv_bwd_kernel(const float* __restrict__ dyhat, const float* __restrict__ raw, VBn obn,
                                                            const double* __restrict__ sums, float inv_n_all, int batch_stats, const float* __restrict__ src,
                                                            VBn in, int in_act, float in_slope, const float* __restrict__ w, float* __restrict__ dW,
                                                            float* __restrict__ dbias, float* __restrict__ dsrc, int accumulate,
                                                            float* __restrict__ dslope, int Tin, int Tout, int stride) {
    __shared__ float red[256];
    const int b = blockIdx.x, c = threadIdx.x & 63, ph = threadIdx.x >> 6;
    float omean, orstd, osc, osh, imean, irstd, isc, ish;
    vbn_coef(obn, c, omean, orstd, osc, osh);
    vbn_coef(in, c, imean, irstd, isc, ish);
    const float m1 = batch_stats ? (float)(sums[c] * (double)inv_n_all) : 0.f, m2 = batch_stats ? (float)(sums[TVH + c] * (double)inv_n_all) : 0.f;
    const size_t oo = ((size_t)b * TVH + c) * Tout, oi = ((size_t)b * TVH + c) * Tin;
    const float w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];
    float gw0 = 0.f, gw1 = 0.f, gw2 = 0.f, gb = 0.f, gsl = 0.f;
    auto ld = [&](int p) {
        if (p < 0 || p >= Tin) return 0.f;
        float v = fmaf(src[oi + p], isc, ish);
        if (in_act == 1) v = prelu(v, in_slope);
        return v;
    };
    auto dxat = [&](int t) {  // BatchNorm adjoint on read (recomputed for each of the <= 3 uses: no LDS tile, see vp_gate_proj_fwd_kernel)
        const float xh = (raw[oo + t] - omean) * orstd;
        return osc * (dyhat[oo + t] - m1 - xh * m2);
    };
    for (int t = ph; t < Tout; t += 4) {
        const float dx = dxat(t);
        const int p = t * stride - 1;
        gw0 = fmaf(dx, ld(p), gw0), gw1 = fmaf(dx, ld(p + 1), gw1), gw2 = fmaf(dx, ld(p + 2), gw2);
        gb += dx;
    }
    if (dsrc) {
        for (int p = ph; p < Tin; p += 4) {
            float du = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int tn = p + 1 - k;
                if (tn >= 0 && tn % stride == 0 && tn / stride < Tout) du = fmaf(k == 0 ? w0 : (k == 1 ? w1 : w2), dxat(tn / stride), du);
            }
            if (in_act == 1) {
                const float y = fmaf(src[oi + p], isc, ish);
                if (y <= 0.f) {
                    gsl = fmaf(du, y, gsl);
                    du *= in_slope;
                }
            }
            dsrc[oi + p] = accumulate ? dsrc[oi + p] + du : du;
        }
    }
    phase_reduce_atomic(gw0, red, dW);  // dW stored [3][64] (tap-major); the host transposes to the parameter's [64][1][3]
    phase_reduce_atomic(gw1, red, dW + TVH);
    phase_reduce_atomic(gw2, red, dW + 2 * TVH);
    if (dbias) phase_reduce_atomic(gb, red, dbias);
    if (dslope && in_act == 1) {
        gsl = wave_sum(gsl);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gsl;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(dslope, red[0] + red[1] + red[2] + red[3]);
    }
}

struct VPoolB {
    float* d[4];  // gradients w.r.t. the four BatchNorm outputs, accumulated into
    int T[4];
};
__global__ __launch_bounds__(256) void vp_pool_bwd_kernel(const float* __restrict__ dg, VPoolB p, int Tg) {
    const int b = blockIdx.x, c = threadIdx.x & 63, ph = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int Ti = p.T[i];
        for (int q = ph; q < Ti; q += 4) {
            float acc = 0.f;
            for (int j = 0; j < Tg; ++j) {
                const int st = (j * Ti) / Tg, en = ((j + 1) * Ti + Tg - 1) / Tg;
                if (q >= st && q < en) acc += dg[((size_t)b * TVH + c) * Tg + j] / (float)(en - st);
            }
            p.d[i][((size_t)b * TVH + c) * Ti + q] += acc;
        }
    }
}

// adjoint of vp_gate_proj_fwd.  dyhat: gradient w.r.t. the projection BatchNorm output (already through its PReLU); dout: the block output
// gradient, which reaches r through the gateway residual.
//   dy = BN adjoint (on read);  dWp += dy . r^T;  dbp += sum dy;  dr = Wp^T dy + dout;  u = x gw + gb:  dgw += sum dr prelu'(u) x, dgb, dgslope;  dx = dr prelu'(u) gw
__global__ __launch_bounds__(256) void vp_gate_proj_bwd_kernel(const float* __restrict__ dyhat, const float* __restrict__ y, VBn ybn,
                                                               const double* __restrict__ sums, float inv_n_all, int batch_stats, const float* __restrict__ dout,
                                                               const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ gw,
                                                               const float* __restrict__ gb, float gslope, const float* __restrict__ Wp,
                                                               float* __restrict__ dWp, float* __restrict__ dbp, float* __restrict__ dgw,
                                                               float* __restrict__ dgb, float* __restrict__ dgslope, float* __restrict__ dx, int T) {
    // grid (B, 8): workgroup (b, y) owns the 64 input channels k = 64 y .. 64 y + 63 and walks the time axis in chunks of 16 (dy chunk = 4 KB of LDS;
    // every workgroup re-derives dy, only y == 0 adds dbp)
    constexpr int CT = 16;
    __shared__ float dys[TVH * CT];
    __shared__ float red[256];
    const int b = blockIdx.x, c = threadIdx.x & 63, ph = threadIdx.x >> 6, k0 = blockIdx.y * 64;
    float mean, rstd, sc, sh;
    vbn_coef(ybn, c, mean, rstd, sc, sh);
    const float m1 = batch_stats ? (float)(sums[c] * (double)inv_n_all) : 0.f, m2 = batch_stats ? (float)(sums[TVH + c] * (double)inv_n_all) : 0.f;
    const float* rb = r + ((size_t)b * TVIN + k0) * T;
    const float* xb = x + ((size_t)b * TVIN + k0) * T;
    const float* db = dout + ((size_t)b * TVIN + k0) * T;
    const int k = k0 + c;
    const float gwk = gw[k], gbk = gb[k];
    float sb = 0.f, gsl = 0.f, aw = 0.f, ab = 0.f;
    float wacc[16];  // dWp[c][k0 + ph + 4 j], j = 0..15
#pragma unroll
    for (int j = 0; j < 16; ++j) wacc[j] = 0.f;
    for (int t0 = 0; t0 < T; t0 += CT) {
        const int nt = min(CT, T - t0);
        __syncthreads();  // previous chunk consumed
        for (int tt = ph; tt < nt; tt += 4) {
            const size_t o = ((size_t)b * TVH + c) * T + t0 + tt;
            const float d = sc * (dyhat[o] - m1 - (y[o] - mean) * rstd * m2);
            dys[c * CT + tt] = d;
            sb += d;
        }
        __syncthreads();
        // dWp[c][k] += sum_t dy[c][t] r[k][t]
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float* rr = rb + (size_t)(ph + 4 * j) * T + t0;
            float acc = 0.f;
            for (int tt = 0; tt < nt; ++tt) acc = fmaf(dys[c * CT + tt], rr[tt], acc);
            wacc[j] += acc;
        }
        // input channel k = k0 + c, time steps ph, ph + 4, ...: dr = Wp^T dy + dout, gateway adjoint
        for (int tt = ph; tt < nt; tt += 4) {
            const int t = t0 + tt;
            float dr = db[(size_t)c * T + t];
            for (int cc = 0; cc < TVH; ++cc) dr = fmaf(Wp[(size_t)cc * TVIN + k], dys[cc * CT + tt], dr);
            const float xv = xb[(size_t)c * T + t], u = fmaf(xv, gwk, gbk);
            float du = dr;
            if (u <= 0.f) {
                gsl = fmaf(dr, u, gsl);
                du = dr * gslope;
            }
            aw = fmaf(du, xv, aw);
            ab += du;
            dx[((size_t)b * TVIN + k) * T + t] = du * gwk;
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) atomicAdd(dWp + (size_t)c * TVIN + k0 + ph + 4 * j, wacc[j]);
    if (blockIdx.y != 0) sb = 0.f;  // dbp once per utterance
    __syncthreads();
    phase_reduce_atomic(sb, red, dbp);
    phase_reduce_atomic(aw, red, dgw + k0);
    phase_reduce_atomic(ab, red, dgb + k0);
    gsl = wave_sum(gsl);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gsl;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dgslope, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace rtfs

using namespace rtfs;

static VBn mk_bn(const double* stats, const float* gamma, const float* beta, float inv_n) { return VBn{stats, gamma, beta, inv_n}; }

#define VP_CHECK(cond) \
    if (!(cond)) return RTFS_EINVAL

extern "C" {

int rtfs_vp_gate_proj_fwd(const float* x, const float* gw, const float* gb, float gslope, const float* Wp, const float* bp, float* r, float* y, double* stats,
                          int B, int T, void* stream) {
    VP_CHECK(B > 0 && T > 0);
    hipLaunchKernelGGL(vp_gate_proj_fwd_kernel, dim3(B, (T + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gw, gb, gslope, Wp, bp, r, y, stats, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_dwconv_fwd(const float* src, const double* in_stats, const float* in_gamma, const float* in_beta, float in_inv_n, int in_act, float in_slope,
                       const float* w0, const float* b0, float* out0, double* stats0, const float* w1, float* out1, double* stats1, int B, int Tin, int Tout,
                       int stride, void* stream) {
    VP_CHECK(B > 0 && Tin > 0 && Tout > 0 && (stride == 1 || stride == 2) && (in_act == 0 || in_act == 1));
    hipLaunchKernelGGL(vp_dwconv_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, src, mk_bn(in_stats, in_gamma, in_beta, in_inv_n), in_act, in_slope,
                       w0, b0, out0, stats0, w1, out1, stats1, Tin, Tout, stride);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_pool_fwd(const float* const* raw, const double* const* stats, const float* const* gamma, const float* const* beta, int T0, int T1, int T2, int T3,
                     float inv_n0, float inv_n1, float inv_n2, float inv_n3, float* g, int B, int Tg, void* stream) {
    VP_CHECK(B > 0 && Tg > 0);
    const int T[4] = {T0, T1, T2, T3};
    const float inv_n[4] = {inv_n0, inv_n1, inv_n2, inv_n3};
    VPool p;
    for (int i = 0; i < 4; ++i) p.raw[i] = raw[i], p.T[i] = T[i], p.bn[i] = mk_bn(stats[i], gamma[i], beta[i], inv_n[i]);
    hipLaunchKernelGGL(vp_pool_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p, g, Tg);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_mix_fwd(const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, float inv_n_loc, const float* gate,
                    const double* gate_stats, const float* gate_g, const float* gate_b, const float* emb, const double* emb_stats, const float* emb_g,
                    const float* emb_b, float inv_n_glob, const float* res, const double* res_stats, const float* res_g, const float* res_b, float* out,
                    int B, int Tn, int To, void* stream) {
    VP_CHECK(B > 0 && Tn > 0 && To > 0);
    hipLaunchKernelGGL(vp_mix_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, loc, mk_bn(loc_stats, loc_g, loc_b, inv_n_loc), gate,
                       mk_bn(gate_stats, gate_g, gate_b, inv_n_glob), emb, mk_bn(emb_stats, emb_g, emb_b, inv_n_glob), res,
                       mk_bn(res_stats, res_g, res_b, inv_n_loc), out, Tn, To);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_resid_fwd(const float* e, const float* Wr, const float* br, const float* r, float* out, int B, int T, void* stream) {
    VP_CHECK(B > 0 && T > 0 && T <= 4096);
    hipLaunchKernelGGL(vp_resid_fwd_kernel, dim3(B, 8), dim3(256), 0, (hipStream_t)stream, e, Wr, br, r, out, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_resid_bwd(const float* dout, const float* e, const float* Wr, float* de, float* dWr, float* dbr, int B, int T, void* stream) {
    VP_CHECK(B > 0 && T > 0 && T <= 4096);
    if (hipMemsetAsync(de, 0, (size_t)B * TVH * T * sizeof(float), (hipStream_t)stream) != hipSuccess) return RTFS_ELAUNCH;
    hipLaunchKernelGGL(vp_resid_bwd_kernel, dim3(B, 8), dim3(256), 0, (hipStream_t)stream, dout, e, Wr, de, dWr, dbr, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_mix_bwd(const float* dout, const float* loc, const double* loc_stats, const float* loc_g, const float* loc_b, float inv_n_loc, const float* gate,
                    const double* gate_stats, const float* gate_g, const float* gate_b, float inv_n_glob, float* dloc, float* dgate, float* demb,
                    float* dres_acc_or_null, int B, int Tn, int To, void* stream) {
    VP_CHECK(B > 0 && Tn > 0 && To > 0);
    hipLaunchKernelGGL(vp_mix_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dout, loc, mk_bn(loc_stats, loc_g, loc_b, inv_n_loc), gate,
                       mk_bn(gate_stats, gate_g, gate_b, inv_n_glob), dloc, dgate, demb, dres_acc_or_null, Tn, To);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_bn_bwd_reduce(const float* dyhat, const float* raw, const double* stats, const float* gamma, const float* beta, float inv_n, double* sums, int B,
                          int T, void* stream) {
    VP_CHECK(B > 0 && T > 0);
    hipLaunchKernelGGL(vp_bn_bwd_reduce_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dyhat, raw, mk_bn(stats, gamma, beta, inv_n), sums, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_dwconv_bwd(const float* dyhat, const float* raw, const double* out_stats, const float* out_gamma, const float* out_beta, float out_inv_n,
                       const double* sums, float inv_n_all, int batch_stats, const float* src, const double* in_stats, const float* in_gamma,
                       const float* in_beta, float in_inv_n, int in_act, float in_slope, const float* w, float* dW, float* dbias_or_null,
                       float* dsrc_or_null, int accumulate, float* dslope_or_null, int B, int Tin, int Tout, int stride, void* stream) {
    VP_CHECK(B > 0 && Tin > 0 && Tout > 0 && Tout <= 4096 && (stride == 1 || stride == 2));
    hipLaunchKernelGGL(vp_dwconv_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dyhat, raw, mk_bn(out_stats, out_gamma, out_beta, out_inv_n), sums,
                       inv_n_all, batch_stats, src, mk_bn(in_stats, in_gamma, in_beta, in_inv_n), in_act, in_slope, w, dW, dbias_or_null, dsrc_or_null,
                       accumulate, dslope_or_null, Tin, Tout, stride);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_pool_bwd(const float* dg, float* const* d, int T0, int T1, int T2, int T3, int B, int Tg, void* stream) {
    VP_CHECK(B > 0 && Tg > 0);
    const int T[4] = {T0, T1, T2, T3};
    VPoolB p;
    for (int i = 0; i < 4; ++i) p.d[i] = d[i], p.T[i] = T[i];
    hipLaunchKernelGGL(vp_pool_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dg, p, Tg);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_vp_gate_proj_bwd(const float* dyhat, const float* y, const double* y_stats, const float* y_gamma, const float* y_beta, float y_inv_n,
                          const double* sums, float inv_n_all, int batch_stats, const float* dout, const float* x, const float* r, const float* gw,
                          const float* gb, float gslope, const float* Wp, float* dWp, float* dbp, float* dgw, float* dgb, float* dgslope, float* dx, int B,
                          int T, void* stream) {
    VP_CHECK(B > 0 && T > 0 && T <= 4096);
    hipLaunchKernelGGL(vp_gate_proj_bwd_kernel, dim3(B, 8), dim3(256), 0, (hipStream_t)stream, dyhat, y, mk_bn(y_stats, y_gamma, y_beta, y_inv_n), sums,
                       inv_n_all, batch_stats, dout, x, r, gw, gb, gslope, Wp, dWp, dbp, dgw, dgb, dgslope, dx, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
