// GlobalAttention of the VP (video) block in the TRAINING step: MultiHeadSelfAttention (LayerNorm, positional
// encoding, nn.MultiheadAttention with 8 heads, dropout, LayerNorm, DropPath; layers/attention.py:28-73) + FeedForwardNetwork (1x1 conv
// 64 -> 128 + gLN, depth-wise k = 3 conv + bias + ReLU, DropPath, 1x1 conv 128 -> 64 + gLN, DropPath; layers/conv_layers.py:218-259) on the
// pooled [B][64][Tg] tensor of the 1-D TDANetBlock (Tg = 7 tokens for 2 s, 13 for 4 s), forward and adjoint, one workgroup per utterance.
//
//   rtfs_vp_attn_fwd   g -> out                                   (the eval kernel csrc/vp.hip holds the same arithmetic inside vp_block_kernel)
//   rtfs_vp_attn_bwd   (g, d out) -> d g, parameter gradients     (recomputes the forward into LDS: nothing but g and the masks is saved)
//
// Stochastic layers: the caller supplies the multiplicative keep-masks (0 or 1 / keep_prob) of one step - attention-probability dropout
// [8][Tg][Tg], element dropout of the attention output [Tg][64], and the three per-utterance DropPath factors - so forward and adjoint see
// the same draw and the kernels stay deterministic functions of their arguments; a null mask pointer means "no dropout" (eval, p = 0).
// Parameter gradients of the B workgroups meet in one [34176]-float buffer through fp32 atomics (the order of the B terms varies).
#include "common.h"

namespace rtfs {

constexpr int AH = 64, AF = 128, AHEADS = 8, AHD = 8, AMAXT = 16;

struct VaOff {  // float offsets into the packed parameter (and gradient) buffer; rtfs_net_amd/models/vp_train.py packs in this order
    static constexpr int ln1g = 0, ln1b = ln1g + AH;
    static constexpr int inw = ln1b + AH, inb = inw + 3 * AH * AH, outw = inb + 3 * AH, outb = outw + AH * AH;
    static constexpr int ln2g = outb + AH, ln2b = ln2g + AH;
    static constexpr int encw = ln2b + AH, encg = encw + AF * AH, encb = encg + AF;
    static constexpr int refw = encb + AF, refb = refw + AF * 3;
    static constexpr int decw = refb + AF, decg = decw + AH * AF, decb = decg + AH;
    static constexpr int total = decb + AH;
};

struct VaMask {  // per-utterance mask block: [8][Tg][Tg] | [Tg][64] | 3 scalars
    __host__ __device__ static int size(int Tg) { return AHEADS * Tg * Tg + Tg * AH + 3; }
};

__device__ __forceinline__ float va_block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// LDS layout (floats) for Tg tokens; the backward-only buffers follow the forward ones
struct VaLds {
    float *G, *XH1, *Y, *QKV, *P, *O, *ZH, *X1, *EH, *E, *R0, *Rd, *DH, *st;  // st: rstd1[16] rstd2[16] rstd_e rstd_d
    float *dX1, *dR, *dE, *dZ, *dO, *DS, *dQKV;
    __device__ VaLds(float* base, int Tg) {
        float* p = base;
        auto take = [&](int n) {
            float* r = p;
            p += n;
            return r;
        };
        G = take(AH * Tg), XH1 = take(AH * Tg), Y = take(AH * Tg), QKV = take(3 * AH * Tg), P = take(AHEADS * Tg * Tg), O = take(AH * Tg);
        ZH = take(AH * Tg), X1 = take(AH * Tg), EH = take(AF * Tg), E = take(AF * Tg), R0 = take(AF * Tg), Rd = take(AF * Tg), DH = take(AH * Tg);
        st = take(64);
        dX1 = take(AH * Tg), dR = take(AF * Tg), dE = take(AF * Tg), dZ = take(AH * Tg), dO = take(AH * Tg), DS = take(AHEADS * Tg * Tg);
        dQKV = take(3 * AH * Tg);
    }
    __host__ __device__ static int floats(int Tg, bool bwd) {
        int n = AH * Tg * 8 + 3 * AH * Tg + AHEADS * Tg * Tg + AF * Tg * 4 + 64;
        if (bwd) n += AH * Tg * 3 + AF * Tg * 2 + AHEADS * Tg * Tg + 3 * AH * Tg;
        return n;
    }
};

// dot product of a global weight row (N floats, 16-byte aligned) with an LDS vector x[k * xs]: all loads first (csrc/vp.hip dot_row)
template <int N>
__device__ __forceinline__ float va_dot_row(const float* __restrict__ wrow, const float* x, int xs, float init) {
    float4 wv[N / 4];
#pragma unroll
    for (int q = 0; q < N / 4; ++q) wv[q] = ld4(wrow + 4 * q);
    float s = init;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        s = fmaf(wv[q].x, x[(4 * q) * xs], s);
        s = fmaf(wv[q].y, x[(4 * q + 1) * xs], s);
        s = fmaf(wv[q].z, x[(4 * q + 2) * xs], s);
        s = fmaf(wv[q].w, x[(4 * q + 3) * xs], s);
    }
    return s;
}

// The forward of one utterance into LDS (every intermediate the adjoint needs); returns nothing, `out` (may be null) receives the result.
__device__ void va_forward(const VaLds& L, const float* __restrict__ gb, const float* __restrict__ Pm, const float* __restrict__ pe,
                           const float* __restrict__ mk, float* __restrict__ out, int Tg, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* m_attn = mk;                                  // [8][Tg][Tg]
    const float* m_el = mk ? mk + AHEADS * Tg * Tg : nullptr;  // [Tg][64]
    const float dp1 = mk ? mk[AHEADS * Tg * Tg + Tg * AH] : 1.f, dp2 = mk ? mk[AHEADS * Tg * Tg + Tg * AH + 1] : 1.f;
    const float dp3 = mk ? mk[AHEADS * Tg * Tg + Tg * AH + 2] : 1.f;
    for (int idx = tid; idx < AH * Tg; idx += 256) L.G[idx] = gb[idx];
    __syncthreads();
    // LayerNorm over the channels of g^T + positional encoding (attention.py:63-65)
    for (int t = w; t < Tg; t += 4) {
        const float v = L.G[lane * Tg + t];
        const float mean = wave_sum(v) * (1.f / 64.f);
        const float d = v - mean;
        const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.f / 64.f) + kEps);
        const float xh = d * rstd;
        L.XH1[t * AH + lane] = xh;
        L.Y[t * AH + lane] = fmaf(xh, Pm[VaOff::ln1g + lane], Pm[VaOff::ln1b + lane]) + pe[t * AH + lane];
        if (lane == 0) L.st[t] = rstd;
    }
    __syncthreads();
    for (int idx = tid; idx < Tg * 3 * AH; idx += 256) {  // in-projection
        const int t = idx / (3 * AH), n = idx - t * 3 * AH;
        L.QKV[idx] = va_dot_row<AH>(Pm + VaOff::inw + n * AH, L.Y + t * AH, 1, Pm[VaOff::inb + n]);
    }
    __syncthreads();
    for (int idx = tid; idx < AHEADS * Tg; idx += 256) {  // per (head, query): softmax(q k^T / sqrt(8)), dropout on the probabilities, . v
        const int h = idx / Tg, tq = idx - h * Tg;
        const float* q = L.QKV + tq * 3 * AH + h * AHD;
        float sc[AMAXT], mx = -1e30f;
        for (int tk = 0; tk < Tg; ++tk) {
            const float* kx = L.QKV + tk * 3 * AH + AH + h * AHD;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < AHD; ++e) s = fmaf(q[e], kx[e], s);
            sc[tk] = s * 0.35355339059327373f;
            mx = fmaxf(mx, sc[tk]);
        }
        float den = 0.f;
        for (int tk = 0; tk < Tg; ++tk) sc[tk] = __expf(sc[tk] - mx), den += sc[tk];
        const float inv = 1.0f / den;
        float* prow = L.P + (h * Tg + tq) * Tg;
        for (int tk = 0; tk < Tg; ++tk) {
            const float p = sc[tk] * inv;
            prow[tk] = p;
            sc[tk] = m_attn ? p * m_attn[(h * Tg + tq) * Tg + tk] : p;
        }
#pragma unroll
        for (int e = 0; e < AHD; ++e) {
            float o = 0.f;
            for (int tk = 0; tk < Tg; ++tk) o = fmaf(sc[tk], L.QKV[tk * 3 * AH + 2 * AH + h * AHD + e], o);
            L.O[tq * AH + h * AHD + e] = o;
        }
    }
    __syncthreads();
    // out-projection, dropout, + residual (Y), LayerNorm2, transpose back, DropPath, + block residual g  -> X1 [64][Tg]
    for (int t = w; t < Tg; t += 4) {
        float a = va_dot_row<AH>(Pm + VaOff::outw + lane * AH, L.O + t * AH, 1, Pm[VaOff::outb + lane]);
        if (m_el) a *= m_el[t * AH + lane];
        const float v = a + L.Y[t * AH + lane];
        const float mean = wave_sum(v) * (1.f / 64.f);
        const float d = v - mean;
        const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.f / 64.f) + kEps);
        const float zh = d * rstd;
        L.ZH[t * AH + lane] = zh;
        L.X1[lane * Tg + t] = fmaf(fmaf(zh, Pm[VaOff::ln2g + lane], Pm[VaOff::ln2b + lane]), dp1, L.G[lane * Tg + t]);
        if (lane == 0) L.st[16 + t] = rstd;
    }
    __syncthreads();
    // FFN encoder 64 -> 128 (no bias) + gLN over (128, Tg)
    float ls = 0.f, lq = 0.f;
    for (int idx = tid; idx < AF * Tg; idx += 256) {
        const int n = idx / Tg, t = idx - n * Tg;
        const float s = va_dot_row<AH>(Pm + VaOff::encw + n * AH, L.X1 + t, Tg, 0.f);
        L.EH[idx] = s;
        ls += s, lq = fmaf(s, s, lq);
    }
    {
        const float n = (float)(AF * Tg);
        const float mean = va_block_sum(ls, red) / n;
        const float var = fmaxf(va_block_sum(lq, red) / n - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + kEps);
        for (int idx = tid; idx < AF * Tg; idx += 256) {
            const int c = idx / Tg;
            const float eh = (L.EH[idx] - mean) * rstd;
            L.EH[idx] = eh;
            L.E[idx] = fmaf(eh, Pm[VaOff::encg + c], Pm[VaOff::encb + c]);
        }
        if (tid == 0) L.st[32] = rstd;
    }
    __syncthreads();
    for (int idx = tid; idx < AF * Tg; idx += 256) {  // refiner: depth-wise k = 3 ('same') + bias + ReLU, then DropPath
        const int c = idx / Tg, t = idx - c * Tg;
        float s = Pm[VaOff::refb + c];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = t - 1 + k;
            if (p >= 0 && p < Tg) s = fmaf(Pm[VaOff::refw + c * 3 + k], L.E[c * Tg + p], s);
        }
        L.R0[idx] = s;
        L.Rd[idx] = fmaxf(s, 0.f) * dp2;
    }
    __syncthreads();
    ls = 0.f, lq = 0.f;
    for (int idx = tid; idx < AH * Tg; idx += 256) {  // decoder 128 -> 64 (no bias) + gLN over (64, Tg)
        const int c = idx / Tg, t = idx - c * Tg;
        const float s = va_dot_row<AF>(Pm + VaOff::decw + c * AF, L.Rd + t, Tg, 0.f);
        L.DH[idx] = s;
        ls += s, lq = fmaf(s, s, lq);
    }
    {
        const float n = (float)(AH * Tg);
        const float mean = va_block_sum(ls, red) / n;
        const float var = fmaxf(va_block_sum(lq, red) / n - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + kEps);
        for (int idx = tid; idx < AH * Tg; idx += 256) {
            const int c = idx / Tg;
            const float dh = (L.DH[idx] - mean) * rstd;
            L.DH[idx] = dh;
            if (out) out[idx] = fmaf(fmaf(dh, Pm[VaOff::decg + c], Pm[VaOff::decb + c]), dp3, L.X1[idx]);
        }
        if (tid == 0) L.st[33] = rstd;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void vp_attn_fwd_kernel(const float* __restrict__ g, const float* __restrict__ Pm, const float* __restrict__ pe,
                                                          const float* __restrict__ masks, float* __restrict__ out, int Tg) {
    extern __shared__ __attribute__((aligned(16))) float va_lds[];
    __shared__ float red[4];
    const VaLds L(va_lds, Tg);
    const int b = blockIdx.x;
    va_forward(L, g + (size_t)b * AH * Tg, Pm, pe, masks ? masks + (size_t)b * VaMask::size(Tg) : nullptr, out + (size_t)b * AH * Tg, Tg, red);
}

// GlobalAttention in EVAL mode for more than AMAXT pooled tokens (utterances longer than 5.1 s: Tg = 27 at 8.5 s, 188 at 120 s; the reference has no length
// limit, attention.py:28-73 + conv_layers.py:218-259).  Same arithmetic as va_forward without the stochastic layers; the per-token intermediates live in
// a global workspace instead of LDS (one workgroup per utterance: a workgroup's own global writes are visible to it after a barrier), nothing is
// kept for an adjoint, and the softmax walks the keys twice (maximum; exponentials, their sum and the weighted values together) instead of holding
// a row of scores in registers.  work: [B][704 Tg] floats = Y [Tg][64] | QKV [Tg][192] | O [Tg][64] | X1 [64][Tg] | E [128][Tg] | Rd [128][Tg] | DH [64][Tg].
__global__ __launch_bounds__(256) void vp_attn_long_fwd_kernel(const float* __restrict__ g, const float* __restrict__ Pm, const float* __restrict__ pe,
                                                               float* __restrict__ out, float* __restrict__ work, int Tg) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* gb = g + (size_t)b * AH * Tg;
    float* ob = out + (size_t)b * AH * Tg;
    float* Y = work + (size_t)b * 704 * Tg;
    float* QKV = Y + AH * Tg;
    float* O = QKV + 3 * AH * Tg;
    float* X1 = O + AH * Tg;
    float* E = X1 + AH * Tg;
    float* Rd = E + AF * Tg;
    float* DH = Rd + AF * Tg;
    for (int t = w; t < Tg; t += 4) {  // LayerNorm over the channels of g^T + positional encoding
        const float v = gb[lane * Tg + t];
        const float mean = wave_sum(v) * (1.f / 64.f);
        const float d = v - mean;
        const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.f / 64.f) + kEps);
        Y[t * AH + lane] = fmaf(d * rstd, Pm[VaOff::ln1g + lane], Pm[VaOff::ln1b + lane]) + pe[t * AH + lane];
    }
    __syncthreads();
    for (int idx = tid; idx < Tg * 3 * AH; idx += 256) {  // in-projection
        const int t = idx / (3 * AH), n = idx - t * 3 * AH;
        QKV[idx] = va_dot_row<AH>(Pm + VaOff::inw + n * AH, Y + t * AH, 1, Pm[VaOff::inb + n]);
    }
    __syncthreads();
    for (int idx = tid; idx < AHEADS * Tg; idx += 256) {  // per (head, query): softmax(q k^T / sqrt(8)) . v
        const int h = idx / Tg, tq = idx - h * Tg;
        float q[AHD];
#pragma unroll
        for (int e = 0; e < AHD; ++e) q[e] = QKV[tq * 3 * AH + h * AHD + e] * 0.35355339059327373f;
        float mx = -1e30f;
        for (int tk = 0; tk < Tg; ++tk) {
            const float4 k0 = ld4(QKV + tk * 3 * AH + AH + h * AHD), k1 = ld4(QKV + tk * 3 * AH + AH + h * AHD + 4);
            const float sc = q[0] * k0.x + q[1] * k0.y + q[2] * k0.z + q[3] * k0.w + q[4] * k1.x + q[5] * k1.y + q[6] * k1.z + q[7] * k1.w;
            mx = fmaxf(mx, sc);
        }
        float den = 0.f, o[AHD];
#pragma unroll
        for (int e = 0; e < AHD; ++e) o[e] = 0.f;
        for (int tk = 0; tk < Tg; ++tk) {
            const float* row = QKV + tk * 3 * AH + h * AHD;
            const float4 k0 = ld4(row + AH), k1 = ld4(row + AH + 4), v0 = ld4(row + 2 * AH), v1 = ld4(row + 2 * AH + 4);
            const float sc = q[0] * k0.x + q[1] * k0.y + q[2] * k0.z + q[3] * k0.w + q[4] * k1.x + q[5] * k1.y + q[6] * k1.z + q[7] * k1.w;
            const float p = __expf(sc - mx);
            den += p;
            o[0] = fmaf(p, v0.x, o[0]), o[1] = fmaf(p, v0.y, o[1]), o[2] = fmaf(p, v0.z, o[2]), o[3] = fmaf(p, v0.w, o[3]);
            o[4] = fmaf(p, v1.x, o[4]), o[5] = fmaf(p, v1.y, o[5]), o[6] = fmaf(p, v1.z, o[6]), o[7] = fmaf(p, v1.w, o[7]);
        }
        const float inv = 1.0f / den;
#pragma unroll
        for (int e = 0; e < AHD; ++e) O[tq * AH + h * AHD + e] = o[e] * inv;
    }
    __syncthreads();
    for (int t = w; t < Tg; t += 4) {  // out-projection + residual (Y), LayerNorm2, transpose back, + block residual g -> X1 [64][Tg]
        const float v = va_dot_row<AH>(Pm + VaOff::outw + lane * AH, O + t * AH, 1, Pm[VaOff::outb + lane]) + Y[t * AH + lane];
        const float mean = wave_sum(v) * (1.f / 64.f);
        const float d = v - mean;
        const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.f / 64.f) + kEps);
        X1[lane * Tg + t] = fmaf(d * rstd, Pm[VaOff::ln2g + lane], Pm[VaOff::ln2b + lane]) + gb[lane * Tg + t];
    }
    __syncthreads();
    float ls = 0.f, lq = 0.f;
    for (int idx = tid; idx < AF * Tg; idx += 256) {  // FFN encoder 64 -> 128 (no bias) + gLN over (128, Tg)
        const int n = idx / Tg, t = idx - n * Tg;
        const float v = va_dot_row<AH>(Pm + VaOff::encw + n * AH, X1 + t, Tg, 0.f);
        E[idx] = v;
        ls += v, lq = fmaf(v, v, lq);
    }
    {
        const float n = (float)(AF * Tg);
        const float mean = va_block_sum(ls, red) / n;
        const float var = fmaxf(va_block_sum(lq, red) / n - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + kEps);
        for (int idx = tid; idx < AF * Tg; idx += 256) {  // (every thread normalises the elements it wrote)
            const int c = idx / Tg;
            E[idx] = fmaf((E[idx] - mean) * rstd, Pm[VaOff::encg + c], Pm[VaOff::encb + c]);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < AF * Tg; idx += 256) {  // refiner: depth-wise k = 3 ('same') + bias + ReLU
        const int c = idx / Tg, t = idx - c * Tg;
        float v = Pm[VaOff::refb + c];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = t - 1 + k;
            if (p >= 0 && p < Tg) v = fmaf(Pm[VaOff::refw + c * 3 + k], E[c * Tg + p], v);
        }
        Rd[idx] = fmaxf(v, 0.f);
    }
    __syncthreads();
    ls = 0.f, lq = 0.f;
    for (int idx = tid; idx < AH * Tg; idx += 256) {  // decoder 128 -> 64 (no bias) + gLN over (64, Tg) + the FFN residual
        const int c = idx / Tg, t = idx - c * Tg;
        const float v = va_dot_row<AF>(Pm + VaOff::decw + c * AF, Rd + t, Tg, 0.f);
        DH[idx] = v;
        ls += v, lq = fmaf(v, v, lq);
    }
    {
        const float n = (float)(AH * Tg);
        const float mean = va_block_sum(ls, red) / n;
        const float var = fmaxf(va_block_sum(lq, red) / n - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + kEps);
        for (int idx = tid; idx < AH * Tg; idx += 256) {
            const int c = idx / Tg;
            ob[idx] = fmaf((DH[idx] - mean) * rstd, Pm[VaOff::decg + c], Pm[VaOff::decb + c]) + X1[idx];
        }
    }
}

__global__ __launch_bounds__(256) void vp_attn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ Pm, const float* __restrict__ pe,
                                                          const float* __restrict__ masks, const float* __restrict__ dout, float* __restrict__ dg,
                                                          float* __restrict__ dP, int Tg) {
    extern __shared__ __attribute__((aligned(16))) float va_lds[];
    __shared__ float red[4];
    const VaLds L(va_lds, Tg);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* mk = masks ? masks + (size_t)b * VaMask::size(Tg) : nullptr;
    va_forward(L, g + (size_t)b * AH * Tg, Pm, pe, mk, nullptr, Tg, red);
    const float* m_attn = mk;
    const float* m_el = mk ? mk + AHEADS * Tg * Tg : nullptr;
    const float dp1 = mk ? mk[AHEADS * Tg * Tg + Tg * AH] : 1.f, dp2 = mk ? mk[AHEADS * Tg * Tg + Tg * AH + 1] : 1.f;
    const float dp3 = mk ? mk[AHEADS * Tg * Tg + Tg * AH + 2] : 1.f;
    const float* dob = dout + (size_t)b * AH * Tg;
    // ---- decoder gLN: dD = dout * dp3 (dX1 starts as dout: the FFN residual)
    float s1 = 0.f, s2 = 0.f;
    for (int idx = tid; idx < AH * Tg; idx += 256) {
        const int c = idx / Tg;
        const float d = dob[idx];
        L.dX1[idx] = d;
        const float u = d * dp3 * Pm[VaOff::decg + c];
        s1 += u, s2 = fmaf(u, L.DH[idx], s2);
    }
    if (tid < AH) {  // d gamma / d beta of the decoder norm: channel tid, sum over t
        float a = 0.f, bb = 0.f;
        for (int t = 0; t < Tg; ++t) {
            const float d = dob[tid * Tg + t] * dp3;
            a = fmaf(d, L.DH[tid * Tg + t], a), bb += d;
        }
        atomicAdd(dP + VaOff::decg + tid, a);
        atomicAdd(dP + VaOff::decb + tid, bb);
    }
    {
        const float n = (float)(AH * Tg);
        const float m1 = va_block_sum(s1, red) / n, m2 = va_block_sum(s2, red) / n, rstd = L.st[33];
        for (int idx = tid; idx < AH * Tg; idx += 256) {
            const int c = idx / Tg;
            const float u = L.dX1[idx] * dp3 * Pm[VaOff::decg + c];
            L.dZ[idx] = rstd * (u - m1 - L.DH[idx] * m2);  // dD0 [64][Tg] (dZ's buffer is free until the attention part)
        }
    }
    __syncthreads();
    // ---- decoder 1x1: dWdec[c][n] += sum_t dD0[c][t] Rd[n][t];  dRd[n][t] = sum_c Wdec[c][n] dD0[c][t]
    for (int idx = tid; idx < AH * AF; idx += 256) {
        const int c = idx / AF, n = idx - c * AF;
        float a = 0.f;
        for (int t = 0; t < Tg; ++t) a = fmaf(L.dZ[c * Tg + t], L.Rd[n * Tg + t], a);
        atomicAdd(dP + VaOff::decw + idx, a);
    }
    for (int idx = tid; idx < AF * Tg; idx += 256) {
        const int n = idx / Tg, t = idx - n * Tg;
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < AH; ++c) a = fmaf(Pm[VaOff::decw + c * AF + n], L.dZ[c * Tg + t], a);
        L.dR[idx] = L.R0[idx] > 0.f ? a * dp2 : 0.f;  // through DropPath and ReLU: dR0
    }
    __syncthreads();
    // ---- refiner (depth-wise k = 3 + bias): parameter gradients, dE
    if (tid < AF) {
        float db = 0.f, dw[3] = {0.f, 0.f, 0.f};
        for (int t = 0; t < Tg; ++t) {
            const float d = L.dR[tid * Tg + t];
            db += d;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int p = t - 1 + k;
                if (p >= 0 && p < Tg) dw[k] = fmaf(d, L.E[tid * Tg + p], dw[k]);
            }
        }
        atomicAdd(dP + VaOff::refb + tid, db);
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(dP + VaOff::refw + tid * 3 + k, dw[k]);
    }
    s1 = 0.f, s2 = 0.f;
    for (int idx = tid; idx < AF * Tg; idx += 256) {
        const int c = idx / Tg, t = idx - c * Tg;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // out position q = t + 1 - k took input t with tap k
            const int q = t + 1 - k;
            if (q >= 0 && q < Tg) a = fmaf(Pm[VaOff::refw + c * 3 + k], L.dR[c * Tg + q], a);
        }
        L.dE[idx] = a;
        const float u = a * Pm[VaOff::encg + c];
        s1 += u, s2 = fmaf(u, L.EH[idx], s2);
    }
    __syncthreads();
    if (tid < AF) {
        float a = 0.f, bb = 0.f;
        for (int t = 0; t < Tg; ++t) a = fmaf(L.dE[tid * Tg + t], L.EH[tid * Tg + t], a), bb += L.dE[tid * Tg + t];
        atomicAdd(dP + VaOff::encg + tid, a);
        atomicAdd(dP + VaOff::encb + tid, bb);
    }
    {
        const float n = (float)(AF * Tg);
        const float m1 = va_block_sum(s1, red) / n, m2 = va_block_sum(s2, red) / n, rstd = L.st[32];
        for (int idx = tid; idx < AF * Tg; idx += 256) {
            const int c = idx / Tg;
            L.dE[idx] = rstd * (L.dE[idx] * Pm[VaOff::encg + c] - m1 - L.EH[idx] * m2);  // dE0
        }
    }
    __syncthreads();
    // ---- encoder 1x1: dWenc[n][c] += sum_t dE0[n][t] X1[c][t];  dX1[c][t] += sum_n Wenc[n][c] dE0[n][t]
    for (int idx = tid; idx < AF * AH; idx += 256) {
        const int n = idx / AH, c = idx - n * AH;
        float a = 0.f;
        for (int t = 0; t < Tg; ++t) a = fmaf(L.dE[n * Tg + t], L.X1[c * Tg + t], a);
        atomicAdd(dP + VaOff::encw + idx, a);
    }
    for (int idx = tid; idx < AH * Tg; idx += 256) {
        const int c = idx / Tg, t = idx - c * Tg;
        float a = L.dX1[idx];
#pragma unroll 8
        for (int n = 0; n < AF; ++n) a = fmaf(Pm[VaOff::encw + n * AH + c], L.dE[n * Tg + t], a);
        L.dX1[idx] = a;  // gradient of X1 = MHSA output: its `+ res` term is d g, the other goes through DropPath into LayerNorm2
    }
    __syncthreads();
    // ---- LayerNorm2 (per token over channels): parameter gradients, dZ [Tg][64]
    if (tid < AH) {
        float a = 0.f, bb = 0.f;
        for (int t = 0; t < Tg; ++t) {
            const float d = L.dX1[tid * Tg + t] * dp1;
            a = fmaf(d, L.ZH[t * AH + tid], a), bb += d;
        }
        atomicAdd(dP + VaOff::ln2g + tid, a);
        atomicAdd(dP + VaOff::ln2b + tid, bb);
    }
    __syncthreads();  // (dZ's buffer held dD0 until here)
    for (int t = w; t < Tg; t += 4) {
        const float u = L.dX1[lane * Tg + t] * dp1 * Pm[VaOff::ln2g + lane], zh = L.ZH[t * AH + lane];
        const float m1 = wave_sum(u) * (1.f / 64.f), m2 = wave_sum(u * zh) * (1.f / 64.f);
        L.dZ[t * AH + lane] = L.st[16 + t] * (u - m1 - zh * m2);  // = dY (residual) ; dA = dZ * m_el
    }
    __syncthreads();
    // ---- out-projection: dWout[c][k] += sum_t dA[t][c] O[t][k]; dbout; dO[t][k] = sum_c dA[t][c] Wout[c][k]
    for (int idx = tid; idx < AH * AH; idx += 256) {
        const int c = idx / AH, k = idx - c * AH;
        float a = 0.f;
        for (int t = 0; t < Tg; ++t) a = fmaf(L.dZ[t * AH + c] * (m_el ? m_el[t * AH + c] : 1.f), L.O[t * AH + k], a);
        atomicAdd(dP + VaOff::outw + idx, a);
    }
    if (tid < AH) {
        float a = 0.f;
        for (int t = 0; t < Tg; ++t) a += L.dZ[t * AH + tid] * (m_el ? m_el[t * AH + tid] : 1.f);
        atomicAdd(dP + VaOff::outb + tid, a);
    }
    for (int idx = tid; idx < Tg * AH; idx += 256) {
        const int t = idx / AH, k = idx - t * AH;
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < AH; ++c) a = fmaf(L.dZ[t * AH + c] * (m_el ? m_el[t * AH + c] : 1.f), Pm[VaOff::outw + c * AH + k], a);
        L.dO[idx] = a;
    }
    __syncthreads();
    // ---- attention core.  Pass 1, thread = (head, query): dS row (softmax adjoint incl. the probability dropout), dq
    for (int idx = tid; idx < AHEADS * Tg; idx += 256) {
        const int h = idx / Tg, tq = idx - h * Tg;
        const float* prow = L.P + (h * Tg + tq) * Tg;
        float dp[AMAXT], sdot = 0.f;
        for (int tk = 0; tk < Tg; ++tk) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < AHD; ++e) a = fmaf(L.dO[tq * AH + h * AHD + e], L.QKV[tk * 3 * AH + 2 * AH + h * AHD + e], a);
            if (m_attn) a *= m_attn[(h * Tg + tq) * Tg + tk];
            dp[tk] = a;
            sdot = fmaf(prow[tk], a, sdot);
        }
        float dq[AHD];
#pragma unroll
        for (int e = 0; e < AHD; ++e) dq[e] = 0.f;
        for (int tk = 0; tk < Tg; ++tk) {
            const float ds = prow[tk] * (dp[tk] - sdot) * 0.35355339059327373f;
            L.DS[(h * Tg + tq) * Tg + tk] = ds;
#pragma unroll
            for (int e = 0; e < AHD; ++e) dq[e] = fmaf(ds, L.QKV[tk * 3 * AH + AH + h * AHD + e], dq[e]);
        }
#pragma unroll
        for (int e = 0; e < AHD; ++e) L.dQKV[tq * 3 * AH + h * AHD + e] = dq[e];
    }
    __syncthreads();
    // Pass 2, thread = (head, key): dk = sum_q dS[q][k] q_q,  dv = sum_q Pd[q][k] dO_q
    for (int idx = tid; idx < AHEADS * Tg; idx += 256) {
        const int h = idx / Tg, tk = idx - h * Tg;
        float dk[AHD], dv[AHD];
#pragma unroll
        for (int e = 0; e < AHD; ++e) dk[e] = 0.f, dv[e] = 0.f;
        for (int tq = 0; tq < Tg; ++tq) {
            const float ds = L.DS[(h * Tg + tq) * Tg + tk];
            float pd = L.P[(h * Tg + tq) * Tg + tk];
            if (m_attn) pd *= m_attn[(h * Tg + tq) * Tg + tk];
#pragma unroll
            for (int e = 0; e < AHD; ++e) {
                dk[e] = fmaf(ds, L.QKV[tq * 3 * AH + h * AHD + e], dk[e]);
                dv[e] = fmaf(pd, L.dO[tq * AH + h * AHD + e], dv[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < AHD; ++e) L.dQKV[tk * 3 * AH + AH + h * AHD + e] = dk[e], L.dQKV[tk * 3 * AH + 2 * AH + h * AHD + e] = dv[e];
    }
    __syncthreads();
    // ---- in-projection: dWin[n][c] += sum_t dQKV[t][n] Y[t][c]; dbin; dY[t][c] += sum_n dQKV[t][n] Win[n][c]
    for (int idx = tid; idx < 3 * AH * AH; idx += 256) {
        const int n = idx / AH, c = idx - n * AH;
        float a = 0.f;
        for (int t = 0; t < Tg; ++t) a = fmaf(L.dQKV[t * 3 * AH + n], L.Y[t * AH + c], a);
        atomicAdd(dP + VaOff::inw + idx, a);
    }
    if (tid < 3 * AH) {
        float a = 0.f;
        for (int t = 0; t < Tg; ++t) a += L.dQKV[t * 3 * AH + tid];
        atomicAdd(dP + VaOff::inb + tid, a);
    }
    for (int idx = tid; idx < Tg * AH; idx += 256) {
        const int t = idx / AH, c = idx - t * AH;
        float a = L.dZ[idx];
#pragma unroll 8
        for (int n = 0; n < 3 * AH; ++n) a = fmaf(L.dQKV[t * 3 * AH + n], Pm[VaOff::inw + n * AH + c], a);
        L.dO[idx] = a;  // dY (dO's buffer is free)
    }
    __syncthreads();
    // ---- LayerNorm1: parameter gradients, d g = dX1 (block residual) + LN adjoint
    if (tid < AH) {
        float a = 0.f, bb = 0.f;
        for (int t = 0; t < Tg; ++t) a = fmaf(L.dO[t * AH + tid], L.XH1[t * AH + tid], a), bb += L.dO[t * AH + tid];
        atomicAdd(dP + VaOff::ln1g + tid, a);
        atomicAdd(dP + VaOff::ln1b + tid, bb);
    }
    float* dgb = dg + (size_t)b * AH * Tg;
    for (int t = w; t < Tg; t += 4) {
        const float u = L.dO[t * AH + lane] * Pm[VaOff::ln1g + lane], xh = L.XH1[t * AH + lane];
        const float m1 = wave_sum(u) * (1.f / 64.f), m2 = wave_sum(u * xh) * (1.f / 64.f);
        dgb[lane * Tg + t] = L.dX1[lane * Tg + t] + L.st[t] * (u - m1 - xh * m2);
    }
}

}  // namespace rtfs

using namespace rtfs;

static int va_set_lds(const void* fn, bool* flags) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return RTFS_ELAUNCH;
    if (!flags[dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess) return RTFS_ELAUNCH;
        flags[dev] = true;
    }
    return RTFS_OK;
}

extern "C" {

int rtfs_vp_attn_param_count(void) { return VaOff::total; }
int rtfs_vp_attn_mask_size(int Tg) { return VaMask::size(Tg); }

// g, out: [B][64][Tg] (2 <= Tg <= 16); params: rtfs_vp_attn_param_count() floats in the order of VaOff; pe: [>= Tg][64] rows of the positional
// encoding; masks: [B][rtfs_vp_attn_mask_size(Tg)] multiplicative keep-masks or NULL (no dropout).
int rtfs_vp_attn_fwd(const float* g, const float* params, const float* pe, const float* masks_or_null, float* out, int B, int Tg, void* stream) {
    if (B <= 0 || Tg < 2 || Tg > AMAXT) return RTFS_EINVAL;
    static bool set[16] = {};
    if (va_set_lds(reinterpret_cast<const void*>(vp_attn_fwd_kernel), set) != RTFS_OK) return RTFS_ELAUNCH;
    hipLaunchKernelGGL(vp_attn_fwd_kernel, dim3(B), dim3(256), VaLds::floats(Tg, false) * sizeof(float), (hipStream_t)stream, g, params, pe, masks_or_null,
                       out, Tg);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// eval mode, 2 <= Tg <= 1024 pooled tokens (no masks, nothing kept for an adjoint): work = [B][rtfs_vp_attn_long_work_floats(Tg)] floats of scratch
int rtfs_vp_attn_long_work_floats(int Tg) { return 704 * Tg; }
int rtfs_vp_attn_long_fwd(const float* g, const float* params, const float* pe, float* out, float* work, int B, int Tg, void* stream) {
    if (B <= 0 || Tg < 2 || Tg > 1024) return RTFS_EINVAL;
    hipLaunchKernelGGL(vp_attn_long_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, g, params, pe, out, work, Tg);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// adjoint: dg [B][64][Tg] is written; dparams [rtfs_vp_attn_param_count()] is ACCUMULATED into (zero it first)
int rtfs_vp_attn_bwd(const float* g, const float* params, const float* pe, const float* masks_or_null, const float* dout, float* dg, float* dparams, int B,
                     int Tg, void* stream) {
    if (B <= 0 || Tg < 2 || Tg > AMAXT) return RTFS_EINVAL;
    static bool set[16] = {};
    if (va_set_lds(reinterpret_cast<const void*>(vp_attn_bwd_kernel), set) != RTFS_OK) return RTFS_ELAUNCH;
    hipLaunchKernelGGL(vp_attn_bwd_kernel, dim3(B), dim3(256), VaLds::floats(Tg, true) * sizeof(float), (hipStream_t)stream, g, params, pe, masks_or_null,
                       dout, dg, dparams, Tg);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
