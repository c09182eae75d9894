// VP (video processing) block of the refinement module, eval mode, as ONE kernel:
// TDANetBlock with is2d = False (separators/tdanet.py:106-133, config yaml:74-92) = gateway -> projection -> 4 depth-wise
// down-samplings -> pooled sum -> GlobalAttention (MHSA with 8 heads + FFN, layers/attention.py:28-73,192-220) -> 4 + 3
// InjectionMultiSum units (layers/fusion.py:54-69) -> residual conv + gateway residual.
//
// The whole block of one utterance is ~13 MFLOP on [64 x <=100] activations: one workgroup per utterance keeps every
// intermediate in LDS (<= 158 KB at Tv = 100) and replaces the ~100 PyTorch launches of the glue path.  The two "large" maps
// (512 -> 64 projection, 64 -> 512 residual conv) run lane = time step: the activation column sits in registers, the weights
// are wave-uniform scalar loads, x / out are read / written coalesced along t.  BatchNorm1d is folded to (scale, shift) on the
// host (eval statistics); training keeps the PyTorch glue (batch statistics / dropout).
//
// params: packed fp32 in the order of vp_off below (rtfs_net_amd/models/hip_path.py: pack_vp_params builds it).
#include "common.h"

namespace rtfs {

constexpr int VIN = 512, VH = 64, VF = 128, VHEADS = 8, VHD = 8, VMAXT = 100, VSM = 8192;

struct VpOff {  // float offsets into the packed parameter buffer
    static constexpr int gw = 0, gb = gw + VIN, gslope = gb + VIN;
    static constexpr int pw = gslope + 4, ps = pw + VH * VIN, psh = ps + VH, pslope = psh + VH;  // scalars padded to 4: rows stay 16-byte aligned
    static constexpr int down = pslope + 4;                 // 4 x (w[64][3], scale[64], shift[64])
    static constexpr int down_sz = VH * 3 + 2 * VH;
    static constexpr int ln1g = down + 4 * down_sz, ln1b = ln1g + VH;
    static constexpr int inw = ln1b + VH, inb = inw + 3 * VH * VH, outw = inb + 3 * VH, outb = outw + VH * VH;
    static constexpr int ln2g = outb + VH, ln2b = ln2g + VH;
    static constexpr int encw = ln2b + VH, encg = encw + VF * VH, encb = encg + VF;
    static constexpr int refw = encb + VF, refb = refw + VF * 3;
    static constexpr int decw = refb + VF, decg = decw + VH * VF, decb = decg + VH;
    static constexpr int ims = decb + VH;                   // 7 units x 3 convs x (w[64][3], scale[64], shift[64])
    static constexpr int ims_conv = VH * 3 + 2 * VH, ims_unit = 3 * ims_conv;
    static constexpr int rw = ims + 7 * ims_unit, rb = rw + VIN * VH;
    static constexpr int total = rb + VIN;
};

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// depth-wise conv k = 3 at output position t of a [64][Tin] LDS tensor: stride 1 'same' (pad 1,1) or stride 2 pad 1
__device__ __forceinline__ float dw3(const float* in, int c, int Tin, int t, int stride, const float* w) {
    const int c0 = t * stride - 1;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = c0 + k;
        if (p >= 0 && p < Tin) s = fmaf(w[c * 3 + k], in[c * Tin + p], s);
    }
    return s;
}

// InjectionMultiSum with the global branch convolved at its own length then nearest-up-sampled (fusion.py:58-61; equal lengths
// coincide with the other branch): out[c][t] = loc(local)[c][t] * sigmoid(gate(glob))[c][src] + emb(glob)[c][src] (+ res[c][t])
__device__ void ims_unit(const float* P, const float* local, int Tn, const float* glob, int To, const float* res, float* out, float* ge, float* gg) {
    // thread = (channel c, time phase): the 3 x (3 taps, scale, shift) of ITS channel are loaded once into registers - a loop
    // over (c, t) pairs would re-fetch them from global memory for every element, one exposed latency each
    const int c = threadIdx.x & 63, ph = threadIdx.x >> 6;
    float w_[3][3], sc_[3], sh_[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float* q = P + e * VpOff::ims_conv;
#pragma unroll
        for (int k = 0; k < 3; ++k) w_[e][k] = q[c * 3 + k];
        sc_[e] = q[VH * 3 + c], sh_[e] = q[VH * 4 + c];
    }
    auto conv = [&](const float* in, int Tin, int t, int e) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int p = t - 1 + k;
            if (p >= 0 && p < Tin) s = fmaf(w_[e][k], in[c * Tin + p], s);
        }
        return fmaf(sc_[e], s, sh_[e]);
    };
    for (int t = ph; t < To; t += 4) {
        ge[c * To + t] = conv(glob, To, t, 1);
        gg[c * To + t] = sigmoidf_fast(conv(glob, To, t, 2));
    }
    __syncthreads();
    for (int t = ph; t < Tn; t += 4) {
        const int src = nearest_src(t, To, Tn);
        float v = fmaf(conv(local, Tn, t, 0), gg[c * To + src], ge[c * To + src]);
        if (res) v += res[c * Tn + t];
        out[c * Tn + t] = v;
    }
    __syncthreads();
}

// dot product of a global weight row (N floats, 16-byte aligned) with an LDS vector x[k * xs]: all N/4 weight loads are issued
// before the first FMA (a rolled scalar loop pays one L2 latency per term)
template <int N>
__device__ __forceinline__ float dot_row(const float* __restrict__ wrow, const float* x, int xs, float init) {
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

__global__ __launch_bounds__(256) void vp_block_kernel(const float* __restrict__ x, const float* __restrict__ P, const float* __restrict__ pe,
                                                       float* __restrict__ out, int Tv) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: weight rows indexed by w become scalar (SMEM) loads
    int T[4];
    T[0] = Tv;
#pragma unroll
    for (int i = 1; i < 4; ++i) T[i] = (T[i - 1] - 1) / 2 + 1;
    const int Tg = T[3], sumT = T[0] + T[1] + T[2] + T[3];
    // The GlobalAttention stage re-uses the W1..W3 span for 576 Tg floats of temporaries: that fits 3 x 64 x T0 from three frames on; with one or two frames
    // (T0 < 3 Tg) the buffers are spaced for 3 Tg columns instead (round 6: until then the stage overran the span and 1 - 2 frames went to the PyTorch modules)
    const int Tw = max(T[0], 3 * Tg);
    float* DS = lds;                       // ds0..ds3, [64][Ti] each
    float* W1 = DS + VH * sumT;            // three [64][T0] work buffers (capacity [64][Tw])
    float* W2 = W1 + VH * Tw;
    float* W3 = W2 + VH * Tw;
    float* SM = W3 + VH * Tw;              // VSM floats: pooled g + attention / IMS temporaries
    float* dsp[4] = {DS, DS + VH * T[0], DS + VH * (T[0] + T[1]), DS + VH * (T[0] + T[1] + T[2])};
    const float* xb = x + (size_t)b * VIN * Tv;
    const float gslope = P[VpOff::gslope], pslope = P[VpOff::pslope];

    // ---- S1: gateway (dw 1x1 + PReLU) + projection 512 -> 64 + BatchNorm + PReLU -> W1 [64][T0].  lane = t, wave = 16 channels.
    for (int t0 = 0; t0 < Tv; t0 += 64) {
        const int t = t0 + lane;
        const bool ok = t < Tv;
        const int tc = ok ? t : Tv - 1;  // clamped: the loads below stay unconditional (a per-load branch serialises them)
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll 1
        for (int k0 = 0; k0 < VIN; k0 += 32) {
            float g[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) g[k] = xb[(size_t)(k0 + k) * Tv + tc];
#pragma unroll
            for (int k = 0; k < 32; ++k) g[k] = prelu(fmaf(g[k], P[VpOff::gw + k0 + k], P[VpOff::gb + k0 + k]), gslope);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float* wr = P + VpOff::pw + (size_t)(16 * w + j) * VIN + k0;  // wave-uniform: scalar loads
#pragma unroll
                for (int k = 0; k < 32; ++k) acc[j] = fmaf(wr[k], g[k], acc[j]);
            }
        }
        if (ok) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = 16 * w + j;
                W1[c * Tv + t] = prelu(fmaf(P[VpOff::ps + c], acc[j], P[VpOff::psh + c]), pslope);
            }
        }
    }
    __syncthreads();
    // ---- S2: down-sampling chain (depth-wise k = 3 + bias + BatchNorm, stride 1 then 2, 2, 2)
    for (int i = 0; i < 4; ++i) {
        const float* dp = P + VpOff::down + i * VpOff::down_sz;
        const float* in = i == 0 ? W1 : dsp[i - 1];
        const int Tin = i == 0 ? T[0] : T[i - 1], To = T[i], stride = i == 0 ? 1 : 2;
        {
            const int c = tid & 63;
            const float w0 = dp[c * 3], w1 = dp[c * 3 + 1], w2 = dp[c * 3 + 2], dsc = dp[VH * 3 + c], dsh = dp[VH * 4 + c];
            for (int t = tid >> 6; t < To; t += 4) {
                const int p0 = t * stride - 1;
                float sacc = 0.f;
                if (p0 >= 0) sacc = w0 * in[c * Tin + p0];
                if (p0 + 1 < Tin) sacc = fmaf(w1, in[c * Tin + p0 + 1], sacc);
                if (p0 + 2 < Tin) sacc = fmaf(w2, in[c * Tin + p0 + 2], sacc);
                dsp[i][c * To + t] = fmaf(dsc, sacc, dsh);
            }
        }
        __syncthreads();
    }
    // ---- S3: g = sum_i adaptive_avg_pool1d(ds_i, Tg) -> SM[0 .. 64*Tg)
    float* G = SM;
    for (int idx = tid; idx < VH * Tg; idx += 256) {
        const int c = idx / Tg, j = idx - c * Tg;
        float s = 0.f;
        for (int i = 0; i < 4; ++i) {
            const int st = (j * T[i]) / Tg, en = ((j + 1) * T[i] + Tg - 1) / Tg;
            float a = 0.f;
            for (int p = st; p < en; ++p) a += dsp[i][c * T[i] + p];
            s += a / (float)(en - st);
        }
        G[idx] = s;
    }
    __syncthreads();
    // ---- S4: GlobalAttention.  Temporaries in the (now free) W1..W3 span, 576 Tg <= 192 T0 floats: Y [Tg][64] (LN1 + PE, the
    // attention residual), QKV [Tg][192], O [Tg][64], FFN hidden E / R2 [128][Tg]
    float* Y = W1;
    float* QKV = Y + Tg * VH;
    float* O = QKV + Tg * 3 * VH;
    float* E = O + Tg * VH;          // FFN hidden [128][Tg]
    float* R2 = E + VF * Tg;         // FFN refined [128][Tg]
    {
        // LayerNorm over channels of g^T, + positional encoding (attention.py:48-52)
        for (int t = w; t < Tg; t += 4) {
            const float v = G[lane * Tg + t];
            const float mean = wave_sum(v) * (1.f / 64.f);
            const float d = v - mean;
            const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.f / 64.f) + kEps);
            Y[t * VH + lane] = fmaf(d * rstd, P[VpOff::ln1g + lane], P[VpOff::ln1b + lane]) + pe[t * VH + lane];
        }
        __syncthreads();
        // in-projection: QKV[t][n] = Y[t] . Win[n] + bin[n]
        for (int idx = tid; idx < Tg * 3 * VH; idx += 256) {
            const int t = idx / (3 * VH), n = idx - t * 3 * VH;
            QKV[idx] = dot_row<VH>(P + VpOff::inw + n * VH, Y + t * VH, 1, P[VpOff::inb + n]);
        }
        __syncthreads();
        // per (head, query): softmax(q k^T / sqrt(8)) v
        for (int idx = tid; idx < VHEADS * Tg; idx += 256) {
            const int h = idx / Tg, tq = idx - h * Tg;
            const float* q = QKV + tq * 3 * VH + h * VHD;
            float sc[16], mx = -1e30f;
            for (int tk = 0; tk < Tg; ++tk) {
                const float* kx = QKV + tk * 3 * VH + VH + h * VHD;
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < VHD; ++e) s = fmaf(q[e], kx[e], s);
                sc[tk] = s * 0.35355339059327373f;
                mx = fmaxf(mx, sc[tk]);
            }
            float den = 0.f;
            for (int tk = 0; tk < Tg; ++tk) sc[tk] = __expf(sc[tk] - mx), den += sc[tk];
            const float inv = 1.0f / den;
#pragma unroll
            for (int e = 0; e < VHD; ++e) {
                float o = 0.f;
                for (int tk = 0; tk < Tg; ++tk) o = fmaf(sc[tk], QKV[tk * 3 * VH + 2 * VH + h * VHD + e], o);
                O[tq * VH + h * VHD + e] = o * inv;
            }
        }
        __syncthreads();
        // out-projection + residual (Y), LayerNorm2, transpose back, + block residual g  -> G (in place)
        for (int t = w; t < Tg; t += 4) {
            const float v = dot_row<VH>(P + VpOff::outw + lane * VH, O + t * VH, 1, P[VpOff::outb + lane]) + Y[t * VH + lane];
            const float mean = wave_sum(v) * (1.f / 64.f);
            const float d = v - mean;
            const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.f / 64.f) + kEps);
            G[lane * Tg + t] += fmaf(d * rstd, P[VpOff::ln2g + lane], P[VpOff::ln2b + lane]);
        }
        __syncthreads();
        // FFN: encoder 64 -> 128 (no bias) + gLN; refiner dw k = 3 + bias + ReLU; decoder 128 -> 64 + gLN; + residual
        float ls = 0.f, lq = 0.f;
        for (int idx = tid; idx < VF * Tg; idx += 256) {
            const int n = idx / Tg, t = idx - n * Tg;
            const float s = dot_row<VH>(P + VpOff::encw + n * VH, G + t, Tg, 0.f);
            E[idx] = s;
            ls += s, lq = fmaf(s, s, lq);
        }
        {
            const float n = (float)(VF * Tg);
            const float mean = block_sum256(ls, red) / n;
            const float var = fmaxf(block_sum256(lq, red) / n - mean * mean, 0.f);
            const float rstd = 1.0f / sqrtf(var + kEps);
            for (int idx = tid; idx < VF * Tg; idx += 256) {
                const int c = idx / Tg;
                E[idx] = fmaf((E[idx] - mean) * rstd, P[VpOff::encg + c], P[VpOff::encb + c]);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < VF * Tg; idx += 256) {
            const int c = idx / Tg, t = idx - c * Tg;
            float s = P[VpOff::refb + c];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int p = t - 1 + k;
                if (p >= 0 && p < Tg) s = fmaf(P[VpOff::refw + c * 3 + k], E[c * Tg + p], s);
            }
            R2[idx] = fmaxf(s, 0.f);
        }
        __syncthreads();
        ls = 0.f, lq = 0.f;
        float dv[4];  // this thread's decoder outputs (64*Tg <= 1024 -> at most 4 per thread)
        for (int idx = tid, q = 0; idx < VH * Tg; idx += 256, ++q) {
            const int c = idx / Tg, t = idx - c * Tg;
            const float s = dot_row<VF>(P + VpOff::decw + c * VF, R2 + t, Tg, 0.f);
            dv[q] = s;
            ls += s, lq = fmaf(s, s, lq);
        }
        {
            const float n = (float)(VH * Tg);
            const float mean = block_sum256(ls, red) / n;
            const float var = fmaxf(block_sum256(lq, red) / n - mean * mean, 0.f);
            const float rstd = 1.0f / sqrtf(var + kEps);
            for (int idx = tid, q = 0; idx < VH * Tg; idx += 256, ++q) {
                const int c = idx / Tg;
                G[idx] += fmaf((dv[q] - mean) * rstd, P[VpOff::decg + c], P[VpOff::decb + c]);
            }
        }
        __syncthreads();
    }
    // ---- S5: fusion + concat chain (tdanet.py:124-129).  GE / GG temporaries after G in SM.
    float* GE = SM + 1024;
    float* GG = GE + VH * T[1];
    const float* ims = P + VpOff::ims;
    ims_unit(ims + 3 * VpOff::ims_unit, dsp[3], T[3], G, Tg, nullptr, W1, GE, GG);        // fused3 -> W1
    ims_unit(ims + 2 * VpOff::ims_unit, dsp[2], T[2], G, Tg, nullptr, W2, GE, GG);        // fused2 -> W2
    ims_unit(ims + 6 * VpOff::ims_unit, W2, T[2], W1, T[3], dsp[2], W3, GE, GG);          // exp2 = concat2(fused2, fused3) + ds2 -> W3
    ims_unit(ims + 1 * VpOff::ims_unit, dsp[1], T[1], G, Tg, nullptr, W1, GE, GG);        // fused1 -> W1
    ims_unit(ims + 5 * VpOff::ims_unit, W1, T[1], W3, T[2], dsp[1], W2, GE, GG);          // exp1 = concat1(fused1, exp2) + ds1 -> W2
    ims_unit(ims + 0 * VpOff::ims_unit, dsp[0], T[0], G, Tg, nullptr, W1, GE, GG);        // fused0 -> W1
    ims_unit(ims + 4 * VpOff::ims_unit, W1, T[0], W2, T[1], dsp[0], W3, GE, GG);          // exp0 = concat0(fused0, exp1) + ds0 -> W3
    // ---- S6: residual conv 64 -> 512 + bias + gateway residual.  lane = t, wave = 128 output channels, exp0 column in registers.
    float* ob = out + (size_t)b * VIN * Tv;
    for (int t0 = 0; t0 < Tv; t0 += 64) {
        const int t = t0 + lane;
        const bool ok = t < Tv;
        const int tc = ok ? t : Tv - 1;
        float e[VH];
#pragma unroll
        for (int c = 0; c < VH; ++c) e[c] = W3[c * Tv + tc];
#pragma unroll 1
        for (int j0 = 0; j0 < 128; j0 += 4) {
            float xv[4], sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = xb[(size_t)(128 * w + j0 + u) * Tv + tc];  // unconditional, 4 rows in flight
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int co = 128 * w + j0 + u;
                const float* wr = P + VpOff::rw + (size_t)co * VH;  // wave-uniform: scalar loads
                float acc1 = P[VpOff::rb + co];
#pragma unroll
                for (int c = 0; c < VH; ++c) acc1 = fmaf(wr[c], e[c], acc1);
                sv[u] = acc1 + prelu(fmaf(xv[u], P[VpOff::gw + co], P[VpOff::gb + co]), gslope);
            }
            if (ok) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ob[(size_t)(128 * w + j0 + u) * Tv + t] = sv[u];
            }
        }
    }
}

}  // namespace rtfs

using namespace rtfs;

extern "C" {

int rtfs_vp_param_count(void) { return VpOff::total; }

// x, out: [B][512][Tv] (the reference's NCT layout of the lip embedding); params: rtfs_vp_param_count() floats; pe: [>= Tg][64] rows
// of the positional-encoding buffer.  1 <= Tv <= 100 (25 fps x 4 s).
int rtfs_vp_block_fwd(const float* x, const float* params, const float* pe, float* out, int B, int Tv, void* stream) {
    if (B <= 0 || Tv < 1 || Tv > VMAXT) return RTFS_EINVAL;
    int T = Tv, sumT = Tv;
    for (int i = 1; i < 4; ++i) T = (T - 1) / 2 + 1, sumT += T;
    if (T > 16) return RTFS_EINVAL;
    const int Tw = Tv > 3 * T ? Tv : 3 * T;  // (the kernel's work-buffer spacing)
    const size_t bytes = ((size_t)VH * (sumT + 3 * Tw) + VSM) * sizeof(float);
    static bool attr_set[16] = {};  // the attribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return RTFS_ELAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(vp_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess)
            return RTFS_ELAUNCH;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(vp_block_kernel, dim3(B), dim3(256), bytes, (hipStream_t)stream, x, params, pe, out, Tv);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
