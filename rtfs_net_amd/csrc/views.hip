// Module-boundary views: what the stage modules of AVNet (encoder, audio_bottleneck, refinement_module, mask_generator, decoder) and the
// forward hooks of their sub-modules need when the reference's module API is used ONE MODULE AT A TIME (`model.encoder(x)`,
// `register_forward_hook`, thop-style per-module profiling: /root/reference/src/models/TDAVNet/base_av_model.py:61-118 calls the stages one
// by one).  The fused forward never materialises a module's output where the next kernel can form it on load (the gateway's output, a
// ConvNormAct's normalised + activated output, NCHW tensors); these kernels materialise exactly those, from the tensors the path does hold.
//
//   rtfs_gln_stats       (sum, sum of squares) per utterance of a channels-last tensor -> a gLN statistics slot (a stage module called on its
//                        own has no producer epilogue that accumulated them)
//   rtfs_norm_act_fwd    y = act(gLN(x)): the output of a ConvNormAct whose conv result x the path keeps un-normalised (conv_layers.py:121-127)
//   rtfs_gateway_fwd     y = prelu(x * w + b): the gateway (depth-wise 1x1 + PReLU, tdanet.py:34-41,108), fused on load everywhere else
//   rtfs_cl_to_nchw / rtfs_nchw_to_cl   layout change at the module boundary ([B][P][C] <-> [B][C][P], P = T*F pixels), through an LDS tile
#include "common.h"

namespace rtfs {

__global__ __launch_bounds__(256) void gln_stats_kernel(const float* __restrict__ x, double* __restrict__ slot, long long n4, int per_wg) {
    __shared__ float red[8];
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * n4 * 4;
    const long long i0 = (long long)blockIdx.x * per_wg, i1 = min(n4, i0 + per_wg);
    float s = 0.f, q = 0.f;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float4 v = ld4(xb + i * 4);
        s += (v.x + v.y) + (v.z + v.w);
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    block_stats_commit(s, q, red, slot, b);
}

// ACT: 0 none, 1 PReLU(slope), 2 ReLU, 3 sigmoid
template <int ACT>
__global__ __launch_bounds__(256) void norm_act_kernel(const float* __restrict__ x, const double* __restrict__ slot, double inv_n,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                       float* __restrict__ y, long long n4, int C) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float mean, rstd;
    stats_finalize(slot, b, inv_n, mean, rstd);
    const int c4 = (int)((i * 4) % C);
    const float4 g = ld4(gamma + c4), be = ld4(beta + c4);
    const size_t o = ((size_t)b * n4 + i) * 4;
    const float4 v = ld4(x + o);
    float4 r = f4((v.x - mean) * rstd * g.x + be.x, (v.y - mean) * rstd * g.y + be.y, (v.z - mean) * rstd * g.z + be.z, (v.w - mean) * rstd * g.w + be.w);
    if (ACT == 1) r = prelu4(r, slope);
    if (ACT == 2) r = f4(fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f));
    if (ACT == 3) r = f4(1.f / (1.f + __expf(-r.x)), 1.f / (1.f + __expf(-r.y)), 1.f / (1.f + __expf(-r.z)), 1.f / (1.f + __expf(-r.w)));
    st4(y + o, r);
}

__global__ __launch_bounds__(256) void gateway_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float slope, float* __restrict__ y, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c4 = (int)((i * 4) % kC);
    const float4 v = ld4(x + i * 4), g = ld4(w + c4), be = ld4(bias + c4);
    st4(y + i * 4, prelu4(f4(v.x * g.x + be.x, v.y * g.y + be.y, v.z * g.z + be.z, v.w * g.w + be.w), slope));
}

// [B][P][C] -> [B][C][P] (TO_NCHW) or back: 64 pixels x 64 channels per workgroup through a padded LDS tile, both sides coalesced
template <bool TO_NCHW>
__global__ __launch_bounds__(256) void layout_kernel(const float* __restrict__ src, float* __restrict__ dst, int P, int C) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const float* s = src + (size_t)b * P * C;
    float* d = dst + (size_t)b * P * C;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = ly + 4 * k;  // row of the SOURCE's fast dimension block
        if (TO_NCHW) {  // source rows = pixels, fast = channels
            if (p0 + r < P && c0 + lx < C) tile[r][lx] = s[(size_t)(p0 + r) * C + c0 + lx];
        } else {        // source rows = channels, fast = pixels
            if (c0 + r < C && p0 + lx < P) tile[r][lx] = s[(size_t)(c0 + r) * P + p0 + lx];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = ly + 4 * k;
        if (TO_NCHW) {
            if (c0 + r < C && p0 + lx < P) d[(size_t)(c0 + r) * P + p0 + lx] = tile[lx][r];
        } else {
            if (p0 + r < P && c0 + lx < C) d[(size_t)(p0 + r) * C + c0 + lx] = tile[lx][r];
        }
    }
}

}  // namespace rtfs

using namespace rtfs;

extern "C" {

int rtfs_gln_stats(const float* x, double* stats, int B, long long per_utt, void* stream) {
    if (B <= 0 || per_utt <= 0 || (per_utt & 3)) return RTFS_EINVAL;
    const long long n4 = per_utt / 4;
    const int per_wg = 256 * 16;
    hipLaunchKernelGGL(gln_stats_kernel, dim3((unsigned)((n4 + per_wg - 1) / per_wg), B), dim3(256), 0, (hipStream_t)stream, x, stats, n4, per_wg);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_norm_act_fwd(const float* x, const double* stats, const float* gamma, const float* beta, int act, float slope, float* y, int B, long long rows,
                      int C, void* stream) {
    if (B <= 0 || rows <= 0 || C <= 0 || (C & 3) || act < 0 || act > 3) return RTFS_EINVAL;
    const long long n4 = rows * C / 4;
    const double inv_n = 1.0 / ((double)rows * C);
    const dim3 grid((unsigned)((n4 + 255) / 256), B);
#define NA(A) hipLaunchKernelGGL(norm_act_kernel<A>, grid, dim3(256), 0, (hipStream_t)stream, x, stats, inv_n, gamma, beta, slope, y, n4, C)
    if (act == 0) NA(0); else if (act == 1) NA(1); else if (act == 2) NA(2); else NA(3);
#undef NA
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_gateway_fwd(const float* x, const float* w, const float* bias, float slope, float* y, long long rows, void* stream) {
    if (rows <= 0) return RTFS_EINVAL;
    const long long n4 = rows * kC / 4;
    hipLaunchKernelGGL(gateway_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias, slope, y, n4);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_cl_to_nchw(const float* src, float* dst, int B, int P, int C, void* stream) {
    if (B <= 0 || P <= 0 || C <= 0) return RTFS_EINVAL;
    hipLaunchKernelGGL(layout_kernel<true>, dim3((P + 63) / 64, (C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, src, dst, P, C);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_nchw_to_cl(const float* src, float* dst, int B, int P, int C, void* stream) {
    if (B <= 0 || P <= 0 || C <= 0) return RTFS_EINVAL;
    hipLaunchKernelGGL(layout_kernel<false>, dim3((P + 63) / 64, (C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, src, dst, P, C);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
