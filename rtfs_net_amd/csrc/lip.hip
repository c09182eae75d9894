// Frozen lip encoder: FRCNNVideoModel with the ResNet-18 trunk, eval mode, channels-last fp32.
//   reference: src/models/videomodels/frcnn_videomodel.py:16-72 (frontend3D + trunk), resnet.py:27-66 (BasicBlock), :68-130 (ResNet)
// Every convolution is an implicit GEMM on the fp32 MFMA pipe with the BatchNorm folded into the weights (scale) and the bias
// (shift) on the host; PReLU (per channel) and the residual add run in the epilogue.  Frames are independent: M = frames x pixels.
//   rtfs_lip_stem_fwd   Conv3d(1->64, 5x7x7, stride (1,2,2), pad (2,3,3)) + BN3d + PReLU(64)        frcnn_videomodel.py:43-55
//   rtfs_lip_maxpool_fwd MaxPool3d((1,3,3), (1,2,2), (0,1,1))                                        frcnn_videomodel.py:55
//   rtfs_conv_nhwc_fwd  Conv2d 3x3 pad 1 / 1x1 pad 0, stride 1|2 (+BN) (+residual) (+PReLU)          resnet.py:5-13, 49-66
//   rtfs_lip_avgpool_fwd AdaptiveAvgPool2d(1) + view(B, T, C).transpose(1, 2)                        resnet.py:124-126, frcnn:66
#include "common.h"

namespace rtfs {

constexpr int kLipLds = 68;  // LDS row stride in floats: an odd number of 16-byte slots

// epilogue shared by the two GEMM kernels: lane owns 4 consecutive output channels of one pixel per register group
struct LipEpi {
    const float* bias;   // [Cout] folded BatchNorm shift (or nullptr)
    const float* slope;  // [Cout] PReLU slopes (or nullptr: no activation)
    const float* res;    // [M][Cout] residual (or nullptr)
    float* out;          // [M][Cout]
    int Cout;
};

template <int WN>
__device__ __forceinline__ void lip_store(const floatx16 (&acc)[1][WN], const LipEpi& e, int co_w, long long px_w, long long M) {
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const long long px = px_w + 32 * n + i;
        if (px >= M) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = co_w + 8 * g + 4 * kh;
            float4 v = acc_group(acc[0][n], g);
            if (e.bias) v = v + ld4(e.bias + co);
            if (e.res) v = v + ld4(e.res + px * e.Cout + co);
            if (e.slope) {
                const float4 a = ld4(e.slope + co);
                v = f4(prelu(v.x, a.x), prelu(v.y, a.y), prelu(v.z, a.z), prelu(v.w, a.w));
            }
            st4(e.out + px * e.Cout + co, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic KS x KS convolution over [N][H][W][Cin] -> [N][Ho][Wo][Cout] as an implicit GEMM: workgroup tile = 64 output channels x
// 128 output pixels, K runs over (tap, 64-channel slice): the pixel operand of one K step is 128 contiguous 256-byte channel runs
// (zero outside the image), the weight operand 64 rows of Wk[Cout][KS*KS*Cin].  Next step's global loads are in flight under the
// current step's MFMAs.  Waves 2 x 2: 32 channels x 64 pixels each.
// ------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256, 2) void conv_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ Wk, LipEpi e, int H, int W,
                                                            int Cin, int Ho, int Wo, int stride, long long M, long long in_bytes) {
    __shared__ __attribute__((aligned(16))) float As[64 * kLipLds];
    __shared__ __attribute__((aligned(16))) float Bs[128 * kLipLds];
    constexpr int PAD = KS / 2;
    const int co0 = blockIdx.y * 64;
    const long long px0 = (long long)blockIdx.x * 128;
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
    const int ktot = KS * KS * Cin;
    // The 8 pixel rows this thread stages.  Per row, once: the byte offset of its window's top-left element and a KS*KS-bit mask of
    // the taps that fall inside the image.  Per K step the address is then ONE add (row offset + the step's uniform tap / channel
    // offset) and the zero padding ONE select: out-of-image taps get an offset past the end of the buffer descriptor, which the
    // hardware answers with zeros (fp32 MFMA does not overlap with VALU work - per-step address arithmetic is paid in full).
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)min(in_bytes, (long long)0x7fffffff), 0x00020000);
    int roff[8];
    unsigned rmask[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        long long px = px0 + r0 + 16 * j;
        const bool ok = px < M;
        if (!ok) px = 0;
        const long long n = px / (Ho * Wo);
        const int rem = (int)(px - n * (Ho * Wo)), oy = rem / Wo, ox = rem - oy * Wo;
        const int iy = oy * stride - PAD, ix = ox * stride - PAD;
        roff[j] = (int)(((n * H + iy) * W + ix) * Cin + c4) * 4;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) {
            const int y = iy + t / KS, x = ix + t % KS;
            if (ok && y >= 0 && y < H && x >= 0 && x < W) mk |= 1u << t;
        }
        rmask[j] = mk;
    }
    float4 va[4], vb[8];
    auto fetch = [&](int tap, int ci0) {
        const int dy = tap / KS, dx = tap - dy * KS;
        const int soff = ((dy * W + dx) * Cin + ci0) * 4;  // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = ld4(Wk + (size_t)(co0 + r0 + 16 * j) * ktot + tap * Cin + ci0 + c4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int off = (rmask[j] >> tap) & 1 ? roff[j] + soff : 0x7ffffff0;
            const floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0));
            vb[j] = f4(v[0], v[1], v[2], v[3]);
        }
    };
    const int wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1;
    floatx16 acc[1][2];
    acc_zero(acc);
    const int nci = Cin / 64, nsteps = KS * KS * nci;
    fetch(0, 0);
    for (int st = 0; st < nsteps; ++st) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) st4(As + (r0 + 16 * j) * kLipLds + c4, va[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) st4(Bs + (r0 + 16 * j) * kLipLds + c4, vb[j]);
        __syncthreads();
        if (st + 1 < nsteps) {
            const int tap = (st + 1) / nci;
            fetch(tap, (st + 1 - tap * nci) * 64);
        }
        mma_block<1, 2>(acc, As + wm * 32 * kLipLds, kLipLds, Bs + wn * 64 * kLipLds, kLipLds, 64);
    }
    lip_store<2>(acc, e, co0 + wm * 32, px0 + wn * 64, M);
}

// ------------------------------------------------------------------------------------------------
// Stem: Conv3d(1 -> 64, 5x7x7, stride (1,2,2)) on zero-padded frames P [B][T+4][H+6][W+6]; K = 245 taps padded to 256 (zero
// weights), Ws [64][256] k-contiguous.  The pixel operand is gathered tap by tap: a thread owns one output pixel and walks the K
// slice through a table of tap offsets (wave-uniform index).  Output [B*T][Hc][Wc][64] (before the max-pool).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void lip_stem_kernel(const float* __restrict__ P, const float* __restrict__ Ws, const int* __restrict__ tapoff,
                                                           LipEpi e, int T, int Hp, int Wp, int Hc, int Wc, long long M) {
    __shared__ __attribute__((aligned(16))) float As[64 * kLipLds];
    __shared__ __attribute__((aligned(16))) float Bs[128 * kLipLds];
    __shared__ int toff[256];
    toff[threadIdx.x] = tapoff[threadIdx.x];
    const long long px0 = (long long)blockIdx.x * 128;
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
    // gather role: pixel gp = tid & 127, K half gk = tid >> 7 (32 taps of the 64-tap slice)
    const int gp = threadIdx.x & 127, gk = threadIdx.x >> 7;
    long long px = px0 + gp;
    if (px >= M) px = M - 1;
    const long long n = px / (Hc * Wc);
    const int rem = (int)(px - n * (Hc * Wc)), oy = rem / Wc, ox = rem - oy * Wc;
    const long long b = n / T;
    const int t = (int)(n - b * T);
    const float* base = P + ((b * (T + 4) + t) * Hp + 2 * oy) * (long long)Wp + 2 * ox;
    const int wave = threadIdx.x >> 6, wm = wave & 1, wn = wave >> 1;
    floatx16 acc[1][2];
    acc_zero(acc);
    __syncthreads();
    for (int k0 = 0; k0 < 256; k0 += 64) {
        float4 va[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = ld4(Ws + (r0 + 16 * j) * 256 + k0 + c4);
        float g[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) g[j] = base[toff[k0 + gk * 32 + j]];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) st4(As + (r0 + 16 * j) * kLipLds + c4, va[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) st4(Bs + gp * kLipLds + gk * 32 + 4 * j, f4(g[4 * j], g[4 * j + 1], g[4 * j + 2], g[4 * j + 3]));
        __syncthreads();
        mma_block<1, 2>(acc, As + wm * 32 * kLipLds, kLipLds, Bs + wn * 64 * kLipLds, kLipLds, 64);
    }
    lip_store<2>(acc, e, wm * 32, px0 + wn * 64, M);
}

// 3x3 stride-2 pad-1 max-pool over [N][H][W][64] -> [N][Ho][Wo][64]; 16 lanes x float4 per pixel
__global__ __launch_bounds__(256) void lip_maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int Ho, int Wo,
                                                          long long M) {
    const long long px = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (px >= M) return;
    const int c4 = (threadIdx.x & 15) * 4;
    const long long n = px / (Ho * Wo);
    const int rem = (int)(px - n * (Ho * Wo)), oy = rem / Wo, ox = rem - oy * Wo;
    float4 m = f4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int iy = 2 * oy - 1 + dy, ix = 2 * ox - 1 + dx;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float4 v = ld4(in + ((n * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * 64 + c4);
            if (ok) m = f4(fmaxf(m.x, v.x), fmaxf(m.y, v.y), fmaxf(m.z, v.z), fmaxf(m.w, v.w));
        }
    st4(out + px * 64 + c4, m);
}

// mean over the HW pixels of frame n = b*T + t, written as out[b][c][t]
__global__ __launch_bounds__(256) void lip_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int HW, int C, int T) {
    const long long n = blockIdx.x;
    const long long b = n / T;
    const int t = (int)(n - b * T);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += in[(n * HW + p) * C + c];
        out[(b * C + c) * T + t] = s / (float)HW;
    }
}


// Mouth-ROI preprocessing (src/datas/transform.py:151-167): uint8 grey frames [B][T][H][W] -> Normalize(0, 255) ->
// crop ch x cw (centre, or per-clip offset) -> optional horizontal flip -> Normalize(mean, std), written straight into the zero-padded
// clip P [B][T+4][ch+6][cw+6] the stem reads (borders included: no memset).  The value map is a 256-entry table computed on the
// host in float64 exactly as numpy evaluates the reference's pipeline, so the result is bit-identical to it.
__global__ __launch_bounds__(256) void lip_roi_kernel(const unsigned char* __restrict__ roi, const int* __restrict__ crop, const float* __restrict__ lut,
                                                      float* __restrict__ P, int T, int H, int W, int ch, int cw, int dy0, int dx0) {
    __shared__ float tab[256];
    tab[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.z, tp = blockIdx.y;  // padded frame index 0 .. T+3
    const int Hp = ch + 6, Wp = cw + 6;
    const int dy = crop ? crop[3 * b] : dy0, dx = crop ? crop[3 * b + 1] : dx0;
    const bool flip = crop ? crop[3 * b + 2] != 0 : false;
    const int t = tp - 2;
    float* out = P + ((size_t)b * (T + 4) + tp) * Hp * Wp;
    const unsigned char* in = roi + ((size_t)b * T + max(min(t, T - 1), 0)) * H * W;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < Hp * Wp; idx += gridDim.x * 256) {
        const int y = idx / Wp, x = idx - y * Wp;
        const int cy = y - 3, cx = x - 3;
        const bool inside = t >= 0 && t < T && cy >= 0 && cy < ch && cx >= 0 && cx < cw;
        const int sx = flip ? cw - 1 - cx : cx;
        const unsigned char u = in[min(max(dy + cy, 0), H - 1) * W + min(max(dx + sx, 0), W - 1)];
        out[idx] = inside ? tab[u] : 0.f;
    }
}

}  // namespace rtfs

using namespace rtfs;

extern "C" {

// P: zero-padded frames [B][T+4][H+6][W+6]; Ws [64][256] (k = (dt*7+dy)*7+dx, BN scale folded, k >= 245 zero); tapoff [256] ints =
// (dt*(H+6) + dy)*(W+6) + dx (0 for k >= 245); bias, slope [64]; out [B*T][Hc][Wc][64], Hc = (H-1)/2 + 1.
int rtfs_lip_stem_fwd(const float* P, const float* Ws, const int* tapoff, const float* bias, const float* slope, float* out, int B, int T, int H,
                      int W, void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0) return RTFS_EINVAL;
    const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;
    const long long M = (long long)B * T * Hc * Wc;
    LipEpi e{bias, slope, nullptr, out, 64};
    hipLaunchKernelGGL(lip_stem_kernel, dim3((unsigned)((M + 127) / 128)), dim3(256), 0, (hipStream_t)stream, P, Ws, tapoff, e, T, H + 6, W + 6, Hc,
                       Wc, M);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// in [N][H][W][64] -> out [N][Ho][Wo][64], Ho = (H-1)/2 + 1
int rtfs_lip_maxpool_fwd(const float* in, float* out, int N, int H, int W, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return RTFS_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long M = (long long)N * Ho * Wo;
    hipLaunchKernelGGL(lip_maxpool_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, (hipStream_t)stream, in, out, H, W, Ho, Wo, M);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// in [N][H][W][Cin]; Wk [Cout][ks*ks*Cin] (k = (dy*ks + dx)*Cin + ci); pad = ks/2; out [N][Ho][Wo][Cout], Ho = (H + 2 pad - ks)/stride + 1.
// bias / slope [Cout] or NULL; res [N][Ho][Wo][Cout] or NULL (added before the activation).  Cin, Cout multiples of 64.
int rtfs_conv_nhwc_fwd(const float* in, const float* Wk, const float* bias, const float* slope, const float* res, float* out, int N, int H, int W,
                       int Cin, int Cout, int ks, int stride, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin % 64 || Cout % 64 || (ks != 1 && ks != 3) || (stride != 1 && stride != 2)) return RTFS_EINVAL;
    const long long in_bytes = (long long)N * H * W * Cin * 4;
    if (in_bytes >= 0x7ffffff0LL) return RTFS_EINVAL;  // 32-bit buffer offsets (2 GB of input activations per call)
    const int pad = ks / 2, Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    const long long M = (long long)N * Ho * Wo;
    LipEpi e{bias, slope, res, out, Cout};
    dim3 grid((unsigned)((M + 127) / 128), Cout / 64);
    if (ks == 3)
        hipLaunchKernelGGL((conv_nhwc_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, in, Wk, e, H, W, Cin, Ho, Wo, stride, M, in_bytes);
    else
        hipLaunchKernelGGL((conv_nhwc_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, in, Wk, e, H, W, Cin, Ho, Wo, stride, M, in_bytes);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// in [B*T][HW][C] -> out [B][C][T]
int rtfs_lip_avgpool_fwd(const float* in, float* out, int B, int T, int HW, int C, void* stream) {
    if (B <= 0 || T <= 0 || HW <= 0 || C <= 0) return RTFS_EINVAL;
    hipLaunchKernelGGL(lip_avgpool_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, in, out, HW, C, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// roi [B][T][H][W] uint8; crop [B][3] ints (dy, dx, flip) or NULL = centre crop (transform.py:96-101: delta = int(round(H - ch) / 2.0));
// lut [256] floats; P [B][T+4][ch+6][cw+6] (whole buffer written).  0 <= dy <= H - ch, 0 <= dx <= W - cw is the caller's contract.
int rtfs_lip_roi_fwd(const unsigned char* roi, const int* crop, const float* lut, float* P, int B, int T, int H, int W, int ch, int cw, void* stream) {
    if (B <= 0 || T <= 0 || ch <= 0 || cw <= 0 || H < ch || W < cw) return RTFS_EINVAL;
    const int plane = (ch + 6) * (cw + 6);
    dim3 grid(min((plane + 255) / 256, 64), T + 4, B);
    hipLaunchKernelGGL(lip_roi_kernel, grid, dim3(256), 0, (hipStream_t)stream, roi, crop, lut, P, T, H, W, ch, cw, (H - ch) / 2, (W - cw) / 2);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
