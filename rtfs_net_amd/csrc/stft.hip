// STFT encoder front end and iSTFT decoder back end.
//
//   rtfs_stft_fwd       torch.stft(n_fft=256, hop=128, periodic hann, center=True, reflect, onesided)
//                       -> stack(re, im) (/root/reference/src/models/TDAVNet/encoder.py:161-173), channels-last [B][T][129][2]
//   rtfs_enc_conv_fwd   Conv2d(2->256, 3x3, 'same', no bias) (encoder.py:146-157,174) + gLN partial sums for the bottleneck
//   rtfs_istft_fwd      ConvTranspose2d(256->2,3x3,pad 1) tap gather + torch.istft(length=L)  (decoder.py:113-130)
//
// One wave per frame: a 256-point complex FFT as 4 radix-4 Stockham passes in LDS (64 butterflies per pass
// = one per lane), twiddles from sincospif.  The decoder's transposed convolution is split into a per-pixel
// GEMM 256 -> 18 taps (rtfs_gemm_rows_fwd) and a 9-neighbour gather that is fused here into the spectrum load.
#include "common.h"

namespace rtfs {

// One radix-4 Stockham FFT of 256 complex points held in LDS (two ping-pong buffers of 256 float2), 64 lanes.
// Returns the buffer that holds the result.  INV: conjugate kernel (no 1/N scaling).  Every wave of the
// workgroup must call it (it synchronises with __syncthreads between passes).
template <bool INV>
__device__ __forceinline__ float2* fft256(float2* a, float2* b, int j) {
    constexpr float sgn = INV ? 1.f : -1.f;
#pragma unroll
    for (int Ns = 1; Ns < 256; Ns *= 4) {
        const int k = j & (Ns - 1);
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = a[j + r * 64];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < 4; ++r) {
                float sn, cs;
                sincospif(sgn * 2.0f * (float)(r * k) / (float)(Ns * 4), &sn, &cs);
                v[r] = make_float2(v[r].x * cs - v[r].y * sn, v[r].x * sn + v[r].y * cs);
            }
        }
        const float2 a0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y);
        const float2 a1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
        const float2 a2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y);
        const float2 d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
        // (v1 - v3) * (sgn * i):  i*(x+iy) = -y + ix
        const float2 a3 = make_float2(-sgn * d.y, sgn * d.x);
        const int j0 = ((j - k) << 2) + k;
        b[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
        b[j0 + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
        b[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
        b[j0 + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
        __syncthreads();
        float2* t = a;
        a = b;
        b = t;
    }
    return a;
}

__device__ __forceinline__ float hann256(int i) { return 0.5f - 0.5f * cospif((float)i * (1.0f / 128.0f)); }

// grid: (ceil(T/4), B); 4 waves per workgroup, one frame each.
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ wav, float* __restrict__ spec, int L, int T) {
    __shared__ float2 buf[4][2][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = min(blockIdx.x * 4 + w, T - 1), b = blockIdx.y;
    const bool valid = blockIdx.x * 4 + w < T;
    const float* x = wav + (size_t)b * L;
    float2* A = buf[w][0];
    float2* Bf = buf[w][1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = lane + r * 64;
        int n = t * kHop - kWin / 2 + i;  // center=True: reflect padding by n_fft/2
        if (n < 0) n = -n;
        if (n >= L) n = 2 * (L - 1) - n;
        A[i] = make_float2(x[n] * hann256(i), 0.f);
    }
    __syncthreads();
    float2* R = fft256<false>(A, Bf, lane);
    float2* out = reinterpret_cast<float2*>(spec) + ((size_t)b * T + t) * kF;
    if (valid)
        for (int k = lane; k < kF; k += 64) out[k] = R[k];
}

// a_emb[b][t][f][c] = sum_{ci,dt,df} W[c][ci][dt][df] * spec[b][t+dt-1][f+df-1][ci]; Wp: [18][256], tap = ci*9+dt*3+df.
// grid: (ceil(T*F/64), B); each wave walks 16 pixels, each lane owns 4 output channels.
constexpr int kEncTiles = 4;
__global__ __launch_bounds__(256) void enc_conv_kernel(const float* __restrict__ spec, const float* __restrict__ Wp, float* __restrict__ a_emb,
                                                       double* __restrict__ stats, int T) {
    // The 3x3 neighbourhood of pixel p = t*F + f is p + {-F,0,F} + {-1,0,1} in the flattened [T][F] spectrogram, so the 64 pixels of a
    // workgroup need ONE contiguous range of 64 + 2(F+1) complex values: staged in LDS with a single coalesced load per thread, the
    // per-pixel taps become wave-uniform LDS reads and nothing in the pixel loop waits on HBM (the kernel is a 1 GB write stream).
    constexpr int HALO = kF + 1, NS = 64 + 2 * HALO;
    __shared__ float2 sps[NS];
    __shared__ float red[8];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int TF = T * kF;
    float4 wr[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) wr[i] = ld4(Wp + i * 256 + lane * 4);
    const float2* sp = reinterpret_cast<const float2*>(spec) + (size_t)b * TF;
    float s = 0.f, q = 0.f;
    // kEncTiles 64-pixel tiles per workgroup: the 18 weight vectors, the launch and the two statistics atomics are paid once per 256 pixels
#pragma unroll 1
    for (int tile = 0; tile < kEncTiles; ++tile) {
    const int p0 = (blockIdx.x * kEncTiles + tile) * 64;
    if (p0 >= TF) break;
    if (tile) __syncthreads();  // the previous tile's taps have been read
    for (int i = threadIdx.x; i < NS; i += 256) {
        const int p = p0 - HALO + i;
        sps[i] = (p >= 0 && p < TF) ? sp[p] : make_float2(0.f, 0.f);
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int p = p0 + w * 16 + i;
        if (p < TF) {  // wave-uniform
            const int t = p / kF, f = p - t * kF;
            float4 acc = f4(0, 0, 0, 0);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    const int tt = t + dt - 1, ff = f + df - 1;
                    const float m = (tt >= 0 && tt < T && ff >= 0 && ff < kF) ? 1.f : 0.f;  // 'same' zero padding (row wrap-around masked)
                    const float2 v = sps[w * 16 + i + HALO + (dt - 1) * kF + (df - 1)];
                    const float re = v.x * m, im = v.y * m;
                    acc = fma4(wr[dt * 3 + df], f4(re, re, re, re), acc);
                    acc = fma4(wr[9 + dt * 3 + df], f4(im, im, im, im), acc);
                }
            st4(a_emb + ((size_t)b * TF + p) * kC + lane * 4, acc);
            s += acc.x + acc.y + acc.z + acc.w;
            q += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
        }
    }
    }
    block_stats_commit(s, q, red, stats, b);
}

// taps: [B][T][F][32]; column o*9 + kt*3 + kf holds sum_c masked[c] * Wdec[c][o][kt][kf] of that pixel.
// spectrum[b][t][f][o] = sum_{kt,kf} taps[b][t-kt+1][f-kf+1][o*9+kt*3+kf]   (ConvTranspose2d stride 1, padding 1)
// frames[b][t][i] = hann[i] * irfft(spectrum[b][t])[i].    grid: (ceil(T/4), B)
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ taps, float* __restrict__ frames, int T) {
    __shared__ float2 buf[4][2][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = min(blockIdx.x * 4 + w, T - 1), b = blockIdx.y;
    const bool valid = blockIdx.x * 4 + w < T;
    float2* A = buf[w][0];
    float2* Bf = buf[w][1];
    const float* tp = taps + (size_t)b * T * kF * 32;
    for (int k = lane; k < kF; k += 64) {
        // the 18 taps of a bin: all loads first, addresses clamped and the out-of-range taps masked afterwards (a load under a branch is
        // waited for before the next one issues: 18 serial L2 latencies per bin in the first version)
        float vr[9], vi[9];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int tt = t - kt + 1, ff = k - kf + 1;
                const float* p = tp + ((size_t)min(max(tt, 0), T - 1) * kF + min(max(ff, 0), kF - 1)) * 32 + kt * 3 + kf;
                vr[kt * 3 + kf] = p[0];
                vi[kt * 3 + kf] = p[9];
            }
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kf = 0; kf < 3; ++kf) {
                const int tt = t - kt + 1, ff = k - kf + 1;
                const bool ok = tt >= 0 && tt < T && ff >= 0 && ff < kF;
                re += ok ? vr[kt * 3 + kf] : 0.f;  // (same summation order as before: bit-identical)
                im += ok ? vi[kt * 3 + kf] : 0.f;
            }
        if (k == 0 || k == 128) im = 0.f;  // C2R ignores the imaginary part of DC and Nyquist
        A[k] = make_float2(re, im);
        if (k > 0 && k < 128) A[256 - k] = make_float2(re, -im);
    }
    __syncthreads();
    float2* R = fft256<true>(A, Bf, lane);
    float* fr = frames + ((size_t)b * T + t) * kWin;
    if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = lane + r * 64;
            fr[i] = R[i].x * (1.0f / 256.0f) * hann256(i);
        }
    }
}

// Overlap-add, window-envelope normalisation, centre trim: torch.istft(length=L) (decoder.py:122-128).
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, float* __restrict__ out, int L, int T) {
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= L) return;
    const int m = n + kWin / 2;
    const int t1 = m / kHop;
    float acc = 0.f, env = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int t = t1 - d;
        const int i = m - t * kHop;
        if (t >= 0 && t < T && i < kWin) {
            const float wv = hann256(i);
            acc += frames[((size_t)b * T + t) * kWin + i];
            env = fmaf(wv, wv, env);
        }
    }
    // samples past (T-1)*hop are still covered by the second half of the last frame (torch.istft keeps them)
    out[(size_t)b * L + n] = env > 1e-11f ? acc / env : 0.f;
}

}  // namespace rtfs

using namespace rtfs;

extern "C" {

int rtfs_stft_fwd(const float* wav, float* spec, int B, int L, void* stream) {
    if (B <= 0 || L < kWin / 2 + 1) return RTFS_EINVAL;
    const int T = 1 + L / kHop;
    hipLaunchKernelGGL(stft_kernel, dim3((T + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, wav, spec, L, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_enc_conv_fwd(const float* spec, const float* Wp, float* a_emb, double* stats, int B, int T, void* stream) {
    if (B <= 0 || T <= 0) return RTFS_EINVAL;
    hipLaunchKernelGGL(enc_conv_kernel, dim3((T * kF + 64 * kEncTiles - 1) / (64 * kEncTiles), B), dim3(256), 0, (hipStream_t)stream, spec, Wp, a_emb, stats, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// frames: workspace [B][T][256].  out: [B][L].
int rtfs_istft_fwd(const float* taps, float* frames, float* out, int B, int L, void* stream) {
    if (B <= 0 || L < kWin / 2 + 1) return RTFS_EINVAL;
    const int T = 1 + L / kHop;
    hipLaunchKernelGGL(istft_frames_kernel, dim3((T + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, taps, frames, T);
    RTFS_LAUNCH_CHECK();
    hipLaunchKernelGGL(istft_ola_kernel, dim3((L + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, frames, out, L, T);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
