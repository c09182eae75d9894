// Shared device helpers for the RTFS-Net hot-path kernels (gfx950 / CDNA4, wave64).
//
// Activation layout everywhere below the C-ABI is CHANNELS-LAST fp32:
//   full resolution   X[b][t][f][c]   (c = 256 "C" or 64 "H")
//   compressed        G[b][t2][f2][c] (c = 64)
// so a "pixel" (b,t,f) owns a contiguous, 16-byte aligned channel vector and every 1x1 convolution is
// a row-major GEMM  [pixels x Cin] x [Cin x Cout].  The reference (NCHW, /root/reference/src/models)
// never sees this layout: only waveforms and lip embeddings cross the boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// the C-ABI every .hip file defines a part of: including it here makes a definition that drifts from its declaration fail to compile
#include "../../include/rtfs_hip.h"

namespace rtfs {

constexpr int kC = 256;   // encoder / bottleneck channels   (config yaml: enc_dec_params.out_chan)
constexpr int kH = 64;    // RTFS block hidden channels      (audio_params.hid_chan)
constexpr int kWin = 256; // STFT window                     (enc_dec_params.win)
constexpr int kHop = 128; // STFT hop                        (enc_dec_params.hop_length)
constexpr int kF = 129;   // kWin/2 + 1 frequency bins
constexpr int kF2 = 64;   // compressed frequency bins       (layer_3.n_freqs)
// gLN statistic slots: one 128-byte line per utterance (first two doubles used).  Packed [B][2], eight utterances shared a line
// and every workgroup's two fp64 atomics queued behind each other at ~27 ns per request (enc_conv: 16k workgroups -> 218 us).
constexpr int kStatStride = 16;
constexpr float kEps = 1e-5f;  // src/models/layers/normalizations.py:5

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---- error codes returned across the C-ABI: RTFS_OK / RTFS_EINVAL / RTFS_ELAUNCH (include/rtfs_hip.h) ----

#define RTFS_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return RTFS_ELAUNCH;  \
    } while (0)

// run-time precision -> template: terms 1 (bf16), 3 (split-bf16) or 6 (fp32-equivalent three-way split); anything else is refused
#define RTFS_TERMS_DISPATCH(terms, CALL1, CALL3, CALL6) \
    do {                                                \
        if ((terms) == 1) return CALL1;                 \
        if ((terms) == 3) return CALL3;                 \
        if ((terms) == 6) return CALL6;                 \
        return RTFS_EINVAL;                             \
    } while (0)

// ---- gLN statistics --------------------------------------------------------------------------------
// A global layer norm (GroupNorm(1,C), normalizations.py:8-17) needs mean/var over (C,T,F) of one
// utterance: a grid-wide reduction.  Producers add per-workgroup partial (sum, sum of squares) into a
// double2 slot per utterance with fp64 atomics; consumers turn the slot into (mean, rstd) when they
// load ("normalise on read").  fp64 accumulation makes the result insensitive to arrival order.
struct Stats {
    const double* slot;  // [B][2]
    float inv_n;         // 1 / (C*T*F)
};

__device__ __forceinline__ void stats_finalize(const double* slot, int b, double inv_n, float& mean, float& rstd) {
    double s = slot[kStatStride * b], q = slot[kStatStride * b + 1];
    double m = s * inv_n;
    double v = q * inv_n - m * m;
    if (v < 0) v = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(v + (double)kEps));
}

__device__ __forceinline__ float row16_sum(float v);
// xor butterfly 32, 16, 8, 4, 2, 1 over the wave, every lane gets the total; the four steps inside a 16-lane row run as DPP adds (row16_sum, same
// operand pairs = same bits as four more __shfl_xor steps, a quarter of their cost: ds_bpermute goes through the LDS crossbar)
__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32, 64);
    v += __shfl_xor(v, 16, 64);
    return row16_sum(v);
}
// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), every lane gets the total: the xor butterfly 8, 4, 2, 1 of
//   for (o = 8; o; o >>= 1) v += __shfl_xor(v, o)
// with the same operand pairs (after the step for bit b the partial sums no longer depend on that bit of the lane index, so rotating the
// row by 2^b reaches a lane that holds exactly what lane ^ 2^b holds) - bit-identical, 4 v_add_f32_dpp instead of 4 ds_bpermute round trips.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide (256 threads) reduction of a (sum, sumsq) pair followed by ONE pair of fp64 atomics.
// `red` is 8 floats of LDS.
__device__ __forceinline__ void block_stats_commit(float s, float q, float* red, double* slot, int b) {
    s = wave_sum(s);
    q = wave_sum(q);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        red[w] = s;
        red[4 + w] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
        double Q = (double)red[4] + (double)red[5] + (double)red[6] + (double)red[7];
        atomicAdd(slot + kStatStride * b, S);
        atomicAdd(slot + kStatStride * b + 1, Q);
    }
}

// ---- spread accumulators (spread.hip) ------------------------------------------------------------------------------
constexpr int kSpread = 32;         // scratch copies; a workgroup adds into copy (its index % kSpread)
constexpr int kSpreadCap = 65536;   // floats per copy (round 4: room for several pending regions, see spread.hip)
constexpr int kSpreadSlots = 8;
struct SpreadOut {                  // up to 8 accumulators of one kernel; region j starts at the 32-float-aligned running offset
    float* dst[kSpreadSlots];
    int n[kSpreadSlots];
    __host__ __device__ int off(int j) const {
        int o = 0;
        for (int k = 0; k < j; ++k) o += (n[k] + 31) & ~31;
        return o;
    }
};
// this workgroup's copy of the scratch
__device__ __forceinline__ float* spread_copy(float* scr, unsigned wg) { return scr + (size_t)(wg & (kSpread - 1)) * kSpreadCap; }
float* spread_scratch();  // this device's zeroed scratch [kSpread][kSpreadCap], at the start of the next free region (nullptr on failure)
// dst[j][i] += sum over copies of the region that starts at `scr`; re-zeroes it.  Immediately (one small launch), or - in deferred mode,
// rtfs_spread_defer - recorded and applied by ONE launch for all pending regions at the next flush; `consumed_now`: the caller reads dst right away
int spread_finish(float* scr, const SpreadOut& o, hipStream_t st, bool consumed_now = false);

// dualpath.hip: the fast-FIR weight-stationary kernel in its ConvTranspose-input-gradient mode (bwd_gemm.hip calls it); 1 = size not eligible
int convt_bwd_input_ffa(const float* dG, const float* Wt, float* dH3, int B, int T2, int dim, hipStream_t stream);

// 1 / (1 + 2^(-x log2 e)) on v_exp_f32 + v_rcp_f32 (1 ulp each).  __frcp_rn expands to the full IEEE division sequence
// (div_scale / rcp / 4 fma / div_fmas / div_fixup: 10 VALU instructions per gate), which dominated the recurrence kernels.
constexpr float kNegLog2e = -1.4426950408889634f;
__device__ __forceinline__ float sigmoid_from_exp2arg(float a) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a)); }  // a = -x log2 e
__device__ __forceinline__ float sigmoidf_fast(float x) { return sigmoid_from_exp2arg(x * kNegLog2e); }
__device__ __forceinline__ float prelu(float x, float a) { return x >= 0.f ? x : a * x; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// wave-uniform base + 32-bit BYTE offset: selects the `global_load v, v_off, s[base]` form (no 64-bit VALU address math; a float
// index would be scaled after the zero-extension and fall back to 64-bit adds)
__device__ __forceinline__ float4 ld4_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st4_off(float* base, unsigned byte_off, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__device__ __forceinline__ float ld1_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st1_off(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Native 4-vector (clang ext_vector_type): `a * b + c` on it lowers to two v_pk_fma_f32 on the aligned register pairs a 16-byte load already
// delivers.  The struct float4 version of the same arithmetic goes through the SLP vectoriser, which pairs lanes of DIFFERENT values and pays
// 2 v_mov per v_pk_fma to build its operands (measured in the depth-wise conv window pass: 70 moves per 35 packed FMAs).
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4v ld4v(const float* p) { return *reinterpret_cast<const float4v*>(p); }
__device__ __forceinline__ float4 to_f4(float4v v) { return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4v to_v4(float4 v) { return float4v{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float4 f4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return f4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float s) { return f4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 sub4(float4 a, float s) { return f4(a.x - s, a.y - s, a.z - s, a.w - s); }
__device__ __forceinline__ float hsum4(float4 a) { return a.x + a.y + a.z + a.w; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return f4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
// (x - mean) * rstd * gamma + beta  ==  x * (rstd*gamma) + (beta - mean*rstd*gamma)
__device__ __forceinline__ float4 norm4(float4 x, float mean, float rstd, float4 g, float4 b) {
    float4 sc = g * rstd;
    float4 sh = f4(b.x - mean * sc.x, b.y - mean * sc.y, b.z - mean * sc.z, b.w - mean * sc.w);
    return fma4(x, sc, sh);
}
__device__ __forceinline__ float4 prelu4(float4 x, float a) { return f4(prelu(x.x, a), prelu(x.y, a), prelu(x.z, a), prelu(x.w, a)); }
// prelu(x) = x + (a - 1) min(x, 0): 2 (packable) instructions per pair instead of compare + select + multiply per element.
// Differs from a*x by one rounding of (a - 1) x + x.   am1 = a - 1.
__device__ __forceinline__ float4 prelu4_minfma(float4 x, float am1) {
    return f4(fmaf(fminf(x.x, 0.f), am1, x.x), fmaf(fminf(x.y, 0.f), am1, x.y), fmaf(fminf(x.z, 0.f), am1, x.z), fmaf(fminf(x.w, 0.f), am1, x.w));
}
__device__ __forceinline__ float4 relu4(float4 x) { return f4(fmaxf(x.x, 0.f), fmaxf(x.y, 0.f), fmaxf(x.z, 0.f), fmaxf(x.w, 0.f)); }
__device__ __forceinline__ float4 sigmoid4(float4 x) {
    return f4(sigmoidf_fast(x.x), sigmoidf_fast(x.y), sigmoidf_fast(x.z), sigmoidf_fast(x.w));
}

// nearest-neighbour source index used by F.interpolate(mode="nearest"): floor(dst * in / out)
// (32-bit: dst * in < 2^31 for every shape on this path; a 64-bit divide would cost ~100 instructions per pixel)
__device__ __forceinline__ int nearest_src(int dst, int in, int out) { return (int)(((unsigned)dst * (unsigned)in) / (unsigned)out); }

// ---- fp32 MFMA tile machinery --------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32]; lane l supplies A[l&31][l>>5] and
// B[l>>5][l&31]; accumulator register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
// Exact fp32 (one rounding per product) at the fp32 vector rate, 64 cycles per instruction.
//
// Both operands are staged in LDS "k-contiguous": As[row][k], Bs[col][k] with a row stride that is an
// odd number of 16-byte slots (36 or 68 floats) so that a wave's ds_read_b128 is bank-conflict free.
// Lane (i = l&31, kh = l>>5) reads the 4 consecutive k values 8q + 4kh .. +3 of its row with ONE
// ds_read_b128 and feeds them to 4 successive MFMAs; the pairing (A k, B k) is what matters, the
// order in which k is summed does not.
// BT = distance in B rows (output columns) between this wave's consecutive N-tiles (32 = adjacent);
// BT == -4 selects the "paired" map for 4 tiles: tiles 0,1 are adjacent, tiles 2,3 sit 128 columns further.
template <int BT>
__device__ __forceinline__ constexpr int btile_row(int n) { return BT == -4 ? ((n >> 1) * 128 + (n & 1) * 32) : n * BT; }

//
// Orientation: the FIRST operand's rows become accumulator rows (spread over the 16 registers of a lane: each
// group of 4 registers = 4 consecutive rows), the SECOND operand's rows become accumulator columns (= lanes).
// Passing the WEIGHTS first and the PIXELS second therefore leaves every lane with 4 consecutive output
// channels of one pixel per register group -> channels-last write-back in 16-byte vectors.  AT is the tile map
// of the first operand, BT of the second.
template <int WM, int WN, int BT = 32, int AT = 32>
__device__ __forceinline__ void mma_block(floatx16 (&acc)[WM][WN], const float* As, int lda, const float* Bs, int ldb, int kdepth) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, kh = lane >> 5;
    const float* ap = As + i * lda + kh * 4;
    const float* bp = Bs + i * ldb + kh * 4;
#pragma unroll 2
    for (int q = 0; q < kdepth; q += 8) {
        float4 a[WM], b[WN];
#pragma unroll
        for (int m = 0; m < WM; ++m) a[m] = ld4(ap + btile_row<AT>(m) * lda + q);
#pragma unroll
        for (int n = 0; n < WN; ++n) b[n] = ld4(bp + btile_row<BT>(n) * ldb + q);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, b[n].x, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, b[n].y, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, b[n].z, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, b[n].w, acc[m][n], 0, 0, 0);
            }
    }
}

// Variant for a B operand stored "n-contiguous" in LDS: Bs[k][col] (row stride ldb floats).
template <int WM, int WN>
__device__ __forceinline__ void mma_block_bn(floatx16 (&acc)[WM][WN], const float* As, int lda, const float* Bs, int ldb, int kdepth) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, kh = lane >> 5;
    const float* ap = As + i * lda + kh * 4;
    const float* bp = Bs + (kh * 4) * ldb + i;
#pragma unroll 2
    for (int q = 0; q < kdepth; q += 8) {
        float4 a[WM];
        float b[WN][4];
#pragma unroll
        for (int m = 0; m < WM; ++m) a[m] = ld4(ap + m * 32 * lda + q);
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) b[n][r] = bp[(q + r) * ldb + n * 32];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, b[n][0], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, b[n][1], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, b[n][2], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, b[n][3], acc[m][n], 0, 0, 0);
            }
    }
}

// ---- bf16 / split-bf16 MFMA (v_mfma_f32_32x32x16_bf16, v_mfma_f32_16x16x32_bf16; 16x the fp32 MFMA rate) --------------------------
// Precision switch of every MFMA kernel, template parameter NT ("terms"):
//   NT = 0  exact fp32 (v_mfma_f32_32x32x2_f32): the default path, untouched by the code below;
//   NT = 1  operands rounded to bfloat16 (RNE), fp32 accumulation:                 a.b ~ a_hi b_hi
//   NT = 3  split-bf16: a = a_hi + a_lo (a_lo = bf16(a - a_hi)), three products:    a.b ~ a_lo b_hi + a_hi b_lo + a_hi b_hi
//           (drops a_lo b_lo ~ 2^-18 |a b|: waveform error ~8e-6 vs fp32 on RTFS-Net-12, tools/bf16_error_model.py).
//   NT = 6  fp32-equivalent: a = a_hi + a_mid + a_lo carries all 24 mantissa bits of the fp32 operand in three bfloat16 values; the six
//           products hh, hm, mh, hl, lh, mm are accumulated in fp32, the dropped ml, lm, ll terms are <= 2^-23 |a b| - the size of ONE fp32
//           rounding.  6 x 32 cycles per K = 16 against 8 x 64 on the fp32 pipe: 2.7x its rate at its accuracy (not bit-identical to it).
//           Operands stay plain fp32 in HBM; the generic (LDS-staged) kernels keep fp32 tiles in LDS as well (three planes do not fit the 16-byte slot
//           below) and split every fragment in registers - ~56 VALU instructions per ds_read_b128, which is why the mode's two hottest kernels have their
//           own forms that split each operand ONCE into three bf16 planes (unfold_ws6_kernel, sru_layer_kernel<., 6, .>: dualpath.hip, round 5), and why
//           some entry points route this mode to their fp32 kernel where that one is weight-stationary (gemm.hip, bwd_gemm.hip).
// PACKED SLOT.  The fp32 kernels stage operands k-contiguously as float4 = 4 consecutive k.  The bf16 paths keep every address,
// stride and swizzle of those layouts and store in the same 16 bytes the 4 hi halves and the 4 lo halves of the same 4 k values:
//      slot = [hi(k0) hi(k1) | hi(k2) hi(k3) | lo(k0) lo(k1) | lo(k2) lo(k3)]        (4 dwords)
// A lane that read slots q and q+1 of its row (k = 8q + 4kh .. +3 and 8(q+1) + 4kh .. +3) owns 8 k values = one A / B fragment of
// the K = 16 instruction; which k a fragment element holds is free as long as both operands agree, and they are built the same way.
// Weights are packed once on the host into the same slots (same byte size and indexing as the fp32 matrix), so weight staging is
// the unchanged plain copy; activations are packed by the staging store (pack4) or in registers (frag_f32).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even): a -> bits 0..15, b -> 16..31
    const float2v f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2v));
}
template <int NT>
__device__ __forceinline__ float4 pack4(float4 v) {
    if constexpr (NT == 0 || NT == 6) return v;  // NT = 6 keeps fp32 tiles and splits in registers (frag_lds)
    const unsigned h0 = pk_bf16(v.x, v.y), h1 = pk_bf16(v.z, v.w);
    unsigned l0 = 0, l1 = 0;
    if constexpr (NT == 3) {
        l0 = pk_bf16(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
        l1 = pk_bf16(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
    }
    return make_float4(__uint_as_float(h0), __uint_as_float(h1), __uint_as_float(l0), __uint_as_float(l1));
}
struct Frag {  // one MFMA operand fragment (8 k values of one row): hi and lo planes (+ mid for the three-way split, NT = 6)
    bf16x8 hi, lo, mid;
};
__device__ __forceinline__ Frag frag_packed(float4 s0, float4 s1) {  // two packed slots of the same row -> fragment (register renaming only)
    const uint4v h = {__float_as_uint(s0.x), __float_as_uint(s0.y), __float_as_uint(s1.x), __float_as_uint(s1.y)};
    const uint4v l = {__float_as_uint(s0.z), __float_as_uint(s0.w), __float_as_uint(s1.z), __float_as_uint(s1.w)};
    return Frag{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, l), __builtin_bit_cast(bf16x8, l)};
}
// three-way split of 8 fp32 values: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (the residues are exact in fp32)
__device__ __forceinline__ Frag frag_split3(float4 a, float4 b) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint4v h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned hj = pk_bf16(v[2 * j], v[2 * j + 1]);
        const float r0 = v[2 * j] - __uint_as_float(hj << 16), r1 = v[2 * j + 1] - __uint_as_float(hj & 0xffff0000u);
        const unsigned mj = pk_bf16(r0, r1);
        const unsigned lj = pk_bf16(r0 - __uint_as_float(mj << 16), r1 - __uint_as_float(mj & 0xffff0000u));
        h[j] = hj, m[j] = mj, l[j] = lj;
    }
    return Frag{__builtin_bit_cast(bf16x8, h), __builtin_bit_cast(bf16x8, l), __builtin_bit_cast(bf16x8, m)};
}
template <int NT>
__device__ __forceinline__ Frag frag_f32(float4 a, float4 b) {  // two fp32 k-quads of the same row -> fragment, packed in registers
    if constexpr (NT == 6) return frag_split3(a, b);
    return frag_packed(pack4<NT>(a), pack4<NT>(b));
}
// fragment from two 16-byte LDS / register slots of the kernel's operand tile: packed slots for NT = 1, 3; plain fp32 k-quads for NT = 6
template <int NT>
__device__ __forceinline__ Frag frag_lds(float4 s0, float4 s1) {
    if constexpr (NT == 6) return frag_split3(s0, s1);
    return frag_packed(s0, s1);
}
template <int NT>
__device__ __forceinline__ void mma32(floatx16& acc, const Frag& a, const Frag& b) {
    if constexpr (NT == 6) {  // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, acc, 0, 0, 0);
    }
    if constexpr (NT == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
// first product of an accumulator chain: C = inline-constant zero (no register clearing)
template <int NT>
__device__ __forceinline__ floatx16 mma32_first(const Frag& a, const Frag& b) {
    floatx16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    floatx16 acc;
    if constexpr (NT == 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, z, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
    } else if constexpr (NT == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, z, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, z, 0, 0, 0);
    }
    return acc;
}
template <int NT>
__device__ __forceinline__ void mma16(floatx4& acc, const Frag& a, const Frag& b) {  // 16x16x32: lane group kk = lane >> 4 supplies 8 k values
    if constexpr (NT == 6) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.mid, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.mid, acc, 0, 0, 0);
    }
    if constexpr (NT == 3) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}

// mma_block on PACKED LDS tiles (same addressing as mma_block; kdepth a multiple of 16).
template <int NT, int WM, int WN, int BT = 32, int AT = 32>
__device__ __forceinline__ void mma_block_p(floatx16 (&acc)[WM][WN], const float* As, int lda, const float* Bs, int ldb, int kdepth) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, kh = lane >> 5;
    const float* ap = As + i * lda + kh * 4;
    const float* bp = Bs + i * ldb + kh * 4;
#pragma unroll 2
    for (int q = 0; q < kdepth; q += 16) {
        Frag a[WM], b[WN];
#pragma unroll
        for (int m = 0; m < WM; ++m) a[m] = frag_lds<NT>(ld4(ap + btile_row<AT>(m) * lda + q), ld4(ap + btile_row<AT>(m) * lda + q + 8));
#pragma unroll
        for (int n = 0; n < WN; ++n) b[n] = frag_lds<NT>(ld4(bp + btile_row<BT>(n) * ldb + q), ld4(bp + btile_row<BT>(n) * ldb + q + 8));
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) mma32<NT>(acc[m][n], a[m], b[n]);
    }
}
// precision dispatch used by the kernels that call mma_block: NT = 0 -> the fp32 block, else the packed block
template <int NT, int WM, int WN, int BT = 32, int AT = 32>
__device__ __forceinline__ void mma_block_nt(floatx16 (&acc)[WM][WN], const float* As, int lda, const float* Bs, int ldb, int kdepth) {
    if constexpr (NT == 0)
        mma_block<WM, WN, BT, AT>(acc, As, lda, Bs, ldb, kdepth);
    else
        mma_block_p<NT, WM, WN, BT, AT>(acc, As, lda, Bs, ldb, kdepth);
}

// mma_block_bn with a PACKED A tile (k-contiguous weights) and a plain fp32 n-contiguous B tile (packed in registers)
template <int NT, int WM, int WN>
__device__ __forceinline__ void mma_block_bn_p(floatx16 (&acc)[WM][WN], const float* As, int lda, const float* Bs, int ldb, int kdepth) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, kh = lane >> 5;
    const float* ap = As + i * lda + kh * 4;
    const float* bp = Bs + (kh * 4) * ldb + i;
#pragma unroll 2
    for (int q = 0; q < kdepth; q += 16) {
        Frag a[WM], b[WN];
#pragma unroll
        for (int m = 0; m < WM; ++m) a[m] = frag_lds<NT>(ld4(ap + m * 32 * lda + q), ld4(ap + m * 32 * lda + q + 8));
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const float* c0 = bp + q * ldb + n * 32;
            b[n] = frag_f32<NT>(f4(c0[0], c0[ldb], c0[2 * ldb], c0[3 * ldb]), f4(c0[8 * ldb], c0[9 * ldb], c0[10 * ldb], c0[11 * ldb]));
        }
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) mma32<NT>(acc[m][n], a[m], b[n]);
    }
}

template <int WM, int WN>
__device__ __forceinline__ void acc_zero(floatx16 (&acc)[WM][WN]) {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
}

// register group g (0..3) of a 32x32 accumulator tile: rows 8g + 4*(lane>>5) .. +3 of column lane&31
__device__ __forceinline__ float4 acc_group(const floatx16& a, int g) { return f4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]); }

// row of accumulator register r inside its 32x32 tile, for this lane
__device__ __forceinline__ int acc_row(int r) { return (r & 3) + 8 * (r >> 2) + 4 * ((threadIdx.x & 63) >> 5); }

// Copy a [rows][BK] k-chunk of a k-contiguous global matrix W[row][ldw] into LDS Bs[row][lds].
// All 256 threads participate; BK/4 lanes cover one row (coalesced 128/256-byte segments).
template <int ROWS, int BK>
struct ChunkRegs {
    static constexpr int kPerThread = ROWS * (BK / 4) / 256;
    float4 v[kPerThread];
    __device__ __forceinline__ void load(const float* __restrict__ W, int ldw, int k0) {
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            int idx = threadIdx.x + i * 256;
            int row = idx / (BK / 4), c4 = idx % (BK / 4);
            v[i] = ld4(W + (size_t)row * ldw + k0 + c4 * 4);
        }
    }
    __device__ __forceinline__ void store(float* Bs, int lds) const {
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            int idx = threadIdx.x + i * 256;
            int row = idx / (BK / 4), c4 = idx % (BK / 4);
            st4(Bs + row * lds + c4 * 4, v[i]);
        }
    }
};

}  // namespace rtfs
