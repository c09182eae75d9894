// The dual-path SRU stage of the RTFS block: DualPathRNN.forward (/root/reference/src/models/layers/
// rnn_layers.py:136-162) for dim=4 (sequences along F, one per (b,t)) and dim=3 (along T, one per (b,f)).
//
//   rtfs_dp_unfold_gemm_fwd   LN4D over C  ->  nn.Unfold((8,1))  ->  X . W0     (rnn_layers.py:146-149 + SRU layer 0)
//   rtfs_sru_scan_fwd         the bidirectional SRU recurrence                  (sru package; oracle/sru_ref.py)
//   rtfs_dp_convt_fwd         ConvTranspose1d(64->64,k=8) + bias + residual     (rnn_layers.py:129,153-156)
//   (SRU layers 1-3 input projections: rtfs_gemm_rows_fwd in gemm.hip)
//
// Unfold is never materialised.  In channels-last layout the 8-position window starting at position l of a
// sequence is rows l..l+7 of a [positions][64] slab, i.e. the unfolded matrix is a Toeplitz VIEW of the slab:
//      X[l][kk*64 + c] = slab[l + kk][c]
// (feature order (kk,c) instead of the reference's (c,kk): the host permutes W0's rows once).  The slab is
// layer-normalised once into LDS and the MFMA A operand is read from it with a row offset per k-chunk.
// ConvTranspose1d is the same structure on the zero-padded SRU output: y[n] = sum_k' hpad[n + k'] . W'[k'].
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace rtfs {

struct SeqMap {  // sequence s -> base element offset; positions are pos_stride apart; channels contiguous
    int seq_div;
    long long stride_hi, stride_lo, pos_stride;
    int npos;  // positions per sequence (F2 or T2)
    int L;     // npos - 8 + 1 windows
    __device__ __forceinline__ size_t base(int s) const { return (size_t)(s / seq_div) * stride_hi + (size_t)(s % seq_div) * stride_lo; }
    // 32-bit BYTE offset of (sequence s, position pos): seq_div is 1 or 64 (a shift), the whole tensor is < 2^32 bytes whenever the large-batch
    // kernel runs (checked on the host) - the staging code of the layer-0 GEMM issued ~60 VALU instructions per 16-byte load with the general form
    int seq_shift;
    unsigned magicL;  // floor(2^32 / L) + 1: n / L == __umulhi(n, magicL) for n L < 2^32 (checked on the host)
    __device__ __forceinline__ unsigned off32(int s, int pos) const {
        return (((unsigned)(s >> seq_shift) * (unsigned)stride_hi + (unsigned)(s & (seq_div - 1)) * (unsigned)stride_lo) + (unsigned)pos * (unsigned)pos_stride) * 4u;
    }
};

constexpr int kSlabRows = 64 + 7;
constexpr int kSlabLd = 68;

// MODE 0: slab = LN4D(G) rows m0..m0+70;  out U0[S][L][256]
// MODE 1: slab = zero-padded h3 rows (m0-7)..(m0+63);  out G[pos] = acc + bias + G[pos]  (in place)
// LDO: row stride of the MODE 0 output.  LDO > N: the workgroup computes column block blockIdx.z (N of the LDO columns) - batch-1-sized launches put four
// 64-column workgroups where one 256-column workgroup walked its 1024 MFMAs per wave alone (round 4).
template <int N, int WM, int WN, int BK, int MODE, int NT = 0, int LDO = N>  // NT != 0 (common.h): Wt host-PACKED, slab packed on store
__global__ __launch_bounds__(256) void toeplitz_gemm_kernel(SeqMap map, const float* __restrict__ src, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ Wt,
                                                            const float* __restrict__ bias, float* __restrict__ dst) {
    constexpr int LDB = BK + 4;
    constexpr int WGN = N / (32 * WN), WGM = 4 / WGN;
    static_assert(WGM * WM * 32 == 64, "workgroup tile is 64 rows");
    __shared__ __attribute__((aligned(16))) float slab[kSlabRows * kSlabLd];
    __shared__ __attribute__((aligned(16))) float Bs[2][N * LDB];

    const int s = blockIdx.y, m0 = blockIdx.x * 64;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = w / WGN, wn = w % WGN;
    const size_t sbase = map.base(s);
    if (LDO > N) Wt += (size_t)blockIdx.z * N * 512, dst += blockIdx.z * N;

    ChunkRegs<N, BK> breg;
    breg.load(Wt, 512, 0);

    // ---- slab ----
#pragma unroll
    for (int it = 0; it < (kSlabRows * 16 + 255) / 256; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int row = idx >> 4, c4 = idx & 15;
        float4 v = f4(0, 0, 0, 0);
        if (MODE == 0) {
            const int pos = m0 + row;
            const bool ok = row < kSlabRows && pos < map.npos;
            v = ld4(src + sbase + (size_t)min(pos, map.npos - 1) * map.pos_stride + c4 * 4);
            // LayerNormalization4D over the 64 channels of this position (normalizations.py:33-37)
            float sum = v.x + v.y + v.z + v.w;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            const float mean = sum * (1.f / 64.f);
            float4 d = f4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
            float sq = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            const float rstd = 1.0f / sqrtf(sq * (1.f / 64.f) + kEps);
            v = ok ? fma4(d * rstd, ld4(gamma + c4 * 4), ld4(beta + c4 * 4)) : f4(0, 0, 0, 0);
        } else {
            const int l = m0 + row - 7;
            v = ld4(src + ((size_t)s * map.L + min(max(l, 0), map.L - 1)) * 64 + c4 * 4);  // clamped + select: the slab loads stay batched
            if (!(row < kSlabRows && l >= 0 && l < map.L)) v = f4(0, 0, 0, 0);
        }
        if (row < kSlabRows) st4(slab + row * kSlabLd + c4 * 4, pack4<NT>(v));
    }
    breg.store(Bs[0], LDB);
    __syncthreads();

    floatx16 acc[WN][WM];  // [weight tile][row tile]: lanes = sequence positions, registers = output channels
    acc_zero(acc);
    constexpr int NK = 512 / BK;
#pragma unroll 1
    for (int kc = 0; kc < NK; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < NK) breg.load(Wt, 512, (kc + 1) * BK);
        const int k0 = kc * BK, kk = k0 >> 6, c0 = k0 & 63;
        mma_block_nt<NT, WN, WM>(acc, Bs[cur] + wn * WN * 32 * LDB, LDB, slab + (wm * WM * 32 + kk) * kSlabLd + c0, kSlabLd, BK);
        if (kc + 1 < NK) breg.store(Bs[cur ^ 1], LDB);
        __syncthreads();
    }

#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int row = m0 + (wm * WM + m) * 32 + (lane & 31);
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = (wn * WN + n) * 32 + 8 * g + 4 * (lane >> 5);
                const float4 v = acc_group(acc[n][m], g);
                if (MODE == 0) {
                    if (row < map.L) st4(dst + ((size_t)s * map.L + row) * LDO + col, v);
                } else {
                    if (row < map.npos) {
                        float* o = dst + sbase + (size_t)row * map.pos_stride + col;
                        st4(o, v + ld4(bias + col) + ld4(o));
                    }
                }
            }
    }
}


// ConvTranspose1d (rnn_layers.py:153-156) as a Toeplitz GEMM with TWO 64-row slabs per workgroup (two sequences, or the two halves of
// one): G[pos] += sum_{k', j} h3[pos + k' - 7][j] * Wt[c][k'*64 + j] + bias[c].  Against the 64-row toeplitz_gemm_kernel every 64 x 64
// weight stage now feeds 128 rows (half the weight bytes and barriers per flop) and a wave owns 1 weight tile x 2 row tiles
// (3 LDS reads per 8 MFMAs instead of 2 per 4).  Slab loads are unconditional (clamped + select).
template <int NT = 0>
__global__ __launch_bounds__(256, 2) void convt_gemm2_kernel(SeqMap map, const float* __restrict__ src, const float* __restrict__ Wt,
                                                             const float* __restrict__ bias, float* __restrict__ dst, int tiles_per_seq, int total_tiles) {
    constexpr int BK = 64, LDB = BK + 4, N = 64;
    __shared__ __attribute__((aligned(16))) float slab[2][kSlabRows * kSlabLd];
    __shared__ __attribute__((aligned(16))) float Bs[2][N * LDB];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int st = w >> 1, wn = w & 1;  // slab of this wave, column half (32 channels)
    int seq[2], m0[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int gt = min((int)blockIdx.x * 2 + q, total_tiles - 1);
        seq[q] = gt / tiles_per_seq;
        m0[q] = (gt - seq[q] * tiles_per_seq) * 64;
    }
    ChunkRegs<N, BK> breg;
    breg.load(Wt, 512, 0);
    constexpr int NIT = (2 * kSlabRows * 16 + 255) / 256;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int q = idx >= kSlabRows * 16 ? 1 : 0;
        const int row = (idx - q * kSlabRows * 16) >> 4, c4 = (idx & 15) * 4;
        const int l = m0[q] + row - 7;
        float4 v = ld4(src + ((size_t)seq[q] * map.L + min(max(l, 0), map.L - 1)) * 64 + c4);
        if (!(l >= 0 && l < map.L)) v = f4(0, 0, 0, 0);
        if (idx < 2 * kSlabRows * 16) st4(slab[q] + row * kSlabLd + c4, pack4<NT>(v));
    }
    breg.store(Bs[0], LDB);
    // the residual rows this lane will add to (G, in place) are fetched here, clamped and unconditional, so that their latency runs under
    // the 8 k-chunks instead of in front of the stores
    const size_t sbase = map.base(seq[st]);
    float4 res[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int row = min(m0[st] + m * 32 + (lane & 31), map.npos - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) res[m][g] = ld4(dst + sbase + (size_t)row * map.pos_stride + wn * 32 + 8 * g + 4 * (lane >> 5));
    }
    __syncthreads();
    floatx16 acc[1][2];
    acc_zero(acc);
#pragma unroll 1
    for (int kc = 0; kc < 8; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < 8) breg.load(Wt, 512, (kc + 1) * BK);
        mma_block_nt<NT, 1, 2>(acc, Bs[cur] + wn * 32 * LDB, LDB, slab[st] + kc * kSlabLd, kSlabLd, BK);
        if (kc + 1 < 8) breg.store(Bs[cur ^ 1], LDB);
        __syncthreads();
    }
    if ((int)blockIdx.x * 2 + st < total_tiles) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = m0[st] + m * 32 + (lane & 31);
            if (row < map.npos) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = wn * 32 + 8 * g + 4 * (lane >> 5);
                    float* o = dst + sbase + (size_t)row * map.pos_stride + col;
                    st4(o, acc_group(acc[0][m], g) + ld4(bias + col) + res[m][g]);
                }
            }
        }
    }
}

// Layer-0 kernel, second generation: workgroup tile 128 rows x 256 columns = TWO 64-row slabs (two frequency
// sequences, or the two halves of one time sequence), 4 waves as 2 (slab) x 2 (column half), each wave 64 x 128
// (4 x 2 accumulator tiles = 128 registers).  Against a 64 x 256 tile this halves the weight bytes staged per
// flop and the barriers per flop; BK = 16 keeps LDS at 79.6 KB so two workgroups share a CU.
// PERSISTENT: a workgroup walks tile pairs blockIdx.x, blockIdx.x + gridDim.x, ...; the raw slab rows of the
// next pair are fetched into registers before the current pair's 32 k-chunks and layer-normalised into LDS after
// its write-back, so slab latency and most of the prologue disappear from the MFMA timeline.
__global__ __launch_bounds__(256, 2) void unfold_gemm128_kernel(SeqMap map, const float* __restrict__ src, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ Wt, float* __restrict__ dst,
                                                                int tiles_per_seq, int total_tiles) {
    constexpr int BK = 16, LDB = BK + 4, N = 256;
    constexpr int NIT = (2 * kSlabRows * 16 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float slab[2][kSlabRows * kSlabLd];
    __shared__ __attribute__((aligned(16))) float Bs[2][N * LDB];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = w >> 1, wn = w & 1;
    const float4 g4 = ld4(gamma + (threadIdx.x & 15) * 4), b4 = ld4(beta + (threadIdx.x & 15) * 4);

    ChunkRegs<N, BK> breg;
    float4 sraw[NIT];
    int seq[2], m0[2];
    auto locate = [&](int pair) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int gt = min(pair * 2 + st, total_tiles - 1);
            seq[st] = gt / tiles_per_seq;
            m0[st] = (gt - seq[st] * tiles_per_seq) * 64;
        }
    };
    // raw rows of both slabs of `pair` -> registers (rows past the sequence end are clamped and zeroed at store time)
    auto fetch_slabs = [&](int pair) {
        int sq[2], mm[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int gt = min(pair * 2 + st, total_tiles - 1);
            sq[st] = gt / tiles_per_seq;
            mm[st] = (gt - sq[st] * tiles_per_seq) * 64;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int st = idx >= kSlabRows * 16 ? 1 : 0;
            const int row = (idx - st * kSlabRows * 16) >> 4;
            const int pos = min(mm[st] + row, map.npos - 1);
            sraw[it] = ld4(src + map.base(sq[st]) + (size_t)pos * map.pos_stride + (threadIdx.x & 15) * 4);
        }
    };
    // LayerNormalization4D over the 64 channels of each position (normalizations.py:33-37) -> LDS slabs
    auto store_slabs = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int st = idx >= kSlabRows * 16 ? 1 : 0;
            const int row = (idx - st * kSlabRows * 16) >> 4;
            const bool inr = idx < 2 * kSlabRows * 16;
            const bool ok = inr && (m0[st] + row < map.npos);
            const float4 v = sraw[it];
            float sum = v.x + v.y + v.z + v.w;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            const float mean = sum * (1.f / 64.f);
            const float4 d = f4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
            float sq = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            const float rstd = 1.0f / sqrtf(sq * (1.f / 64.f) + kEps);
            const float4 y = ok ? fma4(d * rstd, g4, b4) : f4(0, 0, 0, 0);
            if (inr) st4(slab[st] + row * kSlabLd + (threadIdx.x & 15) * 4, y);
        }
    };

    int pair = blockIdx.x;
    const int npairs = (total_tiles + 1) / 2;
    breg.load(Wt, 512, 0);
    fetch_slabs(pair);
    locate(pair);
    store_slabs();
    breg.store(Bs[0], LDB);
    __syncthreads();

    constexpr int NK = 512 / BK;
#pragma unroll 1
    while (true) {
        const int next = pair + gridDim.x;
        const bool has_next = next < npairs;
        if (has_next) fetch_slabs(next);
        floatx16 acc[4][2];  // [weight tile][row tile]
        acc_zero(acc);
#pragma unroll 1
        for (int kc = 0; kc < NK; ++kc) {
            const int cur = kc & 1;
            if (kc + 1 < NK) breg.load(Wt, 512, (kc + 1) * BK);
            const int k0 = kc * BK, kk = k0 >> 6, c0 = k0 & 63;
            mma_block<4, 2>(acc, Bs[cur] + wn * 128 * LDB, LDB, slab[wm] + kk * kSlabLd + c0, kSlabLd, BK);
            if (kc + 1 < NK) breg.store(Bs[cur ^ 1], LDB);
            __syncthreads();
        }
        if (has_next) breg.load(Wt, 512, 0);
        if (pair * 2 + wm < total_tiles) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int row = m0[wm] + m * 32 + (lane & 31);
                if (row < map.L) {
                    float* o = dst + ((size_t)seq[wm] * map.L + row) * N + wn * 128 + 4 * (lane >> 5);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int g = 0; g < 4; ++g) st4(o + n * 32 + 8 * g, acc_group(acc[n][m], g));
                }
            }
        }
        if (!has_next) break;
        pair = next;
        locate(pair);
        store_slabs();  // every wave left the last k-chunk (barrier above): the slabs and Bs[0] are free
        breg.store(Bs[0], LDB);
        __syncthreads();
    }
}

// Layer-0 kernel, third generation: the 64-row tiles are cut from the FLATTENED row index r = s * L + l, so no MFMA row is spent on
// the padding of a sequence to a multiple of 64 (L = 57: 64 -> 57 rows per sequence, -10.9 % of the MFMAs; L = 118: 128 -> 118,
// -7.8 %).  A tile then covers up to three sequences: its slab holds, per sequence segment, the segment's rows plus the 7 halo rows
// (<= 64 + 21 rows), and output row i reads slab row i + 7 g(i) + tap, g = the segment of row i (a per-lane constant).  U0 is
// [S][L][256] = [S*L][256], so the write-back is one contiguous block of rows.
// Work unit = (tile pair, 128-column half): a workgroup owns a CONTIGUOUS range of units (balanced to one unit), stages the pair's two
// slabs once (LayerNormalization4D on the way in) and runs the K loop once per column half with a 64 x 64 wave tile - 64 accumulator
// registers instead of 128, so the fragment reads of a k step are no longer funnelled through one register quad, and the tail of the
// launch is quantised in half-size units.  BK = 32, weight chunk unpadded (128-byte rows, 16-byte slot XOR-swizzled with row & 7),
// 78 KB of LDS: two workgroups per CU.  The raw rows of the next pair are fetched into registers under the last K loop of the
// current one.
constexpr int kFlatRows = 64 + 21;
// Bank-conflict-free segment jumps (round 3): consecutive output rows of a wave read consecutive slab rows, 17 sixteen-byte slots apart, so
// eight / sixteen neighbouring lanes hit distinct slot residues - except across a sequence boundary, where the slab row jumps by the 7 halo
// rows and two lanes of the group met in one bank (PMC, round 2: 28 % of this kernel's LDS cycles were conflict cycles).  Segment g of a
// slab is therefore stored 9 g slots further on: the jump becomes 7 x 17 + 9 = 128 slots = 0 (mod 16) and the residues continue as if the
// rows were consecutive.  Costs 288 bytes of LDS per slab and one per-lane constant.
constexpr int kSegSkew = 36;  // floats (= 9 slots) per segment index
struct FlatTile {  // geometry of one 64-row tile (wave-uniform)
    int r0, s0, l0, n0, n1;  // first flattened row, its (sequence, position); rows of segment 0, 1 (segment 2 = the rest)
};
template <int NT = 0>
__global__ __launch_bounds__(256, 2) void unfold_gemm128f_kernel(SeqMap map, const float* __restrict__ src, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ Wt, float* __restrict__ dst,
                                                                 int S, int total_tiles) {
    constexpr int BK = 32, N = 256, NP = 128;  // columns per pass
    constexpr int NIT = (2 * kFlatRows * 16 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float slab[2][kFlatRows * kSlabLd + 2 * kSegSkew];
    __shared__ __attribute__((aligned(16))) float Bs[2][NP * BK];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const float4 g4 = ld4(gamma + (threadIdx.x & 15) * 4), b4 = ld4(beta + (threadIdx.x & 15) * 4);
    const int L = map.L;
    const long long R = (long long)S * L;

    auto tile_of = [&](int gt) {
        FlatTile t;
        t.r0 = gt * 64;
        t.s0 = (int)__umulhi((unsigned)t.r0, map.magicL);  // = r0 / L (exact: S L^2 < 2^32, checked by the launcher)
        t.l0 = t.r0 - t.s0 * L;
        t.n0 = min(L - t.l0, 64);
        t.n1 = min(L, 64 - t.n0);
        return t;
    };
    // weight chunk of a pass: global -> registers -> LDS (row n = 32 floats = eight 16-byte slots, slot c stored at c ^ (n & 7))
    float4 breg[4];
    auto load_b = [&](int np, int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = threadIdx.x + it * 256, row = idx >> 3, c = idx & 7;
            breg[it] = ld4(Wt + (size_t)(np * NP + row) * 512 + k0 + c * 4);
        }
    };
    auto store_b = [&](float* B) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = threadIdx.x + it * 256, row = idx >> 3, c = idx & 7;
            st4(B + row * BK + ((c ^ (row & 7)) << 2), breg[it]);
        }
    };
    float4 sraw[NIT];
    // slab row j of a tile -> (sequence, position, valid)
    auto slab_row = [&](const FlatTile& t, int j, int& sq, int& pos, int& g) {
        const int e0 = t.n0 + 7, e1 = e0 + t.n1 + 7, n2 = 64 - t.n0 - t.n1;
        g = (j >= e0) + (j >= e1);
        const int jj = j - (g == 0 ? 0 : (g == 1 ? e0 : e1));
        const int ng = g == 0 ? t.n0 : (g == 1 ? t.n1 : n2);
        sq = t.s0 + g;
        pos = (g == 0 ? t.l0 : 0) + jj;
        return ng > 0 && jj < ng + 7 && sq < S && pos < map.npos;
    };
    auto fetch_slabs = [&](int pair) {
        FlatTile t[2] = {tile_of(min(pair * 2, total_tiles - 1)), tile_of(min(pair * 2 + 1, total_tiles - 1))};
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int st = idx >= kFlatRows * 16 ? 1 : 0;
            const int j = min((idx - st * kFlatRows * 16) >> 4, kFlatRows - 1);
            int sq, pos, g;
            slab_row(t[st], j, sq, pos, g);
            sraw[it] = ld4_off(src, map.off32(min(sq, S - 1), min(pos, map.npos - 1)) + (threadIdx.x & 15) * 16u);
        }
    };
    // LayerNormalization4D over the 64 channels of each position (normalizations.py:33-37) -> LDS slabs
    auto store_slabs = [&](int pair) {
        FlatTile t[2] = {tile_of(min(pair * 2, total_tiles - 1)), tile_of(min(pair * 2 + 1, total_tiles - 1))};
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int st = idx >= kFlatRows * 16 ? 1 : 0;
            const int j = (idx - st * kFlatRows * 16) >> 4;
            const bool inr = idx < 2 * kFlatRows * 16;
            int sq, pos, g;
            const bool ok = slab_row(t[st], min(j, kFlatRows - 1), sq, pos, g) && inr;
            const float4 v = sraw[it];
            float sum = v.x + v.y + v.z + v.w;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            const float mean = sum * (1.f / 64.f);
            const float4 d = f4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
            float sqs = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sqs += __shfl_xor(sqs, o, 64);
            const float rstd = 1.0f / sqrtf(sqs * (1.f / 64.f) + kEps);
            const float4 y = ok ? fma4(d * rstd, g4, b4) : f4(0, 0, 0, 0);
            if (inr) st4(slab[st] + j * kSlabLd + g * kSegSkew + (threadIdx.x & 15) * 4, pack4<NT>(y));
        }
    };

    // this workgroup's units [u0, u1): unit u = (pair u >> 1, column half u & 1)
    const int npairs = (total_tiles + 1) / 2;
    const long long U = 2LL * npairs;
    // Balanced over CUs, not over workgroups: the grid is two resident workgroups per CU, and workgroups b and b + gridDim.x / 2 are dispatched to
    // the same CU in practice (a placement assumption that only affects speed).  The units are first split evenly over the gridDim.x / 2 CU slots and
    // each slot's share is then halved - with 7.375 units per workgroup (time path at B = 32) the busiest CU carries 15 units instead of 16.
    int u0, u1;
    if ((gridDim.x & 1) == 0) {
        const int half = gridDim.x >> 1, c = blockIdx.x % half, which = blockIdx.x / half;
        const int c0 = (int)(U * c / half), c1 = (int)(U * (c + 1) / half), first = (c1 - c0 + 1) >> 1;
        u0 = which == 0 ? c0 : c0 + first;
        u1 = which == 0 ? c0 + first : c1;
    } else {
        u0 = (int)(U * blockIdx.x / gridDim.x), u1 = (int)(U * (blockIdx.x + 1) / gridDim.x);
    }
    if (u0 >= u1) return;
    load_b(u0 & 1, 0);
    fetch_slabs(u0 >> 1);
    store_slabs(u0 >> 1);
    store_b(Bs[0]);
    __syncthreads();

    const int sw = i & 7;
    constexpr int NK = 512 / BK;
#pragma unroll 1
    for (int u = u0; u < u1; ++u) {
        const int pair = u >> 1, np = u & 1;
        const bool has_next = u + 1 < u1;
        const bool new_pair = has_next && ((u + 1) >> 1) != pair;
        if (new_pair) fetch_slabs(pair + 1);
        const FlatTile t = tile_of(min(pair * 2 + wm, total_tiles - 1));
        int prow[2];  // slab offset (floats) of this lane's output row in the wave's two row tiles: row ri + 7 g, skewed by its segment g
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int ri = 32 * m + i, g = (ri >= t.n0) + (ri >= t.n0 + t.n1);
            prow[m] = (ri + 7 * g) * kSlabLd + g * kSegSkew;
        }
        floatx16 acc[2][2];  // [weight tile][row tile]
        acc_zero(acc);
#pragma unroll 1
        for (int kc = 0; kc < NK; ++kc) {
            const int cur = kc & 1;
            if (kc + 1 < NK) load_b(np, (kc + 1) * BK);
            const int k0 = kc * BK, kk = k0 >> 6, c0 = k0 & 63;
            const float* ap = Bs[cur] + (wn * 64 + i) * BK;
            const float* sp = slab[wm] + kk * kSlabLd + c0 + 4 * kh;
            if constexpr (NT == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 a[2], b[2];
    #pragma unroll
                    for (int n = 0; n < 2; ++n) a[n] = ld4(ap + n * 32 * BK + (((2 * q + kh) ^ sw) << 2));
    #pragma unroll
                    for (int m = 0; m < 2; ++m) b[m] = ld4(sp + prow[m] + 8 * q);
    #pragma unroll
                    for (int n = 0; n < 2; ++n)
    #pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[n].x, b[m].x, acc[n][m], 0, 0, 0);
                            acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[n].y, b[m].y, acc[n][m], 0, 0, 0);
                            acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[n].z, b[m].z, acc[n][m], 0, 0, 0);
                            acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[n].w, b[m].w, acc[n][m], 0, 0, 0);
                        }
                }
            } else {
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {  // 16 k per step: packed slots (4 q2 + kh) and (4 q2 + 2 + kh) of every row
                    Frag a[2], b[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        a[n] = frag_lds<NT>(ld4(ap + n * 32 * BK + (((4 * q2 + kh) ^ sw) << 2)), ld4(ap + n * 32 * BK + (((4 * q2 + 2 + kh) ^ sw) << 2)));
#pragma unroll
                    for (int m = 0; m < 2; ++m) b[m] = frag_lds<NT>(ld4(sp + prow[m] + 16 * q2), ld4(sp + prow[m] + 16 * q2 + 8));
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int m = 0; m < 2; ++m) mma32<NT>(acc[n][m], a[n], b[m]);
                }
            }
            if (kc + 1 < NK) store_b(Bs[cur ^ 1]);
            __syncthreads();
        }
        if (has_next) load_b((u + 1) & 1, 0);
        if (pair * 2 + wm < total_tiles) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const long long r = (long long)t.r0 + 32 * m + i;
                if (r < R) {
                    float* o = dst + r * N + np * NP + wn * 64 + 4 * kh;
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int g = 0; g < 4; ++g) st4(o + n * 32 + 8 * g, acc_group(acc[n][m], g));
                }
            }
        }
        if (!has_next) break;
        if (new_pair) store_slabs(pair + 1);  // every wave left the last k-chunk (barrier above): the slabs and Bs[0] are free
        store_b(Bs[0]);
        __syncthreads();
    }
}

// Layer-0 kernel, fourth generation (round 3), large batches, fp32: WEIGHT-STATIONARY.  The kernels above re-stage W0 (512 KB) through LDS for
// every tile pair - 16 chunks x 2 column halves, one barrier each - and reach 77 % matrix-pipe occupancy.  Here a workgroup keeps one
// 128-column half of W0 in registers for its whole life (wave w: columns 128 h + 32 w .. + 31 x all 512 k = 256 registers per lane, the MFMA
// A operand; two workgroups - two CUs - share a range of row tiles, one per column half) and walks 64-row tiles cut from the FLATTENED row
// index exactly as unfold_gemm128f_kernel does (same tile geometry, same slab layout incl. the segment skew).  The K loop of a tile - 512
// MFMAs per wave, B fragments read from the LayerNorm-ed slab one step ahead - never stops, not even between tiles: the LayerNorm staging of
// the next tile, the write-back of the previous one, the global loads of the tile after next and the two barriers of a tile all ride between
// its MFMAs (schedule at the tile loop below).  Stores go through a buffer descriptor without a branch and loads are consumed before the
// tile's stores are issued: vmcnt counts both, see ws256_kernel in gemm.hip.  Same products in the same k order per accumulator as the
// LDS-staged kernels (bit-identical U0 with the IEEE 1 / sqrt in the LayerNorm; the shipped form uses v_rsq_f32, see stage_b).
template <int NT = 0>  // 0 fp32; 1 / 3: bf16 / split-bf16 MFMA (common.h; Wt host-PACKED, the slab packed on store)
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void unfold_ws_kernel(SeqMap map, const float* __restrict__ src, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ Wt, float* __restrict__ dst, int S,
                                                           int total_tiles) {
    constexpr int NIT = (kFlatRows * 16 + 255) / 256;  // 6
    __shared__ __attribute__((aligned(16))) float slab[2][(kFlatRows + 1) * kSlabLd + 2 * kSegSkew];  // (+ one scratch row)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int half = blockIdx.x & 1, slot = blockIdx.x >> 1, nslots = gridDim.x >> 1;
#ifdef UW_TIMING
    const unsigned long long uw_t0 = __builtin_amdgcn_s_memtime();
#endif
    const int c4 = (threadIdx.x & 15) * 4;
    const int cst = NT == 0 ? c4 : (c4 >> 4) * 16 + ((c4 >> 2) & 1) * 4 + ((c4 >> 3) & 1) * 2;  // float offset of this thread's channel quad inside a slab row
    const float4 g4 = ld4(gamma + c4), b4 = ld4(beta + c4);
    const int L = map.L;

    float4 wf[64];  // W0 fragments: row n = 128 half + 32 w + i, k = 8 q + 4 kh .. +3  (k = 64 tap + channel)
#pragma unroll
    for (int q = 0; q < 64; ++q) wf[q] = ld4(Wt + (size_t)(128 * half + 32 * w + i) * 512 + 8 * q + 4 * kh);
    // bf16 modes: the host-packed slots [hi(k0 k1) hi(k2 k3) lo(k0 k1) lo(k2 k3)] of step q2 = quads 2 q2, 2 q2 + 1 regrouped ONCE into the
    // 4-register operand tuples of v_mfma_f32_32x32x16_bf16 (hi and lo planes of the 8 k values 16 q2 + 4 kh + {0..3, 8..11})
    bf16x8 whi[NT ? 32 : 1], wlo[NT ? 32 : 1];
    if constexpr (NT != 0) {
#pragma unroll
        for (int q2 = 0; q2 < 32; ++q2) {
            const float4 s0 = wf[2 * q2], s1 = wf[2 * q2 + 1];
            whi[q2] = __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(s0.x), __float_as_uint(s0.y), __float_as_uint(s1.x), __float_as_uint(s1.y)});
            wlo[q2] = __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(s0.z), __float_as_uint(s0.w), __float_as_uint(s1.z), __float_as_uint(s1.w)});
        }
    }

    const int t0 = (int)((long long)total_tiles * slot / nslots), t1 = (int)((long long)total_tiles * (slot + 1) / nslots);
    if (t0 >= t1) return;
    auto tile_of = [&](int gt) {
        FlatTile t;
        t.r0 = gt * 64;
        t.s0 = (int)__umulhi((unsigned)t.r0, map.magicL);  // = r0 / L (exact: S L^2 < 2^32, checked by the launcher)
        t.l0 = t.r0 - t.s0 * L;
        t.n0 = min(L - t.l0, 64);
        t.n1 = min(L, 64 - t.n0);
        return t;
    };
    // slab row j of a tile -> (sequence, position, segment, valid)
    auto slab_row = [&](const FlatTile& t, int j, int& sq, int& pos, int& g) {
        const int e0 = t.n0 + 7, e1 = e0 + t.n1 + 7, n2 = 64 - t.n0 - t.n1;
        g = (j >= e0) + (j >= e1);
        const int jj = j - (g == 0 ? 0 : (g == 1 ? e0 : e1));
        const int ng = g == 0 ? t.n0 : (g == 1 ? t.n1 : n2);
        sq = t.s0 + g;
        pos = (g == 0 ? t.l0 : 0) + jj;
        return ng > 0 && jj < ng + 7 && sq < S && pos < map.npos;
    };
    float4 sraw[NIT];
    unsigned sinfo[NIT];  // where a fetched row goes: float offset in its slab (scratch row behind the slab for rows past it)
    FlatTile tf;  // geometry of the tile being fetched
    auto fetch_begin = [&](int tile) { tf = tile_of(min(tile, t1 - 1)); };  // (tiles past the end re-fetch the last one: L2 hits, never used)
    auto fetch1 = [&](int it) {
        const int j = (int)(threadIdx.x >> 4) + 16 * it;
        const bool inr = j < kFlatRows;
        int sq, pos, g;
        slab_row(tf, min(j, kFlatRows - 1), sq, pos, g);
        sraw[it] = ld4_off(src, map.off32(min(sq, S - 1), min(pos, map.npos - 1)) + (threadIdx.x & 15) * 16u);
        sinfo[it] = (unsigned)((inr ? j * kSlabLd + g * kSegSkew : kFlatRows * kSlabLd + 2 * kSegSkew) + cst);
    };
    // LayerNormalization4D over the 64 channels of a position (normalizations.py:33-37), one 16-row group of the slab at a time, in three
    // pieces that fit between two MFMAs.  Slab rows that no valid output row reads (the unused halo slots of a tile, rows past the last
    // sequence) get the LayerNorm of whatever clamped row was fetched for them - finite values that only meet dropped output rows.
    // (v_rsq_f32 for 1 / sqrt: hipcc expands the IEEE form into ~35 instructions with branches, and every VALU instruction of this kernel is
    // paid in matrix-pipe time; <= 1 ulp of rstd against the LDS-staged kernels, which keep the IEEE form)
    float4 ln_d;
    float ln_s;
    auto stage_a = [&](int it) {
        const float4 v = sraw[it];
        const float mean = row16_sum(v.x + v.y + v.z + v.w) * (1.f / 64.f);
        ln_d = f4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
        ln_s = ln_d.x * ln_d.x + ln_d.y * ln_d.y + ln_d.z * ln_d.z + ln_d.w * ln_d.w;
    };
    auto stage_b = [&]() { ln_s = __builtin_amdgcn_rsqf(row16_sum(ln_s) * (1.f / 64.f) + kEps); };
    // fp32: the 4 channels as they are.  bf16 modes: a row is 4 groups of 16 channels, each stored as [hi plane of the kh = 0 fragment half
    // (channels 0-3, 8-11) | hi, kh = 1 (4-7, 12-15) | lo, kh = 0 | lo, kh = 1], 16 bytes each - a lane's ds_read_b128 is a whole MFMA operand tuple
    // (the same 8 k values, in the same order, as the weight tuples above: no register shuffling in the K loop); this thread's channel quad is
    // 8 bytes of a hi slot and 8 bytes of the lo slot 32 bytes further on
    auto stage_c = [&](float* sl, int it) {
        const float4 y = fma4(ln_d * ln_s, g4, b4);
        if constexpr (NT == 0) {
            st4(sl + sinfo[it], y);
        } else {
            const float4 pk = pack4<NT>(y);
            *reinterpret_cast<float2*>(sl + sinfo[it]) = make_float2(pk.x, pk.y);
            if constexpr (NT == 3) *reinterpret_cast<float2*>(sl + sinfo[it] + 8) = make_float2(pk.z, pk.w);
        }
    };
    // U0 rows through a buffer descriptor: rows past the end (and the "previous tile" of the first one) are dropped by the range check
    const long long R = (long long)S * L;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(R * 1024), 0x00020000);
    const unsigned ocol = (unsigned)(i * 256 + 128 * half + 32 * w + 4 * kh) * 4u;  // this lane's row i, first column quad
    fetch_begin(t0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) fetch1(it);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        stage_a(it);
        stage_b();
        stage_c(slab[0], it);
    }
    fetch_begin(t0 + 1);
#pragma unroll
    for (int it = 0; it < NIT; ++it) fetch1(it);
    // 44 of the 64 weight fragments live in the accumulation registers and are read by the MFMAs from there (bound to that class here - otherwise
    // hipcc uses those registers as spill slots and copies every value back to a VGPR before its MFMA - and only here: the binding waits for the
    // weight loads, which so far were in flight together with the first tile's rows)
    if constexpr (NT == 0) {
#pragma unroll
        for (int q = 0; q < 44; ++q) asm volatile("" : "+a"(wf[q].x), "+a"(wf[q].y), "+a"(wf[q].z), "+a"(wf[q].w));
    } else {
#pragma unroll
        for (int q2 = 0; q2 < 32; ++q2) asm volatile("" : "+a"(whi[q2]));
        if constexpr (NT == 3) {
#pragma unroll
            for (int q2 = 0; q2 < 12; ++q2) asm volatile("" : "+a"(wlo[q2]));
        }
    }
    __syncthreads();
    unsigned prev_base = 0xC0000000u;  // no previous tile yet: every store of its write-back is dropped (the launcher keeps U0 below 2^31 bytes)
    // slab offset (floats) of this lane's output row in the two row tiles of a tile: row ri + 7 g, skewed by its segment g
    auto rows_in = [&](const FlatTile& t, const float* sl, const float* (&bp)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int ri = 32 * m + i, g = (ri >= t.n0) + (ri >= t.n0 + t.n1);
            bp[m] = sl + (ri + 7 * g) * kSlabLd + g * kSegSkew + 4 * kh;
        }
    };
    // The tile loop is unrolled by two: even tiles live in slab[0] and accumulator set A, odd tiles in slab[1] and set B.  The write-back of a
    // tile reads its accumulators in place while the next tile fills the other set, the first fragments of the next tile are read during the
    // last step of the current one, and the two barriers a tile needs sit between MFMAs in the middle of it - the MFMA stream never stops at a
    // tile boundary.  Per tile, one small piece per step, BETWEEN two MFMAs of the step (an LDS / buffer instruction then issues in the shadow of
    // a running MFMA; bunched in front of a step they cost 2-3x their VALU time):
    //   step  1      barrier: every wave has left the previous tile, its slab may be overwritten;
    //   steps 2-19   the NEXT tile's raw rows (requested one tile ago) -> LayerNormalization4D -> that slab;
    //   steps 20-27  the PREVIOUS tile's accumulators -> U0 (8 stores);   steps 28-34  the global loads of the tile after next;
    //   step  40     barrier: the next tile's slab is complete;   step 58: the next tile's row geometry;   step 63: its first fragments.
    // (bf16 / split-bf16: 32 steps of 16 k, two of these slots per step)
    // (Order of the memory operations: the loads of the previous tile are consumed before this tile's stores, this tile's loads come last.)
    floatx16 accA[2], accB[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[m][r] = 0.f, accB[m][r] = 0.f;
    const float* bp[2];
    float4 eb[2][2];      // fp32: [buffer][row tile]
    float4 ebp[2][2][2];  // bf16: [buffer][row tile][slot]
    rows_in(tile_of(t0), slab[0], bp);
    if constexpr (NT == 0) {
        eb[0][0] = ld4(bp[0]), eb[0][1] = ld4(bp[1]);
    } else {
#pragma unroll
        for (int m = 0; m < 2; ++m) ebp[0][m][0] = ld4(bp[m]), ebp[0][m][1] = ld4(bp[m] + 8);  // (hi, lo tuples of step 0)
    }
    auto out1 = [&](const floatx16 (&h)[2], int it, unsigned base) {
        const float4 v = acc_group(h[it >> 2], it & 3);
        __builtin_amdgcn_raw_buffer_store_b128(uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, ru,
                                               (int)(ocol + (base + (unsigned)((it >> 2) * 32 * 1024 + (it & 3) * 32))), 0, 0);
    };
    auto body = [&](auto par, int tile, floatx16 (&acc)[2], const floatx16 (&accp)[2]) {
        constexpr int PAR = decltype(par)::value;
        float* sn = slab[PAR ^ 1];
        const float* bpn[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        // the tile's other work as 64 slots: one per step of the fp32 K loop, two per 16-k step of the bf16 loops
        auto piece = [&](int sl_) {
            if (sl_ == 1) __syncthreads();
            if (sl_ >= 2 && sl_ < 20) {
                if ((sl_ - 2) % 3 == 0) stage_a((sl_ - 2) / 3);
                if ((sl_ - 2) % 3 == 1) stage_b();
                if ((sl_ - 2) % 3 == 2) stage_c(sn, (sl_ - 2) / 3);
            }
            if (sl_ >= 20 && sl_ < 28) out1(accp, sl_ - 20, prev_base);
            if (sl_ == 28) fetch_begin(tile + 2);
            if (sl_ >= 29 && sl_ < 29 + NIT) fetch1(sl_ - 29);
            if (sl_ == 40) __syncthreads();
            if (sl_ == 58) rows_in(tile_of(min(tile + 1, t1 - 1)), sn, bpn);
        };
        // (two half loops: one 64-step body exceeds hipcc's full-unroll budget, and a partially unrolled loop indexes the weight registers
        // dynamically, i.e. puts them in scratch)
        auto half_loop = [&](auto qh) {
            if constexpr (NT == 0) {
#pragma unroll
                for (int qq = 0; qq < 32; ++qq) {
                    const int q = decltype(qh)::value * 32 + qq;
                    if (q + 1 < 64) {
                        const int o = ((q + 1) >> 3) * kSlabLd + ((q + 1) & 7) * 8;  // tap (q + 1) / 8 = slab row offset, channel 8 ((q + 1) % 8)
                        eb[(q + 1) & 1][0] = ld4(bp[0] + o), eb[(q + 1) & 1][1] = ld4(bp[1] + o);
                    } else {
                        eb[0][0] = ld4(bpn[0]), eb[0][1] = ld4(bpn[1]);  // step 0 of the next tile
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float4 e0 = eb[q & 1][0], e1 = eb[q & 1][1];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].x, e0.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].x, e1.x, acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    piece(q);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].y, e0.y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].y, e1.y, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].z, e0.z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].z, e1.z, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].w, e0.w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].w, e1.w, acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // 16 k per step; the operand tuples of step q2 + 1 are read before the MFMAs of step q2
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) {
                    const int q2 = decltype(qh)::value * 16 + qq;
                    float4(&nx)[2][2] = ebp[(q2 + 1) & 1];
                    const float* b0 = q2 + 1 < 32 ? bp[0] + ((q2 + 1) >> 2) * kSlabLd + ((q2 + 1) & 3) * 16 : bpn[0];  // tap (q2 + 1) / 4, channel group (q2 + 1) % 4
                    const float* b1 = q2 + 1 < 32 ? bp[1] + ((q2 + 1) >> 2) * kSlabLd + ((q2 + 1) & 3) * 16 : bpn[1];  // (step 0 of the next tile after the last step)
                    nx[0][0] = ld4(b0), nx[1][0] = ld4(b1);
                    if constexpr (NT == 3) nx[0][1] = ld4(b0 + 8), nx[1][1] = ld4(b1 + 8);
                    __builtin_amdgcn_sched_barrier(0);
                    auto tuple = [](float4 v) { return __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}); };
                    const Frag wq{whi[q2], wlo[q2], wlo[q2]};
                    const Frag f0{tuple(ebp[q2 & 1][0][0]), tuple(ebp[q2 & 1][0][1]), tuple(ebp[q2 & 1][0][1])};
                    const Frag f1{tuple(ebp[q2 & 1][1][0]), tuple(ebp[q2 & 1][1][1]), tuple(ebp[q2 & 1][1][1])};
                    // the two accumulator chains alternate (a dependent 8-pass MFMA issued back to back waits for its predecessor); per accumulator
                    // the order of mma32<NT>: lo.hi, hi.lo, hi.hi
                    if constexpr (NT == 3) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq.lo, f0.hi, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq.lo, f1.hi, acc[1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        piece(2 * q2);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq.hi, f0.lo, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq.hi, f1.lo, acc[1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        piece(2 * q2 + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq.hi, f0.hi, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq.hi, f1.hi, acc[1], 0, 0, 0);
                    } else {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq.hi, f0.hi, acc[0], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        piece(2 * q2);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq.hi, f1.hi, acc[1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        piece(2 * q2 + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        half_loop(std::integral_constant<int, 0>{});
        half_loop(std::integral_constant<int, 1>{});
        bp[0] = bpn[0], bp[1] = bpn[1];
        prev_base = (unsigned)(tile * 64) * 1024u;
    };
    bool last_is_b = false;
#pragma unroll 1
    for (int tile = t0; tile < t1; tile += 2) {
        body(std::integral_constant<int, 0>{}, tile, accA, accB);
        last_is_b = tile + 1 < t1;
        if (!last_is_b) break;
        body(std::integral_constant<int, 1>{}, tile + 1, accB, accA);
    }
    if (last_is_b) {
#pragma unroll
        for (int it = 0; it < 8; ++it) out1(accB, it, prev_base);
    } else {
#pragma unroll
        for (int it = 0; it < 8; ++it) out1(accA, it, prev_base);
    }
#ifdef UW_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(dst) + 2 * blockIdx.x;
        o[0] = __builtin_amdgcn_s_memtime() - uw_t0;
        o[1] = (unsigned long long)(t1 - t0);
    }
#endif
}

// Layer-0 kernel for the six-term split (NT = 6: fp32-equivalent arithmetic on the bf16 pipe, common.h), large batches: WEIGHT-STATIONARY.
// The LDS-staged kernel splits every fp32 fragment it reads into three bf16 planes in registers (~56 VALU instructions per ds_read_b128) and is
// bound by that; here every operand is split ONCE.  Four workgroups of one XCD share a range of 64-row tiles (same flattened tile geometry as
// unfold_ws_kernel), 64 output columns each.  Wave (cn = w & 1, kp = w >> 1) owns columns 32 cn .. + 31 and the taps 4 kp .. 4 kp + 3: 32 x 256
// weights split three ways = 16 steps x (hi, mid, lo) operand tuples of v_mfma_f32_32x32x16_bf16 = 192 registers per lane, bound to the
// accumulation half of the register file.  The slab holds the LayerNorm-ed rows as THREE bf16 planes per row (hi | mid | lo, 128 bytes each,
// row stride 25 sixteen-byte slots, segment skew 1 slot: the 7-row jump at a sequence boundary becomes 176 slots = 0 mod 16), written by the
// staging threads - a lane's ds_read_b128 is a finished operand tuple.  Per tile and wave: 2 row tiles x 16 steps x 6 products = 192 MFMAs of
// 32 cycles (the fp32 fast-FIR kernel: 768 of 32) and 96 fragment reads; the two tap halves of a column block are added through LDS in the
// write-back (each wave of a pair hands one row tile over and stores the other).  Product order per accumulator as mma32<6>.
#ifndef WS6_ABL
#define WS6_ABL 0  // ablation builds (tools/ffa_ablate.sh ws6; wrong results, timing only): 1 no staging, 2 no write-back, 4 no fetch, 8 no barriers, 16 no fragment reads
#endif
constexpr int kW6Row = 100;  // floats per slab row
constexpr int kW6Skew = 4;   // floats per segment index
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void unfold_ws6_kernel(SeqMap map, const float* __restrict__ src,
                                                                                                        const float* __restrict__ gamma,
                                                                                                        const float* __restrict__ beta,
                                                                                                        const float* __restrict__ Wt, float* __restrict__ dst,
                                                                                                        int S, int total_tiles) {
    constexpr int NIT = (kFlatRows * 16 + 255) / 256;  // 6
    __shared__ __attribute__((aligned(16))) float slab[2][(kFlatRows + 1) * kW6Row + 2 * kW6Skew];  // (+ one scratch row)
    __shared__ __attribute__((aligned(16))) float red[4][4 * 64 * 4];  // per wave: the row tile it hands to its partner, [register group][lane]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int cn = w & 1, kp = w >> 1;
    const int xcd = blockIdx.x & 7, qd = (blockIdx.x >> 3) & 3, slot = xcd + 8 * (blockIdx.x >> 5), nslots = gridDim.x >> 2;
    const int c4 = (threadIdx.x & 15) * 4;
    const float4 g4 = ld4(gamma + c4), b4 = ld4(beta + c4);
    const int L = map.L;
#ifdef UW_TIMING
    const unsigned long long uw_t0 = __builtin_amdgcn_s_memtime();
#endif

    // step s = tap 4 kp + s / 4, channels 16 (s % 4) + 8 kh .. + 7 of column 64 qd + 32 cn + i
    bf16x8 whi[16], wmid[16], wlo[16];
    {
        const float* wp = Wt + (size_t)(64 * qd + 32 * cn + i) * 512 + 256 * kp + 8 * kh;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const Frag f = frag_split3(ld4(wp + 16 * s), ld4(wp + 16 * s + 4));
            whi[s] = f.hi, wmid[s] = f.mid, wlo[s] = f.lo;
        }
    }

    const int t0 = (int)((long long)total_tiles * slot / nslots), t1 = (int)((long long)total_tiles * (slot + 1) / nslots);
    if (t0 >= t1) return;
    auto tile_of = [&](int gt) {
        FlatTile t;
        t.r0 = gt * 64;
        t.s0 = (int)__umulhi((unsigned)t.r0, map.magicL);  // = r0 / L (exact: S L^2 < 2^32, checked by the launcher)
        t.l0 = t.r0 - t.s0 * L;
        t.n0 = min(L - t.l0, 64);
        t.n1 = min(L, 64 - t.n0);
        return t;
    };
    auto slab_row = [&](const FlatTile& t, int j, int& sq, int& pos, int& g) {  // slab row j of a tile -> (sequence, position, segment)
        const int e0 = t.n0 + 7, e1 = e0 + t.n1 + 7;
        g = (j >= e0) + (j >= e1);
        const int jj = j - (g == 0 ? 0 : (g == 1 ? e0 : e1));
        sq = t.s0 + g;
        pos = (g == 0 ? t.l0 : 0) + jj;
    };
    float4 sraw[NIT];
    unsigned sinfo[NIT];  // float offset of this thread's 8 hi bytes in the slab (scratch row behind the slab for rows past it)
    FlatTile tf;
    auto fetch_begin = [&](int tile) { tf = tile_of(min(tile, t1 - 1)); };  // (tiles past the end re-fetch the last one: L2 hits, never used)
    auto fetch1 = [&](int it) {
        const int j = (int)(threadIdx.x >> 4) + 16 * it;
        const bool inr = j < kFlatRows;
        int sq, pos, g;
        slab_row(tf, min(j, kFlatRows - 1), sq, pos, g);
        sraw[it] = ld4_off(src, map.off32(min(sq, S - 1), min(pos, map.npos - 1)) + (threadIdx.x & 15) * 16u);
        sinfo[it] = (unsigned)((inr ? j * kW6Row + g * kW6Skew : kFlatRows * kW6Row + 2 * kW6Skew) + (c4 >> 1));
    };
    // LayerNormalization4D over the 64 channels of a position (normalizations.py:33-37) and the three-way split (residues exact in fp32), in four
    // pieces of <= ~15 instructions.  Each piece ends by passing its results through an empty volatile asm: sched_barrier only binds the machine
    // scheduler, the instruction selector is free to sink pure arithmetic to its first use - it merged three pieces into one MFMA gap otherwise.
    float4 ln_d, sp_r;
    float ln_s;
    auto pin4 = [](float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
    auto stage_a = [&](int it) {
        const float4 v = sraw[it];
        const float mean = row16_sum(((v.x + v.y) + v.z) + v.w) * (1.f / 64.f);
        ln_d = f4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
        pin4(ln_d);
    };
    auto stage_b = [&]() {
        ln_s = __builtin_amdgcn_rsqf(row16_sum(fmaf(ln_d.w, ln_d.w, fmaf(ln_d.z, ln_d.z, fmaf(ln_d.y, ln_d.y, ln_d.x * ln_d.x)))) * (1.f / 64.f) + kEps);
        asm volatile("" : "+v"(ln_s));
    };
    auto stage_c = [&](float* sl, int it) {
        const float4 y = f4(fmaf(ln_d.x * ln_s, g4.x, b4.x), fmaf(ln_d.y * ln_s, g4.y, b4.y), fmaf(ln_d.z * ln_s, g4.z, b4.z), fmaf(ln_d.w * ln_s, g4.w, b4.w));
        const unsigned h0 = pk_bf16(y.x, y.y), h1 = pk_bf16(y.z, y.w);
        sp_r = f4(y.x - __uint_as_float(h0 << 16), y.y - __uint_as_float(h0 & 0xffff0000u), y.z - __uint_as_float(h1 << 16),
                  y.w - __uint_as_float(h1 & 0xffff0000u));
        *reinterpret_cast<float2*>(sl + sinfo[it]) = make_float2(__uint_as_float(h0), __uint_as_float(h1));
        pin4(sp_r);
    };
    auto stage_d = [&](float* sl, int it) {
        const float4 r = sp_r;
        const unsigned m0 = pk_bf16(r.x, r.y), m1 = pk_bf16(r.z, r.w);
        const unsigned l0 = pk_bf16(r.x - __uint_as_float(m0 << 16), r.y - __uint_as_float(m0 & 0xffff0000u));
        const unsigned l1 = pk_bf16(r.z - __uint_as_float(m1 << 16), r.w - __uint_as_float(m1 & 0xffff0000u));
        float* o = sl + sinfo[it];
        *reinterpret_cast<float2*>(o + 32) = make_float2(__uint_as_float(m0), __uint_as_float(m1));
        *reinterpret_cast<float2*>(o + 64) = make_float2(__uint_as_float(l0), __uint_as_float(l1));
    };
    const long long R = (long long)S * L;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(R * 1024), 0x00020000);
    // a wave's two row tiles: local 0 = rows 32 kp + i (kept and stored), local 1 = rows 32 (kp ^ 1) + i (handed to the partner wave w ^ 2)
    const unsigned ocol = (unsigned)((32 * kp + i) * 256 + 64 * qd + 32 * cn + 4 * kh) * 4u;
    fetch_begin(t0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) fetch1(it);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        stage_a(it);
        stage_b();
        stage_c(slab[0], it);
        stage_d(slab[0], it);
    }
    fetch_begin(t0 + 1);
#pragma unroll
    for (int it = 0; it < NIT; ++it) fetch1(it);
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" : "+a"(whi[s]), "+a"(wmid[s]), "+a"(wlo[s]));  // (see unfold_ws_kernel)
    __syncthreads();
    unsigned prev_base = 0xC0000000u;  // no previous tile yet: its stores are dropped by the range check (U0 stays below 2^31 bytes)
    // slab offset of this lane's output row in its two row tiles, at this wave's first tap
    auto rows_in = [&](const FlatTile& t, const float* sl, const float* (&bp)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int ri = 32 * (m ^ kp) + i, g = (ri >= t.n0) + (ri >= t.n0 + t.n1);
            bp[m] = sl + (ri + 7 * g + 4 * kp) * kW6Row + g * kW6Skew + 4 * kh;
        }
    };
    // Tile loop unrolled by two (slab / accumulator set A, B) with the other work of a tile in 64 slots between the MFMAs, four per step:
    //   slots 0-1    the PREVIOUS tile's hand-over row tile -> red;   slot 2: barrier (red complete, every wave has left the previous slab);
    //   slots 3-26   the NEXT tile's raw rows -> LayerNormalization4D -> three planes -> that slab (four pieces per 16-row group);
    //   slots 27-31  partner's partial sums (read one slot ahead) + own -> U0 (4 stores);   slots 32-38: the global loads of the tile after next;
    //   slot  44     barrier: the next slab is complete, red may be overwritten;   slot 58: the next tile's row geometry.
    floatx16 accA[2], accB[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[m][r] = 0.f, accB[m][r] = 0.f;
    const float* bp[2];
    float4 eb[2][2][3];  // [buffer][row tile][plane]
    rows_in(tile_of(t0), slab[0], bp);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) eb[0][m][pl] = ld4(bp[m] + 32 * pl);
    float* rmine = red[w] + lane * 4;
    const float* rpart = red[w ^ 2] + lane * 4;
    auto hand1 = [&](const floatx16 (&h)[2], int g) { st4(rmine + g * 256, acc_group(h[1], g)); };
    float4 rv;  // the partner's register group, read one slot before it is used
    auto out0 = [&](int g) { rv = ld4(rpart + g * 256); };
    auto out1 = [&](const floatx16 (&h)[2], int g, unsigned base) {
        const float4 v = acc_group(h[0], g) + rv;
        __builtin_amdgcn_raw_buffer_store_b128(uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, ru,
                                               (int)(ocol + (base + (unsigned)(g * 32))), 0, 0);
    };
    auto tuple = [](float4 v) { return __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}); };
    auto body = [&](auto par, int tile, floatx16 (&acc)[2], const floatx16 (&accp)[2]) {
        constexpr int PAR = decltype(par)::value;
        float* sn = slab[PAR ^ 1];
        const float* bpn[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        auto piece = [&](int sl_) {
            if (sl_ == 0 && !(WS6_ABL & 2)) hand1(accp, 0), hand1(accp, 1);
            if (sl_ == 1 && !(WS6_ABL & 2)) hand1(accp, 2), hand1(accp, 3);
            if (sl_ == 2 && !(WS6_ABL & 8)) __syncthreads();
            if (sl_ >= 3 && sl_ < 27 && !(WS6_ABL & 1)) {
                if ((sl_ - 3) % 4 == 0) stage_a((sl_ - 3) / 4);
                if ((sl_ - 3) % 4 == 1) stage_b();
                if ((sl_ - 3) % 4 == 2) stage_c(sn, (sl_ - 3) / 4);
                if ((sl_ - 3) % 4 == 3) stage_d(sn, (sl_ - 3) / 4);
            }
            if (sl_ >= 28 && sl_ < 32 && !(WS6_ABL & 2)) out1(accp, sl_ - 28, prev_base);
            if (sl_ >= 27 && sl_ < 31 && !(WS6_ABL & 2)) out0(sl_ - 27);
            if (sl_ == 32) fetch_begin(tile + 2);
            if (sl_ >= 33 && sl_ < 33 + NIT && !(WS6_ABL & 4)) fetch1(sl_ - 33);
            if (sl_ == 44 && !(WS6_ABL & 8)) __syncthreads();
            if (sl_ == 58) rows_in(tile_of(min(tile + 1, t1 - 1)), sn, bpn);
        };
        // One step = 12 MFMAs of 32 cycles in four regions of three, each region with one slot of the other work and two of the six fragment reads
        // of the next step.  A wave issues in order: whatever follows an MFMA hides only under THAT MFMA (~6 instructions), so inside a region the
        // scheduler is asked (sched_group_barrier) for the pattern MFMA, <= 6 others, MFMA, <= 6 others, MFMA, rest - a block of 15-20 instructions
        // behind three back-to-back MFMAs cost two thirds of its own issue time (s_memtime build: 9.5k cycles per tile against 6.1k of MFMAs).
#define WS6_REGION(SLOT)                                         \
    piece(SLOT);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
    __builtin_amdgcn_sched_group_barrier(0x496, 6, 0);           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
    __builtin_amdgcn_sched_group_barrier(0x496, 6, 0);           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
    __builtin_amdgcn_sched_group_barrier(0x496, 8, 0);           \
    __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float4(&nx)[2][3] = eb[(s + 1) & 1];
            const float* bn[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) bn[m] = s + 1 < 16 ? bp[m] + ((s + 1) >> 2) * kW6Row + ((s + 1) & 3) * 8 : bpn[m];  // (step 0 of the next tile after the last step)
            const bool rd = !(WS6_ABL & 16) || s == 15;
            const bf16x8 h0 = tuple(eb[s & 1][0][0]), m0 = tuple(eb[s & 1][0][1]), l0 = tuple(eb[s & 1][0][2]);
            const bf16x8 h1 = tuple(eb[s & 1][1][0]), m1 = tuple(eb[s & 1][1][1]), l1 = tuple(eb[s & 1][1][2]);
            // the two accumulator chains alternate; per accumulator the order of mma32<6>: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
            if (rd) nx[0][0] = ld4(bn[0]), nx[1][0] = ld4(bn[1]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[s], h0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[s], h1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], l0, acc[0], 0, 0, 0);
            WS6_REGION(4 * s);
            if (rd) nx[0][2] = ld4(bn[0] + 64), nx[1][2] = ld4(bn[1] + 64);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], l1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wmid[s], m0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wmid[s], m1, acc[1], 0, 0, 0);
            WS6_REGION(4 * s + 1);
            if (rd) nx[0][1] = ld4(bn[0] + 32), nx[1][1] = ld4(bn[1] + 32);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wmid[s], h0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wmid[s], h1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], m0, acc[0], 0, 0, 0);
            WS6_REGION(4 * s + 2);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], m1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], h0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[s], h1, acc[1], 0, 0, 0);
            WS6_REGION(4 * s + 3);
        }
#undef WS6_REGION
        bp[0] = bpn[0], bp[1] = bpn[1];
        prev_base = tile < t1 ? (unsigned)(tile * 64) * 1024u : 0xC0000000u;
    };
    // Always whole pairs, without a branch between the two bodies: a range with an odd number of tiles runs its last tile once more as a dummy whose
    // stores are dropped (<= 1 tile in ~56).  With `if (...) break` in between, the bodies are two basic blocks and hipcc sinks everything the
    // first one prepares for the second - the global loads of the tile after next, the row geometry, ~200 instructions - to the top of the
    // second block, in front of its first MFMA.
#pragma unroll 1
    for (int tile = t0; tile < t1; tile += 2) {
        body(std::integral_constant<int, 0>{}, tile, accA, accB);
        body(std::integral_constant<int, 1>{}, tile + 1, accB, accA);
    }
    // the last tile: hand over, barrier, add and store
#pragma unroll
    for (int g = 0; g < 4; ++g) hand1(accB, g);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) out0(g), out1(accB, g, prev_base);
#ifdef UW_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(dst) + 2 * blockIdx.x;
        o[0] = __builtin_amdgcn_s_memtime() - uw_t0;
        o[1] = (unsigned long long)(t1 - t0);
    }
#endif
}

// Layer-0 kernel, fifth generation (round 5), large batches, fp32: WEIGHT-STATIONARY 2-PARALLEL FAST FIR - 0.75x the MFMAs of the direct form.
// U[l] = sum_{k<8} W_k x[l + k] is a stride-1 correlation over the sequence with 64 x 256 matrix taps.  With the input phases e[m] = x[2m],
// o[m] = x[2m + 1] and the even / odd taps We[j] = W_{2j}, Wo[j] = W_{2j+1} (j < 4):
//      A[m] = sum_j We[j] e[m + j],   B[m] = sum_j Wo[j] o[m + j],   Z[m] = sum_j (We[j] + Wo[j]) (o[m + j] + e[m + j + 1])
//      U[2m] = A[m] + B[m],           U[2m + 1] = Z[m] - A[m + 1] - B[m]
// - three 4-tap correlations over half-rate sequences per PAIR of outputs (768 k-products) instead of two 8-tap ones (1024).  Rows of the GEMM
// are "virtual rows" (sequence s, pair index v), Lv = (L + 2) / 2 per sequence (the last one of an even-length sequence only supplies A[m + 1]),
// flattened over the sequences and cut into 64-row tiles that advance by 63 rows: row 63 of a tile supplies A[m + 1] to row 62 and is computed
// again as row 0 of the next tile (1.6 % of the MFMAs; with the per-sequence extra row 0.775x the direct form's count in total).
// Three weight sets = 768 KB: FOUR workgroups (one XCD) share a range of tiles, 64 output columns each; wave w owns 16 columns x all 768 k =
// 192 registers per lane, bound to the accumulation half of the register file, and runs v_mfma_f32_16x16x4_f32 (A = weights: lane (n, kg) holds
// column n, k = 4 kg + step; B = 16 virtual rows; a lane ends up with 4 consecutive output channels of one row).  LDS holds, double-buffered, three
// planes of LayerNorm-ed rows per tile - E (even positions), O (odd), Z' = O + next E - as "unit" rows u = (positions 2(v + u) - 1, 2(v + u)) with 4
// halo units per sequence segment (a tile spans up to four sequences); segment g is skewed by 12 g sixteen-byte slots so that the +4-row jump
// at a sequence boundary keeps a wave's ds_read_b128 conflict-free.  As in unfold_ws_kernel the K loop (768 MFMAs per wave and tile) never stops:
// staging of the next tile, write-back of the previous one (two accumulator sets; A[m + 1] arrives through two DPP row shifts) and the global
// loads of the tile after next ride between its MFMAs, one piece per 16-MFMA group.  Not bit-compatible with the direct form: (We + Wo) is rounded
// once and Z - A - B cancels a little (fp32 error of U against float64 2.2e-7 against 1.6e-7, tools/ffa_prototype.py).
// Reference: rnn_layers.py:97,146-150 (LayerNormalization4D + nn.Unfold((8, 1)) + the SRU's layer-0 projection).
constexpr int kFfaUnits = 80;                                        // 64 rows + 4 segments x 4 halo units
constexpr int kFfaSkew = 48;                                         // floats (12 slots) per segment index: 4 rows x 17 slots + 12 = 80 = 0 (mod 16)
constexpr int kFfaPlane = kFfaUnits * kSlabLd + 3 * kFfaSkew + 16;   // floats per plane
struct FfaTile {  // geometry of one 64-row tile (wave-uniform): first virtual row r0 = (s0, v0); n0 rows in segment 0
    int s0, v0, n0;
};
// MODE 1 (round 5): the same machinery for the ConvTranspose1d that closes a DualPathRNN (rnn_layers.py:129,153-156): y[n] = sum_k' W'[k'] x[n + k'] over the
// zero-padded SRU output x[p] = h3[p - 7] is the same 8-tap correlation with 64-channel taps and 64 output channels - one workgroup holds all three weight
// sets (192 KB), there is no LayerNorm in the staging (rows outside the sequence come back as zeros from the buffer range check), and the write-back adds bias
// and the residual row of G (fetched earlier in the same tile) and stores in place.
// MODE 2: the input gradient of that ConvTranspose1d (training step; adjoint of rtfs_dp_convt_fwd): dH3[l] = sum_k W[k] dG[l + k] - the unfold correlation on
// the rows of dG without the LayerNorm, 64 output channels, plain stores into [S][L][64].
template <int DIM, int MODE = 0>  // DIM 4: sequences along F (one per (b, t2)), 3: along T (one per (b, f2)) - the row address is shifts and one 24-bit multiply
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void unfold_ffa_kernel(SeqMap map, const float* __restrict__ src,
                                                                                                       const float* __restrict__ gamma,
                                                                                                       const float* __restrict__ beta,
                                                                                                       const float* __restrict__ Wt, float* __restrict__ dst,
                                                                                                       int S, int Lv, unsigned magicLv, int total_tiles) {
    constexpr int NIT = kFfaUnits * 16 / 256;  // 5 staging iterations of (unit row, channel quad) items
    __shared__ __attribute__((aligned(16))) float slab[2][3][kFfaPlane];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, kg = lane >> 4;
    // workgroups b, b + 8, b + 16, b + 24 (one XCD under round-robin dispatch) hold the four 64-column quarters of one tile range
    const int xcd = blockIdx.x & 7, qd = MODE == 0 ? (blockIdx.x >> 3) & 3 : 0, slot = MODE == 0 ? xcd + 8 * (blockIdx.x >> 5) : blockIdx.x,
              nslots = MODE == 0 ? gridDim.x >> 2 : gridDim.x;
    const int col0 = 64 * qd + 16 * w;
    const int c4 = (threadIdx.x & 15) * 4;
    const float4 g4 = MODE == 0 ? ld4(gamma + c4) : f4(0, 0, 0, 0), b4 = MODE == 0 ? ld4(beta + c4) : f4(0, 0, 0, 0);
    const floatx4 bias4 = MODE == 1 ? *reinterpret_cast<const floatx4*>(gamma + col0 + 4 * kg) : floatx4{0.f, 0.f, 0.f, 0.f};  // (MODE 1: `gamma` is the bias)
    const int L = MODE == 1 ? map.npos : map.L;  // outputs per sequence
    const int Lh = map.L;                        // MODE 1: rows of h3 per sequence

    // weights: lane (n = j, kg) holds W[col0 + n][64 tap + 16 cg + 4 kg .. + 3] for q = 4 j' + cg; We: tap 2 j', Wo: tap 2 j' + 1
    float4 wA[16], wB[16], wS[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float* wp = Wt + (size_t)(col0 + j) * 512 + 128 * (q >> 2) + 16 * (q & 3) + 4 * kg;
        wA[q] = ld4(wp), wB[q] = ld4(wp + 64);
    }
    const int t0 = (int)((long long)total_tiles * slot / nslots), t1 = (int)((long long)total_tiles * (slot + 1) / nslots);
    if (t0 >= t1) return;
    auto tile_of = [&](int t) {
        FfaTile g;
        const unsigned r0 = 63u * (unsigned)t;
        g.s0 = (int)__umulhi(r0, magicLv);  // = r0 / Lv (exact: checked by the launcher)
        g.v0 = (int)r0 - g.s0 * Lv;
        g.n0 = min(Lv - g.v0, 64);
        return g;
    };
    // ---- staging: unit row ur = (t >> 4) + 16 it of the tile being fetched: positions (2 (v + u) - 1, 2 (v + u)) of its segment's sequence ----
    float4 rawo[NIT], rawe[NIT];
    int sinfo[NIT];  // float offset of this thread's quad inside a plane of the fetched tile's slab: unit row * ld + segment skew + channel
    FfaTile tf;
    const unsigned lane16 = (threadIdx.x & 15) * 16u;
    const unsigned t2rows = (unsigned)(map.stride_hi >> 12);  // dim 3: T2 (stride_hi = T2 x F2 x 64 floats)
    constexpr unsigned kNowhere = 0xFFFFFF00u;
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, MODE == 1 ? (int)((long long)S * Lh * 256) : 0, 0x00020000);
    auto bld = [](const __amdgpu_buffer_rsrc_t& r, unsigned off) {
        const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return f4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    auto fetch1 = [&](int it) {
        const int ur = (int)(threadIdx.x >> 4) + 16 * it;
        const int b1 = tf.n0 + 4, b2 = b1 + Lv + 4, b3 = b2 + Lv + 4;
        const int g = (ur >= b1) + (ur >= b2) + (ur >= b3);
        const int u = ur - (g == 0 ? 0 : (g == 1 ? b1 : (g == 2 ? b2 : b3)));
        const int pe = 2 * ((g == 0 ? tf.v0 : 0) + u);
        if constexpr (MODE != 1) {
            const unsigned sq = (unsigned)min(tf.s0 + g, S - 1);
            const unsigned p1 = (unsigned)min(pe, map.npos - 1), p0 = (unsigned)min(max(pe - 1, 0), map.npos - 1);
            // byte offset of (sequence, position): dim 4: [s][pos][64]; dim 3: [s >> 6][pos][s & 63][64]
            const unsigned sb = DIM == 4 ? (sq << 14) + lane16 : (__umul24(sq >> 6, t2rows) << 14) + ((sq & 63u) << 8) + lane16;
            constexpr int PS = DIM == 4 ? 8 : 14;
            rawe[it] = ld4_off(src, sb + (p1 << PS));
            rawo[it] = ld4_off(src, sb + (p0 << PS));
        } else {  // x[p] = h3[s][p - 7] (contiguous [S][Lh][64]); zero outside the sequence / past the last one
            const int sq = tf.s0 + g, h1 = pe - 7, h0 = pe - 8;
            const unsigned sb = (__umul24((unsigned)min(sq, S - 1), (unsigned)Lh) << 8) + lane16;
            rawe[it] = bld(rh, (sq < S && (unsigned)h1 < (unsigned)Lh) ? sb + ((unsigned)h1 << 8) : kNowhere);
            rawo[it] = bld(rh, (sq < S && (unsigned)h0 < (unsigned)Lh) ? sb + ((unsigned)h0 << 8) : kNowhere);
        }
        sinfo[it] = ur * kSlabLd + g * kFfaSkew + c4;
    };
    // LayerNormalization4D over the 64 channels of a position (normalizations.py:33-37); 16 lanes = one position (v_rsq_f32 as in unfold_ws_kernel)
    float4 de, dod;
    float se, so;
    auto ln_mean = [&](int it) {  // the two rows of a unit side by side: their DPP chains interleave
        const float4 ve = rawe[it], vo = rawo[it];
        const float me = row16_sum(ve.x + ve.y + ve.z + ve.w) * (1.f / 64.f), mo = row16_sum(vo.x + vo.y + vo.z + vo.w) * (1.f / 64.f);
        de = f4(ve.x - me, ve.y - me, ve.z - me, ve.w - me);
        dod = f4(vo.x - mo, vo.y - mo, vo.z - mo, vo.w - mo);
    };
    auto ln_rstd = [&]() {
        se = __builtin_amdgcn_rsqf(row16_sum(de.x * de.x + de.y * de.y + de.z * de.z + de.w * de.w) * (1.f / 64.f) + kEps);
        so = __builtin_amdgcn_rsqf(row16_sum(dod.x * dod.x + dod.y * dod.y + dod.z * dod.z + dod.w * dod.w) * (1.f / 64.f) + kEps);
    };
#ifndef FFA_ABL
#define FFA_ABL 0  // ablation builds (tools/ffa_ablate.sh), wrong results, timing only: 1 no staging, 2 no write-back, 4 no fetch, 8 no barriers,
#endif             // 16 staging without its LDS stores, 32 staging without the LayerNorm arithmetic, 64 write-back without stores, 128 without its arithmetic
    auto raw_c = [&](float* sl, int off, int it) {  // MODE 1: the rows as they are
        st4(sl + off, rawe[it]);
        st4(sl + kFfaPlane + off, rawo[it]);
        st4(sl + 2 * kFfaPlane + off, rawe[it] + rawo[it]);
    };
    auto ln_c = [&](float* sl, int off) {
        const float4 ye = fma4(de * se, g4, b4), yo = fma4(dod * so, g4, b4);
        if (FFA_ABL & 16) {
            asm volatile("" ::"v"(ye.x + yo.x), "v"(ye.y + yo.y), "v"(ye.z + yo.z), "v"(ye.w + yo.w));
            return;
        }
        st4(sl + off, ye);
        st4(sl + kFfaPlane + off, yo);
        st4(sl + 2 * kFfaPlane + off, ye + yo);
    };
    // ---- output rows: virtual row r = 63 tile + 16 rt + j -> U0 rows (s L + 2 v) and (+ 1), through a buffer descriptor (invalid rows dropped) ----
    const long long R = (long long)S * L;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(dst, 0, MODE == 0 ? (int)(R * 1024) : (MODE == 2 ? (int)(R * 256) : (int)(((long long)(S >> map.seq_shift) * map.stride_hi) * 4)), 0x00020000);
    constexpr unsigned kDrop = 0xC0000000u;
    // byte offsets of the even / odd U0 rows of virtual rows 63 tile + 16 rt + j, rt = 0..3: one division for rt = 0, then + 16 rows with at most one
    // sequence wrap each (Lv >= 21)
    unsigned oe[4], oo[4];
    auto out_offsets = [&](int tile) {  // tile < 0 (no previous tile yet): everything dropped
        const unsigned r = 63u * (unsigned)max(tile, 0) + j;
        int sq = (int)__umulhi(r, magicLv), v = (int)r - (int)__umul24((unsigned)sq, (unsigned)Lv);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const bool ok = tile >= 0 && (16 * rt + j < 63) && sq < S && 2 * v < L;
            unsigned off, step;
            if constexpr (MODE == 0) {
                off = ((__umul24((unsigned)sq, (unsigned)L) + 2u * (unsigned)v) << 10) + (unsigned)(col0 + 4 * kg) * 4u, step = 1024u;
            } else if constexpr (MODE == 2) {
                off = ((__umul24((unsigned)sq, (unsigned)L) + 2u * (unsigned)v) << 8) + (unsigned)(col0 + 4 * kg) * 4u, step = 256u;
            } else {  // G row (sequence sq, position 2 v): dim 4: [s][pos][64]; dim 3: [s >> 6][pos][s & 63][64]
                constexpr int PS = DIM == 4 ? 8 : 14;
                const unsigned usq = (unsigned)sq;
                off = (DIM == 4 ? (usq << 14) : (__umul24(usq >> 6, t2rows) << 14) + ((usq & 63u) << 8)) + ((2u * (unsigned)v) << PS) + (unsigned)(col0 + 4 * kg) * 4u;
                step = 1u << PS;
            }
            oe[rt] = ok ? off : kDrop;
            oo[rt] = (ok && 2 * v + 1 < L) ? off + step : kDrop;
            v += 16;
            if (v >= Lv) v -= Lv, ++sq;
        }
    };
    // slab offset (floats, E plane) of this lane's row in the four 16-row tiles: unit row (16 rt + j) + 4 g, skewed by its segment g
    auto rows_in = [&](const FfaTile& t, const float* sl, const float* (&bp)[4]) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int ri = 16 * rt + j, g = (ri >= t.n0) + (ri >= t.n0 + Lv) + (ri >= t.n0 + 2 * Lv);
            bp[rt] = sl + (ri + 4 * g) * kSlabLd + g * kFfaSkew + 4 * kg;
        }
    };
    // B fragment of step q = 4 j' + cg of sub-GEMM sub (0 A: plane E, unit row + j'; 1 B: plane O, + j' + 1; 2 Z: plane Z', + j' + 1)
    auto frag_off = [](int sub, int q) { return sub * kFfaPlane + ((q >> 2) + (sub != 0)) * kSlabLd + (q & 3) * 16; };

    // ---- prologue: first tile staged directly, second tile's rows requested ----
    tf = tile_of(t0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) fetch1(it);
#pragma unroll
    for (int q = 0; q < 16; ++q) wS[q] = wA[q] + wB[q];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if constexpr (MODE == 0) {
            ln_mean(it);
            ln_rstd();
            ln_c(&slab[0][0][0], sinfo[it]);
        } else {
            raw_c(&slab[0][0][0], sinfo[it], it);
        }
    }
    tf = tile_of(min(t0 + 1, t1 - 1));
#pragma unroll
    for (int it = 0; it < NIT; ++it) fetch1(it);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        asm volatile("" : "+a"(wA[q].x), "+a"(wA[q].y), "+a"(wA[q].z), "+a"(wA[q].w));
        asm volatile("" : "+a"(wB[q].x), "+a"(wB[q].y), "+a"(wB[q].z), "+a"(wB[q].w));
        asm volatile("" : "+a"(wS[q].x), "+a"(wS[q].y), "+a"(wS[q].z), "+a"(wS[q].w));
    }
    __syncthreads();

    floatx4 accA[3][4], accB[3][4];  // [sub-GEMM][16-row tile], two sets: even / odd tiles
#pragma unroll
    for (int sb = 0; sb < 3; ++sb)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) accA[sb][rt] = floatx4{0.f, 0.f, 0.f, 0.f}, accB[sb][rt] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float* bp[4];
    float4 eb[2][4];  // [buffer][16-row tile]
    rows_in(tile_of(t0), &slab[0][0][0], bp);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) eb[0][rt] = ld4(bp[rt] + frag_off(0, 0));

    auto shl1 = [](float lo, float nxt) {  // lane j <- lane j + 1 of `lo` inside its 16-lane row; lane 15 <- lane 0 of `nxt`
        int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, lo), 0x101, 0xf, 0xf, false);   // row_shl:1
        t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, nxt), 0x11f, 0xf, 0xf, false);      // row_shr:15 (only lane 15 has a source)
        return __builtin_bit_cast(float, t);
    };
    floatx4 aps;  // A[m + 1] of the row tile being written back
    auto store_u = [&](floatx4 u, unsigned off) {
        if (FFA_ABL & 64) {
            asm volatile("" ::"v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(off));
            return;
        }
        __builtin_amdgcn_raw_buffer_store_b128(uint4v{__float_as_uint(u[0]), __float_as_uint(u[1]), __float_as_uint(u[2]), __float_as_uint(u[3])}, ru, (int)off, 0, 0);
    };
    floatx4 res[MODE == 1 ? 8 : 1];  // MODE 1: residual rows of G for the previous tile's 8 stores
    // MODE 1: `beta` carries the tensor the residual rows are read from when the output goes to a different buffer (rtfs_dp_convt_fwd_to: the training
    // step keeps the input of a dual-path stage for the adjoint - in place it had to copy G first); nullptr = in place
    const __amdgpu_buffer_rsrc_t rres = MODE == 1 && beta != nullptr ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(beta), 0, (int)(((long long)(S >> map.seq_shift) * map.stride_hi) * 4), 0x00020000) : ru;
    auto res1 = [&](int k) {
        const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(rres, (int)((k & 1) ? oo[k >> 1] : oe[k >> 1]), 0, 0);
        res[MODE == 1 ? k : 0] = floatx4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    };
    auto out1 = [&](const floatx4 (&h)[3][4], int k) {
        const int rt = k >> 1;
        if (FFA_ABL & 128) {
            store_u(h[k & 1][rt], (k & 1) ? oo[rt] : oe[rt]);
            return;
        }
        if ((k & 1) == 0) {
            const floatx4 nx = rt < 3 ? h[0][rt + 1] : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) aps[c] = shl1(h[0][rt][c], nx[c]);
            if constexpr (MODE != 1) store_u(h[0][rt] + h[1][rt], oe[rt]);
            else store_u((h[0][rt] + h[1][rt]) + (bias4 + res[MODE == 1 ? k : 0]), oe[rt]);
        } else {
            if constexpr (MODE != 1) store_u(h[2][rt] - (aps + h[1][rt]), oo[rt]);
            else store_u((h[2][rt] - (aps + h[1][rt])) + (bias4 + res[MODE == 1 ? k : 0]), oo[rt]);
        }
    };
    auto body = [&](auto par, int tile, floatx4 (&acc)[3][4], const floatx4 (&accp)[3][4]) {
        constexpr int PAR = decltype(par)::value;
        float* sn = &slab[PAR ^ 1][0][0];
        const float* bpn[4];
        FfaTile tn;
        // the tile's other work as 48 slots, one per 16-MFMA group:
        //   1 barrier (previous tile left: its slab may be overwritten) | 2-21 next tile's raw rows -> LayerNorm -> that slab | 22-29 previous tile's
        //   accumulators -> U0 (8 stores) | 30-40 loads of the tile after next | 42 barrier (next slab complete) | 44 next tile's row geometry
        auto piece = [&](int sl_) {
            if (sl_ == 0) tn = tile_of(min(tile + 1, t1 - 1));
            if (sl_ == 1 && !(FFA_ABL & 8)) __syncthreads();
            if (sl_ >= 2 && sl_ < 22 && !(FFA_ABL & 1)) {
                const int it = (sl_ - 2) >> 2, ph = (sl_ - 2) & 3;
                if (MODE != 0) {
                    if (ph == 2) raw_c(sn, sinfo[it], it);
                    if (MODE == 1 && it >= 1 && ph < 2) res1(2 * (it - 1) + ph);  // slots 6, 7, 10, 11, 14, 15, 18, 19: the residual rows of the previous tile's outputs
                } else if (FFA_ABL & 32) {
                    if (ph == 2) st4(sn + sinfo[it], rawe[it]), st4(sn + kFfaPlane + sinfo[it], rawo[it]), st4(sn + 2 * kFfaPlane + sinfo[it], rawo[it]);
                } else {
                    if (ph == 0) ln_mean(it);
                    if (ph == 1) ln_rstd();
                    if (ph == 2) ln_c(sn, sinfo[it]);
                }
            }
            if (sl_ == (MODE == 1 ? 2 : 18) && !(FFA_ABL & 2)) out_offsets(tile > t0 ? tile - 1 : -1);
            if (sl_ >= 22 && sl_ < 30 && !(FFA_ABL & 2)) out1(accp, sl_ - 22);
            if (sl_ == 30) tf = tile_of(min(tile + 2, t1 - 1));
            if (sl_ >= 31 && sl_ < 31 + NIT && !(FFA_ABL & 4)) fetch1(sl_ - 31);
            if (sl_ == 42 && !(FFA_ABL & 8)) __syncthreads();
            if (sl_ == 44) rows_in(tn, sn, bpn);
        };
        auto sub_loop = [&](auto sub_, const float4 (&wq)[16]) {
            constexpr int SUB = decltype(sub_)::value;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int lin = 16 * SUB + q;
                if (lin + 1 < 48) {
                    const int o = frag_off((lin + 1) >> 4, (lin + 1) & 15);
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) eb[(lin + 1) & 1][rt] = ld4(bp[rt] + o);
                } else {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) eb[0][rt] = ld4(bpn[rt] + frag_off(0, 0));  // step 0 of the next tile
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4(&e)[4] = eb[lin & 1];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)  // (the chain starts from an inline-constant zero C operand: no register clearing)
                    acc[SUB][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].x, e[rt].x, q == 0 ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[SUB][rt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                piece(lin);
#if !(FFA_ABL & 256)
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[SUB][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].y, e[rt].y, acc[SUB][rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[SUB][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].z, e[rt].z, acc[SUB][rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[SUB][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[q].w, e[rt].w, acc[SUB][rt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        sub_loop(std::integral_constant<int, 0>{}, wA);
        sub_loop(std::integral_constant<int, 1>{}, wB);
        sub_loop(std::integral_constant<int, 2>{}, wS);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) bp[rt] = bpn[rt];
    };
    bool last_is_b = false;
#pragma unroll 1
    for (int tile = t0; tile < t1; tile += 2) {
        body(std::integral_constant<int, 0>{}, tile, accA, accB);
        last_is_b = tile + 1 < t1;
        if (!last_is_b) break;
        body(std::integral_constant<int, 1>{}, tile + 1, accB, accA);
    }
    out_offsets(t1 - 1);
    if constexpr (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) res1(k);
    }
    if (last_is_b) {
#pragma unroll
        for (int k = 0; k < 8; ++k) out1(accB, k);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) out1(accA, k);
    }
}

// ConvTranspose kernel, weight-stationary (round 3, large batches, fp32): the structure of unfold_ws_kernel on the zero-padded SRU output.
// W' (64 x 512 = 128 KB) lives in registers - wave (wn, wm): output channels 32 wn .. + 31 x all 512 k = 256 registers per lane, the MFMA A
// operand; the two waves of a column half hold the same fragments and own one 64-row tile each (wm) - and a workgroup walks pairs of 64-row
// tiles (two frequency sequences, or the two halves of one time sequence).  Between the steps of the uninterrupted K loop (512 MFMAs per wave
// and pair) ride, one small piece per step:
//   steps 0-8    the NEXT pair's h3 rows (requested one pair ago; rows outside the sequence come back as zeros from the buffer range check)
//                -> the other two slabs;
//   steps 9-16   the PREVIOUS pair's accumulators + bias + residual rows (fetched one pair ago) -> G, in place;
//   steps 17-24  this pair's residual rows;   steps 25-33  the h3 rows of the pair after next;
// one barrier per pair.  Same products in the same k order per accumulator as convt_gemm2_kernel => bit-identical G.
template <int NT = 0>  // 0 fp32; 1 / 3: bf16 / split-bf16 MFMA (Wt host-PACKED, the slabs packed on store; operand tuples as in unfold_ws_kernel)
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void convt_ws_kernel(SeqMap map, const float* __restrict__ src,
                                                                                                     const float* __restrict__ Wt,
                                                                                                     const float* __restrict__ bias, float* __restrict__ dst,
                                                                                                     int S, int tiles_per_seq, int total_tiles) {
    constexpr int NIT = (2 * kSlabRows * 16 + 255) / 256;  // 9
    __shared__ __attribute__((aligned(16))) float slab[2][2][(kSlabRows + 1) * kSlabLd];  // [buffer][tile of the pair] (+ one scratch row)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int c4 = (threadIdx.x & 15) * 4;
    const int cst = NT == 0 ? c4 : (c4 >> 4) * 16 + ((c4 >> 2) & 1) * 4 + ((c4 >> 3) & 1) * 2;  // float offset of this thread's channel quad inside a slab row
    const int L = map.L;

    float4 wf[64];  // W' fragments: row n = 32 wn + i, k = 8 q + 4 kh .. +3  (k = 64 tap + channel)
#pragma unroll
    for (int q = 0; q < 64; ++q) wf[q] = ld4(Wt + (size_t)(32 * wn + i) * 512 + 8 * q + 4 * kh);

    const int npairs = (total_tiles + 1) / 2;
    const int p0 = (int)((long long)npairs * blockIdx.x / gridDim.x), p1 = (int)((long long)npairs * (blockIdx.x + 1) / gridDim.x);
    if (p0 >= p1) return;
    // h3 [S][L][64] and G through buffer descriptors: out-of-range offsets read zeros / drop the store
    constexpr unsigned kNowhere = 0xFFFFFF00u;
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)((long long)S * L * 256), 0x00020000);
    const long long gbytes = ((long long)(S >> map.seq_shift) * map.stride_hi) * 4;  // the whole G tensor
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)gbytes, 0x00020000);
    auto bld = [](const __amdgpu_buffer_rsrc_t& r, unsigned off) {
        const uint4v v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return f4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    // staging: thread -> per iteration (tile of the pair, slab row): constants of the thread
    float4 sraw[NIT];
    int s_tile = 0, s_seq[2], s_m0[2];  // the pair being fetched
    auto locate = [&](int pair, int (&sq)[2], int (&mm)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int gt = min(pair * 2 + q, total_tiles - 1);
            sq[q] = gt / tiles_per_seq;
            mm[q] = (gt - sq[q] * tiles_per_seq) * 64;
        }
    };
    auto fetch_begin = [&](int pair) { locate(min(pair, p1 - 1), s_seq, s_m0); };  // (pairs past the end re-fetch the last one: never used)
    auto fetch1 = [&](int it) {
        const int idx = threadIdx.x + it * 256;
        const int q = idx >= kSlabRows * 16 ? 1 : 0;
        const int row = (idx - q * kSlabRows * 16) >> 4;
        const int l = s_m0[q] + row - 7;
        const unsigned off = ((unsigned)(s_seq[q] * L + l) * 64u + (unsigned)c4) * 4u;
        sraw[it] = bld(rh, (unsigned)l < (unsigned)L ? off : kNowhere);  // zero padding = rows outside the sequence
    };
    auto stage1 = [&](float (*sl)[(kSlabRows + 1) * kSlabLd], int it) {
        const int idx = threadIdx.x + it * 256;
        const int q = idx >= kSlabRows * 16 ? 1 : 0;
        const int row = min((idx - q * kSlabRows * 16) >> 4, kSlabRows);  // (the last iteration's surplus threads write the scratch row)
        if constexpr (NT == 0) {
            st4(sl[q] + row * kSlabLd + cst, sraw[it]);
        } else {
            const float4 pk = pack4<NT>(sraw[it]);
            *reinterpret_cast<float2*>(sl[q] + row * kSlabLd + cst) = make_float2(pk.x, pk.y);
            if constexpr (NT == 3) *reinterpret_cast<float2*>(sl[q] + row * kSlabLd + cst + 8) = make_float2(pk.z, pk.w);
        }
    };
    // write-back: lane = (row i of row tile m, channels 32 wn + 8 g + 4 kh .. +3); residual rows fetched one pair ahead
    floatx16 hold0, hold1;
    float4 res[8];
    float4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = ld4(bias + wn * 32 + 8 * g + 4 * kh);
    unsigned roff[2] = {kNowhere, kNowhere}, roff_prev[2] = {kNowhere, kNowhere};  // byte offset of this lane's two output rows (or nowhere)
    auto rows_of = [&](int pair, unsigned (&ro)[2]) {
        int sq[2], mm[2];
        locate(pair, sq, mm);
        const bool live = pair * 2 + wm < total_tiles;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = mm[wm] + 32 * m + i;
            ro[m] = live && row < map.npos ? map.off32(sq[wm], row) + (unsigned)(wn * 32 + 4 * kh) * 4u : kNowhere;
        }
    };
    auto res1 = [&](int it) { res[it] = bld(rg, roff[it >> 2] + (unsigned)(it & 3) * 32u); };
    auto out1 = [&](int it) {
        const float4 v = acc_group(it >> 2 ? hold1 : hold0, it & 3) + bq[it & 3] + res[it];
        __builtin_amdgcn_raw_buffer_store_b128(uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, rg,
                                               (int)(roff_prev[it >> 2] + (unsigned)(it & 3) * 32u), 0, 0);
    };

    fetch_begin(p0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) fetch1(it);
#pragma unroll
    for (int it = 0; it < NIT; ++it) stage1(slab[0], it);
#pragma unroll
    for (int it = 0; it < 8; ++it) res1(it);  // (nowhere: zeros; the same sequence of memory operations as inside the loop)
    fetch_begin(p0 + 1);
#pragma unroll
    for (int it = 0; it < NIT; ++it) fetch1(it);
    bf16x8 whi[NT ? 32 : 1], wlo[NT ? 32 : 1];  // bf16 modes: operand tuples of step q2 (see unfold_ws_kernel)
    if constexpr (NT == 0) {
#pragma unroll
        for (int q = 0; q < 44; ++q) asm volatile("" : "+a"(wf[q].x), "+a"(wf[q].y), "+a"(wf[q].z), "+a"(wf[q].w));  // (see unfold_ws_kernel)
    } else {
#pragma unroll
        for (int q2 = 0; q2 < 32; ++q2) {
            const float4 s0 = wf[2 * q2], s1 = wf[2 * q2 + 1];
            whi[q2] = __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(s0.x), __float_as_uint(s0.y), __float_as_uint(s1.x), __float_as_uint(s1.y)});
            wlo[q2] = __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(s0.z), __float_as_uint(s0.w), __float_as_uint(s1.z), __float_as_uint(s1.w)});
        }
#pragma unroll
        for (int q2 = 0; q2 < 32; ++q2) asm volatile("" : "+a"(whi[q2]));
    }
    __syncthreads();

#pragma unroll 1
    for (int pair = p0; pair < p1; ++pair) {
        const int cur = (pair - p0) & 1;
        float(*sn)[(kSlabRows + 1) * kSlabLd] = slab[cur ^ 1];
        rows_of(pair, roff);
        const float* bp[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) bp[m] = slab[cur][wm] + (32 * m + i) * kSlabLd + 4 * kh;
        floatx16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        auto piece = [&](int q) {  // 64 slots: one per step of the fp32 K loop, two per 16-k step of the bf16 loops
            if (q < NIT) stage1(sn, q);
            if (q >= 9 && q < 17) out1(q - 9);
            if (q >= 17 && q < 25) res1(q - 17);
            if (q == 25) fetch_begin(pair + 2);
            if (q >= 25 && q < 25 + NIT) fetch1(q - 25);
        };
        float4 eb[2][2];      // fp32: [buffer][row tile]
        float4 ebp[2][2][2];  // bf16: [buffer][row tile][hi | lo]
        if constexpr (NT == 0) {
            eb[0][0] = ld4(bp[0]), eb[0][1] = ld4(bp[1]);
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m) ebp[0][m][0] = ld4(bp[m]), ebp[0][m][1] = ld4(bp[m] + 8);
        }
        auto half_loop = [&](auto qh) {
            if constexpr (NT == 0) {
#pragma unroll
                for (int qq = 0; qq < 32; ++qq) {
                    const int q = decltype(qh)::value * 32 + qq;
                    if (q + 1 < 64) {
                        const int o = ((q + 1) >> 3) * kSlabLd + ((q + 1) & 7) * 8;  // tap (q + 1) / 8 = slab row offset, channel 8 ((q + 1) % 8)
                        eb[(q + 1) & 1][0] = ld4(bp[0] + o), eb[(q + 1) & 1][1] = ld4(bp[1] + o);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float4 e0 = eb[q & 1][0], e1 = eb[q & 1][1];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].x, e0.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].x, e1.x, acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    piece(q);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].y, e0.y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].y, e1.y, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].z, e0.z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].z, e1.z, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].w, e0.w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[q].w, e1.w, acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                auto tuple = [](float4 v) { return __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}); };
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) {
                    const int q2 = decltype(qh)::value * 16 + qq;
                    if (q2 + 1 < 32) {
                        const int o = ((q2 + 1) >> 2) * kSlabLd + ((q2 + 1) & 3) * 16;  // tap (q2 + 1) / 4, channel group (q2 + 1) % 4
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            ebp[(q2 + 1) & 1][m][0] = ld4(bp[m] + o);
                            if constexpr (NT == 3) ebp[(q2 + 1) & 1][m][1] = ld4(bp[m] + o + 8);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 h0 = tuple(ebp[q2 & 1][0][0]), l0 = tuple(ebp[q2 & 1][0][1]), h1 = tuple(ebp[q2 & 1][1][0]), l1 = tuple(ebp[q2 & 1][1][1]);
                    if constexpr (NT == 3) {  // per accumulator the order of mma32<3>: lo.hi, hi.lo, hi.hi; the two chains alternate
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[q2], h0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[q2], h1, acc[1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        piece(2 * q2);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[q2], l0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[q2], l1, acc[1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        piece(2 * q2 + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[q2], h0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[q2], h1, acc[1], 0, 0, 0);
                    } else {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[q2], h0, acc[0], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        piece(2 * q2);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi[q2], h1, acc[1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        piece(2 * q2 + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        half_loop(std::integral_constant<int, 0>{});
        half_loop(std::integral_constant<int, 1>{});
        hold0 = acc[0], hold1 = acc[1];
        roff_prev[0] = roff[0], roff_prev[1] = roff[1];
        __syncthreads();  // every wave has read its last fragment of these slabs; the next pair's slabs are complete
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) out1(it);
}

// Bidirectional SRU recurrence, one wave per sequence: lane = dir*32 + j.
//   KM == 4 (layer 0): U[s][l][lane][4] = (u0, u1, u2, x')          -> one 16-byte load per lane per step
//   KM == 3 (layers 1-3): U[s][l][m][lane], m = 0..2, skip input x' = X[s][l][lane] * scale_x
// Loads do not depend on the carried state, so UNR steps are fetched ahead of the dependent chain.
template <int KM>
__global__ __launch_bounds__(256) void sru_scan_kernel(const float* __restrict__ U, const float* __restrict__ X, const float* __restrict__ wc,
                                                       const float* __restrict__ bias, float scale_x, float* __restrict__ Hout, int S, int L) {
    constexpr int UNR = 8;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const int lane = threadIdx.x & 63;
    const bool rev = lane >= 32;
    const float wf = wc[lane], wr = wc[64 + lane], bf = bias[lane], br = bias[64 + lane];
    const float* u = U + (size_t)s * L * 64 * KM + (KM == 4 ? lane * 4 : lane);
    const float* x = X + (size_t)s * L * 64 + lane;
    float* h = Hout + (size_t)s * L * 64 + lane;
    float c = 0.f;
    // Two register sets of UNR steps: the loads of batch i + 1 are issued before the recurrence of batch i (addresses clamped, never under
    // a branch - a load under even a wave-uniform branch is waited for before the next one issues), so a wave pays the HBM latency once,
    // not once per batch: with 2-4 waves per SIMD and a dependent chain per step there is little else to hide it.
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

// ------------------------------------------------------------------------------------------------
// SRU layers 1-3, input projection FUSED into the recurrence: U = h_prev . W never exists in HBM.
// One wave per sequence.  Per 32-step chunk the wave computes its own six 32x32 tiles U_(d,m)[step][j] = h_prev[t_d(step)] . W_(m,d,j)
// on the MFMA pipe (rows = the chunk's 32 steps in the scan order of direction d: t_0 = step, t_1 = L-1-step; columns = the 32
// hidden units) - 192 MFMAs - then every lane swaps one half of its accumulators with lane ^ 32, which leaves lane (d, j) holding
// all 32 steps of ITS column for m = 0,1,2 in registers, and the recurrence runs out of registers.  HBM traffic per layer: read h_prev, write h (2 x 58 MB at B = 32) instead of
// GEMM + scan's 524 MB; the weight (48 KB) sits in LDS for the 4 waves of the workgroup.
//   A operand: lane (i, kh) supplies h_prev[t(i)][32kh + s] at MFMA step s (the K order is free as long as A and B agree), i.e.
//   32 consecutive floats of one row = 8 x 16-byte loads;  B operand: W[(m, d, j=i)][32kh + s] via ds_read_b128.
// ------------------------------------------------------------------------------------------------
// Round 4 (same arithmetic, same bits): (1) ONE 8-wave workgroup per CU shares the 51 KB weight tile, staged with all of a thread's loads in flight (the
// staging loop used to wait for each 16-byte load: ~6 us per workgroup, twice per CU on the freq path); (2) the weight fragments of MFMA group g + 1 are
// requested BEFORE the 8 MFMAs of group g and pinned there with sched_barrier (hipcc placed each ds_read_b128 pair directly in front of its first MFMA
// and waited: ~100 idle cycles of the matrix pipe per 512); (3) the recurrence issues NO global load: the skip inputs x' - the rows the MFMAs just
// consumed as A fragments - are written to a wave-private LDS tile between the first MFMA groups and read back with ds_read_b32.  With loads and the
// 32 per-step stores in flight together every wait for a load is `s_waitcnt vmcnt(0)` (loads and stores complete out of order with respect to each
// other, so hipcc cannot count past a store): four full drains of store acknowledgements + the next chunk's A prefetch per chunk.  Measured on the
// bench shapes (tools/sru_bench.py, same box): see DESIGN.md section 5.
// NW = waves (= sequences) per workgroup: 8 at large batch; 4 below 2048 sequences, where the workgroups do not fill the chip anyway and two waves on a
// SIMD only stretch each other's MFMA phases (batch 1: 33 -> 48 us per launch with 8, measured against the round-3 kernel).
template <bool SAVE_C, int NT = 0, int NW = 8>  // SAVE_C (training): also stores the cell states and the pre-activations U (the adjoint's inputs); NT: common.h
__global__ __launch_bounds__(NW * 64, 2) void sru_layer_kernel(const float* __restrict__ Hprev, const float* __restrict__ Wt, const float* __restrict__ wc,
                                                           const float* __restrict__ bias, float scale_x, float* __restrict__ Hout,
                                                           float* __restrict__ Cout, float* __restrict__ Uout, int S, int L) {
    // LDX: ds_write_b128 of 8 consecutive rows (one lane group) covers 32 distinct banks; the per-step reads are consecutive floats
    // LDW, NT = 6: a weight row is split ONCE, here, into three bf16 planes (hi | mid | lo, 128 bytes each, + 16: 25 sixteen-byte slots) - a lane's
    // ds_read_b128 is a finished operand tuple (round 5; splitting every fragment read in registers cost ~1300 VALU instructions per 32-step chunk)
    constexpr int LDW = NT == 6 ? 100 : 68, LDX = 36;
    __shared__ __attribute__((aligned(16))) float Ws[192 * LDW];
    __shared__ __attribute__((aligned(16))) float Xs[NW][2][32 * LDX];  // per wave: x' of the chunk's 32 steps, [dir][step][j]
    // the gate rows (m = 1, 2) are pre-scaled by -log2(e): the recurrence then needs fma, v_exp, add, v_rcp per gate and nothing else
    {
        constexpr int NS = 3072 / (NW * 64);
        float4 stg[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) stg[k] = ld4(Wt + (size_t)(threadIdx.x + NW * 64 * k) * 4);  // (Wt is the plain fp32 weight for every NT: packed here, after the scaling)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = threadIdx.x + NW * 64 * k, n = idx >> 4, q4 = (idx & 15) * 4;
            if constexpr (NT == 6) {
                const float4 y = stg[k] * (n >= 64 ? kNegLog2e : 1.0f);
                const unsigned h0 = pk_bf16(y.x, y.y), h1 = pk_bf16(y.z, y.w);
                const float4 r = f4(y.x - __uint_as_float(h0 << 16), y.y - __uint_as_float(h0 & 0xffff0000u), y.z - __uint_as_float(h1 << 16),
                                    y.w - __uint_as_float(h1 & 0xffff0000u));
                const unsigned m0 = pk_bf16(r.x, r.y), m1 = pk_bf16(r.z, r.w);
                const unsigned l0 = pk_bf16(r.x - __uint_as_float(m0 << 16), r.y - __uint_as_float(m0 & 0xffff0000u));
                const unsigned l1 = pk_bf16(r.z - __uint_as_float(m1 << 16), r.w - __uint_as_float(m1 & 0xffff0000u));
                float* o = Ws + n * LDW + (q4 >> 1);
                *reinterpret_cast<float2*>(o) = make_float2(__uint_as_float(h0), __uint_as_float(h1));
                *reinterpret_cast<float2*>(o + 32) = make_float2(__uint_as_float(m0), __uint_as_float(m1));
                *reinterpret_cast<float2*>(o + 64) = make_float2(__uint_as_float(l0), __uint_as_float(l1));
            } else
            st4(Ws + n * LDW + q4, pack4<NT>(stg[k] * (n >= 64 ? kNegLog2e : 1.0f)));
        }
    }
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.x * NW + wv;  // wave-uniform: bases below live in SGPRs
    const int nch = (L + 31) >> 5;
    if (s >= S) return;
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const bool rev = lane >= 32;
    const float wf = wc[lane] * kNegLog2e, wr = wc[64 + lane] * kNegLog2e;
    const float bf = bias[lane] * kNegLog2e, br = bias[64 + lane] * kNegLog2e;
    const float* hp = Hprev + (size_t)s * L * 64;
    float* hob = Hout + (size_t)s * L * 64;
    float* cob = Cout + (size_t)s * L * 64;
    float* uob = Uout + (size_t)s * L * 192;  // [l][m][lane]
    // The recurrence step of scan position sl touches row t = sl (forward lanes) or L-1-sl (reverse lanes): as a byte offset from the
    // sequence base that is  off0 + sl * dstr  with per-lane constants - one v_mad per access (wave-uniform base + 32-bit offset form)
    const int dstr = rev ? -256 : 256, off0 = (rev ? (L - 1) * 256 : 0) + lane * 4;
    const int off0u = (rev ? (L - 1) * 768 : 0) + lane * 4;  // same for the [l][3][64] pre-activation rows
    constexpr float kUnscale = 1.0f / kNegLog2e;
    float* xw = &Xs[wv][kh][i * LDX];          // this lane's row of the tile it WRITES: direction kh, local step i (its A-fragment row), columns 32 kh + 4 q of h_prev
    const float* xr = &Xs[wv][kh][i];          // and the column it READS: direction kh (= rev), hidden unit i, local step k at + k * LDX
    float c = 0.f;
#ifdef SRU_TIMING
    unsigned long long tm[24];
    int ntm = 0;
    tm[ntm++] = __builtin_amdgcn_s_memtime();
#endif
    // A fragments of the two directions (rows past the end are clamped; their steps are never scanned); the next chunk's are
    // fetched as soon as the MFMAs have consumed the current ones, i.e. under the 32 recurrence steps
    float4 a0[8], a1[8];
    auto load_a = [&](int sl0) {
        const int ta = min(sl0 + i, L - 1), tb = max(L - 1 - (sl0 + i), 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            a0[q] = ld4(hp + (size_t)ta * 64 + 32 * kh + 4 * q);
            a1[q] = ld4(hp + (size_t)tb * 64 + 32 * kh + 4 * q);
        }
    };
    load_a(0);
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const int sl0 = ch * 32;
        floatx16 acc[2][3];
        int woff = i * LDW + (NT == 6 ? 16 : 32) * kh;
        asm volatile("" : "+v"(woff));  // opaque per chunk: keeps hipcc from hoisting all 48 weight reads (192 VGPRs) out of the chunk
                                        // loop (the OFFSET is laundered, not the pointer, so the reads stay ds_read_b128)
        const float* wp = Ws + woff;
        // MFMA groups g = 3 qq + m: fp32: qq = 0..7 (k quad), 2 fragment reads + 8 MFMAs; bf16 forms: qq = 0..3 (k octet), 4 reads + 2 (x terms) MFMAs
        constexpr int NG = NT == 0 ? 24 : 12, NR = NT == 0 ? 2 : 4;
        float4 bb[2][NR];
        auto read_group = [&](float4(&dst)[NR], int g) {
            const int qq = g / 3, m = g % 3;
            if constexpr (NT == 0) {
                dst[0] = ld4(wp + (m * 64) * LDW + 4 * qq), dst[1] = ld4(wp + (m * 64 + 32) * LDW + 4 * qq);
            } else {
                dst[0] = ld4(wp + (m * 64) * LDW + 8 * qq), dst[1] = ld4(wp + (m * 64) * LDW + 8 * qq + 4);
                dst[2] = ld4(wp + (m * 64 + 32) * LDW + 8 * qq), dst[3] = ld4(wp + (m * 64 + 32) * LDW + 8 * qq + 4);
            }
        };
#ifdef SRU_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tm[ntm++] = __builtin_amdgcn_s_memtime();
#endif
        Frag fa0, fa1;
        if constexpr (NT == 6) {
            // 24 half groups (k octet qq, gate m, direction half hf): three plane reads one half group ahead, six MFMAs; 12 registers of fragments
            // per buffer (whole groups double-buffered spilled); the x' tile goes out during the first eight
            auto tuple = [](float4 v) { return __builtin_bit_cast(bf16x8, uint4v{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}); };
            float4 b6[2][3];
            auto read_half = [&](float4(&dst)[3], int hg) {
                const int g = hg >> 1, hf = hg & 1, qq = g / 3, m = g % 3;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) dst[pl] = ld4(wp + (m * 64 + 32 * hf) * LDW + 4 * qq + 32 * pl);
            };
            read_half(b6[0], 0);
#pragma unroll
            for (int hg = 0; hg < 24; ++hg) {
                const int g = hg >> 1, hf = hg & 1, qq = g / 3, m = g % 3;
                if (hg + 1 < 24) read_half(b6[(hg + 1) & 1], hg + 1);
                if (hg < 8) st4(xw + 4 * hg, kh ? a1[hg] : a0[hg]);
                __builtin_amdgcn_sched_barrier(0);
                if (m == 0 && hf == 0) fa0 = frag_split3(a0[2 * qq], a0[2 * qq + 1]);
                if (m == 0 && hf == 1) fa1 = frag_split3(a1[2 * qq], a1[2 * qq + 1]);
                const float4(&b)[3] = b6[hg & 1];
                const Frag bf{tuple(b[0]), tuple(b[2]), tuple(b[1])};  // (hi, lo, mid)
                if (qq == 0)
                    acc[hf][m] = mma32_first<6>(hf ? fa1 : fa0, bf);
                else
                    mma32<6>(acc[hf][m], hf ? fa1 : fa0, bf);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        read_group(bb[0], 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int qq = g / 3, m = g % 3;
            if (g + 1 < NG) read_group(bb[(g + 1) & 1], g + 1);
            if (g < 8) st4(xw + 4 * g, kh ? a1[g] : a0[g]);  // x' tile, one 16-byte piece per group (both fragments are still live: a*[g] is consumed by groups >= g)
            __builtin_amdgcn_sched_barrier(0);
            const float4(&b)[NR] = bb[g & 1];
            if constexpr (NT == 0) {
                if (qq == 0) {  // the chain starts from a zero C operand (an inline constant): no accumulator clearing
                    floatx16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[qq].x, b[0].x, z, 0, 0, 0);
                    acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[qq].x, b[1].x, z, 0, 0, 0);
                } else {
                    acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[qq].x, b[0].x, acc[0][m], 0, 0, 0);
                    acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[qq].x, b[1].x, acc[1][m], 0, 0, 0);
                }
                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[qq].y, b[0].y, acc[0][m], 0, 0, 0);
                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[qq].y, b[1].y, acc[1][m], 0, 0, 0);
                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[qq].z, b[0].z, acc[0][m], 0, 0, 0);
                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[qq].z, b[1].z, acc[1][m], 0, 0, 0);
                acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[qq].w, b[0].w, acc[0][m], 0, 0, 0);
                acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[qq].w, b[1].w, acc[1][m], 0, 0, 0);
            } else {  // 16 k per step: this lane's k = 32 kh + 8 qq .. + 7 (the fp32 fragments of quads 2 qq, 2 qq + 1, packed in registers)
                if (m == 0) fa0 = frag_f32<NT>(a0[2 * qq], a0[2 * qq + 1]), fa1 = frag_f32<NT>(a1[2 * qq], a1[2 * qq + 1]);
                const Frag b0 = frag_lds<NT>(b[0], b[1]), b1 = frag_lds<NT>(b[2], b[3]);
                if (qq == 0) {
                    acc[0][m] = mma32_first<NT>(fa0, b0);
                    acc[1][m] = mma32_first<NT>(fa1, b1);
                } else {
                    mma32<NT>(acc[0][m], fa0, b0);
                    mma32<NT>(acc[1][m], fa1, b1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (ch + 1 < nch) load_a(sl0 + 32);
        // Half exchange: v_permlane32_swap X, Y swaps lanes 32-63 of X with lanes 0-31 of Y.  With X = a dir-0 accumulator register and
        // Y = the same register of the dir-1 tile, lanes 0-31 end up with (own dir-0 rows, dir-0 rows of lane + 32) and lanes 32-63
        // with (dir-1 rows of lane - 32, own dir-1 rows): every lane holds all 32 steps of ITS direction, no selects.  Afterwards
        // acc[0][m][r] -> local step rho(r), acc[1][m][r] -> rho(r) + 4,  rho(r) = (r & 3) + 8 (r >> 2).
        // Inline asm (the builtin on accumulator elements is miscompiled by this hipcc), so the hazards are ours: nothing may move
        // across the barrier, 20 wait states cover the last MFMA's 16 passes before its result is read, and the swaps only
        // read MFMA results (no VALU write of an operand within two wait states).
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x0 = acc[0][m][r], y0 = acc[1][m][r];
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(y0));
                acc[0][m][r] = x0;
                acc[1][m][r] = y0;
            }
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#ifdef SRU_TIMING
        tm[ntm++] = __builtin_amdgcn_s_memtime();
#endif
        // recurrence: 32 steps out of registers; x' of local step k from the wave's LDS tile (written above by the lanes that held those rows as A
        // fragments: same wave, LDS operations of a wave complete in order - the fence keeps hipcc from moving the reads above the writes)
        // (On gfx950 the fp32 MFMA executes on the SIMD's fp32 vector lanes - its rate IS the vector rate - and holds them for 64 cycles per instruction:
        // next to the co-resident wave's MFMA phase every dependent VALU instruction of this chain waits for an MFMA to retire.  s_memtime stamps
        // (tools/sru_timeline.py): 32 steps take ~19k cycles beside the partner's MFMAs, ~4.5k for two waves scanning together.  Locking the two waves'
        // phases with workgroup barriers - MFMA phases together, recurrences together - measured the same 76-79 us as this free-running form: the
        // vector lanes are busy either way, the barrier skew eats what the shorter scans save.  In the bf16 forms the matrix pipe is a separate unit.)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef SRU_TIMING
        tm[ntm++] = __builtin_amdgcn_s_memtime();
#endif
        auto step = [&](int k) {
            const int sl = sl0 + k;
            const int r = (k & 3) + 4 * (k >> 3), sel = (k >> 2) & 1;
            const float u0 = acc[sel][0][r], u1 = acc[sel][1][r], u2 = acc[sel][2][r];
            const unsigned off = (unsigned)(off0 + sl * dstr);
            const float x = xr[k * LDX] * scale_x;
            const float f = sigmoid_from_exp2arg(fmaf(wf, c, u1) + bf);
            const float rg = sigmoid_from_exp2arg(fmaf(wr, c, u2) + br);
            c = u0 + (c - u0) * f;
            st1_off(hob, off, x + (c - x) * rg);
            if (SAVE_C) {
                st1_off(cob, off, c);
                const unsigned offu = (unsigned)(off0u + sl * (3 * dstr));
                st1_off(uob, offu, u0), st1_off(uob, offu + 256, u1 * kUnscale), st1_off(uob, offu + 512, u2 * kUnscale);
            }
        };
        if (sl0 + 32 <= L) {  // a full chunk: ONE basic block (a branch per step kept every x' read inside its step: LDS latency on the chain, 32 times)
#pragma unroll
            for (int k = 0; k < 32; ++k) step(k);
        } else {  // the sequence's last chunk: steps past the end are skipped (wave-uniform branches)
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (sl0 + k < L) step(k);
        }
#ifdef SRU_TIMING
        tm[ntm++] = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_wave_barrier();  // (the next chunk's x' writes stay behind this chunk's reads)
#ifdef SRU_TIMING
        tm[ntm++] = __builtin_amdgcn_s_memtime();
#endif
    }
#ifdef SRU_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(hob);
        for (int k = 0; k < 24; ++k) o[k] = k < ntm ? tm[k] : 0ull;
    }
#endif
}

// Small-batch form of the fused SRU layer (fp32, inference): ONE WAVE PER (sequence, direction).  sru_layer_kernel gives a sequence one wave, whose
// per-chunk chain is 192 MFMAs (12.3k cycles) + 32 recurrence steps; below ~512 sequences most SIMDs idle while each sequence walks that chain (batch 1:
// 64 / 125 sequences, 29 us per launch = 36 x 29 us = a quarter of the forward); used below 2048 sequences.  Here a wave owns one direction: 96 MFMAs per chunk (three 32 x 32 tiles
// U_m[step][j]), then lanes 32-63 hand their 16 steps of every column to lanes 0-31 (v_permlane32_swap into a second register set) and lanes 0-31 run the
// recurrence of their direction.  Same products in the same order per accumulator, same recurrence arithmetic: bit-identical to sru_layer_kernel.
__global__ __launch_bounds__(256, 2) void sru_layer_dir_kernel(const float* __restrict__ Hprev, const float* __restrict__ Wt, const float* __restrict__ wc,
                                                               const float* __restrict__ bias, float scale_x, float* __restrict__ Hout, int S, int L) {
    constexpr int LDW = 68, LDX = 36;
    __shared__ __attribute__((aligned(16))) float Ws[192 * LDW];
    __shared__ __attribute__((aligned(16))) float Xs[4][32 * LDX];  // per wave: x' of the chunk's 32 steps of its direction, [step][j]
    {
        float4 stg[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) stg[k] = ld4(Wt + (size_t)(threadIdx.x + 256 * k) * 4);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int idx = threadIdx.x + 256 * k, n = idx >> 4, q4 = (idx & 15) * 4;
            st4(Ws + n * LDW + q4, stg[k] * (n >= 64 ? kNegLog2e : 1.0f));
        }
    }
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * 4 + wv;  // (sequence, direction) pair, wave-uniform
    const int s = unit >> 1, d = unit & 1;
    if (s >= S) return;
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const float wf = wc[d * 32 + i] * kNegLog2e, wr = wc[64 + d * 32 + i] * kNegLog2e;
    const float bf = bias[d * 32 + i] * kNegLog2e, br = bias[64 + d * 32 + i] * kNegLog2e;
    const float* hp = Hprev + (size_t)s * L * 64;
    float* hob = Hout + (size_t)s * L * 64;
    const int dstr = d ? -256 : 256, off0 = (d ? (L - 1) * 256 : 0) + (d * 32 + i) * 4;  // byte offset of this column in the row of scan position 0
    float* xw = &Xs[wv][i * LDX];    // row i of the tile this lane writes (lanes with kh == d hold columns 32 d .. 32 d + 31 of their A rows)
    const float* xr = &Xs[wv][i];    // column i, local step k at + k * LDX
    float c = 0.f;
    const int nch = (L + 31) >> 5;
    float4 a[8];
    auto load_a = [&](int sl0) {
        const int t = d ? max(L - 1 - (sl0 + i), 0) : min(sl0 + i, L - 1);  // rows past the end are clamped; their steps are never scanned
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = ld4(hp + (size_t)t * 64 + 32 * kh + 4 * q);
    };
    load_a(0);
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const int sl0 = ch * 32;
        floatx16 acc[3];
        int woff = (d * 32 + i) * LDW + 32 * kh;
        asm volatile("" : "+v"(woff));  // (see sru_layer_kernel)
        const float* wp = Ws + woff;
        float4 bb[2][3];
#pragma unroll
        for (int m = 0; m < 3; ++m) bb[0][m] = ld4(wp + (m * 64) * LDW);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q + 1 < 8) {
#pragma unroll
                for (int m = 0; m < 3; ++m) bb[(q + 1) & 1][m] = ld4(wp + (m * 64) * LDW + 4 * (q + 1));
            }
            if (kh == d) st4(xw + 4 * q, a[q]);  // x' tile: this direction's half of the A rows
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float4 b = bb[q & 1][m];
                if (q == 0) {
                    floatx16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b.x, z, 0, 0, 0);
                } else {
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b.x, acc[m], 0, 0, 0);
                }
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b.y, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b.z, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b.w, acc[m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch + 1 < nch) load_a(sl0 + 32);
        // lanes 32-63 -> lanes 0-31: after the swap oth[m][r] of lane i holds what lane i + 32 had in acc[m][r], i.e. local step rho(r) + 4 (own: rho(r)),
        // rho(r) = (r & 3) + 8 (r >> 2)
        // (the swap's operands must not have been written by a VALU instruction within two wait states - see sru_layer_kernel: the second set is
        // zeroed BEFORE the scheduling barrier, and both registers are swapped in place so hipcc has no copy to insert in front of a swap)
        floatx16 oth[3];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) oth[m][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x0 = acc[m][r], y0 = oth[m][r];
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(y0));
                acc[m][r] = x0;
                oth[m][r] = y0;
            }
        asm volatile("s_nop 1" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto step = [&](int k) {
            const int sl = sl0 + k;
            const int r = (k & 3) + 4 * (k >> 3), sel = (k >> 2) & 1;
            const float u0 = sel ? oth[0][r] : acc[0][r], u1 = sel ? oth[1][r] : acc[1][r], u2 = sel ? oth[2][r] : acc[2][r];
            const unsigned off = (unsigned)(off0 + sl * dstr);
            const float x = xr[k * LDX] * scale_x;
            const float f = sigmoid_from_exp2arg(fmaf(wf, c, u1) + bf);
            const float rg = sigmoid_from_exp2arg(fmaf(wr, c, u2) + br);
            c = u0 + (c - u0) * f;
            if (kh == 0) st1_off(hob, off, x + (c - x) * rg);
        };
        if (sl0 + 32 <= L) {
#pragma unroll
            for (int k = 0; k < 32; ++k) step(k);
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (sl0 + k < L) step(k);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Smallest batches (a few hundred sequences): ONE WORKGROUP PER (sequence, direction).  sru_layer_dir_kernel's wave walks 96 MFMAs + 32 recurrence steps
// per chunk in line (batch 1: 250 / 128 waves on 1024 SIMDs, 19 us per launch, 36 launches = a quarter of the forward).  Here waves 0-2 each compute ONE
// gate's 32-step x 32-unit tile of U per chunk (32 MFMAs; the gate's 32 weight rows live in 32 registers - no weight staging in LDS) and hand it over
// through LDS; wave 3 runs the recurrence of the PREVIOUS chunk meanwhile (two U / x' buffers, one barrier per chunk).  Same products in the same order
// per accumulator, same recurrence arithmetic: bit-identical to sru_layer_kernel / sru_layer_dir_kernel.
constexpr int kSruSplitBelow = 257;  // sequences (up to 512 workgroups); above: sru_layer_dir_kernel.  tools/sru_bench.py f32 <B> <form>, form 2 -> 3:
// 125 x 57: 16.0 -> 11.6 us, 64 x 118: 26.3 -> 15.9, 250 x 57: 16.6 -> 15.0, 128 x 118: 26.8 -> 16.6, 256 x 118: 27.8 -> 22.3; 500 x 57: 18.0 -> 21.7 (not taken)

__global__ __launch_bounds__(256) void sru_layer_split_kernel(const float* __restrict__ Hprev, const float* __restrict__ Wt, const float* __restrict__ wc,
                                                              const float* __restrict__ bias, float scale_x, float* __restrict__ Hout, int S, int L) {
    constexpr int LDU = 104, LDX = 36;  // U rows [3 gates][32] + 8: the two lane halves' rows (4 apart) fall 32 banks apart
    __shared__ __attribute__((aligned(16))) float Us[2][32 * LDU];
    __shared__ __attribute__((aligned(16))) float Xs[2][32 * LDX];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = blockIdx.x >> 1, d = blockIdx.x & 1;
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const float* hp = Hprev + (size_t)s * L * 64;
    const int nch = (L + 31) >> 5;
    if (wv < 3) {
        float4 b[8];
        {
            const float* wr = Wt + (size_t)(wv * 64 + d * 32 + i) * 64 + 32 * kh;
            const float sc = wv >= 1 ? kNegLog2e : 1.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = ld4(wr + 4 * q) * sc;
        }
        float4 a[8];
        auto load_a = [&](int sl0) {
            const int t = d ? max(L - 1 - (sl0 + i), 0) : min(sl0 + i, L - 1);  // rows past the end are clamped; their steps are never scanned
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = ld4(hp + (size_t)t * 64 + 32 * kh + 4 * q);
        };
        load_a(0);
#pragma unroll 1
        for (int ch = 0; ch <= nch; ++ch) {
            if (ch < nch) {
                float* us = Us[ch & 1];
                if (wv == 0 && kh == d) {  // x' tile: this direction's half of the A rows
#pragma unroll
                    for (int q = 0; q < 8; ++q) st4(&Xs[ch & 1][i * LDX + 4 * q], a[q]);
                }
                floatx16 acc;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (q == 0) {
                        floatx16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, z, 0, 0, 0);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
                }
                if (ch + 1 < nch) load_a(32 * (ch + 1));
                // acc[r] of lane (i, kh): local step (r & 3) + 8 (r >> 2) + 4 kh, unit i of gate wv
#pragma unroll
                for (int r = 0; r < 16; ++r) us[((r & 3) + 8 * (r >> 2) + 4 * kh) * LDU + wv * 32 + i] = acc[r];
            }
            __syncthreads();
        }
    } else {
        const float wf = wc[d * 32 + i] * kNegLog2e, wr = wc[64 + d * 32 + i] * kNegLog2e;
        const float bf = bias[d * 32 + i] * kNegLog2e, br = bias[64 + d * 32 + i] * kNegLog2e;
        float* hob = Hout + (size_t)s * L * 64;
        const int dstr = d ? -256 : 256, off0 = (d ? (L - 1) * 256 : 0) + (d * 32 + i) * 4;  // byte offset of this column in the row of scan position 0
        float c = 0.f;
#pragma unroll 1
        for (int ch = 0; ch <= nch; ++ch) {
            if (ch >= 1 && kh == 0) {
                const int sl0 = 32 * (ch - 1);
                const float* us = Us[(ch - 1) & 1] + i;
                const float* xs = Xs[(ch - 1) & 1] + i;
                auto step = [&](int k) {
                    const float u0 = us[k * LDU], u1 = us[k * LDU + 32], u2 = us[k * LDU + 64];
                    const unsigned off = (unsigned)(off0 + (sl0 + k) * dstr);
                    const float x = xs[k * LDX] * scale_x;
                    const float f = sigmoid_from_exp2arg(fmaf(wf, c, u1) + bf);
                    const float rg = sigmoid_from_exp2arg(fmaf(wr, c, u2) + br);
                    c = u0 + (c - u0) * f;
                    st1_off(hob, off, x + (c - x) * rg);
                };
                if (sl0 + 32 <= L) {
#pragma unroll
                    for (int k = 0; k < 32; ++k) step(k);
                } else {
#pragma unroll
                    for (int k = 0; k < 32; ++k)
                        if (sl0 + k < L) step(k);
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace rtfs

using namespace rtfs;

static SeqMap make_map(int dim, int B, int T2) {
    SeqMap m;
    if (dim == 4) {  // along F: one sequence per (b, t2), contiguous [F2][64]
        m.seq_div = 1, m.stride_hi = (long long)kF2 * kH, m.stride_lo = 0, m.pos_stride = kH, m.npos = kF2;
    } else {  // along T: one sequence per (b, f2)
        m.seq_div = kF2, m.stride_hi = (long long)T2 * kF2 * kH, m.stride_lo = kH, m.pos_stride = (long long)kF2 * kH, m.npos = T2;
    }
    m.L = m.npos - 7;
    m.seq_shift = dim == 4 ? 0 : 6;  // seq_div = 1 or kF2 = 64
    m.magicL = (unsigned)((1ULL << 32) / (unsigned)m.L) + 1u;
    return m;
}

template <int NT>
static int unfold_gemm_impl(const float* G, const float* gamma, const float* beta, const float* Wt, float* U0, int B, int T2, int dim, int variant, void* stream) {
    if ((dim != 3 && dim != 4) || B <= 0 || T2 < 8 || variant < 0 || variant > 3) return RTFS_EINVAL;
    SeqMap m = make_map(dim, B, T2);
    const int S = dim == 4 ? B * T2 : B * kF2;
    const int tps = (m.L + 63) / 64, total = S * tps;
    const int npairs = (total + 1) / 2, resident = 2 * 256;  // two 79.6 KB workgroups per CU
    if (npairs < 256 || (NT != 0 && m.L < 32)) {  // small batches: 64-row tiles put twice as many workgroups on the (otherwise half-empty) chip
        if (total < 256)  // ... and below one workgroup per CU, four 64-column workgroups per row tile
            hipLaunchKernelGGL((toeplitz_gemm_kernel<64, 1, 1, 16, 0, NT, 256>), dim3(tps, S, 4), dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, Wt, nullptr, U0);
        else
            hipLaunchKernelGGL((toeplitz_gemm_kernel<256, 2, 2, 16, 0, NT>), dim3(tps, S), dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, Wt, nullptr, U0);
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    }
    // fp32, large batch: the weight-stationary fast-FIR kernel (three half-rate 4-tap correlations, 0.775x the MFMAs; variant 3 = the direct form)
    if constexpr (NT == 0) {
        const int Lv = (m.L + 2) / 2;
        const long long Rv = (long long)S * Lv, ftiles = (Rv + 62) / 63;
        if (variant == 0 && Lv >= 21 && (long long)B * T2 * kF2 * kH * 4 < (1LL << 32) && (Rv + 256) * Lv < (1LL << 32) && (long long)S * m.L * 1024 < (1LL << 31) &&
            ftiles >= 8 * 64) {  // (>= 8 tiles per tile range to pay for the 192 KB weight read of each of its four workgroups)
            const unsigned magicLv = (unsigned)((1ULL << 32) / (unsigned)Lv) + 1u;
            if (dim == 4)
                hipLaunchKernelGGL(unfold_ffa_kernel<4>, dim3(256), dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, Wt, U0, S, Lv, magicLv, (int)ftiles);
            else
                hipLaunchKernelGGL(unfold_ffa_kernel<3>, dim3(256), dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, Wt, U0, S, Lv, magicLv, (int)ftiles);
            RTFS_LAUNCH_CHECK();
            return RTFS_OK;
        }
    }
    // six-term split, large batch: its own weight-stationary kernel (variant 2 keeps the LDS-staged kernel selectable for A/B)
    if constexpr (NT == 6) {
        if ((variant == 0 || variant == 3) && m.L >= 32 && (long long)B * T2 * kF2 * kH * 4 < (1LL << 32) && (long long)S * m.L * m.L < (1LL << 32) &&
            (long long)S * m.L * 1024 < (1LL << 31) && ((long long)S * m.L + 63) / 64 >= 8 * 64) {
            const int ftiles = (int)(((long long)S * m.L + 63) / 64);
            hipLaunchKernelGGL(unfold_ws6_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, Wt, U0, S, ftiles);
            RTFS_LAUNCH_CHECK();
            return RTFS_OK;
        }
    }
    // fp32 (variant 3) / bf16 / split-bf16, large batch: the direct weight-stationary kernel (variant 2 keeps the LDS-staged flattened-tile kernel selectable for A/B)
    if ((NT == 0 || NT == 1 || NT == 3) && (variant == 0 || variant == 3) && m.L >= 32 && (long long)B * T2 * kF2 * kH * 4 < (1LL << 32) && (long long)S * m.L * m.L < (1LL << 32) &&
        (long long)S * m.L * 1024 < (1LL << 31) && ((long long)S * m.L + 63) / 64 >= 8 * 128) {  // (>= 8 tiles per workgroup to pay for its 256 KB weight read)
        const int ftiles = (int)(((long long)S * m.L + 63) / 64);
        hipLaunchKernelGGL(unfold_ws_kernel<(NT == 6 ? 0 : NT)>, dim3(256), dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, Wt, U0, S, ftiles);
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    }
    const bool per_seq_tiles = variant == 1;  // second-generation kernel (tiles padded per sequence), kept selectable for A/B: same bits
    if ((NT != 0 || !per_seq_tiles) && m.L >= 32 && (long long)B * T2 * kF2 * kH * 4 < (1LL << 32) && (long long)S * m.L * m.L < (1LL << 32)) {  // a 64-row tile then spans at most three sequences (1 + L + L >= 64); 32-bit staging offsets
        const int ftiles = (int)(((long long)S * m.L + 63) / 64), fpairs = (ftiles + 1) / 2;
        const int units = 2 * fpairs;  // (tile pair, column half)
        hipLaunchKernelGGL(unfold_gemm128f_kernel<NT>, dim3(units < resident ? units : resident), dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta,
                           Wt, U0, S, ftiles);
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    }
    if (NT != 0) return RTFS_EINVAL;  // (not reached: the bf16 paths are served by the two kernels above)
    hipLaunchKernelGGL(unfold_gemm128_kernel, dim3(npairs < resident ? npairs : resident), dim3(256), 0, (hipStream_t)stream, m, G, gamma, beta, Wt, U0,
                       tps, total);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// rtfs_convt_bwd_input at large batch (fp32): unfold_ffa_kernel<DIM, 2>.  Returns 1 when the size is not eligible (the caller launches the direct kernel).
int rtfs::convt_bwd_input_ffa(const float* dG, const float* Wt, float* dH3, int B, int T2, int dim, hipStream_t stream) {
    SeqMap m = make_map(dim, B, T2);
    const int S = dim == 4 ? B * T2 : B * kF2;
    const int Lv = (m.L + 2) / 2;
    const long long Rv = (long long)S * Lv, ftiles = (Rv + 62) / 63;
    if (!(Lv >= 21 && (long long)B * T2 * kF2 * kH * 4 < (1LL << 32) && (long long)S * m.L * 256 < (1LL << 31) && (Rv + 256) * Lv < (1LL << 32) && ftiles >= 4 * 256)) return 1;
    const unsigned magicLv = (unsigned)((1ULL << 32) / (unsigned)Lv) + 1u;
    if (dim == 4)
        hipLaunchKernelGGL((unfold_ffa_kernel<4, 2>), dim3(256), dim3(256), 0, stream, m, dG, nullptr, nullptr, Wt, dH3, S, Lv, magicLv, (int)ftiles);
    else
        hipLaunchKernelGGL((unfold_ffa_kernel<3, 2>), dim3(256), dim3(256), 0, stream, m, dG, nullptr, nullptr, Wt, dH3, S, Lv, magicLv, (int)ftiles);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

template <int NT>
// Gres: the tensor the residual is read from (nullptr or G: in place).  Only the fast-FIR kernel reads it directly; every other form copies it to G first.
static int convt_impl(const float* H3, const float* Wt, const float* bias, float* G, int B, int T2, int dim, void* stream, int variant = 0, const float* Gres = nullptr) {
    if ((dim != 3 && dim != 4) || B <= 0 || T2 < 8) return RTFS_EINVAL;
    if (Gres == G) Gres = nullptr;
    SeqMap m = make_map(dim, B, T2);
    const int S = dim == 4 ? B * T2 : B * kF2;
    const int tps = (m.npos + 63) / 64, total = S * tps;
    // fp32, large batch: the weight-stationary fast-FIR kernel in its ConvTranspose mode (unfold_ffa_kernel<DIM, 1>: 0.775x the MFMAs of the direct form)
    if constexpr (NT == 0) {
        const int Lv = (m.npos + 2) / 2;
        const long long Rv = (long long)S * Lv, ftiles = (Rv + 62) / 63;
        if (variant == 0 && Lv >= 21 && (long long)B * T2 * kF2 * kH * 4 < (1LL << 31) && (long long)S * m.L * 256 < (1LL << 31) && (Rv + 256) * Lv < (1LL << 32) &&
            ftiles >= 4 * 256) {
            const unsigned magicLv = (unsigned)((1ULL << 32) / (unsigned)Lv) + 1u;
            if (dim == 4)
                hipLaunchKernelGGL((unfold_ffa_kernel<4, 1>), dim3(256), dim3(256), 0, (hipStream_t)stream, m, H3, bias, Gres, Wt, G, S, Lv, magicLv, (int)ftiles);
            else
                hipLaunchKernelGGL((unfold_ffa_kernel<3, 1>), dim3(256), dim3(256), 0, (hipStream_t)stream, m, H3, bias, Gres, Wt, G, S, Lv, magicLv, (int)ftiles);
            RTFS_LAUNCH_CHECK();
            return RTFS_OK;
        }
    }
    if (Gres != nullptr && hipMemcpyAsync(G, Gres, (size_t)B * T2 * kF2 * kH * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return RTFS_ELAUNCH;
    // fp32 (variant 1), bf16 modes; large batch (>= 3 tile pairs per CU - measured: 84.6 vs 90.5 us at 3.9 pairs, 49.4 vs 47.6 us at 2; 32-bit offsets): the direct weight-stationary kernel
    if ((NT == 0 || NT == 1 || NT == 3) && total >= 2 * 3 * 256 && (long long)B * T2 * kF2 * kH * 4 < (1LL << 31) && (long long)S * m.L * 256 < (1LL << 31)) {
        hipLaunchKernelGGL(convt_ws_kernel<(NT == 6 ? 0 : NT)>, dim3(256), dim3(256), 0, (hipStream_t)stream, m, H3, Wt, bias, G, S, tps, total);
        RTFS_LAUNCH_CHECK();
        return RTFS_OK;
    }
    hipLaunchKernelGGL(convt_gemm2_kernel<NT>, dim3((total + 1) / 2), dim3(256), 0, (hipStream_t)stream, m, H3, Wt, bias, G, tps, total);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

extern "C" {

// G: [B][T2][F2][64].  U0: [S][L][256] with S = B*T2 (dim 4) or B*F2 (dim 3), L = npos-7, column = (dir*32+j)*4+m.
// Wt: [256][512], k index = kk*64 + c.

int rtfs_dp_unfold_gemm_fwd(const float* G, const float* gamma, const float* beta, const float* Wt, float* U0, int B, int T2, int dim,
                            int variant, void* stream) {
    return unfold_gemm_impl<0>(G, gamma, beta, Wt, U0, B, T2, dim, variant, stream);
}
// bf16 (terms 1) / split-bf16 (terms 3) MFMA; Wpk = host-packed weight (same indexing as Wt)
int rtfs_dp_unfold_gemm_fwd_bf16(const float* G, const float* gamma, const float* beta, const void* Wpk, float* U0, int B, int T2, int dim, int variant,
                                 int terms, void* stream) {
    const float* W = (const float*)Wpk;
    RTFS_TERMS_DISPATCH(terms, unfold_gemm_impl<1>(G, gamma, beta, W, U0, B, T2, dim, variant, stream), unfold_gemm_impl<3>(G, gamma, beta, W, U0, B, T2, dim, variant, stream),
                        unfold_gemm_impl<6>(G, gamma, beta, W, U0, B, T2, dim, variant, stream));
}

// H3: [S][L][64] -> G[pos] += convT(H3)[pos] + bias  (in place on G).  Wt: [64][512], k index = k'*64 + j, k' = 7-k.
int rtfs_dp_convt_fwd(const float* H3, const float* Wt, const float* bias, float* G, int B, int T2, int dim, void* stream) {
    return convt_impl<0>(H3, Wt, bias, G, B, T2, dim, stream);
}
// variant: 0 = the library's choice (the fast-FIR kernel at large batch), 1 = the direct 8-tap forms (weight-stationary / LDS-staged: the same bits as each other)
int rtfs_dp_convt_fwd_form(const float* H3, const float* Wt, const float* bias, float* G, int B, int T2, int dim, int variant, void* stream) {
    if (variant < 0 || variant > 1) return RTFS_EINVAL;
    return convt_impl<0>(H3, Wt, bias, G, B, T2, dim, stream, variant);
}
int rtfs_dp_convt_fwd_bf16(const float* H3, const void* Wpk, const float* bias, float* G, int B, int T2, int dim, int terms, void* stream) {
    const float* W = (const float*)Wpk;
    RTFS_TERMS_DISPATCH(terms, convt_impl<1>(H3, W, bias, G, B, T2, dim, stream), convt_impl<3>(H3, W, bias, G, B, T2, dim, stream), convt_impl<6>(H3, W, bias, G, B, T2, dim, stream));
}
// Gout = Gin + convT(H3) + bias: the out-of-place form (training step: Gin is kept for the adjoint; no copy of G in front of an in-place launch).  Gin == Gout
// is the in-place call.  The fp32 fast-FIR kernel reads Gin directly; every other form copies Gin to Gout on the stream and runs in place.
int rtfs_dp_convt_fwd_to(const float* H3, const float* Wt, const float* bias, const float* Gin, float* Gout, int B, int T2, int dim, void* stream) {
    return convt_impl<0>(H3, Wt, bias, Gout, B, T2, dim, stream, 0, Gin);
}
int rtfs_dp_convt_fwd_to_bf16(const float* H3, const void* Wpk, const float* bias, const float* Gin, float* Gout, int B, int T2, int dim, int terms, void* stream) {
    const float* W = (const float*)Wpk;
    RTFS_TERMS_DISPATCH(terms, convt_impl<1>(H3, W, bias, Gout, B, T2, dim, stream, 0, Gin), convt_impl<3>(H3, W, bias, Gout, B, T2, dim, stream, 0, Gin),
                        convt_impl<6>(H3, W, bias, Gout, B, T2, dim, stream, 0, Gin));
}

// km = 4: U [S][L][64][4];  km = 3: U [S][L][3][64] and X [S][L][64].  wc, bias: [2][64] (forget | reset).  H: [S][L][64].
int rtfs_sru_scan_fwd(const float* U, const float* X, const float* wc, const float* bias, float scale_x, float* H, int S, int L, int km,
                      void* stream) {
    if (S <= 0 || L <= 0) return RTFS_EINVAL;
    dim3 grid((S + 3) / 4);
    if (km == 4)
        hipLaunchKernelGGL((sru_scan_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, U, X, wc, bias, scale_x, H, S, L);
    else if (km == 3)
        hipLaunchKernelGGL((sru_scan_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, U, X, wc, bias, scale_x, H, S, L);
    else
        return RTFS_EINVAL;
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

// SRU layers 1-3 with the input projection fused: Hprev, Hout [S][L][64]; Wt [192][64], row = m*64 + dir*32 + j (k contiguous);
// Training (both or neither): Cout cell states [S][L][64], Uout pre-activations [S][L][3][64] as rtfs_sru_scan_bwd reads them.
// Replaces rtfs_gemm_rows_fwd(64 -> 192) + rtfs_sru_scan_fwd / rtfs_sru_scan_train_fwd (km = 3).
// form (inference only): 0 = the library's choice by S; 1 = one wave per sequence (sru_layer_kernel), 2 = one wave per (sequence, direction), 3 = one
// workgroup per (sequence, direction) with the gates split over its waves - all three give the same bits
int rtfs_sru_layer_fwd_form(const float* Hprev, const float* Wt, const float* wc, const float* bias, float scale_x, float* Hout, float* Cout_or_null,
                            float* Uout_or_null, int S, int L, int form, void* stream) {
    if (S <= 0 || L <= 0 || Hprev == Hout || (Cout_or_null == nullptr) != (Uout_or_null == nullptr) || form < 0 || form > 3 || (Cout_or_null && form > 1))
        return RTFS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define SRU_L(SAVE, NWV) hipLaunchKernelGGL((sru_layer_kernel<SAVE, 0, NWV>), dim3((S + NWV - 1) / NWV), dim3(NWV * 64), 0, st, Hprev, Wt, wc, bias, scale_x, Hout, Cout_or_null, Uout_or_null, S, L)
    if (form == 0 && !Cout_or_null) form = S < kSruSplitBelow ? 3 : (S < 2048 ? 2 : 1);
    if (form == 3) {
        hipLaunchKernelGGL(sru_layer_split_kernel, dim3(2 * S), dim3(256), 0, st, Hprev, Wt, wc, bias, scale_x, Hout, S, L);
    } else if (form == 2) {  // below 2048 sequences: one wave per (sequence, direction) (same-box A/B: batch 8 7.45 -> 7.23 ms, batch 16 equal; 4096: batch 32 slower)
        hipLaunchKernelGGL(sru_layer_dir_kernel, dim3((2 * S + 3) / 4), dim3(256), 0, st, Hprev, Wt, wc, bias, scale_x, Hout, S, L);
    } else if (S >= 2048) {
        if (Cout_or_null) SRU_L(true, 8); else SRU_L(false, 8);
    } else {
        if (Cout_or_null) SRU_L(true, 4); else SRU_L(false, 4);
    }
#undef SRU_L
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_sru_layer_fwd(const float* Hprev, const float* Wt, const float* wc, const float* bias, float scale_x, float* Hout, float* Cout_or_null,
                       float* Uout_or_null, int S, int L, void* stream) {
    return rtfs_sru_layer_fwd_form(Hprev, Wt, wc, bias, scale_x, Hout, Cout_or_null, Uout_or_null, S, L, 0, stream);
}

// bf16 / split-bf16 variant of rtfs_sru_layer_fwd: Wt is the PLAIN fp32 weight (packed in the kernel after the gate scaling)
int rtfs_sru_layer_fwd_bf16(const float* Hprev, const float* Wt, const float* wc, const float* bias, float scale_x, float* Hout, float* Cout_or_null,
                            float* Uout_or_null, int S, int L, int terms, void* stream) {
    if (S <= 0 || L <= 0 || Hprev == Hout || (terms != 1 && terms != 3 && terms != 6) || (Cout_or_null == nullptr) != (Uout_or_null == nullptr)) return RTFS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define SRU_L(SAVE, NTV, NWV) hipLaunchKernelGGL((sru_layer_kernel<SAVE, NTV, NWV>), dim3((S + NWV - 1) / NWV), dim3(NWV * 64), 0, st, Hprev, Wt, wc, bias, scale_x, Hout, Cout_or_null, Uout_or_null, S, L)
#define SRU_LW(SAVE, NTV) do { if (S >= 2048) SRU_L(SAVE, NTV, 8); else SRU_L(SAVE, NTV, 4); } while (0)
    if (Cout_or_null) {
        if (terms == 1) SRU_LW(true, 1); else if (terms == 3) SRU_LW(true, 3); else SRU_LW(true, 6);
    } else {
        if (terms == 1) SRU_LW(false, 1); else if (terms == 3) SRU_LW(false, 3); else SRU_LW(false, 6);
    }
#undef SRU_LW
#undef SRU_L
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
