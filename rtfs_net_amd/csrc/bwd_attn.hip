// Backward of the TF self-attention stage (MultiHeadSelfAttention2D, layers/attention.py:149-189).
//
//   rtfs_attn_out_norm_bwd   adjoint of (PReLU -> LN4D over (64,F) -> + residual) of attn_concat_proj: dOut -> dYpre, dgamma/dbeta/dslope
//   rtfs_attn_qkv_norm_bwd   adjoint of the 12 x (PReLU -> LN4D over (c,F)) of the Q/K/V projections: dQ,dK,dV -> dYpre96, dgamma/dbeta/dslope
//   rtfs_attn_core_bwd       flash-style adjoint of softmax(QK^T/16) V: recomputes P from the saved log-sum-exp;
//                            one kernel per query tile (dQ, D = rowsum(dO*O)), one per key tile (dK, dV)
//   rtfs_transpose_tok       [tok][64][64] per-token transpose (channels-last <-> [c][f])
// The linear maps themselves (1x1 convs) use rtfs_gemm_rows / rtfs_wgrad / rtfs_colsum_add on channels-last rows.
#include "common.h"

namespace rtfs {

constexpr int kHeadsB = 4;

__device__ __forceinline__ float block_sum(float v, float* red /*4*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---- attn_concat_proj tail ------------------------------------------------------------------------------------------------
// forward per token: z = prelu(ypre); out = (z - mean)*rstd*gamma[f][c] + beta[f][c] + res   (4096 values per token)
// Thread owns elements i = tid + 256k (k < 16) of the [f][c] tile for every token of its chunk, so dgamma/dbeta accumulate
// in registers and are committed once per workgroup.
__global__ __launch_bounds__(256) void attn_out_norm_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ Ypre, float slope,
                                                                const float* __restrict__ gamma_fc, float* __restrict__ dYpre,
                                                                float* __restrict__ scr, int ntok, int tok_per_wg) {
    __shared__ float red[4];
    float ag[16], ab[16], gam[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) ag[k] = ab[k] = 0.f, gam[k] = gamma_fc[threadIdx.x + 256 * k];
    float asl = 0.f;
    const int t0 = blockIdx.x * tok_per_wg, t1 = min(ntok, t0 + tok_per_wg);
    for (int tok = t0; tok < t1; ++tok) {
        const size_t base = (size_t)tok * 4096;
        float yp[16], z[16], g[16];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            yp[k] = Ypre[base + threadIdx.x + 256 * k];
            g[k] = dOut[base + threadIdx.x + 256 * k];
            z[k] = prelu(yp[k], slope);
            s += z[k];
        }
        const float mean = block_sum(s, red) * (1.f / 4096.f);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            z[k] -= mean;
            q = fmaf(z[k], z[k], q);
        }
        const float rstd = 1.0f / sqrtf(block_sum(q, red) * (1.f / 4096.f) + kEps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            z[k] *= rstd;  // zhat
            const float a = g[k] * gam[k];
            s1 += a;
            s2 = fmaf(a, z[k], s2);
            ag[k] = fmaf(g[k], z[k], ag[k]);
            ab[k] += g[k];
        }
        s1 = block_sum(s1, red) * (1.f / 4096.f);
        s2 = block_sum(s2, red) * (1.f / 4096.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float dz = (g[k] * gam[k] - s1 - z[k] * s2) * rstd;
            if (yp[k] <= 0.f) asl = fmaf(dz, yp[k], asl);
            dYpre[base + threadIdx.x + 256 * k] = yp[k] > 0.f ? dz : dz * slope;
        }
    }
    float* mine = spread_copy(scr, blockIdx.x);  // [dgamma 4096 | dbeta 4096 | dslope]
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        atomicAdd(mine + threadIdx.x + 256 * k, ag[k]);
        atomicAdd(mine + 4096 + threadIdx.x + 256 * k, ab[k]);
    }
    asl = block_sum(asl, red);
    if (threadIdx.x == 0) atomicAdd(mine + 8192, asl);
}

// ---- Q/K/V projection tails -----------------------------------------------------------------------------------------------
// Ypre96: [tok][64 f][96] (columns: Q h*4+e | 16 + K h*4+e | 32 + V h*16+c).  Group g (12 per token) = module: 64 f x ncol columns.
// dQ,dK: [B][4][T2][256] (e*64+f), dV: [B][4][T2][1024] (c*64+f).  gamma arrays as in the forward: gq,gk [4][256], gv [4][1024].
// Outputs: dYpre96 (same layout), dgq/dbq/dgk/dbk [4][256], dgv/dbv [4][1024], dslope[12] (module order Q0..3,K0..3,V0..3).
// Element ownership: i = tid + 256k over the 6144 elements in (column n, f) order: n = i / 64, f = i % 64  (k < 24).
__global__ __launch_bounds__(256) void attn_qkv_norm_bwd_kernel(const float* __restrict__ dQ, const float* __restrict__ dK, const float* __restrict__ dV,
                                                                const float* __restrict__ Ypre, const float* __restrict__ slope,
                                                                const float* __restrict__ gq, const float* __restrict__ gk, const float* __restrict__ gv,
                                                                float* __restrict__ dYpre, float* __restrict__ scr, int BT, int T2, int tok_per_wg) {
    constexpr int LDY = 97;
    __shared__ float Yr[64 * LDY];       // raw ypre staged in memory order ([f][96]), read back in (n, f) ownership order
    __shared__ float Ds[64 * LDY];       // dYpre staged in ownership order, written out in memory order
    __shared__ float part[4][48];        // per-wave partial sums: 12 modules x (sum z, sum z^2, sum a, sum a z)
    __shared__ float stat[12][4];        // mean, rstd, s1, s2 per module
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // element ownership: i = tid + 256k -> column n = wave + 4k, f = lane; the module of element k is a function of k alone
    // (k for the Q/K modules, 8 + (k-8)/4 for V), so everything per-module below is a static register index.
    float gam[24], ag[24], ab[24], asl[12];
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        const int n = w + 4 * k, g = k < 8 ? k : 8 + ((k - 8) >> 2);
        const int col0 = g < 8 ? g * 4 : 32 + (g - 8) * 16;
        const int e = (n - col0) * 64 + lane;
        gam[k] = g < 4 ? gq[g * 256 + e] : (g < 8 ? gk[(g - 4) * 256 + e] : gv[(g - 8) * 1024 + e]);
        ag[k] = ab[k] = 0.f;
    }
#pragma unroll
    for (int g = 0; g < 12; ++g) asl[g] = 0.f;

    const int t0 = blockIdx.x * tok_per_wg, t1 = min(BT, t0 + tok_per_wg);
    for (int bt = t0; bt < t1; ++bt) {
        const int b = bt / T2, t = bt % T2;
        float yraw[24], gN[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) yraw[k] = Ypre[(size_t)bt * 6144 + threadIdx.x + 256 * k];
#pragma unroll
        for (int k = 0; k < 24; ++k) {  // incoming gradients, already in ownership order (coalesced in f)
            const int n = w + 4 * k;
            if (k < 4) gN[k] = dQ[(((size_t)b * kHeadsB + (n >> 2)) * T2 + t) * 256 + (n & 3) * 64 + lane];
            else if (k < 8) gN[k] = dK[(((size_t)b * kHeadsB + ((n - 16) >> 2)) * T2 + t) * 256 + ((n - 16) & 3) * 64 + lane];
            else gN[k] = dV[(((size_t)b * kHeadsB + ((n - 32) >> 4)) * T2 + t) * 1024 + ((n - 32) & 15) * 64 + lane];
        }
        __syncthreads();  // previous token's Ds / Yr fully consumed
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int i = threadIdx.x + 256 * k, f = i / 96, n = i - f * 96;
            Yr[f * LDY + n] = yraw[k];
        }
        __syncthreads();
        float yp[24], z[24], ps[48];
#pragma unroll
        for (int q = 0; q < 48; ++q) ps[q] = 0.f;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int n = w + 4 * k, g = k < 8 ? k : 8 + ((k - 8) >> 2);
            yp[k] = Yr[lane * LDY + n];
            z[k] = prelu(yp[k], slope[n]);  // n is wave-uniform: scalar load
            const float a = gN[k] * gam[k];
            ps[4 * g] += z[k], ps[4 * g + 1] = fmaf(z[k], z[k], ps[4 * g + 1]), ps[4 * g + 2] += a, ps[4 * g + 3] = fmaf(a, z[k], ps[4 * g + 3]);
        }
#pragma unroll
        for (int q = 0; q < 48; ++q) {
            const float v = wave_sum(ps[q]);
            if (lane == 0) part[w][q] = v;
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            const int g = threadIdx.x;
            const float cnt = g < 8 ? 256.f : 1024.f;
            float sm[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sm[q] = part[0][4 * g + q] + part[1][4 * g + q] + part[2][4 * g + q] + part[3][4 * g + q];
            const float mean = sm[0] / cnt;
            const float var = fmaxf(sm[1] / cnt - mean * mean, 0.f);
            const float rstd = 1.0f / sqrtf(var + kEps);
            stat[g][0] = mean, stat[g][1] = rstd;
            stat[g][2] = sm[2] / cnt;                              // s1 = mean(a)
            stat[g][3] = (sm[3] - mean * sm[2]) * rstd / cnt;      // s2 = mean(a * zhat)
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int n = w + 4 * k, g = k < 8 ? k : 8 + ((k - 8) >> 2);
            const float mean = stat[g][0], rstd = stat[g][1];
            const float zh = (z[k] - mean) * rstd;
            ag[k] = fmaf(gN[k], zh, ag[k]);
            ab[k] += gN[k];
            const float dz = (gN[k] * gam[k] - stat[g][2] - zh * stat[g][3]) * rstd;
            if (yp[k] <= 0.f) asl[g] = fmaf(dz, yp[k], asl[g]);
            Ds[lane * LDY + n] = yp[k] > 0.f ? dz : dz * slope[n];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int i = threadIdx.x + 256 * k, f = i / 96, n = i - f * 96;
            dYpre[(size_t)bt * 6144 + i] = Ds[f * LDY + n];
        }
    }
    // commit parameter gradients into this workgroup's scratch copy: [dgq 1024 | dbq 1024 | dgk 1024 | dbk 1024 | dgv 4096 | dbv 4096 | dslope 12]
    float* mine = spread_copy(scr, blockIdx.x);
    float *dgq = mine, *dbq = mine + 1024, *dgk = mine + 2048, *dbk = mine + 3072, *dgv = mine + 4096, *dbv = mine + 8192, *dslope = mine + 12288;
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        const int n = w + 4 * k, g = k < 8 ? k : 8 + ((k - 8) >> 2);
        const int col0 = g < 8 ? g * 4 : 32 + (g - 8) * 16;
        const int e = (n - col0) * 64 + lane;
        float *pg, *pb;
        if (g < 4) pg = dgq + g * 256 + e, pb = dbq + g * 256 + e;
        else if (g < 8) pg = dgk + (g - 4) * 256 + e, pb = dbk + (g - 4) * 256 + e;
        else pg = dgv + (g - 8) * 1024 + e, pb = dbv + (g - 8) * 1024 + e;
        atomicAdd(pg, ag[k]);
        atomicAdd(pb, ab[k]);
    }
    __shared__ float red[4];
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        const float v = block_sum(asl[g], red);
        if (threadIdx.x == 0) atomicAdd(dslope + g, v);
    }
}

// ---- core, query-tile kernel: dQ and D --------------------------------------------------------------------------------------
// grid (ceil(T2/32), 4, B).  dO / O are read in the O layout [B][T2][64 ch][64 f] (head h: contiguous 1024 at offset h*1024).
template <int MAXKT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const float* __restrict__ Q, const float* __restrict__ Kx, const float* __restrict__ V,
                                                             const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ LSE,
                                                             float* __restrict__ dQ, float* __restrict__ Dout, int T2) {
    constexpr int LDP = MAXKT * 32 + 4;
    __shared__ __attribute__((aligned(16))) float Ps[32 * LDP];   // P, then dS
    __shared__ float Ds[32], Ls[32];
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int q0 = qt * 32, NT = (T2 + 31) / 32;
    const size_t headoff = ((size_t)b * kHeadsB + h) * T2;
    const float* Qg = Q + headoff * 256;
    const float* Kg = Kx + headoff * 256;
    const float* Vg = V + headoff * 1024;
    auto orow = [&](const float* base, int t) { return base + ((size_t)b * T2 + min(t, T2 - 1)) * 4096 + h * 1024; };

    // D_i = sum_e dO[i][e] * O[i][e]; 8 rows per wave
    for (int rr = 0; rr < 8; ++rr) {
        const int t = q0 + w * 8 + rr;
        const float* po = orow(O, t);
        const float* pd = orow(dO, t);
        float s = 0.f;
        for (int e = lane * 4; e < 1024; e += 256) {
            const float4 a = ld4(po + e), d = ld4(pd + e);
            s += a.x * d.x + a.y * d.y + a.z * d.z + a.w * d.w;
        }
        s = wave_sum(s);
        if (lane == 0) {
            Ds[w * 8 + rr] = s;
            Ls[w * 8 + rr] = LSE[headoff + min(t, T2 - 1)];
            if (t < T2) Dout[headoff + t] = s;
        }
    }
    __syncthreads();
    // P = exp(S - LSE), dP = dO V^T, dS = P (dP - D) / 16; wave w owns key tiles w, w+4, ...
    // Keys are walked in blocks of MAXKT tiles (the LDS dS tile): P is recomputed from the FINAL log-sum-exp, so the blocks are independent
    // and dQ simply accumulates over them - no length limit (round 3; the reference has none, attention.py:171-173)
    const float* qrow = Qg + (size_t)min(q0 + i, T2 - 1) * 256 + 4 * kh;
    const float* drow = orow(dO, q0 + i) + 4 * kh;
    floatx16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll 1
    for (int kb0 = 0; kb0 < NT; kb0 += MAXKT) {
    const int kb1 = min(NT, kb0 + MAXKT);
    if (kb0) __syncthreads();  // the previous block's dS tile has been consumed
    for (int kt = kb0 + w; kt < kb1; kt += 4) {
        const int key = min(kt * 32 + i, T2 - 1);
        const float* krow = Kg + (size_t)key * 256 + 4 * kh;
        const float* vrow = Vg + (size_t)key * 1024 + 4 * kh;
        floatx16 sa, pa;
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[r] = pa[r] = 0.f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) {
            const float4 a = ld4(qrow + 8 * q), kb = ld4(krow + 8 * q);
            sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kb.x, sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kb.y, sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kb.z, sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kb.w, sa, 0, 0, 0);
        }
#pragma unroll 8
        for (int q = 0; q < 128; ++q) {
            const float4 a = ld4(drow + 8 * q), vb = ld4(vrow + 8 * q);
            pa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, vb.x, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, vb.y, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, vb.z, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, vb.w, pa, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r);
            const bool kvalid = kt * 32 + i < T2;
            const float p = kvalid ? __expf(sa[r] * 0.0625f - Ls[row]) : 0.f;
            Ps[row * LDP + (kt - kb0) * 32 + i] = p * (pa[r] - Ds[row]) * 0.0625f;
        }
    }
    __syncthreads();
    // dQ[32 x 256] += dS K; wave w owns features [64w, 64w+64)
    const float* pa_ = Ps + i * LDP + 4 * kh;
    for (int kq = 0; kq < (kb1 - kb0) * 4; ++kq) {
        const float4 p = ld4(pa_ + 8 * kq);
        const int key = kb0 * 32 + 8 * kq + 4 * kh;
        float kb[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* kr = Kg + (size_t)min(key + r, T2 - 1) * 256 + w * 64 + i;
            kb[0][r] = kr[0], kb[1][r] = kr[32];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.x, kb[n][0], acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.y, kb[n][1], acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.z, kb[n][2], acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.w, kb[n][3], acc[n], 0, 0, 0);
        }
    }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = q0 + acc_row(r);
            if (t < T2) dQ[(headoff + t) * 256 + w * 64 + n * 32 + i] = acc[n][r];
        }
}

// ---- core, key-tile kernel: dK and dV -------------------------------------------------------------------------------------
// grid (ceil(T2/32), 4, B): one 32-key tile.  Builds P^T and dS^T for all queries in LDS ([32 keys][queries]), then
// dV = P^T dO (wave w owns 256 of the 1024 features, two passes), dK = dS^T Q (wave w owns 64 of the 256 features).
template <int MAXKT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const float* __restrict__ Q, const float* __restrict__ Kx, const float* __restrict__ V,
                                                              const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Din,
                                                              float* __restrict__ dK, float* __restrict__ dV, int T2) {
    constexpr int LDP = MAXKT * 32 + 4;
    __shared__ __attribute__((aligned(16))) float Pt[32 * LDP];
    __shared__ __attribute__((aligned(16))) float St[32 * LDP];
    const int ktile = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int k0 = ktile * 32, NT = (T2 + 31) / 32;
    const size_t headoff = ((size_t)b * kHeadsB + h) * T2;
    const float* Qg = Q + headoff * 256;
    const float* Kg = Kx + headoff * 256;
    const float* Vg = V + headoff * 1024;
    auto orow = [&](int t) { return dO + ((size_t)b * T2 + min(t, T2 - 1)) * 4096 + h * 1024; };

    // first operand (rows) = keys of this tile, second operand (lanes) = queries of tile qt
    const float* krow = Kg + (size_t)min(k0 + i, T2 - 1) * 256 + 4 * kh;
    const float* vrow = Vg + (size_t)min(k0 + i, T2 - 1) * 1024 + 4 * kh;
    // Queries are walked in blocks of MAXKT tiles (the two LDS tiles); past the first block the dV / dK accumulators start from the partial
    // sums the previous block left in global memory (one block for T2 <= 32 MAXKT: then nothing is re-read) - no length limit
#pragma unroll 1
    for (int qb0 = 0; qb0 < NT; qb0 += MAXKT) {
    const int qb1 = min(NT, qb0 + MAXKT);
    if (qb0) __syncthreads();  // the previous block's tiles have been consumed
    for (int qt = qb0 + w; qt < qb1; qt += 4) {
        const int tq = min(qt * 32 + i, T2 - 1);
        const float* qrow = Qg + (size_t)tq * 256 + 4 * kh;
        const float* drow = orow(qt * 32 + i) + 4 * kh;
        floatx16 sa, pa;
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[r] = pa[r] = 0.f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) {
            const float4 a = ld4(krow + 8 * q), qb = ld4(qrow + 8 * q);
            sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qb.x, sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qb.y, sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qb.z, sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qb.w, sa, 0, 0, 0);
        }
#pragma unroll 8
        for (int q = 0; q < 128; ++q) {
            const float4 a = ld4(vrow + 8 * q), db = ld4(drow + 8 * q);
            pa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, db.x, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, db.y, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, db.z, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, db.w, pa, 0, 0, 0);
        }
        const bool qvalid = qt * 32 + i < T2;
        const float lse = LSE[headoff + tq], dd = Din[headoff + tq];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r);  // key inside the tile
            const bool kvalid = k0 + row < T2;
            const float p = (qvalid && kvalid) ? __expf(sa[r] * 0.0625f - lse) : 0.f;
            Pt[row * LDP + (qt - qb0) * 32 + i] = p;
            St[row * LDP + (qt - qb0) * 32 + i] = p * (pa[r] - dd) * 0.0625f;
        }
    }
    __syncthreads();
    // dV = P^T dO
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const int n0 = w * 256 + pass * 128;
        floatx16 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + acc_row(r);
                acc[n][r] = (qb0 && key < T2) ? dV[(headoff + key) * 1024 + n0 + n * 32 + i] : 0.f;
            }
        const float* pa_ = Pt + i * LDP + 4 * kh;
        for (int qq = 0; qq < (qb1 - qb0) * 4; ++qq) {
            const float4 p = ld4(pa_ + 8 * qq);
            const int tq = qb0 * 32 + 8 * qq + 4 * kh;
            float vb[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* dr = orow(tq + r) + n0 + i;
#pragma unroll
                for (int n = 0; n < 4; ++n) vb[n][r] = dr[n * 32];
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.x, vb[n][0], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.y, vb[n][1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.z, vb[n][2], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.w, vb[n][3], acc[n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + acc_row(r);
                if (key < T2) dV[(headoff + key) * 1024 + n0 + n * 32 + i] = acc[n][r];
            }
    }
    // dK = dS^T Q
    floatx16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + acc_row(r);
            acc[n][r] = (qb0 && key < T2) ? dK[(headoff + key) * 256 + w * 64 + n * 32 + i] : 0.f;
        }
    const float* sa_ = St + i * LDP + 4 * kh;
    for (int qq = 0; qq < (qb1 - qb0) * 4; ++qq) {
        const float4 p = ld4(sa_ + 8 * qq);
        const int tq = qb0 * 32 + 8 * qq + 4 * kh;
        float qb[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* qr = Qg + (size_t)min(tq + r, T2 - 1) * 256 + w * 64 + i;
            qb[0][r] = qr[0], qb[1][r] = qr[32];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.x, qb[n][0], acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.y, qb[n][1], acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.z, qb[n][2], acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p.w, qb[n][3], acc[n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + acc_row(r);
            if (key < T2) dK[(headoff + key) * 256 + w * 64 + n * 32 + i] = acc[n][r];
        }
    }
}

// per-token 64x64 transpose: out[tok][j][i] = in[tok][i][j]
__global__ __launch_bounds__(256) void transpose_tok_kernel(const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float tile[64 * 65];
    const size_t base = (size_t)blockIdx.x * 4096;
    for (int i = threadIdx.x; i < 4096; i += 256) tile[(i >> 6) * 65 + (i & 63)] = in[base + i];
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256) out[base + i] = tile[(i & 63) * 65 + (i >> 6)];
}

}  // namespace rtfs

using namespace rtfs;

extern "C" {

// dOut, Ypre, dYpre: [ntok][64 f][64 c]; gamma_fc/dgamma_fc/dbeta_fc: [64 f][64 c]; dslope: [1]
int rtfs_attn_out_norm_bwd(const float* dOut, const float* Ypre, float slope, const float* gamma_fc, float* dYpre, float* dgamma_fc, float* dbeta_fc,
                           float* dslope, int ntok, void* stream) {
    if (ntok <= 0) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    const int per = 4;
    hipLaunchKernelGGL(attn_out_norm_bwd_kernel, dim3((ntok + per - 1) / per), dim3(256), 0, (hipStream_t)stream, dOut, Ypre, slope, gamma_fc, dYpre,
                       scr, ntok, per);
    RTFS_LAUNCH_CHECK();
    return spread_finish(scr, SpreadOut{{dgamma_fc, dbeta_fc, dslope}, {4096, 4096, 1}}, (hipStream_t)stream);
}

int rtfs_attn_qkv_norm_bwd(const float* dQ, const float* dK, const float* dV, const float* Ypre, const float* slope, const float* gq, const float* gk,
                           const float* gv, float* dYpre, float* dgq, float* dbq, float* dgk, float* dbk, float* dgv, float* dbv, float* dslope, int B,
                           int T2, void* stream) {
    if (B <= 0 || T2 <= 0) return RTFS_EINVAL;
    float* scr = spread_scratch();
    if (!scr) return RTFS_ELAUNCH;
    const int per = 4, BT = B * T2;
    hipLaunchKernelGGL(attn_qkv_norm_bwd_kernel, dim3((BT + per - 1) / per), dim3(256), 0, (hipStream_t)stream, dQ, dK, dV, Ypre, slope, gq, gk, gv, dYpre,
                       scr, BT, T2, per);
    RTFS_LAUNCH_CHECK();
    return spread_finish(scr, SpreadOut{{dgq, dbq, dgk, dbk, dgv, dbv, dslope}, {1024, 1024, 1024, 1024, 4096, 4096, 12}}, (hipStream_t)stream);
}

// Q,K,dQ,dK: [B][4][T2][256]; V,dV: [B][4][T2][1024]; O,dO: [B][T2][64][64] (O layout); LSE, Dws: [B][4][T2]
int rtfs_attn_core_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE, float* Dws, float* dQ,
                       float* dK, float* dV, int B, int T2, void* stream) {
    if (B <= 0 || T2 <= 0) return RTFS_EINVAL;  // (more than 512 compressed frames: the kernels walk keys / queries in blocks of 512)
    dim3 grid((T2 + 31) / 32, kHeadsB, B);
    hipStream_t st = (hipStream_t)stream;
#define CORE_BWD(M)                                                                                                     \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<M>), grid, dim3(256), 0, st, Q, K, V, O, dO, LSE, dQ, Dws, T2);              \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<M>), grid, dim3(256), 0, st, Q, K, V, dO, LSE, (const float*)Dws, dK, dV, T2);
    if (T2 <= 128) { CORE_BWD(4) } else if (T2 <= 256) { CORE_BWD(8) } else { CORE_BWD(16) }
#undef CORE_BWD
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

int rtfs_transpose_tok(const float* in, float* out, int ntok, void* stream) {
    if (ntok <= 0) return RTFS_EINVAL;
    hipLaunchKernelGGL(transpose_tok_kernel, dim3(ntok), dim3(256), 0, (hipStream_t)stream, in, out);
    RTFS_LAUNCH_CHECK();
    return RTFS_OK;
}

}  // extern "C"
