// Train-mode forward pieces and the model's backward pass: what `loss.backward()` does to
// MultiModal_Fusion_Attn_Iter (/root/reference/model/imf_vad.py:109-161) in /root/reference/train/ucf_train.py:103 and
// train/xd_train.py:78, as hand-written fp32 kernels.  The reference has no backward code of its own (autograd derives it), so every
// kernel below cites the forward line it differentiates.
//
//   iefvad_bgemm_f32_kernel      one batched fp32-MFMA contraction C[i,j] = alpha * sum_k A(i,k) B(j,k) (+ R) (gated by G > 0) for
//                                operands stored with either index contiguous.  It serves every product of the backward pass:
//                                  dX = dY W          (A k-contiguous, B j-contiguous)   Linear / in_proj / out_proj input gradients
//                                  dW = dY^T X        (both row-major over the contraction: deterministic split-K over the rows)
//                                  S = q k^T, dP = dA v^T   (both k-contiguous), A = P v, dQ = dS k, dK = dS^T q, dV = P^T dA
//                                  (batched over (chunk, head) with two-level strides)
//   iefvad_softmax_dropout_kernel  softmax over the 256 keys + attention dropout (imf_vad.py:70: nn.MultiheadAttention(dropout=0.1),
//                                  active in train()); keeps P with the mask in its sign bits
//   iefvad_softmax_bwd_kernel      dS = P_d .* dP_d - P * rowsum(P_d .* dP_d)
//   iefvad_scorer_bwd_kernel       classifier (imf_vad.py:150):  dz = dlogit w_c (+ d fused), partial sums for dw_c, db_c
//   iefvad_fusion_bwd_kernel       precision weights + fusion (imf_vad.py:130-144), literal expression graph
//   iefvad_layernorm_bwd_kernel    LayerNorm(768) (imf_vad.py:116-117,122-123), partial sums for d gamma, d beta
//   iefvad_colsum_kernel / iefvad_reduce_parts_kernel   bias gradients and the fixed-order reduction of every partial sum
// No atomics anywhere: every reduction over rows goes through per-workgroup (or per-split) partials that one kernel adds up in
// index order, so gradients are bit-reproducible from run to run.
#pragma once
#include "common.h"
#include "rowops.h"

// ------------------------------------------------------------------------------------------------------------------------
// batched fp32 GEMM on v_mfma_f32_32x32x2_f32
// ------------------------------------------------------------------------------------------------------------------------
// C[z][i, j] = alpha * sum_k A(i, k) * B(j, k)  (+ R[i, j])  (zeroed where G[i, j] <= 0)
//   A_KC:  A(i, k) = A[i * lda + k]   (k contiguous)        else A(i, k) = A[k * lda + i]   (i contiguous)
//   B_KC:  B(j, k) = B[j * ldb + k]                          else B(j, k) = B[k * ldb + j]
// Block tile 128 x BN (BN = 128 or 96: a head is 96 wide), BK = 16, 4 waves, wave w owns rows 32 w .. 32 w + 31 and all BN
// columns (BN / 32 accumulators).  Both tiles sit in LDS k-major ([k][i]): the MFMA fragment of lane (i, h) is element
// (i, k = 2 s + h), i.e. 32 consecutive dwords per half-wave -- conflict-free for ds_read_b32 whatever the pitch (lanes l and
// l + 32 never conflict).  An operand that is contiguous along i is staged with 16-byte loads and stores; one that is contiguous
// along k is loaded 16 bytes along k and transposed on the way into LDS (four ds_write_b32; pitch 130 spreads the 32 lanes of
// a half over 32 banks).  Global loads of tile t + 1 are issued before the MFMAs of tile t and written to the other LDS buffer
// after them: one barrier per tile.
struct BgemmArgs {
    const float* A; const float* B; float* C;
    const float* R;          // nullable: added (same layout as C)
    const float* G;          // nullable: gate, C = G > 0 ? C : 0 (ReLU backward on the saved activation, imf_vad.py:100)
    int M, N, K;
    int lda, ldb, ldc;
    long long a1, a2, b1, b2, c1, c2;     // batch z = z1 * nz2 + z2: operand offsets z1 * x1 + z2 * x2 (floats)
    int nz2;
    float alpha;
    // epilogue extras (zero / null in every product of the backward pass): v = act(alpha * acc + bias[j] + R[i, j])
    const float* bias;       // nullable: [N], broadcast over the rows (and over the batch)
    int act;                 // 0 none; 1 QuickGELU v * sigmoid(1.702 v) (/root/reference/model/module.py:15-17); 2 ELU (layers.py:43-44)
    // A holds the train forward's sign-carrying probabilities (attention_split.h): A(i, k) = sign ? 0 : stored * a_drop, i.e.
    // dropout(P), formed between the tile's load and its LDS store; 0 = A is taken as it is
    float a_drop;
};

#define BG_BM 128
#define BG_BK 16

template <bool KC, int ROWS>
struct BgemmTile {                      // staging of one operand tile: ROWS (i or j) x 16 (k)
    static constexpr int PITCH = KC ? ROWS + 2 : ROWS + 4;
    static constexpr int NQ = ROWS * 4;                               // 16-byte pieces of the tile
    static constexpr int NV = (NQ + 255) / 256;                       // per thread
    // A 96-row tile has 384 pieces for 256 threads: the surplus threads of the second round repeat the LAST piece (same load, same
    // value to the same LDS address) instead of being masked off -- a masked load would cost the prefetch its asynchrony (the
    // compiler waits for it on the spot to merge it with the old register value)
    static __device__ __forceinline__ int piece(int t, int u) { const int idx = t + 256 * u; return idx < NQ ? idx : NQ - 1; }
    // global -> registers
    static __device__ __forceinline__ void load(const float* base, int ld, int k0, f32x4 (&r)[NV], int t) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = piece(t, u);
            if (KC) r[u] = *(const f32x4*)(base + (size_t)(idx >> 2) * ld + k0 + 4 * (idx & 3));
            else r[u] = *(const f32x4*)(base + (size_t)(k0 + idx / (ROWS / 4)) * ld + 4 * (idx % (ROWS / 4)));
        }
    }
    // registers -> LDS image [k][i]
    static __device__ __forceinline__ void store(float* s, const f32x4 (&r)[NV], int t) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = piece(t, u);
            if (KC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s[(4 * (idx & 3) + e) * PITCH + (idx >> 2)] = r[u][e];
            } else {
                *(f32x4*)(s + (idx / (ROWS / 4)) * PITCH + 4 * (idx % (ROWS / 4))) = r[u];
            }
        }
    }
};

template <bool A_KC, bool B_KC, int BN>
__global__ __launch_bounds__(256, 2) void iefvad_bgemm_f32_kernel(BgemmArgs a) {
    typedef BgemmTile<A_KC, BG_BM> TA;
    typedef BgemmTile<B_KC, BN> TB;
    constexpr int NT = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[2 * BG_BK * (TA::PITCH + TB::PITCH)];
    constexpr int ASZ = BG_BK * TA::PITCH, BSZ = BG_BK * TB::PITCH;      // buffer c: A at c * ASZ, B at 2 * ASZ + c * BSZ

    const int t = threadIdx.x, lane = t & 63, w = t >> 6, i = lane & 31, h = lane >> 5;
    // one-dimensional grid over (batch z, tile), walked XCD by XCD: each XCD owns a contiguous range of (z, tile) pairs, so the
    // tiles that share operand panels (the 6 column tiles of a row block; all 36 tiles of a split-K slice) meet in one L2
    const int ntn = a.N / BN;
    const int tiles = (a.M / BG_BM) * ntn;
    const int rid = xcd_remap(blockIdx.x, gridDim.x);
    const int z = rid / tiles, tid = rid - z * tiles;
    const int tm = tid / ntn, tn = tid - tm * ntn;
    const int m0 = tm * BG_BM, n0 = tn * BN;
    const int z1 = z / a.nz2, z2 = z - z1 * a.nz2;
    const float* Ab = a.A + z1 * a.a1 + z2 * a.a2 + (A_KC ? (size_t)m0 * a.lda : (size_t)m0);
    const float* Bb = a.B + z1 * a.b1 + z2 * a.b2 + (B_KC ? (size_t)n0 * a.ldb : (size_t)n0);
    const size_t coff = (size_t)(z1 * a.c1 + z2 * a.c2);

    f32x16 acc[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    f32x4 ra[TA::NV], rb[TB::NV];
    TA::load(Ab, a.lda, 0, ra, t);
    TB::load(Bb, a.ldb, 0, rb, t);
    if (a.a_drop != 0.f) {
#pragma unroll
        for (int u = 0; u < TA::NV; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) ra[u][e] = dropped_from_signed(ra[u][e], a.a_drop);
    }
    TA::store(smem, ra, t);
    TB::store(smem + 2 * ASZ, rb, t);
    __syncthreads();
    const int nk = a.K / BG_BK;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            TA::load(Ab, a.lda, (kt + 1) * BG_BK, ra, t);
            TB::load(Bb, a.ldb, (kt + 1) * BG_BK, rb, t);
        }
        const float* ap = smem + cur * ASZ + h * TA::PITCH + 32 * w + i;
        const float* bp = smem + 2 * ASZ + cur * BSZ + h * TB::PITCH + i;
        // fragments of k-step s + 1 are requested before the MFMAs of k-step s (pinned: hipcc otherwise sinks every read to
        // its first use and the wave waits out an LDS round trip per pair of MFMAs)
        float av[2], bv[2][NT];
        av[0] = ap[0];
#pragma unroll
        for (int b = 0; b < NT; ++b) bv[0][b] = bp[32 * b];
#pragma unroll
        for (int s = 0; s < BG_BK / 2; ++s) {
            const int c = s & 1, n = c ^ 1;
            if (s + 1 < BG_BK / 2) {
                av[n] = ap[2 * (s + 1) * TA::PITCH];
#pragma unroll
                for (int b = 0; b < NT; ++b) bv[n][b] = bp[2 * (s + 1) * TB::PITCH + 32 * b];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bv[c][b], acc[b], 0, 0, 0);
        }
        if (more) {
            if (a.a_drop != 0.f) {      // uniform
#pragma unroll
                for (int u = 0; u < TA::NV; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ra[u][e] = dropped_from_signed(ra[u][e], a.a_drop);
            }
            TA::store(smem + (cur ^ 1) * ASZ, ra, t);
            TB::store(smem + 2 * ASZ + (cur ^ 1) * BSZ, rb, t);
        }
        __syncthreads();
        cur ^= 1;
    }
    // accumulator map (32x32 tile): col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
            const size_t o = coff + (size_t)m * a.ldc + n0 + 32 * b + i;
            float v = a.alpha * acc[b][r];
            if (a.bias) v += a.bias[n0 + 32 * b + i];
            if (a.R) v += a.R[o];
            if (a.G) v = a.G[o] > 0.f ? v : 0.f;
            if (a.act == 1) v = v * (1.0f / (1.0f + expf(-1.702f * v)));
            else if (a.act == 2) v = v > 0.f ? v : expm1f(v);
            a.C[o] = v;
        }
}

// ------------------------------------------------------------------------------------------------------------------------
// attention probabilities, train mode
// ------------------------------------------------------------------------------------------------------------------------
// (the counter-based mask generator `dropout_bits` lives in common.h: the fused train-mode attention kernel draws the same bits)

// One wavefront per score row (256 keys, 4 per lane).  S holds q k^T / sqrt(96) (q is pre-scaled by the in_proj epilogue, as
// F.multi_head_attention_forward scales q before the bmm).  P = softmax(S) is written over S with the dropout mask in its SIGN bits
// (set = dropped; common.h: one tensor carries P = |stored| and dropout(P) = sign ? 0 : stored / (1 - p), as the fused launches of
// attention_split.h keep it): the products that read dropout(P) form it on the way (BgemmArgs.a_drop).
struct SoftmaxDropArgs {
    float* S;                    // [rows, 256] in: scores, out: sign-carrying P
    int drop;                    // dropout in force (p > 0 or an injected mask)
    const unsigned char* keep;   // [rows, 256] injected mask (1 = keep), nullable
    unsigned long long seed;     // RNG stream of this (layer, modality)
    float p;                     // dropout probability
    long long rows;
};
__global__ __launch_bounds__(256) void iefvad_softmax_dropout_kernel(SoftmaxDropArgs a) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    float* sp = a.S + row * 256 + 4 * lane;
    f32x4 v = *(const f32x4*)sp;
    const float mx = wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = expf(v[e] - mx); s += v[e]; }
    s = wave_sum(s);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] / s;
    if (a.drop) {
        const unsigned thr = (unsigned)((double)a.p * 16777216.0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long idx = (unsigned long long)row * 256 + 4 * lane + e;
            const bool k = a.keep ? a.keep[idx] != 0 : dropout_bits(a.seed, idx) >= thr;
            v[e] = with_sign_bit(v[e], k ? 0u : 0x80000000u);
        }
    }
    *(f32x4*)sp = v;
}

// dS = P .* (dP - D), with dP = dPd .* keep / (1 - p) and D = rowsum(P .* dP): since P .* keep / (1 - p) = Pd,
// dS = Pd .* dPd - P * rowsum(Pd .* dPd).  In place over dPd.  `Ps`: the sign-carrying P, `scale` = 1 / (1 - p) as the forward rounded it.
__global__ __launch_bounds__(256) void iefvad_softmax_bwd_kernel(const float* Ps, float scale, float* dPd, long long rows) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t o = (size_t)row * 256 + 4 * lane;
    const f32x4 ps = *(const f32x4*)(Ps + o);
    f32x4 g = *(const f32x4*)(dPd + o);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[e] = dropped_from_signed(ps[e], scale) * g[e]; d += g[e]; }
    d = wave_sum(d);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = g[e] - __builtin_fabsf(ps[e]) * d;
    *(f32x4*)(dPd + o) = g;
}

// ------------------------------------------------------------------------------------------------------------------------
// fixed-order reductions over rows
// ------------------------------------------------------------------------------------------------------------------------
// part[blockIdx.y][j] = sum over the block's rows of Y[r, j]; one thread per column (coalesced across threads), four
// independent chains per thread combined in a fixed order
__global__ __launch_bounds__(256) void iefvad_colsum_kernel(const float* Y, int ld, int rows, int ncols, int rows_per_block, float* part) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ncols) return;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += Y[(size_t)r * ld + j];
        s1 += Y[(size_t)(r + 1) * ld + j];
        s2 += Y[(size_t)(r + 2) * ld + j];
        s3 += Y[(size_t)(r + 3) * ld + j];
    }
    for (; r < r1; ++r) s0 += Y[(size_t)r * ld + j];
    part[(size_t)blockIdx.y * ncols + j] = (s0 + s1) + (s2 + s3);
}

// out[i] = alpha * sum_p part[p][i] in a FIXED order: the partials are cut into 16 contiguous groups, thread (group g, column
// quad c) adds its group's partials of four consecutive elements in index order (eight loads in flight), and the 16 group sums
// are added in group order through LDS.  One workgroup per 64-element strip: a 1024-partial LayerNorm reduction is 64 dependent
// steps per thread instead of 1024, a 32-way split-K reduction streams with every thread loading.
#define RED_STRIP 64
__device__ __forceinline__ void reduce_parts_strip(const float* part, size_t stride, int nparts, size_t n, float* out, float alpha, unsigned blk,
                                                   float (*sm)[RED_STRIP]) {
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t i = (size_t)blk * RED_STRIP + 4 * c;
    const int chunk = (nparts + 15) / 16;
    const int p0 = g * chunk, p1 = (p0 + chunk < nparts) ? p0 + chunk : nparts;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i + 4 <= n && !(stride & 3) && !((uintptr_t)part & 15)) {
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(part + (size_t)(p + u) * stride + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p < p1; ++p) s += *(const f32x4*)(part + (size_t)p * stride + i);
    } else {
        for (int p = p0; p < p1; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < n) s[e] += part[(size_t)p * stride + i + e];
    }
    *(f32x4*)&sm[g][4 * c] = s;
    __syncthreads();
    if (threadIdx.x < RED_STRIP) {
        const size_t j = (size_t)blk * RED_STRIP + threadIdx.x;
        if (j < n) {
            float r = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) r += sm[q][threadIdx.x];
            out[j] = alpha * r;
        }
    }
}
__global__ __launch_bounds__(256) void iefvad_reduce_parts_kernel(const float* part, size_t stride, int nparts, size_t n, float* out, float alpha) {
    __shared__ __attribute__((aligned(16))) float sm[16][RED_STRIP];
    reduce_parts_strip(part, stride, nparts, n, out, alpha, blockIdx.x, sm);
}

// Up to four such reductions in ONE launch (the weight gradient of a Linear, its second half when two parameters share a product
// -- the stacked mu | logvar heads -- and the bias gradient(s) of the same Linear; gamma and beta of a LayerNorm): job j owns
// workgroups [first[j], first[j + 1]).  Same arithmetic, same order: same bits as four launches of the kernel above.
struct ReduceJobs {
    const float* part[4]; size_t stride[4]; int nparts[4]; size_t n[4]; float* out[4]; float alpha[4];
    unsigned first[5];
    int count;
};
__global__ __launch_bounds__(256) void iefvad_reduce_parts_multi_kernel(ReduceJobs a) {
    __shared__ __attribute__((aligned(16))) float sm[16][RED_STRIP];
    int j = 0;
    while (j + 1 < a.count && blockIdx.x >= a.first[j + 1]) ++j;
    reduce_parts_strip(a.part[j], a.stride[j], a.nparts[j], a.n[j], a.out[j], a.alpha[j], blockIdx.x - a.first[j], sm);
}

// the 4 waves' per-lane column partials (lane l owns columns 4 l + 256 j) -> part[block][768], waves added in order
#define BWD_ROWS_PER_BLOCK 32
__device__ __forceinline__ void block_colpart_store(float* smem /* [4][768] */, const f32x4 (&v)[3], float* dst, int lane, int wave) {
#pragma unroll
    for (int j = 0; j < 3; ++j) *(f32x4*)(smem + wave * IEF_D + 4 * lane + 256 * j) = v[j];
    __syncthreads();
    for (int c = threadIdx.x; c < IEF_D; c += 256) dst[c] = (smem[c] + smem[IEF_D + c]) + (smem[2 * IEF_D + c] + smem[3 * IEF_D + c]);
    __syncthreads();
}

// ---- classifier backward (imf_vad.py:150: logits = fused_refined W_c^T + b_c) -----------------------------------------------------
struct ScorerBwdArgs {
    const float* dlogit;     // [rows], nullable (= 0)
    const float* dfused;     // [rows, 768] gradient handed to the `fused` output, nullable
    const float* z;          // [rows, 768] the refined state z_K
    const float* w;          // [768] classifier.weight
    float* g;                // [rows, 768] out: d z_K
    float* part_w;           // [blocks][768] partial sums of dlogit * z
    float* part_b;           // [blocks] partial sums of dlogit
    int rows;
};
__global__ __launch_bounds__(256) void iefvad_scorer_bwd_kernel(ScorerBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[4 * IEF_D];
    __shared__ float sb[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float accb = 0.f;
    f32x4 wv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) wv[j] = *(const f32x4*)(a.w + 4 * lane + 256 * j);
    for (int rr = wave; rr < BWD_ROWS_PER_BLOCK; rr += 4) {
        const int row = blockIdx.x * BWD_ROWS_PER_BLOCK + rr;
        if (row >= a.rows) break;
        const float dl = a.dlogit ? a.dlogit[row] : 0.f;
        accb += dl;
        const size_t base = (size_t)row * IEF_D + 4 * lane;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32x4 zv = *(const f32x4*)(a.z + base + 256 * j);
            f32x4 gv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { gv[e] = dl * wv[j][e]; acc[j][e] += dl * zv[e]; }
            if (a.dfused) {
                const f32x4 f = *(const f32x4*)(a.dfused + base + 256 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[e] += f[e];
            }
            *(f32x4*)(a.g + base + 256 * j) = gv;
        }
    }
    if (lane == 0) sb[wave] = accb;
    block_colpart_store(smem, acc, a.part_w + (size_t)blockIdx.x * IEF_D, lane, wave);
    if (threadIdx.x == 0) a.part_b[blockIdx.x] = (sb[0] + sb[1]) + (sb[2] + sb[3]);
}

// ---- precision weights + fusion backward (imf_vad.py:130-144) ------------------------------------------------------------------------
//   w_m = f exp(-l_m);  den = w_i + w_e + eps;  n_m = w_m / den;  z = n_i mu_i + n_e mu_e
// differentiated node by node as autograd walks the reference's expression graph:
//   d mu_m = gz n_m;   g n_m = gz mu_m (+ the gradient handed to the w_i / w_e outputs)
//   d den = -(g n_i n_i + g n_e n_e) / den;   d w_m = g n_m / den + d den;   d l_m = -w_m d w_m
// plus whatever the loss handed to the image_mu / event_mu / image_logvar / event_logvar outputs directly.  The results leave
// stacked per modality, [rows, 1536] = d mu | d logvar: the A operand of the heads' input gradient and of their weight gradients
// (the forward stacks mu.weight over logvar.weight the same way).
struct FusionBwdArgs {
    const float* mu_i; const float* lv_i; const float* mu_e; const float* lv_e;      // [rows, 768] saved outputs of the heads
    const float* gz;                                                                  // [rows, 768] d z_0
    const float* d_mu_i; const float* d_lv_i; const float* d_mu_e; const float* d_lv_e;   // nullable: direct gradients of the outputs
    const float* d_n_i; const float* d_n_e;                                           // nullable: gradients handed to w_i / w_e
    float* dh_i; float* dh_e;                                                         // [rows, 1536] out
    int rows;
    float factor, eps;
};
__global__ __launch_bounds__(256) void iefvad_fusion_bwd_kernel(FusionBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const size_t base = (size_t)row * IEF_D + 4 * lane, hb = (size_t)row * 2 * IEF_D + 4 * lane;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const size_t o = base + 256 * j;
        const f32x4 mi = *(const f32x4*)(a.mu_i + o), li = *(const f32x4*)(a.lv_i + o);
        const f32x4 me = *(const f32x4*)(a.mu_e + o), le = *(const f32x4*)(a.lv_e + o);
        const f32x4 gz = *(const f32x4*)(a.gz + o);
        f32x4 dmi, dli, dme, dle;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float wi = __fmul_rn(a.factor, expf(-li[e])), we = __fmul_rn(a.factor, expf(-le[e]));
            const float den = __fadd_rn(__fadd_rn(wi, we), a.eps);
            const float ni = wi / den, ne = we / den;
            float gni = gz[e] * mi[e], gne = gz[e] * me[e];
            if (a.d_n_i) gni += a.d_n_i[o + e];
            if (a.d_n_e) gne += a.d_n_e[o + e];
            const float dden = -(gni * ni + gne * ne) / den;
            const float dwi = gni / den + dden, dwe = gne / den + dden;
            dmi[e] = gz[e] * ni;
            dme[e] = gz[e] * ne;
            dli[e] = -wi * dwi;
            dle[e] = -we * dwe;
        }
        if (a.d_mu_i) { const f32x4 x = *(const f32x4*)(a.d_mu_i + o); dmi += x; }
        if (a.d_mu_e) { const f32x4 x = *(const f32x4*)(a.d_mu_e + o); dme += x; }
        if (a.d_lv_i) { const f32x4 x = *(const f32x4*)(a.d_lv_i + o); dli += x; }
        if (a.d_lv_e) { const f32x4 x = *(const f32x4*)(a.d_lv_e + o); dle += x; }
        *(f32x4*)(a.dh_i + hb + 256 * j) = dmi;
        *(f32x4*)(a.dh_i + hb + IEF_D + 256 * j) = dli;
        *(f32x4*)(a.dh_e + hb + 256 * j) = dme;
        *(f32x4*)(a.dh_e + hb + IEF_D + 256 * j) = dle;
    }
}

// ---- LayerNorm(768) backward (imf_vad.py:116-117,122-123; biased variance, eps inside the sqrt) ---------------------------------------
//   xh = (x - mean) rstd;  y = xh g + b
//   d b = sum_rows dy;  d g = sum_rows dy xh;  dx = rstd (dyg - mean(dyg) - xh mean(dyg xh)),  dyg = dy g
// x is the LayerNorm's INPUT (saved by the forward); mean and rstd are recomputed with the forward's own operations.
struct LnBwdArgs {
    const float* x;      // [rows, 768]
    const float* dy;     // [rows, 768]
    const float* g;      // [768]
    float* dx;           // [rows, 768] (may alias dy)
    float* part_g;       // [blocks][768]
    float* part_b;       // [blocks][768]
    int rows;
    float eps;
};
__global__ __launch_bounds__(256) void iefvad_layernorm_bwd_kernel(LnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[4 * IEF_D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 ag[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 ab[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 gv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) gv[j] = *(const f32x4*)(a.g + 4 * lane + 256 * j);
    for (int rr = wave; rr < BWD_ROWS_PER_BLOCK; rr += 4) {
        const int row = blockIdx.x * BWD_ROWS_PER_BLOCK + rr;
        if (row >= a.rows) break;
        const size_t base = (size_t)row * IEF_D + 4 * lane;
        f32x4 v[3], d[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            v[j] = *(const f32x4*)(a.x + base + 256 * j);
            d[j] = *(const f32x4*)(a.dy + base + 256 * j);
        }
        const float rstd = ln_center_rstd(v, a.eps);                // the forward's own statistics, bit for bit (rowops.h)
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = v[j][e] * rstd;
                v[j][e] = xh;
                ab[j][e] += d[j][e];
                ag[j][e] += d[j][e] * xh;
                const float dg = d[j][e] * gv[j][e];
                d[j][e] = dg;
                c1 += dg;
                c2 += dg * xh;
            }
        c1 = wave_sum(c1) * (1.0f / IEF_D);
        c2 = wave_sum(c2) * (1.0f / IEF_D);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (d[j][e] - c1 - v[j][e] * c2);
            *(f32x4*)(a.dx + base + 256 * j) = o;
        }
    }
    block_colpart_store(smem, ag, a.part_g + (size_t)blockIdx.x * IEF_D, lane, wave);
    block_colpart_store(smem, ab, a.part_b + (size_t)blockIdx.x * IEF_D, lane, wave);
}
