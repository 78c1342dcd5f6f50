// Dense projection GEMM for the d=768 layers, exact fp32 on the CDNA4 matrix cores.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T + bias[N] )
//
// Every Linear / in_proj / out_proj of the reference forward
// (/root/reference/model/imf_vad.py:115,121,125-128,148,150) is this contraction with
// K = 768 and W stored [out,in] exactly as torch stores nn.Linear.weight, so both operands
// are K-contiguous ("NT").
//
// Tiling: 128x128 block tile, BK = 32, 256 threads = 4 waves in a 2x2 grid, each wave a 64x64
// sub-tile = 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 VGPRs).  One MFMA takes lane (i, h):
// A[i][k=h], B[k=h][j=i'].  A lane fetches its operands as one ds_read_b128 = 4 consecutive k of
// its row and feeds them to 4 consecutive MFMAs; lane half h takes k = 8s+4h .. 8s+4h+3 of each
// 8-wide k group, so each half sums a different (but complete and disjoint) set of k: the order of
// the K summation differs from a sequential loop, the set of products does not.
// LDS image: [row][32 floats] with the 16-byte chunk index XORed by (row>>1)&7, which makes the
// four 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots (conflict-free).
// Staging is LDS-DMA (global_load_lds_dwordx4, see the comment at the main loop), one k-tile ahead;
// LDS is double buffered, one barrier per tile.
// 64 KB LDS + <256 VGPRs -> 2 blocks per CU, so one block's barrier wait hides under the other's MFMAs.
#pragma once
#include "common.h"

enum GemmEpilogue {
    EPI_BIAS = 0,        // C = acc + bias
    EPI_QKV = 1,         // C = (acc + bias) * (n < qcols ? qscale : 1)      in_proj; q pre-scaled by 1/sqrt(dh)
    EPI_BIAS_RELU = 2,   // C = relu(acc + bias)                              refinement Linear -> ReLU
    EPI_BIAS_RESID = 3,  // C = acc + bias + R                                out_proj + residual (x + attn_out)
    EPI_REFINE = 4,      // C = R - alpha * (acc + bias)                      z - lambda * block(z)
    EPI_HEADS = 5,       // n < 768: C = acc + bias ; n >= 768: C2 = acc + bias   (mu | logvar heads share A)
    EPI_GATE = 6,        // C = R > 0 ? alpha * (acc + bias) : 0   (split kernel only: ReLU backward on the saved activation, train.h)
};

struct GemmProblem {
    const float* A;      // [M, K], row stride lda
    const float* W;      // [N, K], row stride K (dense)
    const float* bias;   // [N]
    float* C;            // [M, ldc]
    const float* R;      // residual input, [M, ldc] (same layout as C); may alias C
    float* C2;           // second output for EPI_HEADS, [M, 768]
};

struct GemmArgs {
    GemmProblem p[2];    // blockIdx.z selects (image / event share shapes, differ in weights)
    int M, N, K;
    int lda, ldc;
    int epi;
    float alpha;         // lambda (EPI_REFINE) or qscale (EPI_QKV)
    int qcols;           // EPI_QKV: columns < qcols are scaled
};

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 32


// Residual operand of the EPI_BIAS_RESID / EPI_REFINE epilogues, fetched in accumulator layout.
// TM x TN = 32x32 accumulators per wave (wave tile 32 TM x 32 TN); (wr, wc) = wave position in the 2x2 wave grid.
template <int TM, int TN>
__device__ __forceinline__ void gemm_prefetch_residual(const GemmArgs& args, const GemmProblem& P, f32x16 (&res)[TM][TN],
                                                       int m0, int n0, int wr, int wc, int i, int h) {
    const int epi = args.epi, ldc = args.ldc;
    if (epi == EPI_BIAS_RESID || epi == EPI_REFINE) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    res[a][b][r] = P.R[(size_t)(m0 + wr * 32 * TM + a * 32 + 4 * h + (r & 3) + 8 * (r >> 2)) * ldc +
                                       n0 + wc * 32 * TN + b * 32 + i];
    }
}

// Epilogue.  Accumulator map (32x32 tile): col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& args, const GemmProblem& P, f32x16 (&acc)[TM][TN],
                                              f32x16 (&res)[TM][TN], int m0, int n0, int wr, int wc, int i, int h) {
    const int epi = args.epi, ldc = args.ldc;
    const float alpha = args.alpha;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wc * 32 * TN + b * 32 + i;
        const float bv = P.bias[n];
        float* Cb = P.C;
        int nn = n;
        float scale = 1.f;
        if (epi == EPI_HEADS && n >= IEF_D) { Cb = P.C2; nn = n - IEF_D; }
        if (epi == EPI_QKV && n < args.qcols) scale = alpha;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int mb = m0 + wr * 32 * TM + a * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                const size_t o = (size_t)m * ldc + nn;
                float v = acc[a][b][r] + bv;
                if (epi == EPI_QKV) v *= scale;
                else if (epi == EPI_BIAS_RELU) v = (v < 0.f) ? 0.f : v;
                else if (epi == EPI_BIAS_RESID) v = v + res[a][b][r];
                else if (epi == EPI_REFINE) v = res[a][b][r] - alpha * v;
                Cb[o] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void iefvad_gemm_f32_kernel(GemmArgs args) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (GEMM_BM + GEMM_BN) * GEMM_BK];   // 64 KB
    const GemmProblem& P = args.p[blockIdx.z];
    const int ntn = args.N / GEMM_BN;
    const int nwg = gridDim.x;
    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const int K = args.K, lda = args.lda;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    // staging: thread t moves chunk (row = (t>>3) + 32 j, ch = t&7), j = 0..3, of both tiles
    const int srow = t >> 3, sch = t & 7;

    // fragment read offsets (floats) inside a tile image
    const int fsw = (i >> 1) & 7;
    int aoff[2], boff[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        aoff[x] = (wr * 64 + x * 32 + i) * GEMM_BK;
        boff[x] = (wc * 64 + x * 32 + i) * GEMM_BK;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Staging: LDS-DMA (global_load_lds_dwordx4), one k-tile ahead, LDS double buffered.  One wave
    // instruction moves 64 x 16 B = 8 rows x 128 B straight into LDS (no VGPR round trip, no ds_write);
    // the LDS image is lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address and
    // again on the fragment read.  Thread t moves chunk (row = (t>>3) + 32 j, slot = t&7) for j = 0..3 of
    // both operand tiles; the eight DMAs of tile kt+1 are issued at the top of tile kt, so they have the
    // whole tile (64 MFMAs per wave) to land before the vmcnt(0) + barrier that ends it.
    const int nk = K / GEMM_BK;
    const int ssw = (srow >> 1) & 7;                          // (row>>1)&7, j-invariant
    // buffer addressing: SGPR descriptor + per-lane byte offset that never changes + scalar byte offset
    // that walks j and the k-tile -> the main loop spends no VALU instruction on addresses.
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * lda), 0,
                                                       (int)((GEMM_BM - 1) * lda + K) * 4, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (size_t)n0 * K), 0,
                                                       (int)((GEMM_BN - 1) * K + K) * 4, 0x00020000);
    const int voA = (srow * lda + ((sch ^ ssw) << 2)) * 4;
    const int voW = (srow * K + ((sch ^ ssw) << 2)) * 4;
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 8 * GEMM_BK;   // wave-uniform LDS offset (floats)
#define GLDS16(rs, vo, so, lp)                                                                      \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        GLDS16(rsA, voA, (32 * j * lda) * 4, smem + wbase + 32 * j * GEMM_BK);
        GLDS16(rsW, voW, (32 * j * K) * 4, smem + 2 * GEMM_BM * GEMM_BK + wbase + 32 * j * GEMM_BK);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
#define GEMM_TILE_BODY(STAGE_NEXT)                                                                         \
    {                                                                                                      \
        const float* As = smem + cur * GEMM_BM * GEMM_BK;                                                  \
        const float* Ws = smem + 2 * GEMM_BM * GEMM_BK + cur * GEMM_BN * GEMM_BK;                          \
        float* Ad = smem + (cur ^ 1) * GEMM_BM * GEMM_BK + wbase;                                          \
        float* Wd = smem + 2 * GEMM_BM * GEMM_BK + (cur ^ 1) * GEMM_BN * GEMM_BK + wbase;                  \
        const int k1 = (kt + 1) * GEMM_BK;                                                                 \
        if (STAGE_NEXT) {                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
                GLDS16(rsA, voA, (32 * j * lda + k1) * 4, Ad + 32 * j * GEMM_BK);                          \
                GLDS16(rsW, voW, (32 * j * K + k1) * 4, Wd + 32 * j * GEMM_BK);                            \
            }                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                  \
        f32x4 fa[2][2], fb[2][2];                                                                          \
        {                                                                                                  \
            const int ch = (h ^ fsw) << 2;                                                                 \
            fa[0][0] = *(const f32x4*)(As + aoff[0] + ch);                                                 \
            fa[0][1] = *(const f32x4*)(As + aoff[1] + ch);                                                 \
            fb[0][0] = *(const f32x4*)(Ws + boff[0] + ch);                                                 \
            fb[0][1] = *(const f32x4*)(Ws + boff[1] + ch);                                                 \
        }                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                    \
            const int c = s & 1, n = c ^ 1;                                                                \
            if (s < 3) { /* fragments of k-step s+1 are fetched a full step (16 MFMAs) ahead */            \
                const int ch = ((2 * (s + 1) + h) ^ fsw) << 2;                                             \
                fa[n][0] = *(const f32x4*)(As + aoff[0] + ch);                                             \
                fa[n][1] = *(const f32x4*)(As + aoff[1] + ch);                                             \
                fb[n][0] = *(const f32x4*)(Ws + boff[0] + ch);                                             \
                fb[n][1] = *(const f32x4*)(Ws + boff[1] + ch);                                             \
                __builtin_amdgcn_sched_barrier(0); /* keep the prefetch ahead of this step's MFMAs */      \
            }                                                                                              \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0][e], fb[c][0][e], acc[0][0], 0, 0, 0); \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0][e], fb[c][1][e], acc[0][1], 0, 0, 0); \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1][e], fb[c][0][e], acc[1][0], 0, 0, 0); \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1][e], fb[c][1][e], acc[1][1], 0, 0, 0); \
            }                                                                                              \
        }                                                                                                  \
    }
    int kt = 0;
    for (; kt + 1 < nk; ++kt) {
        GEMM_TILE_BODY(true)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    // last tile: nothing left to stage; epilogues that read a residual prefetch it here, under the MFMAs
    f32x16 res[2][2];
    gemm_prefetch_residual<2, 2>(args, P, res, m0, n0, wr, wc, i, h);
    GEMM_TILE_BODY(false)
#undef GEMM_TILE_BODY
#undef GLDS16

    gemm_epilogue<2, 2>(args, P, acc, res, m0, n0, wr, wc, i, h);
}


// ------------------------------------------------------------------------------------------------------------
// Low-latency variant for small M (the reference's per-video call pattern: B = 1 .. a few chunks): 64 x 64 block
// tile, 4 waves as 2 x 2 of one 32x32 accumulator, BK = 32, 32 KB LDS.  A launch has 4x the blocks and each
// block a quarter of the MFMA chain of the 128 x 128 kernel (24 k-tiles x 16 MFMAs = 10 us instead of 41 us),
// which is what bounds the time when the grid does not fill 256 CUs anyway.  Same LDS image, same staging and
// the same k summation order per output element: results are bit-identical to iefvad_gemm_f32_kernel.
// ------------------------------------------------------------------------------------------------------------
#define GEMS_BM 64
#define GEMS_BN 64

__global__ __launch_bounds__(256, 2) void iefvad_gemm_f32_small_kernel(GemmArgs args) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (GEMS_BM + GEMS_BN) * GEMM_BK];   // 32 KB
    const GemmProblem& P = args.p[blockIdx.z];
    const int ntn = args.N / GEMS_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * GEMS_BM, n0 = tn * GEMS_BN;
    const int K = args.K, lda = args.lda;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int srow = t >> 3, sch = t & 7;
    const int fsw = (i >> 1) & 7;
    const int aoff = (wr * 32 + i) * GEMM_BK, boff = (wc * 32 + i) * GEMM_BK;

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

    const int nk = K / GEMM_BK;
    const int ssw = (srow >> 1) & 7;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * lda), 0,
                                                       (int)((GEMS_BM - 1) * lda + K) * 4, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (size_t)n0 * K), 0,
                                                       (int)((GEMS_BN - 1) * K + K) * 4, 0x00020000);
    const int voA = (srow * lda + ((sch ^ ssw) << 2)) * 4;
    const int voW = (srow * K + ((sch ^ ssw) << 2)) * 4;
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 8 * GEMM_BK;
#define GLDS16(rs, vo, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
#define GEMS_STAGE(kk, buf)                                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                      \
        GLDS16(rsA, voA, (32 * j * lda + (kk)) * 4, smem + (buf) * GEMS_BM * GEMM_BK + wbase + 32 * j * GEMM_BK); \
        GLDS16(rsW, voW, (32 * j * K + (kk)) * 4,                                                        \
               smem + 2 * GEMS_BM * GEMM_BK + (buf) * GEMS_BN * GEMM_BK + wbase + 32 * j * GEMM_BK);     \
    }
    GEMS_STAGE(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    f32x16 res[1][1];
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            GEMS_STAGE((kt + 1) * GEMM_BK, cur ^ 1)
        } else {
            gemm_prefetch_residual<1, 1>(args, P, res, m0, n0, wr, wc, i, h);
        }
        const float* As = smem + cur * GEMS_BM * GEMM_BK;
        const float* Ws = smem + 2 * GEMS_BM * GEMM_BK + cur * GEMS_BN * GEMM_BK;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = ((2 * s + h) ^ fsw) << 2;
            const f32x4 fa = *(const f32x4*)(As + aoff + ch);
            const f32x4 fb = *(const f32x4*)(Ws + boff + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc[0][0], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    }
#undef GEMS_STAGE
#undef GLDS16
    gemm_epilogue<1, 1>(args, P, acc, res, m0, n0, wr, wc, i, h);
}


// ------------------------------------------------------------------------------------------------------------
// Lowest-latency variant for the reference's per-video call pattern (B = 1 .. a few chunks, M = 256 B rows): 32 x 32
// block tile, 4 waves as 2 x 2 of ONE 16 x 16 accumulator on v_mfma_f32_16x16x4_f32.  A launch has 4x the blocks of the
// 64 x 64 kernel (a 256 x 768 projection: 192 blocks instead of 48 on 256 CUs) and a wave's dependent MFMA chain is
// 192 x 40 cycles = 3.2 us instead of 384 x 64 = 10.2 us.  k-tiles are 1/3 the MFMA time of the 64 x 64 kernel's, shorter
// than a round trip to the Infinity Cache (the weights of a forward, 94 MB, do not stay in the 4 MB L2), so the LDS ring
// has 4 slots of 8 KB filled three k-tiles ahead behind a counted s_waitcnt vmcnt and a raw s_barrier.  Measured (rocprofv3,
// M = 256, N = 768): 12-15 us per launch against 23 us for the 64 x 64 kernel; a k-tile takes ~1100 cycles for 320 of MFMA
// chain -- LDS-DMA issue, fragment-read latency and the barrier of a workgroup that is alone on its CU; seven tiles ahead
// (8 slots) did not help (17 us), so it is not memory latency.
//
// Same summation order per output element as the 32x32x2 kernels -- k = 8s + {0,4,1,5,2,6,3,7}, k-tiles ascending -- so
// the results are bit-identical to them: MFMA j (j = 0, 1) of the 8-group s takes from lane group q = lane >> 4 the
// k = 8s + 2j + (q >> 1) + 4 (q & 1), i.e. the instruction's own k order 0..3 is (0,4,1,5) then (2,6,3,7).  A lane reads
// the 16-byte chunk 2s + (q & 1) of its row (the same image and swizzle as the other kernels; conflict-free for this
// pattern too: the eight even rows of a 16-lane group land on chunk ^ {0..7}) and picks elements (q >> 1) + 2j.
// ------------------------------------------------------------------------------------------------------------
#ifndef GEMT_EXP          // tools/gemm_tune_tiny timing experiments (wrong results when non-zero): 1 no DMA in the loop,
#define GEMT_EXP 0        // 2 no barrier, 4 no MFMA, 8 no fragment reads (bit mask)
#endif
#define GEMT_BM 32
#define GEMT_BN 32
#define GEMT_STAGES 4                                    // ring slots; GEMT_STAGES - 1 k-tiles are in flight (8 slots: 15.2 -> 17.3 us)
#define GEMT_SLOT ((GEMT_BM + GEMT_BN) * GEMM_BK)        // floats per ring slot: 64 rows x 128 B = 8 KB

__global__ __launch_bounds__(256, 2) void iefvad_gemm_f32_tiny_kernel(GemmArgs args) {
    __shared__ __attribute__((aligned(16))) float smem[GEMT_STAGES * GEMT_SLOT];   // 32 KB
    const GemmProblem& P = args.p[blockIdx.z];
    const int ntn = args.N / GEMT_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * GEMT_BM, n0 = tn * GEMT_BN;
    const int K = args.K, lda = args.lda;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;

    // staging: one LDS-DMA instruction = 8 rows x 128 B; wave w moves rows 8w..8w+7 of the A tile and of the W tile
    const int srow = lane >> 3, sch = lane & 7;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * lda), 0,
                                                       (int)((GEMT_BM - 1) * lda + K) * 4, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (size_t)n0 * K), 0,
                                                       (int)((GEMT_BN - 1) * K + K) * 4, 0x00020000);
    const int grow = uw * 8 + srow;                                  // row inside the 32-row tile
    const int ssw = (grow >> 1) & 7;
    const int voA = (grow * lda + ((sch ^ ssw) << 2)) * 4;
    const int voW = (grow * K + ((sch ^ ssw) << 2)) * 4;
#define GLDS16(rs, vo, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
#define GEMT_STAGE(kt_, slot_)                                                                    \
    {                                                                                             \
        float* d = smem + (slot_) * GEMT_SLOT + uw * 8 * GEMM_BK;                                 \
        GLDS16(rsA, voA, (kt_) * GEMM_BK * 4, d);                                                 \
        GLDS16(rsW, voW, (kt_) * GEMM_BK * 4, d + GEMT_BM * GEMM_BK);                             \
    }
    const int arow = wr * 16 + r16, brow = wc * 16 + r16;
    const int fsw = (r16 >> 1) & 7;                                  // (row >> 1) & 7 of both fragment rows (16 | 16 wr)
    const int aoff = arow * GEMM_BK, boff = GEMT_BM * GEMM_BK + brow * GEMM_BK;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nk = K / GEMM_BK;                 // even (the launcher requires K % 64 == 0)
    // Software pipeline across k-tiles: while the eight dependent MFMAs of tile kt run (320 cycles) on fragments already
    // in registers, the wave reads the fragments of tile kt+1 from LDS, so no MFMA ever waits for a ds_read; with the reads
    // inside the tile (first build) a k-tile cost ~1100 cycles: 4 x exposed read latency + DMA issue + barrier skew.
    //   iteration kt:  wait (own pieces of tile kt+1 landed)  ->  barrier  ->  issue the DMA of tile kt+3 (slot of tile
    //                  kt-1, whose readers finished an iteration ago)  ->  read fragments of tile kt+1  ->  MFMAs of tile kt
    // A lane needs elements (q >> 1) and (q >> 1) + 2 of chunk 2g + (q & 1) of its row: two dword reads at a fixed distance
    // of 2 floats (ds_read2_b32), NOT a ds_read_b128 plus selects -- fragment sets that live across loop iterations and are
    // indexed by a lane-dependent element made hipcc build 16-way v_cndmask chains.
    float pa0[8], pb0[8], pa1[8], pb1[8];      // fragment sets of the even / odd k-tiles (loop unrolled by two: no moves)
    int foff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) foff[g] = (((2 * g + (q & 1)) ^ fsw) << 2) + (q >> 1);
#define GEMT_READ(PA, PB, kt_)                                                            \
    {                                                                                     \
        const float* S = smem + ((kt_) % GEMT_STAGES) * GEMT_SLOT;                        \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                   \
            PA[2 * g] = S[aoff + foff[g]];                                                \
            PA[2 * g + 1] = S[aoff + foff[g] + 2];                                        \
            PB[2 * g] = S[boff + foff[g]];                                                \
            PB[2 * g + 1] = S[boff + foff[g] + 2];                                        \
        }                                                                                 \
    }
#define GEMT_MFMA(PA, PB)                                                                 \
    _Pragma("unroll") for (int e = 0; e < 8; ++e)                                         \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(PA[e], PB[e], acc, 0, 0, 0);
#define GEMT_RAW_BARRIER()                    \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");
    // (raw s_barrier: __syncthreads() would put s_waitcnt vmcnt(0) in front of it -- an LDS-DMA is a pending LDS write --
    // and drain the ring on every k-tile)
    // One k-tile: the eight MFMAs form one dependent chain (40 cycles each, 8 of them issue), so everything else of the
    // iteration is issued INSIDE the chain, pinned with sched_barrier: the two LDS-DMA instructions of tile kt+3 behind
    // MFMAs 0 and 1, the fragment reads of tile kt+1 (group g behind MFMA 2 + g).  Issued in front of the chain they cost
    // their full issue time (ablation, tools/gemm_tune_tiny: reads 2.6 us, DMA + barrier 3.5 us of a 13.4 us launch whose MFMA
    // chain is 3.2 us).
#define GEMT_SB() __builtin_amdgcn_sched_barrier(0)
#define GEMT_READ1(PA, PB, S_, g_)                                                        \
    {                                                                                     \
        PA[2 * (g_)] = (S_)[aoff + foff[g_]];                                             \
        PA[2 * (g_) + 1] = (S_)[aoff + foff[g_] + 2];                                     \
        PB[2 * (g_)] = (S_)[boff + foff[g_]];                                             \
        PB[2 * (g_) + 1] = (S_)[boff + foff[g_] + 2];                                     \
    }
#define GEMT_M(PA, PB, e_) if (!(GEMT_EXP & 4)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(PA[e_], PB[e_], acc, 0, 0, 0);
#define GEMT_ITER(CUR_A, CUR_B, NXT_A, NXT_B, kt_)                                        \
    {                                                                                     \
        const bool nxt = (kt_) + 1 < nk, dma = (kt_) + 3 < nk && !(GEMT_EXP & 1);         \
        const float* Sn = smem + (((kt_) + 1) % GEMT_STAGES) * GEMT_SLOT;                 \
        float* Dd = smem + (((kt_) + 3) % GEMT_STAGES) * GEMT_SLOT + uw * 8 * GEMM_BK;    \
        if (nxt) {                                                                        \
            if ((kt_) + 2 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         \
            if (!(GEMT_EXP & 2)) { GEMT_RAW_BARRIER() }                                   \
        }                                                                                 \
        GEMT_SB(); GEMT_M(CUR_A, CUR_B, 0) GEMT_SB();                                     \
        if (dma) GLDS16(rsA, voA, ((kt_) + 3) * GEMM_BK * 4, Dd);                         \
        GEMT_SB(); GEMT_M(CUR_A, CUR_B, 1) GEMT_SB();                                     \
        if (dma) GLDS16(rsW, voW, ((kt_) + 3) * GEMM_BK * 4, Dd + GEMT_BM * GEMM_BK);     \
        GEMT_SB(); GEMT_M(CUR_A, CUR_B, 2) GEMT_SB();                                     \
        if (nxt && !(GEMT_EXP & 8)) GEMT_READ1(NXT_A, NXT_B, Sn, 0)                       \
        GEMT_SB(); GEMT_M(CUR_A, CUR_B, 3) GEMT_SB();                                     \
        if (nxt && !(GEMT_EXP & 8)) GEMT_READ1(NXT_A, NXT_B, Sn, 1)                       \
        GEMT_SB(); GEMT_M(CUR_A, CUR_B, 4) GEMT_SB();                                     \
        if (nxt && !(GEMT_EXP & 8)) GEMT_READ1(NXT_A, NXT_B, Sn, 2)                       \
        GEMT_SB(); GEMT_M(CUR_A, CUR_B, 5) GEMT_SB();                                     \
        if (nxt && !(GEMT_EXP & 8)) GEMT_READ1(NXT_A, NXT_B, Sn, 3)                       \
        GEMT_SB(); GEMT_M(CUR_A, CUR_B, 6) GEMT_M(CUR_A, CUR_B, 7) GEMT_SB();             \
        /* retire the reads of tile kt+1 here (they landed under the chain), so that hipcc's own conservative lgkmcnt  */ \
        /* waits in front of the next tile's MFMAs (loop back-edge) find nothing outstanding                            */ \
        __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0) only; the builtin, so that the compiler's counter model sees it */ \
    }
    GEMT_STAGE(0, 0)
    if (nk > 1) GEMT_STAGE(1, 1)
    if (nk > 2) GEMT_STAGE(2, 2)
    if (nk > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMT_RAW_BARRIER()
    GEMT_READ(pa0, pb0, 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): the loop is entered with nothing outstanding on either path
    for (int kt = 0; kt < nk; kt += 2) {
        GEMT_ITER(pa0, pb0, pa1, pb1, kt)
        GEMT_ITER(pa1, pb1, pa0, pb0, kt + 1)
    }
#undef GEMT_ITER
#undef GEMT_M
#undef GEMT_READ1
#undef GEMT_SB
#undef GEMT_RAW_BARRIER
#undef GEMT_MFMA
#undef GEMT_READ
#undef GEMT_STAGE
#undef GLDS16
    // epilogue, accumulator map of the 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + reg
    const int epi = args.epi, ldc = args.ldc;
    const float alpha = args.alpha;
    const int n = n0 + wc * 16 + r16;
    const float bv = P.bias[n];
    float* Cb = P.C;
    int nn = n;
    float scale = 1.f;
    if (epi == EPI_HEADS && n >= IEF_D) { Cb = P.C2; nn = n - IEF_D; }
    if (epi == EPI_QKV && n < args.qcols) scale = alpha;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * 16 + 4 * q + r;
        const size_t o = (size_t)m * ldc + nn;
        float v = acc[r] + bv;
        if (epi == EPI_QKV) v *= scale;
        else if (epi == EPI_BIAS_RELU) v = (v < 0.f) ? 0.f : v;
        else if (epi == EPI_BIAS_RESID) v = v + P.R[(size_t)m * ldc + n];
        else if (epi == EPI_REFINE) v = P.R[(size_t)m * ldc + n] - alpha * v;
        Cb[o] = v;
    }
}

