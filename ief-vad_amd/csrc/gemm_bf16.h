// Dense projection GEMMs on the 128 x 256 / 3-slot-ring structure, and the bf16 operand path.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T + bias[N] ),  A and W K-contiguous; bias, residual and C fp32;
//   optionally a bf16 copy Cb of the result (the A operand of the next bf16 projection).
//
// Kernels in this file
//   iefvad_gemm_f32_t256_kernel   fp32 operands, v_mfma_f32_32x32x2_f32 -- the throughput kernel of the default
//                                 (parity) mode; bit-identical to the 128x128 and 64x64 kernels of gemm_f32.h
//   iefvad_gemm_bf16_kernel       bf16 operands, v_mfma_f32_16x16x32_bf16 -- throughput mode ("bf16 projections with
//                                 fp32 fusion state", BASELINE config 3)
//   iefvad_gemm_bf16_m32_kernel   the same on v_mfma_f32_32x32x16_bf16 (A/B runs)
//   iefvad_gemm_bf16_v1_kernel    128x128 double-buffer fallback for shapes the ring kernel does not take
// All three ring kernels are one template body (gemm_t256_body): same LDS image, LDS-DMA staging, ring protocol and
// LDS-staged epilogue; only the fragment -> MFMA step differs.
//
// v1 (below): 128x128 block tile, BK = 64 bf16 (128-byte rows: the LDS image, its XOR swizzle, the LDS-DMA staging
// pattern and the fragment addressing are byte-for-byte those of iefvad_gemm_f32_kernel), 4 waves as 2x2 of 64x64,
// v_mfma_f32_32x32x16_bf16 (lane (i, h) supplies A[i][k = 8h + j], B[k = 8h + j][i'], j = 0..7, as one 16-byte
// fragment), double-buffered 64 KB LDS -> 2 blocks per CU.  Its epilogue goes through LDS: each wave parks its
// 64x64 fp32 tile in a private, padded image and re-reads it row-wise, so global stores (and the residual / bias
// loads) are 16 bytes per lane and whole 256-byte row segments per 16 lanes.
#pragma once
#include "common.h"
#include "gemm_f32.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct GemmBProblem {
    const bf16_t* A;     // [M, K] bf16, row stride lda (elements)
    const bf16_t* W;     // [N, K] bf16, dense
    const float* bias;   // [N]
    float* C;            // [M, ldc] fp32 result, nullable
    bf16_t* Cb;          // [M, ldc] bf16 copy of the result, nullable
    const float* R;      // residual, fp32, same layout as C
    float* C2;           // EPI_HEADS second output (fp32, [M, 768])
    // fp16x3 mode (gemm_split.h, F16 path): running max |.| words of the A tensor, of the W matrix (both read) and of the
    // result written to C (updated), nullable
    const float* amaxA;
    const float* amaxW;
    float* amaxC;
};

struct GemmBArgs {
    GemmBProblem p[2];
    int M, N, K;
    int lda, ldc;
    int epi;
    float alpha;
    int qcols;
    int wplane;          // split kernel (gemm_split.h): bytes between the three bf16 planes of W
    int stagger;         // ring kernels: the second resident workgroup of each CU starts `stagger` x 8128 cycles late
};

#define GEMMB_BK 64                 // bf16 elements per k-tile (128 bytes per row)
#define GEMMB_TILE (GEMM_BM * 32)   // floats-equivalent (4-byte units) per operand tile image: 128 rows x 128 B
#define GEMMB_EPI_LD 68             // padded row length (floats) of the per-wave epilogue image

__global__ __launch_bounds__(256, 2) void iefvad_gemm_bf16_v1_kernel(GemmBArgs args) {
    __shared__ __attribute__((aligned(16))) float smem[4 * GEMMB_TILE];   // 64 KB: [A0|A1|W0|W1], reused by the epilogue
    const GemmBProblem& P = args.p[blockIdx.z];
    const int ntn = args.N / GEMM_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const int K = args.K, lda = args.lda;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    const int srow = t >> 3, sch = t & 7;
    const int ssw = (srow >> 1) & 7;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * lda), 0,
                                                       (int)((GEMM_BM - 1) * lda + K) * 2, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (size_t)n0 * K), 0,
                                                       (int)((GEMM_BN - 1) * K + K) * 2, 0x00020000);
    const int voA = srow * lda * 2 + ((sch ^ ssw) << 4);
    const int voW = srow * K * 2 + ((sch ^ ssw) << 4);
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 8 * 32;     // 8 rows x 128 B, in 4-byte units
#define GLDS16(rs, vo, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)

    const int fsw = (i >> 1) & 7;
    int aoff[2], boff[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        aoff[x] = (wr * 64 + x * 32 + i) * 32;
        boff[x] = (wc * 64 + x * 32 + i) * 32;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / GEMMB_BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        GLDS16(rsA, voA, (32 * j * lda) * 2, smem + wbase + 32 * j * 32);
        GLDS16(rsW, voW, (32 * j * K) * 2, smem + 2 * GEMMB_TILE + wbase + 32 * j * 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
#define GEMMB_TILE_BODY(STAGE_NEXT)                                                                          \
    {                                                                                                        \
        const float* As = smem + cur * GEMMB_TILE;                                                           \
        const float* Ws = smem + 2 * GEMMB_TILE + cur * GEMMB_TILE;                                          \
        float* Ad = smem + (cur ^ 1) * GEMMB_TILE + wbase;                                                   \
        float* Wd = smem + 2 * GEMMB_TILE + (cur ^ 1) * GEMMB_TILE + wbase;                                  \
        const int k1 = (kt + 1) * GEMMB_BK;                                                                  \
        if (STAGE_NEXT) {                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
                GLDS16(rsA, voA, (32 * j * lda + k1) * 2, Ad + 32 * j * 32);                                 \
                GLDS16(rsW, voW, (32 * j * K + k1) * 2, Wd + 32 * j * 32);                                   \
            }                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                    \
        bf16x8 fa[2][2], fb[2][2];                                                                           \
        {                                                                                                    \
            const int ch = (h ^ fsw) << 2;                                                                   \
            fa[0][0] = *(const bf16x8*)(As + aoff[0] + ch);                                                  \
            fa[0][1] = *(const bf16x8*)(As + aoff[1] + ch);                                                  \
            fb[0][0] = *(const bf16x8*)(Ws + boff[0] + ch);                                                  \
            fb[0][1] = *(const bf16x8*)(Ws + boff[1] + ch);                                                  \
        }                                                                                                    \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                      \
            const int c = s & 1, n = c ^ 1;                                                                  \
            if (s < 3) {                                                                                     \
                const int ch = ((2 * (s + 1) + h) ^ fsw) << 2;                                               \
                fa[n][0] = *(const bf16x8*)(As + aoff[0] + ch);                                              \
                fa[n][1] = *(const bf16x8*)(As + aoff[1] + ch);                                              \
                fb[n][0] = *(const bf16x8*)(Ws + boff[0] + ch);                                              \
                fb[n][1] = *(const bf16x8*)(Ws + boff[1] + ch);                                              \
                __builtin_amdgcn_sched_barrier(0);                                                           \
            }                                                                                                \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);     \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);     \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);     \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);     \
        }                                                                                                    \
    }
    int kt = 0;
    for (; kt + 1 < nk; ++kt) {
        GEMMB_TILE_BODY(true)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue -------------------------------------------------------------------------------------
    // Row-wise view of the wave's 64x64 tile: lane (rq = lane >> 4, cq = lane & 15) owns columns 4 cq .. 4 cq + 3
    // of rows rq + 4 u, u = 0..15.  The residual is fetched in that view before the last k-tile's MFMAs.
    const int epi = args.epi, ldc = args.ldc;
    const float alpha = args.alpha;
    const int rq = lane >> 4, cq = lane & 15;
    const int ncol = n0 + wc * 64 + 4 * cq;                 // first of this lane's 4 output columns
    const int mrow = m0 + wr * 64 + rq;
    const bool has_resid = (epi == EPI_BIAS_RESID || epi == EPI_REFINE);
    f32x4 res[16];
    if (has_resid) {
#pragma unroll
        for (int u = 0; u < 16; ++u) res[u] = *(const f32x4*)(P.R + (size_t)(mrow + 4 * u) * ldc + ncol);
    }
    GEMMB_TILE_BODY(false)
#undef GEMMB_TILE_BODY
#undef GLDS16
    __syncthreads();                                        // every wave is done reading the operand images
    float* E = smem + wave * (GEMMB_TILE);                  // 16 KB per wave; 64 x 68 floats = 17 KB would overflow:
                                                            // the tile is parked in two 32-row halves (8.5 KB each)
    const f32x4 bv = *(const f32x4*)(P.bias + ncol);
    float* Cb32 = P.C;
    bf16_t* Cb16 = P.Cb;
    int nn = ncol;
    f32x4 scale = {1.f, 1.f, 1.f, 1.f};
    if (epi == EPI_HEADS && ncol >= IEF_D) { Cb32 = P.C2; nn = ncol - IEF_D; }
    if (epi == EPI_QKV && ncol < args.qcols) scale = f32x4{alpha, alpha, alpha, alpha};
#pragma unroll
    for (int a = 0; a < 2; ++a) {                           // 32-row half a of the wave tile
        // park: accumulator (col = i, row = (r&3) + 8(r>>2) + 4h) of both 32-column blocks
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                E[((r & 3) + 8 * (r >> 2) + 4 * h) * GEMMB_EPI_LD + b * 32 + i] = acc[a][b][r];
        // (wave-private image: LDS ops of one wave complete in order, no barrier needed)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int lr = rq + 4 * u;                      // row inside the half
            f32x4 v = *(const f32x4*)(E + lr * GEMMB_EPI_LD + 4 * cq);
            v = v + bv;
            if (epi == EPI_QKV) v = v * scale;
            else if (epi == EPI_BIAS_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];
            } else if (epi == EPI_BIAS_RESID) v = v + res[a * 8 + u];
            else if (epi == EPI_REFINE) v = res[a * 8 + u] - alpha * v;
            const size_t o = (size_t)(mrow + a * 32 + 4 * u) * ldc + nn;
            if (Cb32) *(f32x4*)(Cb32 + o) = v;
            if (Cb16) {
                bf16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (bf16_t)v[e];
                *(bf16x4*)(Cb16 + o) = w;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// Ring structure: 128 x 256 block tile, 4 waves as 2 x 2 of 64 x 128 (8 accumulators), 64-byte k-tile rows (32 bf16 /
// 16 fp32), 3-slot
// LDS ring (3 x 24 KB = 72 KB -> 2 blocks per CU) filled by LDS-DMA two k-tiles ahead with a COUNTED
// s_waitcnt vmcnt(6): the k-tile of this GEMM is only 16 MFMAs (512 cycles) per wave, shorter than an L2 round
// trip, so a one-tile-ahead double buffer (v1) exposes the DMA latency every tile; two tiles of distance
// (>= 2048 cycles with two waves per SIMD) covers it.  The 64 x 128 wave tile needs 6 fragment reads per 16
// MFMAs (v1: 8) and 25 % fewer staged bytes per FLOP.  LDS image per slot: [384 rows][64 B] (A rows then W
// rows), 16-byte chunk index XOR (row>>2)&3 (conflict-free ds_read_b128).  Same k order as v1: bit-identical.
// ------------------------------------------------------------------------------------------------------------
// raw s_barrier (no vmcnt(0): LDS-DMA of later tiles stays in flight across it) fenced for the COMPILER on both
// sides: the intrinsic is IntrNoMem, so without the empty asm a later ds_read could legally be hoisted above it
#define GB2_BARRIER()                       \
    do {                                    \
        asm volatile("" ::: "memory");      \
        __builtin_amdgcn_s_barrier();       \
        asm volatile("" ::: "memory");      \
    } while (0)
#define GB2_BM 128
#define GB2_BN 256
#define GB2_BK 32                                   // bf16 elements per k-tile
#define GB2_SLOT ((GB2_BM + GB2_BN) * 16)           // 4-byte units per ring slot (384 rows x 64 B)
#define GB2_STAGES 3
#define GB2_EPI_LD 132                              // padded row (floats) of the per-wave epilogue image (32 x 128)
#define GB2_LDS_BYTES (GB2_STAGES * GB2_SLOT * 4)   // 73,728 B

// ---- epilogue of the 128 x 256 kernels, through LDS: each wave parks 32 rows x 128 columns of its 64 x 128 tile at a
// time in a private padded image (the ring is dead by then; the caller has put a barrier after its last read) and
// re-reads them row-wise, so bias / residual loads and stores are 16 bytes per lane and 512-byte row segments per
// instruction.  MF16 selects the accumulator map (16x16 tiles in acc16, else 32x32 tiles in acc).
// The wave tile is (32 PASSES) rows x 128 columns with origin (wrow0, wcol0) inside the block tile at (m0, n0).
// `cscale` multiplies the accumulators before the bias (1 except in the fp16x3 path, where it undoes the operand scales);
// `amax_out` (nullable) is the per-chunk running-max slot of C: word (row / 256) receives max |value written|.
// Result stores of the ring / split kernels.  GB2_NT_STORES=1 marks them non-temporal (global_store ... nt): a launch writes
// 0.8-4.8 GB that no workgroup of the SAME launch reads again, while the A panels and W planes its co-resident workgroups
// share must stay in the 4 MB L2 of their XCD.
#ifndef GB2_EPI_SLEEP
#define GB2_EPI_SLEEP 0
#endif
#ifndef GB2_NT_STORES
#define GB2_NT_STORES 1      // bf16 ring kernel +3..6 % with fp32 results, split kernels +0..1 % (profiles/r02_gemm_nt_stores.log)
#endif
#ifndef GB2_NT_RESID
#define GB2_NT_RESID 0       // experiment (round 5): the epilogue's residual / gate rows as non-temporal loads: +4 % in the stand-alone
                             // harness (a 200 MB residual buffer re-read every iteration), nothing in the forward (TRIED.md)
#endif
#if GB2_NT_STORES
#define GB2_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define GB2_STORE(ptr, val) (*(ptr) = (val))
#endif
template <bool MF16, int PASSES>
__device__ __forceinline__ void gemm_wave_epilogue(const GemmBArgs& args, const GemmBProblem& P, float* smem, int m0, int n0,
                                                   int wrow0, int wcol0, f32x16 (&acc)[PASSES][4],
                                                   f32x4 (&acc16)[2 * PASSES][8], float cscale = 1.0f, float* amax_out = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int r16 = lane & 15, q16 = lane >> 4;
    // ---- epilogue through LDS: the wave parks 32 rows x 128 columns at a time and re-reads them row-wise;
    // lane (rq = lane >> 5, cq = lane & 31) owns columns 4 cq .. 4 cq + 3 of rows rq + 2 u, u = 0..15.
    const int epi = args.epi, ldc = args.ldc;
    const float alpha = args.alpha;
    const int rq = lane >> 5, cq = lane & 31;
    const int ncol = n0 + wcol0 + 4 * cq;
    const bool has_resid = (epi == EPI_BIAS_RESID || epi == EPI_REFINE || epi == EPI_GATE);
    float* E = smem + wave * (32 * GB2_EPI_LD);       // 16.9 KB per wave
    const f32x4 bv = *(const f32x4*)(P.bias + ncol);
    float* C32 = P.C;
    bf16_t* C16 = P.Cb;
    int nn = ncol;
    f32x4 scale = {1.f, 1.f, 1.f, 1.f};
    if (epi == EPI_HEADS && ncol >= IEF_D) { C32 = P.C2; nn = ncol - IEF_D; }
    if (epi == EPI_QKV && ncol < args.qcols) scale = f32x4{alpha, alpha, alpha, alpha};
    float vmax = 0.f;
    // bf16-only results (in_proj q|k|v and the refinement's ReLU output in the bf16 mode): a lane takes EIGHT columns of a row,
    // so that a store is 16 bytes per lane like the fp32 ones (half the store instructions of the 8-byte form)
    const bool wide16 = C16 && !C32 && !has_resid && !amax_out && (epi == EPI_BIAS || epi == EPI_QKV || epi == EPI_BIAS_RELU);
    if (wide16) {
        const int rq4 = lane >> 4, c8 = lane & 15;
        const int ncol8 = n0 + wcol0 + 8 * c8;
        const f32x4 b0 = *(const f32x4*)(P.bias + ncol8), b1 = *(const f32x4*)(P.bias + ncol8 + 4);
        const float sc = (epi == EPI_QKV && ncol8 < args.qcols) ? alpha : 1.f;      // qcols is a multiple of 8
#pragma unroll
        for (int a = 0; a < PASSES; ++a) {
            if constexpr (MF16) {
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int b = 0; b < 8; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            E[(x * 16 + 4 * q16 + r) * GB2_EPI_LD + b * 16 + r16] = acc16[2 * a + x][b][r];
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        E[((r & 3) + 8 * (r >> 2) + 4 * h) * GB2_EPI_LD + b * 32 + i] = acc[a][b][r];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = rq4 + 4 * u;
                f32x4 v0 = *(const f32x4*)(E + row * GB2_EPI_LD + 8 * c8);
                f32x4 v1 = *(const f32x4*)(E + row * GB2_EPI_LD + 8 * c8 + 4);
                v0 = v0 * cscale + b0;
                v1 = v1 * cscale + b1;
                if (epi == EPI_QKV) { v0 = v0 * sc; v1 = v1 * sc; }
                bf16x8 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x0 = v0[e], x1 = v1[e];
                    if (epi == EPI_BIAS_RELU) { x0 = x0 < 0.f ? 0.f : x0; x1 = x1 < 0.f ? 0.f : x1; }
                    w[e] = (bf16_t)x0;
                    w[4 + e] = (bf16_t)x1;
                }
                GB2_STORE((bf16x8*)(C16 + (size_t)(m0 + wrow0 + a * 32 + row) * ldc + ncol8), w);
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < PASSES; ++a) {
        const int mrow = m0 + wrow0 + a * 32 + rq;
        f32x4 res[16];
        if (has_resid) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                res[u] = GB2_NT_RESID ? __builtin_nontemporal_load((const f32x4*)(P.R + (size_t)(mrow + 2 * u) * ldc + ncol))
                                      : *(const f32x4*)(P.R + (size_t)(mrow + 2 * u) * ldc + ncol);
        }
        if constexpr (MF16) {
            // 16x16 accumulator map: col = lane & 15, row = 4 (lane >> 4) + reg; this pass takes row sub-tiles 2a, 2a+1
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int b = 0; b < 8; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        E[(x * 16 + 4 * q16 + r) * GB2_EPI_LD + b * 16 + r16] = acc16[2 * a + x][b][r];
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    E[((r & 3) + 8 * (r >> 2) + 4 * h) * GB2_EPI_LD + b * 32 + i] = acc[a][b][r];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            f32x4 v = *(const f32x4*)(E + (rq + 2 * u) * GB2_EPI_LD + 4 * cq);
            v = v * cscale + bv;
            if (epi == EPI_QKV) v = v * scale;
            else if (epi == EPI_BIAS_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];
            } else if (epi == EPI_BIAS_RESID) v = v + res[u];
            else if (epi == EPI_REFINE) v = res[u] - alpha * v;
            else if (epi == EPI_GATE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = res[u][e] > 0.f ? alpha * v[e] : 0.f;
            }
            const size_t o = (size_t)(mrow + 2 * u) * ldc + nn;
            if (C32) GB2_STORE((f32x4*)(C32 + o), v);
            if (C16) {
                bf16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (bf16_t)v[e];
                GB2_STORE((bf16x4*)(C16 + o), w);
            }
            if (amax_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vmax = amax_fold(vmax, v[e]);
            }
#if GB2_EPI_SLEEP
            __builtin_amdgcn_s_sleep(GB2_EPI_SLEEP);      // experiment: spread the stores over the partner's main loop
#endif
        }
    }
    if (amax_out)      // a wave tile lies inside one chunk; its word: (32-row band of the chunk, 128-column tile), N <= 4096
        amax_store_part(amax_out, (m0 + wrow0) / IEF_T, (((m0 + wrow0) % IEF_T) / 32) * (args.N / 128) + (n0 + wcol0) / 128, wave_max(vmax), lane);
}

// the 2 x 2 wave layout of the 128 x 256 kernels: wave (wr, wc) owns the 64 x 128 tile at (64 wr, 128 wc)
template <bool MF16>
__device__ __forceinline__ void gemm_t256_epilogue(const GemmBArgs& args, const GemmBProblem& P, float* smem, int m0, int n0,
                                                   f32x16 (&acc)[2][4], f32x4 (&acc16)[4][8]) {
    const int wave = threadIdx.x >> 6;
    gemm_wave_epilogue<MF16, 2>(args, P, smem, m0, n0, (wave >> 1) * 64, (wave & 1) * 128, acc, acc16);
}

// One body, two element types: F32 = false -> bf16 operands (32 elements per 64-byte row, v_mfma_f32_32x32x16_bf16);
// F32 = true -> fp32 operands (16 elements per row; each 16-byte fragment feeds four v_mfma_f32_32x32x2_f32, lane
// half h taking k = 8s+4h..+3 exactly as iefvad_gemm_f32_kernel does, so the fp32 results are bit-identical to it).
// MODE 2 = bf16 operands on v_mfma_f32_16x16x32_bf16: one 16-byte fragment (lane (r = lane & 15, q = lane >> 4) holds
// row r, k = 8q .. 8q+7) covers the whole 32-wide k-tile, so a tile is 4 A + 8 B fragment reads feeding 32 MFMAs of
// 16 cycles -- the same reads, FLOPs and cycles as MODE 0 -- but the chip holds a higher clock on this shape
// (MI355X_MICROARCH.md, DVFS item 7; measured here: see DESIGN.md).  Its LDS image uses the chunk swizzle
// G[(row>>2)&3], G = {2,0,1,3}, which makes the (row, q) access pattern of the 16x16 fragment conflict-free.
enum { T256_BF16_32 = 0, T256_F32 = 1, T256_BF16_16 = 2, T256_BF16_16P = 3 };   // 3 = 2 with the pinned issue order below

// BM = 128: 4 waves (2 x 2 of 64 x 128), two workgroups per CU.  BM = 256 (MODE 3 only): 8 waves (4 x 2 of 64 x 128), ONE workgroup
// per CU, 256 x 256 block tile: a k-tile stages 32 KB for twice the MFMAs of two 128 x 256 tiles' 48 KB -- the bf16 main
// loop is bound by what a CU can take in through LDS-DMA (~46-68 GB/s; 47 B/clk at the full MFMA rate with 128 x 256 tiles).
template <int MODE, int BM = GB2_BM>
__device__ __forceinline__ void gemm_t256_body(const GemmBArgs& args, float* smem) {
    constexpr int NTHR = 2 * BM;                        // 256 or 512 threads
    constexpr int RPJ = NTHR / 4;                       // rows one staging instruction of the whole workgroup covers (64 / 128)
    constexpr int NJA = BM / RPJ, NJW = GB2_BN / RPJ;   // staging instructions per wave and k-tile: A (2), W (4 / 2)
    constexpr int SLOT = (BM + GB2_BN) * 16;            // 4-byte units per ring slot
#ifdef GB2_CLOCK_DIAG
    const unsigned long long dg_entry = __builtin_amdgcn_s_memtime();
#endif
    constexpr bool F32 = (MODE == T256_F32);
    constexpr bool MF16 = (MODE == T256_BF16_16 || MODE == T256_BF16_16P);
    constexpr bool PIPE = (MODE == T256_BF16_16P);
    constexpr int EB = F32 ? 4 : 2;                     // bytes per operand element
    constexpr int BKE = 64 / EB;                        // elements per k-tile row (64 bytes)
    const GemmBProblem& P = args.p[blockIdx.z];
    const int ntn = args.N / GB2_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * BM, n0 = tn * GB2_BN;
    const int K = args.K, lda = args.lda;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    // De-phasing.  All workgroups of a launch start together and take the same time, so chip-wide every main loop
    // (matrix pipe busy, HBM idle) and every epilogue (128 KB of stores per workgroup: HBM saturated, matrix pipe idle)
    // coincide.  The launch's first 512 workgroups are the two residents of each CU; the second 256 of them (dispatch
    // order is linear in blockIdx -- an observed property used for speed only) wait about half a tile time once, and
    // from then on one resident of a CU stores while the other multiplies.
    if (args.stagger != 0) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.z;
        const int n = args.stagger > 0 ? args.stagger : -args.stagger;
        // which of the first 512 workgroups share a CU is not documented: > 0 assumes (b, b + 256), < 0 assumes consecutive
        // workgroups of one XCD, (b, b + 8)
        const bool late = args.stagger > 0 ? (lin >= 256 && lin < 512) : (lin < 512 && ((lin >> 3) & 1));
        if (n >= 1000) {
            // uniform phases: the first 512 workgroups (every initial resident) start at one of 64 evenly spaced offsets
            // over (n - 1000) x 64 x 64 cycles, so that chip-wide the store traffic is a steady stream, not a burst per tile
            if (lin < 512) {
                const int ph = (lin * 29) & 63;
                for (int w = 0; w < ph; ++w)
                    for (int u = 0; u < n - 1000; ++u) __builtin_amdgcn_s_sleep(1);
            }
        } else if (late)
            for (int w = 0; w < n; ++w) __builtin_amdgcn_s_sleep(127);
    }

    // staging: thread t moves chunk (row = (t>>2) + RPJ j, slot chunk = t&3); A: j < NJA, W: j < NJW
    const int srow = t >> 2, sch = t & 3;
    auto swz = [](int row) { const int x = (row >> 2) & 3; return MF16 ? ((0xD2 >> (2 * x)) & 3) : x; };
    const int ssw = swz(srow);                          // rows srow + 64 j share it
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)P.A + (size_t)m0 * lda * EB), 0,
                                                       (int)((BM - 1) * lda + K) * EB, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)P.W + (size_t)n0 * K * EB), 0,
                                                       (int)((GB2_BN - 1) * K + K) * EB, 0x00020000);
    const int voA = srow * lda * EB + ((sch ^ ssw) << 4);
    const int voW = srow * K * EB + ((sch ^ ssw) << 4);
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 16 * 16;    // this wave's 16 rows x 64 B, 4-byte units
#define GLDS16(rs, vo, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
#define GB2_STAGE(tile, slotbase)                                                                     \
    {                                                                                                 \
        float* Ad = smem + (slotbase) + wbase;                                                        \
        float* Wd = Ad + BM * 16;                                                                     \
        const int kk = (tile) * BKE;                                                               \
        _Pragma("unroll") for (int j = 0; j < NJA; ++j) GLDS16(rsA, voA, (RPJ * j * lda + kk) * EB, Ad + RPJ * j * 16); \
        _Pragma("unroll") for (int j = 0; j < NJW; ++j) GLDS16(rsW, voW, (RPJ * j * K + kk) * EB, Wd + RPJ * j * 16);   \
    }

    const int fsw = (i >> 2) & 3;
    int aoff[2], boff[4];
#pragma unroll
    for (int x = 0; x < 2; ++x) aoff[x] = (wr * 64 + x * 32 + i) * 16;
#pragma unroll
    for (int x = 0; x < 4; ++x) boff[x] = BM * 16 + (wc * 128 + x * 32 + i) * 16;
    // 16x16x32 fragments: lane (r16, q16) reads row (16 x + r16), chunk q16 (swizzled) of the A / W image
    const int r16 = lane & 15, q16 = lane >> 4;
    const int f16 = (q16 ^ swz(r16)) << 2;              // rows 16 x + r16 share (row>>2)&3 with r16
    const int a16 = (wr * 64 + r16) * 16 + f16;
    const int b16 = BM * 16 + (wc * 128 + r16) * 16 + f16;

    f32x16 acc[2][4];
    f32x4 acc16[4][8];
    if constexpr (MF16) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }

#define GB2_COMPUTE(slotbase)                                                                               \
    if constexpr (MF16) {                                                                                   \
        const float* S = smem + (slotbase);                                                                 \
        f32x4 ga[4], gb[8];                                                                                 \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) ga[x] = *(const f32x4*)(S + a16 + x * 16 * 16);       \
        _Pragma("unroll") for (int x = 0; x < 8; ++x) gb[x] = *(const f32x4*)(S + b16 + x * 16 * 16);       \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                       \
            _Pragma("unroll") for (int b = 0; b < 8; ++b)                                                   \
                acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                      \
                    __builtin_bit_cast(bf16x8, ga[a]), __builtin_bit_cast(bf16x8, gb[b]), acc16[a][b], 0, 0, 0); \
    } else {                                                                                                \
        const float* S = smem + (slotbase);                                                                 \
        f32x4 fa[2][2], fb[2][4];   /* 16-byte fragments: 4 fp32 or 8 bf16 */                                \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                     \
            const int ch = ((2 * s + h) ^ fsw) << 2;                                                        \
            _Pragma("unroll") for (int x = 0; x < 2; ++x) fa[s][x] = *(const f32x4*)(S + aoff[x] + ch);     \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) fb[s][x] = *(const f32x4*)(S + boff[x] + ch);     \
        }                                                                                                   \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                     \
            if constexpr (F32) {                                                                            \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                               \
                    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                           \
                        _Pragma("unroll") for (int b = 0; b < 4; ++b)                                       \
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][a][e], fb[s][b][e], acc[a][b], 0, 0, 0); \
            } else {                                                                                        \
                _Pragma("unroll") for (int a = 0; a < 2; ++a)                                               \
                    _Pragma("unroll") for (int b = 0; b < 4; ++b)                                           \
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                \
                            __builtin_bit_cast(bf16x8, fa[s][a]), __builtin_bit_cast(bf16x8, fb[s][b]), acc[a][b], 0, 0, 0); \
            }                                                                                               \
        }                                                                                                   \
    }


    // ---- MODE 3: the k-tile of MODE 2 with its issue order pinned (sched_group_barrier).  A k-tile is 32 MFMAs of 16
    // cycles per wave; in MODE 2 the wave first issues its six LDS-DMA instructions of tile kt+2 (60-180 cycles EACH when
    // the vector-memory queue is busy), then twelve fragment reads, then the MFMAs -- with two waves per SIMD in lockstep
    // the matrix pipe idles through both preambles.  Here the wave reads ga[0..3], gb[0..1], and every step b (4 MFMAs of
    // column tile b) carries one fragment read (gb[b+2]) and, for b < 6, ONE of the six DMA instructions between its
    // MFMAs, so the matrix pipe has work from ~100 cycles after the barrier until the end of the tile.
#define GB2_DMA1(n_, tile, slotbase)                                                                       \
    {                                                                                                      \
        float* Ad = smem + (slotbase) + wbase;                                                             \
        float* Wd = Ad + BM * 16;                                                                          \
        const int kk = (tile) * BKE;                                                                       \
        if ((n_) < NJA) GLDS16(rsA, voA, (RPJ * (n_) * lda + kk) * EB, Ad + RPJ * (n_) * 16);              \
        else if ((n_) < NJA + NJW) GLDS16(rsW, voW, (RPJ * ((n_) - NJA) * K + kk) * EB, Wd + RPJ * ((n_) - NJA) * 16); \
    }
#define GB2_PSTEP(b_, DMA_)                                                                                \
    {                                                                                                      \
        if ((b_) + 2 < 8) gb[((b_) + 2) & 7] = *(const f32x4*)(S + b16 + (((b_) + 2) & 7) * 16 * 16);      \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                      \
            acc16[a][b_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                        \
                __builtin_bit_cast(bf16x8, ga[a]), __builtin_bit_cast(bf16x8, gb[b_]), acc16[a][b_], 0, 0, 0); \
        if (DMA_) { GB2_DMA1(b_, dma_tile, dma_slot) }                                                     \
        if ((b_) + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                 \
        if (DMA_) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                 \
    }
#define GB2_COMPUTE_PIPE(slotbase, DMA_)                                                                   \
    {                                                                                                      \
        const float* S = smem + (slotbase);                                                                \
        f32x4 ga[4], gb[8];                                                                                \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) ga[x] = *(const f32x4*)(S + a16 + x * 16 * 16);      \
        gb[0] = *(const f32x4*)(S + b16);                                                                  \
        gb[1] = *(const f32x4*)(S + b16 + 16 * 16);                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                                 \
        GB2_PSTEP(0, DMA_) GB2_PSTEP(1, DMA_) GB2_PSTEP(2, DMA_) GB2_PSTEP(3, DMA_)                        \
        GB2_PSTEP(4, DMA_) GB2_PSTEP(5, DMA_) GB2_PSTEP(6, false) GB2_PSTEP(7, false)                      \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }

    const int nk = K / BKE;          // >= 2
    int s0 = 0, s1 = SLOT, s2 = 2 * SLOT;     // slot of tile kt, kt+1, kt+2 (float offsets), rotated each tile
    GB2_STAGE(0, s0)
    GB2_STAGE(1, s1)
    if constexpr (NJA + NJW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    GB2_BARRIER();
#ifdef GB2_CLOCK_DIAG
    const unsigned long long dg_c0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dg_last = dg_c0, dg_acc[3] = {0, 0, 0};   // cycles in: k-tile body | vmcnt + lgkmcnt wait | barrier
#define GB2_DIAG_STAMP(i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); dg_acc[i] += now - dg_last; dg_last = now; }
#else
#define GB2_DIAG_STAMP(i)
#endif
    int kt = 0;
    if constexpr (PIPE) {
        for (; kt + 2 < nk; ++kt) {
            const int dma_tile = kt + 2, dma_slot = s2;   // s2 held tile kt-1: every wave passed the barrier after reading it
            GB2_COMPUTE_PIPE(s0, true)
            GB2_DIAG_STAMP(0)
            // own DMAs of tile kt+1 landed; the NJA + NJW of tile kt+2 stay in flight
            if constexpr (NJA + NJW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GB2_DIAG_STAMP(1)
            GB2_BARRIER();
            GB2_DIAG_STAMP(2)
            const int tmp = s0; s0 = s1; s1 = s2; s2 = tmp;
        }
        {
            const int dma_tile = 0, dma_slot = 0;
            (void)dma_tile; (void)dma_slot;
            GB2_COMPUTE_PIPE(s0, false)                       // tile nk-2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GB2_BARRIER();
            GB2_COMPUTE_PIPE(s1, false)                       // tile nk-1
        }
    } else {
    for (; kt + 2 < nk; ++kt) {
        GB2_STAGE(kt + 2, s2)                         // s2 held tile kt-1: every wave passed the barrier after reading it
        GB2_COMPUTE(s0)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // own DMAs of tile kt+1 landed; tile kt+2's six stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();
        const int tmp = s0; s0 = s1; s1 = s2; s2 = tmp;
    }
    GB2_COMPUTE(s0)                                   // tile nk-2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();
    GB2_COMPUTE(s1)                                   // tile nk-1
    }
#ifdef GB2_CLOCK_DIAG
    if (threadIdx.x == 0 && P.C2) {   // diagnostic build only: in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz
        unsigned long long* dg = (unsigned long long*)P.C2 + 2 * (blockIdx.x + gridDim.x * blockIdx.z);
        dg[0] = __builtin_amdgcn_s_memtime() - dg_c0;
        dg[1] = __builtin_amdgcn_s_memrealtime() - dg_r0;
        unsigned long long* dx = (unsigned long long*)P.C2 + 2 * gridDim.x * gridDim.z + 3 * (blockIdx.x + gridDim.x * blockIdx.z);
        dx[0] = dg_acc[0]; dx[1] = dg_acc[1]; dx[2] = dg_acc[2];
    }
    const unsigned long long dg_loop_end = __builtin_amdgcn_s_memtime();
#endif
#undef GB2_DIAG_STAMP
#undef GB2_COMPUTE_PIPE
#undef GB2_PSTEP
#undef GB2_DMA1
#undef GB2_COMPUTE
#undef GB2_STAGE
#undef GLDS16
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();                     // every wave is done with the ring: reuse it for the epilogue

    gemm_t256_epilogue<MF16>(args, P, smem, m0, n0, acc, acc16);
#ifdef GB2_CLOCK_DIAG
    if (threadIdx.x == 0 && P.C2) {
        unsigned long long* dy = (unsigned long long*)P.C2 + 5 * gridDim.x * gridDim.z + 2 * (blockIdx.x + gridDim.x * blockIdx.z);
        dy[0] = dg_c0 - dg_entry;                                   // prologue: kernel entry -> main loop
        dy[1] = __builtin_amdgcn_s_memtime() - dg_loop_end;         // epilogue (stores issued, not necessarily landed)
    }
#endif
}

// the bf16 projection kernel: 16x16x32 MFMA shape (+4..9 % over the 32x32x16 shape at equal cycles: higher clock)
__global__ __launch_bounds__(256, 2) void iefvad_gemm_bf16_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_t256_body<T256_BF16_16>(args, smem);
}

// MODE 3 on the 256 x 256 block tile (8 waves, one workgroup per CU)
#define GB3_BM 256
#define GB3_LDS_BYTES (8 * 32 * GB2_EPI_LD * 4)        // the epilogue's eight parking images (135,168 B) > the 3-slot ring (98,304 B)
__global__ __launch_bounds__(512, 2) void iefvad_gemm_bf16_w256_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_t256_body<T256_BF16_16P, GB3_BM>(args, smem);
}

// MODE 3: the same k-tile with the issue order pinned (A/B candidate of tools/gemm_tune_bf16)
__global__ __launch_bounds__(256, 2) void iefvad_gemm_bf16_pipe_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_t256_body<T256_BF16_16P>(args, smem);
}

// the same kernel on the 32x32x16 shape, kept for A/B runs in tools/gemm_tune_bf16
__global__ __launch_bounds__(256, 2) void iefvad_gemm_bf16_m32_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_t256_body<T256_BF16_32>(args, smem);
}

// fp32 operands on the same 128x256 / 3-slot-ring structure (A and W of GemmBProblem then point to fp32 data)
__global__ __launch_bounds__(256, 2) void iefvad_gemm_f32_t256_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_t256_body<T256_F32>(args, smem);
}
