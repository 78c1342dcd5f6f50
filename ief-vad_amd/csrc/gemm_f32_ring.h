// Ring-buffered variant of the fp32 projection GEMM: same 128x128 block tile, 2x2 waves of 64x64, same
// epilogues and the same k summation order as iefvad_gemm_f32_kernel (results are bit-identical), but the
// K dimension is staged in 16-wide tiles through a 4-slot LDS ring (4 x 16 KB = the same 64 KB):
// the LDS-DMA loads of tile t+3 are issued while tile t is computed, and the end-of-tile wait is a COUNTED
// s_waitcnt vmcnt(8) that only retires tile t+1, leaving two tiles in flight across the barrier.  A slow
// DMA (an L2 miss under load) therefore has three tiles to land instead of one.
// LDS image per slot: [row][16 floats], 16-byte chunk index XORed with (row>>2)&3 (conflict-free
// ds_read_b128: a 16-lane group covers four rows x four chunks of a 256-byte bank row).
#pragma once
#include "gemm_f32.h"

#define RING_BK 16
#define RING_STAGES 4
#define RING_SLOT ((GEMM_BM + GEMM_BN) * RING_BK)   // floats per ring slot (A tile then W tile)

__global__ __launch_bounds__(256, 2) void iefvad_gemm_f32_ring_kernel(GemmArgs args) {
    __shared__ __attribute__((aligned(16))) float smem[RING_STAGES * RING_SLOT];   // 64 KB
    const GemmProblem& P = args.p[blockIdx.z];
    const int ntn = args.N / GEMM_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const int K = args.K, lda = args.lda;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;

    // staging: thread t moves chunk (row = (t>>2) + 64 j, slot chunk = t&3), j = 0..1, of both operand tiles;
    // one wave instruction covers 16 rows x 64 B = 1 KB of the lane-linear LDS image
    const int srow = t >> 2, sch = t & 3;
    const int ssw = (srow >> 2) & 3;                      // ((row + 64 j) >> 2) & 3 is j-invariant
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * lda), 0,
                                                       (int)((GEMM_BM - 1) * lda + K) * 4, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (size_t)n0 * K), 0,
                                                       (int)((GEMM_BN - 1) * K + K) * 4, 0x00020000);
    const int voA = (srow * lda + ((sch ^ ssw) << 2)) * 4;
    const int voW = (srow * K + ((sch ^ ssw) << 2)) * 4;
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 16 * RING_BK;   // wave-uniform: its 16 rows
#define GLDS16(rs, vo, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
#define RING_STAGE_TILE(tile, slot)                                                                   \
    {                                                                                                 \
        float* Ad = smem + (slot) * RING_SLOT + wbase;                                                \
        float* Wd = Ad + GEMM_BM * RING_BK;                                                           \
        const int kk = (tile) * RING_BK;                                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
            GLDS16(rsA, voA, (64 * j * lda + kk) * 4, Ad + 64 * j * RING_BK);                         \
            GLDS16(rsW, voW, (64 * j * K + kk) * 4, Wd + 64 * j * RING_BK);                           \
        }                                                                                             \
    }

    const int fsw = (i >> 2) & 3;
    int aoff[2], boff[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        aoff[x] = (wr * 64 + x * 32 + i) * RING_BK;
        boff[x] = GEMM_BM * RING_BK + (wc * 64 + x * 32 + i) * RING_BK;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / RING_BK;    // >= 3 (K = 768 -> 48)
    RING_STAGE_TILE(0, 0)
    RING_STAGE_TILE(1, 1)
    RING_STAGE_TILE(2, 2)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // tile 0 landed (tiles 1, 2 = 8 DMAs still in flight)
    __builtin_amdgcn_s_barrier();

#define RING_COMPUTE(slot)                                                                                      \
    {                                                                                                           \
        const float* S = smem + (slot) * RING_SLOT;                                                             \
        f32x4 fa[2][2], fb[2][2];                                                                               \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                         \
            const int ch = ((2 * s + h) ^ fsw) << 2;                                                            \
            fa[s][0] = *(const f32x4*)(S + aoff[0] + ch);                                                       \
            fa[s][1] = *(const f32x4*)(S + aoff[1] + ch);                                                       \
            fb[s][0] = *(const f32x4*)(S + boff[0] + ch);                                                       \
            fb[s][1] = *(const f32x4*)(S + boff[1] + ch);                                                       \
        }                                                                                                       \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                         \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                     \
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0][e], fb[s][0][e], acc[0][0], 0, 0, 0); \
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][0][e], fb[s][1][e], acc[0][1], 0, 0, 0); \
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1][e], fb[s][0][e], acc[1][0], 0, 0, 0); \
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][1][e], fb[s][1][e], acc[1][1], 0, 0, 0); \
            }                                                                                                   \
        }                                                                                                       \
    }

    int slot = 0;
    int kt = 0;
    for (; kt + 3 < nk; ++kt) {
        RING_STAGE_TILE(kt + 3, (slot + 3) & 3)          // that slot held tile kt-1: every wave is past it
        __builtin_amdgcn_sched_barrier(0);
        RING_COMPUTE(slot)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); // own DMAs of tile kt+1 done; kt+2, kt+3 stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = (slot + 1) & 3;
    }
    // drain: tiles nk-3, nk-2, nk-1 (no more staging)
    RING_COMPUTE(slot)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    slot = (slot + 1) & 3;
    RING_COMPUTE(slot)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    slot = (slot + 1) & 3;
    f32x16 res[2][2];
    gemm_prefetch_residual(args, P, res, m0, n0, wr, wc, i, h);
    RING_COMPUTE(slot)
#undef RING_COMPUTE
#undef RING_STAGE_TILE
#undef GLDS16
    gemm_epilogue(args, P, acc, res, m0, n0, wr, wc, i, h);
}
