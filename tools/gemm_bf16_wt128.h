// EXPERIMENT (tools/gemm_tune_bf16 only, not dispatched by the library): the bf16 projection on a 256 x 256 block tile with
// FOUR waves of 128 x 128 -- 64 accumulator tiles of v_mfma_f32_16x16x32_bf16 = 256 accumulator registers per lane, one wave
// per SIMD -- instead of eight waves of 64 x 128.  A k-tile then costs a wave 16 fragment reads for 64 MFMAs (0.25 per
// MFMA against 0.375), i.e. a third less LDS read traffic per FLOP, which is what the vendor library's kernels of this shape
// do.  Same LDS images, ring protocol, k order and epilogue as iefvad_gemm_bf16_w256_kernel: bit-identical results.
#pragma once
#include "../ief-vad_amd/csrc/gemm_bf16.h"

#define GW_BM 256
#define GW_BN 256
#define GW_SLOT ((GW_BM + GW_BN) * 16)               // 4-byte units per ring slot (32 KB)
#define GW_LDS_BYTES (3 * GW_SLOT * 4)               // 98,304 B (the four epilogue parks need 67,584)

__global__ __launch_bounds__(256, 1) void iefvad_gemm_bf16_wt128_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BKE = 32, EB = 2;
    const GemmBProblem& P = args.p[blockIdx.z];
    const int ntn = args.N / GW_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * GW_BM, n0 = tn * GW_BN;
    const int K = args.K, lda = args.lda;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // staging: one instruction of the workgroup = 64 rows x 64 B; A: rows srow + 64 j (j < 4), W likewise
    const int srow = t >> 2, sch = t & 3;
    auto swz = [](int row) { return (0xD2 >> (2 * ((row >> 2) & 3))) & 3; };
    const int ssw = swz(srow);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * lda), 0, (int)((GW_BM - 1) * lda + K) * EB, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (size_t)n0 * K), 0, (int)((GW_BN - 1) * K + K) * EB, 0x00020000);
    const int voA = srow * lda * EB + ((sch ^ ssw) << 4);
    const int voW = srow * K * EB + ((sch ^ ssw) << 4);
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 16 * 16;
#define GW_GLDS(rs, vo, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
#define GW_DMA1(n_, tile, slotbase)                                                                        \
    {                                                                                                      \
        float* Dst = smem + (slotbase) + wbase;                                                            \
        const int kk = (tile) * BKE;                                                                       \
        if ((n_) < 4) GW_GLDS(rsA, voA, (64 * (n_) * lda + kk) * EB, Dst + 64 * (n_) * 16);                \
        else GW_GLDS(rsW, voW, (64 * ((n_) - 4) * K + kk) * EB, Dst + GW_BM * 16 + 64 * ((n_) - 4) * 16);  \
    }

    const int r16 = lane & 15, q16 = lane >> 4;
    const int f16 = (q16 ^ swz(r16)) << 2;
    const int a16 = (wr * 128 + r16) * 16 + f16;
    const int b16 = GW_BM * 16 + (wc * 128 + r16) * 16 + f16;

    f32x4 acc16[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#define GW_PSTEP(b_, DMA_)                                                                                 \
    {                                                                                                      \
        if ((b_) + 2 < 8) gb[((b_) + 2) & 7] = *(const f32x4*)(S + b16 + (((b_) + 2) & 7) * 16 * 16);      \
        _Pragma("unroll") for (int a = 0; a < 8; ++a)                                                      \
            acc16[a][b_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                        \
                __builtin_bit_cast(bf16x8, ga[a]), __builtin_bit_cast(bf16x8, gb[b_]), acc16[a][b_], 0, 0, 0); \
        if (DMA_) { GW_DMA1(b_, dma_tile, dma_slot) }                                                      \
        if ((b_) + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                 \
        if (DMA_) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                 \
    }
#define GW_COMPUTE(slotbase, DMA_)                                                                         \
    {                                                                                                      \
        const float* S = smem + (slotbase);                                                                \
        f32x4 ga[8], gb[8];                                                                                \
        _Pragma("unroll") for (int x = 0; x < 8; ++x) ga[x] = *(const f32x4*)(S + a16 + x * 16 * 16);      \
        gb[0] = *(const f32x4*)(S + b16);                                                                  \
        gb[1] = *(const f32x4*)(S + b16 + 16 * 16);                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);                                                \
        GW_PSTEP(0, DMA_) GW_PSTEP(1, DMA_) GW_PSTEP(2, DMA_) GW_PSTEP(3, DMA_)                            \
        GW_PSTEP(4, DMA_) GW_PSTEP(5, DMA_) GW_PSTEP(6, DMA_) GW_PSTEP(7, DMA_)                            \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }

    const int nk = K / BKE;
    int s0 = 0, s1 = GW_SLOT, s2 = 2 * GW_SLOT;
    {
        const int dma_tile = 0, dma_slot = s0;
        GW_DMA1(0, dma_tile, dma_slot) GW_DMA1(1, dma_tile, dma_slot) GW_DMA1(2, dma_tile, dma_slot) GW_DMA1(3, dma_tile, dma_slot)
        GW_DMA1(4, dma_tile, dma_slot) GW_DMA1(5, dma_tile, dma_slot) GW_DMA1(6, dma_tile, dma_slot) GW_DMA1(7, dma_tile, dma_slot)
    }
    {
        const int dma_tile = 1, dma_slot = s1;
        GW_DMA1(0, dma_tile, dma_slot) GW_DMA1(1, dma_tile, dma_slot) GW_DMA1(2, dma_tile, dma_slot) GW_DMA1(3, dma_tile, dma_slot)
        GW_DMA1(4, dma_tile, dma_slot) GW_DMA1(5, dma_tile, dma_slot) GW_DMA1(6, dma_tile, dma_slot) GW_DMA1(7, dma_tile, dma_slot)
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    GB2_BARRIER();
    for (int kt = 0; kt + 2 < nk; ++kt) {
        const int dma_tile = kt + 2, dma_slot = s2;
        GW_COMPUTE(s0, true)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();
        const int tmp = s0; s0 = s1; s1 = s2; s2 = tmp;
    }
    {
        const int dma_tile = 0, dma_slot = 0;
        (void)dma_tile; (void)dma_slot;
        GW_COMPUTE(s0, false)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();
        GW_COMPUTE(s1, false)
    }
#undef GW_COMPUTE
#undef GW_PSTEP
#undef GW_DMA1
#undef GW_GLDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();
    f32x16 unused[4][4];
    gemm_wave_epilogue<true, 4>(args, P, smem, m0, n0, wr * 128, wc * 128, unused, acc16);
}

// ------------------------------------------------------------------------------------------------------------
// EXPERIMENT 2: the eight-wave 256 x 256 kernel (iefvad_gemm_bf16_w256_kernel) with the k-tile's ONE barrier moved to the
// middle of the tile and the next tile's first fragments prefetched behind it.  In the production loop a wave that leaves
// the end-of-tile barrier has to issue six fragment reads and wait out their latency before its first MFMA (in-kernel
// stamps: 558 of a k-tile's 1444 cycles are the barrier and what follows it).  Here, per tile kt:
//   steps 0..3   16 MFMAs on fragments already in registers (ga: prefetched during tile kt-1; gb streamed two ahead)
//   mid          s_waitcnt vmcnt(0): this wave's pieces of tile kt+1 (issued during tile kt-1) have landed;  s_barrier
//                -> every wave is past the first half of tile kt: tile kt+1 is complete in LDS, the slot of tile kt-1 is free
//   steps 4..7   16 MFMAs; the four LDS-DMA instructions of tile kt+2 (into the slot of tile kt-1), one per step; in steps
//                6 and 7 the reads of tile kt+1's ga[0..3], gb[0..1]
// Still one barrier per k-tile; RAW: a fragment of tile kt+1 is read only after the mid barrier of tile kt, which every
// wave passes after its vmcnt(0); WAR: the slot of tile kt-1 is refilled only after the mid barrier of tile kt, and a wave
// at that barrier has consumed every fragment of tile kt-1.
__global__ __launch_bounds__(512, 2) void iefvad_gemm_bf16_w256q_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BKE = 32, EB = 2, BM = 256;
    constexpr int SLOT = (BM + GB2_BN) * 16;
    const GemmBProblem& P = args.p[blockIdx.z];
    const int ntn = args.N / GB2_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * BM, n0 = tn * GB2_BN;
    const int K = args.K, lda = args.lda;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int srow = t >> 2, sch = t & 3;                 // 128 rows per staging instruction of the workgroup
    auto swz = [](int row) { return (0xD2 >> (2 * ((row >> 2) & 3))) & 3; };
    const int ssw = swz(srow);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * lda), 0, (int)((BM - 1) * lda + K) * EB, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(P.W + (size_t)n0 * K), 0, (int)((GB2_BN - 1) * K + K) * EB, 0x00020000);
    const int voA = srow * lda * EB + ((sch ^ ssw) << 4);
    const int voW = srow * K * EB + ((sch ^ ssw) << 4);
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 16 * 16;
#define GQ_GLDS(rs, vo, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
#define GQ_DMA1(n_, tile, slotbase)                                                                        \
    {                                                                                                      \
        float* Dst = smem + (slotbase) + wbase;                                                            \
        const int kk = (tile) * BKE;                                                                       \
        if ((n_) < 2) GQ_GLDS(rsA, voA, (128 * (n_) * lda + kk) * EB, Dst + 128 * (n_) * 16);              \
        else GQ_GLDS(rsW, voW, (128 * ((n_) - 2) * K + kk) * EB, Dst + BM * 16 + 128 * ((n_) - 2) * 16);   \
    }
    const int r16 = lane & 15, q16 = lane >> 4;
    const int f16 = (q16 ^ swz(r16)) << 2;
    const int a16 = (wr * 64 + r16) * 16 + f16;
    const int b16 = BM * 16 + (wc * 128 + r16) * 16 + f16;

    f32x4 acc16[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 ga[4], gb[8], gan[4], gbn[2];
#define GQ_MFMA4(b_)                                                                                       \
    _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                          \
        acc16[a][b_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                            \
            __builtin_bit_cast(bf16x8, ga[a]), __builtin_bit_cast(bf16x8, gb[b_]), acc16[a][b_], 0, 0, 0);
    // first half: steps 0..3, each reads the B fragment two steps ahead
#define GQ_STEP_A(b_)                                                                                      \
    {                                                                                                      \
        gb[(b_) + 2] = *(const f32x4*)(S + b16 + ((b_) + 2) * 16 * 16);                                    \
        GQ_MFMA4(b_)                                                                                       \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                 \
    }
    // second half: steps 4..7: B fragments 6, 7 (steps 4, 5), one LDS-DMA of tile kt+2 each, the next tile's first fragments
#define GQ_STEP_B(b_, DMA_, PRE_)                                                                          \
    {                                                                                                      \
        if ((b_) + 2 < 8) gb[((b_) + 2) & 7] = *(const f32x4*)(S + b16 + (((b_) + 2) & 7) * 16 * 16);      \
        if ((PRE_) && (b_) == 6) { gan[0] = *(const f32x4*)(Sn + a16); gan[1] = *(const f32x4*)(Sn + a16 + 256); gan[2] = *(const f32x4*)(Sn + a16 + 512); } \
        if ((PRE_) && (b_) == 7) { gan[3] = *(const f32x4*)(Sn + a16 + 768); gbn[0] = *(const f32x4*)(Sn + b16); gbn[1] = *(const f32x4*)(Sn + b16 + 256); } \
        GQ_MFMA4(b_)                                                                                       \
        if (DMA_) { GQ_DMA1((b_) - 4, dma_tile, dma_slot) }                                                \
        if ((b_) + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
        if ((PRE_) && (b_) >= 6) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                 \
        if (DMA_) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                 \
    }
#define GQ_TILE(cur_, nxt_, DMA_, MID_)                                                                    \
    {                                                                                                      \
        const float* S = smem + (cur_);                                                                    \
        const float* Sn = smem + (nxt_);                                                                   \
        (void)Sn;                                                                                          \
        GQ_STEP_A(0) GQ_STEP_A(1) GQ_STEP_A(2) GQ_STEP_A(3)                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (MID_) {                                                                                        \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
            GB2_BARRIER();                                                                                 \
        }                                                                                                  \
        GQ_STEP_B(4, DMA_, MID_) GQ_STEP_B(5, DMA_, MID_) GQ_STEP_B(6, DMA_, MID_) GQ_STEP_B(7, DMA_, MID_) \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (MID_) {                                                                                        \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) ga[x] = gan[x];                                  \
            gb[0] = gbn[0]; gb[1] = gbn[1];                                                                \
        }                                                                                                  \
    }

    const int nk = K / BKE;          // >= 3
    int s0 = 0, s1 = SLOT, s2 = 2 * SLOT;
    {
        const int dma_tile = 0, dma_slot = s0;
        GQ_DMA1(0, dma_tile, dma_slot) GQ_DMA1(1, dma_tile, dma_slot) GQ_DMA1(2, dma_tile, dma_slot) GQ_DMA1(3, dma_tile, dma_slot)
    }
    {
        const int dma_tile = 1, dma_slot = s1;
        GQ_DMA1(0, dma_tile, dma_slot) GQ_DMA1(1, dma_tile, dma_slot) GQ_DMA1(2, dma_tile, dma_slot) GQ_DMA1(3, dma_tile, dma_slot)
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // tile 0 landed (tile 1 may still be in flight)
    GB2_BARRIER();
    {
        const float* S = smem + s0;
#pragma unroll
        for (int x = 0; x < 4; ++x) ga[x] = *(const f32x4*)(S + a16 + x * 256);
        gb[0] = *(const f32x4*)(S + b16);
        gb[1] = *(const f32x4*)(S + b16 + 256);
    }
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
        const int dma_tile = kt + 2, dma_slot = s2;
        GQ_TILE(s0, s1, true, true)
        const int tmp = s0; s0 = s1; s1 = s2; s2 = tmp;
    }
    {
        const int dma_tile = 0, dma_slot = 0;
        (void)dma_tile; (void)dma_slot;
        GQ_TILE(s0, s1, false, true)                      // tile nk-2: its mid barrier publishes tile nk-1
        GQ_TILE(s1, s1, false, false)                     // tile nk-1
    }
#undef GQ_TILE
#undef GQ_STEP_B
#undef GQ_STEP_A
#undef GQ_MFMA4
#undef GQ_DMA1
#undef GQ_GLDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();                    // every wave is done with the ring: the epilogue parks reuse it
    f32x16 unused[2][4];
    gemm_wave_epilogue<true, 2>(args, P, smem, m0, n0, wr * 64, wc * 128, unused, acc16);
}
