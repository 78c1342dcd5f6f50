// EXPERIMENT (tools/gemm_tune_bf16 only, not dispatched by the library): the bf16 projection on a 256 x 256 block tile with
// FOUR waves of 128 x 128 -- 64 accumulator tiles of v_mfma_f32_16x16x32_bf16 = 256 accumulator registers per lane, one wave
// per SIMD -- instead of eight waves of 64 x 128.  A k-tile then costs a wave 16 fragment reads for 64 MFMAs (0.25 per
// MFMA against 0.375), i.e. a third less LDS read traffic per FLOP, which is what the vendor library's kernels of this shape
// do.  Same LDS images, ring protocol, k order and epilogue as iefvad_gemm_bf16_w256_kernel: bit-identical results.
#pragma once
#include "gemm_bf16.h"

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
