// bf16 mode: attention out_proj + bias + residual + LayerNorm (+ the whitening LayerNorm after the last layer) in one
// kernel (/root/reference/model/imf_vad.py:116-117,121-123).
//
// Unfused, the out_proj epilogue writes y = attn W_o^T + b_o + x (fp32, 3 KB per row and modality) and the LayerNorm
// kernel reads it straight back.  LayerNorm needs whole 768-wide rows, so here a workgroup OWNS rows: block tile 128 rows x
// 768 columns, 8 waves as 2 x 4 of 64 x 192 (48 accumulator tiles of v_mfma_f32_16x16x32_bf16 = 192 registers per lane),
// k-tiles of 32 bf16 through a 2-slot LDS ring (a slot is 128 A rows + 768 W rows x 64 B = 56 KB; three slots do not
// fit), filled by LDS-DMA one k-tile ahead, the seven DMA instructions of a wave spread between its 48 MFMAs; same LDS
// image / swizzle / fragment addressing and the same k order as the other ring kernels, so y is bit-identical to theirs.
// Epilogue, in four passes of 16 rows per wave: the waves park their 16 x 192 accumulator slabs in LDS, and after a
// workgroup barrier each wave takes four whole rows of the pass -- lane l owns columns 4 l + 256 j exactly as in
// iefvad_layernorm_kernel -- adds bias and the fp32 residual row (16-byte coalesced loads), runs ln_row (rowops.h: the
// LayerNorm kernel's own code, twice after the last layer) and stores the fp32 row (the next layer's residual; skipped
// after the last layer) and its bf16 copy (the next projection's operand).  y never exists in memory.
#pragma once
#include "gemm_bf16.h"
#include "rowops.h"

#define OL_BM 128
#define OL_SLOT ((OL_BM + IEF_D) * 16)               // 4-byte units per ring slot: A | W, 64-byte rows (57,344 B)
#define OL_LDS_BYTES (2 * OL_SLOT * 4)               // 114,688 B (the epilogue's parks need 100,352)
#define OL_EPI_LD 196                                // padded row (floats) of a wave's 16 x 192 park (196 = 4 mod 64 banks)

struct OutLnProblem {
    const bf16_t* A;         // attention output [M, 768] bf16
    const bf16_t* W;         // out_proj weight [768, 768] bf16
    const float* bias;       // [768]
    const float* R;          // residual: the layer's fp32 input rows [M, 768]
    const float* g1; const float* b1;   // LayerNorm
    const float* g2; const float* b2;   // whitening LayerNorm (nullable: skip)
    float* y;                // [M, 768] fp32 output, nullable
    bf16_t* yb;              // [M, 768] bf16 output, nullable
};
struct OutLnArgs {
    OutLnProblem p[2];       // one per modality (blockIdx.y)
    int M;                   // multiple of 128
    float eps;
};

__global__ __launch_bounds__(512, 2) void iefvad_outproj_ln_bf16_kernel(OutLnArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = IEF_D, BKE = 32, EB = 2;
    const OutLnProblem& P = args.p[blockIdx.y];
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * OL_BM;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 2, wc = wave & 3;

    // staging: thread t moves the 16-byte chunk (row = (t>>2) + 128 j, slot chunk = t&3); A: one instruction, W: six
    const int srow = t >> 2, sch = t & 3;
    auto swz = [](int row) { return (0xD2 >> (2 * ((row >> 2) & 3))) & 3; };       // gemm_bf16.h, MODE 2 images
    const int ssw = swz(srow);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(P.A + (size_t)m0 * K), 0, OL_BM * K * EB, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)P.W, 0, IEF_D * K * EB, 0x00020000);
    const int vo = srow * K * EB + ((sch ^ ssw) << 4);
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 16 * 16;    // this wave's 16 rows x 64 B, 4-byte units
#define OL_GLDS(rs, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
#define OL_DMA1(n_, tile, slotbase)                                                                        \
    {                                                                                                      \
        float* Dst = smem + (slotbase) + wbase;                                                            \
        const int kk = (tile) * BKE * EB;                                                                  \
        if ((n_) == 0) OL_GLDS(rsA, kk, Dst);                                                              \
        else OL_GLDS(rsW, 128 * ((n_) - 1) * K * EB + kk, Dst + OL_BM * 16 + 128 * ((n_) - 1) * 16);       \
    }

    // 16x16x32 fragments: lane (r16, q16) reads row (16 x + r16), chunk q16 (swizzled)
    const int r16 = lane & 15, q16 = lane >> 4;
    const int f16 = (q16 ^ swz(r16)) << 2;
    const int a16 = (wr * 64 + r16) * 16 + f16;
    const int b16 = OL_BM * 16 + (wc * 192 + r16) * 16 + f16;

    f32x4 acc16[4][12];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 12; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one column tile b: read the fragment two tiles ahead, four MFMAs, at most one LDS-DMA of the next k-tile between them
#define OL_PSTEP(b_, DMA_)                                                                                 \
    {                                                                                                      \
        if ((b_) + 2 < 12) gb[(b_) + 2 < 12 ? (b_) + 2 : 0] = *(const f32x4*)(S + b16 + ((b_) + 2) * 16 * 16); \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                      \
            acc16[a][b_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                        \
                __builtin_bit_cast(bf16x8, ga[a]), __builtin_bit_cast(bf16x8, gb[b_]), acc16[a][b_], 0, 0, 0); \
        if ((DMA_) && (b_) < 7) { OL_DMA1(b_, dma_tile, dma_slot) }                                        \
        if ((b_) + 2 < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                 \
        if ((DMA_) && (b_) < 7) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                 \
    }
#define OL_COMPUTE(slotbase, DMA_)                                                                         \
    {                                                                                                      \
        const float* S = smem + (slotbase);                                                                \
        f32x4 ga[4], gb[12];                                                                               \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) ga[x] = *(const f32x4*)(S + a16 + x * 16 * 16);      \
        gb[0] = *(const f32x4*)(S + b16);                                                                  \
        gb[1] = *(const f32x4*)(S + b16 + 16 * 16);                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                                 \
        OL_PSTEP(0, DMA_) OL_PSTEP(1, DMA_) OL_PSTEP(2, DMA_) OL_PSTEP(3, DMA_) OL_PSTEP(4, DMA_) OL_PSTEP(5, DMA_)   \
        OL_PSTEP(6, DMA_) OL_PSTEP(7, DMA_) OL_PSTEP(8, DMA_) OL_PSTEP(9, DMA_) OL_PSTEP(10, DMA_) OL_PSTEP(11, DMA_) \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }

    constexpr int nk = K / BKE;      // 24
    int cur = 0, nxt = OL_SLOT;
    {
        const int dma_tile = 0, dma_slot = 0;
        OL_DMA1(0, dma_tile, dma_slot) OL_DMA1(1, dma_tile, dma_slot) OL_DMA1(2, dma_tile, dma_slot) OL_DMA1(3, dma_tile, dma_slot)
        OL_DMA1(4, dma_tile, dma_slot) OL_DMA1(5, dma_tile, dma_slot) OL_DMA1(6, dma_tile, dma_slot)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GB2_BARRIER();
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int dma_tile = kt + 1, dma_slot = nxt;     // `nxt` held tile kt-1: every wave passed the barrier after reading it
        OL_COMPUTE(cur, true)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of tile kt+1 landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    {
        const int dma_tile = 0, dma_slot = 0;
        (void)dma_tile; (void)dma_slot;
        OL_COMPUTE(cur, false)                            // tile nk-1
    }
#undef OL_COMPUTE
#undef OL_PSTEP
#undef OL_DMA1
#undef OL_GLDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();                    // every wave is done with the ring: the parks reuse it

    // ---- epilogue: four passes; pass p handles row tile p (16 rows) of every wave, i.e. block rows 16 p .. 16 p + 15 of both
    // 64-row bands.  Wave w then owns rows 4 (w & 3) .. + 3 of band w >> 2, all 768 columns.
    float* Eown = smem + wave * (16 * OL_EPI_LD);
    const float* Eband = smem + (wave & ~3) * (16 * OL_EPI_LD);        // the four parks of this wave's row band
    f32x4 bias3[3];
    int eoff[3];                                                         // where columns 4 lane + 256 j live in the band's parks
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = 4 * lane + 256 * j;
        bias3[j] = *(const f32x4*)(P.bias + c);
        eoff[j] = (c / 192) * (16 * OL_EPI_LD) + (c % 192);
    }
    // the four residual rows of a pass are requested together, right after the pass is parked (its accumulators are free)
    f32x4 rnext[4][3];
#define OL_FETCH_RES(p_)                                                                                   \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                        \
        const float* rp = P.R + (size_t)(m0 + wr * 64 + 16 * (p_) + 4 * (wave & 3) + u) * IEF_D + 4 * lane; \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) rnext[u][j] = *(const f32x4*)(rp + 256 * j);         \
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int b = 0; b < 12; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) Eown[(4 * q16 + r) * OL_EPI_LD + b * 16 + r16] = acc16[p][b][r];
        __builtin_amdgcn_sched_barrier(0);
        OL_FETCH_RES(p)               // in flight while the parks complete and the workgroup meets
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();                // all parks of this pass are complete
        f32x4 rcur[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) rcur[u][j] = rnext[u][j];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = 4 * (wave & 3) + u;                           // row inside the 16-row slab
            const size_t row = (size_t)(m0 + wr * 64 + 16 * p + rr);
            f32x4 v[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f32x4 e = *(const f32x4*)(Eband + rr * OL_EPI_LD + eoff[j]);
                v[j] = (e + bias3[j]) + rcur[u][j];
            }
            ln_row(v, P.g1, P.b1, lane, args.eps);
            if (P.g2 != nullptr) ln_row(v, P.g2, P.b2, lane, args.eps);
            if (P.y) {
                float* yp = P.y + row * IEF_D + 4 * lane;
#pragma unroll
                for (int j = 0; j < 3; ++j) *(f32x4*)(yp + 256 * j) = v[j];
            }
            if (P.yb) {
                bf16_t* yb = P.yb + row * IEF_D + 4 * lane;
#pragma unroll
                for (int j = 0; j < 3; ++j) *(bf16x4_t*)(yb + 256 * j) = to_bf16x4(v[j]);
            }
        }
        if (p < 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GB2_BARRIER();            // every reader is done with this pass's parks: they may be overwritten
        }
    }
#undef OL_FETCH_RES
}
