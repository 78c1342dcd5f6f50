// bf16 mode: out_proj + bias + residual + LayerNorm (+ whitening LayerNorm) (/root/reference/model/imf_vad.py:116-117,121-123) as a
// PERSISTENT row-block kernel (round 4).  Same arithmetic as outproj_ln_chain_bf16.h (round 3), whose phase stamps
// (profiles/r03_outproj_chain_phase_stamps.log) showed where a 64-row block's 67 k cycles go: 20 k until the 96 KB activation image
// is in LDS (a workgroup alone on its CU, nothing to overlap the HBM latency with), 20 k main loop, 27 k LayerNorm epilogue with the
// memory system idle.  Here a workgroup walks blocks b, b + G, b + 2 G, ... and the image of the NEXT block is fetched during the
// epilogue of the current one:
//   * the image region is dead once the main loop is done, so the next image goes straight into it by LDS-DMA (buffer_load ... lds:
//     no registers -- the epilogue needs them all -- and no ds_write; the XOR swizzle sits on the per-lane SOURCE address, the LDS
//     side is lane-linear), 12 one-KB pieces per wave;
//   * the epilogue therefore parks its accumulators in a region of its own, 16 rows at a time ([16][772] fp32 = 48 KB: image 96 KB +
//     park 48 KB + affine vectors 15 KB = 159.25 of the 160 KB), each wave normalising two whole rows per quarter with the LayerNorm
//     kernel's own per-row operations (ln_rows<2>);
//     the pieces are issued behind the epilogue's LAST residual request (in-order retirement: anything requested after them waits for them);
//   * at the top of the next block every wave waits for its own DMA pieces by COUNT (vmcnt retires in order: only the weight pieces
//     and residual rows requested after them may still be in flight), then the workgroup barrier, then the first fragment read.
// Products, k order and epilogue arithmetic are those of the round-3 kernel: every output is bit-identical to it and to the
// two-kernel path (tests/test_gpu_bf16.py).
#pragma once
#include "outproj_ln_chain_bf16.h"

#define OP_PARK_ROWS 16
#define OP_IMG_OFF 0
#define OP_PARK_OFF OC_IMG_BYTES                                         // 98,304
#define OP_PARK_BYTES (OP_PARK_ROWS * OC_PARK_LD * 4)                    // 49,408
#define OP_AFF_OFF (OP_PARK_OFF + OP_PARK_BYTES)                         // 147,712
#define OP_LDS_BYTES (OP_AFF_OFF + 5 * IEF_D * 4)                        // 163,072 <= 163,840
#define OP_DMA_PER_WAVE 12                                               // 6144 16-byte chunks / (8 waves x 64 lanes)

// HAS_Y / HAS_YB (fp32 / bf16 output present) are template parameters, and the next image is requested for EVERY block, so that hipcc
// knows how many vector-memory operations follow each residual request: with `if (P.y)` around the stores and `if (nxt < nblk)` around
// the DMA it had to assume none, and its s_waitcnt for the residual rows of the next quarter (vmcnt counts in order, stores included)
// also waited for the acknowledgements of the stores just issued -- and, in the last quarter, for the whole next image.
template <bool HAS_Y, bool HAS_YB>
__global__ __launch_bounds__(512, 2) void iefvad_outproj_ln_pchain_bf16_kernel(OutLnChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    const OutLnChainProblem& P = args.p[blockIdx.y];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int nblk = args.M / OC_BM;
    const bool two = P.g2 != nullptr;

    // De-phasing experiment (IEFVAD_OL_STAGGER = n: every other workgroup starts n x 8128 cycles late): all workgroups start together and
    // a block takes the same time everywhere, so chip-wide the epilogues (all of a block's HBM traffic) coincide with each other.
    if (args.stagger > 0 && (blockIdx.x & 1))
        for (int i = 0; i < args.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    // ---- once per workgroup: bias and the LayerNorms' affine terms -> LDS (5 x 768 floats, as in the round-3 kernel)
    {
        f32x4 aff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int slot = (2 * wave + i) < 15 ? (2 * wave + i) : 14, k = slot / 3;
            const float* sp = k == 0 ? P.bias : k == 1 ? P.g1 : k == 2 ? P.b1 : k == 3 ? (two ? P.g2 : P.g1) : (two ? P.b2 : P.b1);
            aff[i] = *(const f32x4*)(sp + 4 * ((slot % 3) * 64 + lane));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int slot = (2 * wave + i) < 15 ? (2 * wave + i) : 14;
            *(f32x4*)(lds + OP_AFF_OFF + (slot * 64 + lane) * 16) = aff[i];
        }
    }
    // ---- the image by LDS-DMA: piece i of this wave fills LDS chunks L = (12 wave + i) 64 + lane (lane-linear); chunk L is row
    // r = L / 96, physical position pc = L % 96, and holds logical chunk c = (pc & ~15) | ((pc ^ r) & 15) of that row
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)((size_t)args.M * IEF_D * 2), 0x00020000);
    int vo[OP_DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < OP_DMA_PER_WAVE; ++i) {
        const int L = (OP_DMA_PER_WAVE * wave + i) * 64 + lane, r = L / 96, pc = L - r * 96;
        vo[i] = (r * 96 + ((pc & ~15) | ((pc ^ r) & 15))) * 16;
    }
#define OP_IMAGE_DMA(blk_)                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < OP_DMA_PER_WAVE; ++i)                                                                     \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(lds + OP_IMG_OFF + (OP_DMA_PER_WAVE * wave + i) * 1024), \
                                                 16, vo[i], (blk_) * (OC_BM * IEF_D * 2), 0, 0)
    OP_IMAGE_DMA(blockIdx.x);
    __builtin_amdgcn_sched_barrier(0);

    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(P.stream + (size_t)wave * args.wave_stride), 0, (int)args.wave_stride, 0x00020000);
    const int vlane = lane * 16;
#define OP_LOAD(piece_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (int)((piece_) << 10), 0))
    int rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rd[j] = m * (IEF_D * 2) + (((4 * j + q) ^ m) & 15) * 16;
    const float* affl = (const float*)(lds + OP_AFF_OFF) + 4 * lane;      // + 768 i (bias, g1, b1, g2, b2) + 256 j
    float* park = (float*)(lds + OP_PARK_OFF);

#ifdef OC_DIAG
    unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dt0 = 0, dt1;      // image wait | main loop | drain + DMA issue | parks + barriers | LayerNorm + stores | blocks
#define OP_T(i) do { dt1 = __builtin_amdgcn_s_memtime(); dsum[i] += dt1 - dt0; dt0 = dt1; } while (0)
#define OP_T0() do { dt0 = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OP_T(i)
#define OP_T0()
#endif
    // The residual rows of a block's FIRST quarter are requested in the last quarter of the block before it (here for the first block):
    // requested at the top of their own block they sat in front of the weight ring's refills, and a wave's vector-memory operations
    // retire in order -- the ring (six pieces = 400 cycles ahead) stalled for one HBM latency at the start of every main loop.
    f32x4 res[2][2][3];      // [quarter parity][row][16-byte piece]
#define OP_FETCH_RES(m0_, quarter_)                                                                           \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                          \
        const float* rp = P.R + (size_t)((m0_) + OP_PARK_ROWS * (quarter_) + 2 * wave + u) * IEF_D + 4 * lane; \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) res[(quarter_) & 1][u][j] = *(const f32x4*)(rp + 256 * j); \
    }
    OP_FETCH_RES((int)blockIdx.x * OC_BM, 0)
    __builtin_amdgcn_sched_barrier(0);
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int m0 = blk * OC_BM;
        OP_T0();
        // requests of this block: the first weight pieces
        f32x4 rg[OC_DEPTH];
#pragma unroll
        for (int s = 0; s < OC_DEPTH; ++s) rg[s] = OP_LOAD(s);
        __builtin_amdgcn_sched_barrier(0);
        // this wave's image pieces (and the first quarter's residual rows) were requested before those 6 loads
        asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();                        // every wave's pieces have landed (and the affine vectors of the first block)
        OP_T(0);

        f32x4 acc[4][OC_NB];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < OC_NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        int p = 0;
#pragma unroll 1
        for (int k4 = 0; k4 < OC_KT / 4; ++k4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 ga[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) ga[a] = *(const f32x4*)(lds + OP_IMG_OFF + rd[j] + a * (16 * IEF_D * 2) + k4 * 256);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < OC_NB; ++b) {
                    const f32x4 w = rg[b];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, ga[a]), acc[a][b], 0, 0, 0);
                    rg[b] = OP_LOAD(p + OC_DEPTH);
                    __builtin_amdgcn_sched_barrier(0);
                    ++p;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < OC_DEPTH; ++s) asm volatile("" :: "v"(rg[s]));      // the read-ahead (zero pad pieces) lands before its registers are reused
        OP_T(1);
        // No barrier here: the first quarter parks into a region of its own (its readers left it behind the previous block's last
        // barrier), and the barrier that follows those parks is also the one that declares the image dead -- it is overwritten (by the
        // next image's DMA) two quarters later.  The waves leave the main loop up to ~4 k cycles apart (wave 0 wins the weight stream's
        // arbitration); the early ones now park and request their residual rows while the others finish.
        const int nxt = blk + (int)gridDim.x;
#ifndef OP_PROBE_NORES
        // Inside the epilogue the residual rows run TWO quarters ahead of their use (a CU's share of the HBM stream is ~10 bytes per
        // cycle while the chip is busy: one quarter ahead, every quarter waited ~1.9 k cycles for its rows); nothing of it is issued
        // in front of the weight ring.
        OP_FETCH_RES(m0, 1)
#endif
        OP_T(2);

        // ---- epilogue: four quarters of 16 rows; wave w normalises rows 2 w, 2 w + 1 of each quarter
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
            for (int b = 0; b < OC_NB; ++b) *(f32x4*)(park + m * OC_PARK_LD + 96 * wave + 16 * b + 4 * q) = acc[qt][b];
            f32x4 rcur[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 3; ++j) rcur[u][j] = res[qt & 1][u][j];
#ifndef OP_PROBE_NORES      // timing probe (wrong results): the residual rows of quarters 1 - 3 are not fetched
            // (two quarters ahead; with quarters 0 AND 1 requested at the top of the block it was measured SLOWER, 16.6 -> 17.3 ms per step:
            // the weight ring's pieces retire in order behind every HBM request in front of them; TRIED.md)
            if (qt < 2) { OP_FETCH_RES(m0, qt + 2) }     // in flight while this quarter and the next are normalised
#endif
            // ... and the next block's first quarter (behind a workgroup's last block: its own rows again, requested and not used)
            if (qt == 3) { OP_FETCH_RES((nxt < nblk ? nxt : blk) * OC_BM, 0) }
            // The next block's image is requested HERE, behind the last residual request: a wave's vector-memory operations retire
            // in order, so a residual row requested after the 12 KB of image pieces would arrive only when those have landed (the
            // first build issued them at the top of the epilogue and every quarter waited for the image: no overlap left).  Behind it
            // in the queue are only stores and the next block's first requests, which wait for the image anyway.
            // (behind the last block of a workgroup `nxt` lies outside the descriptor's range: the pieces are dropped by the range check)
            if (qt == 2) { OP_IMAGE_DMA(nxt); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !defined(OP_PROBE_NOBAR) || OP_PROBE_NOBAR < 2      // timing probes only (wrong results): -DOP_PROBE_NOBAR=1 drops the four end barriers, =2 the park barriers too
            GB2_BARRIER();                    // all parks of this quarter are complete
#endif
            OP_T(3);
            f32x4 v[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const f32x4 e = *(const f32x4*)(park + (2 * wave + u) * OC_PARK_LD + 4 * lane + 256 * j);
                    v[u][j] = (e + *(const f32x4*)(affl + 256 * j)) + rcur[u][j];
                }
            ln_rows<2>(v, affl + IEF_D - 4 * lane, affl + 2 * IEF_D - 4 * lane, lane, args.eps);       // ln_rows adds 4 lane itself
            if (two) ln_rows<2>(v, affl + 3 * IEF_D - 4 * lane, affl + 4 * IEF_D - 4 * lane, lane, args.eps);
#ifdef OP_PROBE_NOSTORE     // timing probe (wrong results): the normalised rows are not stored
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 3; ++j) asm volatile("" :: "v"(v[u][j]));
#else
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const size_t row = (size_t)(m0 + OP_PARK_ROWS * qt + 2 * wave + u);
                if constexpr (HAS_Y) {
                    float* yp = P.y + row * IEF_D + 4 * lane;
#pragma unroll
                    for (int j = 0; j < 3; ++j) *(f32x4*)(yp + 256 * j) = v[u][j];
                }
                if constexpr (HAS_YB) {
                    bf16_t* yb = P.yb + row * IEF_D + 4 * lane;
#pragma unroll
                    for (int j = 0; j < 3; ++j) *(bf16x4_t*)(yb + 256 * j) = to_bf16x4(v[u][j]);
                }
            }
#endif
            OP_T(4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef OP_PROBE_NOBAR
            GB2_BARRIER();                    // every reader is done with this quarter's parks: they may be overwritten
#endif
            OP_T(3);
        }
#ifdef OC_DIAG
        dsum[5] += 1;
#endif
#undef OP_FETCH_RES
    }
#ifdef OC_DIAG
    if (args.diag && t == 0)
        for (int i = 0; i < 6; ++i) args.diag[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = dsum[i];
#endif
#undef OP_LOAD
#undef OP_IMAGE_DMA
}
