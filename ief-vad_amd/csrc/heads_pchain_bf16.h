// bf16 mode: the mu / log-variance heads of both modalities and the precision-weighted fusion
// (/root/reference/model/imf_vad.py:125-144) as a PERSISTENT row-block kernel (round 4).  Same arithmetic as heads_chain_bf16.h
// (round 3), whose stamps (profiles/r03_rowblock_phase_stamps.log) put 24 k of a block's 62 k cycles into waiting for its two 96 KB
// activation images (x_i at entry, x_e between the phases), a workgroup alone on its CU with nothing to overlap the HBM latency with.
// Here a workgroup keeps its column third and walks the 64-row blocks b, b + G, ...; all image traffic is LDS-DMA (buffer_load ... lds:
// no registers, no ds_write; the XOR swizzle sits on the per-lane SOURCE address) and runs under other work:
//   region A (96 KB)  x_i of the block, whole; later its first 32 KB hold the last third of x_e's k range (k >= 512)
//   region B (64 KB)  x_e of the block, k < 512                                    A + B = all 160 KB of the CU's LDS
//   * during the fusion epilogue of block n (registers only): x_i of block n + 1 -> A, x_e[k < 512] of block n + 1 -> B;
//   * after phase 1 (x_i dead): x_e[k >= 512] -> A[0 : 32 KB], while phase 2 already multiplies k < 512 out of B.  A wave's
//     vector-memory operations retire in order, so once its weight ring consumes a piece requested after those four DMA pieces they
//     have landed; a workgroup barrier in front of k = 512 makes every wave's pieces visible to all.
// Products, k order, fuse_elem and the stores are the round-3 kernel's: every output is bit-identical to it and to the unfused path.
#pragma once
#include "heads_chain_bf16.h"

#define HP_A_OFF 0
#define HP_B_OFF OC_IMG_BYTES                         // 98,304
#define HP_LDS_BYTES (OC_IMG_BYTES + 64 * 512 * 2)    // 163,840: the whole LDS of a CU
#define HP_DMA_A 12                                   // 1 KB pieces per wave: 64 rows x 96 chunks of 16 B
#define HP_DMA_B 8                                    //                        64 rows x 64 chunks (k < 512)
#define HP_DMA_C 4                                    //                        64 rows x 32 chunks (k >= 512)

__global__ __launch_bounds__(512, 2) void iefvad_heads_pchain_bf16_kernel(HeadsChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int c3 = blockIdx.y;
    const int nblk = args.M / HC_BM;

    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(args.stream + (size_t)(8 * c3 + wave) * args.wave_stride), 0, (int)args.wave_stride, 0x00020000);
    const int vlane = lane * 16;
#define HP_LOAD(piece_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (int)((piece_) << 10), 0))
    const auto rsI = __builtin_amdgcn_make_buffer_rsrc((void*)args.A[0], 0, (int)((size_t)args.M * IEF_D * 2), 0x00020000);
    const auto rsE = __builtin_amdgcn_make_buffer_rsrc((void*)args.A[1], 0, (int)((size_t)args.M * IEF_D * 2), 0x00020000);
    // per-lane source offsets of the three image shapes: LDS chunk L (lane-linear) is row r = L / W, physical position pc = L % W of a
    // W-chunk row, and holds logical chunk c0 + ((pc & ~15) | ((pc ^ r) & 15)) of the source row (96 chunks of 16 B)
    int voA[HP_DMA_A], voB[HP_DMA_B], voC[HP_DMA_C];
#pragma unroll
    for (int i = 0; i < HP_DMA_A; ++i) {
        const int L = (HP_DMA_A * wave + i) * 64 + lane, r = L / 96, pc = L - r * 96;
        voA[i] = (r * 96 + ((pc & ~15) | ((pc ^ r) & 15))) * 16;
    }
#pragma unroll
    for (int i = 0; i < HP_DMA_B; ++i) {
        const int L = (HP_DMA_B * wave + i) * 64 + lane, r = L >> 6, pc = L & 63;
        voB[i] = (r * 96 + ((pc & ~15) | ((pc ^ r) & 15))) * 16;
    }
#pragma unroll
    for (int i = 0; i < HP_DMA_C; ++i) {
        const int L = (HP_DMA_C * wave + i) * 64 + lane, r = L >> 5, pc = L & 31;
        voC[i] = (r * 96 + 64 + ((pc & ~15) | ((pc ^ r) & 15))) * 16;
    }
#define HP_DMA(rs_, vo_, n_, off_, blk_)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < (n_); ++i)                                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(lds + (off_) + ((n_) * wave + i) * 1024), \
                                                 16, vo_[i], (blk_) * (HC_BM * IEF_D * 2), 0, 0)
    // fragment read offsets: row m of a 16-row tile, chunk (4 j + q) of a 16-chunk group, per row pitch of the three shapes
    int rdA[4], rdB[4], rdC[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sw = (((4 * j + q) ^ m) & 15) * 16;
        rdA[j] = HP_A_OFF + m * 1536 + sw;
        rdB[j] = HP_B_OFF + m * 1024 + sw;
        rdC[j] = HP_A_OFF + m * 512 + sw;
    }
    const int cbase = 256 * c3 + 32 * wave + 4 * q;
    // the biases of this lane's columns, once per workgroup and BEFORE any image request: a load issued behind 20 KB of DMA pieces
    // in the wave's in-order queue would hold the epilogue until the whole image has landed
    f32x4 bmi[2], bli[2], bme[2], ble[2];
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        const int col = cbase + 16 * tile;
        bmi[tile] = *(const f32x4*)(args.bias[0] + col); bli[tile] = *(const f32x4*)(args.bias[0] + IEF_D + col);
        bme[tile] = *(const f32x4*)(args.bias[1] + col); ble[tile] = *(const f32x4*)(args.bias[1] + IEF_D + col);
    }

    HP_DMA(rsI, voA, HP_DMA_A, HP_A_OFF, blockIdx.x);
    HP_DMA(rsE, voB, HP_DMA_B, HP_B_OFF, blockIdx.x);
    __builtin_amdgcn_sched_barrier(0);
#ifdef HC_DIAG      // phase sums per workgroup: top wait | phase 1 | barrier + DMA of x_e's last third | phase 2 (k < 512) | k = 512 barrier | phase 2 (k >= 512) | drain + barrier + next images' DMA issue | fusion epilogue
    unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dt0 = 0, dt1;
#define HP_T(i) do { dt1 = __builtin_amdgcn_s_memtime(); dsum[i] += dt1 - dt0; dt0 = dt1; } while (0)
#define HP_T0() do { dt0 = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HP_T(i)
#define HP_T0()
#endif

    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int m0 = blk * HC_BM;
        HP_T0();
        f32x4 rg[HC_DEPTH];
#pragma unroll
        for (int s = 0; s < HC_DEPTH; ++s) rg[s] = HP_LOAD(s);
        __builtin_amdgcn_sched_barrier(0);
        // this wave's image pieces (A and B) were requested before those 8 loads
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        GB2_BARRIER();
        HP_T(0);

        // acc[phase][2 hd + tile][a]: lane (m, q) holds row 16 a + m, columns 256 c3 + 32 wave + 16 tile + 4 q .. + 3
        f32x4 acc[2][4][4];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[ph][b][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        int p = 0;
#define HP_MAIN(ph_, rd_, pitch16_, k4lo_, k4hi_, k4base_, hook_k4_, hook_)                                        \
    _Pragma("unroll 1") for (int k4 = (k4lo_); k4 < (k4hi_); ++k4) {                                                \
        if (k4 == (hook_k4_)) { hook_ }                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
            f32x4 ga[4];                                                                                            \
            _Pragma("unroll") for (int a = 0; a < 4; ++a) ga[a] = *(const f32x4*)(lds + rd_[j] + a * (pitch16_) + (k4 - (k4base_)) * 256); \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                         \
                const f32x4 w = rg[(j & 1) * 4 + b];                                                                \
                _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                       \
                    acc[ph_][b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, ga[a]), acc[ph_][b][a], 0, 0, 0); \
                rg[(j & 1) * 4 + b] = HP_LOAD(p + HC_DEPTH);                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
                ++p;                                                                                                \
            }                                                                                                       \
        }                                                                                                           \
    }
        HP_MAIN(0, rdA, 16 * 1536, 0, OC_KT / 4, 0, -1, )
        HP_T(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();                        // every wave is done reading x_i: region A may be overwritten
        // (moving this barrier behind the first four k-steps of phase 2 changes nothing, 55.3 -> 55.0 k cycles per block: the wait is not
        // idle time -- wave 0 wins the weight stream's arbitration and the CU's 64 B per cycle serve the other waves meanwhile)
        HP_DMA(rsE, voC, HP_DMA_C, HP_A_OFF, blk);
        __builtin_amdgcn_sched_barrier(0);
        HP_T(2);
        HP_MAIN(1, rdB, 16 * 1024, 0, 4, 0, -1, )      // k < 512 out of region B
        HP_T(3);
        // the four DMA pieces above are older than every weight piece this wave has consumed in the last 14 k-steps: they have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();                        // ... in every wave
        HP_T(4);
        HP_MAIN(1, rdC, 16 * 512, 4, OC_KT / 4, 4, -1, )
        HP_T(5);
#undef HP_MAIN
#pragma unroll
        for (int s = 0; s < HC_DEPTH; ++s) asm volatile("" :: "v"(rg[s]));      // the read-ahead (zero pad pieces) lands before its registers are reused
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();                        // every wave is done with both regions
        const int nxt = blk + (int)gridDim.x;
        if (nxt < nblk) {
            HP_DMA(rsI, voA, HP_DMA_A, HP_A_OFF, nxt);
            HP_DMA(rsE, voB, HP_DMA_B, HP_B_OFF, nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
        HP_T(6);

        // ---- epilogue, in registers (heads_chain_bf16.h's, operation for operation)
        float si[4] = {0.f, 0.f, 0.f, 0.f}, se[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
            const int col = cbase + 16 * tile;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const size_t o = (size_t)(m0 + 16 * a + m) * IEF_D + col;
                const f32x4 mi = acc[0][tile][a] + bmi[tile];
                const f32x4 li = acc[0][2 + tile][a] + bli[tile];
                const f32x4 me = acc[1][tile][a] + bme[tile];
                const f32x4 le = acc[1][2 + tile][a] + ble[tile];
                f32x4 ni, ne, z;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float nie, nee, ze;
                    fuse_elem(mi[e], li[e], me[e], le[e], args.factor, args.eps, nie, nee, ze);
                    ni[e] = nie; ne[e] = nee; z[e] = ze;
                    si[a] += nie;
                    se[a] += nee;
                }
                if (args.mu[0]) GB2_STORE((f32x4*)(args.mu[0] + o), mi);
                if (args.lv[0]) GB2_STORE((f32x4*)(args.lv[0] + o), li);
                if (args.mu[1]) GB2_STORE((f32x4*)(args.mu[1] + o), me);
                if (args.lv[1]) GB2_STORE((f32x4*)(args.lv[1] + o), le);
                if (args.n[0]) GB2_STORE((f32x4*)(args.n[0] + o), ni);
                if (args.n[1]) GB2_STORE((f32x4*)(args.n[1] + o), ne);
                *(f32x4*)(args.z + o) = z;
                if (args.zb) *(bf16x4_t*)(args.zb + o) = to_bf16x4(z);
            }
        }
        if (args.nsum_part) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float x = si[a], y = se[a];
                x += __shfl_xor(x, 16, 64); y += __shfl_xor(y, 16, 64);
                x += __shfl_xor(x, 32, 64); y += __shfl_xor(y, 32, 64);
                if (q == 0) {
                    float* pp = args.nsum_part + ((size_t)(m0 + 16 * a + m) * 2) * HC_NPART + 8 * c3 + wave;
                    pp[0] = x;
                    pp[HC_NPART] = y;
                }
            }
        }
        HP_T(7);
    }
#ifdef HC_DIAG
    if (args.diag && t == 0)
        for (int i = 0; i < 8; ++i) args.diag[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = dsum[i];
#endif
#undef HP_LOAD
#undef HP_DMA
}
