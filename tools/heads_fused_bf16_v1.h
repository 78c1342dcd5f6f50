// bf16 mode: the mu / log-variance heads of BOTH modalities and the precision-weighted fusion in one kernel
// (/root/reference/model/imf_vad.py:125-144).
//
// Unfused, the heads projection writes four [N, 768] fp32 tensors (12 KB per snippet) that the fusion kernel reads back
// at once; in the scores-only output mode nothing else ever reads them.  Here one workgroup owns, for 256 rows and 64
// output columns, all four quantities of a (row, column): its 256-column tile is
//     [ mu_i c0..c0+63 | logvar_i c0..c0+63 | mu_e c0..c0+63 | logvar_e c0..c0+63 ]
// i.e. the left half multiplies the image rows x_i, the right half the event rows x_e.  The main loop is the 256 x 256 /
// 8-wave / 3-slot-ring loop of iefvad_gemm_bf16_w256_kernel (gemm_bf16.h; same LDS images, swizzle, pinned issue order,
// same summation order per element, so mu / logvar are bit-identical to the unfused projection); a ring slot holds two A
// panels (x_i, x_e: 2 x 16 KB) and the gathered W rows (16 KB): 48 KB, three slots = 144 KB, one workgroup per CU.
// Epilogue: the waves park their tiles in LDS as before; after a workgroup barrier each wave reads mu_i, logvar_i (from
// the image-side wave's park) and mu_e, logvar_e (from the event-side wave's) for 16 rows x 64 columns per pass, adds the
// biases, applies iefvad's fusion formula (rowops.h: fuse_elem, the same operations in the same order as
// iefvad_fusion_kernel) and stores z (+ its bf16 copy) and whichever of mu / logvar / n_i / n_e the caller asked for, 256
// bytes per row and instruction.  Row means of n_i, n_e (test.py:131-136) leave as per-column-block partial sums,
// finished by iefvad_rowmean_finish_kernel in a fixed order (no atomics: bit-reproducible).
#pragma once
#include "gemm_bf16.h"
#include "rowops.h"

#define HF_BM 256
#define HF_COLS 64                                   // output columns per workgroup
#define HF_NBLK (IEF_D / HF_COLS)                    // 12 column blocks
#define HF_SLOT ((2 * HF_BM + 256) * 16)             // 4-byte units per ring slot: A_i | A_e | W, 64-byte rows
#define HF_LDS_BYTES (3 * HF_SLOT * 4)               // 147,456 B (the epilogue's eight parks need 135,168)

struct HeadsFusedArgs {
    const bf16_t* A[2];      // x_i, x_e: [M, 768] bf16 (the LayerNorm kernel's operand copies)
    const bf16_t* W[2];      // head matrices [1536, 768] bf16: rows 0..767 mu, 768..1535 log-variance
    const float* bias[2];    // [1536] each
    float* mu[2];            // [M, 768] fp32, nullable
    float* lv[2];            // [M, 768] fp32, nullable
    float* n[2];             // normalised precision weights n_i, n_e [M, 768], nullable
    float* z;                // fused state [M, 768] fp32
    bf16_t* zb;              // its bf16 copy (operand of the first refinement projection), nullable
    float* nsum_part;        // [M][2][HF_NBLK] partial row sums of n_i, n_e over this workgroup's 64 columns, nullable
    int M;                   // multiple of 256
    float factor, eps;
};

__global__ __launch_bounds__(512, 2) void iefvad_heads_fused_bf16_kernel(HeadsFusedArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = IEF_D, BKE = 32, EB = 2;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / HF_NBLK, tn = bid - tm * HF_NBLK;
    const int m0 = tm * HF_BM, c0 = tn * HF_COLS;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // staging: thread t moves the 16-byte chunk (row = (t>>2) + 128 j, slot chunk = t&3) of each image, j = 0, 1
    const int srow = t >> 2, sch = t & 3;
    auto swz = [](int row) { return (0xD2 >> (2 * ((row >> 2) & 3))) & 3; };       // gemm_bf16.h, MODE 2 images
    const int ssw = swz(srow);
    const auto rsAi = __builtin_amdgcn_make_buffer_rsrc((void*)(args.A[0] + (size_t)m0 * K), 0, HF_BM * K * EB, 0x00020000);
    const auto rsAe = __builtin_amdgcn_make_buffer_rsrc((void*)(args.A[1] + (size_t)m0 * K), 0, HF_BM * K * EB, 0x00020000);
    // W image rows 128 j + srow: j = modality; rows 0..63 of the half are mu columns c0.., rows 64..127 log-variance columns
    const auto rsWi = __builtin_amdgcn_make_buffer_rsrc((void*)(args.W[0] + (size_t)c0 * K), 0, (IEF_D + HF_COLS) * K * EB, 0x00020000);
    const auto rsWe = __builtin_amdgcn_make_buffer_rsrc((void*)(args.W[1] + (size_t)c0 * K), 0, (IEF_D + HF_COLS) * K * EB, 0x00020000);
    const int voA = srow * K * EB + ((sch ^ ssw) << 4);
    const int voW = ((srow >= 64 ? IEF_D : 0) + (srow & 63)) * K * EB + ((sch ^ ssw) << 4);
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 16 * 16;    // this wave's 16 rows x 64 B, 4-byte units
#define HF_GLDS(rs, vo, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
    // the six LDS-DMA instructions of a k-tile: A_i rows 0..127, 128..255, A_e likewise, W of modality i, of modality e
#define HF_DMA1(n_, tile, slotbase)                                                                        \
    {                                                                                                      \
        float* D = smem + (slotbase) + wbase;                                                              \
        const int kk = (tile) * BKE * EB;                                                                  \
        if ((n_) == 0) HF_GLDS(rsAi, voA, kk, D);                                                          \
        else if ((n_) == 1) HF_GLDS(rsAi, voA, 128 * K * EB + kk, D + 128 * 16);                           \
        else if ((n_) == 2) HF_GLDS(rsAe, voA, kk, D + HF_BM * 16);                                        \
        else if ((n_) == 3) HF_GLDS(rsAe, voA, 128 * K * EB + kk, D + HF_BM * 16 + 128 * 16);              \
        else if ((n_) == 4) HF_GLDS(rsWi, voW, kk, D + 2 * HF_BM * 16);                                    \
        else HF_GLDS(rsWe, voW, kk, D + 2 * HF_BM * 16 + 128 * 16);                                        \
    }
#define HF_STAGE(tile, slotbase) \
    { HF_DMA1(0, tile, slotbase) HF_DMA1(1, tile, slotbase) HF_DMA1(2, tile, slotbase) HF_DMA1(3, tile, slotbase) HF_DMA1(4, tile, slotbase) HF_DMA1(5, tile, slotbase) }

    // 16x16x32 fragments: lane (r16, q16) reads row (16 x + r16), chunk q16 (swizzled) of its A image / of the W image
    const int r16 = lane & 15, q16 = lane >> 4;
    const int f16 = (q16 ^ swz(r16)) << 2;
    const int a16 = wc * HF_BM * 16 + (wr * 64 + r16) * 16 + f16;         // wc = 0: image side, 1: event side
    const int b16 = 2 * HF_BM * 16 + (wc * 128 + r16) * 16 + f16;

    f32x4 acc16[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#define HF_PSTEP(b_, DMA_)                                                                                 \
    {                                                                                                      \
        if ((b_) + 2 < 8) gb[((b_) + 2) & 7] = *(const f32x4*)(S + b16 + (((b_) + 2) & 7) * 16 * 16);      \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                      \
            acc16[a][b_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                        \
                __builtin_bit_cast(bf16x8, ga[a]), __builtin_bit_cast(bf16x8, gb[b_]), acc16[a][b_], 0, 0, 0); \
        if (DMA_) { HF_DMA1(b_, dma_tile, dma_slot) }                                                      \
        if ((b_) + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                 \
        if (DMA_) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                 \
    }
#define HF_COMPUTE(slotbase, DMA_)                                                                         \
    {                                                                                                      \
        const float* S = smem + (slotbase);                                                                \
        f32x4 ga[4], gb[8];                                                                                \
        _Pragma("unroll") for (int x = 0; x < 4; ++x) ga[x] = *(const f32x4*)(S + a16 + x * 16 * 16);      \
        gb[0] = *(const f32x4*)(S + b16);                                                                  \
        gb[1] = *(const f32x4*)(S + b16 + 16 * 16);                                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                                 \
        HF_PSTEP(0, DMA_) HF_PSTEP(1, DMA_) HF_PSTEP(2, DMA_) HF_PSTEP(3, DMA_)                            \
        HF_PSTEP(4, DMA_) HF_PSTEP(5, DMA_) HF_PSTEP(6, false) HF_PSTEP(7, false)                          \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }

    constexpr int nk = K / BKE;      // 24
    int s0 = 0, s1 = HF_SLOT, s2 = 2 * HF_SLOT;
    HF_STAGE(0, s0)
    HF_STAGE(1, s1)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    GB2_BARRIER();
    for (int kt = 0; kt + 2 < nk; ++kt) {
        const int dma_tile = kt + 2, dma_slot = s2;      // s2 held tile kt-1: every wave passed the barrier after reading it
        HF_COMPUTE(s0, true)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // own DMAs of tile kt+1 landed; the six of tile kt+2 stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();
        const int tmp = s0; s0 = s1; s1 = s2; s2 = tmp;
    }
    {
        const int dma_tile = 0, dma_slot = 0;
        (void)dma_tile; (void)dma_slot;
        HF_COMPUTE(s0, false)                             // tile nk-2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();
        HF_COMPUTE(s1, false)                             // tile nk-1
    }
#undef HF_COMPUTE
#undef HF_PSTEP
#undef HF_STAGE
#undef HF_DMA1
#undef HF_GLDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();                    // every wave is done with the ring: the parks reuse it

    // ---- epilogue.  Park image of a wave: 32 rows x 128 columns (padded to GB2_EPI_LD): columns 0..63 mu, 64..127 logvar.
    float* Eown = smem + wave * (32 * GB2_EPI_LD);
    const float* Ei = smem + (wave & ~1) * (32 * GB2_EPI_LD);          // the image-side wave of this row band
    const float* Ee = smem + (wave | 1) * (32 * GB2_EPI_LD);           // the event-side wave
    const int c16 = lane & 15, rq = lane >> 4;
    const int col = c0 + 4 * c16;
    const f32x4 bmi = *(const f32x4*)(args.bias[0] + col), bli = *(const f32x4*)(args.bias[0] + IEF_D + col);
    const f32x4 bme = *(const f32x4*)(args.bias[1] + col), ble = *(const f32x4*)(args.bias[1] + IEF_D + col);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Eown[(x * 16 + 4 * q16 + r) * GB2_EPI_LD + b * 16 + r16] = acc16[2 * a + x][b][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();                // both parks of every row band are complete
        // this wave finishes rows 16 wc .. 16 wc + 15 of the pass, all 64 columns: lane (rq, c16) -> rows rq + 4 u, columns 4 c16..+3
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int prow = 16 * wc + rq + 4 * u;
            const size_t o = (size_t)(m0 + wr * 64 + 32 * a + prow) * IEF_D + col;
            const f32x4 mi = *(const f32x4*)(Ei + prow * GB2_EPI_LD + 4 * c16) + bmi;
            const f32x4 li = *(const f32x4*)(Ei + prow * GB2_EPI_LD + 64 + 4 * c16) + bli;
            const f32x4 me = *(const f32x4*)(Ee + prow * GB2_EPI_LD + 4 * c16) + bme;
            const f32x4 le = *(const f32x4*)(Ee + prow * GB2_EPI_LD + 64 + 4 * c16) + ble;
            f32x4 ni, ne, z;
            float si = 0.f, se = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float nie, nee, ze;
                fuse_elem(mi[e], li[e], me[e], le[e], args.factor, args.eps, nie, nee, ze);
                ni[e] = nie; ne[e] = nee; z[e] = ze;
                si += nie;
                se += nee;
            }
            if (args.mu[0]) GB2_STORE((f32x4*)(args.mu[0] + o), mi);
            if (args.lv[0]) GB2_STORE((f32x4*)(args.lv[0] + o), li);
            if (args.mu[1]) GB2_STORE((f32x4*)(args.mu[1] + o), me);
            if (args.lv[1]) GB2_STORE((f32x4*)(args.lv[1] + o), le);
            if (args.n[0]) GB2_STORE((f32x4*)(args.n[0] + o), ni);
            if (args.n[1]) GB2_STORE((f32x4*)(args.n[1] + o), ne);
            *(f32x4*)(args.z + o) = z;                      // read again by the refinement: not a streaming store
            if (args.zb) {
                bf16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (bf16_t)z[e];
                *(bf16x4*)(args.zb + o) = w;
            }
            if (args.nsum_part) {
                // sum over the 16 lanes that share the row (xor 1, 2, 4, 8 stay inside the 16-lane group)
#pragma unroll
                for (int s = 1; s < 16; s <<= 1) { si += __shfl_xor(si, s, 64); se += __shfl_xor(se, s, 64); }
                if (c16 == 0) {
                    float* pp = args.nsum_part + ((size_t)(m0 + wr * 64 + 32 * a + prow) * 2) * HF_NBLK + tn;
                    pp[0] = si;
                    pp[HF_NBLK] = se;
                }
            }
        }
        if (a == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GB2_BARRIER();            // the partner has read this wave's park: it may be overwritten
        }
    }
}

// row means of the normalised weights from the fused kernel's partial sums: fixed order, one thread per row
// (np partials per row and modality: HF_NBLK here, HC_NPART from heads_chain_bf16.h; a multiple of 4).  One thread per (row, modality):
// its np floats are contiguous, neighbouring threads read neighbouring 16-byte vectors.
__global__ __launch_bounds__(256) void iefvad_rowmean_finish_kernel(const float* part, float* n_i_mean, float* n_e_mean, int nrows, int np) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int row = id >> 1, mod = id & 1;
    if (row >= nrows) return;
    float* dst = mod ? n_e_mean : n_i_mean;
    if (!dst) return;
    const float* p = part + ((size_t)row * 2 + mod) * np;
    float s = 0.f;
    for (int b = 0; b < np; b += 4) {
        const f32x4 v = *(const f32x4*)(p + b);
        s += v[0]; s += v[1]; s += v[2]; s += v[3];
    }
    dst[row] = s * (1.0f / IEF_D);
}
