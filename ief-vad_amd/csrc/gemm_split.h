// fp32-accurate projection GEMM on the bf16 matrix cores ("split" mode).
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T + bias[N] ),  A fp32 in HBM, W pre-split into three bf16 planes.
//
// MI355X multiplies bf16 16x faster than fp32 (2.5 PFLOP/s vs 157 TFLOP/s dense), so an fp32 product is cheaper as a
// sum of bf16 products than as one v_mfma_f32_32x32x2_f32.  Every fp32 value is the exact sum of three bf16 values
// obtained by round-to-nearest of the running remainder,
//       x = x1 + x2 + x3,   x1 = bf16(x),  x2 = bf16(x - x1),  x3 = bf16(x - x1 - x2)
// (|x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|, and what is left after x3 has at most 8 significant bits: it IS x3; the
// subtractions are exact in fp32), and a bf16 x bf16 product is exact in the fp32 accumulator.  Of the nine partial
// products the kernel keeps the six of relative size up to 2^-16,
//       a3 w1 + a1 w3 + a2 w2 + a2 w1 + a1 w2 + a1 w1          (accumulated in that order, smallest first),
// and drops a2 w3 + a3 w2 + a3 w3 <= 2^-23 |a w| in the worst case (both remainders at their half-ulp bound), ~2^-27 |a w|
// typically -- against the up to 2^-24 of the RUNNING SUM that every fp32 accumulation step rounds away.  What dominates
// is fp32 accumulation error, the same kind the fp32 MFMA path (and the reference's CPU GEMM) has; measured against
// fp64 the split kernel's error is below the fp32 MFMA kernel's (tools/gemm_tune_split, tests/test_gpu_bf16x6.py); the parity
// tests hold this mode to the SAME gates as the fp32 mode (tests/helpers.py TOL_*).  Non-finite inputs differ:
// inf - inf in the remainder turns an inf operand into NaN (the fp32 path would propagate inf).
//
// Two configurations of one template body (NA = 16-row tiles per wave):
//   NA = 4  128 x 256 block tile, 4 waves as 2 x 2 of 64 x 128 (32 accumulators of v_mfma_f32_16x16x32_bf16), ONE
//           workgroup per CU (one wave per SIMD, ~330 of its 512 VGPR + AGPR); a k-tile (k = 32) is 192 MFMAs = 3072
//           cycles per wave.  LDS 128 KB: 2 x A slot [128 rows][128 B fp32] + 2 x W slot 3 planes x [256 rows][64 B bf16]
//   NA = 2  128 x 128 block tile, 4 waves as 4 x 1 of 32 x 128 (16 accumulators), TWO workgroups per CU (80 KB LDS, <= 256
//           registers): one workgroup's prologue, barrier waits and epilogue (every CU storing at once is an HBM
//           burst) run under the other's MFMAs; each wave splits only its own 32 rows (no duplicated split work).
// Common: a one-tile-ahead double buffer filled by LDS-DMA.
//   A is staged as fp32 (no extra HBM pass, producers keep writing fp32) and split in registers: the fragments of
//   k-tile t+1 are read and split (11 VALU per element pair: v_cvt_pk_bf16_f32, shift / mask, exact subtractions) while
//   the MFMAs of k-tile t run.  The A ring therefore runs one tile ahead of the W ring.
//   W planes are split once, at iefvad_set_weights (iefvad_split_planes_kernel).
//   LDS reads per wave and tile (NA = 4): 8 KB (A) + 24 KB (W) for 3072 MFMA cycles (the plain bf16 kernel: 12 KB per 512).
// With one or two waves per SIMD little hides a stall of a wave's in-order instruction stream, so the issue ORDER is
// pinned with sched_group_barrier: every non-MFMA instruction of a step (split VALU, LDS reads, LDS-DMA) sits between two
// MFMAs -- an MFMA leaves 8 of its 16 issue cycles free (MI355X_MICROARCH.md, vector-instruction issue cost).
// Swizzles (ds_read_b128 is served in the 16-lane groups of MI355X_MICROARCH.md, LDS): A image, 16-byte chunk index
// XOR ((row>>1)&5): lane (r, q) reads chunks 2q, 2q+1 of row r, conflict-free; W image as in gemm_bf16.h (G[(row>>2)&3]).
// Measured (tools/gemm_tune_split, M = 65,536, bias + fp32 store): NA = 2 218-225 TFLOP/s fp32-equivalent (1.3 PFLOP/s of
// bf16 MFMA), NA = 4 201-209, the fp32 MFMA kernel 137-141.  In-kernel stamps: NA = 2 main loop 3324 cycles per k-tile
// and wave against 3072 of pure MFMA for a SIMD's two waves, at 1.70 GHz; NA = 4 4150-4400 cycles against 3072 at
// 2.07-2.25 GHz (the chip trades clock for MFMA density: the kernel is MFMA-power-bound, all-zero operands run 24 % faster),
// prologue 5.5 k and epilogue 11.7 k cycles per block (all CUs store at once with one workgroup per CU).
// The F16 instantiation of the same body (two fp16 terms, three products, scaled operands) is the opt-in fp16x3 mode.
#pragma once
#include "gemm_bf16.h"

#ifndef GS_EXP_NODMA      // timing experiments of tools/gemm_tune_split (wrong results when set)
#define GS_EXP_NODMA 0
#endif
#ifndef GS_EXP_NOSPLIT
#define GS_EXP_NOSPLIT 0
#endif
#ifndef GS_EXP_NOBAR
#define GS_EXP_NOBAR 0
#endif
#ifndef GS_PN              // column tiles per L2-resident group (0 = all of them), see the tile map in gemm_split_body
#define GS_PN 6
#endif
#define GS_BM 128
#define GS_BK 32
#define GS_A_SLOT (GS_BM * 32)                  // 4-byte units: 128 rows x 128 B
#define GS_BN_OF(NA) ((NA) == 4 ? 256 : 128)
#define GS_LDS_BYTES_OF(NA) ((2 * GS_A_SLOT + 2 * 3 * GS_BN_OF(NA) * 16) * 4)   // 131,072 B (NA = 4) / 81,920 B (NA = 2)
#define GS_BN GS_BN_OF(4)                       // the one-workgroup-per-CU configuration
#define GS_LDS_BYTES GS_LDS_BYTES_OF(4)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// two fp32 -> one dword of two bf16 (round to nearest even): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// The split x = p1 + p2 + p3 (exact for finite x) of four fp32 values, one plane per call: the plane is bf16(r) of
// the running remainder r (two packed dwords); `peel` then subtracts it from r exactly (bf16 -> fp32 is a shift or a
// mask).  11 VALU instructions per element pair for the three planes.
struct Split4 {
    f32x4 r;
    __device__ __forceinline__ void plane(unsigned& d0, unsigned& d1, bool peel) {
        d0 = cvt_pk_bf16(r[0], r[1]);
        d1 = cvt_pk_bf16(r[2], r[3]);
        if (peel) {
            r[0] -= __builtin_bit_cast(float, d0 << 16);
            r[1] -= __builtin_bit_cast(float, d0 & 0xffff0000u);
            r[2] -= __builtin_bit_cast(float, d1 << 16);
            r[3] -= __builtin_bit_cast(float, d1 & 0xffff0000u);
        }
    }
    // experiment: the same with fp16 terms (round to nearest even)
    // (one v_cvt_pk_f16_f32 per pair; the remainder is taken from the PACKED value -- v_cvt_f32_f16 on either half --
    // so that nothing is converted twice: 8 VALU per pair for the two planes)
    __device__ __forceinline__ void plane_f16(unsigned& d0, unsigned& d1, bool peel) {
        const f16x2 a = __builtin_convertvector(f32x2{r[0], r[1]}, f16x2), b = __builtin_convertvector(f32x2{r[2], r[3]}, f16x2);
        d0 = __builtin_bit_cast(unsigned, a);
        d1 = __builtin_bit_cast(unsigned, b);
        if (peel) {
            const f32x2 fa = __builtin_convertvector(a, f32x2), fb = __builtin_convertvector(b, f32x2);
            r[0] -= fa[0]; r[1] -= fa[1]; r[2] -= fb[0]; r[3] -= fb[1];
        }
    }
};

// weights -> three bf16 planes (plane stride n elements), once per iefvad_set_weights
__global__ __launch_bounds__(256) void iefvad_split_planes_kernel(const float* __restrict__ src, bf16_t* __restrict__ planes, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        Split4 s;
        s.r = *(const f32x4*)(src + i);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            unsigned d0, d1;
            s.plane(d0, d1, p < 2);
            *(uint2*)(planes + (size_t)p * n + i) = make_uint2(d0, d1);
        }
    }
}

// The same for up to SPLIT_MANY_MAX matrices in ONE launch (blockIdx.y = matrix): a training step changes every weight, so every
// step re-splits all 10 + 2 K projection matrices and, for the backward's dX products, their transposes -- 90 launches of 5 us (the
// stand-alone split, a transpose into scratch, the split of the scratch) were 0.46 ms of kernel time plus their launch gaps in a 27 ms
// step.  `rows` = 0: planes of the matrix as it is (n elements); rows = n_out > 0: planes of the TRANSPOSE of W [n_out, 768], i.e. of
// W^T [768, n_out] (64 x 32 tiles through LDS; a thread splits two neighbouring elements of a W^T row and stores one dword per plane).
// Bit for bit the planes of iefvad_split_planes_kernel (the same v_cvt_pk_bf16_f32 / exact remainders).
#define SPLIT_MANY_MAX 32
struct SplitManyArgs {
    const float* src[SPLIT_MANY_MAX];
    bf16_t* dst[SPLIT_MANY_MAX];
    unsigned n[SPLIT_MANY_MAX];          // elements per matrix (a multiple of 4; transposed: of 64 x 32)
    int rows[SPLIT_MANY_MAX];
    int count;
};
__global__ __launch_bounds__(256) void iefvad_split_planes_many_kernel(SplitManyArgs a) {
    const int z = blockIdx.y;
    const float* __restrict__ src = a.src[z];
    bf16_t* __restrict__ planes = a.dst[z];
    const size_t n = a.n[z];
    const int n_out = a.rows[z];
    if (n_out == 0) {
        const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
        for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
            Split4 s;
            s.r = *(const f32x4*)(src + i);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                unsigned d0, d1;
                s.plane(d0, d1, p < 2);
                *(uint2*)(planes + (size_t)p * n + i) = make_uint2(d0, d1);
            }
        }
        return;
    }
    // W [n_out, 768] -> W^T [768, n_out]: tile = 64 rows of W (r) x 32 columns (c); out element (c, r)
    __shared__ float tile[64][33];
    const int ncols = (int)(n / (size_t)n_out);       // 768
    const int tr = n_out / 64, tc = ncols / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
        const int r0 = (t / tc) * 64, c0 = (t % tc) * 32;
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[ty + 8 * k][tx] = src[(size_t)(r0 + ty + 8 * k) * ncols + c0 + tx];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = ty + 8 * k;                  // row of W^T inside the tile; this thread: elements r0 + 2 tx, r0 + 2 tx + 1
            float x0 = tile[2 * tx][c], x1 = tile[2 * tx + 1][c];
            const size_t o = (size_t)(c0 + c) * n_out + r0 + 2 * tx;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned d = cvt_pk_bf16(x0, x1);
                *(unsigned*)(planes + (size_t)p * n + o) = d;
                if (p < 2) {
                    x0 -= __builtin_bit_cast(float, d << 16);
                    x1 -= __builtin_bit_cast(float, d & 0xffff0000u);
                }
            }
        }
        __syncthreads();
    }
}

// F16 = false: three bf16 terms per operand, six products (the production arithmetic).
// F16 = true (compute = fp16x3, opt-in): two fp16 terms per operand (22 bits), three products h1 g1 + h1 g2 + h2 g1 --
// half the MFMAs; products good to ~2^-20.4 |a w| worst case (2^-23 typical).  fp16 has a 5-bit exponent, so both operands are scaled by powers of two (exact):
// A by 2^(13 - floor(log2 amaxA)) from the running max |A| word OF THE TILE'S CHUNK that its producer kernel maintained (P.amaxA), W at
// iefvad_set_weights by the same rule (P.amaxW); the epilogue multiplies the accumulators by the inverse (cscale).
// Scaled maxima sit in [2^13, 2^14); elements more than 2^27 below their tensor's maximum fall into fp16's subnormals
// (absolute error <= 2^-38 of the maximum).  Null amax pointers (tools/gemm_tune_split) mean unscaled operands.
template <int NA, bool F16 = false>
__device__ __forceinline__ void gemm_split_body(const GemmBArgs& args, float* smem) {
    constexpr int NP = F16 ? 2 : 3;                    // planes per operand
    constexpr int NT = F16 ? 3 : 6;                    // products per multiply-add
    constexpr int BN = GS_BN_OF(NA);
    constexpr int W_PLANE = BN * 16;                   // 4-byte units: BN rows x 64 B
    constexpr int W_SLOT = NP * W_PLANE;
    constexpr int W_BASE = 2 * GS_A_SLOT;
    constexpr int WROWS = BN / 4;                      // W rows staged per wave and plane
    constexpr int NM = NT * NA;                        // MFMAs per step (24 / 12)
#ifdef GB2_CLOCK_DIAG
    const unsigned long long dg_entry = __builtin_amdgcn_s_memtime();
#endif
    const GemmBProblem& P = args.p[blockIdx.z];
    // Tile map.  xcd_remap gives each XCD a contiguous range of `bid`; the workgroups resident on an XCD at one time are
    // consecutive bids.  Column tiles are taken in groups of PN: inside a group the order is (row panel, column tile of
    // the group), so the resident set is (64 / PN) row panels x PN column tiles: each A panel is fetched once for its PN
    // co-running blocks and the group's W planes (PN x 590 KB at K = 768) stay in the 4 MB L2 instead of being re-read
    // from the Infinity Cache by every panel (N = 2304 with all 18 column tiles in flight: 10.6 MB of W planes per panel,
    // profiles/r01_gemm_split_hbm_traffic.json).  A is re-read once per group (N / (128 PN) times per launch).
    const int ntn = args.N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    {
        const int pn = (GS_PN > 0 && ntn % GS_PN == 0) ? GS_PN : ntn;
        const int ntm = args.M / GS_BM;
        const int grp = bid / (ntm * pn), rem = bid - grp * (ntm * pn);
        tm = rem / pn;
        tn = grp * pn + (rem - tm * pn);
    }
    const int m0 = tm * GS_BM, n0 = tn * BN;
#ifdef GS_EXPERIMENT_A_ALIAS      // tools/gemm_tune_split_alias: every row panel reads one of GS_EXPERIMENT_A_ALIAS panels (A L2-resident)
    const int m0a = (tm % GS_EXPERIMENT_A_ALIAS) * GS_BM;
#else
    const int m0a = m0;
#endif
    const int K = args.K, lda = args.lda;
    const int wplane = args.wplane;                    // bytes between the planes of W
    float ascale = 1.0f, cscale = 1.0f;                // fp16x3: operand scale of A, inverse of both scales
    if constexpr (F16) {
        if (P.amaxA && P.amaxW) {
            const int ea = 13 - amax_exponent(amax_read_chunk(P.amaxA, m0 / IEF_T)), ew = 13 - amax_exponent(amax_read(P.amaxW));
            ascale = __builtin_ldexpf(1.0f, ea);
            cscale = __builtin_ldexpf(1.0f, -ea - ew);
        }
    }

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wrow0 = NA == 4 ? (wave >> 1) * 64 : wave * 32;      // origin of the wave tile inside the block tile
    const int wcol0 = NA == 4 ? (wave & 1) * 128 : 0;
    const int r16 = lane & 15, q16 = lane >> 4;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);

    // ---- staging (LDS-DMA, lane-linear 1 KB images; the swizzle is applied to the SOURCE chunk) ----
    // A: one instruction = 8 rows x 128 B; wave w, instruction j -> rows 32 w + 8 j + (lane >> 3), chunk lane & 7
    // W: one instruction = 16 rows x 64 B; wave w, plane p, instruction j -> rows WROWS w + 16 j + (lane >> 2), chunk lane & 3
    const int nrecA = (int)((GS_BM - 1) * lda + K) * 4, nrecW = (NP - 1) * wplane + (int)((BN - 1) * K + K) * 2;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)P.A + (size_t)m0a * lda * 4), 0, nrecA, 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)P.W + (size_t)n0 * K * 2), 0, nrecW, 0x00020000);
    const int arow = lane >> 3, achk = lane & 7;
    int voA[2];                                        // row bit 3 = j & 1 enters the swizzle
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) voA[jj] = arow * lda * 4 + ((achk ^ (((arow >> 1) & 1) | (jj << 2))) << 4);
    const int wrow = lane >> 2, wchk = lane & 3;
    const int voW = wrow * K * 2 + ((wchk ^ ((0xD2 >> (2 * ((wrow >> 2) & 3))) & 3)) << 4);
#define GLDS16(rs, vo, so, lp) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
    auto stage_a = [&](int tile, int slot) {
        float* Ad = smem + slot * GS_A_SLOT + uwave * 32 * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) GLDS16(rsA, voA[j & 1], ((uwave * 32 + j * 8) * lda + tile * GS_BK) * 4, Ad + j * 8 * 32);
    };
    auto stage_w = [&](int tile, int slot) {
        float* Wd = smem + W_BASE + slot * W_SLOT + uwave * WROWS * 16;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int j = 0; j < WROWS / 16; ++j)
                GLDS16(rsW, voW, p * wplane + ((uwave * WROWS + j * 16) * K + tile * GS_BK) * 2, Wd + p * W_PLANE + j * 16 * 16);
    };

    // ---- fragment addresses (4-byte units) ----
    const int swa = (r16 >> 1) & 5;
    const int a_lo = (wrow0 + r16) * 32 + (((2 * q16) ^ swa) << 2);
    const int a_hi = (wrow0 + r16) * 32 + (((2 * q16 + 1) ^ swa) << 2);
    const int b_of = (wcol0 + r16) * 16 + ((q16 ^ ((0xD2 >> (2 * ((r16 >> 2) & 3))) & 3)) << 2);

    f32x4 acc16[NA][8];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ap[NA][NP], an[NA][NP];                      // planes of the A fragments: current k-tile / next k-tile
    auto mfma_row = [&](int b, int pa, const u32x4& wv) {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if constexpr (F16)
                acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ap[a][pa]),
                                                                     __builtin_bit_cast(f16x8, wv), acc16[a][b], 0, 0, 0);
            else
                acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[a][pa]),
                                                                      __builtin_bit_cast(bf16x8, wv), acc16[a][b], 0, 0, 0);
        }
    };
    // half `hf` (k = 8q + 4 (hf & 1) .. + 3) of row-tile (hf >> 1) of the NEXT k-tile: fp32 fragment -> three planes
    auto split_half = [&](int hf, const f32x4& v) {
        Split4 sp;
        if constexpr (F16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sp.r[e] = v[e] * ascale;       // scalar multiplies: v_pk_mul_f32 is slow beside MFMAs
        } else {
            sp.r = v;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            unsigned d0, d1;
            if constexpr (F16) sp.plane_f16(d0, d1, p < NP - 1); else sp.plane(d0, d1, p < NP - 1);
            an[hf >> 1][p][2 * (hf & 1)] = d0;
            an[hf >> 1][p][2 * (hf & 1) + 1] = d1;
        }
    };
#define GS_FENCE() __builtin_amdgcn_sched_barrier(0)
#define GS_PIPE(mask) __builtin_amdgcn_sched_group_barrier(mask, 1, 0)

    const int nk = K / GS_BK;
    stage_a(0, 0);
    stage_a(1, 1);
    stage_w(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GB2_BARRIER();
#pragma unroll
    for (int hf = 0; hf < 2 * NA; ++hf)                // k-tile 0 is split up front (exposed once per block)
        split_half(hf, *(const f32x4*)(smem + (hf >> 1) * 16 * 32 + ((hf & 1) ? a_hi : a_lo)));
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int p = 0; p < NP; ++p) ap[a][p] = an[a][p];
#ifdef GB2_CLOCK_DIAG
    const unsigned long long dg_c0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dg_last = dg_c0, dg_acc[3] = {0, 0, 0};   // cycles in: MFMA body | vmcnt + lgkmcnt wait | barrier
#define GS_DIAG_STAMP(i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); dg_acc[i] += now - dg_last; dg_last = now; }
#else
#define GS_DIAG_STAMP(i)
#endif

    // One k-tile per iteration: W of slot (kt & 1) against the planes in `ap`; the fp32 A fragments of k-tile kt+1 (slot
    // (kt+1) & 1) are read and split into `an` on the way.  Step B of 8 is the 6 NA MFMAs of column-tile B (six terms x
    // NA row-tiles) with the other work of the step issued BETWEEN them:
    //   NA = 4: steps 0..5 split one half-fragment each (22 VALU), step 6 two (44); NA = 2: steps 0..3 one each;
    //   step 7 moves `an` into `ap` (12 NA v_mov);
    //   the wave's LDS-DMA instructions (A of k-tile kt+2, then W of k-tile kt+1) go out 4 per step in steps 0..3
    //   (NA = 4) / 2 per step in steps 0..4 (NA = 2): issued back to back they fill the vector-memory queue and stall
    //   the wave's MFMA stream;
    //   every step reads the W fragments (and the fp32 A half) of the next step.
    // sched_group_barrier pins that interleaving (hipcc otherwise lumps the VALU work in front of the MFMAs).  On the
    // last k-tile the split works on stale LDS data that is never used, and the DMA descriptors have zero records.
    for (int kt = 0; kt < nk; ++kt) {
        const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)P.W + (size_t)n0 * K * 2), 0,
                                                          (kt + 1 < nk) ? nrecW : 0, 0x00020000);
        const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)P.A + (size_t)m0a * lda * 4), 0,
                                                          (kt + 2 < nk) ? nrecA : 0, 0x00020000);
        float* Wd = smem + W_BASE + ((kt + 1) & 1) * W_SLOT + uwave * WROWS * 16;
        float* Ad = smem + (kt & 1) * GS_A_SLOT + uwave * 32 * 32;
        const int kW = (kt + 1) * GS_BK * 2, kA = (kt + 2) * GS_BK * 4;
        auto dma = [&](int d) {                        // d = 0..3: A, d = 4..: W (plane-major)
            if (GS_EXP_NODMA) return;
            if (d < 4) {
                GLDS16(rA, voA[d & 1], (uwave * 32 + d * 8) * lda * 4 + kA, Ad + d * 8 * 32);
            } else {
                const int p = (d - 4) / (WROWS / 16), j = (d - 4) % (WROWS / 16);      /* d - 4 < NP WROWS / 16 */
                GLDS16(rW, voW, p * wplane + (uwave * WROWS + j * 16) * K * 2 + kW, Wd + p * W_PLANE + j * 16 * 16);
            }
        };
        const float* Wv = smem + W_BASE + (kt & 1) * W_SLOT + b_of;
        const float* Av = smem + ((kt + 1) & 1) * GS_A_SLOT;
        auto a_half = [&](int hf) { return *(const f32x4*)(Av + (hf >> 1) * 16 * 32 + ((hf & 1) ? a_hi : a_lo)); };
        u32x4 w[NP];
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NP; ++p) w[p] = *(const u32x4*)(Wv + p * W_PLANE);
        v0 = a_half(0);
        GS_FENCE();
        // (B and g are literals below: sched_group_barrier takes integer constant expressions only)
#define GS_NSPLIT(B) (NA == 4 ? ((B) < 6 ? 1 : (B) == 6 ? 2 : 0) : ((B) < 4 ? 1 : 0))       /* half-fragments split in step B */
#define GS_NAREAD(B) ((B) < 7 ? GS_NSPLIT((B) + 1) : 0)                                      /* fp32 A reads for step B+1 */
#define GS_NREAD(B) (((B) < 7 ? NP : 0) + GS_NAREAD(B))
#define GS_NDMA(B) (NA == 4 ? ((B) < 4 ? (F16 ? 3 : 4) : 0) : (F16 ? ((B) < 4 ? 2 : 0) : ((B) < 5 ? 2 : 0)))
#define GS_NVALU(B) ((B) == 7 ? 4 * NP * NA : GS_EXP_NOSPLIT ? 0 : (F16 ? 12 : 22) * GS_NSPLIT(B))
#define GS_VSLOT(B, g) (((g) + 1) * GS_NVALU(B) / NM - (g) * GS_NVALU(B) / NM)
#define GS_SLOT(B, g)                                                                                           \
        if ((g) < NM) {                                                                                         \
            GS_PIPE(0x008);                                                                                     \
            if (GS_VSLOT(B, g) > 0) __builtin_amdgcn_sched_group_barrier(0x002, GS_VSLOT(B, g) > 0 ? GS_VSLOT(B, g) : 1, 0); \
            if (GS_NDMA(B) > 0 && !GS_EXP_NODMA && (g) % (NM / 4) == 2 && (g) / (NM / 4) < GS_NDMA(B)) GS_PIPE(0x020); \
        }
#define GS_STEP(B)                                                                                              \
        {                                                                                                       \
            u32x4 wn[NP];                                                                                       \
            f32x4 vn0 = {0.f, 0.f, 0.f, 0.f}, vn1 = {0.f, 0.f, 0.f, 0.f};                                       \
            if ((B) < 7) {                                                                                      \
                _Pragma("unroll") for (int p = 0; p < NP; ++p) wn[p] = *(const u32x4*)(Wv + p * W_PLANE + ((B) + 1) * 16 * 16); \
                if (GS_NAREAD(B) == 1) vn0 = a_half((B) + 1);                                                   \
                if (GS_NAREAD(B) == 2) { vn0 = a_half((B) + 1); vn1 = a_half((B) + 2); }                        \
            }                                                                                                   \
            if constexpr (!F16) {                                                                               \
                mfma_row(B, NP - 1, w[0]);      /* a3 w1 */                                                     \
                mfma_row(B, 0, w[NP - 1]);      /* a1 w3 */                                                     \
                mfma_row(B, 1, w[1]);           /* a2 w2 */                                                     \
            }                                                                                                   \
            mfma_row(B, 1, w[0]);      /* a2 w1 */                                                              \
            mfma_row(B, 0, w[1]);      /* a1 w2 */                                                              \
            mfma_row(B, 0, w[0]);      /* a1 w1 */                                                              \
            if (!GS_EXP_NOSPLIT) {                                                                              \
                if (GS_NSPLIT(B) >= 1) split_half(B, v0);                                                       \
                if (GS_NSPLIT(B) == 2) split_half((B) + 1, v1);                                                 \
            }                                                                                                   \
            if ((B) == 7) {                                                                                     \
                _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                  \
                    _Pragma("unroll") for (int p = 0; p < NP; ++p) ap[a][p] = an[a][p];                         \
            }                                                                                                   \
            _Pragma("unroll") for (int d = 0; d < GS_NDMA(B); ++d) dma(GS_NDMA(B) * (B) + d);                   \
            if (GS_NREAD(B) > 0) __builtin_amdgcn_sched_group_barrier(0x100, GS_NREAD(B) > 0 ? GS_NREAD(B) : 1, 0); \
            GS_SLOT(B, 0) GS_SLOT(B, 1) GS_SLOT(B, 2) GS_SLOT(B, 3) GS_SLOT(B, 4) GS_SLOT(B, 5)                 \
            GS_SLOT(B, 6) GS_SLOT(B, 7) GS_SLOT(B, 8) GS_SLOT(B, 9) GS_SLOT(B, 10) GS_SLOT(B, 11)               \
            GS_SLOT(B, 12) GS_SLOT(B, 13) GS_SLOT(B, 14) GS_SLOT(B, 15) GS_SLOT(B, 16) GS_SLOT(B, 17)           \
            GS_SLOT(B, 18) GS_SLOT(B, 19) GS_SLOT(B, 20) GS_SLOT(B, 21) GS_SLOT(B, 22) GS_SLOT(B, 23)           \
            GS_FENCE();                                                                                         \
            if ((B) < 7) {                                                                                      \
                _Pragma("unroll") for (int p = 0; p < NP; ++p) w[p] = wn[p];                                    \
                if (GS_NAREAD(B) >= 1) v0 = vn0;                                                                \
                if (GS_NAREAD(B) == 2) v1 = vn1;                                                                \
            }                                                                                                   \
        }
        GS_STEP(0) GS_STEP(1) GS_STEP(2) GS_STEP(3) GS_STEP(4) GS_STEP(5) GS_STEP(6) GS_STEP(7)
#undef GS_STEP
#undef GS_SLOT
#undef GS_VSLOT
#undef GS_NVALU
#undef GS_NDMA
#undef GS_NREAD
#undef GS_NAREAD
#undef GS_NSPLIT
        GS_DIAG_STAMP(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GS_DIAG_STAMP(1)
        if (!GS_EXP_NOBAR) GB2_BARRIER();
        GS_DIAG_STAMP(2)
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
#undef GS_PIPE
#undef GS_FENCE
#undef GLDS16
    // (the last tile ended with lgkmcnt(0) + barrier: the ring is dead, the epilogue image may overwrite it)
    f32x16 unused[NA / 2][4];
    gemm_wave_epilogue<true, NA / 2>(args, P, smem, m0, n0, wrow0, wcol0, unused, acc16, cscale, F16 ? P.amaxC : nullptr);
#ifdef GB2_CLOCK_DIAG
    if (threadIdx.x == 0 && P.C2) {
        unsigned long long* dy = (unsigned long long*)P.C2 + 5 * gridDim.x * gridDim.z + 2 * (blockIdx.x + gridDim.x * blockIdx.z);
        dy[0] = dg_c0 - dg_entry;                                   // prologue: entry -> main loop
        dy[1] = __builtin_amdgcn_s_memtime() - dg_loop_end;         // epilogue (stores issued, not necessarily landed)
    }
#endif
}

// 128 x 256, one workgroup per CU
__global__ __launch_bounds__(256, 1) void iefvad_gemm_split_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_split_body<4>(args, smem);
}

// 128 x 128, two workgroups per CU
__global__ __launch_bounds__(256, 2) void iefvad_gemm_split_n128_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_split_body<2>(args, smem);
}

// fp16x3 (opt-in): 128 x 128, two workgroups per CU
__global__ __launch_bounds__(256, 2) void iefvad_gemm_split_f16_n128_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_split_body<2, true>(args, smem);
}

#ifdef GS_EXPERIMENT_F16     // tools/gemm_tune_split only
__global__ __launch_bounds__(256, 1) void iefvad_gemm_split_f16_kernel(GemmBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_split_body<4, true>(args, smem);
}
#endif

// weights -> two fp16 planes scaled by 2^(13 - floor(log2 max|W|)) (plane stride n elements); `amax` was filled by
// iefvad_amax_kernel over the same matrix
__global__ __launch_bounds__(256) void iefvad_split_planes_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ planes,
                                                                      size_t n, const float* __restrict__ amax) {
    const float scale = __builtin_ldexpf(1.0f, 13 - amax_exponent(amax_read(amax)));
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        Split4 s;
        s.r = *(const f32x4*)(src + i) * scale;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned d0, d1;
            s.plane_f16(d0, d1, p < 1);
            *(uint2*)(planes + (size_t)p * n + i) = make_uint2(d0, d1);
        }
    }
}
