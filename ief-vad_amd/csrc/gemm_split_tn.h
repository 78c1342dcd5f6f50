// Weight gradients in the bf16x6 arithmetic: C[i, j] = sum_r A[r, i] B[r, j] with BOTH operands row-major over the contraction index
// (dW = dY^T X of a Linear, /root/reference/model/imf_vad.py:115-150 under autograd; train.h `launch_dw`), fp32-accurate as
// gemm_split.h: every fp32 operand element is the exact sum of three bf16 terms and a product is accumulated in fp32 from the six
// largest bf16 MFMA products.
//
// Neither operand can be pre-split (both are activations of the step) and neither has the contraction index contiguous, so:
//   * tiles are k-MAJOR fp32 images in LDS, [16 r][128 columns], filled by LDS-DMA (one 512-byte row segment per half wave) into a
//     ring of THREE buffers: a tile is 24 MFMAs per wave (0.4 us), so a tile requested one tile ahead would not have landed; requested
//     two ahead and waited for by COUNT (`s_waitcnt vmcnt(4)`: the four requests of the youngest tile may still be in flight; with
//     -DTN_NBUF=4, three ahead and vmcnt(8): measured 1 % slower at its two workgroups per CU); the
//     MFMA fragment of lane (i, h) -- eight consecutive r of ONE column -- is eight ds_read_b32 down a column: lanes i = 0..31 read
//     consecutive dwords, and the 16-byte chunks of rows with (r >> 3) odd are XOR-swizzled by 8 (32 floats; on the per-lane SOURCE
//     address) so that the two lane halves (r and r + 8) use different banks;
//   * each wave splits the fragments it reads in registers (split8 of attention_split.h: 22 VALU per 8 elements and plane set):
//     four fragments (two 32-column tiles of A, two of B) per 16-r tile feed 2 x 2 x 6 = 24 MFMAs.
// 128 x 128 output tile, 4 waves (2 x 2) of 64 x 64, three workgroups per CU (48 KB of LDS: three buffers x (A + B) x 8 KB; 114
// VGPRs).  Split-K over the rows in grid.z slices; each slice writes its partial tile, the caller reduces them in index order
// (deterministic).  The workgroups of the first column block also sum the columns of their A tiles: the bias gradient.
#pragma once
#include "attention_split.h"

static int g_num_cus = 256;            // set by iefvad_create: the host picks as many row slices as fill the chip's workgroup slots once

struct TnArgs {
    const float* A;      // [nk_total * 16, lda]: dY
    const float* B;      // [nk_total * 16, ldb]: X
    float* C;            // [slices][M][ldc] partial products
    int M, N;            // M a multiple of 128, N of the kernel's column block (128 / 256)
    int nk_total;        // 16-row k-tiles in all; slice z takes k-tiles [z nk_total / slices, (z + 1) nk_total / slices)
    int slices;
    int lda, ldb, ldc;
    int tiles_n;         // N / column block
    float* colsum;       // nullable: [slices][M] partial column sums of A (the bias gradient of the same Linear: db = dY^T 1), fp32
};

#define TN_BK 16
#ifndef TN_NBUF
#define TN_NBUF 3
#endif
#define TN_A_FLOATS (TN_BK * 128)                            // the A image of a k-tile: 8 KB
#define TN_B_FLOATS(NB) (TN_BK * 64 * (NB))                  // the B image: 8 KB (NB = 2) / 16 KB (NB = 4)
#define TN_LDS_BYTES_OF(NB) (TN_NBUF * (TN_A_FLOATS + TN_B_FLOATS(NB)) * 4)      // [buffer b: A | B]
#define TN_LDS_BYTES TN_LDS_BYTES_OF(2)

// NB = 32-column tiles of B per wave: 2 = the 128 x 128 block above (three workgroups per CU); 4 = a 128 x 256 block, 64 x 128 per wave
// (round 5: six fragment splits per 48 MFMAs instead of four per 24 -- the kernel is bound by the VALU issue of its splits, 7.3 VALU
// per MFMA against the ~7 issue slots an MFMA leaves -- 72 KB of LDS, two workgroups per CU).  The row slices are RAGGED (any count):
// the host picks as many slices as fill the chip's workgroup slots once.
template <int NB>
__device__ __forceinline__ void gemm_split_tn_body(const TnArgs& args, float* tn_smem) {
    constexpr int BN = 64 * NB;                      // columns of B per workgroup
    constexpr int RPP = 256 / BN;                    // B rows per 1 KB DMA piece (2 / 1)
    constexpr int BPW = (TN_BK / RPP) / 4;           // B pieces per wave and tile (2 / 4)
    constexpr int REQ = 2 + BPW;                     // DMA requests per wave and tile
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = (args.M / 128) * args.tiles_n;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int z = bid / tiles, tt = bid - z * tiles;
    const int tm = tt / args.tiles_n, tn = tt - tm * args.tiles_n;
    const int lda = args.lda, ldb = args.ldb;
    const int kt_begin = (int)((long long)z * args.nk_total / args.slices);
    const int nk = (int)((long long)(z + 1) * args.nk_total / args.slices) - kt_begin;
    const int R = nk * TN_BK;

    // DMA, A: piece p (1 KB) of a tile = rows 2 p, 2 p + 1; lane L fills row 2 p + (L >> 5), 16-byte position L & 31, with the source
    // chunk (L & 31) ^ 8 ((row >> 3) & 1); wave w sends pieces 2 w, 2 w + 1.  B: RPP rows per piece (BN / 4 chunks per row), the same
    // swizzle; wave w sends pieces BPW w .. BPW w + BPW - 1.
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(args.A + (size_t)kt_begin * TN_BK * lda + tm * 128), 0, (R - 1) * lda * 4 + 512, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(args.B + (size_t)kt_begin * TN_BK * ldb + tn * BN), 0, (R - 1) * ldb * 4 + BN * 4, 0x00020000);
    int voa[2], vob[BPW];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = 2 * (2 * wave + p) + (lane >> 5), c = (lane & 31) ^ (((row >> 3) & 1) << 3);
        voa[p] = row * lda * 4 + c * 16;
    }
#pragma unroll
    for (int p = 0; p < BPW; ++p) {
        const int row = RPP * (BPW * wave + p) + lane / (64 / RPP), c = (lane % (64 / RPP)) ^ (((row >> 3) & 1) << 3);
        vob[p] = row * ldb * 4 + c * 16;
    }
    auto dma = [&](int kt, int buf) {
        float* dA = tn_smem + buf * (TN_A_FLOATS + TN_B_FLOATS(NB)) + (2 * wave) * 256;
        float* dB = tn_smem + buf * (TN_A_FLOATS + TN_B_FLOATS(NB)) + TN_A_FLOATS + (BPW * wave) * 256;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dA + p * 256), 16, voa[p], kt * (TN_BK * lda * 4), 0, 0);
#pragma unroll
        for (int p = 0; p < BPW; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(dB + p * 256), 16, vob[p], kt * (TN_BK * ldb * 4), 0, 0);
    };

    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment of 32-column tile t2 of an image with row pitch P: element j = img[(8 h + j) * P + 32 (t2 ^ h) + i]
#define TN_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), c_, 0, 0, 0)
#define TN_SIX(x_, y_, c_) TN_MFMA(x_[2], y_[0], c_); TN_MFMA(x_[0], y_[2], c_); TN_MFMA(x_[1], y_[1], c_); TN_MFMA(x_[1], y_[0], c_); TN_MFMA(x_[0], y_[1], c_); TN_MFMA(x_[0], y_[0], c_)
    // s_waitcnt vmcnt(n), n <= 15
#define TN_WAIT_VM(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | (n_))

#ifdef TN_PROBE_NOSPLIT
    u32x4 pa[2][3] = {}, pb[NB][3] = {};
#endif
    const bool do_cs = args.colsum != nullptr && tn == 0;
    const int cs_c = t & 127, cs_h = t >> 7;
    float cs = 0.f;
#pragma unroll
    for (int b = 0; b < TN_NBUF - 1; ++b)
        if (b < nk) dma(b, b);
    // tile 0 has landed when at most the requests of the tiles behind it are in flight
    if (nk >= TN_NBUF - 1 && TN_NBUF == 4) TN_WAIT_VM(2 * REQ); else if (nk >= 2) TN_WAIT_VM(REQ); else TN_WAIT_VM(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + TN_NBUF - 1 < nk) dma(kt + TN_NBUF - 1, (kt + TN_NBUF - 1) % TN_NBUF);      // the buffer tile kt - 1 has left
        const float* imgA = tn_smem + (kt % TN_NBUF) * (TN_A_FLOATS + TN_B_FLOATS(NB));
        const float* imgB = imgA + TN_A_FLOATS;
#ifndef TN_PROBE_NOSPLIT
        u32x4 pa[2][3], pb[NB][3];
#endif
#pragma unroll
        for (int f = 0; f < 2 + NB; ++f) {
            const float* p = f < 2 ? imgA + 8 * h * 128 + 32 * ((2 * wm + f) ^ h) + i : imgB + 8 * h * BN + 32 * ((NB * wn + (f - 2)) ^ h) + i;
            constexpr int PA = 128;
            const int P = f < 2 ? PA : BN;
            const f32x4 lo = {p[0], p[P], p[2 * P], p[3 * P]}, hi = {p[4 * P], p[5 * P], p[6 * P], p[7 * P]};
#ifdef TN_PROBE_NOSPLIT      // timing probe only (wrong results): the fragment reads stay, the register split runs for the first k-tile only
            asm volatile("" :: "v"(lo), "v"(hi));
            if (kt == 0)
#endif
            split8<false, 3>(lo, hi, f < 2 ? pa[f] : pb[f - 2], 1.0f);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) { TN_SIX(pa[a], pb[b], acc[a][b]); }
        if (do_cs) {         // the column sums ride along in the workgroups of the first column block: 8 rows of one column per thread
            const float* q = imgA + 8 * cs_h * 128 + (cs_c ^ (32 * cs_h));
#pragma unroll
            for (int j = 0; j < 8; ++j) cs += q[j * 128];
        }
        // tile kt + 1 has landed when at most the requests of the tiles behind it are in flight
        const int young = nk - kt - 2;
        if (young >= 2 && TN_NBUF == 4) TN_WAIT_VM(2 * REQ);
        else if (young >= 1) TN_WAIT_VM(REQ);
        else TN_WAIT_VM(0);
        __syncthreads();                     // ... for every wave, and every wave has left this tile's images
    }
#undef TN_SIX
#undef TN_MFMA
#undef TN_WAIT_VM
    if (do_cs) {             // (the loop ended with a barrier: the ring is dead)
        tn_smem[t] = cs;
        __syncthreads();
        if (t < 128) args.colsum[(size_t)z * args.M + tm * 128 + t] = tn_smem[t] + tn_smem[t + 128];
    }
    // partial tile out: accumulator column = lane & 31 (the B column), row = (r & 3) + 8 (r >> 2) + 4 h (the A column)
    float* C = args.C + (size_t)z * args.M * args.ldc + (size_t)(tm * 128 + wm * 64) * args.ldc + tn * BN + wn * (32 * NB);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(size_t)(32 * a + (r & 3) + 8 * (r >> 2) + 4 * h) * args.ldc + 32 * b + i] = acc[a][b][r];
}

__global__ __launch_bounds__(256, TN_NBUF == 3 ? 3 : 2) void iefvad_gemm_split_tn_kernel(TnArgs args) {
    extern __shared__ __attribute__((aligned(16))) float tn_smem[];
    gemm_split_tn_body<2>(args, tn_smem);
}

__global__ __launch_bounds__(256, 2) void iefvad_gemm_split_tn256_kernel(TnArgs args) {
    extern __shared__ __attribute__((aligned(16))) float tn_smem[];
    gemm_split_tn_body<4>(args, tn_smem);
}
