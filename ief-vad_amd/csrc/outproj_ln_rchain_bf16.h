// bf16 mode: out_proj + bias + residual + LayerNorm (+ whitening LayerNorm) (/root/reference/model/imf_vad.py:116-117,121-123), third
// form (round 5): the persistent row-block kernel of outproj_ln_pchain_bf16.h with the LayerNorm done IN THE ACCUMULATOR REGISTERS.
//
// Round 4's phase stamps (profiles/r04_outproj_pchain_phase_stamps.log) put 31 k of a 64-row block's 58 k cycles into the epilogue,
// and 18.4 k of those not into LayerNorm arithmetic (12.2 k) but into moving the accumulators through LDS so that a wave owns
// whole rows: 4 x (park 16 rows, barrier, normalise, barrier).  A row-per-wave layout is what ln_row's summation order asks for:
//     lane l of the row's wave holds columns 256 j + 4 l + e (j = 0..2, e = 0..3); s_l = hsum((v_0 + v_1) + v_2); the 64 s_l go through a
//     binary tree over the bits of l in ascending order (wave_sum, common.h).
// The same tree can be walked without moving the data if the OUTPUT COLUMNS of the MFMA tiles are dealt to the waves accordingly.  The
// weights are the A operand here (a lane holds four consecutive output columns of one row), and which matrix rows a wave streams is
// the pack kernel's choice (iefvad_wstream_pack_kernel, colmap = 1):
//     wave w, column tile b (0..5), lane (m, q), element e   ->   column 256 (b >> 1) + 32 w + 16 (b & 1) + 4 q + e
// i.e. the lane is the "virtual lane" l = 8 w + 4 (b & 1) + q of row 16 a + m for BOTH values of b & 1, and b >> 1 is ln_row's j.  Then
//     s_l            = hsum((acc[a][half] + acc[a][2 + half]) + acc[a][4 + half])             in the lane          (ln_row's own expression)
//     bits 0, 1 of l = q        two cross-row exchanges between the four lanes of a row (ds_bpermute: __shfl_xor 16, 32)
//     bit  2         = half     one add in the lane
//     bits 3, 4, 5   = w        through LDS: [64 rows][8 waves] partials (2 KB), one barrier, ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7))
// -- operation for operation the tree of wave_sum, so mean, variance and every output keep the bits of ln_row: the kernel is
// bit-identical to the LayerNorm kernel, to the round-3 / round-4 kernels and to the two-kernel path (tests/test_gpu_bf16.py compares them).
// Per block: 2 exchanges per LayerNorm (mean, centred variance) = 4 barriers with the whitening LayerNorm, against 9 before; no park
// traffic (196 KB written and read per block); LDS 115 KB instead of 159 KB.  Residual rows and the stores use the accumulator layout:
// 64-byte (fp32) / 32-byte (bf16) pieces per lane quartet, two adjacent pieces per row and j from a wave (b even / odd) = one 128-byte line.
#pragma once
#include "outproj_ln_pchain_bf16.h"

#define OR_XCH_OFF OC_IMG_BYTES                                          // 98,304: exchange buffers, 2 x [64 rows][8 waves] floats
#define OR_XCH_BYTES (64 * 8 * 4)
#define OR_AFF_OFF (OR_XCH_OFF + 2 * OR_XCH_BYTES)                       // 102,400: bias | g1 | b1 | g2 | b2
#define OR_VO_OFF (OR_AFF_OFF + 5 * IEF_D * 4)                           // 117,760: the image DMA's per-lane source offsets, [12 pieces][512 threads] ints
#define OR_LDS_BYTES (OR_VO_OFF + OP_DMA_PER_WAVE * 512 * 4)             // 142,336

// the column a lane's element e of tile b belongs to (see above); also the pack kernel's row choice
__host__ __device__ __forceinline__ int or_col(int w, int b, int q) { return 256 * (b >> 1) + 32 * w + 16 * (b & 1) + 4 * q; }

// W [768, 768] bf16 -> per wave w: pieces (kt, b), lane (r, q): 8 bf16 = W[or_col(w, b, 0) + r][32 kt + 8 q .. + 7]  (r = 4 q' + e of the output)
__global__ __launch_bounds__(256) void iefvad_wstream_pack_colmap_kernel(const bf16_t* W, char* stream) {
    const size_t per_wave = (size_t)(OC_PIECES + OC_PAD_PIECES) * 64;
    const size_t total = 8 * per_wave;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(u / per_wave);
        const size_t v = u - (size_t)w * per_wave;
        const int lane = (int)(v & 63);
        const int piece = (int)(v >> 6);
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (piece < OC_PIECES) {
            const int kt = piece / OC_NB, b = piece % OC_NB, r = lane & 15, q = lane >> 4;
            val = *(const f32x4*)(W + (size_t)(or_col(w, b, 0) + r) * IEF_D + 32 * kt + 8 * q);
        }
        *(f32x4*)(stream + u * 16) = val;
    }
}

// One row statistic of the block's 64 rows.  h[a][half] is this lane's hsum for virtual lane 8 w + 4 half + q of row 16 a + m; returns
// the full-row sum of rows 16 a + m (a = 0..3) in every lane, through wave_sum's tree: bits 0, 1 (q), bit 2 (half), bits 3 - 5 (the waves).
__device__ __forceinline__ void or_row_totals(float (&h)[4][2], float (&tot)[4], float* xch, int m, int q, int wave) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int f = 0; f < 2; ++f) h[a][f] += __shfl_xor(h[a][f], 16, 64);        // bit 0 of the virtual lane
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int f = 0; f < 2; ++f) h[a][f] += __shfl_xor(h[a][f], 32, 64);        // bit 1
    if (q == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) xch[(16 * a + m) * 8 + wave] = h[a][0] + h[a][1];   // bit 2
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x4 lo = *(const f32x4*)(xch + (16 * a + m) * 8), hi = *(const f32x4*)(xch + (16 * a + m) * 8 + 4);
        tot[a] = ((lo[0] + lo[1]) + (lo[2] + lo[3])) + ((hi[0] + hi[1]) + (hi[2] + hi[3]));      // bits 3, 4, 5: the waves
    }
}

// LayerNorm of the block's 64 rows, in place, in the accumulator layout: ln_row's operations (rowops.h: ln_center_rstd + the affine
// step) on the three chunks j = 0..2 of each of the lane's eight virtual lanes.  g, b: the affine vectors in LDS, natural column order.
// affl = LDS address of the affine block + this lane's column base (ONE register; every vector below is that plus an immediate offset)
template <int G_OFF, int B_OFF>
__device__ __forceinline__ void or_layernorm(f32x4 (&v)[4][OC_NB], const char* affl, float* xch0, float* xch1, int m, int q, int wave, float eps) {
    float h[4][2], tot[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int f = 0; f < 2; ++f) h[a][f] = ln_hsum((v[a][f] + v[a][2 + f]) + v[a][4 + f]);
    or_row_totals(h, tot, xch0, m, q, wave);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float mean = tot[a] * (1.0f / IEF_D);
        const f32x4 m4 = {mean, mean, mean, mean};
#pragma unroll
        for (int bb = 0; bb < OC_NB; ++bb) v[a][bb] = v[a][bb] - m4;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            f32x4 sq = v[a][f] * v[a][f];
            sq = v[a][2 + f] * v[a][2 + f] + sq;
            sq = v[a][4 + f] * v[a][4 + f] + sq;
            h[a][f] = ln_hsum(sq);
        }
    }
    or_row_totals(h, tot, xch1, m, q, wave);
    float rstd[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) rstd[a] = 1.0f / sqrtf(tot[a] * (1.0f / IEF_D) + eps);
#pragma unroll
    for (int bb = 0; bb < OC_NB; ++bb) {
        const int c = (256 * (bb >> 1) + 16 * (bb & 1)) * 4;
        const f32x4 gv = *(const f32x4*)(affl + G_OFF + c), bv = *(const f32x4*)(affl + B_OFF + c);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 r4 = {rstd[a], rstd[a], rstd[a], rstd[a]};
            v[a][bb] = (v[a][bb] * r4) * gv + bv;
        }
    }
}

__global__ __launch_bounds__(512, 2) void iefvad_outproj_ln_rchain_bf16_kernel(OutLnChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    const OutLnChainProblem& P = args.p[blockIdx.y];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int nblk = args.M / OC_BM;
    const bool two = P.g2 != nullptr;

    // ---- once per workgroup: bias and the LayerNorms' affine terms -> LDS (5 x 768 floats, natural column order)
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
            *(f32x4*)(lds + OR_AFF_OFF + (slot * 64 + lane) * 16) = aff[i];
        }
    }
    // ---- the image by LDS-DMA, as outproj_ln_pchain_bf16.h
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)((size_t)args.M * IEF_D * 2), 0x00020000);
    // The twelve per-lane source offsets live in LDS, not in registers: twelve more live registers made hipcc spill them, and a spilled
    // offset comes back through scratch memory -- a vector-memory operation that, retiring in order behind the DMA pieces already issued,
    // serialised the image fetch (first build: 25 k cycles per block in this phase)
    int* vo_lds = (int*)(lds + OR_VO_OFF) + t;
#pragma unroll
    for (int i = 0; i < OP_DMA_PER_WAVE; ++i) {
        const int L = (OP_DMA_PER_WAVE * wave + i) * 64 + lane, r = L / 96, pc = L - r * 96;
        vo_lds[512 * i] = (r * 96 + ((pc & ~15) | ((pc ^ r) & 15))) * 16;
    }
#define OR_IMAGE_DMA(blk_)                                                                                                          \
    do {                                                                                                                            \
        int vo_[OP_DMA_PER_WAVE];                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < OP_DMA_PER_WAVE; ++i) vo_[i] = vo_lds[512 * i];                                       \
        _Pragma("unroll") for (int i = 0; i < OP_DMA_PER_WAVE; ++i)                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(lds + (OP_DMA_PER_WAVE * wave + i) * 1024), 16, vo_[i], \
                                                     (blk_) * (OC_BM * IEF_D * 2), 0, 0);                                           \
    } while (0)
    OR_IMAGE_DMA(blockIdx.x);
    __builtin_amdgcn_sched_barrier(0);

    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(P.stream + (size_t)wave * args.wave_stride), 0, (int)args.wave_stride, 0x00020000);
    const int vlane = lane * 16;
#define OR_LOAD(piece_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (int)((piece_) << 10), 0))
    int rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rd[j] = m * (IEF_D * 2) + (((4 * j + q) ^ m) & 15) * 16;
    // bias | g1 | b1 | g2 | b2 (768 floats each) are read at ONE per-lane address plus compile-time offsets (ds_read's offset field): left to
    // itself hipcc keeps a register per (array, tile) address -- 30 of them -- across the block loop and spills them
    int affoff = OR_AFF_OFF + (32 * wave + 4 * q) * 4;
    asm volatile("" : "+v"(affoff));
    const char* affl = lds + affoff;
#define OR_AFF(k_, b_) (*(const f32x4*)(affl + ((k_) * IEF_D + 256 * ((b_) >> 1) + 16 * ((b_) & 1)) * 4))
    float* xch0 = (float*)(lds + OR_XCH_OFF);
    float* xch1 = (float*)(lds + OR_XCH_OFF + OR_XCH_BYTES);
    const int cbase = 32 * wave + 4 * q;                                 // this lane's first column of tile b = cbase + a compile-time constant
#define OR_COL(b_) (cbase + 256 * ((b_) >> 1) + 16 * ((b_) & 1))

#ifdef OC_DIAG
    unsigned long long dsum[7] = {0, 0, 0, 0, 0, 0, 0}, dt0 = 0, dt1;   // image wait | main loop | residual requests + image-free barrier + DMA issue | residual add | LayerNorm(s) | stores | blocks
#define OR_T(i) do { dt1 = __builtin_amdgcn_s_memtime(); dsum[i] += dt1 - dt0; dt0 = dt1; } while (0)
#define OR_T0() do { dt0 = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OR_T(i)
#define OR_T0()
#endif
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int m0 = blk * OC_BM;
        int mrow = m;                         // per-block copy the compiler cannot see through: row addresses are formed where they are used
        asm volatile("" : "+v"(mrow));        // instead of being kept (and spilled) across the main loop
        OR_T0();
        f32x4 rg[OC_DEPTH];
#pragma unroll
        for (int s = 0; s < OC_DEPTH; ++s) rg[s] = OR_LOAD(s);
        // residual rows in the accumulator layout: 16-byte pieces of rows 16 a + m.  Two row groups are requested now and arrive under the
        // main loop, the other two after it, into the same registers
        f32x4 res[2][OC_NB];
#define OR_FETCH_RES(a0_)                                                                                          \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                \
        const float* rp = P.R + (size_t)(m0 + 16 * ((a0_) + u) + mrow) * IEF_D;                                   \
        _Pragma("unroll") for (int b = 0; b < OC_NB; ++b) res[u][b] = *(const f32x4*)(rp + OR_COL(b));            \
    }
#define OR_ADD_RES(a0_)                                                                                            \
    _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                  \
        _Pragma("unroll") for (int b = 0; b < OC_NB; ++b)                                                          \
            acc[(a0_) + u][b] = (acc[(a0_) + u][b] + OR_AFF(0, b)) + res[u][b];
        OR_FETCH_RES(0)
        __builtin_amdgcn_sched_barrier(0);
        // this wave's image pieces were requested before those 18 loads
        asm volatile("s_waitcnt vmcnt(18)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();
        OR_T(0);

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
                for (int a = 0; a < 4; ++a) ga[a] = *(const f32x4*)(lds + rd[j] + a * (16 * IEF_D * 2) + k4 * 256);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < OC_NB; ++b) {
                    const f32x4 w = rg[b];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, ga[a]), acc[a][b], 0, 0, 0);
                    rg[b] = OR_LOAD(p + OC_DEPTH);
                    __builtin_amdgcn_sched_barrier(0);
                    ++p;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < OC_DEPTH; ++s) asm volatile("" :: "v"(rg[s]));
        OR_T(1);
        // (acc + bias) + residual: the first two row groups' rows arrived under the main loop; the other two are requested now, and the next
        // block's image behind the LAST of them (in-order retirement: whatever is requested after the image pieces waits for them)
        OR_ADD_RES(0)
        OR_FETCH_RES(2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();                        // every wave is done with the image: it may be overwritten
        const int nxt = blk + (int)gridDim.x;
        if (nxt < nblk) { OR_IMAGE_DMA(nxt); }
        OR_T(2);
        OR_ADD_RES(2)
        OR_T(3);
        or_layernorm<IEF_D * 4, 2 * IEF_D * 4>(acc, affl, xch0, xch1, m, q, wave, args.eps);
        if (two) or_layernorm<3 * IEF_D * 4, 4 * IEF_D * 4>(acc, affl, xch0, xch1, m, q, wave, args.eps);
        OR_T(4);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const size_t row = (size_t)(m0 + 16 * a + mrow);
            if (P.y) {
                float* yp = P.y + row * IEF_D;
#pragma unroll
                for (int b = 0; b < OC_NB; ++b) *(f32x4*)(yp + OR_COL(b)) = acc[a][b];
            }
            if (P.yb) {
                bf16_t* yb = P.yb + row * IEF_D;
#pragma unroll
                for (int b = 0; b < OC_NB; ++b) *(bf16x4_t*)(yb + OR_COL(b)) = to_bf16x4(acc[a][b]);
            }
        }
        OR_T(5);
#ifdef OC_DIAG
        dsum[6] += 1;
#endif
    }
#ifdef OC_DIAG
    if (args.diag && t == 0)
        for (int i = 0; i < 7; ++i) args.diag[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = dsum[i];
#endif
#undef OR_LOAD
#undef OR_IMAGE_DMA
#undef OR_COL
#undef OR_AFF
#undef OR_FETCH_RES
#undef OR_ADD_RES
}
