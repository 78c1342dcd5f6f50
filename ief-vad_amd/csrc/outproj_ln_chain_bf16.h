// bf16 mode: attention out_proj + bias + residual + LayerNorm (+ the whitening LayerNorm after the last layer)
// (/root/reference/model/imf_vad.py:116-117,121-123), second design (round 3), on the refinement chain's structure
// (refine_chain_bf16.h): a 512-thread workgroup owns 64 rows x all 768 columns;
//   * the 64 x 768 bf16 attention output is ONE 96 KB image in LDS (B operand of every wave's MFMAs; same swizzle);
//   * every wave streams ITS OWN 96 columns of W_o straight into a six-deep register ring (1 KB pieces in fragment order,
//     iefvad_wstream_pack_kernel at iefvad_set_weights): no LDS ring, no barrier in the main loop;
//   * the fp32 residual rows a wave will normalise (4 of the first 32, in the LayerNorm kernel's lane layout: 48 registers)
//     are requested BEFORE the main loop and arrive under it; the second 32 rows' residual is requested while the first 32
//     are being normalised.  That overlap is the point: the first fused kernel (outproj_ln_bf16.h: 128 x 768 per
//     workgroup, LDS-DMA k-tile ring) spent ~25 us per tile multiplying with HBM idle and then ~50 us moving residual and
//     output rows with the matrix pipe idle (0.18 of the pipe, SQ_WAIT_ANY 0.49);
//   * epilogue: the accumulators are parked in LDS 32 rows at a time ([32][772] fp32 over the dead image) and each wave
//     runs ln_row -- the LayerNorm kernel's own code -- on four whole rows: (acc + bias) + residual, LayerNorm, optional
//     whitening LayerNorm, fp32 and bf16 row stores of 512-byte segments.  Same products in the same k order, same epilogue
//     arithmetic: every output is bit-identical to the first fused kernel and to the two-kernel path.
#pragma once
#include "gemm_bf16.h"
#include "rowops.h"

#define OC_BM 64
#define OC_IMG_BYTES (OC_BM * IEF_D * 2)            // 98,304
#define OC_PARK_LD 772                               // floats per parked row (768 + 4: conflict-free 16-byte tile stores)
#define OC_PARK_BYTES (32 * OC_PARK_LD * 4)          // 98,816
#define OC_AFF_OFF OC_PARK_BYTES                     // behind the park: bias | g1 | b1 | g2 | b2, 768 floats each (15 KB)
#define OC_LDS_BYTES (OC_PARK_BYTES + 5 * IEF_D * 4) // 114,176 B
#define OC_KT (IEF_D / 32)
#define OC_NB 6
#define OC_DEPTH 6
#define OC_PIECES (OC_KT * OC_NB)                    // 144 per wave
#define OC_PAD_PIECES OC_DEPTH

// npass = 1: a [768, 768] matrix (out_proj); npass = 3: the packed [2304, 768] in_proj matrix, pass = q | k | v (inproj_chain_bf16.h)
static inline size_t wstream_wave_stride_bytes(int npass = 1) { return (size_t)(npass * OC_PIECES + OC_PAD_PIECES) * 1024; }
static inline size_t wstream_bytes(int npass = 1) { return 8 * wstream_wave_stride_bytes(npass); }

// W [768 npass, 768] bf16 -> per wave w: pieces (pass, kt, b), lane (r, q): 8 bf16 = W[768 pass + 96 w + 16 b + r][32 kt + 8 q .. + 7]
__global__ __launch_bounds__(256) void iefvad_wstream_pack_kernel(const bf16_t* W, char* stream, int npass) {
    const size_t per_wave = (size_t)(npass * OC_PIECES + OC_PAD_PIECES) * 64;
    const size_t total = 8 * per_wave;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(u / per_wave);
        const size_t v = u - (size_t)w * per_wave;
        const int lane = (int)(v & 63);
        const int piece = (int)(v >> 6);
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (piece < npass * OC_PIECES) {
            const int pass = piece / OC_PIECES, pp = piece % OC_PIECES;
            const int kt = pp / OC_NB, b = pp % OC_NB, r = lane & 15, q = lane >> 4;
            val = *(const f32x4*)(W + (size_t)(IEF_D * pass + 96 * w + 16 * b + r) * IEF_D + 32 * kt + 8 * q);
        }
        *(f32x4*)(stream + u * 16) = val;
    }
}

struct OutLnChainProblem {
    const bf16_t* A;         // attention output [M, 768] bf16
    const char* stream;      // iefvad_wstream_pack_kernel(out_proj.weight)
    const float* bias;       // [768]
    const float* R;          // residual: the layer's fp32 input rows [M, 768] (may be the buffer y points to: in place)
    const float* g1; const float* b1;   // LayerNorm
    const float* g2; const float* b2;   // whitening LayerNorm (nullable: skip)
    float* y;                // [M, 768] fp32 output, nullable
    bf16_t* yb;              // [M, 768] bf16 output, nullable
};
struct OutLnChainArgs {
    OutLnChainProblem p[2];  // one per modality (blockIdx.y)
    int M;                   // multiple of 64
    float eps;
    unsigned wave_stride;
    int stagger;             // x 8128 cycles of start delay for every other one of the first 256 workgroups (0 = none)
    unsigned long long* diag; // OC_DIAG builds only: 8 s_memtime stamps per workgroup (tools/outproj_diag.py)
};
#ifdef OC_DIAG
#define OC_STAMP(i) do { if (args.diag && t == 0) args.diag[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OC_STAMP(i)
#endif

__global__ __launch_bounds__(512, 2) void iefvad_outproj_ln_chain_bf16_kernel(OutLnChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    const OutLnChainProblem& P = args.p[blockIdx.y];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * OC_BM;
    OC_STAMP(0);
    // De-phasing: the launch's first 256 workgroups (one per CU) start together and every block takes the same time, so chip-wide
    // the load phases (HBM saturated, matrix pipe idle) and the multiply phases (HBM idle) coincide.  Every other one of them
    // waits about half a block time once; from then on half of the CUs load or store while the other half multiplies.
    if (args.stagger > 0 && blockIdx.y == 0 && blockIdx.x < 256 && (blockIdx.x & 1))
        for (int i = 0; i < args.stagger; ++i) __builtin_amdgcn_s_sleep(127);

    // ---- requests in the order they are needed: the activation image (must be in LDS before the first MFMA), the first weight
    // pieces, then the residual rows of the first half (needed only after the main loop).  vmcnt completes in order: with the
    // residual in front, the image's wait also waited for 96 KB that nobody needs for another 10 us (9.6 us before the first
    // MFMA instead of ~4: profiles/r03_outproj_chain_phase_stamps.log).
    // The image: 64 rows x 96 chunks of 16 bytes, chunk c of row r at r * 1536 + ((c & ~15) | ((c ^ r) & 15)) * 16
    f32x4 tmp[12];
    {
        const bf16_t* Ab = P.A + (size_t)m0 * IEF_D;
#pragma unroll
        for (int i = 0; i < 12; ++i) tmp[i] = *(const f32x4*)((const char*)Ab + (size_t)(t + 512 * i) * 16);
    }
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(P.stream + (size_t)wave * args.wave_stride), 0, (int)args.wave_stride, 0x00020000);
    const int vlane = lane * 16;
#define OC_LOAD(piece_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (int)((piece_) << 10), 0))
    f32x4 rg[OC_DEPTH];
#pragma unroll
    for (int s = 0; s < OC_DEPTH; ++s) rg[s] = OC_LOAD(s);
    // the residual rows this wave normalises in the first half (rows 4 wave .. + 3 of the block), LayerNorm lane layout
    f32x4 res[4][3];
#define OC_FETCH_RES(half_)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                         \
        const float* rp = P.R + (size_t)(m0 + 32 * (half_) + 4 * wave + u) * IEF_D + 4 * lane;             \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) res[u][j] = *(const f32x4*)(rp + 256 * j);           \
    }
    OC_FETCH_RES(0)       // (requested after the image barrier instead: image ready 20 k -> 13 k cycles, main loop 20 k -> 27 k: the
                          // 96 KB take their time wherever they sit in the wave's in-order queue; profiles/r03_outproj_chain_phase_stamps.log)
    // bias and the LayerNorms' affine terms go to LDS once per workgroup (15 KB behind the park): ln_row loads them from global
    // memory per row -- six dependent L2 round trips per row that two resident waves per SIMD cannot hide (3.4 k cycles per
    // row in the first build) -- and keeping them in registers (60) beside two residual sets made hipcc spill
    const bool two = P.g2 != nullptr;
    f32x4 aff[2];
    {
        // 5 arrays x 192 float4 = 15 slots of 64; slot 2 wave + i, so the array is wave-uniform (a per-lane choice of the
        // pointer makes hipcc fetch the pointer itself with a vector load and wait vmcnt(0) for it)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int slot = (2 * wave + i) < 15 ? (2 * wave + i) : 14, k = slot / 3;
            const float* sp = k == 0 ? P.bias : k == 1 ? P.g1 : k == 2 ? P.b1 : k == 3 ? (two ? P.g2 : P.g1) : (two ? P.b2 : P.b1);
            aff[i] = *(const f32x4*)(sp + 4 * ((slot % 3) * 64 + lane));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int id = t + 512 * i, r = id / 96, c = id - r * 96;
        *(f32x4*)(lds + r * (IEF_D * 2) + ((c & ~15) | ((c ^ r) & 15)) * 16) = tmp[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int slot = (2 * wave + i) < 15 ? (2 * wave + i) : 14;
        *(f32x4*)(lds + OC_AFF_OFF + (slot * 64 + lane) * 16) = aff[i];
    }
    int rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rd[j] = m * (IEF_D * 2) + (((4 * j + q) ^ m) & 15) * 16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();
    OC_STAMP(1);

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
                const f32x4 w = rg[b];                      // six pieces per k-step: ring position = column tile
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, ga[a]), acc[a][b], 0, 0, 0);
                rg[b] = OC_LOAD(p + OC_DEPTH);
                __builtin_amdgcn_sched_barrier(0);
                ++p;
            }
        }
    }
#undef OC_LOAD
#pragma unroll
    for (int s = 0; s < OC_DEPTH; ++s) asm volatile("" :: "v"(rg[s]));      // the read-ahead (zero pad pieces) must land before the wave ends
    OC_STAMP(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();                        // every wave is done with the image: the parks reuse its space
    OC_STAMP(3);

    // ---- epilogue: two halves of 32 rows; wave w normalises rows 4 w .. 4 w + 3 of each half
    const float* affl = (const float*)(lds + OC_AFF_OFF) + 4 * lane;      // + 768 i (bias, g1, b1, g2, b2) + 256 j
    float* park = (float*)lds;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < OC_NB; ++b)
                *(f32x4*)(park + (16 * a2 + m) * OC_PARK_LD + 96 * wave + 16 * b + 4 * q) = acc[2 * half + a2][b];
        f32x4 rcur[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) rcur[u][j] = res[u][j];
        if (half == 0) { OC_FETCH_RES(1) }       // in flight while the first 32 rows are normalised (their accumulators are parked)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();                    // all parks of this half are complete
        OC_STAMP(4 + 2 * half);
        // the wave's four rows together (ln_rows: the rows' reduction chains interleaved; per row exactly ln_row's operations)
        f32x4 v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f32x4 e = *(const f32x4*)(park + (4 * wave + u) * OC_PARK_LD + 4 * lane + 256 * j);
                v[u][j] = (e + *(const f32x4*)(affl + 256 * j)) + rcur[u][j];
            }
        ln_rows<4>(v, affl + IEF_D - 4 * lane, affl + 2 * IEF_D - 4 * lane, lane, args.eps);       // ln_rows adds 4 lane itself
        if (two) ln_rows<4>(v, affl + 3 * IEF_D - 4 * lane, affl + 4 * IEF_D - 4 * lane, lane, args.eps);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t row = (size_t)(m0 + 32 * half + 4 * wave + u);
            if (P.y) {
                float* yp = P.y + row * IEF_D + 4 * lane;
#pragma unroll
                for (int j = 0; j < 3; ++j) *(f32x4*)(yp + 256 * j) = v[u][j];
            }
            if (P.yb) {
                bf16_t* yb = P.yb + row * IEF_D + 4 * lane;
#pragma unroll
                for (int j = 0; j < 3; ++j) *(bf16x4_t*)(yb + 256 * j) = to_bf16x4(v[u][j]);
            }
        }
        OC_STAMP(5 + 2 * half);
        if (half == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GB2_BARRIER();                // every reader is done with this half's parks: they may be overwritten
        }
    }
#undef OC_FETCH_RES
}
