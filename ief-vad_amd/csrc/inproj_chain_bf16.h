// bf16 mode: the attention in_proj (q | k | v = x W_in^T + b, q pre-scaled for the softmax;
// /root/reference/model/imf_vad.py:115,121 -> nn.MultiheadAttention's packed in_proj) on the row-block structure of
// outproj_ln_chain_bf16.h: a 512-thread workgroup owns 64 rows x all 2304 output columns;
//   * the 64 x 768 activation block is ONE 96 KB bf16 image in LDS, loaded once -- from the bf16 rows the previous layer's
//     LayerNorm wrote, or (A32, the first layer) from the fp32 rows themselves, rounded to bf16 on the way in: the
//     stand-alone cast kernel and its bf16 copy of the inputs disappear;
//   * every wave streams ITS OWN 96 columns of q, then of k, then of v (three passes over k, 96 accumulators) as 1 KB pieces
//     in fragment order (iefvad_wstream_pack_kernel), six in flight in registers; the stream runs on across the passes, so
//     a pass's epilogue overlaps the next pass's first pieces; no barrier after the image is in place;
//   * a pass's epilogue goes through wave-private 3 KB LDS tiles (16 rows x 96 columns bf16, two per wave): bias, q scale, bf16, 16-byte
//     non-temporal stores of 192-byte row segments.
// The 256 x 256 ring kernel (gemm_bf16.h) re-reads a 96 KB A panel nine times through L2 for 256 rows and stops at every
// k-tile barrier; here the weights move (3.5 MB per block, the measured 110-120 GB/s per CU) and the rows stay.
// Same products in the same k order, the ring kernel's epilogue arithmetic ((acc + b) * s -> bf16): bit-identical q | k | v
// (IEFVAD_ROWBLOCK_OFF=1 selects the ring kernel for the A/B).
#pragma once
#include "outproj_ln_chain_bf16.h"

#define IC_BM 64
#define IC_NPASS 3
#define IC_STAGE_LD 104                                   // bf16 per staged row (208 B)
#define IC_STAGE_BYTES (16 * IC_STAGE_LD * 2)             // 3,328 B per wave
#define IC_LDS_BYTES (OC_IMG_BYTES + 16 * IC_STAGE_BYTES) // 151,552 B: two staging tiles per wave

struct InProjChainProblem {
    const void* A;           // [M, 768] rows: fp32 (A32) or bf16
    const char* stream;      // iefvad_wstream_pack_kernel(in_proj_weight, 3 passes)
    const float* bias;       // [2304]
    bf16_t* C;               // bf16 q | k | v, head-major: [3][8 heads][M][96] (AttnBArgs.head_major): a wave's 96 columns of a pass are one head
};
struct InProjChainArgs {
    InProjChainProblem p[2]; // one per modality (blockIdx.y)
    int M;                   // multiple of 64
    float alpha;             // scale of the q columns
    unsigned wave_stride;
    unsigned long long* diag; // IC_DIAG builds only: 8 s_memtime stamps per workgroup (tools/rowblock_diag.py)
};
#ifdef IC_DIAG
#define IC_STAMP(i) do { if (args.diag && t == 0) args.diag[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define IC_STAMP(i)
#endif

template <bool A32>
__device__ __forceinline__ void inproj_chain_body(const InProjChainArgs& args, char* lds) {
    const InProjChainProblem& P = args.p[blockIdx.y];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * IC_BM;
    IC_STAMP(0);

    // ---- the image: 64 rows x 96 chunks of 16 bytes (8 bf16), chunk c of row r at r * 1536 + ((c & ~15) | ((c ^ r) & 15)) * 16
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(P.stream + (size_t)wave * args.wave_stride), 0, (int)args.wave_stride, 0x00020000);
    const int vlane = lane * 16;
#define IC_LOAD(piece_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (int)((piece_) << 10), 0))
    f32x4 rg[OC_DEPTH];
    if constexpr (A32) {
        const float* Af = (const float*)P.A + (size_t)m0 * IEF_D;
        f32x4 lo[12], hi[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const float* ap = Af + (size_t)(t + 512 * i) * 8;
            lo[i] = *(const f32x4*)ap;
            hi[i] = *(const f32x4*)(ap + 4);
        }
#pragma unroll
        for (int s = 0; s < OC_DEPTH; ++s) rg[s] = IC_LOAD(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int id = t + 512 * i, r = id / 96, c = id - r * 96;
            const bf16x4_t a = to_bf16x4(lo[i]), b = to_bf16x4(hi[i]);      // the cast kernel's rounding
            bf16x8 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) { w[e] = a[e]; w[4 + e] = b[e]; }
            *(bf16x8*)(lds + r * (IEF_D * 2) + ((c & ~15) | ((c ^ r) & 15)) * 16) = w;
        }
    } else {
        const bf16_t* Ab = (const bf16_t*)P.A + (size_t)m0 * IEF_D;
        f32x4 tmp[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) tmp[i] = *(const f32x4*)((const char*)Ab + (size_t)(t + 512 * i) * 16);
#pragma unroll
        for (int s = 0; s < OC_DEPTH; ++s) rg[s] = IC_LOAD(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int id = t + 512 * i, r = id / 96, c = id - r * 96;
            *(f32x4*)(lds + r * (IEF_D * 2) + ((c & ~15) | ((c ^ r) & 15)) * 16) = tmp[i];
        }
    }
    int rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rd[j] = m * (IEF_D * 2) + (((4 * j + q) ^ m) & 15) * 16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();
    IC_STAMP(1);

    bf16_t* stage0 = (bf16_t*)(lds + OC_IMG_BYTES + wave * (2 * IC_STAGE_BYTES));      // two tiles: row tile a + 1 is written while a's rows are read back
    int p = 0;
#pragma unroll 1
    for (int pass = 0; pass < IC_NPASS; ++pass) {
        // the pass's bias in the accumulator's lane order: columns 768 pass + 96 wave + 16 b + 4 q .. + 3
        f32x4 bv[OC_NB];
        {
            const float* bp = P.bias + IEF_D * pass + 96 * wave + 4 * q;
#pragma unroll
            for (int b = 0; b < OC_NB; ++b) bv[b] = *(const f32x4*)(bp + 16 * b);
        }
        f32x4 acc[4][OC_NB];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < OC_NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                    rg[b] = IC_LOAD(p + OC_DEPTH);
                    __builtin_amdgcn_sched_barrier(0);
                    ++p;
                }
            }
        }
        IC_STAMP(2 + 2 * pass);
        // ---- epilogue of the pass: (acc + bias) * scale -> bf16, 16 rows at a time through the wave's private tile.
        // accumulator tile (a, b): lane (m, q) holds row 16 a + m, columns 16 b + 4 q .. + 3
        const float sc = pass == 0 ? args.alpha : 1.f;
        bf16_t* cbase = P.C + ((size_t)(pass * IEF_H + wave) * args.M + m0) * IEF_DH;      // plane (pass, head = wave), contiguous 12 KB per wave
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bf16_t* stage = stage0 + (a & 1) * (IC_STAGE_BYTES / 2);
#pragma unroll
            for (int b = 0; b < OC_NB; ++b) {
                f32x4 v = acc[a][b] + bv[b];
                v = v * sc;
                *(bf16x4_t*)(stage + m * IC_STAGE_LD + 16 * b + 4 * q) = to_bf16x4(v);
            }
            // (wave-private: LDS operations of one wave complete in order, no barrier)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int id = lane + 64 * i, row = id / 12, ch = id - row * 12;
                const bf16x8 w = *(const bf16x8*)(stage + row * IC_STAGE_LD + 8 * ch);
                GB2_STORE((bf16x8*)(cbase + (size_t)(16 * a + row) * IEF_DH + 8 * ch), w);
            }
        }
        IC_STAMP(3 + 2 * pass);
    }
#undef IC_LOAD
#pragma unroll
    for (int s = 0; s < OC_DEPTH; ++s) asm volatile("" :: "v"(rg[s]));      // the read-ahead (zero pad pieces) must land before the wave ends
}

__global__ __launch_bounds__(512, 2) void iefvad_inproj_chain_bf16_kernel(InProjChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    inproj_chain_body<false>(args, (char*)smem);
}

// first layer: fp32 rows in, rounded to bf16 while the image is built
__global__ __launch_bounds__(512, 2) void iefvad_inproj_chain_f32in_kernel(InProjChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    inproj_chain_body<true>(args, (char*)smem);
}
