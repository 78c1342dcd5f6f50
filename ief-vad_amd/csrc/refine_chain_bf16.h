// bf16 mode: the whole refinement tail -- K steps z <- z - lambda (W2 relu(W1 z + b1) + b2) and the scorer
// (/root/reference/model/imf_vad.py:146-150) -- in ONE kernel with the fusion state on chip.
//
// Unfused, every step is two projection launches that move 12 KB per snippet through HBM (bf16 z in, bf16 h out and
// back in, fp32 z in and out, bf16 z out): 2K launches, HBM-bound.  Here a workgroup owns 64 rows for
// the whole chain:
//   * z stays FP32 IN REGISTERS (8 waves x [64 rows x 96 columns] = 96 VGPRs per lane), next to 48 accumulator registers
//     (a wave computes its 96 columns in two passes of 48);
//   * ONE 96 KB bf16 activation image in LDS alternates between bf16(z) (operand of W1) and h (operand of W2): each is
//     dead when the other is written (two workgroup barriers per projection, nothing else synchronises the waves);
//   * the weights stream straight into registers: every wave reads ITS OWN 96 output columns of W1_0, W2_0, W1_1, ... as
//     one linear sequence of 1 KB pieces (one piece = one MFMA fragment of the wave, laid out in lane order by
//     iefvad_chain_pack_kernel at iefvad_set_weights; the biases ride the same stream as pieces in the accumulator's lane
//     order), six pieces (24 registers) in flight per wave, each register set refilled right after the four MFMAs that
//     read it -- no LDS ring, no barrier, hipcc's own counted s_waitcnt vmcnt, and the stream keeps running across
//     projection boundaries.  (Round 3's first version filled a private LDS ring by LDS-DMA: an LDS-DMA instruction costs
//     its wave 60-180 issue cycles per KB beside 64 cycles of MFMA work for that KB, and the kernel sat at 0.50 MFMA-busy and
//     65 GB/s of weight stream per CU where the stream alone runs 110-120: tools/ingest_probe.hip, tools/refine_chain_bf16_v1_ldsdma_ring.h.)
//   * the MFMA is issued with the operands swapped (A = weight fragment, B = activation fragment), so a lane holds four
//     CONSECUTIVE output columns of one row: h / bf16(z) go back into the image as one 8-byte LDS store per tile, and the
//     products and their k order are those of the projection kernels (gemm_bf16.h): z and the logits are bit-identical
//     to the 2K-launch path (tests/test_gpu_bf16.py, IEFVAD_ROWBLOCK_OFF=8 is the A/B switch);
//   * the scorer runs on the resident state: the final z is parked in LDS (fp32, 32 rows at a time) and reduced by the
//     scorer kernel's own code, 4 bytes per snippet leave the chip (plus z itself when the caller asked for `fused`).
// HBM traffic of the tail per snippet: 3 KB in (z from the heads + fusion kernel), 4 B out, against 10 x 12 KB.
// The weight stream is what feeds it: 2K x 1.2 MB per 64 rows from L2 (every workgroup reads the same bytes).
#pragma once
#include "gemm_bf16.h"
#include "rowops.h"

#define RC_BM 64
#define RC_IMG_BYTES (RC_BM * IEF_D * 2)                    // 98,304 B: [64 rows][768 k] bf16, 16-byte chunks XOR-swizzled
#define RC_NW 8                                             // waves per workgroup
#define RC_KT (IEF_D / 32)                                  // 24 k-steps of 32
#define RC_NB 6                                             // 16-column tiles per wave (96 columns)
#define RC_WCOLS (16 * RC_NB)                               // columns per wave
// A wave computes its 64 x 96 block in TWO passes over k, three column tiles (48 accumulator registers) each: with all six
// tiles live beside the 96 registers of z hipcc spills ~140 registers per projection around the main loop.  The activation
// image is read twice per projection instead.
#ifndef RC_NPASS
#define RC_NPASS 2                                          // -DRC_NPASS=3 (32 accumulator registers, 10 pieces in flight): measured equal (profiles/r03_chain_kernel_ab.log)
#endif
#define RC_NBP (RC_NB / RC_NPASS)                           // column tiles per pass
#define RC_PASS_PIECES (RC_KT * RC_NBP + RC_NBP)            // a pass's weight fragments + its bias pieces (75 / 50)
#define RC_PIECES (RC_NPASS * RC_PASS_PIECES)               // pieces per projection and wave
#if RC_NPASS == 2
#define RC_DEPTH 6                                          // weight pieces in flight per wave (a ring of DEPTH x 4 registers)
#define RC_KUNROLL 4                                        // k-steps per loop iteration: KUNROLL x NBP pieces = 0 mod DEPTH
#else
#define RC_DEPTH 10                                         // 50 pieces per pass = 0 mod 10; the k loop is fully unrolled
#define RC_KUNROLL RC_KT
#endif
#define RC_PAD_PIECES RC_DEPTH                              // zero pieces behind a wave's stream (the ring reads ahead)
#define RC_PARK_BYTES ((RC_NPASS - 1) * 4 * RC_NBP * 64 * 8) // per wave: the h tiles of a first projection's earlier passes wait here
#define RC_LDS_BYTES (RC_IMG_BYTES + RC_NW * RC_PARK_BYTES) // 147,456 B

static inline size_t chain_wave_stride_bytes(int K) { return ((size_t)2 * K * RC_PIECES + RC_PAD_PIECES) * 1024; }
static inline size_t chain_stream_bytes(int K) { return RC_NW * chain_wave_stride_bytes(K); }

// ---- the weight stream.  Per wave w and projection g: pass 0's pieces, then pass 1's; a pass = its 72 weight pieces in
// (k-step kt, column tile b) order, then its three bias pieces.  Weight piece, lane l = (r = l & 15, q = l >> 4):
// 8 bf16 = W_g[96 w + 16 b + r][32 kt + 8 q .. + 7], i.e. the lane's A-operand fragment of v_mfma_f32_16x16x32_bf16;
// bias piece b: lane (m, q) holds bias_g[96 w + 16 b + 4 q .. + 3], the four columns its accumulator registers of tile b
// cover.  One thread per 16 bytes.
struct ChainPackArgs {
    const bf16_t* W[2 * IEFVAD_MAX_STEPS];     // [768, 768] bf16 each: W1_0, W2_0, W1_1, ...
    const float* bias[2 * IEFVAD_MAX_STEPS];
    char* stream;
    int K;
};

__global__ __launch_bounds__(256) void iefvad_chain_pack_kernel(ChainPackArgs a) {
    const size_t per_wave = ((size_t)2 * a.K * RC_PIECES + RC_PAD_PIECES) * 64;      // 16-byte units
    const size_t total = RC_NW * per_wave;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(u / per_wave);
        const size_t v = u - (size_t)w * per_wave;
        const int lane = (int)(v & 63);
        const size_t piece = v >> 6;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (piece < (size_t)2 * a.K * RC_PIECES) {
            const int g = (int)(piece / RC_PIECES), pg = (int)(piece % RC_PIECES);
            const int pass = pg / RC_PASS_PIECES, pi = pg % RC_PASS_PIECES;
            if (pi < RC_KT * RC_NBP) {
                const int kt = pi / RC_NBP, b = RC_NBP * pass + pi % RC_NBP;
                const int r = lane & 15, q = lane >> 4;
                val = *(const f32x4*)(a.W[g] + (size_t)(RC_WCOLS * w + 16 * b + r) * IEF_D + 32 * kt + 8 * q);
            } else {
                // bias piece b of the pass, in the accumulator's lane order: lane (m, q) needs columns 16 b + 4 q .. + 3
                const int b = RC_NBP * pass + (pi - RC_KT * RC_NBP), q = lane >> 4;
                val = *(const f32x4*)(a.bias[g] + RC_WCOLS * w + 16 * b + 4 * q);
            }
        }
        *(f32x4*)(a.stream + u * 16) = val;
    }
}

struct ChainArgs {
    const float* z_in;       // [M, 768] fp32: the fused state z_0
    const char* stream;      // iefvad_chain_pack_kernel's output
    const float* cls_w;      // classifier.weight [768]
    const float* cls_b;      // classifier.bias [1]
    float* z_out;            // [M, 768] fp32 final state, nullable
    float* logits;           // [M]
    int M;                   // multiple of 64
    int K;                   // >= 1
    float lambda;
    unsigned wave_stride;    // bytes between the waves' streams
    unsigned long long* diag; // RC_DIAG builds only: 16 s_memtime stamps per workgroup over projections 2 and 3 (tools/rowblock_diag.py chain)
};
#ifdef RC_DIAG
#define RC_STAMP(i) do { if (args.diag && t == 0 && (g == 2 || g == 3)) args.diag[(size_t)blockIdx.x * 16 + (g - 2) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RC_STAMP(i)
#endif

__global__ __launch_bounds__(512, 2) void iefvad_refine_chain_bf16_kernel(ChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * RC_BM;

    // ---- the wave's weight stream: RC_DEPTH pieces in flight, in registers
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(args.stream + (size_t)wave * args.wave_stride), 0,
                                                      (int)args.wave_stride, 0x00020000);
    const int vlane = lane * 16;
#define RC_LOAD(piece_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (int)((piece_) << 10), 0))
    f32x4 rg[RC_DEPTH];
#pragma unroll
    for (int s = 0; s < RC_DEPTH; ++s) rg[s] = RC_LOAD(s);

    // ---- state: z[a][b] = rows 16 a + m, columns 96 wave + 16 b + 4 q .. + 3 (the swapped-operand accumulator map)
    f32x4 z[4][RC_NB];
    {
        const float* zp = args.z_in + (size_t)(m0 + m) * IEF_D + RC_WCOLS * wave + 4 * q;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < RC_NB; ++b) z[a][b] = *(const f32x4*)(zp + (size_t)16 * a * IEF_D + 16 * b);
    }
    // image addressing.  Chunk c (16 bytes = 8 k) of row r lives at r * 1536 + ((c & ~15) | ((c ^ r) & 15)) * 16: conflict-free
    // for the 16-lane groups of the ds_read_b128 fragment reads (lane (m, q) reads row 16 a + m, chunk 4 kt + q: the groups
    // {0-3, 12-15, 20-27}, ... of MI355X_MICROARCH.md cover all sixteen 16-byte slots of a bank row) and 2-way -- the minimum
    // for sixteen rows at one 8-byte column -- for the ds_write_b64 tile stores (c ^ 2 r, the first version, was 4-way there:
    // 2.4e8 conflict cycles per launch, profiles/r03_kernel_pmc_summary_bf16.txt).
    const int sw = m;
    int rd[4];                                      // fragment read offsets for kt & 3 = 0..3 (add 24,576 a + 256 (kt >> 2))
#pragma unroll
    for (int j = 0; j < 4; ++j) rd[j] = m * (IEF_D * 2) + (((4 * j + q) ^ sw) & 15) * 16;
    // store of tile (a, b): row 16 a + m, columns 96 wave + 16 b + 4 q .. + 3 -> chunk 12 wave + 2 b + (q >> 1), half q & 1
    // (mo, qo: the lane's m, q as the epilogue sees them -- made opaque there, so that hipcc recomputes these few-instruction
    // offsets per tile instead of hoisting all 24 of them out of the projection loop into registers it then has to spill)
    int mo = m, qo = q;
    auto img_off = [&](int a, int b) {
        const int c = (RC_WCOLS / 8) * wave + 2 * b + (qo >> 1);
        return (16 * a + mo) * (IEF_D * 2) + ((c & ~15) | ((c ^ mo) & 15)) * 16 + (qo & 1) * 8;
    };
    char* park = lds + RC_IMG_BYTES + wave * RC_PARK_BYTES + lane * 8;      // tile (a, b) of pass pp: + ((pp * 4 + a) * RC_NBP + b) * 512
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < RC_NB; ++b) *(bf16x4_t*)(lds + img_off(a, b)) = to_bf16x4(z[a][b]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();

    const float lambda = args.lambda;
    const int G = 2 * args.K;
    int p = 0;                                      // next piece of the stream to consume (pieces p .. p + 5 are in flight)
    for (int g = 0; g < G; ++g) {
        const bool first = (g & 1) == 0;            // h = relu(z W1^T + b1); else z <- z - lambda (h W2^T + b2)
#pragma unroll
        for (int pass = 0; pass < RC_NPASS; ++pass) {
            // ring position of the pass's piece i: (i + pass x PASS_PIECES) % DEPTH (two passes: a pass is 75 pieces = 3 mod 6, a
            // projection 150 = 0 mod 6; three passes: 50 = 0 mod 10)
            constexpr int kRingOff = 0;
            const int ring_off = (pass * RC_PASS_PIECES) % RC_DEPTH;
            f32x4 acc[4][RC_NBP];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < RC_NBP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            RC_STAMP(3 * pass);
#if RC_KUNROLL == RC_KT
#pragma unroll
#else
#pragma unroll 1
#endif
            for (int k4 = 0; k4 < RC_KT / RC_KUNROLL; ++k4) {
#pragma unroll
                for (int jj = 0; jj < RC_KUNROLL; ++jj) {
                    const int j = jj & 3, k4q = (RC_KUNROLL == RC_KT) ? (jj >> 2) : k4;      // kt = 4 k4q + j
                    f32x4 ga[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) ga[a] = *(const f32x4*)(lds + rd[j] + a * (16 * IEF_D * 2) + k4q * 256);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < RC_NBP; ++b) {
                        const int ri = (jj * RC_NBP + b + ring_off + kRingOff) % RC_DEPTH;      // the iteration's pieces = 0 mod DEPTH
                        const f32x4 w = rg[ri];
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, ga[a]),
                                                                                acc[a][b], 0, 0, 0);
                        rg[ri] = RC_LOAD(p + RC_DEPTH);
                        __builtin_amdgcn_sched_barrier(0);      // keep the refill HERE: hipcc otherwise sinks the loads and drains the ring
                        ++p;
                    }
                }
            }

            // ---- the pass's epilogue; its bias pieces are the next ones of the ring
            RC_STAMP(3 * pass + 1);
            asm volatile("" : "+v"(mo), "+v"(qo));
            f32x4 bias[RC_NBP];
#pragma unroll
            for (int b = 0; b < RC_NBP; ++b) {
                const int ri = (RC_KT * RC_NBP + b + ring_off) % RC_DEPTH;
                bias[b] = rg[ri];
                rg[ri] = RC_LOAD(p + RC_DEPTH);
                ++p;
            }
            if (pass == RC_NPASS - 1) GB2_BARRIER();      // every wave is done reading the image: it may be rewritten
            RC_STAMP(3 * pass + 2);
            if (first) {
#pragma unroll
                for (int b = 0; b < RC_NBP; ++b)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        f32x4 v = acc[a][b] + bias[b];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];
                        if (pass < RC_NPASS - 1)      // the image is still being read: the tile waits in the wave's own park
                            *(bf16x4_t*)(park + ((pass * 4 + a) * RC_NBP + b) * 512) = to_bf16x4(v);
                        else {
#pragma unroll
                            for (int pp = 0; pp < RC_NPASS - 1; ++pp)
                                *(bf16x4_t*)(lds + img_off(a, RC_NBP * pp + b)) = *(const bf16x4_t*)(park + ((pp * 4 + a) * RC_NBP + b) * 512);
                            *(bf16x4_t*)(lds + img_off(a, RC_NBP * pass + b)) = to_bf16x4(v);
                        }
                    }
            } else {
#pragma unroll
                for (int b = 0; b < RC_NBP; ++b)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const f32x4 v = acc[a][b] + bias[b];
                        const int bb = RC_NBP * pass + b;
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[a][bb][e] = __builtin_fmaf(-lambda, v[e], z[a][bb][e]);
                        if (pass == RC_NPASS - 1) {       // bf16(z) -> image for the next step (after the last step nobody reads it)
#pragma unroll
                            for (int pp = 0; pp < RC_NPASS; ++pp)
                                *(bf16x4_t*)(lds + img_off(a, RC_NBP * pp + b)) = to_bf16x4(z[a][RC_NBP * pp + b]);
                        }
                    }
            }
            if (pass == RC_NPASS - 1) {
                RC_STAMP(6);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                GB2_BARRIER();                            // the image holds the next operand
                RC_STAMP(7);
            }
        }
    }
#undef RC_LOAD
    // the ring's read-ahead (zero pad pieces) is still in flight: keep it alive until it has landed
#pragma unroll
    for (int s = 0; s < RC_DEPTH; ++s) asm volatile("" :: "v"(rg[s]));

    // ---- final state out (only when the caller asked for `fused`)
    if (args.z_out) {
        float* zp = args.z_out + (size_t)(m0 + m) * IEF_D + RC_WCOLS * wave + 4 * q;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < RC_NB; ++b) *(f32x4*)(zp + (size_t)16 * a * IEF_D + 16 * b) = z[a][b];
    }

    // ---- scorer (imf_vad.py:150) on the resident state: 32 rows at a time through LDS (fp32, [32][768] = 96 KB, the image's
    // space), then iefvad_scorer_kernel's own reduction: lane l owns columns 4 l + 256 j, wave w rows 4 w .. 4 w + 3.
    f32x4 wv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) wv[j] = *(const f32x4*)(args.cls_w + 4 * lane + 256 * j);
    const float cb = args.cls_b[0];
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < RC_NB; ++b)
                *(f32x4*)(lds + (size_t)((16 * a2 + m) * IEF_D + RC_WCOLS * wave + 16 * b + 4 * q) * 4) = z[2 * hlf + a2][b];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GB2_BARRIER();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = 4 * wave + u;
            const float* zp = (const float*)lds + r * IEF_D + 4 * lane;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f32x4 zv = *(const f32x4*)(zp + 256 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) s += zv[e] * wv[j][e];
            }
            s = wave_sum(s);
            if (lane == 0) args.logits[m0 + 32 * hlf + r] = s + cb;
        }
        if (hlf == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GB2_BARRIER();
        }
    }
}
