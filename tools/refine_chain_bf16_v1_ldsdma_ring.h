// bf16 mode: the whole refinement tail -- K steps z <- z - lambda (W2 relu(W1 z + b1) + b2) and the scorer
// (/root/reference/model/imf_vad.py:146-150) -- in ONE kernel with the fusion state on chip.
//
// Unfused, every step is two projection launches that move 12 KB per snippet through HBM (bf16 z in, bf16 h out and
// back in, fp32 z in and out, bf16 z out): 2K launches, HBM-bound.  Here a workgroup owns 64 rows for
// the whole chain:
//   * z stays FP32 IN REGISTERS (8 waves x [64 rows x 96 columns] = 96 VGPRs per lane), next to 96 accumulator registers;
//   * ONE 96 KB bf16 activation image in LDS alternates between bf16(z) (operand of W1) and h (operand of W2): each is
//     dead when the other is written (two workgroup barriers per projection, nothing else synchronises the waves);
//   * the weights never touch registers on their way in: every wave streams ITS OWN 96 output columns of W1_0, W2_0,
//     W1_1, ... as one linear sequence of 1 KB pieces (one piece = one MFMA fragment of the wave, laid out in lane
//     order by iefvad_chain_pack_kernel at iefvad_set_weights; the projection's 96 bias values ride the same stream as
//     one more piece) through a PRIVATE 8-piece LDS ring filled by LDS-DMA, ordered by the wave's own counted
//     s_waitcnt vmcnt(7) -- no barrier, no shared ring, and the stream keeps running across projection boundaries;
//   * the MFMA is issued with the operands swapped (A = weight fragment, B = activation fragment), so a lane holds four
//     CONSECUTIVE output columns of one row: h / bf16(z) go back into the image as one 8-byte LDS store per tile, and the
//     products and their k order are those of the projection kernels (gemm_bf16.h): z and the logits are bit-identical
//     to the 2K-launch path (tests/test_gpu_bf16.py, IEFVAD_NO_CHAIN=1 is the A/B switch);
//   * the scorer runs on the resident state: the final z is parked in LDS (fp32, 32 rows at a time) and reduced by the
//     scorer kernel's own code, 4 bytes per snippet leave the chip (plus z itself when the caller asked for `fused`).
// HBM traffic of the tail per snippet: 3 KB in (z from the heads + fusion kernel), 4 B out, against 10 x 12 KB.
// The weight stream is what bounds it: 2K x 1.18 MB per 64 rows from L2 (every workgroup reads the same bytes).
#pragma once
#include "gemm_bf16.h"
#include "rowops.h"

#define RC_BM 64
#define RC_IMG_BYTES (RC_BM * IEF_D * 2)                    // 98,304 B: [64 rows][768 k] bf16, 16-byte chunks XOR-swizzled
#define RC_NW 8                                             // waves per workgroup
#define RC_SLOTS 8                                          // pieces of 1 KB in a wave's private ring
#define RC_LDS_BYTES (RC_IMG_BYTES + RC_NW * RC_SLOTS * 1024)   // 163,840 B = all of a CU's LDS
#define RC_KT (IEF_D / 32)                                  // 24 k-steps of 32
#define RC_NB 6                                             // 16-column tiles per wave (96 columns)
#define RC_WCOLS (16 * RC_NB)                               // columns per wave
// A wave computes its 64 x 96 block in TWO passes over k, three column tiles (48 accumulator registers) each: with all six
// tiles live beside the 96 registers of z hipcc spills ~140 registers per projection around the main loop, and the spill
// reloads (ordinary loads) drain the LDS-DMA ring.  The activation image is read twice per projection instead.
#define RC_NPASS 2
#define RC_NBP (RC_NB / RC_NPASS)                           // column tiles per pass
#define RC_PASS_PIECES (RC_KT * RC_NBP + 1)                 // a pass's 72 weight fragments + its 48 bias values
#define RC_PIECES (RC_NPASS * RC_PASS_PIECES)               // pieces per projection and wave
#define RC_PAD_PIECES RC_SLOTS                              // zero pieces behind a wave's stream (the ring reads ahead)

static inline size_t chain_wave_stride_bytes(int K) { return ((size_t)2 * K * RC_PIECES + RC_PAD_PIECES) * 1024; }
static inline size_t chain_stream_bytes(int K) { return RC_NW * chain_wave_stride_bytes(K); }

// ---- the weight stream.  Piece (wave w, projection g, k-step kt, column tile b), lane l = (r = l & 15, q = l >> 4):
// 8 bf16 = W_g[96 w + 16 b + r][32 kt + 8 q .. + 7], i.e. the lane's A-operand fragment of v_mfma_f32_16x16x32_bf16;
// piece (w, g, bias): floats 0..95 = bias_g[96 w ..], the rest zero.  One thread per 16 bytes.
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
            } else if (lane < 4 * RC_NBP) {
                val = *(const f32x4*)(a.bias[g] + RC_WCOLS * w + 16 * RC_NBP * pass + 4 * lane);
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
};

__global__ __launch_bounds__(512, 2) void iefvad_refine_chain_bf16_kernel(ChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * RC_BM;

    // ---- the wave's weight stream and its private ring
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(args.stream + (size_t)wave * args.wave_stride), 0,
                                                      (int)args.wave_stride, 0x00020000);
    const int ring = RC_IMG_BYTES + wave * (RC_SLOTS * 1024);
    const int vlane = lane * 16;
#define RC_DMA(piece_)                                                                                              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + ring + (((piece_) & (RC_SLOTS - 1)) << 10)), \
                                             16, vlane, (int)((piece_) << 10), 0, 0)
#define RC_WAIT_PIECE() asm volatile("s_waitcnt vmcnt(7)" ::: "memory")      /* RC_SLOTS - 1 */
#pragma unroll
    for (int s = 0; s < RC_SLOTS; ++s) RC_DMA(s);

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
#ifdef RC_SWZ2
    const int sw = (2 * m) & 15;      // A/B: the first version's swizzle
#else
    const int sw = m;
#endif
    int rd[4];                                      // fragment read offsets for kt & 3 = 0..3 (add 24,576 a + 256 (kt >> 2))
#pragma unroll
    for (int j = 0; j < 4; ++j) rd[j] = m * (IEF_D * 2) + (((4 * j + q) ^ sw) & 15) * 16;
    // store of tile (a, b): row 16 a + m, columns 96 wave + 16 b + 4 q .. + 3 -> chunk 12 wave + 2 b + (q >> 1), half q & 1
    auto img_off = [&](int a, int b) {
        const int c = (RC_WCOLS / 8) * wave + 2 * b + (q >> 1);
        return (16 * a + m) * (IEF_D * 2) + ((c & ~15) | ((c ^ sw) & 15)) * 16 + (q & 1) * 8;
    };
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < RC_NB; ++b) *(bf16x4_t*)(lds + img_off(a, b)) = to_bf16x4(z[a][b]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();

    const float lambda = args.lambda;
    const int G = 2 * args.K;
    int p = 0;                                      // next piece of the stream to consume
    for (int g = 0; g < G; ++g) {
        const bool first = (g & 1) == 0;            // h = relu(z W1^T + b1); else z <- z - lambda (h W2^T + b2)
        bf16x4_t hp[4][RC_NBP];                     // pass 0 of a first projection: its h tiles wait here for the image
#pragma unroll
        for (int pass = 0; pass < RC_NPASS; ++pass) {
            f32x4 acc[4][RC_NBP];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < RC_NBP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

            // Software-pipelined over the pass's 72 pieces: while the four MFMAs of piece p issue, the fragment of piece p + 1 is
            // already on its way from the ring into the other register set, and the ring slot of piece p is being refilled.  (The
            // first version read, waited, refilled and only then multiplied: a wave's own read latency + DMA issue + MFMA chain,
            // ~260 cycles per piece, bounded the kernel at 0.50 MFMA-busy and 65 GB/s of weight stream per CU, while the stream
            // alone runs at 110 GB/s: tools/ingest_probe.hip, profiles/r03_chain_ingest_probe.log.)
            f32x4 gbuf[2];
            RC_WAIT_PIECE();                                          // the pass's first piece has landed
            gbuf[0] = *(const f32x4*)(lds + ring + ((p & (RC_SLOTS - 1)) << 10) + vlane);
#pragma unroll 1
            for (int k4 = 0; k4 < RC_KT / 4; ++k4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 ga[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) ga[a] = *(const f32x4*)(lds + rd[j] + a * (16 * IEF_D * 2) + k4 * 256);
#pragma unroll
                    for (int b = 0; b < RC_NBP; ++b) {
                        constexpr int kPar = 0;
                        const int cur = (j * RC_NBP + b + kPar) & 1;      // 12 pieces per iteration: the parity is the same in every iteration
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // piece p (and ga) are in registers: slot p may be refilled
                        RC_DMA(p + RC_SLOTS);
                        RC_WAIT_PIECE();                                      // pieces p + 1 .. p + 8 in flight -> p + 1 has landed
                        gbuf[cur ^ 1] = *(const f32x4*)(lds + ring + (((p + 1) & (RC_SLOTS - 1)) << 10) + vlane);   // (after the pass's last piece: the bias piece, unused)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gbuf[cur]), __builtin_bit_cast(bf16x8, ga[a]),
                                                                                acc[a][b], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        ++p;
                    }
                }
            }

            // ---- the pass's epilogue.  Its bias piece: 48 floats, lane (m, q) needs floats 16 b + 4 q .. + 3.
            RC_WAIT_PIECE();
            const char* bp = lds + ring + ((p & (RC_SLOTS - 1)) << 10) + 16 * q;
            if (pass == RC_NPASS - 1) GB2_BARRIER();      // every wave is done reading the image: it may be rewritten
            if (first) {
#pragma unroll
                for (int b = 0; b < RC_NBP; ++b) {
                    const f32x4 bv = *(const f32x4*)(bp + 64 * b);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        f32x4 v = acc[a][b] + bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] < 0.f) ? 0.f : v[e];
                        if (pass == 0) hp[a][b] = to_bf16x4(v);
                        else {
                            *(bf16x4_t*)(lds + img_off(a, b)) = hp[a][b];
                            *(bf16x4_t*)(lds + img_off(a, RC_NBP + b)) = to_bf16x4(v);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < RC_NBP; ++b) {
                    const f32x4 bv = *(const f32x4*)(bp + 64 * b);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const f32x4 v = acc[a][b] + bv;
                        const int bb = RC_NBP * pass + b;
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[a][bb][e] = __builtin_fmaf(-lambda, v[e], z[a][bb][e]);
                        if (pass == RC_NPASS - 1) {       // bf16(z) -> image for the next step (after the last step nobody reads it)
                            *(bf16x4_t*)(lds + img_off(a, b)) = to_bf16x4(z[a][b]);
                            *(bf16x4_t*)(lds + img_off(a, bb)) = to_bf16x4(z[a][bb]);
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            RC_DMA(p + RC_SLOTS);                         // the bias piece's slot is free
            ++p;
            if (pass == RC_NPASS - 1) GB2_BARRIER();      // the image holds the next operand
        }
    }
#undef RC_DMA
#undef RC_WAIT_PIECE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's read-ahead (pad pieces) must land before the LDS is released

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
