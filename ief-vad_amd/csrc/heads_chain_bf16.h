// bf16 mode: the mu / log-variance heads of BOTH modalities and the precision-weighted fusion
// (/root/reference/model/imf_vad.py:125-144) on the row-block structure of inproj_chain_bf16.h / outproj_ln_chain_bf16.h.
//
// A 512-thread workgroup owns 64 rows x one THIRD of the columns (256) of all four heads; a wave owns 32 columns:
//   phase 1: the 64 x 768 bf16 image of x_i in LDS; the wave streams W_mu_i and W_logvar_i of its 32 columns (1 KB pieces in
//            fragment order, eight in flight in registers) -> 2 heads x 2 column tiles x 4 row tiles of accumulators;
//   swap:    the x_e rows replace the image (their loads were requested when phase 1 ended);
//   phase 2: mu_e, logvar_e the same way -> all four quantities of a (row, column) now sit in ONE lane (128 accumulator
//            registers): biases, fuse_elem (rowops.h: the fusion kernel's own operations) and the stores follow in registers,
//            no LDS park, no partner wave.
// Weights are read once per 64 rows (4.7 MB over the three column thirds), the same L2 -> CU stream per row as the
// in_proj kernel's; round 2's 256 x 64 ring kernel (tools/heads_fused_bf16_v1.h) read 196 KB of A and W per k-tile barrier for the same work.
// Same products in the same k order, same epilogue arithmetic: mu, logvar, n_i, n_e and z are bit-identical to
// the unfused path (heads GEMM + fusion kernel: IEFVAD_ROWBLOCK_OFF=4); the row sums of n_i / n_e leave as 24 partials per row (32
// columns each) and are finished in a fixed order by iefvad_rowmean_finish_kernel.
#pragma once
#include "outproj_ln_chain_bf16.h"

#define HC_BM 64
#define HC_THIRDS 3
#define HC_NPART (HC_THIRDS * 8)                     // row-sum partials per row and modality
#define HC_DEPTH 8                                   // pieces in flight per wave = two k-steps
#define HC_PHASE_PIECES (OC_KT * 4)                  // 96: per k-step mu tile 0, mu tile 1, logvar tile 0, logvar tile 1
#define HC_PIECES (2 * HC_PHASE_PIECES)
#define HC_LDS_BYTES OC_IMG_BYTES

static inline size_t heads_stream_wave_stride_bytes() { return (size_t)(HC_PIECES + HC_DEPTH) * 1024; }
static inline size_t heads_stream_bytes() { return (size_t)HC_THIRDS * 8 * heads_stream_wave_stride_bytes(); }

// head matrices [1536, 768] bf16 (rows 0..767 mu, 768..1535 log-variance) of both modalities -> per (third c3, wave w):
// piece (phase = modality, kt, hd, tile), lane (r, q): 8 bf16 = W[phase][768 hd + 256 c3 + 32 w + 16 tile + r][32 kt + 8 q .. + 7]
__global__ __launch_bounds__(256) void iefvad_heads_pack_kernel(const bf16_t* Wi, const bf16_t* We, char* stream) {
    const size_t per_wave = (size_t)(HC_PIECES + HC_DEPTH) * 64;
    const size_t total = (size_t)HC_THIRDS * 8 * per_wave;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int cw = (int)(u / per_wave);          // 8 c3 + w
        const size_t v = u - (size_t)cw * per_wave;
        const int lane = (int)(v & 63);
        const int piece = (int)(v >> 6);
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (piece < HC_PIECES) {
            const int phase = piece / HC_PHASE_PIECES, pp = piece % HC_PHASE_PIECES;
            const int kt = pp >> 2, hd = (pp >> 1) & 1, tile = pp & 1, r = lane & 15, q = lane >> 4;
            const bf16_t* W = phase ? We : Wi;
            val = *(const f32x4*)(W + (size_t)(IEF_D * hd + 32 * cw + 16 * tile + r) * IEF_D + 32 * kt + 8 * q);
        }
        *(f32x4*)(stream + u * 16) = val;
    }
}

struct HeadsChainArgs {
    const bf16_t* A[2];      // x_i, x_e: [M, 768] bf16
    const char* stream;      // iefvad_heads_pack_kernel
    const float* bias[2];    // [1536] each
    float* mu[2];            // [M, 768] fp32, nullable
    float* lv[2];
    float* n[2];             // normalised precision weights n_i, n_e, nullable
    float* z;                // fused state [M, 768] fp32
    bf16_t* zb;              // its bf16 copy, nullable
    float* nsum_part;        // [M][2][HC_NPART] partial row sums of n_i, n_e, nullable
    int M;                   // multiple of 64
    float factor, eps;
    unsigned wave_stride;
    unsigned long long* diag; // HC_DIAG builds only: 8 s_memtime stamps per workgroup (tools/rowblock_diag.py)
};
#ifdef HC_DIAG
#define HC_STAMP(i) do { if (args.diag && t == 0) args.diag[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HC_STAMP(i)
#endif

__global__ __launch_bounds__(512, 2) void iefvad_heads_chain_bf16_kernel(HeadsChainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m = lane & 15, q = lane >> 4;
    const int c3 = blockIdx.y;
    const int m0 = blockIdx.x * HC_BM;
    HC_STAMP(0);

    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(args.stream + (size_t)(8 * c3 + wave) * args.wave_stride), 0, (int)args.wave_stride, 0x00020000);
    const int vlane = lane * 16;
#define HC_LOAD(piece_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (int)((piece_) << 10), 0))
    // the image: 64 rows x 96 chunks of 16 bytes, chunk c of row r at r * 1536 + ((c & ~15) | ((c ^ r) & 15)) * 16
    f32x4 tmp[12];
#define HC_FETCH_IMAGE(mod_)                                                                                \
    {                                                                                                       \
        const bf16_t* Ab = args.A[mod_] + (size_t)m0 * IEF_D;                                      \
        _Pragma("unroll") for (int i = 0; i < 12; ++i) tmp[i] = *(const f32x4*)((const char*)Ab + (size_t)(t + 512 * i) * 16); \
    }
#define HC_WRITE_IMAGE()                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 12; ++i) {                                                        \
        const int id = t + 512 * i, r = id / 96, c = id - r * 96;                                           \
        *(f32x4*)(lds + r * (IEF_D * 2) + ((c & ~15) | ((c ^ r) & 15)) * 16) = tmp[i];                      \
    }
    HC_FETCH_IMAGE(0)
    f32x4 rg[HC_DEPTH];
#pragma unroll
    for (int s = 0; s < HC_DEPTH; ++s) rg[s] = HC_LOAD(s);
    __builtin_amdgcn_sched_barrier(0);
    HC_WRITE_IMAGE()
    int rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rd[j] = m * (IEF_D * 2) + (((4 * j + q) ^ m) & 15) * 16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();
    HC_STAMP(1);

    // acc[phase][2 hd + tile][a]: lane (m, q) holds row 16 a + m, columns 256 c3 + 32 wave + 16 tile + 4 q .. + 3
    f32x4 acc[2][4][4];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[ph][b][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    int p = 0;
#define HC_MAIN(ph_)                                                                                        \
    _Pragma("unroll 1") for (int k4 = 0; k4 < OC_KT / 4; ++k4) {                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
            f32x4 ga[4];                                                                                    \
            _Pragma("unroll") for (int a = 0; a < 4; ++a) ga[a] = *(const f32x4*)(lds + rd[j] + a * (16 * IEF_D * 2) + k4 * 256); \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                 \
                const f32x4 w = rg[(j & 1) * 4 + b];                                                        \
                _Pragma("unroll") for (int a = 0; a < 4; ++a)                                               \
                    acc[ph_][b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, ga[a]), acc[ph_][b][a], 0, 0, 0); \
                rg[(j & 1) * 4 + b] = HC_LOAD(p + HC_DEPTH);                                                \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                ++p;                                                                                        \
            }                                                                                               \
        }                                                                                                   \
    }
    HC_MAIN(0)
    HC_STAMP(2);
    // ---- swap the image: x_e.  Requested only now: a load that misses L2 ahead of the weight pieces in the wave's in-order
    // queue stalls the ring behind it for the HBM latency, and touching the rows early (one dword per 64 bytes at entry) does
    // not help: under the 110 GB/s per CU weight stream a line lives ~1 us in the XCD's 4 MB L2
    // (profiles/r03_rowblock_phase_stamps.log).
    HC_FETCH_IMAGE(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();                        // every wave is done reading x_i
    HC_WRITE_IMAGE()
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GB2_BARRIER();
    HC_STAMP(3);
    HC_MAIN(1)
    HC_STAMP(4);
#undef HC_MAIN
#undef HC_LOAD
#undef HC_FETCH_IMAGE
#undef HC_WRITE_IMAGE
#pragma unroll
    for (int s = 0; s < HC_DEPTH; ++s) asm volatile("" :: "v"(rg[s]));      // the read-ahead (zero pad pieces) must land before the wave ends

    // ---- epilogue, in registers
    const int cbase = 256 * c3 + 32 * wave + 4 * q;
    float si[4] = {0.f, 0.f, 0.f, 0.f}, se[4] = {0.f, 0.f, 0.f, 0.f};        // per row tile a: this lane's share of the row sums
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        const int col = cbase + 16 * tile;
        const f32x4 bmi = *(const f32x4*)(args.bias[0] + col), bli = *(const f32x4*)(args.bias[0] + IEF_D + col);
        const f32x4 bme = *(const f32x4*)(args.bias[1] + col), ble = *(const f32x4*)(args.bias[1] + IEF_D + col);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const size_t o = (size_t)(m0 + 16 * a + m) * IEF_D + col;
            const f32x4 mi = acc[0][tile][a] + bmi;
            const f32x4 li = acc[0][2 + tile][a] + bli;
            const f32x4 me = acc[1][tile][a] + bme;
            const f32x4 le = acc[1][2 + tile][a] + ble;
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
            *(f32x4*)(args.z + o) = z;                      // read again by the refinement: not a streaming store
            if (args.zb) *(bf16x4_t*)(args.zb + o) = to_bf16x4(z);
        }
    }
    if (args.nsum_part) {
        // the four lanes (q = 0..3: lane bits 4, 5) that share a row
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
    HC_STAMP(5);
}

// row means of the normalised weights from the kernel's partial sums: fixed order, one thread per (row, modality) -- its np =
// HC_NPART floats are contiguous, neighbouring threads read neighbouring 16-byte vectors
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
