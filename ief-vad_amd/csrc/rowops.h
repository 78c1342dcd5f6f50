// Row-wise (one wavefront per 768-wide snippet row) stages of the forward: LayerNorm, the
// precision-weighted fusion, the scorer and the input cast.  All are HBM-bound streaming passes:
// 16-byte coalesced loads (lane l owns columns 4l + 256 j, j = 0..2) and 64-lane shuffle reductions.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include "common.h"

#define ROW_WAVES 4   // rows per 256-thread block

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4_t to_bf16x4(f32x4 v) {
    bf16x4_t w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (__bf16)v[e];      // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return w;
}

// ---- LayerNorm(768), eps inside the sqrt, biased variance; optionally a second LayerNorm applied to
// the result (the "whitening" LN that directly follows the last encoder LN, imf_vad.py:116-117,122-123).
struct LnArgs {
    const float* x[2];       // [N, 768] input (already x + attn_out from the out_proj epilogue)
    float* y[2];             // [N, 768] fp32 output (nullable)
    __bf16* yb[2];           // [N, 768] bf16 copy of the output (nullable): A operand of the next bf16 projection
    const float* g1[2];      // first LN weight/bias
    const float* b1[2];
    const float* g2[2];      // second LN (nullable: skip)
    const float* b2[2];
    int nrows;
    float eps;
    float* amax[2];          // fp16x3 mode: running max |y| of the output tensor (nullable)
};

// The per-row arithmetic is written on 4-vectors (hipcc maps them to v_pk_add / v_pk_mul / v_pk_fma_f32: two fp32 lanes per
// instruction): a LayerNorm epilogue of a row-block kernel is VALU-issue bound (~1,300 scalar instructions per wave and 64-row
// block at two waves per SIMD), not memory bound.  ln_row and ln_rows perform the SAME operations per row in the same order, so the
// stand-alone kernel and every fused epilogue agree bit for bit.
__device__ __forceinline__ float ln_hsum(f32x4 t) { return (t[0] + t[1]) + (t[2] + t[3]); }

// centre the row in place and return 1 / sqrt(var + eps): THE statistics of a LayerNorm row -- ln_row and the backward kernel
// (backward.h: iefvad_layernorm_bwd_kernel recomputes instead of storing them) share these operations, so both see the same bits
__device__ __forceinline__ float ln_center_rstd(f32x4 (&v)[3], float eps) {
    const float mean = wave_sum(ln_hsum((v[0] + v[1]) + v[2])) * (1.0f / IEF_D);
    const f32x4 m4 = {mean, mean, mean, mean};
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = v[j] - m4;
    f32x4 sq = v[0] * v[0];
    sq = v[1] * v[1] + sq;
    sq = v[2] * v[2] + sq;
    const float var = wave_sum(ln_hsum(sq)) * (1.0f / IEF_D);
    return 1.0f / sqrtf(var + eps);
}

__device__ __forceinline__ void ln_row(f32x4 (&v)[3], const float* g, const float* b, int lane, float eps) {
    const float rstd = ln_center_rstd(v, eps);
    const f32x4 r4 = {rstd, rstd, rstd, rstd};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const f32x4 gv = *(const f32x4*)(g + 4 * lane + 256 * j);
        const f32x4 bv = *(const f32x4*)(b + 4 * lane + 256 * j);
        v[j] = (v[j] * r4) * gv + bv;
    }
}

// ln_row on R rows of one wave at a time: the same operations per row in the same order, with the R independent reduction
// chains interleaved.  For callers with few resident waves (outproj_ln_*chain_bf16.h: two per SIMD).
template <int R>
__device__ __forceinline__ void ln_rows(f32x4 (&v)[R][3], const float* g, const float* b, int lane, float eps) {
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = ln_hsum((v[r][0] + v[r][1]) + v[r][2]);
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = wave_sum(s[r]);      // R independent DPP chains: the scheduler interleaves them
    float ss[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float mean = s[r] * (1.0f / IEF_D);
        const f32x4 m4 = {mean, mean, mean, mean};
#pragma unroll
        for (int j = 0; j < 3; ++j) v[r][j] = v[r][j] - m4;
        f32x4 sq = v[r][0] * v[r][0];
        sq = v[r][1] * v[r][1] + sq;
        sq = v[r][2] * v[r][2] + sq;
        ss[r] = ln_hsum(sq);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) ss[r] = wave_sum(ss[r]);
    f32x4 gv[3], bv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gv[j] = *(const f32x4*)(g + 4 * lane + 256 * j);
        bv[j] = *(const f32x4*)(b + 4 * lane + 256 * j);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float var = ss[r] * (1.0f / IEF_D);
        const float rstd = 1.0f / sqrtf(var + eps);
        const f32x4 r4 = {rstd, rstd, rstd, rstd};
#pragma unroll
        for (int j = 0; j < 3; ++j) v[r][j] = (v[r][j] * r4) * gv[j] + bv[j];
    }
}

__global__ __launch_bounds__(256) void iefvad_layernorm_kernel(LnArgs a) {
    const int mod = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
    if (row >= a.nrows) return;
    const float* xp = a.x[mod] + (size_t)row * IEF_D + 4 * lane;
    f32x4 v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = *(const f32x4*)(xp + 256 * j);
    ln_row(v, a.g1[mod], a.b1[mod], lane, a.eps);
    if (a.g2[mod] != nullptr) ln_row(v, a.g2[mod], a.b2[mod], lane, a.eps);
    if (a.y[mod]) {
        float* yp = a.y[mod] + (size_t)row * IEF_D + 4 * lane;
#pragma unroll
        for (int j = 0; j < 3; ++j) *(f32x4*)(yp + 256 * j) = v[j];
    }
    if (a.yb[mod]) {
        __bf16* yb = a.yb[mod] + (size_t)row * IEF_D + 4 * lane;
#pragma unroll
        for (int j = 0; j < 3; ++j) *(bf16x4_t*)(yb + 256 * j) = to_bf16x4(v[j]);
    }
    if (a.amax[mod]) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = amax_fold(m, v[j][e]);
        amax_store_part(a.amax[mod], row / IEF_T, row % IEF_T, wave_max(m), lane);
    }
}

// ---- Student-t / Gaussian precision weights + normalised inverse-variance fusion
// (/root/reference/model/imf_vad.py:130-144):
//   w_m = factor * exp(-logvar_m) ; den = w_i + w_e + eps ; n_m = w_m / den ; z = n_i*mu_i + n_e*mu_e
// The literal formula is kept (including its inf/inf = NaN behaviour for logvar < -88.7).
// one element of the fusion; the literal operation order of the reference (shared with heads_fused_bf16.h)
__device__ __forceinline__ void fuse_elem(float mi, float li, float me, float le, float factor, float eps,
                                          float& ni, float& ne, float& z) {
    const float wi = __fmul_rn(factor, expf(-li));
    const float we = __fmul_rn(factor, expf(-le));
    const float den = __fadd_rn(__fadd_rn(wi, we), eps);
    ni = wi / den;
    ne = we / den;
    z = __fadd_rn(__fmul_rn(ni, mi), __fmul_rn(ne, me));
}

struct FusionArgs {
    const float* mu_i; const float* lv_i; const float* mu_e; const float* lv_e;   // [N, 768]
    float* n_i; float* n_e;     // [N, 768], nullable
    float* z;                   // [N, 768]
    __bf16* zb;                 // [N, 768] bf16 copy of z (nullable)
    float* n_i_mean; float* n_e_mean;   // [N], nullable: mean over D (test.py:131-136)
    int nrows;
    float factor, eps;
    float* z_amax;              // fp16x3 mode: running max |z| (nullable)
};

__global__ __launch_bounds__(256) void iefvad_fusion_kernel(FusionArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
    if (row >= a.nrows) return;
    const size_t base = (size_t)row * IEF_D + 4 * lane;
    float si = 0.f, se = 0.f, zm = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const size_t o = base + 256 * j;
        const f32x4 mi = *(const f32x4*)(a.mu_i + o), li = *(const f32x4*)(a.lv_i + o);
        const f32x4 me = *(const f32x4*)(a.mu_e + o), le = *(const f32x4*)(a.lv_e + o);
        f32x4 ni, ne, z;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float nie, nee, ze;
            fuse_elem(mi[e], li[e], me[e], le[e], a.factor, a.eps, nie, nee, ze);
            ni[e] = nie; ne[e] = nee; z[e] = ze;
            si += ni[e];
            se += ne[e];
            zm = amax_fold(zm, z[e]);
        }
        if (a.n_i) *(f32x4*)(a.n_i + o) = ni;
        if (a.n_e) *(f32x4*)(a.n_e + o) = ne;
        *(f32x4*)(a.z + o) = z;
        if (a.zb) *(bf16x4_t*)(a.zb + o) = to_bf16x4(z);
    }
    if (a.z_amax) amax_store_part(a.z_amax, row / IEF_T, row % IEF_T, wave_max(zm), lane);
    if (a.n_i_mean || a.n_e_mean) {
        si = wave_sum(si) * (1.0f / IEF_D);
        se = wave_sum(se) * (1.0f / IEF_D);
        if (lane == 0) {
            if (a.n_i_mean) a.n_i_mean[row] = si;
            if (a.n_e_mean) a.n_e_mean[row] = se;
        }
    }
}

// ---- scorer: logits = z . w_c + b_c  (classifier = Linear(768, 1), imf_vad.py:107,150)
__global__ __launch_bounds__(256) void iefvad_scorer_kernel(const float* z, const float* w, const float* b,
                                                             float* logits, int nrows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const float* zp = z + (size_t)row * IEF_D + 4 * lane;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const f32x4 zv = *(const f32x4*)(zp + 256 * j);
        const f32x4 wv = *(const f32x4*)(w + 4 * lane + 256 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += zv[e] * wv[e];
    }
    s = wave_sum(s);
    if (lane == 0) logits[row] = s + b[0];
}

// ---- input cast: the reference's `.to(torch.float)` (imf_vad.py:41-42) for fp16 / bf16 feature files, and
// the bf16 operand copies of the bf16-projection mode.  Two sources per launch (blockIdx.y), fp32 and/or
// bf16 destinations (nullable).
template <typename T>
__global__ __launch_bounds__(256) void iefvad_cast_kernel(const T* in0, const T* in1, float* out0, float* out1,
                                                          __bf16* ob0, __bf16* ob1, size_t n) {
    const T* in = blockIdx.y ? in1 : in0;
    float* out = blockIdx.y ? out1 : out0;
    __bf16* ob = blockIdx.y ? ob1 : ob0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx * 4 < n; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t o = idx * 4;   // n is a multiple of 4
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)in[o + e];
        if (out) *(f32x4*)(out + o) = v;
        if (ob) *(bf16x4_t*)(ob + o) = to_bf16x4(v);
    }
}

// The same cast with a per-ROW scale folded in (iefvad_forward_scaled: the robustness sweep's attenuated time steps,
// /root/reference/test2.py:71-77 `x[:, idx] = x[:, idx] * 0.01`): the product is formed in fp32 and, for fp16 / bf16 sources, rounded
// to the source type before the widening -- torch multiplies a half tensor in fp32 and stores a half tensor, and the model then
// widens it (imf_vad.py:41-42).  A NULL scale vector, or a scale of exactly 1, leaves the row's bits alone.
template <typename T>
__global__ __launch_bounds__(256) void iefvad_cast_scaled_kernel(const T* in0, const T* in1, float* out0, float* out1, __bf16* ob0, __bf16* ob1,
                                                                 size_t n, const float* s0, const float* s1) {
    const T* in = blockIdx.y ? in1 : in0;
    float* out = blockIdx.y ? out1 : out0;
    __bf16* ob = blockIdx.y ? ob1 : ob0;
    const float* sc = blockIdx.y ? s1 : s0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx * 4 < n; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t o = idx * 4;   // n is a multiple of 4, and so is the row length: the four elements share a row
        const float f = sc ? sc[o / IEF_D] : 1.0f;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (float)in[o + e];
            if (f != 1.0f) {
                // the fp32 product must exist as such before it is narrowed: left to itself hipcc folds fpext -> fmul -> fptrunc into
                // v_fma_mixlo_f16, which rounds the exact product ONCE to fp16 -- torch rounds it to fp32 first (opmath) and then to
                // fp16, and once in a few million elements the two disagree (found by tests/test_gpu_config2.py, round 5)
                float prod = v[e] * f;
                asm volatile("" : "+v"(prod));
                v[e] = (float)(T)prod;
            }
        }
        if (out) *(f32x4*)(out + o) = v;
        if (ob) *(bf16x4_t*)(ob + o) = to_bf16x4(v);
    }
}

// ---- running max |x| of two tensors (blockIdx.y): the raw inputs of the first projection in fp16x3 mode
__global__ __launch_bounds__(256) void iefvad_amax_kernel(const float* in0, const float* in1, float* out0, float* out1, size_t n) {
    const float* in = blockIdx.y ? in1 : in0;
    float* out = blockIdx.y ? out1 : out0;
    float m = 0.f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx * 4 < n; idx += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(in + idx * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m = amax_fold(m, v[e]);
    }
    amax_publish(out, wave_max(m), threadIdx.x & 63);
}

// ---- per-chunk max |x| of the raw inputs (fp16x3 mode): one workgroup per (chunk, modality), no atomics
__global__ __launch_bounds__(256) void iefvad_amax_chunk_kernel(const float* in0, const float* in1, float* out0, float* out1) {
    __shared__ float part[4];
    const float* in = (blockIdx.y ? in1 : in0) + (size_t)blockIdx.x * IEF_T * IEF_D;
    float* out = blockIdx.y ? out1 : out0;
    float m = 0.f;
    for (int i = threadIdx.x * 4; i < IEF_T * IEF_D; i += 256 * 4) {
        const f32x4 v = *(const f32x4*)(in + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) m = amax_fold(m, v[e]);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)blockIdx.x * IEF_AMAX_PARTS] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}
