// Unmasked temporal self-attention over one 256-snippet window, exact fp32 on the matrix cores.
//
// nn.MultiheadAttention(768, 8, batch_first=True)(x, x, x) in eval mode
// (/root/reference/model/imf_vad.py:69-72,115,121): per (chunk, head) softmax(Q K^T / sqrt(96)) V over
// ALL 256 keys -- padding_mask is accepted and ignored by the reference (imf_vad.py:40-44), so zero
// padded rows are attended to, and this kernel does the same.
//
// One workgroup = one (head, chunk, modality); 8 waves x 32 query rows.  Q arrives pre-scaled by
// 1/sqrt(96) from the in_proj epilogue.  The kernel computes the TRANSPOSED score tile
// S^T = K Q^T, so that in the 32x32 accumulator the lane index is the query and the 16 registers
// are keys: the whole softmax row of a query lives in one lane pair (lanes q and q+32), needs no
// LDS, and the probabilities are already laid out as the A operand of the P V product
// (A[i = query][k = key], lane (i, h) supplying key (r&3) + 8(r>>2) + 4h of accumulator register r).
// K and V are staged through one 100 KB LDS image [256][100 floats] (row padded 96 -> 100 so the
// ds_read_b128 operand fetches are bank-conflict free); K is read as MFMA A operand, then the same
// image is refilled with V, read as B operand with ds_read_b32.
#pragma once
#include "common.h"

struct AttnArgs {
    const float* qkv[2];   // [N, 2304] per modality: q | k | v, head h at columns h*96
    float* out[2];         // [N, 768] per modality: concat over heads (fp32), or
    __bf16* outb[2];       // the same as bf16 when non-null (A operand of a bf16 out_proj)
};

#define ATT_LDK 100
#define ATT_LDS_BYTES (IEF_T * ATT_LDK * 4)

__global__ __launch_bounds__(512, 2) void iefvad_attention_f32_kernel(AttnArgs args) {
    extern __shared__ __attribute__((aligned(16))) float kv[];   // [256][100]
    const int head = blockIdx.x, chunk = blockIdx.y, mod = blockIdx.z;
    const float* qkv = args.qkv[mod] + (size_t)chunk * IEF_T * (3 * IEF_D) + head * IEF_DH;
    const size_t obase = (size_t)chunk * IEF_T * IEF_D + head * IEF_DH;
    float* out = args.out[mod] ? args.out[mod] + obase : nullptr;
    __bf16* outb = args.outb[mod] ? args.outb[mod] + obase : nullptr;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 31, h = lane >> 5;

    // Q fragment: lane (i, h) holds Q[wave*32 + i][8s + 4h .. +3], s = 0..11 (B operand of K Q^T)
    f32x4 q[12];
    {
        const float* qp = qkv + (size_t)(wave * 32 + i) * (3 * IEF_D) + 4 * h;
#pragma unroll
        for (int s = 0; s < 12; ++s) q[s] = *(const f32x4*)(qp + 8 * s);
    }
    // stage K: 256 rows x 24 chunks of 16 B
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int c = t + 512 * j;
        const int row = c / 24, ch = c - row * 24;
        *(f32x4*)(kv + row * ATT_LDK + ch * 4) = *(const f32x4*)(qkv + (size_t)row * (3 * IEF_D) + IEF_D + ch * 4);
    }
    __syncthreads();

    // S^T[key][query] = sum_d K[key][d] Q[query][d]
    f32x16 st[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
        const float* kp = kv + (kt * 32 + i) * ATT_LDK + 4 * h;
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const f32x4 ka = *(const f32x4*)(kp + 8 * s);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], q[s][e], st[kt], 0, 0, 0);
        }
    }

    // softmax over the 256 keys of query (wave*32 + i): 128 values in this lane, 128 in lane i+32
    float mx = st[0][0];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(st[kt][r] - mx);
            st[kt][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][r] *= inv;

    __syncthreads();   // every wave is done reading K
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int c = t + 512 * j;
        const int row = c / 24, ch = c - row * 24;
        *(f32x4*)(kv + row * ATT_LDK + ch * 4) = *(const f32x4*)(qkv + (size_t)row * (3 * IEF_D) + 2 * IEF_D + ch * 4);
    }
    __syncthreads();

    // O[query][d] = sum_key P[query][key] V[key][d]
    f32x16 o[3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float* vp = kv + key * ATT_LDK + i;
            const float pa = st[kt][r];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, vp[dt * 32], o[dt], 0, 0, 0);
        }
    }
    // store: accumulator col = d (lane & 31), row = query (r&3) + 8(r>>2) + 4h
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qrow = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (outb) outb[(size_t)qrow * IEF_D + dt * 32 + i] = (__bf16)o[dt][r];
            else out[(size_t)qrow * IEF_D + dt * 32 + i] = o[dt][r];
        }
}
