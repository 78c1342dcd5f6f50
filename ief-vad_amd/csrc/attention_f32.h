// Unmasked temporal self-attention over one 256-snippet window, exact fp32 on the matrix cores.
//
// nn.MultiheadAttention(768, 8, batch_first=True)(x, x, x) in eval mode
// (/root/reference/model/imf_vad.py:69-72,115,121): per (chunk, head) softmax(Q K^T / sqrt(96)) V over
// ALL 256 keys -- padding_mask is accepted and ignored by the reference (imf_vad.py:40-44), so zero
// padded rows are attended to, and this kernel does the same.
//
// Q arrives pre-scaled by log2(e)/sqrt(96) from the in_proj epilogue, so the softmax is exp2(s - max) on
// v_exp_f32 (1 ulp; the scale folds into the one fp32 multiply q already gets).  The kernel computes the TRANSPOSED score tile
// S^T = K Q^T, so that in the 32x32 accumulator the lane index is the query and the 16 registers
// are keys: the whole softmax row of a query lives in one lane pair (lanes q and q+32), needs no
// LDS, and the probabilities are already laid out as the A operand of the P V product
// (A[i = query][k = key], lane (i, h) supplying key (r&3) + 8(r>>2) + 4h of accumulator register r).
// K and V tiles are staged in LDS as [64][100 floats] (row padded 96 -> 100 so the ds_read_b128 operand
// fetches are bank-conflict free); K is read as MFMA A operand, V as B operand with ds_read_b32.
#pragma once
#include "common.h"

struct AttnArgs {
    const float* qkv[2];   // [N, 2304] per modality: q | k | v, head h at columns h*96
    float* out[2];         // [N, 768] per modality: concat over heads (fp32), or
    __bf16* outb[2];       // the same as bf16 when non-null (A operand of a bf16 out_proj)
    int nchunks;           // chunks per modality in this launch
    float* amax[2];        // fp16x3 mode (attention_split.h only): running max |out| per modality, nullable
    const float* amax_in[2];   // fp16x3 attention: running max |q k v| of the input tensor (operand scale)
    const RaggedChunk* chunks; // *_rows_kernel: row-compressed chunks (common.h); row r of the window = row min(r, valid)
};

#define ATT_LDK 100
#define ATT_TK 64                                   // keys per staged tile
#define ATT_TILE (ATT_TK * ATT_LDK)                 // floats per LDS tile image (25.6 KB)

// One workgroup = (head, chunk, modality, query half): 4 waves x 32 queries.  K then V stream through a
// double-buffered LDS tile of 64 keys ([64][100 floats], 2 x 25.6 KB): while the waves run the 96 MFMAs of
// tile i, the global loads of tile i+1 are in flight into registers, and they are written to the other LDS
// buffer after the MFMAs (issue-early / write-late staging).  51 KB of LDS and <= 256 VGPRs let two workgroups
// share a CU, so one workgroup's softmax and barriers hide under the other's matrix work.
template <bool RG>
__device__ __forceinline__ void attention_f32_body(const AttnArgs& args, float* kv) {
    // grid (8 heads, 2 query halves, chunks x modalities): the two halves of a (chunk, head) are 8 apart in linear
    // block order, i.e. dispatched back to back onto the SAME XCD (round-robin over 8), so the second half's K / V
    // re-read hits that XCD's L2 instead of HBM
    const int head = blockIdx.x, qhalf = blockIdx.y, chunk = blockIdx.z % args.nchunks, mod = blockIdx.z / args.nchunks;
    int row0 = chunk * IEF_T, last = IEF_T - 1;      // first row of the window in the row set, last distinct row of the window
    if constexpr (RG) {
        const RaggedChunk c = args.chunks[chunk];
        row0 = c.enc_row;
        last = ragged_rows(c.valid) - 1;
        if (qhalf * 128 > last) return;              // every query of this half is a pad row: nobody reads its output
    }
#define ATT_ROW(r) (RG ? ((r) < last ? (r) : last) : (r))
    const float* qkv = args.qkv[mod] + (size_t)row0 * (3 * IEF_D) + head * IEF_DH;
    const size_t obase = (size_t)row0 * IEF_D + head * IEF_DH;
    float* out = args.out[mod] ? args.out[mod] + obase : nullptr;
    __bf16* outb = args.outb[mod] ? args.outb[mod] + obase : nullptr;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int q0 = qhalf * 128 + wave * 32;

    // Q fragment: lane (i, h) holds Q[q0 + i][8s + 4h .. +3], s = 0..11 (B operand of K Q^T)
    f32x4 q[12];
    {
        const float* qp = qkv + (size_t)ATT_ROW(q0 + i) * (3 * IEF_D) + 4 * h;
#pragma unroll
        for (int s = 0; s < 12; ++s) q[s] = *(const f32x4*)(qp + 8 * s);
    }
    // staging map of a 64-row x 24-chunk (4 floats) tile: thread t moves rows (t >> 3) + 32 (j & 1), chunks (t & 7) + 8 (j >> 1), j = 0..5 --
    // eight lanes cover one 128-byte line and every offset is a CONSTANT added to two per-thread bases (attention_split.h's map).  The
    // first version kept a row and a chunk index per piece (c = t + 256 j, twelve registers): with them the kernel needed 263 registers
    // and spilled seven, and a spill reload inside the tile loop is a vector-memory load that waits for the prefetched next tile.
    const int srow0 = t >> 3, sch0 = t & 7;
    const float* gsrc = qkv + (RG ? 0 : (size_t)srow0 * (3 * IEF_D)) + sch0 * 4;      // RG: the row is clamped per load
    float* ldst = kv + srow0 * ATT_LDK + sch0 * 4;
    f32x4 stg[6];
    // tile ti: ti < 4 -> keys 64 ti .. of K (column block IEF_D), else of V (column block 2 IEF_D)
#define ATT_LOAD(ti)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                             \
        stg[j] = *(const f32x4*)(gsrc + (size_t)(RG ? ATT_ROW(((ti) & 3) * ATT_TK + 32 * (j & 1) + srow0)    \
                                                      : ((ti) & 3) * ATT_TK + 32 * (j & 1)) * (3 * IEF_D) +   \
                                 ((ti) < 4 ? IEF_D : 2 * IEF_D) + 32 * (j >> 1));
#define ATT_WRITE(buf)                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                             \
        *(f32x4*)(ldst + (buf) * ATT_TILE + 32 * (j & 1) * ATT_LDK + 32 * (j >> 1)) = stg[j];

    ATT_LOAD(0)
    ATT_WRITE(0)
    __syncthreads();

    f32x16 st[8];
    f32x16 o[3];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

#pragma unroll
    for (int ti = 0; ti < 8; ++ti) {
        if (ti + 1 < 8) {
            ATT_LOAD(ti + 1)
            // keep the loads HERE: left alone, hipcc sinks them behind the tile's 96 MFMAs (to save 24 registers) and the
            // wave then waits out the whole load latency in front of the LDS write, every tile
            __builtin_amdgcn_sched_barrier(0);
        }
        const float* T = kv + (ti & 1) * ATT_TILE;
        if (ti < 4) {
            // S^T[key][query] = sum_d K[key][d] Q[query][d] for the two 32-key sub-tiles of this tile
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float* kp = T + (u * 32 + i) * ATT_LDK + 4 * h;
#pragma unroll
                for (int s = 0; s < 12; ++s) {
                    const f32x4 ka = *(const f32x4*)(kp + 8 * s);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        st[2 * ti + u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], q[s][e], st[2 * ti + u], 0, 0, 0);
                }
            }
            if (ti == 3) {
                // softmax over the 256 keys of query q0 + i: 128 values in this lane, 128 in lane i + 32
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
                        const float p = __builtin_amdgcn_exp2f(st[kt][r] - mx);   // scores arrive in log2 units
                        st[kt][r] = p;
                        sum += p;
                    }
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kt][r] *= inv;
            }
        } else {
            // O[query][d] += sum over this tile's 64 keys of P[query][key] V[key][d]
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float* vp = T + key * ATT_LDK + i;
                    const float pa = st[2 * (ti - 4) + u][r];
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt)
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, vp[dt * 32], o[dt], 0, 0, 0);
                }
        }
        if (ti + 1 < 8) {
            ATT_WRITE((ti + 1) & 1)     // the other buffer: its previous tile (ti - 1) was released by the last barrier
            __syncthreads();
        }
    }
#undef ATT_LOAD
#undef ATT_WRITE
    // store: accumulator col = d (lane & 31), row = query (r&3) + 8(r>>2) + 4h
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qrow = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (RG && qrow > last) continue;
            if (outb) outb[(size_t)qrow * IEF_D + dt * 32 + i] = (__bf16)o[dt][r];
            else out[(size_t)qrow * IEF_D + dt * 32 + i] = o[dt][r];
        }
#undef ATT_ROW
}

__global__ __launch_bounds__(256, 2) void iefvad_attention_f32_kernel(AttnArgs args) {
    __shared__ __attribute__((aligned(16))) float kv[2 * ATT_TILE];
    attention_f32_body<false>(args, kv);
}

// row-compressed chunks of a whole-video pass (ragged.h)
__global__ __launch_bounds__(256, 2) void iefvad_attention_f32_rows_kernel(AttnArgs args) {
    __shared__ __attribute__((aligned(16))) float kv[2 * ATT_TILE];
    attention_f32_body<true>(args, kv);
}
