// Unmasked temporal self-attention over one 256-snippet window, bf16 MFMA operands / fp32 softmax
// (throughput mode).  Same semantics as attention_f32.h (/root/reference/model/imf_vad.py:69-72,115,121):
// softmax(Q K^T / sqrt(96)) V over ALL 256 keys of the chunk, per head.
//
// One workgroup = (head, chunk, modality, query half): 4 waves x 32 queries; K then V of the head pass
// through one 52 KB LDS image, so two workgroups fit a CU and one's staging / softmax phases run under the
// other's MFMAs.  q, k, v arrive as bf16 from the in_proj epilogue, q pre-scaled by log2(e)/sqrt(96) so the
// softmax is exp2(s - max) on v_exp_f32.
//   S^T = K Q^T   v_mfma_f32_32x32x16_bf16, A = K rows (ds_read_b128 from a 208-byte-row image: 96 bf16 +
//                 16 B pad -> conflict-free), B = Q rows held in registers.  As in the fp32 kernel the
//                 accumulator then has the query on the lane and keys in the registers: the softmax row of a
//                 query sits in one lane pair, and P is already the A operand of the next product.
//   O = P V       registers 8s..8s+7 of a key tile, converted pairwise to bf16, ARE the A fragment of k-step s;
//                 its k order is permuted (element j of lane half h = key 16s + 8(j>>2) + 4h + (j&3)), so the B
//                 fragment takes V rows in that order: two ds_read_b64_tr_b16 (4 consecutive keys x the
//                 lane's d column each) from a row-major [key][96] image (192-byte rows, conflict-free).
#pragma once
#include "common.h"
#include "gemm_bf16.h"

struct AttnBArgs {
    const bf16_t* qkv[2];   // [N, 2304] bf16 per modality: q | k | v, head h at columns h*96
    bf16_t* out[2];         // [N, 768] bf16 per modality
    int nchunks;            // chunks per modality in this launch
    const RaggedChunk* chunks;   // *_rows_kernel: row-compressed chunks (common.h); row r of the window = row min(r, valid)
    int head_major;         // 0: qkv is [N, 2304] as above; 1: [3 (q, k, v)][8 heads][N][96] -- what inproj_chain_bf16.h writes: a
    int nrows;              //    head's K / V tile is one contiguous 48 KB block (whole 128-byte lines) instead of 192-byte row segments
};

#define ATTB_KROW 104   // K image row length in bf16 elements (208 B = 13 x 16 B)
#define ATTB_VROW 96    // V image row length in bf16 elements (192 B)
#define ATTB_LDS_BYTES (IEF_T * ATTB_KROW * 2)   // 53,248 B, V image (49,152 B) reuses it

template <bool RG>
__device__ __forceinline__ void attention_bf16_body(const AttnBArgs& args, bf16_t* kv) {
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
#define ATTB_ROW(r) (RG ? ((r) < last ? (r) : last) : (r))
    // q / k / v rows of this head: base pointers and the row stride of the layout
    const int rs = args.head_major ? IEF_DH : 3 * IEF_D;
    const size_t plane = args.head_major ? (size_t)IEF_H * args.nrows * IEF_DH : (size_t)IEF_D;
    const bf16_t* qb = args.qkv[mod] + (args.head_major ? ((size_t)head * args.nrows + row0) * IEF_DH : (size_t)row0 * (3 * IEF_D) + head * IEF_DH);
    const bf16_t* kb = qb + plane;
    const bf16_t* vb_ = kb + plane;
    bf16_t* out = args.out[mod] + (size_t)row0 * IEF_D + head * IEF_DH;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int q0 = qhalf * 128 + wave * 32;          // first query row of this wave

    // Q fragment (B operand of K Q^T): lane (i, h) holds Q[q0 + i][16 s + 8 h .. +7], s = 0..5
    bf16x8 q[6];
    {
        const bf16_t* qp = qb + (size_t)ATTB_ROW(q0 + i) * rs + 8 * h;
#pragma unroll
        for (int s = 0; s < 6; ++s) q[s] = *(const bf16x8*)(qp + 16 * s);
    }
    // stage K: 256 rows x 12 chunks of 16 B -> padded rows
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int c = t + 256 * j;
        const int row = c / 12, ch = c - row * 12;
        *(bf16x8*)(kv + row * ATTB_KROW + ch * 8) = *(const bf16x8*)(kb + (size_t)ATTB_ROW(row) * rs + ch * 8);
    }
    __syncthreads();

    // V is fetched NOW, into registers (12 x 16 B per thread), and written to LDS after the softmax: its load latency runs
    // under the 48 MFMAs of K Q^T and the softmax instead of standing between two barriers
    bf16x8 vstage[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int c = t + 256 * j;
        const int row = c / 12, ch = c - row * 12;
        vstage[j] = *(const bf16x8*)(vb_ + (size_t)ATTB_ROW(row) * rs + ch * 8);
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x16 st[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
        const bf16_t* kp = kv + (kt * 32 + i) * ATTB_KROW + 8 * h;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const bf16x8 ka = *(const bf16x8*)(kp + 16 * s);
            st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, q[s], st[kt], 0, 0, 0);
        }
    }

    // softmax (base 2) over the 256 keys of query q0 + i: 128 values here, 128 in lane i + 32
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
            const float p = __builtin_amdgcn_exp2f(st[kt][r] - mx);
            st[kt][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    __syncthreads();   // every wave is done reading K
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int c = t + 256 * j;
        const int row = c / 12, ch = c - row * 12;
        *(bf16x8*)(kv + row * ATTB_VROW + ch * 8) = vstage[j];
    }
    __syncthreads();

    // O[query][d] = sum_key P[query][key] V[key][d]
    f32x16 o[3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    // transposed-read addressing: inside each 16-lane group, lane 4 qq + p supplies row qq, columns 4 p .. 4 p + 3
    const int l16 = lane & 15;
    const int tr_row = l16 >> 2;                       // qq
    const int tr_col = ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
    const bf16_t* vbase = kv + (4 * h + tr_row) * ATTB_VROW + tr_col;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 pa;
#pragma unroll
            for (int j = 0; j < 8; ++j) pa[j] = (bf16_t)(st[kt][8 * s + j] * inv);
            const bf16_t* vp = vbase + (kt * 32 + 16 * s) * ATTB_VROW;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(vp + dt * 32));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(vp + 8 * ATTB_VROW + dt * 32));
                bf16x8 vb;
#pragma unroll
                for (int j = 0; j < 4; ++j) { vb[j] = lo[j]; vb[4 + j] = hi[j]; }
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vb, o[dt], 0, 0, 0);
            }
        }
    }
    // store: accumulator col = d (lane & 31), row = query (r&3) + 8(r>>2) + 4h
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qrow = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (!RG || qrow <= last) out[(size_t)qrow * IEF_D + dt * 32 + i] = (bf16_t)o[dt][r];
        }
#undef ATTB_ROW
}

__global__ __launch_bounds__(256, 2) void iefvad_attention_bf16_kernel(AttnBArgs args) {
    __shared__ __attribute__((aligned(16))) bf16_t kv[IEF_T * ATTB_KROW];
    attention_bf16_body<false>(args, kv);
}

// row-compressed chunks of a whole-video pass (ragged.h)
__global__ __launch_bounds__(256, 2) void iefvad_attention_bf16_rows_kernel(AttnBArgs args) {
    __shared__ __attribute__((aligned(16))) bf16_t kv[IEF_T * ATTB_KROW];
    attention_bf16_body<true>(args, kv);
}
