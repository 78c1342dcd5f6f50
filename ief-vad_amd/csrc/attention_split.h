// Unmasked temporal self-attention over one 256-snippet window, fp32-accurate on the bf16 matrix cores (bf16x6 mode).
//
// Same semantics and the same S^T = K Q^T structure as attention_f32.h (/root/reference/model/imf_vad.py:69-72,115,121):
// per (chunk, head) softmax(Q K^T / sqrt(96)) V over ALL 256 keys, q pre-scaled by log2(e)/sqrt(96) in the in_proj
// epilogue, fp32 q | k | v in, fp32 out.  Both contractions are computed as in gemm_split.h: each fp32 operand is the
// exact sum of three bf16 terms (round-to-nearest of the running remainder) and a product is accumulated in fp32 from
// the six largest bf16 MFMA products (k3 q1 + k1 q3 + k2 q2 + k2 q1 + k1 q2 + k1 q1, and the same for P V); the dropped
// terms are <= 2^-23 of the product in the worst case, ~2^-27 typically (gemm_split.h).  Softmax is fp32 on the accumulators, as in the fp32 kernel.
//
// One workgroup = (head, chunk, modality, query half): 4 waves x 32 queries, two workgroups per CU (78 KB LDS).
// K then V stream through a double-buffered 64-key tile.  The STAGING threads split the tile: global fp32 loads of tile
// i+1 are issued before the MFMAs of tile i, split in registers after them (132 VALU per thread) and written to the
// other LDS buffer as three bf16 plane images, so the MFMA phase reads ready bf16 fragments and no split is repeated
// per wave:
//   K planes [64 keys][104 bf16] (208-byte rows: conflict-free ds_read_b128, as attention_bf16.h)  A operand of K Q^T
//   V planes [64 keys][ 96 bf16] (192-byte rows)  B operand of P V through ds_read_b64_tr_b16
// v_mfma_f32_32x32x16_bf16: lane (i, h) supplies K[key i][d = 16 s + 8 h .. +7] / Q[query i][same d]; in the 32x32
// accumulator the lane is the query and the registers are keys, so registers 8 s2 .. 8 s2 + 7 of a key tile ARE (after
// the three-term split) the A fragments of k-step s2 of P V, in the permuted key order 16 s2 + 8 (j >> 2) + 4 h + (j & 3)
// that the transposed V read reproduces (attention_bf16.h).  Q is split once into registers (72 VGPRs).
// Per 64-key tile and wave: 72 MFMAs of 32 cycles against 96 of 64 cycles in the fp32 kernel.
// F16 = true (compute = fp16x3): two fp16 terms per operand and three products (gemm_split.h, F16): q, k, v are scaled by
// s = 2^(13 - floor(log2 max|qkv|)) (running max from the in_proj epilogue), the scores by s^-2 before the softmax, P by
// 2^13 and the output by 2^-13 / s.  Two planes per tile: 52 KB of LDS.
#pragma once
#include "attention_f32.h"
#include "gemm_split.h"

#define ATS_KROW 104                                   // bf16 elements per K-plane row (208 B)
#define ATS_VROW 96                                    // bf16 elements per V-plane row (192 B)
#define ATS_KPLANE (ATT_TK * ATS_KROW)                 // bf16 elements per plane of a K tile
#define ATS_VPLANE (ATT_TK * ATS_VROW)
#define ATS_BUF (3 * ATS_KPLANE)                       // bf16 elements per tile buffer (the K layout is the larger)
#define ATS_LDS_BYTES (2 * ATS_BUF * 2)                // 79,872 B

// eight fp32 -> three bf16x8 planes (exact three-term split) / two fp16x8 planes of the scaled values
template <bool F16, int NP>
__device__ __forceinline__ void split8(const f32x4& lo, const f32x4& hi, u32x4 (&pl)[NP], float scale) {
    Split4 a, b;
    if constexpr (F16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a.r[e] = lo[e] * scale; b.r[e] = hi[e] * scale; }
    } else {
        a.r = lo;
        b.r = hi;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        unsigned d0, d1, d2, d3;
        if constexpr (F16) { a.plane_f16(d0, d1, p < NP - 1); b.plane_f16(d2, d3, p < NP - 1); }
        else { a.plane(d0, d1, p < NP - 1); b.plane(d2, d3, p < NP - 1); }
        pl[p] = u32x4{d0, d1, d2, d3};
    }
}

#define ATS_MFMA(a, b, c)                                                                                                      \
    if constexpr (F16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); \
    else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
// c += x * y from the planes of x (A operand) and y (B operand), smallest terms first (six bf16 / three fp16 products)
#define ATS_SIX(xp, yp, c)                                                                   \
    if constexpr (!F16) { ATS_MFMA(xp[NP - 1], yp[0], c); ATS_MFMA(xp[0], yp[NP - 1], c); ATS_MFMA(xp[1], yp[1], c); } \
    ATS_MFMA(xp[1], yp[0], c); ATS_MFMA(xp[0], yp[1], c); ATS_MFMA(xp[0], yp[0], c)

__device__ __forceinline__ int wave_id_of(int t) { return t >> 6; }

// TRAIN (round 5, the train-mode forward of train.h): the same kernel, but the probabilities leave the chip -- autograd's backward needs
// P = softmax(S) and Pd = dropout(P) -- and q arrives scaled by 1 / sqrt(96) only (the backward differentiates THAT scale), so the
// scores are multiplied by log2(e) inside the exp2.  ONE [chunks, 8, 256, 256] fp32 tensor carries both: a probability is never
// negative, so its sign bit is free and holds the dropout mask (set = dropped; -0.0 for a dropped zero).  P = |stored| and
// Pd = sign ? 0 : stored / (1 - p) (dropped_from_signed below, the forward's own multiplication) are bit for bit what two tensors
// held -- half the buffer, half the forward's stores and a third less to read in the backward.  The dropout mask is the injected one
// (`keep`, one byte per element) or common.h's counter-based bits, element for element those of the stand-alone softmax kernel.
// Replaces three launches (S = q k^T on the fp32 MFMA kernel, softmax + dropout, Pd v) that wrote and re-read S.
// (dropped_from_signed / with_sign_bit: common.h)
struct AttnTrainArgs {
    float* P[2];                    // per modality: the sign-carrying probabilities
    int drop[2];                    // dropout in force (p > 0 or an injected mask): sign bits are written / interpreted
    const unsigned char* keep[2];   // nullable: injected masks, [chunks, 8, 256, 256] bytes (1 = keep)
    unsigned long long seed[2];
    float drop_p[2];
    // BWD (the first half of the same kernel with other operands: d Pd^T = v d att^T per (chunk, head), then the softmax backward on the
    // accumulators -- a lane holds a whole query row, so rowsum(Pd .* d Pd) is an in-lane sum and one shfl_xor):
    // d S = Pd .* d Pd - P rowsum(Pd .* d Pd) leaves as fp32; replaces the d Pd product and the stand-alone softmax backward.  Both
    // passes read the one sign-carrying tensor (the second finds it in the cache hierarchy)
    const float* dO[2];             // d att: [rows, 768], head h at columns 96 h
    float* dS[2];                   // [chunks, 8, 256, 256]
    // ... and the second half runs on d S where the forward runs on P: d q = q_scale d S k (the k rows staged where the forward stages
    // v), written into the q block of d qkv [rows, 2304]; the stand-alone d q product (which re-read d S) is gone as well
    float* dQ[2];
    float q_scale;
};

// MODE 0: eval; 1: train-mode forward (TRAIN); 2: train-mode backward, d S (BWD); 3: TRAIN with an injected mask (an instantiation of
// its own: with both mask sources in one kernel hipcc spilled 56 registers around the softmax, the hash-only kernel none)
template <bool F16, bool RG = false, int MODE = 0>
__device__ __forceinline__ void attention_split_body(const AttnArgs args, bf16_t* kvs, const AttnTrainArgs* tx = nullptr) {
    constexpr bool TRAIN = MODE == 1 || MODE == 3, BWD = MODE == 2, INJECTED = MODE == 3;
    constexpr int NT = 8;                            // staged 64-key tiles: k then v (BWD: v, then k)
    constexpr int NP = F16 ? 2 : 3;
    // grid (8 heads, 2 query halves, chunks x modalities), see attention_f32.h
    const int head = blockIdx.x, qhalf = blockIdx.y, chunk = blockIdx.z % args.nchunks, mod = blockIdx.z / args.nchunks;
    int row0 = chunk * IEF_T, last = IEF_T - 1;      // first row of the window in the row set, last distinct row of the window
    if constexpr (RG) {
        const RaggedChunk c = args.chunks[chunk];
        row0 = c.enc_row;
        last = ragged_rows(c.valid) - 1;
        if (qhalf * 128 > last) return;              // every query of this half is a pad row: nobody reads its output
    }
#define ATS_ROW(r) (RG ? ((r) < last ? (r) : last) : (r))
    const float* qkv = args.qkv[mod] + (size_t)row0 * (3 * IEF_D) + head * IEF_DH;
    float* out = args.out[mod] + (size_t)row0 * IEF_D + head * IEF_DH;
    float qs = 1.0f, s_inv2 = 1.0f, o_inv = 1.0f;      // fp16x3: operand scale of q / k / v, score and output rescale
    constexpr float kPScale = 8192.0f;                 // P <= 1 -> 2^13
    if constexpr (F16) {
        const int e = 13 - amax_exponent(amax_read_chunk(args.amax_in[mod], chunk));
        qs = __builtin_ldexpf(1.0f, e);
        s_inv2 = __builtin_ldexpf(1.0f, -2 * e);
        o_inv = __builtin_ldexpf(1.0f, -13 - e);
    }

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int q0 = qhalf * 128 + wave * 32;

    // staging map: a 64-row x 24-chunk (4 floats) tile; thread t moves rows srow + 32 (j & 1), chunks (t & 7) + 8 (j >> 1),
    // j = 0..5: eight lanes cover one 128-byte line, and every offset is a constant added to two per-thread bases.
    // WHICH row an 8-lane group takes depends on the plane pitch, so that the four 64-byte row segments a half-wave writes with
    // one ds_write_b64 fall into four disjoint 16-bank ranges: V planes (192-byte rows = 48 banks) take consecutive rows
    // (0, 48, 32, 16); K planes (208-byte rows = 52 banks: consecutive rows start at banks 0, 52, 40, 28 and overlap in 12 banks --
    // 3.8e7 conflict cycles per launch in rounds 2 - 4) take rows R, R + 4, R + 8, R + 12 (52 x 4 = 208 = 16 mod 64: banks 0, 16, 32, 48).
    const int sch0 = t & 7;
    const int srow_v = t >> 3;
    const int srow_k = 16 * (wave_id_of(t) >> 1) + 2 * (wave_id_of(t) & 1) + 4 * ((t >> 3) & 3) + ((t >> 5) & 1);
    const float* gsrc_v = qkv + (RG ? 0 : (size_t)srow_v * (3 * IEF_D)) + sch0 * 4;     // RG: the row is clamped per load
    const float* gsrc_k = qkv + (RG ? 0 : (size_t)srow_k * (3 * IEF_D)) + sch0 * 4;
    f32x4 stg[6];
    // tile ti: ti < 4 -> keys 64 ti .. of K (column block IEF_D), else of V (column block 2 IEF_D)
#define ATS_LOAD(ti)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                             \
        stg[j] = *(const f32x4*)(((ti) < 4 ? gsrc_k : gsrc_v) + (size_t)(RG ? ATS_ROW(((ti) & 3) * ATT_TK + 32 * (j & 1) + ((ti) < 4 ? srow_k : srow_v)) \
                                                      : ((ti) & 3) * ATT_TK + 32 * (j & 1)) * (3 * IEF_D) +       \
                                 (((ti) < 4) != BWD ? IEF_D : 2 * IEF_D) + 32 * (j >> 1));
    // split the staged fp32 chunks and write the three bf16 plane images of tile ti into buffer `buf`
#define ATS_WRITE(ti, buf)                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                           \
        Split4 sp;                                                                                            \
        if constexpr (F16) { _Pragma("unroll") for (int e = 0; e < 4; ++e) sp.r[e] = stg[j][e] * qs; }        \
        else sp.r = stg[j];                                                                                   \
        const int rowlen = (ti) < 4 ? ATS_KROW : ATS_VROW, plane = (ti) < 4 ? ATS_KPLANE : ATS_VPLANE;        \
        bf16_t* dst = kvs + (buf) * ATS_BUF + (((ti) < 4 ? srow_k : srow_v) + 32 * (j & 1)) * rowlen + sch0 * 4 + 32 * (j >> 1); \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {                                                      \
            unsigned d0, d1;                                                                                  \
            if constexpr (F16) sp.plane_f16(d0, d1, p < NP - 1); else sp.plane(d0, d1, p < NP - 1);           \
            *(uint2*)(dst + p * plane) = make_uint2(d0, d1);                                                  \
        }                                                                                                     \
    }

    ATS_LOAD(0)
    // Q planes (B operand of K Q^T): lane (i, h) holds Q[q0 + i][16 s + 8 h .. +7], s = 0..5
    u32x4 qp[6][NP];
    {
        const float* qptr = BWD ? tx->dO[mod] + (size_t)(row0 + q0 + i) * IEF_D + head * IEF_DH + 8 * h
                                : qkv + (size_t)ATS_ROW(q0 + i) * (3 * IEF_D) + 8 * h;
#pragma unroll
        for (int s = 0; s < 6; ++s)
            split8<F16, NP>(*(const f32x4*)(qptr + 16 * s), *(const f32x4*)(qptr + 16 * s + 4), qp[s], qs);
    }
    ATS_WRITE(0, 0)
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

    // transposed-read addressing of the V planes (attention_bf16.h): inside each 16-lane group, lane 4 qq + pp supplies
    // row qq, columns 4 pp .. 4 pp + 3
    const int l16 = lane & 15;
    const int tr_off = (4 * h + (l16 >> 2)) * ATS_VROW + ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;

#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        if (ti + 1 < NT) { ATS_LOAD(ti + 1) }
        const bf16_t* T = kvs + (ti & 1) * ATS_BUF;
        if (ti < 4) {
            // S^T[key][query] = sum_d K[key][d] Q[query][d] for the two 32-key sub-tiles of this tile
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16_t* kp = T + (u * 32 + i) * ATS_KROW + 8 * h;
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    u32x4 ka[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) ka[p] = *(const u32x4*)(kp + p * ATS_KPLANE + 16 * s);
                    ATS_SIX(ka, qp[s], st[2 * ti + u]);
                }
            }
            // TRAIN / BWD: the [256 x 256] tensors (P, Pd, d S) cross the chip boundary ROW-MAJOR.  In the accumulator a lane holds its
            // query's keys in groups of four, so direct 16-byte accesses touch 32 rows per instruction (32-byte pieces: the first build's
            // forward ran at 1.9 TB/s of stores).  After tile 3 the tile buffer that held it is dead until tile 5 is written: every wave
            // takes a [32 queries][64 keys + 4] fp32 slice of it (8.5 KB) and moves 64 keys at a time through it -- the global side is
            // then four whole 256-byte row segments per instruction.  Wave-private: no workgroup barrier beyond the one that frees the
            // buffer.
#define ATS_TL(row_, col_) (tl + (row_) * 68 + (col_))
            // registers (st) -> slice -> memory rows q0 .. q0 + 31 through descriptor `rs_` (based at row q0, key 0; 32 KB): the lane's
            // part of the address is ONE register (vrow), the rest scalar offsets -- 64-bit addresses per (row group, batch) cost 16 spills
#define ATS_PUT_ROWS_B(rs_, b4)                                                                                        \
    {                                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                  \
            _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                              \
                *(f32x4*)ATS_TL(i, 32 * u + 8 * g + 4 * h) = f32x4{st[2 * (b4) + u][4 * g], st[2 * (b4) + u][4 * g + 1], \
                                                                   st[2 * (b4) + u][4 * g + 2], st[2 * (b4) + u][4 * g + 3]}; \
        _Pragma("unroll") for (int sr = 0; sr < 8; ++sr)                                                               \
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *(const f32x4*)ATS_TL(4 * sr + (lane >> 4), (lane & 15) * 4)), \
                                                   rs_, vrow, (4 * sr * IEF_T + 64 * (b4)) * 4, 0);                    \
    }
#define ATS_PUT_ROWS(rs_) _Pragma("unroll") for (int b4 = 0; b4 < 4; ++b4) ATS_PUT_ROWS_B(rs_, b4)
            // memory rows -> slice; afterwards ATS_TL(i, 32 u + 8 g + 4 h) is this lane's group g of sub-tile 2 b4 + u
#define ATS_GET_ROWS(rs_, b4_)                                                                                         \
    _Pragma("unroll") for (int sr = 0; sr < 8; ++sr)                                                                   \
        *(f32x4*)ATS_TL(4 * sr + (lane >> 4), (lane & 15) * 4) =                                                       \
            __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_, vrow, (4 * sr * IEF_T + 64 * (b4_)) * 4, 0));
#define ATS_ROWS_RSRC(ptr_) __builtin_amdgcn_make_buffer_rsrc((void*)((ptr_) + wrow), 0, 32 * IEF_T * 4, 0x00020000)
            if (ti == 3 && BWD) {
                // st = d Pd[query q0 + i][key] (128 keys in this lane, 128 in lane i + 32); register 4 g + e of sub-tile kt is key
                // 32 kt + 8 g + 4 h + e
                __syncthreads();                   // every wave is done with tile 3: its buffer is scratch now
                float* tl = (float*)(kvs + (3 & 1) * ATS_BUF) + wave * (32 * 68);
                const size_t wrow = ((size_t)(chunk * IEF_H + head) * IEF_T + __builtin_amdgcn_readfirstlane(q0)) * IEF_T;
                const int vrow = ((lane >> 4) * IEF_T + (lane & 15) * 4) * 4;
                const auto Pw = ATS_ROWS_RSRC(tx->P[mod]);
                const float dscale = tx->drop[mod] ? (float)(1.0 / (1.0 - (double)tx->drop_p[mod])) : 1.0f;      // the forward's factor
                float d = 0.f;
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) {
                    ATS_GET_ROWS(Pw, b4)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 ps = *(const f32x4*)ATS_TL(i, 32 * u + 8 * g + 4 * h);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                st[2 * b4 + u][4 * g + e] *= dropped_from_signed(ps[e], dscale);
                                d += st[2 * b4 + u][4 * g + e];
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);      // one batch of loads ahead at most: hipcc otherwise hoists all four and spills
                }
                d += __shfl_xor(d, 32, 64);
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) {
                    ATS_GET_ROWS(Pw, b4)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 ps = *(const f32x4*)ATS_TL(i, 32 * u + 8 * g + 4 * h);
#pragma unroll
                            for (int e = 0; e < 4; ++e) st[2 * b4 + u][4 * g + e] -= __builtin_fabsf(ps[e]) * d;      // d S stays in the registers: the A operand of d q = d S k
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const auto Sw = ATS_ROWS_RSRC(tx->dS[mod]);
                ATS_PUT_ROWS(Sw)
            }
            if (ti == 3 && !BWD) {
                // softmax over the 256 keys of query q0 + i: 128 values in this lane, 128 in lane i + 32
                // (fp16x3: the accumulators hold s^2 x the scores; exp2((st - mx) s^-2) in one fma)
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
                        const float p = __builtin_amdgcn_exp2f(F16 ? (st[kt][r] - mx) * s_inv2 : TRAIN ? (st[kt][r] - mx) * 1.4426950408889634f
                                                                                                           : st[kt][r] - mx);   // log2 units
                        st[kt][r] = p;
                        sum += p;
                    }
                sum += __shfl_xor(sum, 32, 64);
                const float inv = (F16 ? kPScale : 1.0f) / sum;      // fp16x3: P is carried as 2^13 P
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kt][r] *= inv;
                if constexpr (TRAIN) {
                    // register 4 g + e of sub-tile kt is key 32 kt + 8 g + 4 h + e of query q0 + i
                    __syncthreads();               // every wave is done with tile 3: its buffer is scratch now
                    float* tl = (float*)(kvs + (3 & 1) * ATS_BUF) + wave * (32 * 68);
                    const size_t wrow = ((size_t)(chunk * IEF_H + head) * IEF_T + __builtin_amdgcn_readfirstlane(q0)) * IEF_T;
                    const int vrow = ((lane >> 4) * IEF_T + (lane & 15) * 4) * 4;
                    const auto Pw = ATS_ROWS_RSRC(tx->P[mod]);
                    const bool drop = tx->drop[mod] != 0;      // uniform
                    const float scale = (float)(1.0 / (1.0 - (double)tx->drop_p[mod]));
                    if (drop) {
                        // the mask goes into the sign bits (dropped = negative): dropped <=> mask byte 0 / 24 random bits < thr <=> the
                        // difference below is negative, and its sign bit IS the flag (no compare, no lane mask to keep alive)
                        const unsigned thr = (unsigned)((double)tx->drop_p[mod] * 16777216.0);
                        const unsigned char* kp = tx->keep[mod];
                        const unsigned long long seed = tx->seed[mod];
                        const size_t prow = wrow + (size_t)i * IEF_T + 4 * h;      // element index of this lane's first key
                        if constexpr (INJECTED) {
#pragma unroll
                            for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const unsigned kb = *(const unsigned*)(kp + prow + 32 * kt + 8 * g);      // four mask bytes (the index is a multiple of 4)
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        st[kt][4 * g + e] = with_sign_bit(st[kt][4 * g + e], (((kb >> (8 * e)) & 0xffu) - 1u) & 0x80000000u);
                                    if (g == 3) __builtin_amdgcn_sched_barrier(0);      // four mask loads ahead at most (all 32: 48 spilled registers)
                                }
                        } else {
#pragma unroll
                            for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
                                for (int g = 0; g < 4; ++g)
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        st[kt][4 * g + e] = with_sign_bit(st[kt][4 * g + e],
                                                                          (dropout_bits(seed, prow + 32 * kt + 8 * g + e) - thr) & 0x80000000u);
                                __builtin_amdgcn_sched_barrier(0);      // sixteen hashes interleaved at most
                            }
                        }
                    }
                    ATS_PUT_ROWS(Pw)           // one copy of the store sequence for both cases
                    if (drop) {                // Pd stays in the registers for P V
#pragma unroll
                        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) st[kt][r] = dropped_from_signed(st[kt][r], scale);
                    }
                }
            }
        } else {
            // O[query][d] += sum over this tile's 64 keys of P[query][key] V[key][d]
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const f32x16& pr = st[2 * (ti - 4) + u];
                    u32x4 pp[NP];
                    split8<F16, NP>(f32x4{pr[8 * s2], pr[8 * s2 + 1], pr[8 * s2 + 2], pr[8 * s2 + 3]},
                                    f32x4{pr[8 * s2 + 4], pr[8 * s2 + 5], pr[8 * s2 + 6], pr[8 * s2 + 7]}, pp, 1.0f);
                    const bf16_t* vp = T + (u * 32 + 16 * s2) * ATS_VROW + tr_off;
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) {
                        u32x4 vb[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                (__attribute__((address_space(3))) bf16x4*)(vp + p * ATS_VPLANE + dt * 32));
                            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                (__attribute__((address_space(3))) bf16x4*)(vp + p * ATS_VPLANE + 8 * ATS_VROW + dt * 32));
                            const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                            vb[p] = u32x4{l2.x, l2.y, h2.x, h2.y};
                        }
                        ATS_SIX(pp, vb, o[dt]);
                    }
                }
        }
        if (ti + 1 < NT) {
            ATS_WRITE(ti + 1, (ti + 1) & 1)     // the other buffer: its previous tile (ti - 1) was released by the last barrier
            __syncthreads();
        }
    }
#undef ATS_LOAD
#undef ATS_WRITE
#undef ATS_TL
#undef ATS_PUT_ROWS
#undef ATS_PUT_ROWS_B
#undef ATS_GET_ROWS
#undef ATS_ROWS_RSRC
    // store: accumulator col = d (lane & 31), row = query (r&3) + 8(r>>2) + 4h
    if constexpr (BWD) {
        float* dq = tx->dQ[mod] + (size_t)row0 * (3 * IEF_D) + head * IEF_DH;
        const float qsc = tx->q_scale;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dq[(size_t)(q0 + (r & 3) + 8 * (r >> 2) + 4 * h) * (3 * IEF_D) + dt * 32 + i] = o[dt][r] * qsc;
        return;
    }
    float om = 0.f;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qrow = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float ov = F16 ? o[dt][r] * o_inv : o[dt][r];
            if (!RG || qrow <= last) out[(size_t)qrow * IEF_D + dt * 32 + i] = ov;
            if constexpr (F16) om = amax_fold(om, ov);
        }
    if constexpr (F16) {      // running max |out| for the out_proj operand scale
        if (args.amax[mod]) amax_store_part(args.amax[mod], chunk, head * 8 + qhalf * 4 + (threadIdx.x >> 6), wave_max(om), lane);
    }
}
#undef ATS_SIX
#undef ATS_MFMA
#undef ATS_ROW

__global__ __launch_bounds__(256, 2) void iefvad_attention_split_kernel(AttnArgs args) {
    extern __shared__ __attribute__((aligned(16))) bf16_t kvs[];
    attention_split_body<false>(args, kvs);
}

// row-compressed chunks of a whole-video pass (ragged.h)
__global__ __launch_bounds__(256, 2) void iefvad_attention_split_rows_kernel(AttnArgs args) {
    extern __shared__ __attribute__((aligned(16))) bf16_t kvs[];
    attention_split_body<false, true>(args, kvs);
}

// train-mode forward: P is stored with the dropout mask in its sign bits (train.h)
__global__ __launch_bounds__(256, 2) void iefvad_attention_split_train_kernel(AttnArgs args, AttnTrainArgs tx) {
    extern __shared__ __attribute__((aligned(16))) bf16_t kvs[];
    attention_split_body<false, false, 1>(args, kvs, &tx);
}

// the same with the caller's mask (tx.keep non-null for both modalities) instead of the generator's
__global__ __launch_bounds__(256, 2) void iefvad_attention_split_train_mask_kernel(AttnArgs args, AttnTrainArgs tx) {
    extern __shared__ __attribute__((aligned(16))) bf16_t kvs[];
    attention_split_body<false, false, 3>(args, kvs, &tx);
}

// train-mode backward: d S from d att, v and the sign-carrying P (train.h)
__global__ __launch_bounds__(256, 2) void iefvad_attention_split_ds_kernel(AttnArgs args, AttnTrainArgs tx) {
    extern __shared__ __attribute__((aligned(16))) bf16_t kvs[];
    attention_split_body<false, false, 2>(args, kvs, &tx);
}

// fp16x3
__global__ __launch_bounds__(256, 2) void iefvad_attention_split_f16_kernel(AttnArgs args) {
    extern __shared__ __attribute__((aligned(16))) bf16_t kvs[];
    attention_split_body<true>(args, kvs);
}
