// Shared device/host helpers for libiefvad (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define IEF_D 768      // embed dim
#define IEF_T 256      // snippets per chunk
#define IEF_H 8        // heads
#define IEF_DH 96      // head dim

// One 256-snippet chunk of a whole-video pass (ragged.h).  The encoder's row set holds the chunk at rows enc_row ..: either all
// 256 rows (valid rows, then zero rows), or -- row-compressed -- the valid rows followed by ONE zero row standing for all
// 256 - valid pad rows of the chunk: pad rows are identical in every layer (same input, row-wise projections, and attention
// gives equal queries equal outputs), so every kernel that needs "row r of the chunk" reads row min(r, valid).
struct RaggedChunk {
    int src_row;   // first packed row of the chunk, relative to the pass's first packed row
    int valid;     // 1..256 valid rows
    int video;     // index of the video in the call (selects the NaN flag)
    int enc_row;   // first row of the chunk in the encoder's row set
};
// rows the chunk occupies in a row-compressed set
__host__ __device__ __forceinline__ int ragged_rows(int valid) { return valid < IEF_T ? valid + 1 : IEF_T; }

// Blocks are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Remap so that each
// XCD walks a contiguous range of logical tile ids; bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Sum / maximum over the 64 lanes, result in every lane (wave-uniform).  DPP cross-lane operands instead of six ds_bpermute round
// trips through the LDS crossbar (~100 cycles each, and a row-wise epilogue runs four such chains per row: outproj_ln_*chain,
// LayerNorm backward ...): quad_perm (xor 1, xor 2), row_half_mirror, row_mirror -- after each step the lanes of the group hold
// the same value, so which lane pairs with which does not matter --, then row_bcast:15 / row_bcast:31 fold the four 16-lane rows
// and lane 63 holds the total.  The summation TREE is the xor butterfly's taken in ascending order (1, 2, 4, 8, 16, 32): every
// kernel reduces through these two functions, so all of them stay bit-identical to each other.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_take<0xB1, 0xF>(v, v);          // quad_perm [1,0,3,2]
    v += dpp_take<0x4E, 0xF>(v, v);          // quad_perm [2,3,0,1]
    v += dpp_take<0x141, 0xF>(v, v);         // row_half_mirror: the other quad of the 8-lane group
    v += dpp_take<0x140, 0xF>(v, v);         // row_mirror: the other half of the 16-lane row
    v += dpp_take<0x142, 0xA>(0.f, v);       // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xC>(0.f, v);       // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_take<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_take<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_take<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_take<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_take<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_take<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Running maximum of |x| over a tensor (fp16x3 mode: the power-of-two operand scale of the next projection), kept in
// IEF_AMAX_WAYS device words: a producer workgroup updates word (blockIdx.x % WAYS) -- one hot word would serialise the
// 10^5 waves of a row-wise kernel -- and the consumer takes the maximum of all of them.  Non-negative floats order like
// their bit patterns, so the update is an unsigned atomicMax; the word is read first and the atomic skipped when it would
// not grow (after the first few workgroups it rarely does).  NaNs are ignored by fmaxf upstream.
#define IEF_AMAX_WAYS 32
#define IEF_AMAX_STRIDE 32          // floats between the words of one tensor: one 128-byte line each
#define IEF_AMAX_FLOATS (IEF_AMAX_WAYS * IEF_AMAX_STRIDE)
// fold |v| into a running maximum, ignoring non-finite values: an inf / NaN element (the reference lets them propagate,
// test.py:90-95 only replaces NaN) must poison its own chunk, not the power-of-two operand scale of the whole tensor
__device__ __forceinline__ float amax_fold(float m, float v) {
    const float a = fabsf(v);
    return fmaxf(m, a < __builtin_inff() ? a : 0.f);
}
__device__ __forceinline__ void amax_publish(float* slot, float wave_amax, int lane) {
    if (lane == 0) {
        float* word = slot + (blockIdx.x % IEF_AMAX_WAYS) * IEF_AMAX_STRIDE;
        const unsigned b = __float_as_uint(wave_amax);
        if (b > *(volatile unsigned*)word) atomicMax((unsigned*)word, b);
    }
}
__device__ __forceinline__ float amax_read(const float* slot) {
    const int lane = threadIdx.x & 63;
    return wave_max(lane < IEF_AMAX_WAYS ? slot[lane * IEF_AMAX_STRIDE] : 0.f);
}

// Activations carry their running maximum PER CHUNK (256 rows): chunks are independent batch rows of the forward (attention
// never crosses them) and every 128-row GEMM tile lies inside one, so a per-chunk operand scale costs nothing and a huge
// outlier in one video cannot take precision away from another (round 2; the per-tensor scale of round 1 did).  No atomics:
// a (tensor, chunk) owns IEF_AMAX_PARTS words, every producing WAVE stores the maximum of what it wrote into a word of its
// own (row-wise kernels: part = row % 256; GEMM epilogues: part = wave tile index inside the chunk; attention: part =
// (head, query half, wave)), the words nobody writes stay zero from the micro-batch's memset, and a consumer wave reads all
// 256 words with one 16-byte load per lane.  (One word per chunk updated with atomicMax was tried first: the workgroups
// resident at one time all belong to the same few chunks, i.e. to ONE cache line -- LayerNorm 9 -> 59 ms per step.)
// Weights keep one scale per matrix (the multi-way words above).
#define IEF_AMAX_PARTS 256
__device__ __forceinline__ void amax_store_part(float* slot, int chunk, int part, float wave_amax, int lane) {
    if (lane == 0) slot[(size_t)chunk * IEF_AMAX_PARTS + part] = wave_amax;
}
__device__ __forceinline__ float amax_read_chunk(const float* slot, int chunk) {       // call with the whole wave
    const int lane = threadIdx.x & 63;
    const f32x4 v = *(const f32x4*)(slot + (size_t)chunk * IEF_AMAX_PARTS + 4 * lane);
    return wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
}

// floor(log2(amax)) for a finite positive amax; 13 (scale 2^0 below) for zero / non-finite.  Capped at 30: beyond |x| ~ 2^30
// the scaled value overflows fp16 and the element's chunk becomes NaN (the reference's fp32 chain overflows to NaN within
// that chunk from ~1e19 on); with per-chunk scales no other chunk is affected either way.
#define IEF_AMAX_EXP_CAP 30
__device__ __forceinline__ int amax_exponent(float amax) {
    const unsigned b = __float_as_uint(amax);
    const int e = (int)((b >> 23) & 0xff);
    if (e == 0 || e == 0xff) return 13;
    return e - 127 < IEF_AMAX_EXP_CAP ? e - 127 : IEF_AMAX_EXP_CAP;
}

// The train forward's sign-carrying attention probabilities (attention_split.h TRAIN): stored = P with the sign bit set where the
// element was dropped.  Both helpers take the value BY VALUE on purpose: __builtin_bit_cast applied directly to an element of an
// ext_vector_type (st[kt][r]) read element 0 of the vector for every r (hipcc 7.2: the OR below then mixed st[kt][0] into all sixteen
// elements of a sub-tile -- found by comparing the stored tensor with softmax(q k^T)).
__device__ __forceinline__ float dropped_from_signed(float ps, float scale) {      // dropout(P) = sign ? 0 : P / (1 - p)
    return __builtin_bit_cast(int, ps) < 0 ? 0.f : ps * scale;
}
__device__ __forceinline__ float with_sign_bit(float x, unsigned signbit) {         // signbit: 0 or 0x80000000
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) | signbit);
}

// Counter-based uniform bits for the attention-dropout mask (train mode): a 32-bit mix (murmur3's finaliser, twice) of (seed, element
// index); the same (seed, index) always gives the same bits, so nothing but the seed has to be remembered, and the fused attention
// kernel (attention_split.h, TRAIN) and the stand-alone softmax kernel (backward.h) draw the same mask.  torch draws its mask from
// Philox in an order of its own: a p > 0 run is statistically, not bit-wise, the reference's ("parity unpinned"; p = 0 and injected
// masks are pinned).  Round 5 replaced the 64-bit splitmix finaliser (twelve quarter-rate 32-bit multiplies per element: more VALU
// time than the attention kernel's MFMAs) by this form: four.  `dropout_mix` is the per-row part.
__device__ __forceinline__ unsigned fmix32(unsigned x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    return x ^ (x >> 16);
}
__device__ __forceinline__ unsigned dropout_bits(unsigned long long seed, unsigned long long idx) {
    const unsigned lo = (unsigned)idx, hi = (unsigned)(idx >> 32);
    const unsigned a = fmix32(lo ^ (unsigned)seed);
    return fmix32(a + (unsigned)(seed >> 32) + hi * 0x9E3779B1u) >> 8;      // 24 uniform bits
}
