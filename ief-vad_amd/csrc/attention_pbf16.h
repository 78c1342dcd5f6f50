// bf16 attention, persistent form: ONE 8-wave workgroup per CU walks (modality, chunk, head) items; a head's K and V are staged
// ONCE for both query halves, by LDS-DMA, and the next item's K travels while the current item is computed.
// Same semantics, products and summation order per query as attention_bf16.h (/root/reference/model/imf_vad.py:69-72,115,121):
// the two kernels are bit-identical, which is how this one is tested.
//
// Why: attention_bf16.h runs two 4-wave workgroups per CU, each loading K (wait), computing, writing V through registers.  Its HBM
// traffic is already compulsory (the second half's K / V re-read hits L2), but every workgroup starts with an exposed load and the
// K / V bytes enter the CU twice.  Here (147,456 B of LDS: two K images, one V image):
//   K image       [256 keys][96] bf16, 192-byte rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3: the ds_read_b128 of the K
//                 fragments (lane = key row, 16 lanes per group) is conflict-free without row padding, so the image is lane-linear
//                 for the DMA; the swizzle sits on the per-lane SOURCE address
//   V image       [256 keys][96] bf16 row-major (ds_read_b64_tr_b16, as attention_bf16.h), lane-linear as it is
// Per item: every wave requests 6 KB of V (this item) and 6 KB of K (next item), computes S^T = K Q^T and the softmax of its 32
// queries, requests the next item's Q into the registers S has just released, then waits ONCE (vmcnt(0): everything it waits for was
// requested a whole S + softmax phase earlier), barrier, P V, stores, barrier.  Stores are never waited for.
// Tried: three images rotating through (K, V, free) so that V(n+1) is requested behind the mid-item barrier and something is always
// in flight -- slower (0.74 ms per launch against 0.69): with the score tile (128), the output tile (48) and the next Q (24) live the
// compiler has no register left to run LDS reads ahead of the MFMAs, and the runtime image offsets cost the immediate addressing.
#pragma once
#include "attention_bf16.h"

#define APB_IMG_BYTES (IEF_T * IEF_DH * 2)          // 49,152 B
#define APB_LDS_BYTES (3 * APB_IMG_BYTES)           // K[0] | K[1] | V
#define APB_PIECES 6                                // 1 KB DMA pieces per wave and image (48 per image)

template <bool RG>
__device__ __forceinline__ void attention_pbf16_body(const AttnBArgs& args, char* lds) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int q0 = (wave >> 2) * 128 + (wave & 3) * 32;      // first query row of this wave (waves 0..3: first half, 4..7: second)
    const int rs = args.head_major ? IEF_DH : 3 * IEF_D;      // row stride of a head's q / k / v rows (elements)
    const size_t plane = args.head_major ? (size_t)IEF_H * args.nrows * IEF_DH : (size_t)IEF_D;
    const int total = 2 * args.nchunks * IEF_H, G = gridDim.x;

    // DMA piece p of this wave fills LDS chunks L = (6 wave + p) 64 + lane of an image: row r = L / 12, position L % 12
    // (one packed word per piece: row | K chunk << 8 | V chunk << 12 -- the score tile leaves no registers to spare)
    int pw[APB_PIECES];
#pragma unroll
    for (int p = 0; p < APB_PIECES; ++p) {
        const int L = (APB_PIECES * wave + p) * 64 + lane, r = L / 12, pos = L - 12 * r;
        pw[p] = r | ((pos ^ ((r >> 2) & 3)) << 8) | (pos << 12);      // K: the position holds logical chunk pos ^ swizzle(row)
    }
    // fragment addressing
    const int ksw = (i >> 2) & 3;                   // swizzle of key row 32 kt + i
    // the swizzle touches the two low bits of a chunk index only: chunk (2 s + h) ^ ksw = 4 (s >> 1) + (s odd ? (2 + h) ^ ksw : h ^ ksw) -- TWO
    // per-lane byte offsets plus immediates instead of six registers (with six the row-compressed variant spilled two of them, and their
    // reload -- a vector-memory load behind the K pieces of the NEXT item just requested -- made every item wait for that prefetch)
    const int ko[2] = {(h ^ ksw) << 4, ((2 + h) ^ ksw) << 4};
    const int l16 = lane & 15;
    const int tr_row = l16 >> 2;
    const int tr_col = ((lane >> 4) & 1) * 16 + (l16 & 3) * 4;
    const bf16_t* vbase = (const bf16_t*)(lds + 2 * APB_IMG_BYTES) + (4 * h + tr_row) * ATTB_VROW + tr_col;

    struct Item { const bf16_t* qb; bf16_t* out; int last; };
    auto item_of = [&](int it) {
        // head fastest: the eight 192-byte segments of an output row are written at about the same time and meet in L2 (chunk fastest,
        // i.e. sequential 48 KB tiles per operand stream, measured 11 % slower)
        const int head = it & (IEF_H - 1), cm = it >> 3, chunk = cm % args.nchunks, mod = cm / args.nchunks;
        int row0 = chunk * IEF_T, last = IEF_T - 1;
        if constexpr (RG) {
            const RaggedChunk c = args.chunks[chunk];
            row0 = c.enc_row;
            last = ragged_rows(c.valid) - 1;
        }
        Item r;
        r.qb = args.qkv[mod] + (args.head_major ? ((size_t)head * args.nrows + row0) * IEF_DH : (size_t)row0 * (3 * IEF_D) + head * IEF_DH);
        r.out = args.out[mod] + (size_t)row0 * IEF_D + head * IEF_DH;
        r.last = last;
        return r;
    };
#define APB_ROW(r, last_) (RG ? ((r) < (last_) ? (r) : (last_)) : (r))
    // image <- the 256 rows of one operand of an item (`base` = the head's first row of that operand)
    auto dma = [&](const bf16_t* base, int last_, char* img, int cshift) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (IEF_T - 1) * rs * 2 + IEF_DH * 2, 0x00020000);
#pragma unroll
        for (int p = 0; p < APB_PIECES; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(img + (APB_PIECES * wave + p) * 1024), 16,
                                                     APB_ROW(pw[p] & 255, last_) * rs * 2 + ((pw[p] >> cshift) & 15) * 16, 0, 0, 0);
    };
    auto load_q = [&](bf16x8 (&q)[6], const Item& it_) {
        const bf16_t* qp = it_.qb + (size_t)APB_ROW(q0 + i, it_.last) * rs + 8 * h;
#pragma unroll
        for (int s = 0; s < 6; ++s) q[s] = *(const bf16x8*)(qp + 16 * s);
    };

    int it = blockIdx.x;
    if (it >= total) return;
    Item cur = item_of(it);
    bf16x8 q[6];
    dma(cur.qb + plane, cur.last, lds, 8);                  // K of the first item -> K image 0
    load_q(q, cur);
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0) (the builtin, not inline asm: the compiler's own counting sees it)
    __syncthreads();

    for (int n = 0; it < total; it += G, ++n) {
        const int nxt = it + G;
        const bool more = nxt < total;
        const Item next = item_of(more ? nxt : it);
        const char* kimg = lds + (n & 1) * APB_IMG_BYTES;
        dma(cur.qb + 2 * plane, cur.last, lds + 2 * APB_IMG_BYTES, 12);                        // V of this item
        if (more) dma(next.qb + plane, next.last, lds + ((n + 1) & 1) * APB_IMG_BYTES, 8);   // K of the next item
        const bool active = !RG || (wave >> 2) * 128 <= cur.last;      // RG: a half whose queries are all pad rows computes nothing
        __builtin_amdgcn_sched_barrier(0);

        f32x16 st[8];
        float inv = 0.f;
        if (active) {
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
                const char* kp = kimg + (kt * 32 + i) * (IEF_DH * 2);
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    const bf16x8 ka = *(const bf16x8*)(kp + ko[s & 1] + 64 * (s >> 1));      // chunk (2 s + h) ^ ksw
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, q[s], st[kt], 0, 0, 0);
                }
            }
        }
        if (more) load_q(q, next);                    // the registers S has just released
        if (active) {
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
            inv = 1.0f / sum;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): V, the next K and the next Q, all requested before the softmax
        __syncthreads();                                       // ... by every wave

        if (active) {
            f32x16 o[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
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
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + dt * 32));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vp + 8 * ATTB_VROW + dt * 32));
                        bf16x8 vb;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { vb[j] = lo[j]; vb[4 + j] = hi[j]; }
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vb, o[dt], 0, 0, 0);
                    }
                }
            }
            // store: accumulator col = d (lane & 31), row = query (r & 3) + 8 (r >> 2) + 4 h.  Buffer stores: one per-lane offset
            // register instead of sixteen 64-bit address pairs (the registers the LDS reads need to run ahead of the MFMAs); RG: the
            // descriptor ends behind row `last`, so the rows of pad queries are dropped by the range check
            const auto ro = __builtin_amdgcn_make_buffer_rsrc((void*)cur.out, 0, (RG ? cur.last + 1 : IEF_T) * IEF_D * 2 - (IEF_D - IEF_DH) * 2, 0x00020000);
            const int vo = ((q0 + 4 * h) * IEF_D + i) * 2;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)o[dt][r]), ro,
                                                          vo + ((r & 3) + 8 * (r >> 2)) * (IEF_D * 2), dt * 64, 0);
        }
        __syncthreads();                                       // every wave has read V: the next item's V may land
        cur = next;
    }
#undef APB_ROW
}

__global__ __launch_bounds__(512, 2) void iefvad_attention_pbf16_kernel(AttnBArgs args) {
    extern __shared__ __attribute__((aligned(16))) char apb_lds[];
    attention_pbf16_body<false>(args, apb_lds);
}

// row-compressed chunks of a whole-video pass (ragged.h)
__global__ __launch_bounds__(512, 2) void iefvad_attention_pbf16_rows_kernel(AttnBArgs args) {
    extern __shared__ __attribute__((aligned(16))) char apb_lds[];
    attention_pbf16_body<true>(args, apb_lds);
}
