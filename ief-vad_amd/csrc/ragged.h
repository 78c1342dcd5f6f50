// Whole videos in, scores of their snippets out: the reference's loader + test() prologue folded into the library
// (/root/reference/data/tools.py:100-114 process_split, /root/reference/test.py:76-121).
//
// The reference pads every video to whole 256-snippet chunks on the HOST (process_split), scans the padded tensor for NaN and
// copies it to the device (test.py:90-95), runs the model on all chunk rows and slices the padding away again (test.py:121).
// On ShanghaiTech / MSAD-sized lists 84 % of the chunk rows are padding, on UCF / XD-sized lists 35-40 %, and the host's two
// passes over them bound the evaluation loop.  Here the caller hands over ONLY the valid rows ([sum(len), 768], the videos
// concatenated in list order) and their lengths; on the device
//   * iefvad_nanflag_kernel finds, per video and modality, whether any element is NaN (the condition of test.py:90,93);
//   * iefvad_scatter_rows_kernel lays the rows out as zero-padded chunks (the chunker; the all-zero chunk of a
//     len % 256 == 0 video, whose rows test.py:121 slices away, is not built), applies torch.nan_to_num(nan=0.0) -- NaN -> 0,
//     +-inf -> the source dtype's max / min -- to the videos whose flag is set, widens to fp32 (imf_vad.py:41-42) and, in
//     bf16 mode, writes the bf16 operand copy as well;
//   * attention is over the full zero-padded window (imf_vad.py:115: unmasked by design), but the pad rows of a chunk are
//     identical in every layer, so the encoder's row set holds ONE of them per chunk (RaggedChunk, common.h): the projections
//     and LayerNorms run on valid + 1 rows per chunk, the attention kernels read row min(r, valid) for row r of the window
//     (attention_*.h, *_rows_kernel) -- every product and every sum of the dense computation, from fewer distinct rows;
//   * everything behind the encoder (imf_vad.py:125-150: heads, fusion, K refinement steps, scorer -- 56 % of the FLOPs, all
//     row-wise) runs on the same row set; iefvad_rows_out_kernel picks the valid rows' results (test.py:121).
//   fp16x3 (per-chunk operand scales) and IEFVAD_DENSE_ENCODER=1 keep whole chunks in the encoder;
//   iefvad_compact_rows_kernel then gathers the valid rows of the last LayerNorm's output for the tail.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include "common.h"
#include "rowops.h"

template <typename T> struct RaggedLimits;
template <> struct RaggedLimits<float> { static __device__ float max() { return 3.40282347e38f; } };
template <> struct RaggedLimits<__half> { static __device__ float max() { return 65504.f; } };
template <> struct RaggedLimits<__hip_bfloat16> { static __device__ float max() { return 3.38953139e38f; } };

// gridDim.z workgroups per (chunk, modality), each scanning every gridDim.z-th 4096-element tile (a pass of 60 - 130 chunks is
// 120 - 260 workgroups otherwise: a few per CU, each a chain of memory latencies): any NaN among the chunk's valid elements ->
// flags[2 video + modality] = 1
#define IEF_RAGGED_SLICES 4
template <typename T>
__global__ __launch_bounds__(256) void iefvad_nanflag_kernel(const T* img, const T* ev, const RaggedChunk* chunks, int* flags) {
    const RaggedChunk c = chunks[blockIdx.x];
    const T* src = (blockIdx.y ? ev : img) + (size_t)c.src_row * IEF_D;
    const int n = c.valid * IEF_D;                     // a multiple of 768, hence of 4
    bool bad = false;
    for (int base = blockIdx.z * 4096 + threadIdx.x * 4; base < n; base += gridDim.z * 4096) {
        float v[16];                                   // four 16-byte (8-byte) loads in flight per lane
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * u + e] = (base + 1024 * u < n) ? (float)src[base + 1024 * u + e] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) bad |= (v[e] != v[e]);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) flags[2 * c.video + blockIdx.y] = 1;      // benign race: every writer stores 1
}

// gridDim.z workgroups per (chunk, modality), 16-row groups dealt round-robin: valid rows from the packed input (fixed up if the
// video's flag is set), zeros behind them (nrows = 256: whole chunks; nrows = 0: row-compressed, one zero row)
template <typename T>
__global__ __launch_bounds__(256) void iefvad_scatter_rows_kernel(const T* img, const T* ev, const RaggedChunk* chunks, const int* flags,
                                                                  float* out0, float* out1, __bf16* ob0, __bf16* ob1, int nrows) {
    const RaggedChunk c = chunks[blockIdx.x];
    const int m = blockIdx.y;
    const T* src = (m ? ev : img) + (size_t)c.src_row * IEF_D;
    float* out = (m ? out1 : out0) + (size_t)c.enc_row * IEF_D;
    __bf16* ob = m ? ob1 : ob0;
    if (ob) ob += (size_t)c.enc_row * IEF_D;
    const bool fix = flags && flags[2 * c.video + m] != 0;
    const float big = RaggedLimits<T>::max();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nr = nrows ? nrows : ragged_rows(c.valid);
    // four rows of the wave per trip: all their loads are issued before the first store (a 40-row chunk is 2-3 trips of pure latency)
    for (int r0 = wave + 16 * blockIdx.z; r0 < nr; r0 += 16 * gridDim.z) {
        f32x4 v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 4 * u;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int col = 4 * lane + 256 * j;
                v[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < c.valid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][j][e] = (float)src[(size_t)r * IEF_D + col + e];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 4 * u;
            if (r >= nr) break;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int col = 4 * lane + 256 * j;
                f32x4 w = v[u][j];
                if (fix && r < c.valid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = w[e];
                        w[e] = (x != x) ? 0.f : (x > big ? big : (x < -big ? -big : x));     // torch.nan_to_num(nan=0.0)
                    }
                }
                *(f32x4*)(out + (size_t)r * IEF_D + col) = w;
                if (ob) *(bf16x4_t*)(ob + (size_t)r * IEF_D + col) = to_bf16x4(w);
            }
        }
    }
}

// one workgroup per (chunk, modality): the chunk's valid rows of x -> the compact row set (packed order)
struct CompactArgs {
    const float* x[2];      // [chunks * 256, 768] fp32, nullable
    float* xc[2];
    const __bf16* xb[2];    // the same in bf16, nullable
    __bf16* xcb[2];
    const RaggedChunk* chunks;
};
__global__ __launch_bounds__(256) void iefvad_compact_rows_kernel(CompactArgs a) {
    const RaggedChunk c = a.chunks[blockIdx.x];
    const int m = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < c.valid; r += 4) {
        const size_t s = ((size_t)c.enc_row + r) * IEF_D + 4 * lane, d = ((size_t)c.src_row + r) * IEF_D + 4 * lane;
        if (a.x[m]) {
#pragma unroll
            for (int j = 0; j < 3; ++j) *(f32x4*)(a.xc[m] + d + 256 * j) = *(const f32x4*)(a.x[m] + s + 256 * j);
        }
        if (a.xb[m]) {
#pragma unroll
            for (int j = 0; j < 3; ++j) *(bf16x4_t*)(a.xcb[m] + d + 256 * j) = *(const bf16x4_t*)(a.xb[m] + s + 256 * j);
        }
    }
}

// per-row results of a pass -> the caller's packed vectors.  chunks == nullptr: the rows are already in packed order (the tail
// ran on the compact row set): copy n values; else gather the valid rows of every chunk.
__global__ __launch_bounds__(256) void iefvad_rows_out_kernel(const float* s0, const float* s1, const float* s2, float* d0, float* d1,
                                                              float* d2, const RaggedChunk* chunks, int n) {
    if (!chunks) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < n) {
            if (d0) d0[i] = s0[i];
            if (d1) d1[i] = s1[i];
            if (d2) d2[i] = s2[i];
        }
        return;
    }
    const RaggedChunk c = chunks[blockIdx.x];
    const int r = threadIdx.x;
    if (r < c.valid) {
        const size_t s = (size_t)c.enc_row + r, d = (size_t)c.src_row + r;
        if (d0) d0[d] = s0[s];
        if (d1) d1[d] = s1[s];
        if (d2) d2[d] = s2[s];
    }
}
