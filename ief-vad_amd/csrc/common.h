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

// Blocks are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Remap so that each
// XCD walks a contiguous range of logical tile ids; bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
