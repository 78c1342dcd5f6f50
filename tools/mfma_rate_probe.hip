// MFMA issue-rate probe on RANDOM operands with every CU busy: what the matrix pipe sustains, per operand type, once the chip's power
// management has settled -- the question behind "would an int8-sliced fp32 emulation beat the bf16x6 one?" (TRIED.md, round 5).
//   v_mfma_f32_32x32x16_bf16   16,384 MAC in 32 cycles (MI355X_MICROARCH.md)
//   v_mfma_i32_32x32x32_i8     32,768 MAC in 32 cycles
// Each wave runs NITER x 8 back-to-back MFMAs on 8 independent accumulators (two waves per SIMD), operands random per lane and rotated
// every iteration so the multiplier inputs toggle as they do in a GEMM.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(const int* seed, float* sink, int niter) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    i32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[i][e] = seed[(t * 16 + i * 4 + e) & 0xFFFFF];
            b[i][e] = seed[(t * 16 + i * 4 + e + 7777) & 0xFFFFF];
        }
    f32x16 accf[8];
    i32x16 acci[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accf[k][r] = 0.f; acci[k][r] = 0; }
    for (int it = 0; it < niter; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const i32x4 x = a[(k + it) & 3], y = b[(k * 3 + it) & 3];
            if (MODE == 0) accf[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), accf[k], 0, 0, 0);
            else acci[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, acci[k], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += MODE == 0 ? accf[k][r] : (float)acci[k][r];
    if (s == 12345.678f) sink[t] = s;
}

int main(int argc, char** argv) {
    const int zero = argc > 1 && atoi(argv[1]) == 0;
    std::vector<int> h(1 << 20);
    srand(1);
    for (auto& v : h) {
        // bf16 pairs with sane exponents (|x| in [0.25, 4)); as int8 quadruples the same bits are just random bytes
        unsigned lo = 0x3E80 + (rand() % 0x0180) + ((rand() & 1) << 15), hi = 0x3E80 + (rand() % 0x0180) + ((rand() & 1) << 15);
        v = zero ? 0 : (int)(lo | (hi << 16));
    }
    int* d; float* sink;
    hipMalloc(&d, h.size() * 4); hipMalloc(&sink, 1 << 24);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 2, niter = 40000;        // 2 workgroups of 4 waves per CU = 2 waves per SIMD
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, d, sink, niter);
            else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d, sink, niter);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double macs = (double)blocks * 4 * niter * 8 * (mode == 0 ? 16384.0 : 32768.0);
            printf("%s %s operands: %.1f ms, %.2f P-MAC/s (%.2f P-op/s), one 32x32 MFMA per %.1f ns per SIMD\n", mode == 0 ? "bf16 32x32x16" : "i8   32x32x32",
                   zero ? "ZERO" : "random", ms, macs / ms / 1e12, 2 * macs / ms / 1e12, ms * 1e6 / ((double)niter * 8 * 2));
        }
    }
    return 0;
}
