// What can a CU take in from L2 when every workgroup streams the SAME bytes (the refinement chain's weight stream)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ingest_probe tools/ingest_probe.hip && tools/ingest_probe
// 256 workgroups x 8 waves; wave w of every workgroup streams region w (2.9 MB) of one buffer, 1 KB per instruction:
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds) into a private 8 KB ring, DEPTH pieces in flight per wave
//   mode 1: buffer_load_dwordx4 into registers, DEPTH in flight per wave
//   mode 2: two LDS-DMA pieces and one register piece per step (the mix the chain kernel could use)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void probe(const char* buf, unsigned wave_stride, int pieces, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (size_t)wave * wave_stride), 0, (int)wave_stride, 0x00020000);
    const int ring = wave * 8192, vlane = lane * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#define DMA(p_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + ring + (((p_) & 7) << 10)), 16, vlane, (int)((p_) << 10), 0, 0)
    if constexpr (MODE == 0) {
        for (int s = 0; s < DEPTH; ++s) DMA(s);
        for (int p = 0; p < pieces; ++p) {
            if constexpr (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if constexpr (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            const f32x4 v = *(const f32x4*)(lds + ring + ((p & 7) << 10) + vlane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DMA(p + DEPTH);
            acc += v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 1) {
        f32x4 r[DEPTH];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) r[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, s << 10, 0));
        for (int p = 0; p < pieces; p += DEPTH) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                acc += r[s];
                r[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (p + DEPTH + s) << 10, 0));
            }
        }
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) acc += r[s];
    } else {
        // per step: DMA pieces 3 s, 3 s + 1 (ring of 8 = 4 steps ahead), register piece 3 s + 2 (4 steps ahead)
        f32x4 r[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            DMA(2 * s);
            DMA(2 * s + 1);
            r[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (pieces + s) << 10, 0));
        }
        const int steps = pieces / 3;
        for (int s0 = 0; s0 < steps; s0 += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = s0 + j;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int pd = 2 * s + b;
                    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    const f32x4 v = *(const f32x4*)(lds + ring + ((pd & 7) << 10) + vlane);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    DMA(pd + 8);
                    acc += v;
                }
                acc += r[j];          // hipcc waits for this one itself (vmcnt(0) beside the DMAs: the trap the chain kernel would avoid with asm)
                r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (pieces + s + 4) << 10, 0));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123456.789f) sink[threadIdx.x] = acc[0];
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* buf, unsigned stride, int pieces, float* sink, double bytes_per_wg) {
    CK(hipFuncSetAttribute((const void*)probe<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(256), dim3(512), 65536, 0, buf, stride, pieces, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    printf("%-44s %7.3f ms   %6.1f GB/s per CU   %5.2f TB/s chip-wide\n", name, best, bytes_per_wg / (best * 1e-3) / 1e9, 256 * bytes_per_wg / (best * 1e-3) / 1e12);
}

int main() {
    const int pieces = 2880;                    // 2.9 MB per wave: the chain's 20 projections
    const unsigned stride = (pieces * 2 + 64) * 1024;
    char* buf; float* sink;
    CK(hipMalloc((void**)&buf, (size_t)8 * stride));
    CK(hipMemset(buf, 1, (size_t)8 * stride));
    CK(hipMalloc((void**)&sink, 4096));
    const double bytes = 8.0 * pieces * 1024;
    run<0, 8>("LDS-DMA, 8 KB in flight per wave", buf, stride, pieces, sink, bytes);
    run<0, 4>("LDS-DMA, 4 KB in flight per wave", buf, stride, pieces, sink, bytes);
    run<0, 2>("LDS-DMA, 2 KB in flight per wave", buf, stride, pieces, sink, bytes);
    run<1, 8>("register loads, 8 KB in flight per wave", buf, stride, pieces, sink, bytes);
    run<1, 16>("register loads, 16 KB in flight per wave", buf, stride, pieces, sink, bytes);
    run<1, 4>("register loads, 4 KB in flight per wave", buf, stride, pieces, sink, bytes);
    run<2, 0>("2 LDS-DMA + 1 register piece per step", buf, stride, pieces / 3 * 3, sink, 8.0 * (pieces / 3 * 3) * 1024);
    return 0;
}
