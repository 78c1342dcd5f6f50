// Timing experiments on the 32 x 32 fp32 GEMM (per-video pattern): hipcc -DGEMT_EXP=<mask> ... ; not part of the library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../ief-vad_amd/csrc/gemm_f32.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 768, K = 768, nW = 20;
    float *A, *W, *bias, *C;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)nW * N * K * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    std::vector<float> h((size_t)nW * N * K);
    srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4));
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.epi = EPI_BIAS;
    g.p[0].A = A; g.p[0].bias = bias; g.p[0].C = C;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round) {
        for (int kind = 0; kind < 2; ++kind) {
            const int iters = 400;
            CK(hipEventRecord(e0));
            for (int it = 0; it < iters; ++it) {
                g.p[0].W = W + (size_t)(it % nW) * N * K;          // a different weight matrix per launch, as in the forward
                if (kind == 0) hipLaunchKernelGGL(iefvad_gemm_f32_tiny_kernel, dim3((M / GEMT_BM) * (N / GEMT_BN)), dim3(256), 0, 0, g);
                else hipLaunchKernelGGL(iefvad_gemm_f32_small_kernel, dim3((M / GEMS_BM) * (N / GEMS_BN)), dim3(256), 0, 0, g);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("EXP=%d %s M=%d N=%d: %.2f us per launch (back to back)\n", GEMT_EXP, kind ? "64x64" : "32x32", M, N, ms / iters * 1e3);
        }
    }
    return 0;
}
