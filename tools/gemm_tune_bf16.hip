// Stand-alone timing harness for the bf16 projection GEMM kernels (not part of the product library).
// hipcc --offload-arch=gfx950 -O3 -o gemm_tune_bf16 gemm_tune_bf16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <math.h>
#include "../ief-vad_amd/csrc/gemm_bf16.h"
#include "gemm_bf16_wt128.h"      // a rejected tiling kept with the tuner, not in the product

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Variant { const char* name; int kind; int epi; int nz; bool c32, c16; };

static float time_variant(const Variant& v, GemmBArgs g, int iters) {
    g.epi = v.epi;
    g.stagger = getenv("GB2_STAGGER") ? atoi(getenv("GB2_STAGGER")) : 0;
    if (v.epi == EPI_REFINE) g.alpha = 0.5f;
    for (int m = 0; m < 2; ++m) { if (!v.c32) g.p[m].C = nullptr; if (!v.c16) g.p[m].Cb = nullptr; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) {
        if (v.kind == 2) {
            dim3 grid((g.M / GB2_BM) * (g.N / GB2_BN), 1, v.nz);
            hipLaunchKernelGGL(iefvad_gemm_bf16_m32_kernel, grid, dim3(256), GB2_LDS_BYTES, 0, g);
        } else if (v.kind == 7) {
            dim3 grid((g.M / GB3_BM) * (g.N / GB2_BN), 1, v.nz);
            hipLaunchKernelGGL(iefvad_gemm_bf16_w256q_kernel, grid, dim3(512), GB3_LDS_BYTES, 0, g);
        } else if (v.kind == 6) {
            dim3 grid((g.M / GW_BM) * (g.N / GW_BN), 1, v.nz);
            hipLaunchKernelGGL(iefvad_gemm_bf16_wt128_kernel, grid, dim3(256), GW_LDS_BYTES, 0, g);
        } else if (v.kind == 5) {
            dim3 grid((g.M / GB3_BM) * (g.N / GB2_BN), 1, v.nz);
            hipLaunchKernelGGL(iefvad_gemm_bf16_w256_kernel, grid, dim3(512), GB3_LDS_BYTES, 0, g);
        } else if (v.kind == 4) {
            dim3 grid((g.M / GB2_BM) * (g.N / GB2_BN), 1, v.nz);
            hipLaunchKernelGGL(iefvad_gemm_bf16_pipe_kernel, grid, dim3(256), GB2_LDS_BYTES, 0, g);
        } else if (v.kind == 3) {
            dim3 grid((g.M / GB2_BM) * (g.N / GB2_BN), 1, v.nz);
            hipLaunchKernelGGL(iefvad_gemm_bf16_kernel, grid, dim3(256), GB2_LDS_BYTES, 0, g);
        } else {
            dim3 grid((g.M / GEMM_BM) * (g.N / GEMM_BN), 1, v.nz);
            hipLaunchKernelGGL(iefvad_gemm_bf16_v1_kernel, grid, dim3(256), 0, 0, g);
        }
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536, K = 768;
    const int iters = argc > 2 ? atoi(argv[2]) : 100, rounds = 5;
    const int Ns[2] = {768, 2304};
    bf16_t *A, *W, *Cb; float *bias, *C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)2304 * K * 2)); CK(hipMalloc(&bias, 2304 * 4));
    CK(hipMalloc(&C, (size_t)M * 2304 * 4)); CK(hipMalloc(&Cb, (size_t)M * 2304 * 2));
    std::vector<unsigned short> h((size_t)M * K);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)2304 * K * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 2304 * 4)); CK(hipMemset(C, 0, (size_t)M * 2304 * 4));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GB2_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_bf16_m32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GB2_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_bf16_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GB2_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_bf16_w256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GB3_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_bf16_wt128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_bf16_w256q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GB3_LDS_BYTES));
    const Variant vs[] = {{"v1 bias  C32", 1, EPI_BIAS, 1, true, false}, {"m32 bias  C32", 2, EPI_BIAS, 1, true, false},
                          {"m32 bias  C16 only", 2, EPI_BIAS, 1, false, true}, {"m32 relu  C16 only", 2, EPI_BIAS_RELU, 1, false, true},
                          {"m32 refine C32+C16", 2, EPI_REFINE, 1, true, true}, {"m32 resid C32", 2, EPI_BIAS_RESID, 1, true, false},
                          {"m32 bias  C32 z=2", 2, EPI_BIAS, 2, true, false}, {"m32 none (no stores)", 2, EPI_BIAS, 1, false, false},
                          {"m16 bias  C32", 3, EPI_BIAS, 1, true, false}, {"m16 relu  C16 only", 3, EPI_BIAS_RELU, 1, false, true},
                          {"m16 refine C32+C16", 3, EPI_REFINE, 1, true, true}, {"m16 bias  C32 z=2", 3, EPI_BIAS, 2, true, false},
                          {"m16 none (no stores)", 3, EPI_BIAS, 1, false, false},
                          {"pipe bias  C32", 4, EPI_BIAS, 1, true, false}, {"pipe relu  C16 only", 4, EPI_BIAS_RELU, 1, false, true},
                          {"pipe refine C32+C16", 4, EPI_REFINE, 1, true, true}, {"pipe bias  C32 z=2", 4, EPI_BIAS, 2, true, false},
                          {"pipe none (no stores)", 4, EPI_BIAS, 1, false, false},
                          {"w256 bias  C32", 5, EPI_BIAS, 1, true, false}, {"w256 relu  C16 only", 5, EPI_BIAS_RELU, 1, false, true},
                          {"w256 refine C32+C16", 5, EPI_REFINE, 1, true, true}, {"w256 bias  C32 z=2", 5, EPI_BIAS, 2, true, false},
                          {"w256 none (no stores)", 5, EPI_BIAS, 1, false, false},
                          {"wt128 bias  C32", 6, EPI_BIAS, 1, true, false}, {"wt128 relu  C16 only", 6, EPI_BIAS_RELU, 1, false, true},
                          {"wt128 refine C32+C16", 6, EPI_REFINE, 1, true, true}, {"wt128 bias  C32 z=2", 6, EPI_BIAS, 2, true, false},
                          {"wt128 none (no stores)", 6, EPI_BIAS, 1, false, false},
                          {"w256q bias  C32", 7, EPI_BIAS, 1, true, false}, {"w256q relu  C16 only", 7, EPI_BIAS_RELU, 1, false, true},
                          {"w256q refine C32+C16", 7, EPI_REFINE, 1, true, true}, {"w256q bias  C32 z=2", 7, EPI_BIAS, 2, true, false},
                          {"w256q none (no stores)", 7, EPI_BIAS, 1, false, false}};
    const int nv = sizeof(vs) / sizeof(vs[0]);
    for (int ni = 0; ni < 2; ++ni) {
        GemmBArgs g; memset(&g, 0, sizeof(g));
        g.M = M; g.N = Ns[ni]; g.K = K; g.lda = K; g.ldc = Ns[ni];
        g.p[0].A = A; g.p[0].W = W; g.p[0].bias = bias; g.p[0].C = C; g.p[0].Cb = Cb; g.p[0].R = C; g.p[1] = g.p[0];
        {   // v1 vs v2 bit-compare
            std::vector<float> c1((size_t)M * g.N), c2((size_t)M * g.N);
            time_variant(vs[0], g, 1); CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemset(C, 0, c1.size() * 4));
            time_variant(vs[1], g, 1); CK(hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t q = 0; q < c1.size(); ++q) bad += (c1[q] != c2[q]);
            printf("N=%d: m32 vs v1: %zu mismatching elements of %zu\n", g.N, bad, c1.size());
            CK(hipMemset(C, 0, c1.size() * 4));
            time_variant(vs[8], g, 1); CK(hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost));
            double md = 0; for (size_t q = 0; q < c1.size(); ++q) { double d = fabs((double)c1[q] - c2[q]); if (d > md) md = d; }
            printf("N=%d: m16 vs v1: max abs diff %.3g (different fp32 summation order inside a k-tile)\n", g.N, md);
            std::vector<float> c3((size_t)M * g.N);
            CK(hipMemset(C, 0, c1.size() * 4));
            time_variant(vs[13], g, 1); CK(hipMemcpy(c3.data(), C, c3.size() * 4, hipMemcpyDeviceToHost));
            size_t bad3 = 0; for (size_t q = 0; q < c3.size(); ++q) bad3 += (c3[q] != c2[q]);
            printf("N=%d: pipe vs m16: %zu mismatching elements of %zu\n", g.N, bad3, c3.size());
            CK(hipMemset(C, 0, c1.size() * 4));
            time_variant(vs[18], g, 1); CK(hipMemcpy(c3.data(), C, c3.size() * 4, hipMemcpyDeviceToHost));
            bad3 = 0; for (size_t q = 0; q < c3.size(); ++q) bad3 += (c3[q] != c2[q]);
            printf("N=%d: w256 vs m16: %zu mismatching elements of %zu\n", g.N, bad3, c3.size());
            CK(hipMemset(C, 0, c1.size() * 4));
            time_variant(vs[23], g, 1); CK(hipMemcpy(c3.data(), C, c3.size() * 4, hipMemcpyDeviceToHost));
            bad3 = 0; for (size_t q = 0; q < c3.size(); ++q) bad3 += (c3[q] != c2[q]);
            printf("N=%d: wt128 vs m16: %zu mismatching elements of %zu\n", g.N, bad3, c3.size());
            CK(hipMemset(C, 0, c1.size() * 4));
            time_variant(vs[28], g, 1); CK(hipMemcpy(c3.data(), C, c3.size() * 4, hipMemcpyDeviceToHost));
            bad3 = 0; for (size_t q = 0; q < c3.size(); ++q) bad3 += (c3[q] != c2[q]);
            printf("N=%d: w256q vs m16: %zu mismatching elements of %zu\n", g.N, bad3, c3.size());
        }
        std::vector<std::vector<float>> t(nv);
        for (int r = 0; r < rounds; ++r)
            for (int v = 0; v < nv; ++v) t[v].push_back(time_variant(vs[v], g, iters));
        for (int v = 0; v < nv; ++v) {
            std::sort(t[v].begin(), t[v].end());
            const double fl = 2.0 * M * g.N * K * vs[v].nz;
            printf("  %-22s N=%-5d median %.3f ms %7.1f TF   best %7.1f   worst %7.1f\n", vs[v].name, g.N, t[v][rounds / 2],
                   fl / t[v][rounds / 2] * 1e-9, fl / t[v][0] * 1e-9, fl / t[v][rounds - 1] * 1e-9);
        }
    }
#ifdef GB2_CLOCK_DIAG
    for (int which = 0; which < 2; ++which) {   // in-kernel clock and per-k-tile cycle split of the main loop under sustained load
        unsigned long long* dclk; CK(hipMalloc(&dclk, 64 * 8192));
        GemmBArgs g; memset(&g, 0, sizeof(g));
        g.M = M; g.N = 768; g.K = K; g.lda = K; g.ldc = 768; g.epi = EPI_BIAS;
        g.p[0].A = A; g.p[0].W = W; g.p[0].bias = bias; g.p[0].C = nullptr; g.p[0].Cb = Cb; g.p[0].C2 = (float*)dclk; g.p[1] = g.p[0];
        dim3 grid((M / (which ? GB3_BM : GB2_BM)) * (768 / GB2_BN), 1, 1);
        for (int it = 0; it < 4000; ++it) {
            if (which) hipLaunchKernelGGL(iefvad_gemm_bf16_w256_kernel, grid, dim3(512), GB3_LDS_BYTES, 0, g);
            else hipLaunchKernelGGL(iefvad_gemm_bf16_pipe_kernel, grid, dim3(256), GB2_LDS_BYTES, 0, g);
        }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> c(7 * grid.x); CK(hipMemcpy(c.data(), dclk, c.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz; std::vector<unsigned long long> cyc, p0, p1, p2, q0, q1;
        for (size_t b = 0; b < grid.x; ++b) { q0.push_back(c[5 * grid.x + 2 * b]); q1.push_back(c[5 * grid.x + 2 * b + 1]); }
        std::sort(q0.begin(), q0.end()); std::sort(q1.begin(), q1.end());
        for (size_t b = 0; b < grid.x; ++b) {
            ghz.push_back((double)c[2 * b] / (double)c[2 * b + 1] * 0.1); cyc.push_back(c[2 * b]);
            p0.push_back(c[2 * grid.x + 3 * b]); p1.push_back(c[2 * grid.x + 3 * b + 1]); p2.push_back(c[2 * grid.x + 3 * b + 2]);
        }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end()); std::sort(p0.begin(), p0.end()); std::sort(p1.begin(), p1.end()); std::sort(p2.begin(), p2.end());
        printf("%s (bf16 out): clock median %.3f GHz; wave 0 main loop %llu cycles (24 k-tiles, ideal 24 x 512 = 12288); per k-tile over the first 22: body %.0f, vmcnt/lgkmcnt wait %.0f, barrier %.0f\n",
               which ? "w256 256x256 8 waves" : "pipe 128x256 4 waves x 2", ghz[ghz.size() / 2], cyc[cyc.size() / 2], p0[p0.size() / 2] / 22.0, p1[p1.size() / 2] / 22.0, p2[p2.size() / 2] / 22.0);
        printf("    prologue (entry -> loop) %llu cycles, epilogue (loop end -> last store issued) %llu cycles (medians over workgroups)\n", q0[q0.size() / 2], q1[q1.size() / 2]);
    }
#endif
    return 0;
}
