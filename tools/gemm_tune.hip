// Stand-alone tuning harness for the fp32 MFMA projection GEMM (not part of the product library).
// Builds ablation variants of the 128x128x32 kernel to see where the cycles go, and reads the in-kernel
// clock (s_memtime / s_memrealtime) under load.   hipcc --offload-arch=gfx950 -O3 -o gemm_tune gemm_tune.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../ief-vad_amd/csrc/gemm_f32.h"
#include "../ief-vad_amd/csrc/gemm_bf16.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// ABL bit 3: global loads but no ds_write; bit 4: ds_write but no global loads
// ABL bit 0: skip epilogue stores; bit 1: no global loads / ds_writes after the first tile; bit 2: no barrier
template <int ABL>
__global__ __launch_bounds__(256, 2) void gemm_abl(GemmArgs args, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (GEMM_BM + GEMM_BN) * GEMM_BK];
    const GemmProblem& P = args.p[blockIdx.z];
    const int ntn = args.N / GEMM_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
    const int K = args.K, lda = args.lda;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1, i = lane & 31, h = lane >> 5;
    const int srow = t >> 3, sch = t & 7;
    const float* gA = P.A + (size_t)(m0 + srow) * lda + sch * 4;
    const float* gW = P.W + (size_t)(n0 + srow) * K + sch * 4;
    const int sdst = srow * GEMM_BK + ((sch ^ ((srow >> 1) & 7)) << 2);
    const int fsw = (i >> 1) & 7;
    int aoff[2], boff[2];
    for (int x = 0; x < 2; ++x) { aoff[x] = (wr * 64 + x * 32 + i) * GEMM_BK; boff[x] = (wc * 64 + x * 32 + i) * GEMM_BK; }
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 ra[4], rw[4];
    const int nk = K / GEMM_BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) { ra[j] = *(const f32x4*)(gA + (size_t)(32 * j) * lda); rw[j] = *(const f32x4*)(gW + (size_t)(32 * j) * K); }
    {
        float* As = smem; float* Ws = smem + 2 * GEMM_BM * GEMM_BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) { *(f32x4*)(As + sdst + 32 * j * GEMM_BK) = ra[j]; *(f32x4*)(Ws + sdst + 32 * j * GEMM_BK) = rw[j]; }
    }
    __syncthreads();
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk) && !(ABL & 2);
        const bool do_load = more && !(ABL & 16), do_write = more && !(ABL & 8);
        if (do_load) {
            const int k1 = (kt + 1) * GEMM_BK;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ra[j] = *(const f32x4*)(gA + (size_t)(32 * j) * lda + k1); rw[j] = *(const f32x4*)(gW + (size_t)(32 * j) * K + k1); }
        }
        const float* As = smem + cur * GEMM_BM * GEMM_BK;
        const float* Ws = smem + 2 * GEMM_BM * GEMM_BK + cur * GEMM_BN * GEMM_BK;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = ((2 * s + h) ^ fsw) << 2;
            f32x4 fa[2], fb[2];
            fa[0] = *(const f32x4*)(As + aoff[0] + ch); fa[1] = *(const f32x4*)(As + aoff[1] + ch);
            fb[0] = *(const f32x4*)(Ws + boff[0] + ch); fb[1] = *(const f32x4*)(Ws + boff[1] + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][e], fb[0][e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][e], fb[1][e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][e], fb[0][e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][e], fb[1][e], acc[1][1], 0, 0, 0);
            }
        }
        if (more && !do_write) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { asm volatile("" ::"v"(ra[j]), "v"(rw[j])); }
        }
        if (do_write) {
            float* Ad = smem + (cur ^ 1) * GEMM_BM * GEMM_BK; float* Wd = smem + 2 * GEMM_BM * GEMM_BK + (cur ^ 1) * GEMM_BN * GEMM_BK;
#pragma unroll
            for (int j = 0; j < 4; ++j) { *(f32x4*)(Ad + sdst + 32 * j * GEMM_BK) = ra[j]; *(f32x4*)(Wd + sdst + 32 * j * GEMM_BK) = rw[j]; }
        }
        if (!(ABL & 4)) __syncthreads();
        if (!(ABL & 2)) cur ^= 1;
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (clk && t == 0) { clk[2 * (blockIdx.x + gridDim.x * blockIdx.z)] = c1 - c0; clk[2 * (blockIdx.x + gridDim.x * blockIdx.z) + 1] = r1 - r0; }
    if (ABL & 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[a][b][r]));
        return;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wc * 64 + b * 32 + i;
        const float bv = P.bias[n];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int mb = m0 + wr * 64 + a * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) P.C[(size_t)(mb + (r & 3) + 8 * (r >> 2)) * args.ldc + n] = acc[a][b][r] + bv;
        }
    }
}

template <int ABL>
static void run(const char* name, GemmArgs g, int nz, unsigned long long* dclk, int iters) {
    dim3 grid((g.M / GEMM_BM) * (g.N / GEMM_BN), 1, nz);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(gemm_abl<ABL>, grid, dim3(256), 0, 0, g, dclk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(gemm_abl<ABL>, grid, dim3(256), 0, 0, g, dclk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    std::vector<unsigned long long> clk(2 * grid.x * nz);
    CK(hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; for (size_t b = 0; b < clk.size() / 2; ++b) { cyc += clk[2 * b]; rt += clk[2 * b + 1]; }
    const double flops = 2.0 * g.M * g.N * g.K * nz;
    printf("%-28s M=%d N=%d K=%d z=%d  %.3f ms  %.1f TF  loop cycles/block %.0f  in-kernel clock %.2f GHz\n", name, g.M, g.N, g.K, nz, ms,
           flops / ms * 1e-9, cyc / (clk.size() / 2), cyc / rt * 0.1);
}

static double g_last_clock = 0, g_last_cycles = 0;
struct Variant { const char* name; int kind; int epi; int nz; };   // kind: -1 library kernel, else ablation mask

static float time_variant(const Variant& v, GemmArgs g, unsigned long long* dclk, int iters) {
    g.epi = v.epi;
    if (v.epi == EPI_REFINE) g.alpha = 0.5f;
    dim3 grid((g.M / GEMM_BM) * (g.N / GEMM_BN), 1, v.nz);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) {
        switch (v.kind) {
            case -1: hipLaunchKernelGGL(iefvad_gemm_f32_kernel, grid, dim3(256), 0, 0, g); break;
            case -3: {
                GemmBArgs b; memset(&b, 0, sizeof(b));
                b.M = g.M; b.N = g.N; b.K = g.K; b.lda = g.lda; b.ldc = g.ldc; b.epi = g.epi; b.alpha = g.alpha; b.qcols = g.qcols;
                for (int m = 0; m < 2; ++m) { b.p[m].A = (const bf16_t*)g.p[m].A; b.p[m].W = (const bf16_t*)g.p[m].W; b.p[m].bias = g.p[m].bias;
                                              b.p[m].C = g.p[m].C; b.p[m].R = g.p[m].R; b.p[m].C2 = g.p[m].C2; }
                dim3 grid2((g.M / GB2_BM) * (g.N / GB2_BN), 1, v.nz);
                hipLaunchKernelGGL(iefvad_gemm_f32_t256_kernel, grid2, dim3(256), GB2_LDS_BYTES, 0, b);
                break;
            }
            case 0: hipLaunchKernelGGL(gemm_abl<0>, grid, dim3(256), 0, 0, g, dclk); break;
            case 1: hipLaunchKernelGGL(gemm_abl<1>, grid, dim3(256), 0, 0, g, dclk); break;
            case 2: hipLaunchKernelGGL(gemm_abl<2>, grid, dim3(256), 0, 0, g, dclk); break;
            case 3: hipLaunchKernelGGL(gemm_abl<3>, grid, dim3(256), 0, 0, g, dclk); break;
            case 8: hipLaunchKernelGGL(gemm_abl<8>, grid, dim3(256), 0, 0, g, dclk); break;
            case 16: hipLaunchKernelGGL(gemm_abl<16>, grid, dim3(256), 0, 0, g, dclk); break;
            default: hipLaunchKernelGGL(gemm_abl<7>, grid, dim3(256), 0, 0, g, dclk); break;
        }
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    if (v.kind >= 0) {
        std::vector<unsigned long long> clk(2 * (size_t)grid.x);
        CK(hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0; for (size_t b = 0; b < clk.size() / 2; ++b) { cyc += clk[2 * b]; rt += clk[2 * b + 1]; }
        g_last_clock = cyc / rt * 0.1; g_last_cycles = cyc / (clk.size() / 2);
    } else { g_last_clock = 0; g_last_cycles = 0; }
    return ms / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768, K = 768;
    const int iters = argc > 2 ? atoi(argv[2]) : 100, rounds = 5;
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_f32_t256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GB2_LDS_BYTES));
    const int Ns[2] = {768, 2304};
    float *A, *W, *bias, *C; unsigned long long* dclk;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)2304 * K * 4)); CK(hipMalloc(&bias, 2304 * 4));
    CK(hipMalloc(&C, (size_t)M * 2304 * 4)); CK(hipMalloc(&dclk, 2 * 8 * (size_t)(M / 128) * 18 * 2));
    std::vector<float> h((size_t)M * K);
    srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)2304 * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), 2304 * 4, hipMemcpyHostToDevice));
    const Variant vs[] = {{"v1 full", 0, EPI_BIAS, 1}, {"LIB bias", -1, EPI_BIAS, 1}, {"T256 bias", -3, EPI_BIAS, 1}, {"T256 refine R=C", -3, EPI_REFINE, 1}, {"T256 relu", -3, EPI_BIAS_RELU, 1}, {"T256 resid R=C", -3, EPI_BIAS_RESID, 1}, {"T256 bias z=2", -3, EPI_BIAS, 2}, {"LIB relu", -1, EPI_BIAS_RELU, 1},
                          {"LIB refine R=C", -1, EPI_REFINE, 1}, {"LIB resid R=C", -1, EPI_BIAS_RESID, 1},
                          {"LIB bias z=2", -1, EPI_BIAS, 2}, {"v1 no stores", 1, EPI_BIAS, 1}, {"v1 loads, no ds_write", 8, EPI_BIAS, 1}, {"v1 ds_write, no loads", 16, EPI_BIAS, 1}, {"v1 no loads", 2, EPI_BIAS, 1},
                          {"v1 no loads/stores", 3, EPI_BIAS, 1}, {"v1 mfma only", 7, EPI_BIAS, 1}};
    const int nv = sizeof(vs) / sizeof(vs[0]);
    for (int ni = 0; ni < 2; ++ni) {
        GemmArgs g; memset(&g, 0, sizeof(g));
        g.M = M; g.N = Ns[ni]; g.K = K; g.lda = K; g.ldc = Ns[ni];
        g.p[0].A = A; g.p[0].W = W; g.p[0].bias = bias; g.p[0].C = C; g.p[0].R = C; g.p[1] = g.p[0];
        // correctness: library kernel vs the v1 ablation-0 kernel, bit for bit
        {
            std::vector<float> c1((size_t)M * g.N), c2((size_t)M * g.N);
            time_variant(vs[0], g, dclk, 1); CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemset(C, 0, c1.size() * 4));
            time_variant(vs[1], g, dclk, 1); CK(hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t q = 0; q < c1.size(); ++q) bad += (c1[q] != c2[q]);
            printf("N=%d: library kernel vs v1: %zu mismatching elements of %zu\n", g.N, bad, c1.size());
            CK(hipMemset(C, 0, c1.size() * 4));
            time_variant(vs[2], g, dclk, 1); CK(hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost));
            bad = 0; for (size_t q = 0; q < c1.size(); ++q) bad += (c1[q] != c2[q]);
            printf("N=%d: t256 kernel vs v1: %zu mismatching elements of %zu\n", g.N, bad, c1.size());
        }
        std::vector<std::vector<float>> t(nv);
        std::vector<double> clkv(nv), cycv(nv);
        for (int r = 0; r < rounds; ++r)
            for (int v = 0; v < nv; ++v) { t[v].push_back(time_variant(vs[v], g, dclk, iters)); clkv[v] = g_last_clock; cycv[v] = g_last_cycles; }
        for (int v = 0; v < nv; ++v) {
            std::sort(t[v].begin(), t[v].end());
            const double fl = 2.0 * M * g.N * K * vs[v].nz;
            printf("  %-22s N=%-5d median %.3f ms %6.1f TF   best %6.1f   worst %6.1f   clock %.2f GHz  loop cyc/block %.0f\n", vs[v].name, g.N, t[v][rounds / 2],
                   fl / t[v][rounds / 2] * 1e-9, fl / t[v][0] * 1e-9, fl / t[v][rounds - 1] * 1e-9, clkv[v], cycv[v]);
        }
    }
    return 0;
}
