// Stand-alone accuracy + timing harness for the split-bf16 (fp32-accurate) projection GEMM (not part of the library).
// hipcc --offload-arch=gfx950 -O3 -o gemm_tune_split gemm_tune_split.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <math.h>
#define GS_EXPERIMENT_F16
#include "../ief-vad_amd/csrc/gemm_split.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static unsigned short f2bf(float f) {   // round to nearest even (finite inputs)
    unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

struct Variant { const char* name; int kind; int epi; bool c32; };

static GemmBArgs gh;   // fp16 two-term experiment: same A, W as two fp16 planes
static float run(const Variant& v, GemmBArgs gs, GemmBArgs gf, int iters) {
    gs.epi = gf.epi = gh.epi = v.epi; gs.alpha = gf.alpha = gh.alpha = 0.5f;
    gh.p[0].C = gs.p[0].C;
    if (!v.c32) { gs.p[0].C = gf.p[0].C = gh.p[0].C = nullptr; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) {
        dim3 grid((gs.M / GS_BM) * (gs.N / GS_BN), 1, 1);
        dim3 grid2((gs.M / GS_BM) * (gs.N / 128), 1, 1);
        if (v.kind == 0) hipLaunchKernelGGL(iefvad_gemm_split_kernel, grid, dim3(256), GS_LDS_BYTES, 0, gs);
        else if (v.kind == 2) hipLaunchKernelGGL(iefvad_gemm_split_n128_kernel, grid2, dim3(256), GS_LDS_BYTES_OF(2), 0, gs);
        else if (v.kind == 3) hipLaunchKernelGGL(iefvad_gemm_split_f16_kernel, grid, dim3(256), GS_LDS_BYTES, 0, gh);
        else if (v.kind == 4) hipLaunchKernelGGL(iefvad_gemm_split_f16_n128_kernel, grid2, dim3(256), GS_LDS_BYTES_OF(2), 0, gh);
        else hipLaunchKernelGGL(iefvad_gemm_f32_t256_kernel, grid, dim3(256), GB2_LDS_BYTES, 0, gf);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536, K = 768;
    const int iters = argc > 2 ? atoi(argv[2]) : 50, rounds = 5;
    const int Ns[2] = {768, 2304};
    const int NW = 2304;
    float *A, *Wf, *bias, *C, *R; bf16_t* Wp;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&Wf, (size_t)NW * K * 4)); CK(hipMalloc(&Wp, (size_t)3 * NW * K * 2));
    CK(hipMalloc(&bias, NW * 4)); CK(hipMalloc(&C, (size_t)M * NW * 4)); CK(hipMalloc(&R, (size_t)M * NW * 4));
    std::vector<float> hA((size_t)M * K), hW((size_t)NW * K), hb(NW);
    srand(1);
    for (auto& v : hA) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 1.7f;
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.036f;    // ~ U(-1/sqrt(768), 1/sqrt(768))
    for (auto& v : hb) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.036f;
    if (getenv("GS_ZERO")) {    // DVFS probe: all-zero operands toggle no multiplier bits (MI355X_MICROARCH.md, DVFS give-back)
        for (auto& v : hA) v = 0.f;
        for (auto& v : hW) v = 0.f;
    }
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wf, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), NW * 4, hipMemcpyHostToDevice));
    CK(hipMemset(R, 0, (size_t)M * NW * 4));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_split_n128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS_BYTES_OF(2)));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_split_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_split_f16_n128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS_BYTES_OF(2)));
    CK(hipFuncSetAttribute((const void*)iefvad_gemm_f32_t256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GB2_LDS_BYTES));
    for (int ni = 0; ni < 2; ++ni) {
        const int N = Ns[ni];
        {   // planes of the first N rows, plane stride N*K
            std::vector<unsigned short> hp((size_t)3 * N * K);
            for (size_t q = 0; q < (size_t)N * K; ++q) {
                const float x = hW[q];
                const unsigned short b1 = f2bf(x); const float r1 = x - bf2f(b1);
                const unsigned short b2 = f2bf(r1); const float r2 = r1 - bf2f(b2);
                hp[q] = b1; hp[(size_t)N * K + q] = b2; hp[(size_t)2 * N * K + q] = f2bf(r2);
            }
            CK(hipMemcpy(Wp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
        }
        GemmBArgs gs; memset(&gs, 0, sizeof(gs));
        gs.M = M; gs.N = N; gs.K = K; gs.lda = K; gs.ldc = N; gs.wplane = N * K * 2;
        gs.p[0].A = (const bf16_t*)A; gs.p[0].W = Wp; gs.p[0].bias = bias; gs.p[0].C = C; gs.p[0].R = R; gs.p[1] = gs.p[0];
        GemmBArgs gf = gs; gf.p[0].W = (const bf16_t*)Wf; gf.p[1] = gf.p[0];
        {   // two fp16 planes of W (round to nearest even), plane stride N*K
            std::vector<_Float16> hh((size_t)2 * N * K);
            for (size_t q = 0; q < (size_t)N * K; ++q) {
                const float x = hW[q];
                const _Float16 h1 = (_Float16)x;
                hh[q] = h1; hh[(size_t)N * K + q] = (_Float16)(x - (float)h1);
            }
            static _Float16* Wh = nullptr;
            if (!Wh) CK(hipMalloc(&Wh, (size_t)2 * NW * K * 2));
            CK(hipMemcpy(Wh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice));
            gh = gs; gh.p[0].W = (const bf16_t*)Wh; gh.p[1] = gh.p[0];
        }
        // accuracy against a double-precision dot product on sampled outputs
        std::vector<float> cs((size_t)M * N), cf((size_t)M * N);
        const Variant vs0 = {"split", 0, EPI_BIAS, true}, vf0 = {"f32", 1, EPI_BIAS, true};
        CK(hipMemset(C, 0, cs.size() * 4)); run(vs0, gs, gf, 1); CK(hipMemcpy(cs.data(), C, cs.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(C, 0, cs.size() * 4)); run(vf0, gs, gf, 1); CK(hipMemcpy(cf.data(), C, cf.size() * 4, hipMemcpyDeviceToHost));
        double es = 0, ef = 0, ss = 0, sf = 0; size_t cnt = 0;
        for (int m = 0; m < M; m += 37)
            for (int n = 0; n < N; n += 5) {
                double d = hb[n];
                for (int k = 0; k < K; ++k) d += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
                const double a = fabs(cs[(size_t)m * N + n] - d), b = fabs(cf[(size_t)m * N + n] - d);
                es = std::max(es, a); ef = std::max(ef, b); ss += a * a; sf += b * b; ++cnt;
            }
        double md = 0; for (size_t q = 0; q < cs.size(); ++q) md = std::max(md, (double)fabs(cs[q] - cf[q]));
        {   // the 128 x 128 configuration must give the same bits as the 128 x 256 one (same k order per output)
            const Variant vn0 = {"split n128", 2, EPI_BIAS, true};
            std::vector<float> cn((size_t)M * N);
            CK(hipMemset(C, 0, cs.size() * 4)); run(vn0, gs, gf, 1); CK(hipMemcpy(cn.data(), C, cn.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t q = 0; q < cs.size(); ++q) bad += (cn[q] != cs[q]);
            printf("N=%d: split n128 vs split: %zu mismatching elements of %zu\n", N, bad, cs.size());
        }
        printf("N=%d accuracy vs fp64 on %zu sampled outputs: split max %.3g rms %.3g | fp32 MFMA max %.3g rms %.3g | split vs fp32 MFMA max %.3g\n",
               N, cnt, es, sqrt(ss / cnt), ef, sqrt(sf / cnt), md);
        {   // fp16 two-term experiment: accuracy of the same sampled outputs
            const Variant vh = {"f16x3 n128", 4, EPI_BIAS, true};
            std::vector<float> ch((size_t)M * N);
            CK(hipMemset(C, 0, ch.size() * 4)); run(vh, gs, gf, 1); CK(hipMemcpy(ch.data(), C, ch.size() * 4, hipMemcpyDeviceToHost));
            double eh = 0, sh = 0; size_t c2 = 0;
            for (int m = 0; m < M; m += 37)
                for (int n = 0; n < N; n += 5) {
                    double d = hb[n];
                    for (int k = 0; k < K; ++k) d += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
                    const double a = fabs(ch[(size_t)m * N + n] - d);
                    eh = std::max(eh, a); sh += a * a; ++c2;
                }
            printf("N=%d fp16 two-term / three-product experiment: max %.3g rms %.3g\n", N, eh, sqrt(sh / c2));
        }
        const Variant vs[] = {{"split bias C32", 0, EPI_BIAS, true}, {"split refine C32", 0, EPI_REFINE, true}, {"split none", 0, EPI_BIAS, false},
                              {"n128  bias C32", 2, EPI_BIAS, true}, {"n128  refine C32", 2, EPI_REFINE, true}, {"n128  none", 2, EPI_BIAS, false},
                              {"f16x3 256 bias C32", 3, EPI_BIAS, true}, {"f16x3 256 refine", 3, EPI_REFINE, true}, {"f16x3 256 none", 3, EPI_BIAS, false},
                              {"f16x3 128 bias C32", 4, EPI_BIAS, true}, {"f16x3 128 refine", 4, EPI_REFINE, true}, {"f16x3 128 none", 4, EPI_BIAS, false},
                              {"f32   bias C32", 1, EPI_BIAS, true}, {"f32   refine C32", 1, EPI_REFINE, true}, {"f32   none", 1, EPI_BIAS, false}};
        const int nv = sizeof(vs) / sizeof(vs[0]);
        std::vector<std::vector<float>> t(nv);
        for (int r = 0; r < rounds; ++r)
            for (int v = 0; v < nv; ++v) t[v].push_back(run(vs[v], gs, gf, iters));
        for (int v = 0; v < nv; ++v) {
            std::sort(t[v].begin(), t[v].end());
            const double fl = 2.0 * M * N * K;
            printf("  %-18s N=%-5d median %.3f ms %7.1f TF-equivalent   best %7.1f   worst %7.1f\n", vs[v].name, N, t[v][rounds / 2],
                   fl / t[v][rounds / 2] * 1e-9, fl / t[v][0] * 1e-9, fl / t[v][rounds - 1] * 1e-9);
        }
    }
#ifdef GB2_CLOCK_DIAG
    {   // in-kernel clock and cycles of the split main loop under sustained load
        unsigned long long* dclk; CK(hipMalloc(&dclk, 64 * 8192));
        GemmBArgs g; memset(&g, 0, sizeof(g));
        g.M = M; g.N = 768; g.K = K; g.lda = K; g.ldc = 768; g.epi = EPI_BIAS; g.wplane = 768 * K * 2;
        g.p[0].A = (const bf16_t*)A; g.p[0].W = Wp; g.p[0].bias = bias; g.p[0].C = C; g.p[0].C2 = (float*)dclk; g.p[1] = g.p[0];
      for (int cfgi = 0; cfgi < 3; ++cfgi) {
        dim3 grid((M / GS_BM) * (768 / (cfgi ? 128 : GS_BN)), 1, 1);
        printf("%s\n", cfgi == 2 ? "fp16x3, 128 x 128, two workgroups per CU:" : cfgi ? "128 x 128, two workgroups per CU:" : "128 x 256, one workgroup per CU:");
        GemmBArgs g2 = gh; g2.epi = EPI_BIAS; g2.p[0].C = C; g2.p[0].C2 = (float*)dclk; g2.p[1] = g2.p[0];
        for (int it = 0; it < 4000; ++it) {
            if (cfgi == 2) hipLaunchKernelGGL(iefvad_gemm_split_f16_n128_kernel, grid, dim3(256), GS_LDS_BYTES_OF(2), 0, g2);
            else if (cfgi) hipLaunchKernelGGL(iefvad_gemm_split_n128_kernel, grid, dim3(256), GS_LDS_BYTES_OF(2), 0, g);
            else hipLaunchKernelGGL(iefvad_gemm_split_kernel, grid, dim3(256), GS_LDS_BYTES, 0, g);
        }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> c(7 * grid.x); CK(hipMemcpy(c.data(), dclk, c.size() * 8, hipMemcpyDeviceToHost));
        {
            std::vector<unsigned long long> p0, p1, p2;
            for (size_t b = 0; b < grid.x; ++b) { p0.push_back(c[2 * grid.x + 3 * b]); p1.push_back(c[2 * grid.x + 3 * b + 1]); p2.push_back(c[2 * grid.x + 3 * b + 2]); }
            std::sort(p0.begin(), p0.end()); std::sort(p1.begin(), p1.end()); std::sort(p2.begin(), p2.end());
            printf("wave 0, per k-tile (median over blocks): MFMA body %.0f cycles, vmcnt/lgkmcnt wait %.0f, barrier wait %.0f (ideal body 3072)\n",
                   p0[p0.size() / 2] / 24.0, p1[p1.size() / 2] / 24.0, p2[p2.size() / 2] / 24.0);
        }
        {
            std::vector<unsigned long long> q0, q1;
            for (size_t b = 0; b < grid.x; ++b) { q0.push_back(c[5 * grid.x + 2 * b]); q1.push_back(c[5 * grid.x + 2 * b + 1]); }
            std::sort(q0.begin(), q0.end()); std::sort(q1.begin(), q1.end());
            printf("wave 0: prologue %llu cycles, epilogue %llu cycles (median over blocks)\n", q0[q0.size() / 2], q1[q1.size() / 2]);
        }
        std::vector<double> ghz; std::vector<unsigned long long> cyc;
        for (size_t b = 0; b < grid.x; ++b) { ghz.push_back((double)c[2 * b] / (double)c[2 * b + 1] * 0.1); cyc.push_back(c[2 * b]); }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        printf("split main loop in-kernel clock: median %.3f GHz (min %.3f max %.3f); loop cycles median %llu (ideal 24 x 3072 = 73728 / 24 x 1536 = 36864 per wave)\n",
               ghz[ghz.size() / 2], ghz.front(), ghz.back(), cyc[cyc.size() / 2]);
      }
    }
#endif
    return 0;
}
