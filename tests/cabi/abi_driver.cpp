// Torch-free consumer of libiefvad.so: includes only include/iefvad.h and the HIP runtime API.
//   abi_driver <blob.bin> <B> <L> <K> <out_logits.bin> [compute]
// blob.bin: float32 stream written by tests/test_gpu_cabi_driver.py: every state_dict tensor in the order of
// iefvad_amd.synth.state_dict_keys(L, K), 3 floats of padding, then img [B,256,768], then ev [B,256,768].
// Writes B*256 fp32 logits.  Exit code 0 on success; any ABI error prints iefvad_last_error().
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "iefvad.h"

#define HIPCK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
#define ABICK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, iefvad_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s blob B L K out [compute]\n", argv[0]); return 1; }
    const int B = atoi(argv[2]), L = atoi(argv[3]), K = atoi(argv[4]);
    const int compute = argc > 6 ? atoi(argv[6]) : IEFVAD_COMPUTE_F32;
    const size_t D = 768, T = 256, DD = D * D;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("blob"); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t nfloat = (size_t)ftell(f) / 4;
    fseek(f, 0, SEEK_SET);
    std::vector<float> host(nfloat);
    if (fread(host.data(), 4, nfloat, f) != nfloat) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    const size_t nweights = 2 * (size_t)L * (3 * DD + 3 * D + DD + D + 2 * D) + 4 * D + 4 * (DD + D) + (size_t)K * 2 * (DD + D) + D + 1 + 3;
    if (nfloat != nweights + 2 * (size_t)B * T * D) { fprintf(stderr, "blob has %zu floats, expected %zu\n", nfloat, nweights + 2 * (size_t)B * T * D); return 1; }
    float* dev = nullptr;
    HIPCK(hipMalloc((void**)&dev, nfloat * 4));
    HIPCK(hipMemcpy(dev, host.data(), nfloat * 4, hipMemcpyHostToDevice));

    // carve the blob exactly in state_dict registration order
    iefvad_weights w;
    memset(&w, 0, sizeof(w));
    const float* p = dev;
    auto take = [&](size_t n) { const float* q = p; p += n; return q; };
    for (int m = 0; m < 2; ++m) {
        for (int l = 0; l < L; ++l) {
            w.in_proj_w[m][l] = take(3 * DD); w.in_proj_b[m][l] = take(3 * D);
            w.out_proj_w[m][l] = take(DD); w.out_proj_b[m][l] = take(D);
        }
        for (int l = 0; l < L; ++l) { w.norm_w[m][l] = take(D); w.norm_b[m][l] = take(D); }
    }
    for (int m = 0; m < 2; ++m) { w.whiten_w[m] = take(D); w.whiten_b[m] = take(D); }
    w.mu_w[0] = take(DD); w.mu_b[0] = take(D);           // image_mu
    w.mu_w[1] = take(DD); w.mu_b[1] = take(D);           // event_mu
    w.logvar_w[0] = take(DD); w.logvar_b[0] = take(D);   // image_logvar
    w.logvar_w[1] = take(DD); w.logvar_b[1] = take(D);   // event_logvar
    for (int k = 0; k < K; ++k) {
        w.ref_w1[k] = take(DD); w.ref_b1[k] = take(D);
        w.ref_w2[k] = take(DD); w.ref_b2[k] = take(D);
    }
    w.cls_w = take(D); w.cls_b = take(1);
    (void)take(3);   // pad: the feature blocks must be 16-byte aligned (the forward checks it)
    const float* img = take((size_t)B * T * D);
    const float* ev = take((size_t)B * T * D);

    iefvad_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = IEFVAD_ABI_VERSION; cfg.embed_dim = 768; cfg.seq_len = 256; cfg.num_heads = 8;
    cfg.num_layers = L; cfg.num_steps = K; cfg.noise_model = IEFVAD_NOISE_STUDENT_T; cfg.compute = compute;
    cfg.lambda_ref = 0.5f; cfg.nu = 8.0f; cfg.epsilon = 1e-8f; cfg.micro_batch = 0;
    iefvad_handle* h = nullptr;
    ABICK(iefvad_create(&cfg, &h));
    hipStream_t stream;
    HIPCK(hipStreamCreate(&stream));
    ABICK(iefvad_set_weights(h, &w, stream));
    const size_t wsb = iefvad_workspace_bytes(h, B);
    void* ws = nullptr;
    HIPCK(hipMalloc(&ws, wsb));
    float* logits = nullptr;
    HIPCK(hipMalloc((void**)&logits, (size_t)B * T * 4));
    iefvad_outputs out;
    memset(&out, 0, sizeof(out));
    out.logits = logits;
    ABICK(iefvad_forward(h, img, ev, IEFVAD_IN_F32, B, ws, wsb, &out, stream));
    iefvad_stage_times st;
    ABICK(iefvad_forward_timed(h, img, ev, IEFVAD_IN_F32, B, ws, wsb, &out, stream, &st));
    HIPCK(hipStreamSynchronize(stream));
    std::vector<float> hl((size_t)B * T);
    HIPCK(hipMemcpy(hl.data(), logits, hl.size() * 4, hipMemcpyDeviceToHost));
    FILE* o = fopen(argv[5], "wb");
    if (!o || fwrite(hl.data(), 4, hl.size(), o) != hl.size()) { fprintf(stderr, "cannot write output\n"); return 1; }
    fclose(o);
    // error path: a too-small workspace must be refused with a message, not crash
    if (iefvad_forward(h, img, ev, IEFVAD_IN_F32, B, ws, 16, &out, stream) == 0 || strstr(iefvad_last_error(), "workspace") == nullptr) {
        fprintf(stderr, "small workspace was not refused\n");
        return 4;
    }
    // whole videos (iefvad_forward_videos): the B chunk blocks re-read as B videos of T rows each plus, from the same rows, one
    // short video -- every full chunk must score exactly as in the dense forward (same chunk content, row-independent tail;
    // bit for bit in the F32 and BF16 modes, which is what this driver is run in for B = 2 and B = 64)
    {
        std::vector<int32_t> lens((size_t)B, T);
        lens.back() = 100;                               // the last chunk's first 100 rows as a short video
        const size_t rows = (size_t)(B - 1) * T + 100;
        float *vl = nullptr, *vwi = nullptr, *vwe = nullptr;
        HIPCK(hipMalloc((void**)&vl, rows * 4)); HIPCK(hipMalloc((void**)&vwi, rows * 4)); HIPCK(hipMalloc((void**)&vwe, rows * 4));
        const size_t vws = iefvad_videos_workspace_bytes(h, lens.data(), B);
        if (vws == 0 || vws > wsb) { fprintf(stderr, "videos workspace %zu vs %zu\n", vws, wsb); return 6; }
        ABICK(iefvad_forward_videos(h, img, ev, IEFVAD_IN_F32, lens.data(), B, 1, ws, wsb, vl, vwi, vwe, stream));
        HIPCK(hipStreamSynchronize(stream));
        std::vector<float> hv(rows);
        HIPCK(hipMemcpy(hv.data(), vl, rows * 4, hipMemcpyDeviceToHost));
        if ((compute == IEFVAD_COMPUTE_F32 || compute == IEFVAD_COMPUTE_BF16) && B > 1 &&
            memcmp(hv.data(), hl.data(), (size_t)(B - 1) * T * 4) != 0) { fprintf(stderr, "forward_videos differs from the dense forward\n"); return 6; }
        for (size_t i = 0; i < rows; ++i)
            if (!(hv[i] == hv[i])) { fprintf(stderr, "forward_videos produced a NaN at row %zu\n", i); return 6; }
        lens[0] = 0;
        if (iefvad_forward_videos(h, img, ev, IEFVAD_IN_F32, lens.data(), B, 1, ws, wsb, vl, vwi, vwe, stream) == 0) { fprintf(stderr, "a zero-length video was accepted\n"); return 6; }
        (void)hipFree(vl); (void)hipFree(vwi); (void)hipFree(vwe);
    }
    // the score gather with no torch in the process: librccl is bound from the system ROCm by dlopen; a
    // one-rank communicator exercises id -> init -> count -> all-gather (equal counts) -> the unequal-count bookkeeping
    {
        unsigned char ident[IEFVAD_COMM_ID_BYTES];
        ABICK(iefvad_comm_unique_id(ident));
        iefvad_comm* comm = nullptr;
        ABICK(iefvad_comm_create(ident, 1, 0, &comm));
        if (iefvad_comm_nranks(comm) != 1) { fprintf(stderr, "comm_nranks != 1\n"); return 5; }
        float* gathered = nullptr;
        HIPCK(hipMalloc((void**)&gathered, (size_t)B * T * 4));
        HIPCK(hipMemsetAsync(gathered, 0xff, (size_t)B * T * 4, stream));
        ABICK(iefvad_gather_scores(comm, logits, (size_t)B * T, nullptr, gathered, (size_t)B * T, stream));
        HIPCK(hipStreamSynchronize(stream));
        std::vector<float> hg((size_t)B * T);
        HIPCK(hipMemcpy(hg.data(), gathered, hg.size() * 4, hipMemcpyDeviceToHost));
        if (memcmp(hg.data(), hl.data(), hg.size() * 4) != 0) { fprintf(stderr, "gathered scores differ from the local ones\n"); return 5; }
        const int64_t counts[1] = {(int64_t)T};          // only the first chunk's scores
        HIPCK(hipMemsetAsync(gathered, 0xff, (size_t)B * T * 4, stream));
        ABICK(iefvad_gather_scores(comm, logits, 0, counts, gathered, (size_t)B * T, stream));
        HIPCK(hipStreamSynchronize(stream));
        HIPCK(hipMemcpy(hg.data(), gathered, T * 4, hipMemcpyDeviceToHost));
        if (memcmp(hg.data(), hl.data(), T * 4) != 0) { fprintf(stderr, "counted gather differs\n"); return 5; }
        if (iefvad_gather_scores(nullptr, logits, 1, nullptr, gathered, 1, stream) == 0) { fprintf(stderr, "null comm accepted\n"); return 5; }
        if (iefvad_gather_scores(comm, logits, (size_t)B * T, nullptr, gathered, (size_t)B * T - 1, stream) == 0) { fprintf(stderr, "short `gathered` accepted\n"); return 5; }
        if (iefvad_gather_scores(comm, logits, 8, nullptr, logits + 4, 8, stream) == 0) { fprintf(stderr, "overlapping buffers accepted\n"); return 5; }
        iefvad_comm_destroy(comm);
        (void)hipFree(gathered);
    }
    // the metric tail for a consumer without sklearn: AUC / AP of the logits against a deterministic frame-level
    // ground truth (frame j positive iff the top three bits of j * 2654435761 mod 2^32 are zero), 16 frames per snippet
    double h_metric[2] = {0.0, 0.0};
    {
        const size_t n = (size_t)B * T;
        std::vector<uint8_t> hgt(n * 16);
        for (size_t j = 0; j < hgt.size(); ++j) hgt[j] = (((uint32_t)j * 2654435761u) >> 29) == 0;
        uint8_t* dgt = nullptr;
        double* dm = nullptr;
        void* mws = nullptr;
        const size_t mwsb = iefvad_auc_ap_workspace_bytes((int64_t)n);
        if (mwsb == 0) { fprintf(stderr, "iefvad_auc_ap_workspace_bytes returned 0\n"); return 7; }
        HIPCK(hipMalloc((void**)&dgt, hgt.size())); HIPCK(hipMalloc((void**)&dm, 16)); HIPCK(hipMalloc(&mws, mwsb));
        HIPCK(hipMemcpyAsync(dgt, hgt.data(), hgt.size(), hipMemcpyHostToDevice, stream));
        ABICK(iefvad_auc_ap(logits, dgt, (int64_t)n, 16, dm, dm + 1, mws, mwsb, stream));
        HIPCK(hipStreamSynchronize(stream));
        HIPCK(hipMemcpy(h_metric, dm, 16, hipMemcpyDeviceToHost));
        if (!(h_metric[0] > 0.0 && h_metric[0] < 1.0 && h_metric[1] > 0.0 && h_metric[1] < 1.0)) { fprintf(stderr, "AUC %g / AP %g out of range\n", h_metric[0], h_metric[1]); return 7; }
        (void)hipFree(dgt); (void)hipFree(dm); (void)hipFree(mws);
    }
    printf("metrics AUC %.15f AP %.15f\n", h_metric[0], h_metric[1]);
    // two of the VadCLIP-residue entries from plain C++: the distance adjacency against its closed form, and one graph
    // convolution with that adjacency and an identity weight (out = adj x + x: checked on the host for one output row)
    {
        const int Tn = 128, Dn = 128;
        float *dadj = nullptr, *dx = nullptr, *dw = nullptr, *dout = nullptr;
        void* gws = nullptr;
        const size_t gwsb = iefvad_gcn_workspace_bytes(1, Tn, Dn, Dn, 1);
        HIPCK(hipMalloc((void**)&dadj, Tn * Tn * 4)); HIPCK(hipMalloc((void**)&dx, Tn * Dn * 4)); HIPCK(hipMalloc((void**)&dw, Dn * Dn * 4));
        HIPCK(hipMalloc((void**)&dout, Tn * Dn * 4)); HIPCK(hipMalloc(&gws, gwsb));
        std::vector<float> hx((size_t)Tn * Dn), hw((size_t)Dn * Dn, 0.f), hadj((size_t)Tn * Tn), ho((size_t)Tn * Dn);
        for (size_t j = 0; j < hx.size(); ++j) hx[j] = (float)((j * 2654435761u >> 20) & 1023) / 1024.f - 0.5f;
        for (int j = 0; j < Dn; ++j) hw[(size_t)j * Dn + j] = 1.f;
        HIPCK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        ABICK(iefvad_distance_adj(1, Tn, dadj, stream));
        ABICK(iefvad_gcn_forward(dx, dadj, dw, nullptr, nullptr, nullptr, 1, 0, 1, Tn, Dn, Dn, dout, gws, gwsb, stream));
        HIPCK(hipStreamSynchronize(stream));
        HIPCK(hipMemcpy(hadj.data(), dadj, hadj.size() * 4, hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int i = 0; i < Tn; ++i)
            for (int j = 0; j < Tn; ++j) {
                const double want = exp(-fabs((double)(i - j)) / exp(1.0));
                if (fabs(hadj[(size_t)i * Tn + j] - want) > worst) worst = fabs(hadj[(size_t)i * Tn + j] - want);
            }
        if (worst > 1e-6) { fprintf(stderr, "iefvad_distance_adj differs from exp(-|i-j|/e) by %g\n", worst); return 8; }
        for (int c = 0; c < Dn; c += 37) {
            double want = hx[(size_t)5 * Dn + c];
            for (int j = 0; j < Tn; ++j) want += (double)hadj[(size_t)5 * Tn + j] * hx[(size_t)j * Dn + c];
            if (fabs(ho[(size_t)5 * Dn + c] - want) > 1e-4) { fprintf(stderr, "iefvad_gcn_forward row 5 col %d: %g vs %g\n", c, ho[(size_t)5 * Dn + c], want); return 8; }
        }
        (void)hipFree(dadj); (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dout); (void)hipFree(gws);
    }

    // the training step's two helpers from plain C++: the loops' NaN rule (ucf_train.py:50-53) and the many-matrix split (a matrix and
    // its transpose in one launch: plane 0 of the transposed split at (c, r) is plane 0 of the plain split at (r, c))
    {
        const int Rn = 128, Cn = 64;
        const size_t n = (size_t)Rn * Cn;
        std::vector<float> ha(n), hb(n);
        for (size_t j = 0; j < n; ++j) { ha[j] = (float)((j * 2654435761u >> 18) & 4095) / 512.f - 4.f; hb[j] = ha[j] * 0.25f; }
        ha[17] = NAN; ha[99] = INFINITY; hb[5] = -INFINITY;      // a: NaN -> repaired (its infinity too); b: no NaN -> every bit kept
        float *da = nullptr, *db = nullptr;
        unsigned* dflags = nullptr;
        uint16_t *dp = nullptr, *dpt = nullptr;
        HIPCK(hipMalloc((void**)&da, n * 4)); HIPCK(hipMalloc((void**)&db, n * 4)); HIPCK(hipMalloc((void**)&dflags, 8));
        HIPCK(hipMalloc((void**)&dp, 3 * n * 2)); HIPCK(hipMalloc((void**)&dpt, 3 * n * 2));
        HIPCK(hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice)); HIPCK(hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice));
        ABICK(iefvad_nan_rule(da, db, n, dflags, stream));
        const float* srcs[2] = {db, db};
        void* dsts[2] = {dp, dpt};
        const size_t ns[2] = {n, n};
        const int32_t rows[2] = {0, Rn};
        ABICK(iefvad_split_bf16x3_many(srcs, dsts, ns, rows, 2, stream));
        HIPCK(hipStreamSynchronize(stream));
        unsigned hflags[2];
        std::vector<float> ra(n), rb(n);
        std::vector<uint16_t> hp(3 * n), hpt(3 * n);
        HIPCK(hipMemcpy(hflags, dflags, 8, hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(ra.data(), da, n * 4, hipMemcpyDeviceToHost)); HIPCK(hipMemcpy(rb.data(), db, n * 4, hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(hp.data(), dp, 3 * n * 2, hipMemcpyDeviceToHost)); HIPCK(hipMemcpy(hpt.data(), dpt, 3 * n * 2, hipMemcpyDeviceToHost));
        if (hflags[0] != 1 || hflags[1] != 0 || ra[17] != 0.f || ra[99] != 3.402823466e+38f || memcmp(rb.data(), hb.data(), n * 4) != 0 || ra[18] != ha[18]) {
            fprintf(stderr, "iefvad_nan_rule: flags %u %u, a[17] %g a[99] %g\n", hflags[0], hflags[1], ra[17], ra[99]);
            return 9;
        }
        for (int p = 0; p < 3; ++p)
            for (int r = 0; r < Rn; r += 7)
                for (int c = 0; c < Cn; c += 5)
                    if (hp[p * n + (size_t)r * Cn + c] != hpt[p * n + (size_t)c * Rn + r]) { fprintf(stderr, "iefvad_split_bf16x3_many: plane %d (%d, %d)\n", p, r, c); return 9; }
        (void)hipFree(da); (void)hipFree(db); (void)hipFree(dflags); (void)hipFree(dp); (void)hipFree(dpt);
    }

    printf("abi_driver OK: B=%d L=%d K=%d total %.3f ms, %d GEMM launches, forward_videos OK, gather through librccl OK, auc_ap / distance_adj / gcn_forward / nan_rule / split_many OK\n", B, L, K, st.total_ms, st.gemm_launches);
    iefvad_destroy(h);
    (void)hipFree(ws); (void)hipFree(logits); (void)hipFree(dev); (void)hipStreamDestroy(stream);
    return 0;
}
