// Orchestration of the train-mode forward and of the model's backward pass (include/iefvad.h: iefvad_train_forward /
// iefvad_train_backward; kernels in backward.h).  Included by iefvad.hip behind the handle, launch_proj and the error helpers.
//
// What the reference does per training step (/root/reference/train/ucf_train.py:43-106, train/xd_train.py:35-78): model.train(),
// outputs = model(img, ev, None, prompt_text, lengths), the three loss terms, loss.backward(), optimizer.step().  Here:
//   iefvad_train_forward   the forward of imf_vad.py:109-161 in train mode (attention dropout, imf_vad.py:70) that KEEPS what the
//                          backward needs, in a caller-owned buffer:
//                            per modality and layer: the layer input x_l, q | k | v (q pre-scaled by 1 / sqrt(96)), the attention
//                            probabilities P (ONE tensor, the dropout mask in its sign bits -- attention_split.h, backward.h),
//                            the attention output, the pre-LayerNorm sum;
//                            the last LayerNorm output, the whitened rows, mu and logvar of both modalities;
//                            the refinement states z_0 .. z_K and hidden activations h_0 .. h_{K-1}
//   iefvad_train_backward  the gradients of every parameter, given the gradients of the eight outputs
// Arithmetic: the dense projections of the forward run on the handle's kernels (fp32 MFMA, or the exact bf16x6 split when
// the batch fills its grid); every product of the backward and the two attention products of the train forward run on
// iefvad_bgemm_f32_kernel (fp32 MFMA).  Reductions over rows are fixed-order (no atomics): gradients are bit-reproducible.
#pragma once

// ---- layout of the caller's training buffer, in floats ----------------------------------------------------------------------------
struct TrainLayout {
    size_t U;                       // one [rows, 768] tensor
    size_t PU;                      // one [B, 8, 256, 256] tensor
    size_t x[2][IEFVAD_MAX_LAYERS + 1], qkv[2][IEFVAD_MAX_LAYERS], att[2][IEFVAD_MAX_LAYERS], s[2][IEFVAD_MAX_LAYERS];
    size_t P[2][IEFVAD_MAX_LAYERS];
    size_t E[2], mu[2], lv[2];
    size_t z[IEFVAD_MAX_STEPS + 1], hid[IEFVAD_MAX_STEPS];
    size_t logits;
    // backward scratch
    size_t g, da, dh[2], gx, datt, dqkv, dP, part, cpart, rpart;
    size_t part_floats, cpart_floats, rpart_floats;
    size_t total;
};

static const int kColsumRows = 128;

static int splitk_splits(int rows, int n_out) {
    const int tiles = (n_out / 128) * (IEF_D / 128);
    int want = (1024 + tiles - 1) / tiles, s = 1;
    while (s < want && s < 64) s <<= 1;
    while (s > 1 && ((rows / 16) % s)) s >>= 1;
    return s;
}

// Train-mode attention runs as two fused launches per layer in bf16x6 (attention_split.h TRAIN / BWD) and as the three-launch path in
// the f32 arithmetic (or with IEFVAD_TRAIN_ATTN=unfused, the A/B test's switch, read when a forward starts; the backward follows its
// forward's record).  Both keep P and dropout(P) as ONE sign-carrying tensor: the layout is the same.
static bool train_attn_fused(const iefvad_handle* h) {
    const char* v = getenv("IEFVAD_TRAIN_ATTN");
    return h->cfg.compute == IEFVAD_COMPUTE_BF16X6 && !(v && v[0] == 'u');
}

static TrainLayout train_layout(int L, int K, int B) {
    TrainLayout t;
    memset(&t, 0, sizeof(t));
    const size_t rows = (size_t)B * IEF_T;
    t.U = rows * IEF_D;
    t.PU = (size_t)B * IEF_H * IEF_T * IEF_T;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
    for (int m = 0; m < 2; ++m) {
        for (int l = 0; l <= L; ++l) t.x[m][l] = take(t.U);
        for (int l = 0; l < L; ++l) {
            t.qkv[m][l] = take(3 * t.U);
            t.att[m][l] = take(t.U);
            t.s[m][l] = take(t.U);
            t.P[m][l] = take(t.PU);
        }
        t.E[m] = take(t.U);
        t.mu[m] = take(t.U);
        t.lv[m] = take(t.U);
    }
    for (int k = 0; k <= K; ++k) t.z[k] = take(t.U);
    for (int k = 0; k < K; ++k) t.hid[k] = take(t.U);
    t.logits = take(rows);
    t.g = take(t.U);
    t.da = take(t.U);
    // Scratch of the backward's second half (fusion, heads, encoder layers).  By then the refinement steps have been differentiated and
    // their saved states z_0 .. z_K, h_0 .. h_{K-1} -- (2K + 1) U contiguous floats -- are dead: the scratch tensors live THERE as far as
    // they fit (all of them from K = 6 on), largest first.  The backward therefore consumes the buffer: one backward per forward
    // (iefvad_train_backward retires the forward's record).
    {
        size_t dead = t.z[0];
        const size_t dead_end = t.z[0] + (size_t)(2 * K + 1) * t.U;
        auto place = [&](size_t n) {
            const size_t n64 = (n + 63) & ~(size_t)63;
            if (dead + n64 <= dead_end) { const size_t r = dead; dead += n64; return r; }
            return take(n);
        };
        t.dqkv = place(3 * t.U);
        t.dP = place(t.PU);
        t.dh[0] = place(2 * t.U);
        t.dh[1] = place(2 * t.U);
        t.gx = place(t.U);
        t.datt = place(t.U);
    }
    size_t pf = 0;
    const int nouts[3] = {IEF_D, 2 * IEF_D, 3 * IEF_D};
    for (int n : nouts) {
        const size_t f = (size_t)splitk_splits((int)rows, n) * n * IEF_D;
        if (f > pf) pf = f;
    }
    t.part_floats = pf;
    t.part = take(pf);
    t.cpart_floats = ((rows + kColsumRows - 1) / kColsumRows) * 3 * IEF_D;
    t.cpart = take(t.cpart_floats);
    t.rpart_floats = ((rows + BWD_ROWS_PER_BLOCK - 1) / BWD_ROWS_PER_BLOCK) * (2 * IEF_D + 1);
    t.rpart = take(t.rpart_floats);
    t.total = o;
    return t;
}

extern "C" size_t iefvad_train_workspace_bytes(const iefvad_handle* h, int32_t B) {
    if (!h || B <= 0 || B > 4096) return 0;
    return train_layout(h->cfg.num_layers, h->cfg.num_steps, B).total * sizeof(float) + 256;
}

// ---- launch helpers -----------------------------------------------------------------------------------------------------------------
static int launch_bgemm(const BgemmArgs& a, bool akc, bool bkc, int nz, hipStream_t stream) {
    const int bn = (a.N % 128 == 0) ? 128 : 96;
    if (a.M % BG_BM || a.N % bn || a.K % BG_BK || a.M <= 0 || a.N <= 0 || a.K <= 0 || nz <= 0 || a.nz2 <= 0)
        return fail("bgemm: shape M=%d N=%d K=%d is not a multiple of the 128 x %d x 16 tile", a.M, a.N, a.K, bn);
    if ((a.lda | a.ldb) & 3) return fail("bgemm: leading dimensions must be multiples of 4 floats");
    dim3 grid((unsigned)((a.M / BG_BM) * (a.N / bn) * nz));
#define BG_LAUNCH(AK, BK_, BN_) hipLaunchKernelGGL((iefvad_bgemm_f32_kernel<AK, BK_, BN_>), grid, dim3(256), 0, stream, a)
    if (bn == 128) {
        if (akc && bkc) BG_LAUNCH(true, true, 128);
        else if (akc) BG_LAUNCH(true, false, 128);
        else if (bkc) BG_LAUNCH(false, true, 128);
        else BG_LAUNCH(false, false, 128);
    } else {
        if (akc && bkc) BG_LAUNCH(true, true, 96);
        else if (akc) BG_LAUNCH(true, false, 96);
        else if (bkc) BG_LAUNCH(false, true, 96);
        else BG_LAUNCH(false, false, 96);
    }
#undef BG_LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
}

// bf16x6 handles: the three-plane splits of every TRANSPOSED projection matrix, so that dX = dY W runs as the NT product
// dX[rows, 768] = dY[rows, n_out] (W^T)[768, n_out]^T on iefvad_gemm_split_n128_kernel (fp32-accurate, 1.7x the fp32 MFMA rate)
static int ensure_train_planes(iefvad_handle* h, hipStream_t stream) {
    if (h->cfg.compute != IEFVAD_COMPUTE_BF16X6 || h->tplanes_valid) return 0;
    const int L = h->cfg.num_layers, K = h->cfg.num_steps;
    const size_t D = IEF_D, DD = D * D;
    const size_t nb = 2 * (size_t)L * (3 * DD + DD) + 2 * (2 * DD) + (size_t)K * 2 * DD;
    if (!h->arena_st) HIP_TRY(hipMalloc((void**)&h->arena_st, 3 * nb * sizeof(bf16_t)));
    if (!h->zero_bias) {
        HIP_TRY(hipMalloc((void**)&h->zero_bias, 3 * D * sizeof(float)));
        HIP_TRY(hipMemsetAsync(h->zero_bias, 0, 3 * D * sizeof(float), stream));
    }
    bf16_t* q = h->arena_st;
    SplitMany sm(stream);                     // one launch for all of them (gemm_split.h: transpose and split in one pass)
    auto tsplit = [&](bf16_t** dst, const float* W, int n_out) -> int {      // W [n_out, 768] -> planes of W^T [768, n_out]
        *dst = q;
        q += 3 * (size_t)n_out * D;
        return sm.add(W, *dst, (size_t)n_out * D, n_out);
    };
    for (int m = 0; m < 2; ++m) {
        for (int l = 0; l < L; ++l) {
            if (int rc = tsplit(&h->in_wst[m][l], h->in_w[m][l], 3 * IEF_D)) return rc;
            if (int rc = tsplit(&h->out_wst[m][l], h->out_w[m][l], IEF_D)) return rc;
        }
        if (int rc = tsplit(&h->head_wst[m], h->head_w[m], 2 * IEF_D)) return rc;
    }
    for (int k = 0; k < K; ++k) {
        if (int rc = tsplit(&h->ref_w1st[k], h->ref_w1[k], IEF_D)) return rc;
        if (int rc = tsplit(&h->ref_w2st[k], h->ref_w2[k], IEF_D)) return rc;
    }
    if (int rc = sm.flush()) return rc;
    h->tplanes_valid = true;
    return 0;
}

// dX[rows, n_in] = alpha * dY[rows, n_out] * W[n_out, n_in] (+ R) (gated by G): the input gradient of a Linear whose weight is stored
// [out, in] as torch stores it.  `planes_t` (bf16x6 handles, full grids): the same product on the split kernel.
static int launch_dx(iefvad_handle* h, const bf16_t* planes_t, const float* dY, int ldy, const float* W, int n_out, int n_in, float* dX,
                     const float* R, const float* G, float alpha, int rows, hipStream_t stream) {
    if (planes_t && n_in == IEF_D && ldy == n_out && split_eligible(rows, IEF_D, n_out, 1) && (!R || !G) && (G || alpha == 1.f)) {
        GemmBArgs g;
        memset(&g, 0, sizeof(g));
        g.M = rows; g.N = IEF_D; g.K = n_out; g.lda = n_out; g.ldc = IEF_D; g.alpha = alpha;
        g.epi = G ? EPI_GATE : (R ? EPI_BIAS_RESID : EPI_BIAS);
        g.wplane = IEF_D * n_out * 2;
        g.p[0].A = (const bf16_t*)dY;            // fp32 data behind the typed pointer
        g.p[0].W = planes_t; g.p[0].bias = h->zero_bias; g.p[0].C = dX; g.p[0].R = G ? G : R;
        Timer tm;
        return launch_gemm_split(g, 1, stream, tm, ST_REFINE);
    }
    BgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = dY; a.B = W; a.C = dX; a.R = R; a.G = G;
    a.M = rows; a.N = n_in; a.K = n_out; a.lda = ldy; a.ldb = n_in; a.ldc = n_in; a.nz2 = 1; a.alpha = alpha;
    (void)h;
    return launch_bgemm(a, true, false, 1, stream);
}

// a batch of fixed-order reductions for one launch of iefvad_reduce_parts_multi_kernel (null destinations are skipped)
struct ReduceBatch {
    ReduceJobs j;
    ReduceBatch() { memset(&j, 0, sizeof(j)); }
    void add(const float* part, size_t stride, int nparts, size_t n, float* out, float alpha) {
        if (!out || n == 0 || j.count == 4) return;
        const int c = j.count++;
        j.part[c] = part; j.stride[c] = stride; j.nparts[c] = nparts; j.n[c] = n; j.out[c] = out; j.alpha[c] = alpha;
        j.first[c + 1] = j.first[c] + (unsigned)((n + RED_STRIP - 1) / RED_STRIP);
    }
    void launch(hipStream_t stream) {
        if (j.count == 0) return;
        hipLaunchKernelGGL(iefvad_reduce_parts_multi_kernel, dim3(j.first[j.count]), dim3(256), 0, stream, j);
    }
};

// dW[n_out, 768] = alpha * dY^T X over the rows, split-K with fixed-order reduction; dW may be null (frozen parameter).
// `dW2` (nullable) receives rows [n_split, n_out) as a tensor of its own (the stacked mu | logvar heads).
static int launch_dw(const float* dY, int ldy, int n_out, const float* X, float* dW, float* dW2, int n_split, float alpha, int rows,
                     float* part, size_t part_floats, hipStream_t stream, bool split_tn = false, float* db = nullptr, float* db2 = nullptr,
                     float* cpart = nullptr, size_t cpart_floats = 0, bool* db_done = nullptr) {
    if (db_done) *db_done = false;
    if (!dW && !dW2) return 0;
    const int splits = splitk_splits(rows, n_out);
    const size_t per = (size_t)n_out * IEF_D;
    if ((size_t)splits * per > part_floats) return fail("train_backward: split-K scratch too small");
    const int ms = rows / splits;
    if (split_tn && ms % TN_BK == 0 && n_out % 128 == 0 && (ldy & 3) == 0) {
        // bf16x6 handles: the same partial products on the bf16 matrix pipe (gemm_split_tn.h), the same fixed-order reduction behind them.
        // Default: the 128 x 256 block (64 x 128 per wave, two workgroups per CU) over as many RAGGED row slices as fill the chip's
        // workgroup slots once (at most the slices the scratch was sized for); IEFVAD_TN=128 keeps the 128 x 128 block over the
        // power-of-two slices (A/B).
        const char* tn_env = getenv("IEFVAD_TN");
        const bool wide = !(tn_env && tn_env[0] == '1');
        TnArgs g;
        memset(&g, 0, sizeof(g));
        g.A = dY; g.B = X; g.C = part; g.M = n_out; g.N = IEF_D; g.lda = ldy; g.ldb = IEF_D; g.ldc = IEF_D;
        g.nk_total = rows / TN_BK;
        int slices = splits;
        if (wide) {
            const int tiles = (n_out / 128) * (IEF_D / 256);
            slices = (2 * g_num_cus) / tiles;
            if (slices > g.nk_total / 8) slices = g.nk_total / 8;
            if (slices > splits) slices = splits;          // the scratch (part, cpart) holds `splits` partial results
            if (slices < 1) slices = 1;
        }
        g.slices = slices;
        g.tiles_n = wide ? IEF_D / 256 : IEF_D / 128;
        // the bias gradient of the same Linear (column sums of dY) rides along in the first column block's workgroups
        const bool with_db = (db || db2) && cpart && db_done && (size_t)slices * n_out <= cpart_floats;
        g.colsum = with_db ? cpart : nullptr;
        if (wide)
            hipLaunchKernelGGL(iefvad_gemm_split_tn256_kernel, dim3((unsigned)((n_out / 128) * g.tiles_n * slices)), dim3(256), TN_LDS_BYTES_OF(4), stream, g);
        else
            hipLaunchKernelGGL(iefvad_gemm_split_tn_kernel, dim3((unsigned)((n_out / 128) * g.tiles_n * slices)), dim3(256), TN_LDS_BYTES, stream, g);
        HIP_TRY(hipGetLastError());
        // one launch reduces the weight partials (both halves of a stacked product) and the bias partials that rode along
        ReduceBatch rb;
        rb.add(part, per, slices, (size_t)(n_split > 0 ? n_split : n_out) * IEF_D, dW, alpha);
        if (dW2) rb.add(part + (size_t)n_split * IEF_D, per, slices, (size_t)(n_out - n_split) * IEF_D, dW2, alpha);
        if (with_db) {
            rb.add(cpart, (size_t)n_out, slices, (size_t)(n_split > 0 ? n_split : n_out), db, alpha);
            if (db2) rb.add(cpart + n_split, (size_t)n_out, slices, (size_t)(n_out - n_split), db2, alpha);
            *db_done = true;
        }
        rb.launch(stream);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    BgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = dY; a.B = X; a.C = part;
    a.M = n_out; a.N = IEF_D; a.K = ms; a.lda = ldy; a.ldb = IEF_D; a.ldc = IEF_D;
    a.a1 = (long long)ms * ldy; a.b1 = (long long)ms * IEF_D; a.c1 = (long long)per; a.nz2 = 1; a.alpha = 1.f;
    if (int rc = launch_bgemm(a, false, false, splits, stream)) return rc;
    if (dW) {
        const size_t n = (size_t)(n_split > 0 ? n_split : n_out) * IEF_D;
        hipLaunchKernelGGL(iefvad_reduce_parts_kernel, dim3((unsigned)((n + RED_STRIP - 1) / RED_STRIP)), dim3(256), 0, stream, part, per, splits, n, dW, alpha);
    }
    if (dW2) {
        const size_t n = (size_t)(n_out - n_split) * IEF_D;
        hipLaunchKernelGGL(iefvad_reduce_parts_kernel, dim3((unsigned)((n + RED_STRIP - 1) / RED_STRIP)), dim3(256), 0, stream, part + (size_t)n_split * IEF_D,
                           per, splits, n, dW2, alpha);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// db[ncols] = alpha * column sums of Y[rows, ld]; db2 (nullable) takes columns [n_split, ncols)
static int launch_db(const float* Y, int ld, int ncols, float* db, float* db2, int n_split, float alpha, int rows, float* cpart,
                     hipStream_t stream) {
    if (!db && !db2) return 0;
    const int nblk = (rows + kColsumRows - 1) / kColsumRows;
    hipLaunchKernelGGL(iefvad_colsum_kernel, dim3((ncols + 255) / 256, nblk), dim3(256), 0, stream, Y, ld, rows, ncols, kColsumRows, cpart);
    if (db) {
        const size_t n = n_split > 0 ? n_split : ncols;
        hipLaunchKernelGGL(iefvad_reduce_parts_kernel, dim3((unsigned)((n + RED_STRIP - 1) / RED_STRIP)), dim3(256), 0, stream, cpart, (size_t)ncols, nblk, n, db, alpha);
    }
    if (db2) {
        const size_t n = ncols - n_split;
        hipLaunchKernelGGL(iefvad_reduce_parts_kernel, dim3((unsigned)((n + RED_STRIP - 1) / RED_STRIP)), dim3(256), 0, stream, cpart + n_split, (size_t)ncols, nblk,
                           n, db2, alpha);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// What a train-mode forward leaves on the handle for its backward: which buffer it filled, at which batch size, and whether the
// stored probabilities carry a dropout mask.  A handful of forwards may be outstanding (one record per training buffer).
// Five saved tensors are also outputs (mu and logvar of both modalities, the refined state z_K = `fused`) and two are the inputs: where
// the caller hands over fp32 buffers of its own, the forward writes / reads THOSE and the backward reads them again (x0, mu, lv, zK
// below) -- seven device-to-device copies of [rows, 768] per step less (0.33 ms of a 26 ms step).  The caller keeps them untouched
// until the backward has run (include/iefvad.h).
struct TrainRecord {
    const void* ws; int B; bool drop[2][IEFVAD_MAX_LAYERS]; float drop_p[2][IEFVAD_MAX_LAYERS]; bool fused; unsigned long long stamp;
    const float* x0[2]; float* mu[2]; float* lv[2]; float* zK;
};
struct TrainState { TrainRecord rec[8]; unsigned long long clock = 0; };

static void release_train(iefvad_handle* h) {
    delete h->train;
    h->train = nullptr;
}

static int train_check(const iefvad_handle* h, int32_t B, const void* ws, size_t ws_bytes, const char* who) {
    if (!h) return fail("%s: null handle", who);
    if (!h->weights_set) return fail("%s: weights not set", who);
    if (h->cfg.compute != IEFVAD_COMPUTE_F32 && h->cfg.compute != IEFVAD_COMPUTE_BF16X6)
        return fail("%s: training runs in the fp32-accurate arithmetics only (compute f32 or bf16x6)", who);
    if (B <= 0 || B > 4096) return fail("%s: B = %d outside 1..4096", who, B);
    const size_t need = iefvad_train_workspace_bytes(h, B);
    if (!ws || ws_bytes < need) return fail("%s: training buffer too small (%zu < %zu bytes)", who, ws_bytes, need);
    if ((uintptr_t)ws & 255) return fail("%s: the training buffer must be 256-byte aligned", who);
    return 0;
}

// ---- the trainers' NaN rule on the device ----------------------------------------------------------------------------------------------
// /root/reference/train/ucf_train.py:50-53 (xd_train.py:40-43): `if torch.isnan(x).any(): x = torch.nan_to_num(x, nan=0.0)` per input
// tensor -- in torch an isnan pass that writes a bool tensor, a reduction, a HOST read of the flag (the step's launches cannot run
// ahead of the device) and, when it fires, a rewriting pass.  Here: one scan of both tensors (a flag word each, set by any thread
// that sees a NaN) and a repair launch whose workgroups return at once when their tensor's flag is clear; nothing comes back to the
// host.  The repair is torch.nan_to_num(x, nan=0.0) element for element (NaN -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX), IN PLACE.
__global__ __launch_bounds__(256) void iefvad_nan_scan_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n4, unsigned* flags) {
    const float* x = blockIdx.y ? b : a;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = ((const f32x4*)x)[i];
        bad |= (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) flags[blockIdx.y] = 1u;      // every writer writes the same value
}
__global__ __launch_bounds__(256) void iefvad_nan_fix_kernel(float* a, float* b, size_t n4, const unsigned* flags) {
    if (!flags[blockIdx.y]) return;
    float* x = blockIdx.y ? b : a;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = ((f32x4*)x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f = v[e];
            v[e] = f != f ? 0.f : f == __builtin_inff() ? 3.402823466e+38f : f == -__builtin_inff() ? -3.402823466e+38f : f;
        }
        ((f32x4*)x)[i] = v;
    }
}
extern "C" int iefvad_nan_rule(float* a, float* b, size_t n, void* flags_ws, void* stream_) {
    if (!a || !b || !flags_ws) return fail("iefvad_nan_rule: null argument");
    if (n % 4 || (((uintptr_t)a | (uintptr_t)b) & 15) || ((uintptr_t)flags_ws & 7)) return fail("iefvad_nan_rule: n %% 4 == 0 and 16-byte aligned tensors, please");
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipMemsetAsync(flags_ws, 0, 8, stream));
    hipLaunchKernelGGL(iefvad_nan_scan_kernel, dim3(2048, 2), dim3(256), 0, stream, a, b, n / 4, (unsigned*)flags_ws);
    hipLaunchKernelGGL(iefvad_nan_fix_kernel, dim3(2048, 2), dim3(256), 0, stream, a, b, n / 4, (const unsigned*)flags_ws);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- train-mode forward -------------------------------------------------------------------------------------------------------------
extern "C" int iefvad_train_forward(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B,
                                    const iefvad_train_options* opt, void* train_ws, size_t train_ws_bytes, const iefvad_outputs* out,
                                    void* stream_) {
    if (!h) return fail("iefvad_train_forward: null handle");
    const bool fused = train_attn_fused(h);
    if (int rc = train_check(h, B, train_ws, train_ws_bytes, "iefvad_train_forward")) return rc;
    if (!img || !ev || !out || !opt) return fail("iefvad_train_forward: null argument");
    if (in_dtype != IEFVAD_IN_F32 && in_dtype != IEFVAD_IN_F16 && in_dtype != IEFVAD_IN_BF16)
        return fail("iefvad_train_forward: unknown in_dtype %d", in_dtype);
    if (((uintptr_t)img | (uintptr_t)ev) & 15) return fail("iefvad_train_forward: buffers must be 16-byte aligned");
    if (((uintptr_t)out->fused | (uintptr_t)out->image_mu | (uintptr_t)out->event_mu | (uintptr_t)out->image_logvar | (uintptr_t)out->event_logvar) & 15)
        return fail("iefvad_train_forward: output buffers must be 16-byte aligned");
    const iefvad_config& c = h->cfg;
    const int L = c.num_layers, K = c.num_steps;
    for (int m = 0; m < 2; ++m)
        for (int l = 0; l < L; ++l)
            if (!(opt->dropout_p[m][l] >= 0.f && opt->dropout_p[m][l] < 1.f))
                return fail("iefvad_train_forward: dropout_p[%d][%d] = %g outside [0, 1)", m, l, (double)opt->dropout_p[m][l]);
    hipStream_t stream = (hipStream_t)stream_;
    const TrainLayout t = train_layout(L, K, B);
    float* ws = (float*)train_ws;
    const int rows = B * IEF_T;
    Timer tm;
    if (!h->train) {
        h->train = new (std::nothrow) TrainState();
        if (!h->train) return fail("iefvad_train_forward: out of host memory");
        memset(h->train->rec, 0, sizeof(h->train->rec));
    }
    TrainRecord* rec = nullptr;
    {   // the record of this forward replaces the one of the same buffer, else the oldest
        TrainState& ts = *h->train;
        int slot = 0;
        for (int i = 0; i < 8; ++i) {
            if (ts.rec[i].ws == train_ws) { slot = i; break; }
            if (ts.rec[i].stamp < ts.rec[slot].stamp) slot = i;
        }
        TrainRecord& r = ts.rec[slot];
        rec = &r;
        r.ws = train_ws; r.B = B; r.fused = fused; r.stamp = ++ts.clock;
        const bool f32in = in_dtype == IEFVAD_IN_F32;
        r.x0[0] = f32in ? (const float*)img : ws + t.x[0][0];
        r.x0[1] = f32in ? (const float*)ev : ws + t.x[1][0];
        r.mu[0] = out->image_mu ? out->image_mu : ws + t.mu[0];
        r.mu[1] = out->event_mu ? out->event_mu : ws + t.mu[1];
        r.lv[0] = out->image_logvar ? out->image_logvar : ws + t.lv[0];
        r.lv[1] = out->event_logvar ? out->event_logvar : ws + t.lv[1];
        r.zK = out->fused ? out->fused : ws + t.z[K];
        for (int m = 0; m < 2; ++m)
            for (int l = 0; l < IEFVAD_MAX_LAYERS; ++l) {
                r.drop[m][l] = l < L && (opt->dropout_p[m][l] > 0.f || opt->keep_mask);
                r.drop_p[m][l] = l < L ? opt->dropout_p[m][l] : 0.f;
            }
    }
    const bool splitmb = c.compute == IEFVAD_COMPUTE_BF16X6 && split_eligible(rows, IEF_D, IEF_D, 1);
    const float qscale = 1.0f / sqrtf((float)IEF_DH);
    const float factor = (c.noise_model == IEFVAD_NOISE_STUDENT_T) ? (c.nu + 1.0f) / c.nu : 1.0f;

    // `.to(torch.float)` (imf_vad.py:41-42) into the saved layer-0 inputs; fp32 inputs are read where they are
    auto xin = [&](int m, int l) -> const float* { return l == 0 ? rec->x0[m] : ws + t.x[m][l]; };
    auto zst = [&](int k) -> float* { return k == K ? rec->zK : ws + t.z[k]; };
    if (in_dtype != IEFVAD_IN_F32) {
        const int rc = (in_dtype == IEFVAD_IN_F16)
                           ? launch_cast<__half>(img, ev, ws + t.x[0][0], ws + t.x[1][0], nullptr, nullptr, t.U, 2, stream)
                           : launch_cast<__hip_bfloat16>(img, ev, ws + t.x[0][0], ws + t.x[1][0], nullptr, nullptr, t.U, 2, stream);
        if (rc) return rc;
    }

    for (int l = 0; l < L; ++l) {
        Proj p;
        memset(&p, 0, sizeof(p));
        p.N = 3 * IEF_D; p.ldc = 3 * IEF_D; p.epi = EPI_QKV; p.qcols = IEF_D; p.nz = 2;
        p.alpha = qscale;                               // q * 1/sqrt(dh) before the bmm, as F.multi_head_attention_forward does
        for (int m = 0; m < 2; ++m) {
            p.A32[m] = xin(m, l); p.W32[m] = h->in_w[m][l]; p.Ws[m] = h->in_ws[m][l]; p.bias[m] = h->in_b[m][l];
            p.C[m] = ws + t.qkv[m][l];
        }
        if (int rc = launch_proj(p, c.compute, splitmb, rows, stream, tm, ST_QKV)) return rc;

        // bf16x6: S = q k^T -> softmax -> dropout -> Pd v in ONE launch for both modalities (attention_split.h, TRAIN: the eval kernel
        // with the sign-carrying P stored for the backward); IEFVAD_TRAIN_ATTN=unfused keeps the three launches below (A/B: read per
        // call, the test flips it between two forwards).  The fp32 arithmetic keeps them: its products stay on the fp32 MFMA instruction.
        if (fused) {
            AttnArgs aa;
            AttnTrainArgs tx;
            memset(&aa, 0, sizeof(aa));
            memset(&tx, 0, sizeof(tx));
            aa.nchunks = B;
            for (int m = 0; m < 2; ++m) {
                const float pdrop = opt->dropout_p[m][l];
                const bool drop = pdrop > 0.f || opt->keep_mask;
                aa.qkv[m] = ws + t.qkv[m][l];
                aa.out[m] = ws + t.att[m][l];
                tx.P[m] = ws + t.P[m][l];
                tx.drop[m] = drop ? 1 : 0;
                tx.keep[m] = opt->keep_mask ? opt->keep_mask + ((size_t)m * L + l) * t.PU : nullptr;
                tx.seed[m] = opt->seed * 0x100000001B3ull + (unsigned long long)(m * IEFVAD_MAX_LAYERS + l + 1) * 0x9E3779B97F4A7C15ull;
                tx.drop_p[m] = pdrop;
            }
            if (opt->keep_mask)
                hipLaunchKernelGGL(iefvad_attention_split_train_mask_kernel, dim3(IEF_H, 2, 2 * B), dim3(256), ATS_LDS_BYTES, stream, aa, tx);
            else
                hipLaunchKernelGGL(iefvad_attention_split_train_kernel, dim3(IEF_H, 2, 2 * B), dim3(256), ATS_LDS_BYTES, stream, aa, tx);
            HIP_TRY(hipGetLastError());
        } else
        for (int m = 0; m < 2; ++m) {
            float* qkv = ws + t.qkv[m][l];
            const float pdrop = opt->dropout_p[m][l];
            const bool drop = pdrop > 0.f || opt->keep_mask;
            // S = q k^T per (chunk, head)
            BgemmArgs a;
            memset(&a, 0, sizeof(a));
            a.A = qkv; a.B = qkv + IEF_D; a.C = ws + t.P[m][l];
            a.M = IEF_T; a.N = IEF_T; a.K = IEF_DH; a.lda = a.ldb = 3 * IEF_D; a.ldc = IEF_T;
            a.a1 = a.b1 = (long long)IEF_T * 3 * IEF_D; a.a2 = a.b2 = IEF_DH;
            a.c1 = (long long)IEF_H * IEF_T * IEF_T; a.c2 = (long long)IEF_T * IEF_T; a.nz2 = IEF_H; a.alpha = 1.f;
            if (int rc = launch_bgemm(a, true, true, B * IEF_H, stream)) return rc;
            SoftmaxDropArgs sa;
            sa.S = ws + t.P[m][l];
            sa.drop = drop ? 1 : 0;
            sa.keep = opt->keep_mask ? opt->keep_mask + ((size_t)m * L + l) * t.PU : nullptr;
            sa.seed = opt->seed * 0x100000001B3ull + (unsigned long long)(m * IEFVAD_MAX_LAYERS + l + 1) * 0x9E3779B97F4A7C15ull;
            sa.p = pdrop;
            sa.rows = (long long)B * IEF_H * IEF_T;
            hipLaunchKernelGGL(iefvad_softmax_dropout_kernel, dim3((unsigned)((sa.rows + 3) / 4)), dim3(256), 0, stream, sa);
            HIP_TRY(hipGetLastError());
            // attention output = dropout(P) v
            memset(&a, 0, sizeof(a));
            a.A = ws + t.P[m][l]; a.B = qkv + 2 * IEF_D; a.C = ws + t.att[m][l];
            if (drop) a.a_drop = (float)(1.0 / (1.0 - (double)pdrop));      // dropout(P) from the sign-carrying tensor, in the A tile's staging
            a.M = IEF_T; a.N = IEF_DH; a.K = IEF_T; a.lda = IEF_T; a.ldb = 3 * IEF_D; a.ldc = IEF_D;
            a.a1 = (long long)IEF_H * IEF_T * IEF_T; a.a2 = (long long)IEF_T * IEF_T;
            a.b1 = (long long)IEF_T * 3 * IEF_D; a.b2 = IEF_DH; a.c1 = (long long)IEF_T * IEF_D; a.c2 = IEF_DH; a.nz2 = IEF_H; a.alpha = 1.f;
            if (int rc = launch_bgemm(a, true, false, B * IEF_H, stream)) return rc;
        }

        memset(&p, 0, sizeof(p));
        p.N = IEF_D; p.ldc = IEF_D; p.epi = EPI_BIAS_RESID; p.nz = 2;
        for (int m = 0; m < 2; ++m) {
            p.A32[m] = ws + t.att[m][l]; p.W32[m] = h->out_w[m][l]; p.Ws[m] = h->out_ws[m][l]; p.bias[m] = h->out_b[m][l];
            p.C[m] = ws + t.s[m][l]; p.R[m] = xin(m, l);
        }
        if (int rc = launch_proj(p, c.compute, splitmb, rows, stream, tm, ST_OUT)) return rc;

        LnArgs la;
        memset(&la, 0, sizeof(la));
        la.nrows = rows; la.eps = 1e-5f;
        for (int m = 0; m < 2; ++m) {
            la.x[m] = ws + t.s[m][l]; la.g1[m] = h->norm_w[m][l]; la.b1[m] = h->norm_b[m][l]; la.y[m] = ws + t.x[m][l + 1];
        }
        hipLaunchKernelGGL(iefvad_layernorm_kernel, dim3((rows + ROW_WAVES - 1) / ROW_WAVES, 2), dim3(256), 0, stream, la);
        HIP_TRY(hipGetLastError());
    }
    {   // whitening LayerNorm (imf_vad.py:117,123) as a launch of its own: the backward needs its input, the last norm's output
        LnArgs la;
        memset(&la, 0, sizeof(la));
        la.nrows = rows; la.eps = 1e-5f;
        for (int m = 0; m < 2; ++m) {
            la.x[m] = ws + t.x[m][L]; la.g1[m] = h->whiten_w[m]; la.b1[m] = h->whiten_b[m]; la.y[m] = ws + t.E[m];
        }
        hipLaunchKernelGGL(iefvad_layernorm_kernel, dim3((rows + ROW_WAVES - 1) / ROW_WAVES, 2), dim3(256), 0, stream, la);
        HIP_TRY(hipGetLastError());
    }
    {
        Proj p;
        memset(&p, 0, sizeof(p));
        p.N = 2 * IEF_D; p.ldc = IEF_D; p.epi = EPI_HEADS; p.nz = 2;
        for (int m = 0; m < 2; ++m) {
            p.A32[m] = ws + t.E[m]; p.W32[m] = h->head_w[m]; p.Ws[m] = h->head_ws[m]; p.bias[m] = h->head_b[m];
            p.C[m] = rec->mu[m]; p.C2[m] = rec->lv[m];
        }
        if (int rc = launch_proj(p, c.compute, splitmb, rows, stream, tm, ST_HEAD)) return rc;
    }
    {
        FusionArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.mu_i = rec->mu[0]; fa.lv_i = rec->lv[0]; fa.mu_e = rec->mu[1]; fa.lv_e = rec->lv[1];
        fa.n_i = out->w_i; fa.n_e = out->w_e; fa.z = zst(0);
        fa.n_i_mean = out->w_i_mean; fa.n_e_mean = out->w_e_mean;
        fa.nrows = rows; fa.factor = factor; fa.eps = c.epsilon;
        hipLaunchKernelGGL(iefvad_fusion_kernel, dim3((rows + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, stream, fa);
        HIP_TRY(hipGetLastError());
    }
    for (int k = 0; k < K; ++k) {
        Proj p;
        memset(&p, 0, sizeof(p));
        p.N = IEF_D; p.ldc = IEF_D; p.epi = EPI_BIAS_RELU; p.nz = 1;
        p.A32[0] = zst(k); p.W32[0] = h->ref_w1[k]; p.Ws[0] = h->ref_w1s[k]; p.bias[0] = h->ref_b1[k]; p.C[0] = ws + t.hid[k];
        if (int rc = launch_proj(p, c.compute, splitmb, rows, stream, tm, ST_REFINE)) return rc;
        memset(&p, 0, sizeof(p));
        p.N = IEF_D; p.ldc = IEF_D; p.epi = EPI_REFINE; p.alpha = c.lambda_ref; p.nz = 1;
        p.A32[0] = ws + t.hid[k]; p.W32[0] = h->ref_w2[k]; p.Ws[0] = h->ref_w2s[k]; p.bias[0] = h->ref_b2[k];
        p.C[0] = zst(k + 1); p.R[0] = zst(k);
        if (int rc = launch_proj(p, c.compute, splitmb, rows, stream, tm, ST_REFINE)) return rc;
    }
    float* logits = out->logits ? out->logits : ws + t.logits;
    hipLaunchKernelGGL(iefvad_scorer_kernel, dim3((rows + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, stream, rec->zK, h->cls_w, h->cls_b,
                       logits, rows);
    HIP_TRY(hipGetLastError());
    // (the dict entries of imf_vad.py:152-161 that are also saved tensors were written in place: TrainRecord)
    return 0;
}

// ---- backward -------------------------------------------------------------------------------------------------------------------------
extern "C" int iefvad_train_backward(iefvad_handle* h, int32_t B, void* train_ws, size_t train_ws_bytes, const iefvad_output_grads* dout,
                                     const iefvad_weight_grads* dw, void* stream_) {
    if (!h) return fail("iefvad_train_backward: null handle");
    if (!dout || !dw) return fail("iefvad_train_backward: null argument");
    const TrainRecord* rec = nullptr;
    if (h->train)
        for (int i = 0; i < 8; ++i)
            if (h->train->rec[i].ws == train_ws && h->train->rec[i].stamp) rec = &h->train->rec[i];
    if (!rec || rec->B != B) return fail("iefvad_train_backward: no iefvad_train_forward with B = %d has filled this training buffer", B);
    if (int rc = train_check(h, B, train_ws, train_ws_bytes, "iefvad_train_backward")) return rc;
    const iefvad_config& c = h->cfg;
    const int L = c.num_layers, K = c.num_steps;
    hipStream_t stream = (hipStream_t)stream_;
    const TrainLayout t = train_layout(L, K, B);
    float* ws = (float*)train_ws;
    const int rows = B * IEF_T;
    const int nblk = (rows + BWD_ROWS_PER_BLOCK - 1) / BWD_ROWS_PER_BLOCK;
    const float factor = (c.noise_model == IEFVAD_NOISE_STUDENT_T) ? (c.nu + 1.0f) / c.nu : 1.0f;
    const float qscale = 1.0f / sqrtf((float)IEF_DH);
    float* part = ws + t.part;
    float* cpart = ws + t.cpart;
    float* rpart = ws + t.rpart;
    float* g = ws + t.g;
    float* da = ws + t.da;
    if (int rc = ensure_train_planes(h, stream)) return rc;
    const bool tp = h->cfg.compute == IEFVAD_COMPUTE_BF16X6 && h->tplanes_valid;      // dX on the split kernel where the grid fills it
    static const bool tn_off = [] { const char* v = getenv("IEFVAD_TRAIN_TN"); return v && v[0] == '0'; }();
    const bool tn = h->cfg.compute == IEFVAD_COMPUTE_BF16X6 && !tn_off;              // dW on the split TN kernel (gemm_split_tn.h)

    // classifier (imf_vad.py:150)
    {
        ScorerBwdArgs sa;
        sa.dlogit = dout->logits; sa.dfused = dout->fused; sa.z = rec->zK; sa.w = h->cls_w; sa.g = g;
        sa.part_w = rpart; sa.part_b = rpart + (size_t)nblk * IEF_D; sa.rows = rows;
        hipLaunchKernelGGL(iefvad_scorer_bwd_kernel, dim3(nblk), dim3(256), 0, stream, sa);
        ReduceBatch rb;
        rb.add(rpart, (size_t)IEF_D, nblk, (size_t)IEF_D, dw->cls_w, 1.f);
        rb.add(rpart + (size_t)nblk * IEF_D, (size_t)1, nblk, (size_t)1, dw->cls_b, 1.f);
        rb.launch(stream);
        HIP_TRY(hipGetLastError());
    }
    // refinement steps, last to first (imf_vad.py:146-149): z_{k+1} = z_k - lambda (W2 relu(W1 z_k + b1) + b2); g = d z_{k+1}
    for (int k = K - 1; k >= 0; --k) {
        const float nl = -c.lambda_ref;
        { bool db_done = false;
            if (int rc = launch_dw(g, IEF_D, IEF_D, ws + t.hid[k], dw->ref_w2[k], nullptr, 0, nl, rows, part, t.part_floats, stream, tn, dw->ref_b2[k], nullptr, cpart, t.cpart_floats, &db_done)) return rc;
            if (!db_done)
                if (int rc = launch_db(g, IEF_D, IEF_D, dw->ref_b2[k], nullptr, 0, nl, rows, cpart, stream)) return rc; }
        // d a = (-lambda g W2) gated by h > 0  (ReLU backward on the saved activation)
        if (int rc = launch_dx(h, tp ? h->ref_w2st[k] : nullptr, g, IEF_D, h->ref_w2[k], IEF_D, IEF_D, da, nullptr, ws + t.hid[k], nl, rows, stream)) return rc;
        { bool db_done = false;
            if (int rc = launch_dw(da, IEF_D, IEF_D, k == K ? rec->zK : ws + t.z[k], dw->ref_w1[k], nullptr, 0, 1.f, rows, part, t.part_floats, stream, tn, dw->ref_b1[k], nullptr, cpart, t.cpart_floats, &db_done)) return rc;
            if (!db_done)
                if (int rc = launch_db(da, IEF_D, IEF_D, dw->ref_b1[k], nullptr, 0, 1.f, rows, cpart, stream)) return rc; }
        // d z_k = g + d a W1   (in place: every element of g is read by the thread that overwrites it)
        if (int rc = launch_dx(h, tp ? h->ref_w1st[k] : nullptr, da, IEF_D, h->ref_w1[k], IEF_D, IEF_D, g, g, nullptr, 1.f, rows, stream)) return rc;
    }
    // fusion (imf_vad.py:130-144) -> d mu | d logvar of both modalities, stacked
    {
        FusionBwdArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.mu_i = rec->mu[0]; fa.lv_i = rec->lv[0]; fa.mu_e = rec->mu[1]; fa.lv_e = rec->lv[1];
        fa.gz = g;
        fa.d_mu_i = dout->image_mu; fa.d_lv_i = dout->image_logvar; fa.d_mu_e = dout->event_mu; fa.d_lv_e = dout->event_logvar;
        fa.d_n_i = dout->w_i; fa.d_n_e = dout->w_e;
        fa.dh_i = ws + t.dh[0]; fa.dh_e = ws + t.dh[1];
        fa.rows = rows; fa.factor = factor; fa.eps = c.epsilon;
        hipLaunchKernelGGL(iefvad_fusion_bwd_kernel, dim3((rows + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, stream, fa);
        HIP_TRY(hipGetLastError());
    }
    for (int m = 0; m < 2; ++m) {
        const float* dhm = ws + t.dh[m];
        float* gx = ws + t.gx;
        // heads (imf_vad.py:125-128): mu.weight | logvar.weight are stacked [1536, 768] in the handle
        { bool db_done = false;
            if (int rc = launch_dw(dhm, 2 * IEF_D, 2 * IEF_D, ws + t.E[m], dw->mu_w[m], dw->logvar_w[m], IEF_D, 1.f, rows, part, t.part_floats, stream, tn, dw->mu_b[m], dw->logvar_b[m], cpart, t.cpart_floats, &db_done)) return rc;
            if (!db_done)
                if (int rc = launch_db(dhm, 2 * IEF_D, 2 * IEF_D, dw->mu_b[m], dw->logvar_b[m], IEF_D, 1.f, rows, cpart, stream)) return rc; }
        if (int rc = launch_dx(h, tp ? h->head_wst[m] : nullptr, dhm, 2 * IEF_D, h->head_w[m], 2 * IEF_D, IEF_D, gx, nullptr, nullptr, 1.f, rows, stream)) return rc;
        // whitening LayerNorm (imf_vad.py:117,123), then the layers last to first
        auto ln_bwd = [&](const float* x, const float* gamma, float* dgamma, float* dbeta) -> int {
            LnBwdArgs la;
            la.x = x; la.dy = gx; la.g = gamma; la.dx = gx; la.part_g = rpart; la.part_b = rpart + (size_t)nblk * IEF_D; la.rows = rows; la.eps = 1e-5f;
            hipLaunchKernelGGL(iefvad_layernorm_bwd_kernel, dim3(nblk), dim3(256), 0, stream, la);
            ReduceBatch rb;
            rb.add(rpart, (size_t)IEF_D, nblk, (size_t)IEF_D, dgamma, 1.f);
            rb.add(rpart + (size_t)nblk * IEF_D, (size_t)IEF_D, nblk, (size_t)IEF_D, dbeta, 1.f);
            rb.launch(stream);
            HIP_TRY(hipGetLastError());
            return 0;
        };
        if (int rc = ln_bwd(ws + t.x[m][L], h->whiten_w[m], dw->whiten_w[m], dw->whiten_b[m])) return rc;
        for (int l = L - 1; l >= 0; --l) {
            // x_{l+1} = LayerNorm(s_l), s_l = x_l + out_proj(attention(x_l))   (imf_vad.py:115-116)
            if (int rc = ln_bwd(ws + t.s[m][l], h->norm_w[m][l], dw->norm_w[m][l], dw->norm_b[m][l])) return rc;
            // gx = d s_l
            float* datt = ws + t.datt;
            float* dqkv = ws + t.dqkv;
            float* dP = ws + t.dP;
            const float* qkv = ws + t.qkv[m][l];
            const float* P = ws + t.P[m][l];
            // P carries the dropout mask in its sign bits (when a probability or a mask was in force in the forward): dropout(P) is formed
            // from it where it is read (the d S launch / the softmax backward, the A operand of d v)
            const bool signed_p = rec->drop[m][l];
            const float drop_scale = signed_p ? (float)(1.0 / (1.0 - (double)rec->drop_p[m][l])) : 1.0f;
            { bool db_done = false;
            if (int rc = launch_dw(gx, IEF_D, IEF_D, ws + t.att[m][l], dw->out_proj_w[m][l], nullptr, 0, 1.f, rows, part, t.part_floats, stream, tn, dw->out_proj_b[m][l], nullptr, cpart, t.cpart_floats, &db_done)) return rc;
            if (!db_done)
                if (int rc = launch_db(gx, IEF_D, IEF_D, dw->out_proj_b[m][l], nullptr, 0, 1.f, rows, cpart, stream)) return rc; }
            if (int rc = launch_dx(h, tp ? h->out_wst[m][l] : nullptr, gx, IEF_D, h->out_w[m][l], IEF_D, IEF_D, datt, nullptr, nullptr, 1.f, rows, stream)) return rc;
            const long long sQ = (long long)IEF_T * 3 * IEF_D, sP1 = (long long)IEF_H * IEF_T * IEF_T, sP2 = (long long)IEF_T * IEF_T,
                            sA = (long long)IEF_T * IEF_D;
            BgemmArgs a;
            // bf16x6: d S = Pd .* d Pd - P rowsum(Pd .* d Pd) with d Pd = d att v^T in ONE launch (attention_split.h, BWD: the product on the
            // split arithmetic, the softmax backward on its accumulators); after an IEFVAD_TRAIN_ATTN=unfused forward the product stays on
            // the fp32 MFMA kernel with the stand-alone softmax backward (A/B)
            const bool ds_fused = rec->fused;
            if (ds_fused) {
                AttnArgs aa;
                AttnTrainArgs tx;
                memset(&aa, 0, sizeof(aa));
                memset(&tx, 0, sizeof(tx));
                aa.nchunks = B;
                aa.qkv[0] = qkv;
                tx.dO[0] = datt; tx.P[0] = const_cast<float*>(P); tx.drop[0] = signed_p ? 1 : 0; tx.drop_p[0] = rec->drop_p[m][l]; tx.dS[0] = dP;
                tx.dQ[0] = dqkv; tx.q_scale = qscale;
                hipLaunchKernelGGL(iefvad_attention_split_ds_kernel, dim3(IEF_H, 2, B), dim3(256), ATS_LDS_BYTES, stream, aa, tx);
                HIP_TRY(hipGetLastError());
            } else {
            // d Pd = d att v^T
            memset(&a, 0, sizeof(a));
            a.A = datt; a.B = qkv + 2 * IEF_D; a.C = dP;
            a.M = IEF_T; a.N = IEF_T; a.K = IEF_DH; a.lda = IEF_D; a.ldb = 3 * IEF_D; a.ldc = IEF_T;
            a.a1 = sA; a.a2 = IEF_DH; a.b1 = sQ; a.b2 = IEF_DH; a.c1 = sP1; a.c2 = sP2; a.nz2 = IEF_H; a.alpha = 1.f;
            if (int rc = launch_bgemm(a, true, true, B * IEF_H, stream)) return rc;
            }
            // d v = Pd^T d att
            memset(&a, 0, sizeof(a));
            a.A = P; a.B = datt; a.C = dqkv + 2 * IEF_D;
            a.M = IEF_T; a.N = IEF_DH; a.K = IEF_T; a.lda = IEF_T; a.ldb = IEF_D; a.ldc = 3 * IEF_D;
            a.a1 = sP1; a.a2 = sP2; a.b1 = sA; a.b2 = IEF_DH; a.c1 = sQ; a.c2 = IEF_DH; a.nz2 = IEF_H; a.alpha = 1.f;
            if (signed_p) a.a_drop = drop_scale;
            if (int rc = launch_bgemm(a, false, false, B * IEF_H, stream)) return rc;
            // d S = Pd .* d Pd - P rowsum(Pd .* d Pd)
            if (!ds_fused) {
                const long long srows = (long long)B * IEF_H * IEF_T;
                hipLaunchKernelGGL(iefvad_softmax_bwd_kernel, dim3((unsigned)((srows + 3) / 4)), dim3(256), 0, stream, P, drop_scale, dP, srows);
                HIP_TRY(hipGetLastError());
            }
            // d q (before the 1/sqrt(96) scale) = qscale * d S k  (ds_fused: done by the same launch, on the d S in its registers)
            if (!ds_fused) {
            memset(&a, 0, sizeof(a));
            a.A = dP; a.B = qkv + IEF_D; a.C = dqkv;
            a.M = IEF_T; a.N = IEF_DH; a.K = IEF_T; a.lda = IEF_T; a.ldb = 3 * IEF_D; a.ldc = 3 * IEF_D;
            a.a1 = sP1; a.a2 = sP2; a.b1 = sQ; a.b2 = IEF_DH; a.c1 = sQ; a.c2 = IEF_DH; a.nz2 = IEF_H; a.alpha = qscale;
            if (int rc = launch_bgemm(a, true, false, B * IEF_H, stream)) return rc;
            }
            // d k = d S^T q_scaled
            memset(&a, 0, sizeof(a));
            a.A = dP; a.B = qkv; a.C = dqkv + IEF_D;
            a.M = IEF_T; a.N = IEF_DH; a.K = IEF_T; a.lda = IEF_T; a.ldb = 3 * IEF_D; a.ldc = 3 * IEF_D;
            a.a1 = sP1; a.a2 = sP2; a.b1 = sQ; a.b2 = IEF_DH; a.c1 = sQ; a.c2 = IEF_DH; a.nz2 = IEF_H; a.alpha = 1.f;
            if (int rc = launch_bgemm(a, false, false, B * IEF_H, stream)) return rc;
            // in_proj (packed q | k | v, imf_vad.py:69-72)
            { bool db_done = false;
            if (int rc = launch_dw(dqkv, 3 * IEF_D, 3 * IEF_D, l == 0 ? rec->x0[m] : ws + t.x[m][l], dw->in_proj_w[m][l], nullptr, 0, 1.f, rows, part, t.part_floats, stream, tn, dw->in_proj_b[m][l], nullptr, cpart, t.cpart_floats, &db_done)) return rc;
            if (!db_done)
                if (int rc = launch_db(dqkv, 3 * IEF_D, 3 * IEF_D, dw->in_proj_b[m][l], nullptr, 0, 1.f, rows, cpart, stream)) return rc; }
            // d x_l = d s_l (residual) + d qkv W_in; the input features need no gradient
            if (l > 0)
                if (int rc = launch_dx(h, tp ? h->in_wst[m][l] : nullptr, dqkv, 3 * IEF_D, h->in_w[m][l], 3 * IEF_D, IEF_D, gx, gx, nullptr, 1.f, rows, stream)) return rc;
        }
    }
    const_cast<TrainRecord*>(rec)->stamp = 0;      // the scratch tensors have overwritten the saved refinement states
    return 0;
}
