// SURVEY.md 8 rows a12 / a13 (= f-5): the modules of /root/reference/model/layers.py and /root/reference/model/module.py that the
// north star names.  Upstream they are residue of a deleted model/VADCLIP.py -- nothing imports them, a checkpoint has no key for
// them -- so they are built at MODULE level: every class of the two files is one C entry below, on the library's MFMA GEMM
// (iefvad_bgemm_f32_kernel: fp32 v_mfma_f32_32x32x2_f32, operands with either index contiguous, bias / residual / activation
// epilogue), its LayerNorm kernel and a handful of row kernels (one wavefront per row, wave reductions) for what lies between the
// products.  fp32 throughout, as the reference computes them.
//
//   iefvad_similarity_adj    layers.py:114-163  SimilarityAdj: theta = x W0 (weight0 serves theta AND phi, weight1 is never read, :132-133),
//                                               cosine similarity, F.threshold(0.7, 0), row softmax (inside [:len, :len] with seq_len)
//   iefvad_distance_adj      layers.py:166-179  DistanceAdj: exp(-|i - j| / e); the sigma parameter is never read
//   iefvad_gcn_forward       layers.py:64-111   GraphConvolution: adj (x W) (+ bias) + residual (identity, or Conv1d(k = 5, pad = 2) over
//                                               time when the widths differ), optionally the QuickGELU VadCLIP applies behind it
//   iefvad_gat_forward       layers.py:12-49    GraphAttentionLayer (eval): h = x W, e_ij = LeakyReLU(a [h_i | h_j]), masked by adj > 0
//                                               with -9e15, row softmax, h' = att h, ELU when concat
//   iefvad_resblock_forward  module.py:20-43    ResidualAttentionBlock, sequence-first [T, B, 768]: x + MHA(ln_1 x) with the block's additive
//                                               attn_mask and the key padding mask, then x + c_proj(QuickGELU(c_fc(ln_2 x)))
//                                               (module.py:46-54 Transformer = a chain of these: the Python mirror loops over them)
// Included by iefvad.hip behind launch_bgemm (train.h).
#pragma once

// ---- row kernels --------------------------------------------------------------------------------------------------------------------
// norms[r] = || x[r, :] ||_2   (torch.norm(theta, p=2, dim=2), layers.py:136-137)
__global__ __launch_bounds__(256) void iefvad_rownorm_kernel(const float* x, float* norms, int rows, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (size_t)row * d;
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *(const f32x4*)(p + c);
        s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    s = wave_sum(s);
    if (lane == 0) norms[row] = sqrtf(s);
}

// adjacency rows in place: sim[b, i, :] -> softmax_j( threshold( sim_ij / (n_i n_j + 1e-20), 0.7 ) ) over j < len; rows / columns
// beyond len are zero (layers.py:139-159: `output` starts as zeros and only [:len, :len] is assigned)
__global__ __launch_bounds__(256) void iefvad_simadj_kernel(float* sim, const float* norms, const int* seq_len, int B, int T) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long long)B * T) return;
    const int b = (int)(row / T), i = (int)(row - (long long)b * T);
    const int len = seq_len ? (seq_len[b] < T ? seq_len[b] : T) : T;
    float* p = sim + row * T;
    if (i >= len) {
        for (int j = lane; j < T; j += 64) p[j] = 0.f;
        return;
    }
    const float ni = norms[(size_t)b * T + i];
    const float* nb = norms + (size_t)b * T;
    float mx = -__builtin_inff();
    for (int j = lane; j < len; j += 64) {
        float v = p[j] / (ni * nb[j] + 1e-20f);
        v = v > 0.7f ? v : 0.f;                               // F.threshold(x, 0.7, 0)
        p[j] = v;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < len; j += 64) {
        const float e = expf(p[j] - mx);
        p[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < T; j += 64) p[j] = j < len ? p[j] / sum : 0.f;
}

// out[b, i, j] = exp(-|i - j| / e)   (layers.py:172-178; torch.exp(torch.tensor(1.)) is the fp32 e)
__global__ __launch_bounds__(256) void iefvad_distadj_kernel(float* out, int B, int T) {
    const float e1 = expf(1.0f);
    const size_t n = (size_t)B * T * T;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        const int j = (int)(idx % T), i = (int)((idx / T) % T);
        out[idx] = expf(-(float)(i > j ? i - j : j - i) / e1);
    }
}

// s1[r] = h[r, :] . a[0:F], s2[r] = h[r, :] . a[F:2F]   (the two halves of matmul([h_i | h_j], a), layers.py:32-33)
__global__ __launch_bounds__(256) void iefvad_gat_scores_kernel(const float* h, const float* a, float* s1, float* s2, int rows, int F) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = h + (size_t)row * F;
    float x = 0.f, y = 0.f;
    for (int c = lane; c < F; c += 64) { x += p[c] * a[c]; y += p[c] * a[F + c]; }
    x = wave_sum(x);
    y = wave_sum(y);
    if (lane == 0) { s1[row] = x; s2[row] = y; }
}

// att[i, :] = softmax_j( adj_ij > 0 ? LeakyReLU(s1_i + s2_j) : -9e15 )   (layers.py:33-37)
__global__ __launch_bounds__(256) void iefvad_gat_attention_kernel(const float* s1, const float* s2, const float* adj, float* att, int N, float slope) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    const float si = s1[i];
    const float* ar = adj + (size_t)i * N;
    float* o = att + (size_t)i * N;
    float mx = -__builtin_inff();
    for (int j = lane; j < N; j += 64) {
        float e = si + s2[j];
        e = e > 0.f ? e : slope * e;
        e = ar[j] > 0.f ? e : -9e15f;
        o[j] = e;
        mx = fmaxf(mx, e);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float e = expf(o[j] - mx);
        o[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < N; j += 64) o[j] = o[j] / sum;
}

// Conv1d weight [O, C, 5] (torch layout) -> five [O, C] tap matrices, so that every tap is an ordinary k-contiguous B operand
__global__ __launch_bounds__(256) void iefvad_conv_taps_kernel(const float* w, float* taps, int O, int C) {
    const size_t n = (size_t)O * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256)
#pragma unroll
        for (int d = 0; d < 5; ++d) taps[(size_t)d * n + idx] = w[idx * 5 + d];
}

// xpad[b, 2 + t, :] = x[b, t, :], two zero rows in front of and behind every sequence (padding = 2 of the k = 5 convolution)
__global__ __launch_bounds__(256) void iefvad_pad_time_kernel(const float* x, float* xpad, int B, int T, int C) {
    const size_t n4 = (size_t)B * (T + 4) * C / 4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * 256) {
        const size_t e = idx * 4, r = e / C, c = e - r * C;
        const int b = (int)(r / (T + 4)), tp = (int)(r - (size_t)b * (T + 4));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tp >= 2 && tp < T + 2) v = *(const f32x4*)(x + ((size_t)b * T + (tp - 2)) * C + c);
        *(f32x4*)(xpad + e) = v;
    }
}

// [A, B, D] -> [B, A, D] (sequence-first <-> batch-first), 16 bytes per thread
__global__ __launch_bounds__(256) void iefvad_swap01_kernel(const float* src, float* dst, int A, int Bn, int D) {
    const size_t n4 = (size_t)A * Bn * D / 4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * 256) {
        const size_t e = idx * 4, r = e / D, c = e - r * D;
        const int a = (int)(r / Bn), b = (int)(r - (size_t)a * Bn);
        *(f32x4*)(dst + ((size_t)b * A + a) * D + c) = *(const f32x4*)(src + e);
    }
}

// P[b, h, i, :] = softmax_j( S_ij * scale + attn_mask[i, j] + (key_padding[b, j] ? -inf : 0) ), in place.  A fully masked row is NaN, as
// torch's softmax of a row of -inf is (nn.MultiheadAttention with need_weights=False, module.py:33-36).
__global__ __launch_bounds__(256) void iefvad_masked_softmax_kernel(float* S, const float* attn_mask, const unsigned char* key_padding, int B, int H, int T,
                                                                   float scale) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long long)B * H * T) return;
    const int i = (int)(row % T), b = (int)(row / ((long long)H * T));
    float* p = S + row * T;
    const float* am = attn_mask ? attn_mask + (size_t)i * T : nullptr;
    const unsigned char* kp = key_padding ? key_padding + (size_t)b * T : nullptr;
    float mx = -__builtin_inff();
    for (int j = lane; j < T; j += 64) {
        float v = p[j] * scale;
        if (am) v += am[j];
        if (kp && kp[j]) v = -__builtin_inff();
        p[j] = v;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) {
        const float e = expf(p[j] - mx);          // mx = -inf (everything masked): -inf - -inf = NaN, as in torch
        p[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < T; j += 64) p[j] = p[j] / sum;
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
static unsigned vc_blocks(size_t n, size_t per_block) { size_t b = (n + per_block - 1) / per_block; return (unsigned)(b > 65535 ? 65535 : (b ? b : 1)); }
static size_t vc_align(size_t floats) { return (floats + 63) & ~(size_t)63; }

// C[M, N] = act(alpha A B^T-or-B + bias + R) through launch_bgemm; `bkc`: B is [N, K] (torch Linear weight) else [K, N]
static int vc_gemm(const float* A, int lda, const float* Bm, int ldb, bool bkc, float* C, int ldc, int M, int N, int K, const float* bias, const float* R,
                   int act, hipStream_t stream, int nbatch = 1, long long a1 = 0, long long b1 = 0, long long c1 = 0, int nz2 = 1, long long a2 = 0,
                   long long b2 = 0, long long c2 = 0, bool akc = true) {
    BgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.B = Bm; a.C = C; a.R = R; a.bias = bias; a.act = act;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.a1 = a1; a.b1 = b1; a.c1 = c1; a.a2 = a2; a.b2 = b2; a.c2 = c2; a.nz2 = nz2; a.alpha = 1.f;
    return launch_bgemm(a, akc, bkc, nbatch, stream);
}

static int vc_dims_ok(const char* who, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 128 || (N % 128 && N % 96) || K % 16)
        return fail("%s: the products run on 128 x (128 | 96) x 16 MFMA tiles; got rows %d, columns %d, contraction %d", who, M, N, K);
    return 0;
}

extern "C" size_t iefvad_similarity_adj_workspace_bytes(int32_t B, int32_t T, int32_t d_out) {
    if (B <= 0 || T <= 0 || d_out <= 0) return 0;
    return (vc_align((size_t)B * T * d_out) + vc_align((size_t)B * T)) * sizeof(float);
}

extern "C" int iefvad_similarity_adj(const float* x, const float* weight0, const int32_t* seq_len, int32_t B, int32_t T, int32_t d_in, int32_t d_out,
                                     float* adj, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!x || !weight0 || !adj || !workspace) return fail("iefvad_similarity_adj: null argument");
    if (B <= 0) return fail("iefvad_similarity_adj: B = %d", B);
    if (int rc = vc_dims_ok("iefvad_similarity_adj", T, d_out, d_in)) return rc;
    if (int rc = vc_dims_ok("iefvad_similarity_adj", T, T, d_out)) return rc;
    if (workspace_bytes < iefvad_similarity_adj_workspace_bytes(B, T, d_out)) return fail("iefvad_similarity_adj: workspace too small");
    if (((uintptr_t)workspace | (uintptr_t)x | (uintptr_t)weight0 | (uintptr_t)adj) & 15) return fail("iefvad_similarity_adj: buffers must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    float* theta = (float*)workspace;
    float* norms = theta + vc_align((size_t)B * T * d_out);
    const int rows = B * T;
    // theta = x W0 ([rows, d_in] x [d_in, d_out]); phi is the same product (layers.py:132-133)
    if (int rc = vc_gemm(x, d_in, weight0, d_out, false, theta, d_out, rows, d_out, d_in, nullptr, nullptr, 0, stream)) return rc;
    hipLaunchKernelGGL(iefvad_rownorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, theta, norms, rows, d_out);
    // sim[b] = theta_b theta_b^T
    if (int rc = vc_gemm(theta, d_out, theta, d_out, true, adj, T, T, T, d_out, nullptr, nullptr, 0, stream, B, (long long)T * d_out, (long long)T * d_out,
                         (long long)T * T))
        return rc;
    hipLaunchKernelGGL(iefvad_simadj_kernel, dim3((unsigned)(((long long)rows + 3) / 4)), dim3(256), 0, stream, adj, norms, (const int*)seq_len, B, T);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int iefvad_distance_adj(int32_t B, int32_t T, float* adj, void* stream_) {
    if (!adj || B <= 0 || T <= 0) return fail("iefvad_distance_adj: bad argument");
    hipLaunchKernelGGL(iefvad_distadj_kernel, dim3(vc_blocks((size_t)B * T * T, 1024)), dim3(256), 0, (hipStream_t)stream_, adj, B, T);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" size_t iefvad_gcn_workspace_bytes(int32_t B, int32_t T, int32_t d_in, int32_t d_out, int32_t residual) {
    if (B <= 0 || T <= 0 || d_in <= 0 || d_out <= 0) return 0;
    size_t f = vc_align((size_t)B * T * d_out);                                                    // support
    if (residual == 2) f += vc_align((size_t)B * (T + 4) * d_in) + vc_align((size_t)5 * d_out * d_in) + vc_align((size_t)B * T * d_out);
    return f * sizeof(float);
}

extern "C" int iefvad_gcn_forward(const float* x, const float* adj, const float* weight, const float* bias, const float* conv_w, const float* conv_b,
                                  int32_t residual, int32_t act, int32_t B, int32_t T, int32_t d_in, int32_t d_out, float* out, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    if (!x || !adj || !weight || !out || !workspace) return fail("iefvad_gcn_forward: null argument");
    if (B <= 0) return fail("iefvad_gcn_forward: B = %d", B);
    if (residual < 0 || residual > 2 || act < 0 || act > 1) return fail("iefvad_gcn_forward: residual %d / act %d", residual, act);
    if (residual == 1 && d_in != d_out) return fail("iefvad_gcn_forward: the identity residual needs d_in == d_out (layers.py:80-81)");
    if (residual == 2 && (!conv_w || !conv_b)) return fail("iefvad_gcn_forward: the Conv1d residual needs its weight [d_out, d_in, 5] and bias");
    if (int rc = vc_dims_ok("iefvad_gcn_forward", T, d_out, d_in)) return rc;
    if (int rc = vc_dims_ok("iefvad_gcn_forward", T, d_out, T)) return rc;
    if (workspace_bytes < iefvad_gcn_workspace_bytes(B, T, d_in, d_out, residual)) return fail("iefvad_gcn_forward: workspace too small");
    if (((uintptr_t)workspace | (uintptr_t)x | (uintptr_t)adj | (uintptr_t)weight | (uintptr_t)out) & 15) return fail("iefvad_gcn_forward: buffers must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const int rows = B * T;
    float* support = (float*)workspace;
    // support = x W   (W is [d_in, d_out], layers.py:94)
    if (int rc = vc_gemm(x, d_in, weight, d_out, false, support, d_out, rows, d_out, d_in, nullptr, nullptr, 0, stream)) return rc;
    const float* R = residual == 1 ? x : nullptr;
    if (residual == 2) {
        // Conv1d(k = 5, padding = 2) over time as five shifted products on a zero-padded copy (layers.py:84, 100-104)
        float* xpad = support + vc_align((size_t)rows * d_out);
        float* taps = xpad + vc_align((size_t)B * (T + 4) * d_in);
        float* res = taps + vc_align((size_t)5 * d_out * d_in);
        hipLaunchKernelGGL(iefvad_pad_time_kernel, dim3(vc_blocks((size_t)B * (T + 4) * d_in / 4, 256)), dim3(256), 0, stream, x, xpad, B, T, d_in);
        hipLaunchKernelGGL(iefvad_conv_taps_kernel, dim3(vc_blocks((size_t)d_out * d_in, 256)), dim3(256), 0, stream, conv_w, taps, d_out, d_in);
        HIP_TRY(hipGetLastError());
        for (int d = 0; d < 5; ++d)
            if (int rc = vc_gemm(xpad + (size_t)d * d_in, d_in, taps + (size_t)d * d_out * d_in, d_in, true, res, d_out, T, d_out, d_in, d == 0 ? conv_b : nullptr,
                                 d == 0 ? nullptr : res, 0, stream, B, (long long)(T + 4) * d_in, 0, (long long)T * d_out))
                return rc;
        R = res;
    }
    // out = adj support (+ bias) (+ residual), per sequence
    return vc_gemm(adj, T, support, d_out, false, out, d_out, T, d_out, T, bias, R, act, stream, B, (long long)T * T, (long long)T * d_out, (long long)T * d_out);
}

extern "C" size_t iefvad_gat_workspace_bytes(int32_t N, int32_t f_out) {
    if (N <= 0 || f_out <= 0) return 0;
    return (vc_align((size_t)N * f_out) + 2 * vc_align((size_t)N) + vc_align((size_t)N * N)) * sizeof(float);
}

extern "C" int iefvad_gat_forward(const float* input, const float* adj, const float* W, const float* a, float alpha, int32_t concat, int32_t N, int32_t f_in,
                                  int32_t f_out, float* out, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!input || !adj || !W || !a || !out || !workspace) return fail("iefvad_gat_forward: null argument");
    if (int rc = vc_dims_ok("iefvad_gat_forward", N, f_out, f_in)) return rc;
    if (int rc = vc_dims_ok("iefvad_gat_forward", N, f_out, N)) return rc;
    if (workspace_bytes < iefvad_gat_workspace_bytes(N, f_out)) return fail("iefvad_gat_forward: workspace too small");
    if (((uintptr_t)workspace | (uintptr_t)input | (uintptr_t)W | (uintptr_t)out) & 15) return fail("iefvad_gat_forward: buffers must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    float* h = (float*)workspace;
    float* s1 = h + vc_align((size_t)N * f_out);
    float* s2 = s1 + vc_align((size_t)N);
    float* att = s2 + vc_align((size_t)N);
    if (int rc = vc_gemm(input, f_in, W, f_out, false, h, f_out, N, f_out, f_in, nullptr, nullptr, 0, stream)) return rc;      // h = input W (layers.py:29)
    hipLaunchKernelGGL(iefvad_gat_scores_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, h, a, s1, s2, N, f_out);
    hipLaunchKernelGGL(iefvad_gat_attention_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, s1, s2, adj, att, N, alpha);
    HIP_TRY(hipGetLastError());
    // h' = attention h; F.elu when concat (layers.py:39-44); dropout is off outside train()
    return vc_gemm(att, N, h, f_out, false, out, f_out, N, f_out, N, nullptr, nullptr, concat ? 2 : 0, stream);
}

extern "C" size_t iefvad_resblock_workspace_bytes(int32_t T, int32_t B, int32_t n_head) {
    if (T <= 0 || B <= 0 || n_head <= 0) return 0;
    const size_t U = vc_align((size_t)B * T * IEF_D);
    return (11 * U + vc_align((size_t)B * n_head * T * T)) * sizeof(float);
}

extern "C" int iefvad_resblock_forward(const float* x, const iefvad_resblock_weights* w, const float* attn_mask, const uint8_t* key_padding_mask, int32_t T,
                                       int32_t B, int32_t d_model, int32_t n_head, float* out, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!x || !w || !out || !workspace) return fail("iefvad_resblock_forward: null argument");
    if (d_model != IEF_D) return fail("iefvad_resblock_forward: d_model = %d (the LayerNorm kernel is built for %d)", d_model, IEF_D);
    if (n_head <= 0 || IEF_D % n_head) return fail("iefvad_resblock_forward: n_head = %d", n_head);
    const int dh = IEF_D / n_head;
    if (B <= 0) return fail("iefvad_resblock_forward: B = %d", B);
    if (int rc = vc_dims_ok("iefvad_resblock_forward", T, T, dh)) return rc;
    if (int rc = vc_dims_ok("iefvad_resblock_forward", T, dh, T)) return rc;
    if (workspace_bytes < iefvad_resblock_workspace_bytes(T, B, n_head)) return fail("iefvad_resblock_forward: workspace too small");
    if (((uintptr_t)workspace | (uintptr_t)x | (uintptr_t)out) & 15) return fail("iefvad_resblock_forward: buffers must be 16-byte aligned");
    const float* const* wp = (const float* const*)w;
    for (int i = 0; i < 12; ++i)
        if (!wp[i]) return fail("iefvad_resblock_forward: weight pointer %d is null", i);
    hipStream_t stream = (hipStream_t)stream_;
    const int rows = B * T, D = IEF_D;
    const size_t U = vc_align((size_t)rows * D);
    float* xb = (float*)workspace;            // batch-first copy of x
    float* y = xb + U;                        // LayerNorm outputs
    float* qkv = y + U;                       // 3 U
    float* att = qkv + 3 * U;
    float* x1 = att + U;
    float* hfc = x1 + U;                      // 4 U
    float* S = hfc + 4 * U;
    const unsigned cp = vc_blocks((size_t)rows * D / 4, 256);
    hipLaunchKernelGGL(iefvad_swap01_kernel, dim3(cp), dim3(256), 0, stream, x, xb, T, B, D);
    auto ln = [&](const float* in, const float* g, const float* b, float* o) {
        LnArgs la;
        memset(&la, 0, sizeof(la));
        la.nrows = rows; la.eps = 1e-5f; la.x[0] = in; la.g1[0] = g; la.b1[0] = b; la.y[0] = o;
        hipLaunchKernelGGL(iefvad_layernorm_kernel, dim3((rows + ROW_WAVES - 1) / ROW_WAVES, 1), dim3(256), 0, stream, la);
    };
    // x = x + attention(ln_1(x))   (module.py:40)
    ln(xb, w->ln_1_w, w->ln_1_b, y);
    HIP_TRY(hipGetLastError());
    if (int rc = vc_gemm(y, D, w->in_proj_w, D, true, qkv, 3 * D, rows, 3 * D, D, w->in_proj_b, nullptr, 0, stream)) return rc;
    const long long sQ = (long long)T * 3 * D, sS1 = (long long)n_head * T * T, sS2 = (long long)T * T;
    if (int rc = vc_gemm(qkv, 3 * D, qkv + D, 3 * D, true, S, T, T, T, dh, nullptr, nullptr, 0, stream, B * n_head, sQ, sQ, sS1, n_head, dh, dh, sS2)) return rc;
    hipLaunchKernelGGL(iefvad_masked_softmax_kernel, dim3((unsigned)(((long long)B * n_head * T + 3) / 4)), dim3(256), 0, stream, S, attn_mask,
                       (const unsigned char*)key_padding_mask, B, n_head, T, 1.0f / sqrtf((float)dh));
    HIP_TRY(hipGetLastError());
    if (int rc = vc_gemm(S, T, qkv + 2 * D, 3 * D, false, att, D, T, dh, T, nullptr, nullptr, 0, stream, B * n_head, sS1, sQ, (long long)T * D, n_head, sS2, dh, dh))
        return rc;
    if (int rc = vc_gemm(att, D, w->out_proj_w, D, true, x1, D, rows, D, D, w->out_proj_b, xb, 0, stream)) return rc;
    // x = x + c_proj(QuickGELU(c_fc(ln_2(x))))   (module.py:41)
    ln(x1, w->ln_2_w, w->ln_2_b, y);
    HIP_TRY(hipGetLastError());
    if (int rc = vc_gemm(y, D, w->c_fc_w, D, true, hfc, 4 * D, rows, 4 * D, D, w->c_fc_b, nullptr, 1, stream)) return rc;
    if (int rc = vc_gemm(hfc, 4 * D, w->c_proj_w, 4 * D, true, xb, D, rows, D, 4 * D, w->c_proj_b, x1, 0, stream)) return rc;
    hipLaunchKernelGGL(iefvad_swap01_kernel, dim3(cp), dim3(256), 0, stream, xb, out, B, T, D);
    HIP_TRY(hipGetLastError());
    return 0;
}
