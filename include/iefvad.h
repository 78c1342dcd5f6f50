/*
 * iefvad.h -- C ABI of libiefvad.so: the MI355X (gfx950) implementation of IEF-VAD's
 * uncertainty-weighted image-event fusion inference forward.
 *
 * The reference has NO native/FFI/plugin boundary for this path: the boundary is a Python
 * nn.Module call, `outputs = model(img, ev, padding_mask, text, lengths)`
 * (/root/reference/test.py:111-117, train/ucf_test.py:104-110, train/xd_test.py:98-104,
 * test2.py:62,79), implemented by `MMFMIL.forward` -> `MultiModal_Fusion_Attn_Iter.forward`
 * (/root/reference/model/imf_vad.py:40-44, :109-161).  The entry points below are what a
 * Python `MMFMIL` shim binds with ctypes to replace that forward; INTEGRATION.md shows the
 * binding.  Plain C types only: pointers are DEVICE pointers (HIP), sizes are in elements
 * or bytes as stated, the stream is a `hipStream_t` passed as `void*`.
 *
 * Ownership: the caller allocates and owns every input, output and workspace buffer.  The
 * library owns the handle and its repacked copies of the weights.
 * Errors: every int-returning function returns 0 on success, non-zero on failure, and never
 * throws or aborts across the ABI; `iefvad_last_error()` returns a thread-local message.
 * Threading: a handle is bound to the device current at `iefvad_create` and is not
 * re-entrant; kernels are enqueued on the caller's stream and the call returns without
 * synchronising.
 */
#ifndef IEFVAD_H
#define IEFVAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IEFVAD_ABI_VERSION 8
#define IEFVAD_MAX_LAYERS 8   /* args.visual_layers (reference default 2, parser.py:5)            */
#define IEFVAD_MAX_STEPS 64   /* args.num_refinement_steps (reference default 10, test.py:406)    */

/* noise_model of MultiModal_Fusion_Attn_Iter (/root/reference/model/imf_vad.py:130-138) */
enum { IEFVAD_NOISE_GAUSSIAN = 0, IEFVAD_NOISE_STUDENT_T = 1 };
/* element type of the img / ev feature blocks handed to iefvad_forward (the reference casts
 * whatever arrives with `.to(torch.float)`, imf_vad.py:41-42) */
enum { IEFVAD_IN_F32 = 0, IEFVAD_IN_F16 = 1, IEFVAD_IN_BF16 = 2 };
/* arithmetic of the dense projections:
 *   F32    = fp32 MFMA (v_mfma_f32_32x32x2_f32), bit-reproducible across batch sizes (parity mode);
 *   BF16   = operands rounded to bf16, fp32 accumulation and fp32 fusion state (throughput mode);
 *   BF16X6 = fp32 operands, each split exactly into three bf16 terms and multiplied as six bf16 MFMA
 *            products with fp32 accumulation: fp32-accurate (held to the F32 mode's tolerances, product
 *            error <= 2^-23 relative worst case, ~2^-27 typical) at the bf16 matrix-core rate; everything else is the F32 path.
 *            Small batches (grids that would not fill the chip) use the F32 kernels, so results are
 *            fp32-accurate but not bit-identical across batch sizes in this mode.
 *   FP16X3 = opt-in, near-fp32: the BF16X6 data flow with two fp16 terms per operand and three products per
 *            multiply-add (22-bit products, half the MFMAs).  Operands are scaled by powers of two from running
 *            max |.| words that the producing kernels maintain, so the result is range-safe; error against fp64 is
 *            ~1.8x the fp32 MFMA path's (rms) per projection, below it end to end. */
enum { IEFVAD_COMPUTE_F32 = 0, IEFVAD_COMPUTE_BF16 = 1, IEFVAD_COMPUTE_BF16X6 = 2, IEFVAD_COMPUTE_FP16X3 = 3 };

typedef struct iefvad_handle iefvad_handle;

/* Replaces the constructor arguments that reach the computation:
 * MMFMIL.__init__ -> MultiModal_Fusion_Attn_Iter(embed_dim, num_layers, num_heads,
 * num_refinement_steps, lambda_ref, noise_model, nu) (/root/reference/model/imf_vad.py:30-38),
 * plus epsilon (always 1e-8 in the reference, :58) and visual_length T. */
typedef struct iefvad_config {
    int32_t abi_version;   /* IEFVAD_ABI_VERSION */
    int32_t embed_dim;     /* D, must be 768 */
    int32_t seq_len;       /* T, must be 256 */
    int32_t num_heads;     /* H, must be 8 (head dim 96) */
    int32_t num_layers;    /* L, 1..IEFVAD_MAX_LAYERS */
    int32_t num_steps;     /* K, 0..IEFVAD_MAX_STEPS */
    int32_t noise_model;   /* IEFVAD_NOISE_* */
    int32_t compute;       /* IEFVAD_COMPUTE_* */
    float lambda_ref;
    float nu;
    float epsilon;
    int32_t micro_batch;   /* chunks processed per internal pass; 0 = library default */
    int32_t graph_chunks;  /* calls with B <= graph_chunks replay a cached hipGraph of the forward (the reference's
                              per-video pattern, /root/reference/test.py:76-117: B = 1 .. a few chunks, 31 launches of a
                              few microseconds each, host-bound when launched one by one); 0 = library default (8),
                              negative = never.  Results are bit-identical to the direct launches. */
} iefvad_config;

/* Device pointers to the reference's state_dict tensors, fp32, laid out
 * as torch stores them: Linear.weight is [out, in] row-major.  Index 0 = image, 1 = event.
 * Replaces `model.load_state_dict(...)` (/root/reference/test.py:377-378). */
typedef struct iefvad_weights {
    const float* in_proj_w[2][IEFVAD_MAX_LAYERS];   /* {m}_attn_layers.l.in_proj_weight [3D, D] */
    const float* in_proj_b[2][IEFVAD_MAX_LAYERS];   /* ....in_proj_bias   [3D] */
    const float* out_proj_w[2][IEFVAD_MAX_LAYERS];  /* ....out_proj.weight [D, D] */
    const float* out_proj_b[2][IEFVAD_MAX_LAYERS];  /* ....out_proj.bias   [D] */
    const float* norm_w[2][IEFVAD_MAX_LAYERS];      /* {m}_norms.l.weight [D] */
    const float* norm_b[2][IEFVAD_MAX_LAYERS];      /* {m}_norms.l.bias   [D] */
    const float* whiten_w[2];                       /* whiten_{m}.weight [D] */
    const float* whiten_b[2];                       /* whiten_{m}.bias   [D] */
    const float* mu_w[2];                           /* {m}_mu.weight [D, D] */
    const float* mu_b[2];                           /* {m}_mu.bias   [D] */
    const float* logvar_w[2];                       /* {m}_logvar.weight [D, D] */
    const float* logvar_b[2];                       /* {m}_logvar.bias   [D] */
    const float* ref_w1[IEFVAD_MAX_STEPS];          /* refinement_blocks.k.0.weight [D, D] */
    const float* ref_b1[IEFVAD_MAX_STEPS];          /* refinement_blocks.k.0.bias   [D] */
    const float* ref_w2[IEFVAD_MAX_STEPS];          /* refinement_blocks.k.2.weight [D, D] */
    const float* ref_b2[IEFVAD_MAX_STEPS];          /* refinement_blocks.k.2.bias   [D] */
    const float* cls_w;                             /* classifier.weight [1, D] */
    const float* cls_b;                             /* classifier.bias   [1] */
} iefvad_weights;

/* Device output buffers; every pointer is optional (NULL = not materialised).  The eight
 * dict entries of the reference's return value (/root/reference/model/imf_vad.py:152-161),
 * all fp32, row-major [B*T, D] (logits: [B*T]), plus the per-row means over D of w_i / w_e
 * that the harness derives on the host (/root/reference/test.py:131-136). */
typedef struct iefvad_outputs {
    float* fused;         /* [B*T, D] */
    float* logits;        /* [B*T]    */
    float* image_mu;      /* [B*T, D] */
    float* event_mu;      /* [B*T, D] */
    float* image_logvar;  /* [B*T, D] */
    float* event_logvar;  /* [B*T, D] */
    float* w_i;           /* [B*T, D]  normalised image weight */
    float* w_e;           /* [B*T, D]  normalised event weight */
    float* w_i_mean;      /* [B*T]     mean over D of w_i */
    float* w_e_mean;      /* [B*T]     mean over D of w_e */
} iefvad_outputs;

/* Per-stage device time of the last iefvad_forward_timed call, milliseconds (hipEvents on the
 * caller's stream).  Measurement aid for bench.py's roofline block. */
typedef struct iefvad_stage_times {
    float total_ms;
    float qkv_gemm_ms;
    float attention_ms;
    float out_gemm_ms;
    float layernorm_ms;
    float head_gemm_ms;
    float fusion_ms;
    float refine_gemm_ms;
    float scorer_ms;
    float cast_ms;           /* input casts and bf16 operand copies */
    int32_t gemm_launches;   /* number of dense-projection GEMM launches in the pass */
} iefvad_stage_times;

int iefvad_abi_version(void);

/* Build a handle on the current HIP device.  Unsupported dimensions or noise_model -> error. */
int iefvad_create(const iefvad_config* cfg, iefvad_handle** out);

/* Copy/repack the weights into library-owned device memory (stream-ordered on `stream`). */
int iefvad_set_weights(iefvad_handle* h, const iefvad_weights* w, void* stream);

/* Bytes of caller-provided scratch that iefvad_forward needs for a call with B chunks. */
size_t iefvad_workspace_bytes(const iefvad_handle* h, int32_t B);

/* The forward: img, ev are device pointers to [B, T, D] blocks of `in_dtype`, contiguous.
 * Enqueues on `stream` (hipStream_t) and returns without synchronising. */
int iefvad_forward(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B,
                   void* workspace, size_t workspace_bytes, const iefvad_outputs* out, void* stream);

/* Same forward, bracketing each stage with hipEvents; synchronises the stream before returning. */
int iefvad_forward_timed(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B,
                         void* workspace, size_t workspace_bytes, const iefvad_outputs* out, void* stream,
                         iefvad_stage_times* times);

/* ---- whole videos --------------------------------------------------------------------------------------------------
 * The hot loop of the reference's evaluation, one level up from the forward: for every video the loader pads the [len, D]
 * feature files to whole T = 256 chunks on the host (process_split, /root/reference/data/tools.py:100-114; dataset.py:34-52),
 * test() scans the padded tensors for NaN and replaces them (`if torch.isnan(x).any(): x = torch.nan_to_num(x, nan=0.0)`,
 * /root/reference/test.py:90-95), runs the model on every chunk row and keeps `logits.reshape(-1)[0:len]`
 * (test.py:119-121) and the first len row means of w_i / w_e (test.py:131-138).  iefvad_forward_videos does exactly that
 * for a batch of videos with only the VALID rows crossing the boundary:
 *   img_rows, ev_rows   device, [sum(lengths), D] of `in_dtype`: the videos' feature rows concatenated in list order (no padding)
 *   lengths             HOST array of nvideos snippet counts (>= 1)
 *   nan_to_num          non-zero: the conditional replacement above, decided per WHOLE video and per modality on the device: every
 *                       chunk of the call is scanned before the first micro-batch pass lays out its rows, so a video that
 *                       straddles passes is treated as the reference treats its one tensor
 *                       (NaN -> 0, +-inf -> the largest / smallest finite value of `in_dtype`); zero: rows are used as they are
 *   logits, w_i_mean, w_e_mean   device, [sum(lengths)] fp32 each, the means nullable: per-snippet results in the same order
 * Chunks are laid out on the device (zero padded; the all-zero chunk that process_split appends to a len % 256 == 0 video,
 * whose rows test.py:121 slices away, is not built).  Attention is unmasked over the padded window, as in the reference,
 * but the pad rows of a window are identical in every layer: the row set holds ONE of them per chunk, the row-wise stages
 * (projections, LayerNorms and everything behind the encoder, imf_vad.py:125-150) run on valid + 1 rows per chunk, and the
 * attention kernels read row min(r, valid) for row r of the window -- every product and sum of the padded computation.
 * Results equal those of iefvad_forward on the host-padded chunks (bit for bit in the F32 and BF16 modes).  FP16X3 (one
 * operand scale per whole chunk) and IEFVAD_DENSE_ENCODER=1 (environment, read at iefvad_create) keep whole chunks in the
 * encoder and gather the valid rows behind it.
 * Workspace: iefvad_videos_workspace_bytes.  Enqueued on `stream`; `lengths` is consumed before the call returns. */
size_t iefvad_videos_workspace_bytes(const iefvad_handle* h, const int32_t* lengths, int32_t nvideos);
int iefvad_forward_videos(iefvad_handle* h, const void* img_rows, const void* ev_rows, int32_t in_dtype,
                          const int32_t* lengths, int32_t nvideos, int32_t nan_to_num, void* workspace,
                          size_t workspace_bytes, float* logits, float* w_i_mean, float* w_e_mean, void* stream);

/* The whole evaluation LIST in one call: host rows in, per-snippet results on the device.  Replaces the Python loop around
 * iefvad_forward_videos (one iteration per video in the reference, /root/reference/test.py:76-121; one per packed batch in this
 * package until round 3): the library cuts the list into passes of >= batch_chunks chunks (whole videos; 0 = 128), a worker thread
 * gathers the rows of the next passes into a ring of four pinned staging slots (host_threads copy threads, 0 = 8, at most 16) while
 * this thread sends pass k on an internal copy stream and enqueues its forward on one of TWO internal non-blocking compute streams
 * (alternating; both wait for what `stream` held at entry, and `stream` waits for both before the call returns); results land in
 * list order.
 *   img_rows, ev_rows   HOST arrays of nvideos HOST pointers: video v's [lengths[v], D] feature rows of `in_dtype`, contiguous
 *                       (e.g. the first lengths[v] rows of the zero-padded tensor a DataLoader delivers)
 *   wire_dtype          the element type the rows cross PCIe in: `in_dtype` (the rows as they are), or IEFVAD_IN_BF16 for F32 rows
 *                       on a BF16-mode handle -- the throughput mode's down-conversion on the wire: the gather
 *                       threads round to bf16 (nearest even, as the device would) while they stage, half the bytes are copied, and the
 *                       forward runs as iefvad_forward_videos does on IEFVAD_IN_BF16 rows.  The bf16 mode rounds these rows for its
 *                       first projection anyway; what changes is the first layer's residual, which reads the rounded row too
 *                       (and nan_to_num's +-inf replacement is the bf16 extreme).  Other combinations are refused.
 *   logits, w_i_mean, w_e_mean   DEVICE, [sum(lengths)] fp32 each (the means nullable), valid once `stream` has run
 * Returns when every pass has been enqueued and every host row has been read (the caller's host tensors are free again);
 * staging slots, device input slots and the workspace belong to the handle.  Same results as iefvad_forward_videos on the
 * same batches. */
int iefvad_forward_videos_host(iefvad_handle* h, const void* const* img_rows, const void* const* ev_rows, int32_t in_dtype,
                               int32_t wire_dtype, const int32_t* lengths, int32_t nvideos, int32_t nan_to_num, int32_t batch_chunks,
                               int32_t host_threads, float* logits, float* w_i_mean, float* w_e_mean, void* stream);

/* ---- training-side loss head: forward, and its gradients w.r.t. the model's outputs -----------------
 * The three terms the reference's trainers add up (/root/reference/train/ucf_train.py:68-101, train/xd_train.py:60-75), as
 * device reductions over tensors iefvad_forward already produces:
 *   out[0] classification  CLAS2(logits, labels, lengths) (/root/reference/train/loss.py:18-30): per video the mean of the
 *                          int(len / 16 + 1) largest sigmoid(logit[0:len]), binary cross entropy against targets
 *   out[1] reg = out[2] + out[3]: mean(1 - cosine_similarity(normalize(mu_i), normalize(mu_e))) + mean(| ||mu_i|| - ||mu_e|| |)
 *   out[4] kl = out[5] + out[6]: -0.5 mean(1 + l - mu^2 - exp(l)) per modality, l = logvar (Gaussian) or
 *                          logvar + log(nu / (nu + 1)) (StudentT; the trainers read model.temporal.nu, ucf_train.py:94-95)
 *   out[7] total = classification + lambda_reg * reg + lambda_kl * kl   (1, 1 in ucf_train.py:100-102; 0.01, 0.01 in xd_train.py:73-75)
 * logits [B, T]; the four 768-d tensors [B*T, 768]; lengths int32 [B] and targets fp32 [B] (1 = abnormal, i.e.
 * 1 - labels[:, 0], loss.py:20) on the DEVICE; out: 8 fp32 on the device.  T must be 256.  Deterministic (no atomics).
 * `workspace`: iefvad_loss_workspace_bytes(B, T) device bytes, 16-byte aligned.
 * The four 768-d pointers may all be NULL: then only out[0] (and out[7] = out[0]) is computed -- CLAS2 alone.
 * iefvad_loss_backward: the gradients of grad_scale * total with respect to logits [B, T] and the four 768-d tensors (what
 * `loss.backward()` hands to the model's outputs in /root/reference/train/ucf_train.py:103): CLAS2 through torch's BCE
 * backward ((p - y) / max((1 - p) p, 1e-12)) and the top-k selection, the cosine / norm regulariser, the KL terms.
 * `grad_scale_dev` (nullable) is the upstream gradient as a DEVICE scalar that multiplies grad_scale inside the kernels, so an
 * autograd node never has to read it back to the host.  Each
 * d_* pointer may be NULL (that gradient is not written).  The model's own backward pass: iefvad_train_backward below. */
size_t iefvad_loss_workspace_bytes(int32_t B, int32_t T);
int iefvad_loss_forward(const float* logits, const float* image_mu, const float* event_mu, const float* image_logvar,
                        const float* event_logvar, const int32_t* lengths, const float* targets, int32_t B, int32_t T,
                        int32_t noise_model, float nu, float lambda_reg, float lambda_kl, float* out, void* workspace,
                        size_t workspace_bytes, void* stream);
int iefvad_loss_backward(const float* logits, const float* image_mu, const float* event_mu, const float* image_logvar,
                         const float* event_logvar, const int32_t* lengths, const float* targets, int32_t B, int32_t T,
                         int32_t noise_model, float nu, float lambda_reg, float lambda_kl, float grad_scale,
                         float* d_logits, float* d_image_mu, float* d_event_mu, float* d_image_logvar,
                         float* d_event_logvar, const float* grad_scale_dev, void* stream);

/* ---- training: the train-mode forward and the model's backward pass ----------------------------------------------
 * What `model.train(); outputs = model(...); loss.backward()` does to the model in the reference's trainers
 * (/root/reference/train/ucf_train.py:43,60-66,103; train/xd_train.py:37,46-52,78): the forward of
 * /root/reference/model/imf_vad.py:109-161 with the attention dropout of nn.MultiheadAttention(dropout=0.1) (imf_vad.py:70) active,
 * and the gradient of every parameter of SURVEY.md Appendix B.  The reference has no backward code of its own (autograd derives it);
 * the entry points below are what a torch.autograd.Function around the model binds.
 *
 * iefvad_train_forward computes the eight outputs like iefvad_forward and keeps what the backward needs in `train_ws`
 * (iefvad_train_workspace_bytes; 256-byte aligned; owned by the caller, who keeps it untouched until the matching
 * iefvad_train_backward has run on the same stream).  compute must be IEFVAD_COMPUTE_F32 or IEFVAD_COMPUTE_BF16X6 (fp32-accurate);
 * the whole batch runs as one pass (B <= 4096).  Five of the outputs are also saved tensors (fused, image_mu, event_mu, image_logvar,
 * event_logvar) and fp32 inputs are saved as they are: the forward writes / reads the caller's buffers in place and the backward
 * READS THEM AGAIN, so `img`, `ev` (in_dtype f32) and those five output buffers stay untouched until iefvad_train_backward has run
 * (a null output pointer keeps that tensor in train_ws).  ONE backward per forward: the backward's scratch tensors overwrite saved states
 * it has already differentiated, so a second iefvad_train_backward on the same buffer fails ("no iefvad_train_forward ...") until a
 * new forward has filled it.
 *   dropout_p[m][l]  the probability nn.MultiheadAttention m / layer l drops an attention weight with (0 = none); kept weights
 *                    are scaled by 1 / (1 - p) as torch.nn.functional.dropout does
 *   seed             stream of the library's counter-based mask generator for this step.  torch draws its mask from its own Philox
 *                    stream, so a p > 0 run is statistically, not bit-wise, the reference's; p = 0 and injected masks are exact
 *   keep_mask        device, nullable: uint8 [2][L][B][8][T][T], 1 = keep -- an injected mask replaces the generator (tests, replay)
 * iefvad_train_backward: `dout` holds the gradients handed to the eight outputs (each nullable = zero; [B*T, D], logits [B*T]);
 * every non-null pointer of `dw` ([same shape as the parameter], fp32) receives that parameter's gradient (overwritten, not
 * accumulated).  Reductions over rows are fixed-order: the same inputs give the same bits on every run. */
typedef struct iefvad_train_options {
    float dropout_p[2][IEFVAD_MAX_LAYERS];
    uint64_t seed;
    const uint8_t* keep_mask;
} iefvad_train_options;

typedef struct iefvad_output_grads {
    const float* fused;  const float* logits;  const float* image_mu;  const float* event_mu;
    const float* image_logvar;  const float* event_logvar;  const float* w_i;  const float* w_e;
} iefvad_output_grads;

/* gradient destinations, field for field the layout of iefvad_weights */
typedef struct iefvad_weight_grads {
    float* in_proj_w[2][IEFVAD_MAX_LAYERS];
    float* in_proj_b[2][IEFVAD_MAX_LAYERS];
    float* out_proj_w[2][IEFVAD_MAX_LAYERS];
    float* out_proj_b[2][IEFVAD_MAX_LAYERS];
    float* norm_w[2][IEFVAD_MAX_LAYERS];
    float* norm_b[2][IEFVAD_MAX_LAYERS];
    float* whiten_w[2];
    float* whiten_b[2];
    float* mu_w[2];
    float* mu_b[2];
    float* logvar_w[2];
    float* logvar_b[2];
    float* ref_w1[IEFVAD_MAX_STEPS];
    float* ref_b1[IEFVAD_MAX_STEPS];
    float* ref_w2[IEFVAD_MAX_STEPS];
    float* ref_b2[IEFVAD_MAX_STEPS];
    float* cls_w;
    float* cls_b;
} iefvad_weight_grads;

size_t iefvad_train_workspace_bytes(const iefvad_handle* h, int32_t B);
int iefvad_train_forward(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B,
                         const iefvad_train_options* opt, void* train_ws, size_t train_ws_bytes, const iefvad_outputs* out,
                         void* stream);
int iefvad_train_backward(iefvad_handle* h, int32_t B, void* train_ws, size_t train_ws_bytes, const iefvad_output_grads* dout,
                          const iefvad_weight_grads* dw, void* stream);

/* The trainers' input rule (/root/reference/train/ucf_train.py:50-53, xd_train.py:40-43: `if torch.isnan(x).any(): x =
 * torch.nan_to_num(x, nan=0.0)`, per tensor) for the two fp32 input tensors of a step, without the host read of the flag: one scan,
 * one repair launch that returns at once for a tensor without NaN; a tensor WITH a NaN is rewritten IN PLACE as torch.nan_to_num
 * does (NaN -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX).  a, b: device, n floats each (n % 4 == 0, 16-byte aligned); flags_ws: 8 device
 * bytes (8-byte aligned) that hold the two flags afterwards (1 = that tensor had a NaN). */
int iefvad_nan_rule(float* a, float* b, size_t n, void* flags_ws, void* stream);

/* One torch.optim.AdamW step (as /root/reference/train/ucf_train.py:28 constructs it: betas (0.9, 0.999), eps 1e-8, weight_decay
 * 0.01, amsgrad off) on one flat fp32 tensor, in place, in torch's operation order; `step` counts from 1.  The hyper-parameters
 * travel as doubles: torch forms 1 - lr wd, 1 - beta1, 1 - beta2, lr / (1 - beta1^t) and sqrt(1 - beta2^t) in Python floats
 * and rounds each to fp32 once; so does this.  All pointers are device pointers of n floats. */
int iefvad_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                      double beta2, double eps, double weight_decay, int32_t step, void* stream);

/* The same update for many tensors in ONE launch (a model of SURVEY.md Appendix B has 78 parameter tensors, most of them 768-element
 * vectors): `table` is a DEVICE array of `count` entries, each with its four device pointers, its element count and the index of
 * its first 4096-element chunk (first_chunk[0] = 0, first_chunk[i+1] = first_chunk[i] + ceil(n[i] / 4096)); total_chunks = the sum.
 * All tensors take the same hyper-parameters and step count.  Element for element the arithmetic of iefvad_adamw_step. */
typedef struct iefvad_adamw_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    uint64_t n;
    uint64_t first_chunk;
} iefvad_adamw_tensor;
int iefvad_adamw_step_multi(const iefvad_adamw_tensor* table, int32_t count, uint64_t total_chunks, double lr, double beta1, double beta2,
                            double eps, double weight_decay, int32_t step, void* stream);

/* Host helper of the whole-video path (the loader side, /root/reference/data/dataset.py:34-52 + test.py:90-95's `.to(device)`):
 * dst[0 ..) = srcs[0] | srcs[1] | ... (nbytes[i] bytes each), copied by up to `threads` host threads.  `dst` is normally a
 * pinned staging buffer that one asynchronous copy then sends to the device as iefvad_forward_videos's img_rows / ev_rows.
 * Pure host code: all pointers are HOST pointers. */
int iefvad_host_gather(void* dst, const void* const* srcs, const size_t* nbytes, int64_t count, int32_t threads);
/* The same gather with the rows narrowed on the way: srcs[i] holds nbytes[i] bytes of fp32 (multiples of 64), dst receives them as
 * bf16 (sum(nbytes) / 2 bytes; round to nearest even, NaN stays NaN with its sign, overflow to +-inf -- the device's conversion).
 * This is what iefvad_forward_videos_host's copy threads run for wire_dtype = IEFVAD_IN_BF16. */
int iefvad_host_gather_bf16(void* dst, const void* const* srcs, const size_t* nbytes, int64_t count, int32_t threads);

/* iefvad_forward with a per-row scale on the INPUT features: row r of modality m enters the model as x[r] * scale_m[r] (the
 * product rounded to the input type, as torch does for fp16 / bf16 tensors, before imf_vad.py:41-42 widens it); a NULL vector
 * scales nothing.  This is the perturbation of the reference's robustness sweep (/root/reference/test2.py:71-77:
 * `v_p[:, indexs] = v_p[:, indexs] * 0.01` on a random subset of time steps) folded into the input load, so the packed clean
 * features can stay resident on the device across the sweep's twelve levels.  img_row_scale, ev_row_scale: DEVICE [B*T] fp32. */
int iefvad_forward_scaled(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B, const float* img_row_scale,
                          const float* ev_row_scale, void* workspace, size_t workspace_bytes, const iefvad_outputs* out, void* stream);

/* ---- metric tail of the evaluation loop --------------------------------------------------------------
 * What /root/reference/test.py:158-159 (train/ucf_test.py:153-154, train/xd_test.py:146-147) computes with sklearn on the host,
 *     ROC1 = roc_auc_score(gt, np.repeat(ap1, 16));   AP1 = average_precision_score(gt, np.repeat(ap1, 16))
 * on the DEVICE from the n per-snippet scores and the frame-level ground truth, without materialising the x`repeat` copy: one
 * radix sort of n (score, positive-frames-of-the-snippet) pairs, a scan, a reduction over the tie groups.  Thresholds are the
 * DISTINCT score values as in sklearn's _binary_clf_curve (tied snippets share one threshold; -0.0 == +0.0); the AUC numerator is
 * accumulated exactly in 64-bit integers, the AP terms in doubles in a fixed order: deterministic, and equal to sklearn to ~1e-15.
 *   scores      DEVICE [n] fp32 (e.g. sigmoid of iefvad_forward_videos_host's logits, or iefvad_gather_scores's output)
 *   gt_frames   DEVICE [n * repeat] bytes, non-zero = anomalous frame (the reference's gt.npy holds 0.0 / 1.0, test.py:379)
 *   auc, ap     DEVICE doubles, each nullable (not both); valid once `stream` has run.  A NaN score makes both NaN; only one class
 *               in gt_frames makes *auc NaN (sklearn raises there) and, with no positive frame, *ap 0 (as sklearn returns)
 *   workspace   DEVICE, iefvad_auc_ap_workspace_bytes(n) bytes (16 n + ~1.1 KB per 4096 snippets), 256-byte aligned
 * n * repeat must stay below 2^32 frames.  Enqueued on `stream`; returns without synchronising. */
size_t iefvad_auc_ap_workspace_bytes(int64_t n);
int iefvad_auc_ap(const float* scores, const uint8_t* gt_frames, int64_t n, int32_t repeat, double* auc, double* ap,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- the VadCLIP-residue modules the north star names (SURVEY.md 8 rows a12 / a13 = f-5) ------------------------------------------
 * /root/reference/model/layers.py and /root/reference/model/module.py are dead code upstream (nothing imports them, a checkpoint holds
 * no key of theirs), so they are exposed at MODULE level: one entry per class, fp32, on the library's MFMA GEMM + LayerNorm kernels
 * (csrc/vadclip.h).  All tensors are DEVICE fp32, row-major, 16-byte aligned; every entry takes a caller-owned workspace of
 * iefvad_*_workspace_bytes(...) bytes and enqueues on `stream`.  The products run on 128 x (128 | 96) x 16 tiles: T (and N for the
 * GAT layer) must be a multiple of 128, feature widths multiples of 128 (or 96) and of 16.
 *
 * iefvad_similarity_adj  layers.py:114-163  x [B,T,d_in], weight0 [d_in,d_out] (weight1 is never read upstream, :132-133),
 *                        seq_len int32 [B] on the device or NULL -> adj [B,T,T]: cosine similarity of x W0 rows, F.threshold(0.7, 0),
 *                        row softmax (over [:len, :len] when seq_len is given; zero outside)
 * iefvad_distance_adj    layers.py:166-179  adj[b,i,j] = exp(-|i - j| / e)
 * iefvad_gcn_forward     layers.py:64-111   out = adj (x W) (+ bias) + residual.  weight [d_in,d_out]; residual 0 none, 1 identity (d_in ==
 *                        d_out), 2 Conv1d(d_in, d_out, kernel_size=5, padding=2) over time with conv_w [d_out,d_in,5], conv_b [d_out];
 *                        act 1 applies QuickGELU to the result (what VadCLIP's caller did), 0 leaves it
 * iefvad_gat_forward     layers.py:12-49    input [N,f_in], adj [N,N], W [f_in,f_out], a [2 f_out]; alpha = LeakyReLU slope;
 *                        concat != 0 applies ELU; eval semantics (dropout inactive)
 * iefvad_resblock_forward module.py:20-43   x, out [T,B,768] sequence-first; attn_mask [T,T] additive fp32 or NULL; key_padding_mask
 *                        [B,T] bytes (non-zero = padding) or NULL; n_head with 768 / n_head in {96, 128} */
typedef struct iefvad_resblock_weights {
    const float* in_proj_w;   /* attn.in_proj_weight [2304,768] */
    const float* in_proj_b;   /* [2304] */
    const float* out_proj_w;  /* attn.out_proj.weight [768,768] */
    const float* out_proj_b;
    const float* ln_1_w; const float* ln_1_b;
    const float* ln_2_w; const float* ln_2_b;
    const float* c_fc_w;      /* mlp.c_fc.weight [3072,768] */
    const float* c_fc_b;
    const float* c_proj_w;    /* mlp.c_proj.weight [768,3072] */
    const float* c_proj_b;
} iefvad_resblock_weights;
size_t iefvad_similarity_adj_workspace_bytes(int32_t B, int32_t T, int32_t d_out);
int iefvad_similarity_adj(const float* x, const float* weight0, const int32_t* seq_len, int32_t B, int32_t T, int32_t d_in, int32_t d_out,
                          float* adj, void* workspace, size_t workspace_bytes, void* stream);
int iefvad_distance_adj(int32_t B, int32_t T, float* adj, void* stream);
size_t iefvad_gcn_workspace_bytes(int32_t B, int32_t T, int32_t d_in, int32_t d_out, int32_t residual);
int iefvad_gcn_forward(const float* x, const float* adj, const float* weight, const float* bias, const float* conv_w, const float* conv_b,
                       int32_t residual, int32_t act, int32_t B, int32_t T, int32_t d_in, int32_t d_out, float* out, void* workspace,
                       size_t workspace_bytes, void* stream);
size_t iefvad_gat_workspace_bytes(int32_t N, int32_t f_out);
int iefvad_gat_forward(const float* input, const float* adj, const float* W, const float* a, float alpha, int32_t concat, int32_t N, int32_t f_in,
                       int32_t f_out, float* out, void* workspace, size_t workspace_bytes, void* stream);
size_t iefvad_resblock_workspace_bytes(int32_t T, int32_t B, int32_t n_head);
int iefvad_resblock_forward(const float* x, const iefvad_resblock_weights* w, const float* attn_mask, const uint8_t* key_padding_mask, int32_t T,
                            int32_t B, int32_t d_model, int32_t n_head, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- unit entry for the bf16 mode's row-block kernels (per-kernel parity tests) -----------------------------------------
 * Launches ONE production kernel of compute = IEFVAD_COMPUTE_BF16 -- the symbol, grid and LDS size iefvad_forward uses at that row
 * count (the persistent variants from two 64-row blocks per workgroup on) -- on caller-supplied DEVICE rows, with the handle's
 * weights.  `rows` is a multiple of 64; both modalities ride one launch as in the forward.  Stages and the fields they read:
 *   IEFVAD_UNIT_INPROJ      layer l: x[m] = the layer's input rows [rows,768] (fp32 for l == 0, bf16 otherwise) ->
 *                           y[m] = bf16 q | k | v, head-major [3][8 heads][rows][96], q pre-scaled by log2(e)/sqrt(96)
 *                           (/root/reference/model/imf_vad.py:115,121: nn.MultiheadAttention's packed in_proj)
 *   IEFVAD_UNIT_OUTPROJ_LN  layer l: x[m] = attention output bf16 [rows,768], resid[m] = the layer's fp32 input rows ->
 *                           LayerNorm(resid + x W_o^T + b_o) (imf_vad.py:116,122), and for l == L-1 the whitening LayerNorm
 *                           behind it (:117,:123): y[m] fp32 and / or yb[m] bf16 [rows,768] (each nullable, not both)
 *   IEFVAD_UNIT_HEADS       x[m] = whitened rows bf16 [rows,768] -> mu[m], logvar[m], w[m] (each nullable) and the fused z
 *                           (imf_vad.py:125-144)
 *   IEFVAD_UNIT_REFINE      x[0] = z_0 fp32 [rows,768] -> z = z_K (nullable), logits [rows] (imf_vad.py:146-150)
 * All pointers 16-byte aligned.  Enqueued on `stream`. */
#define IEFVAD_UNIT_INPROJ 0
#define IEFVAD_UNIT_OUTPROJ_LN 1
#define IEFVAD_UNIT_HEADS 2
#define IEFVAD_UNIT_REFINE 3
typedef struct iefvad_unit_io {
    const void* x[2];
    const float* resid[2];
    void* y[2];
    void* yb[2];
    float* mu[2];
    float* logvar[2];
    float* w[2];
    float* z;
    float* logits;
} iefvad_unit_io;
int iefvad_rowblock_unit(iefvad_handle* h, int32_t stage, int32_t layer, int32_t rows, const iefvad_unit_io* io, void* stream);

/* Stand-alone dense projection C[M,N] = A[M,K] * W[N,K]^T + bias[N] on the library's GEMM
 * kernels (unit tests and the roofline micro-benchmark).  M % 128 == 0, N % 128 == 0, K % 64 == 0.
 * compute = IEFVAD_COMPUTE_F32: A and W are fp32; IEFVAD_COMPUTE_BF16: A and W are bf16 (same shapes);
 * IEFVAD_COMPUTE_BF16X6: A is fp32, W is the three-plane split [3][N][K] bf16 made by iefvad_split_bf16x3
 * (K % 64 == 0); bias and C are fp32 in all three. */
int iefvad_gemm_bias(const void* A, const void* W, const float* bias, float* C,
                     int32_t M, int32_t N, int32_t K, int32_t compute, void* stream);

/* The exact three-term bf16 split of n fp32 values (n % 4 == 0): planes[0..n) = bf16(x),
 * planes[n..2n) = bf16(x - p0), planes[2n..3n) = bf16(x - p0 - p1), round-to-nearest-even; x == p0 + p1 + p2
 * for finite x.  What iefvad_set_weights applies to every projection matrix in IEFVAD_COMPUTE_BF16X6. */
int iefvad_split_bf16x3(const float* src, void* planes, size_t n, void* stream);

/* The same for `count` matrices in ONE launch per 32 of them -- what iefvad_set_weights and the training backward use (a training
 * step re-splits every projection matrix and its transpose: 90 small launches before).  src / planes / n / rows are HOST arrays of
 * device pointers and sizes.  rows[i] == 0: planes[i] receives the split of src[i] as above (n[i] % 4 == 0).  rows[i] == R > 0:
 * src[i] is a row-major [R, n[i] / R] matrix and planes[i] receives the split of its TRANSPOSE [n[i] / R, R] (R % 64 == 0,
 * (n[i] / R) % 32 == 0) -- the operand of dX = dY W as an NT product.  Bit for bit the planes of iefvad_split_bf16x3. */
int iefvad_split_bf16x3_many(const float* const* src, void* const* planes, const size_t* n, const int32_t* rows, int32_t count,
                             void* stream);

/* ---- multi-GPU score gather ------------------------------------------------------------------
 * The reference has no collective at all (/root/reference/main.py:7 imports torch.distributed and never uses it);
 * what it fixes is the ORDER of the score vector: per-video scores are concatenated in test-list order and the
 * ground truth is indexed by the running snippet offset (/root/reference/test.py:123-129,153).  Videos shard across
 * ranks in contiguous ranges, so the rank-order concatenation of the ranks' score vectors is that order, and one
 * RCCL exchange over xGMI closes the evaluation.  librccl.so.1 is bound at run time (dlopen; a copy the process has
 * already loaded -- PyTorch's -- is preferred), so single-GPU users never load it.
 *
 * One process per GPU.  Rank 0 calls iefvad_comm_unique_id and hands the 128 bytes to the other ranks by any
 * host-side channel (the Python harness uses the torch.distributed store); every rank then calls
 * iefvad_comm_create with the device it will gather on current.  */
#define IEFVAD_COMM_ID_BYTES 128
typedef struct iefvad_comm iefvad_comm;

int iefvad_comm_unique_id(void* id_bytes /* host, IEFVAD_COMM_ID_BYTES */);
int iefvad_comm_create(const void* id_bytes, int32_t nranks, int32_t rank, iefvad_comm** out);
/* number of ranks RCCL itself reports for the communicator (ncclCommCount); 0 for a NULL handle */
int32_t iefvad_comm_nranks(const iefvad_comm* c);
void iefvad_comm_destroy(iefvad_comm* c);

/* RCCL's own version number (ncclGetVersion) once librccl is bound, 0 when it cannot be bound in this process.  Lets every
 * rank check -- and agree, over whatever host channel it has -- BEFORE the collective iefvad_comm_create that all ranks
 * will get through it. */
int32_t iefvad_rccl_version(void);

/* gathered[offset(r) .. offset(r) + counts[r]) = rank r's `local[0 .. counts[r])`, for every r, on every rank;
 * offset(r) = counts[0] + ... + counts[r-1].  `local`, `gathered` are device pointers to fp32; `counts` is a HOST
 * array of nranks element counts that every rank passes identically (shards are cut from the shared, ordered test
 * list -- harness.partition_by_snippets -- so no count exchange is needed), or NULL when every rank contributes
 * `count` elements.  `gathered_capacity` = the number of fp32 elements `gathered` can hold; a call whose ranks
 * contribute more fails before anything is enqueued.  `local` may be this rank's own slot of `gathered` (in place);
 * any other overlap is an error.  Equal counts run as ONE ncclAllGather; unequal counts as one grouped
 * ncclSend/ncclRecv exchange (every pair has a direct xGMI link) plus a device-to-device copy of the rank's own slice.
 * Enqueued on `stream` (hipStream_t); returns without synchronising. */
int iefvad_gather_scores(iefvad_comm* c, const float* local, size_t count, const int64_t* counts, float* gathered,
                         size_t gathered_capacity, void* stream);

/* The exchange iefvad_gather_scores would enqueue for (nranks, rank, counts | count), as data -- host only, no GPU, no
 * communicator: summary[0..4] = {equal counts (one all-gather) ? 1 : 0, this rank's offset, this rank's count, total
 * elements, number of point-to-point steps}; steps[4 i ..] = {peer, elements sent to it, offset at which its slice is
 * received, elements received}, one entry per peer in rank order (unequal counts only; `steps` may be NULL, else it holds
 * 4 (nranks - 1) values).  Zero-length transfers are listed with count 0 and are not issued. */
int iefvad_gather_plan(int32_t nranks, int32_t rank, const int64_t* counts, int64_t count, int64_t* summary, int64_t* steps);

const char* iefvad_last_error(void);
void iefvad_destroy(iefvad_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* IEFVAD_H */
