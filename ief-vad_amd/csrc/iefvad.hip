// libiefvad.so -- C ABI (include/iefvad.h) and launch orchestration of the IEF-VAD fusion forward
// on MI355X.  Replaces MMFMIL.forward -> MultiModal_Fusion_Attn_Iter.forward
// (/root/reference/model/imf_vad.py:40-44, :109-161).  gfx950 only; no host fallback.
#include "../../include/iefvad.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <exception>
#include <new>
#include <thread>
#include <vector>

#include "attention_f32.h"
#include "attention_bf16.h"
#include "attention_pbf16.h"
#include "attention_split.h"
#include "common.h"
#include "gather.h"
#include "gemm_f32.h"
#include "gemm_bf16.h"
#include "gemm_split.h"
#include "rowops.h"
#include "refine_chain_bf16.h"
#include "outproj_ln_chain_bf16.h"
#include "outproj_ln_pchain_bf16.h"
#include "outproj_ln_rchain_bf16.h"
#include "inproj_chain_bf16.h"
#include "heads_chain_bf16.h"
#include "heads_pchain_bf16.h"
#include "ragged.h"
#include "loss.h"
#include "gemm_split_tn.h"
#include "backward.h"
#include "metrics.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                   \
    } while (0)

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
struct iefvad_handle {
    iefvad_config cfg;
    int device;
    bool weights_set;
    // IEFVAD_ROWBLOCK_OFF=<mask> at iefvad_create (A/B runs and the bit-identity tests): bf16 mode takes the stage off its row-block kernel
    //   1 in_proj -> ring GEMM (+ the stand-alone cast); 2 out_proj + LayerNorm -> ring GEMM + LayerNorm kernel;
    //   4 heads + fusion -> ring GEMM + fusion kernel; 8 refinement chain -> 2K projection launches + the scorer kernel
    bool no_heads_fusion, no_ln_fusion;
    char* iproj_stream[2][IEFVAD_MAX_LAYERS];   // bf16 mode: in_proj weights in per-wave fragment order, q | k | v passes (inproj_chain_bf16.h)
    char* heads_stream;    // bf16 mode: the four head matrices in per-wave fragment order (heads_chain_bf16.h)
    int chain_min_blocks;  // bf16 mode: the refinement chain kernel takes a micro-batch from this many 64-row blocks on (IEFVAD_CHAIN_MIN_BLOCKS overrides)
    int rowblock_min_wgs;  // bf16 mode: the row-block kernels take a projection from this many workgroups on (IEFVAD_ROWBLOCK_MIN_WGS overrides)
    bool no_inproj_chain;
    char* oproj_stream[2][IEFVAD_MAX_LAYERS];   // bf16 mode: out_proj weights in per-wave fragment order (outproj_ln_chain_bf16.h)
    char* oproj_stream_r[2][IEFVAD_MAX_LAYERS]; // the same matrices with the output columns dealt to the waves for the in-register LayerNorm (outproj_ln_rchain_bf16.h)
    bool dense_encoder;    // IEFVAD_DENSE_ENCODER=1: whole-video passes run the encoder on whole 256-row chunks (pad rows computed), the tail on the gathered valid rows
    bool no_chain;
    char* chain_stream;    // bf16 mode: the refinement weights in the chain kernel's per-wave piece order (refine_chain_bf16.h)
    float* arena;          // one allocation holding every repacked weight
    size_t arena_floats;
    // pointers into the arena
    float* in_w[2][IEFVAD_MAX_LAYERS];
    float* in_b[2][IEFVAD_MAX_LAYERS];
    float* out_w[2][IEFVAD_MAX_LAYERS];
    float* out_b[2][IEFVAD_MAX_LAYERS];
    float* norm_w[2][IEFVAD_MAX_LAYERS];
    float* norm_b[2][IEFVAD_MAX_LAYERS];
    float* whiten_w[2];
    float* whiten_b[2];
    float* head_w[2];      // [1536, 768] = mu.weight stacked over logvar.weight
    float* head_b[2];      // [1536]
    float* ref_w1[IEFVAD_MAX_STEPS];
    float* ref_b1[IEFVAD_MAX_STEPS];
    float* ref_w2[IEFVAD_MAX_STEPS];
    float* ref_b2[IEFVAD_MAX_STEPS];
    float* cls_w;
    float* cls_b;
    // bf16 copies of the projection matrices (IEFVAD_COMPUTE_BF16 only)
    bf16_t* arena_b;
    bf16_t* in_wb[2][IEFVAD_MAX_LAYERS];
    bf16_t* out_wb[2][IEFVAD_MAX_LAYERS];
    bf16_t* head_wb[2];
    bf16_t* ref_w1b[IEFVAD_MAX_STEPS];
    bf16_t* ref_w2b[IEFVAD_MAX_STEPS];
    // three-plane bf16 splits [3][N][768] of the projection matrices (IEFVAD_COMPUTE_BF16X6 only)
    bf16_t* arena_s;
    bf16_t* in_ws[2][IEFVAD_MAX_LAYERS];
    bf16_t* out_ws[2][IEFVAD_MAX_LAYERS];
    bf16_t* head_ws[2];
    bf16_t* ref_w1s[IEFVAD_MAX_STEPS];
    bf16_t* ref_w2s[IEFVAD_MAX_STEPS];
    // IEFVAD_COMPUTE_FP16X3: two scaled fp16 planes [2][N][768] per projection matrix, the running-max words of the
    // matrices ([0, kAmaxActBase) of amax_dev, filled at set_weights) and of the activations of the current micro-batch
    _Float16* arena_h;
    float* amax_dev;
    _Float16* in_wh[2][IEFVAD_MAX_LAYERS];  const float* in_wa[2][IEFVAD_MAX_LAYERS];
    _Float16* out_wh[2][IEFVAD_MAX_LAYERS]; const float* out_wa[2][IEFVAD_MAX_LAYERS];
    _Float16* head_wh[2];                   const float* head_wa[2];
    _Float16* ref_w1h[IEFVAD_MAX_STEPS];    const float* ref_w1a[IEFVAD_MAX_STEPS];
    _Float16* ref_w2h[IEFVAD_MAX_STEPS];    const float* ref_w2a[IEFVAD_MAX_STEPS];
    struct EventPool* events;   // hipEvents of iefvad_forward_timed, reused across calls
    struct GraphCache* graphs;  // hipGraphs of small-batch forwards (cfg.graph_chunks)
    struct MetaRing* meta;      // pinned / device metadata buffers of iefvad_forward_videos
    int num_cus;                // compute units of the device: grid size of the persistent row-block kernels
    // training in the bf16x6 arithmetic: three-plane splits of the TRANSPOSED projection matrices ([3][768][n_out]: dX = dY W as an NT
    // product on the split kernel), rebuilt by the first train-mode forward after every iefvad_set_weights (train.h)
    bf16_t* arena_st; float* zero_bias; bool tplanes_valid;
    bf16_t* in_wst[2][IEFVAD_MAX_LAYERS]; bf16_t* out_wst[2][IEFVAD_MAX_LAYERS]; bf16_t* head_wst[2];
    bf16_t* ref_w1st[IEFVAD_MAX_STEPS]; bf16_t* ref_w2st[IEFVAD_MAX_STEPS];
    struct HostPipe* hostpipe;  // staging slots, copy stream and workspace of iefvad_forward_videos_host (hostpipe.h)
    struct TrainState* train;   // records of the train-mode forwards whose backward is outstanding (train.h)
    const float* row_scale[2];  // set for the duration of one iefvad_forward_scaled call: per-row input scales (image, event), nullable
};
static const int kAmaxActBase = 256;   // running-max slots of the projection matrices (multi-way words); behind them the activations'
static int amax_act_tensors(int L, int K) { return 2 + 6 * L + 2 * K + 1; }   // inputs, per layer att|x|qkv x 2 modalities, z_0..z_K, h_0..h_{K-1}
static size_t amax_words(int L, int K, int mb_chunks) {
    return (size_t)kAmaxActBase * IEF_AMAX_FLOATS + (size_t)amax_act_tensors(L, K) * mb_chunks * IEF_AMAX_PARTS;
}

// chunks per internal pass: 256 (65,536 rows, 2.8 GB of workspace) in fp32 mode; the bf16 kernels are ~100 us
// each at that size and gain another 7 % from 4x longer launches (1024 chunks, 11 GB of workspace)
// the split modes (bf16x6 / fp16x3) gain 2.4 % from 4x longer launches as well (fewer kernel-boundary tails)
static const int kDefaultMicroBatchF32 = 256;
static const int kDefaultMicroBatchBF16 = 1024;

static int micro_batch(const iefvad_handle* h) {
    if (h->cfg.micro_batch > 0) return h->cfg.micro_batch < 16384 ? h->cfg.micro_batch : 16384;   // attention grid.z = 2 x chunks
    return h->cfg.compute == IEFVAD_COMPUTE_F32 ? kDefaultMicroBatchF32 : kDefaultMicroBatchBF16;
}

extern "C" int iefvad_abi_version(void) { return IEFVAD_ABI_VERSION; }

extern "C" const char* iefvad_last_error(void) { return g_err; }

extern "C" int iefvad_create(const iefvad_config* cfg, iefvad_handle** out) {
    if (!cfg || !out) return fail("iefvad_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != IEFVAD_ABI_VERSION)
        return fail("iefvad_create: abi_version %d, library is %d", cfg->abi_version, IEFVAD_ABI_VERSION);
    if (cfg->embed_dim != IEF_D || cfg->seq_len != IEF_T || cfg->num_heads != IEF_H)
        return fail("iefvad_create: kernels are built for D=768, T=256, H=8 (got D=%d T=%d H=%d)",
                    cfg->embed_dim, cfg->seq_len, cfg->num_heads);
    if (cfg->num_layers < 1 || cfg->num_layers > IEFVAD_MAX_LAYERS)
        return fail("iefvad_create: num_layers %d outside 1..%d", cfg->num_layers, IEFVAD_MAX_LAYERS);
    if (cfg->num_steps < 0 || cfg->num_steps > IEFVAD_MAX_STEPS)
        return fail("iefvad_create: num_steps %d outside 0..%d", cfg->num_steps, IEFVAD_MAX_STEPS);
    if (cfg->noise_model != IEFVAD_NOISE_GAUSSIAN && cfg->noise_model != IEFVAD_NOISE_STUDENT_T)
        return fail("Unsupported noise_model. Choose 'Gaussian' or 'StudentT'.");   // imf_vad.py:138
    if (cfg->compute != IEFVAD_COMPUTE_F32 && cfg->compute != IEFVAD_COMPUTE_BF16 && cfg->compute != IEFVAD_COMPUTE_BF16X6 &&
        cfg->compute != IEFVAD_COMPUTE_FP16X3)
        return fail("iefvad_create: unknown compute mode %d", cfg->compute);
    if (cfg->noise_model == IEFVAD_NOISE_STUDENT_T && !(cfg->nu != 0.f))
        return fail("iefvad_create: nu must be non-zero for StudentT");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("iefvad_create: no HIP device");
    iefvad_handle* h = new (std::nothrow) iefvad_handle();
    if (!h) return fail("iefvad_create: out of host memory");
    memset(h, 0, sizeof(*h));
    h->cfg = *cfg;
    {
        const char* v = getenv("IEFVAD_ROWBLOCK_OFF");
        const int off = v ? atoi(v) : 0;
        h->no_inproj_chain = off & 1; h->no_ln_fusion = off & 2; h->no_heads_fusion = off & 4; h->no_chain = off & 8;
    }
    { const char* v = getenv("IEFVAD_DENSE_ENCODER"); h->dense_encoder = v && v[0] == '1'; }
    { const char* v = getenv("IEFVAD_ROWBLOCK_MIN_WGS"); h->rowblock_min_wgs = (v && atoi(v) > 0) ? atoi(v) : 128; }      // tools/rowblock_threshold_probe.py
    { const char* v = getenv("IEFVAD_CHAIN_MIN_BLOCKS"); h->chain_min_blocks = (v && atoi(v) > 0) ? atoi(v) : 4; }         // one chunk: 4 blocks take one block time, 2K launches more
    hipError_t e = hipGetDevice(&h->device);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_gemm_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GB2_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_gemm_bf16_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GB2_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_gemm_bf16_w256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GB3_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_refine_chain_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RC_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_outproj_ln_chain_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                OC_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_outproj_ln_pchain_bf16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                OP_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_outproj_ln_pchain_bf16_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                OP_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_outproj_ln_pchain_bf16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                OP_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_outproj_ln_rchain_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                OR_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_heads_pchain_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HP_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_gemm_split_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_gemm_split_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES_OF(4));
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_attention_pbf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, APB_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_attention_pbf16_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, APB_LDS_BYTES);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, h->device);
        h->num_cus = e == hipSuccess ? prop.multiProcessorCount : 256;
        g_num_cus = h->num_cus;
    }
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_gemm_f32_t256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GB2_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_gemm_split_n128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GS_LDS_BYTES_OF(2));
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_attention_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ATS_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_attention_split_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ATS_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_attention_split_train_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ATS_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_attention_split_ds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ATS_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_attention_split_train_mask_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ATS_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_heads_chain_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HC_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_inproj_chain_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                IC_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_inproj_chain_f32in_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                IC_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_attention_split_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ATS_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)iefvad_gemm_split_f16_n128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GS_LDS_BYTES_OF(2));
    if (e == hipSuccess && cfg->compute == IEFVAD_COMPUTE_FP16X3) e = hipMalloc((void**)&h->amax_dev, amax_words(cfg->num_layers, cfg->num_steps, micro_batch(h)) * sizeof(float));
    if (e != hipSuccess) {
        delete h;
        return fail("iefvad_create: %s", hipGetErrorString(e));
    }
    *out = h;
    return 0;
}

static void release_events(iefvad_handle* h);
static void release_graphs(iefvad_handle* h);
static void release_meta(iefvad_handle* h);
static void release_train(iefvad_handle* h);
static void release_hostpipe(iefvad_handle* h);

extern "C" void iefvad_destroy(iefvad_handle* h) {
    if (!h) return;
    if (h->arena) (void)hipFree(h->arena);
    if (h->arena_b) (void)hipFree(h->arena_b);
    if (h->chain_stream) (void)hipFree(h->chain_stream);
    if (h->heads_stream) (void)hipFree(h->heads_stream);
    for (int m = 0; m < 2; ++m)
        for (int l = 0; l < IEFVAD_MAX_LAYERS; ++l) {
            if (h->oproj_stream[m][l]) (void)hipFree(h->oproj_stream[m][l]);
            if (h->oproj_stream_r[m][l]) (void)hipFree(h->oproj_stream_r[m][l]);
            if (h->iproj_stream[m][l]) (void)hipFree(h->iproj_stream[m][l]);
        }
    if (h->arena_s) (void)hipFree(h->arena_s);
    if (h->arena_st) (void)hipFree(h->arena_st);
    if (h->zero_bias) (void)hipFree(h->zero_bias);
    if (h->arena_h) (void)hipFree(h->arena_h);
    if (h->amax_dev) (void)hipFree(h->amax_dev);
    release_events(h);
    release_graphs(h);
    release_meta(h);
    release_train(h);
    release_hostpipe(h);
    delete h;
}

template <typename T>
static int launch_cast(const void* in0, const void* in1, float* o0, float* o1, bf16_t* b0, bf16_t* b1, size_t n, int nsrc,
                       hipStream_t stream) {
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(iefvad_cast_kernel<T>, dim3((unsigned)blocks, nsrc), dim3(256), 0, stream, (const T*)in0,
                       (const T*)in1, o0, o1, b0, b1, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int launch_cast_scaled(const void* in0, const void* in1, float* o0, float* o1, bf16_t* b0, bf16_t* b1, size_t n, const float* s0,
                              const float* s1, hipStream_t stream) {
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(iefvad_cast_scaled_kernel<T>, dim3((unsigned)blocks, 2), dim3(256), 0, stream, (const T*)in0, (const T*)in1, o0, o1, b0, b1,
                       n, s0, s1);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int launch_split_planes(const float* src, bf16_t* planes, size_t n, hipStream_t stream) {
    if (n % 4) return fail("split_bf16x3: n = %zu is not a multiple of 4", n);
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(iefvad_split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, planes, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

// batches of iefvad_split_planes_many_kernel launches (gemm_split.h): add() queues a matrix, flush() launches what is queued
struct SplitMany {
    SplitManyArgs a;
    hipStream_t stream;
    explicit SplitMany(hipStream_t s) : stream(s) { a.count = 0; }
    int flush() {
        if (a.count == 0) return 0;
        hipLaunchKernelGGL(iefvad_split_planes_many_kernel, dim3(96, a.count), dim3(256), 0, stream, a);
        a.count = 0;
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // rows = 0: planes of src [n]; rows = n_out: planes of the transpose of src [n_out, n / n_out]
    int add(const float* src, bf16_t* dst, size_t n, int rows) {
        if (n % 4 || n > 0xffffffffu) return fail("split_bf16x3: n = %zu is not a multiple of 4 below 2^32", n);
        if (rows && (rows % 64 || (n / rows) % 32 || n % rows)) return fail("split_bf16x3: transposed split of a [%d, %zu] matrix", rows, n / rows);
        a.src[a.count] = src; a.dst[a.count] = dst; a.n[a.count] = (unsigned)n; a.rows[a.count] = rows;
        if (++a.count == SPLIT_MANY_MAX) return flush();
        return 0;
    }
};

extern "C" int iefvad_split_bf16x3_many(const float* const* src, void* const* planes, const size_t* n, const int32_t* rows, int32_t count,
                                        void* stream) {
    if (!src || !planes || !n || !rows || count < 0) return fail("iefvad_split_bf16x3_many: null argument");
    SplitMany sm((hipStream_t)stream);
    for (int i = 0; i < count; ++i) {
        if (!src[i] || !planes[i] || rows[i] < 0) return fail("iefvad_split_bf16x3_many: entry %d is null or has negative rows", i);
        if (((uintptr_t)src[i] | (uintptr_t)planes[i]) & 15) return fail("iefvad_split_bf16x3_many: entry %d is not 16-byte aligned", i);
        if (int rc = sm.add(src[i], (bf16_t*)planes[i], n[i], rows[i])) return rc;
    }
    return sm.flush();
}

extern "C" int iefvad_split_bf16x3(const float* src, void* planes, size_t n, void* stream) {
    if (!src || !planes) return fail("iefvad_split_bf16x3: null argument");
    return launch_split_planes(src, (bf16_t*)planes, n, (hipStream_t)stream);
}

// up to COPY_MANY_MAX device-to-device copies in one launch: blockIdx.y = tensor, 32 workgroups stride over it (16-byte moves when
// both ends are 16-byte aligned)
#define COPY_MANY_MAX 96
struct CopyManyArgs {
    float* dst[COPY_MANY_MAX];
    const float* src[COPY_MANY_MAX];
    unsigned count[COPY_MANY_MAX];
    int n;
};
__global__ __launch_bounds__(256) void iefvad_copy_many_kernel(CopyManyArgs a) {
    float* d = a.dst[blockIdx.y];
    const float* s = a.src[blockIdx.y];
    const unsigned n = a.count[blockIdx.y];
    const unsigned tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {
        const unsigned n4 = n >> 2;
        for (unsigned i = tid; i < n4; i += nth) ((f32x4*)d)[i] = ((const f32x4*)s)[i];
        for (unsigned i = (n4 << 2) + tid; i < n; i += nth) d[i] = s[i];
    } else {
        for (unsigned i = tid; i < n; i += nth) d[i] = s[i];
    }
}

extern "C" int iefvad_set_weights(iefvad_handle* h, const iefvad_weights* w, void* stream_) {
    if (!h || !w) return fail("iefvad_set_weights: null argument");
    hipStream_t stream = (hipStream_t)stream_;
    const int L = h->cfg.num_layers, K = h->cfg.num_steps;
    const size_t D = IEF_D, DD = D * D;
    // validate pointers first
    for (int m = 0; m < 2; ++m) {
        for (int l = 0; l < L; ++l)
            if (!w->in_proj_w[m][l] || !w->in_proj_b[m][l] || !w->out_proj_w[m][l] || !w->out_proj_b[m][l] ||
                !w->norm_w[m][l] || !w->norm_b[m][l])
                return fail("iefvad_set_weights: null encoder weight (modality %d layer %d)", m, l);
        if (!w->whiten_w[m] || !w->whiten_b[m] || !w->mu_w[m] || !w->mu_b[m] || !w->logvar_w[m] || !w->logvar_b[m])
            return fail("iefvad_set_weights: null head weight (modality %d)", m);
    }
    for (int k = 0; k < K; ++k)
        if (!w->ref_w1[k] || !w->ref_b1[k] || !w->ref_w2[k] || !w->ref_b2[k])
            return fail("iefvad_set_weights: null refinement weight (step %d)", k);
    if (!w->cls_w || !w->cls_b) return fail("iefvad_set_weights: null classifier weight");

    const size_t total = 2 * L * (3 * DD + 3 * D + DD + D + 2 * D) + 2 * (2 * D) + 2 * (2 * DD + 2 * D) +
                         (size_t)K * (2 * DD + 2 * D) + D + 4;
    if (!h->arena) {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMalloc((void**)&h->arena, total * sizeof(float)));
        h->arena_floats = total;
    }
    float* p = h->arena;
    // the 38 + 4 K tensors go into the arena in batches of one launch (a training step re-uploads every parameter: 78 copies of
    // 6 us each were 2 % of it)
    CopyManyArgs cm;
    cm.n = 0;
    auto flush = [&]() -> hipError_t {
        if (cm.n == 0) return hipSuccess;
        hipLaunchKernelGGL(iefvad_copy_many_kernel, dim3(32, cm.n), dim3(256), 0, stream, cm);
        cm.n = 0;
        return hipGetLastError();
    };
    auto put = [&](float** dst, const float* src, size_t n) -> hipError_t {
        *dst = p;
        p += n;
        cm.dst[cm.n] = *dst; cm.src[cm.n] = src; cm.count[cm.n] = (unsigned)n;
        if (++cm.n == COPY_MANY_MAX) return flush();
        return hipSuccess;
    };
    for (int m = 0; m < 2; ++m) {
        for (int l = 0; l < L; ++l) {
            HIP_TRY(put(&h->in_w[m][l], w->in_proj_w[m][l], 3 * DD));
            HIP_TRY(put(&h->in_b[m][l], w->in_proj_b[m][l], 3 * D));
            HIP_TRY(put(&h->out_w[m][l], w->out_proj_w[m][l], DD));
            HIP_TRY(put(&h->out_b[m][l], w->out_proj_b[m][l], D));
            HIP_TRY(put(&h->norm_w[m][l], w->norm_w[m][l], D));
            HIP_TRY(put(&h->norm_b[m][l], w->norm_b[m][l], D));
        }
        HIP_TRY(put(&h->whiten_w[m], w->whiten_w[m], D));
        HIP_TRY(put(&h->whiten_b[m], w->whiten_b[m], D));
        // mu | logvar heads share their A operand: stack them into one [1536, 768] projection
        HIP_TRY(put(&h->head_w[m], w->mu_w[m], DD));
        float* dummy;
        HIP_TRY(put(&dummy, w->logvar_w[m], DD));
        HIP_TRY(put(&h->head_b[m], w->mu_b[m], D));
        HIP_TRY(put(&dummy, w->logvar_b[m], D));
    }
    for (int k = 0; k < K; ++k) {
        HIP_TRY(put(&h->ref_w1[k], w->ref_w1[k], DD));
        HIP_TRY(put(&h->ref_b1[k], w->ref_b1[k], D));
        HIP_TRY(put(&h->ref_w2[k], w->ref_w2[k], DD));
        HIP_TRY(put(&h->ref_b2[k], w->ref_b2[k], D));
    }
    HIP_TRY(put(&h->cls_w, w->cls_w, D));
    HIP_TRY(put(&h->cls_b, w->cls_b, 1));
    HIP_TRY(flush());
    if ((size_t)(p - h->arena) > h->arena_floats) return fail("iefvad_set_weights: arena overflow");
    if (h->cfg.compute == IEFVAD_COMPUTE_BF16) {
        // bf16 (round-to-nearest-even) copies of every projection matrix; biases, LayerNorm and the scorer stay fp32
        const size_t nb = 2 * (size_t)L * (3 * DD + DD) + 2 * (2 * DD) + (size_t)K * 2 * DD;
        if (!h->arena_b) HIP_TRY(hipMalloc((void**)&h->arena_b, nb * sizeof(bf16_t)));
        bf16_t* q = h->arena_b;
        auto conv = [&](bf16_t** dst, const float* src, size_t n) -> int {
            *dst = q;
            q += n;
            return launch_cast<float>(src, nullptr, nullptr, nullptr, *dst, nullptr, n, 1, stream);
        };
        for (int m = 0; m < 2; ++m) {
            for (int l = 0; l < L; ++l) {
                if (int rc = conv(&h->in_wb[m][l], h->in_w[m][l], 3 * DD)) return rc;
                if (int rc = conv(&h->out_wb[m][l], h->out_w[m][l], DD)) return rc;
            }
            if (int rc = conv(&h->head_wb[m], h->head_w[m], 2 * DD)) return rc;
        }
        for (int k = 0; k < K; ++k) {
            if (int rc = conv(&h->ref_w1b[k], h->ref_w1[k], DD)) return rc;
            if (int rc = conv(&h->ref_w2b[k], h->ref_w2[k], DD)) return rc;
        }
        for (int m = 0; m < 2; ++m)
            for (int l = 0; l < L; ++l) {
                if (!h->oproj_stream[m][l]) HIP_TRY(hipMalloc((void**)&h->oproj_stream[m][l], wstream_bytes()));
                hipLaunchKernelGGL(iefvad_wstream_pack_kernel, dim3(256), dim3(256), 0, stream, h->out_wb[m][l], h->oproj_stream[m][l], 1);
                HIP_TRY(hipGetLastError());
                if (!h->oproj_stream_r[m][l]) HIP_TRY(hipMalloc((void**)&h->oproj_stream_r[m][l], wstream_bytes()));
                hipLaunchKernelGGL(iefvad_wstream_pack_colmap_kernel, dim3(256), dim3(256), 0, stream, h->out_wb[m][l], h->oproj_stream_r[m][l]);
                HIP_TRY(hipGetLastError());
                if (!h->iproj_stream[m][l]) HIP_TRY(hipMalloc((void**)&h->iproj_stream[m][l], wstream_bytes(IC_NPASS)));
                hipLaunchKernelGGL(iefvad_wstream_pack_kernel, dim3(512), dim3(256), 0, stream, h->in_wb[m][l], h->iproj_stream[m][l], IC_NPASS);
                HIP_TRY(hipGetLastError());
            }
        if (!h->heads_stream) HIP_TRY(hipMalloc((void**)&h->heads_stream, heads_stream_bytes()));
        hipLaunchKernelGGL(iefvad_heads_pack_kernel, dim3(512), dim3(256), 0, stream, h->head_wb[0], h->head_wb[1], h->heads_stream);
        HIP_TRY(hipGetLastError());
        if (K > 0) {
            // the same bf16 matrices (and the fp32 biases) once more, in the chain kernel's per-wave piece order
            if (!h->chain_stream) HIP_TRY(hipMalloc((void**)&h->chain_stream, chain_stream_bytes(K)));
            ChainPackArgs pa;
            memset(&pa, 0, sizeof(pa));
            for (int k = 0; k < K; ++k) {
                pa.W[2 * k] = h->ref_w1b[k]; pa.bias[2 * k] = h->ref_b1[k];
                pa.W[2 * k + 1] = h->ref_w2b[k]; pa.bias[2 * k + 1] = h->ref_b2[k];
            }
            pa.stream = h->chain_stream;
            pa.K = K;
            hipLaunchKernelGGL(iefvad_chain_pack_kernel, dim3(2048), dim3(256), 0, stream, pa);
            HIP_TRY(hipGetLastError());
        }
    }
    if (h->cfg.compute == IEFVAD_COMPUTE_BF16X6) {
        // exact three-term bf16 split of every projection matrix (gemm_split.h); biases, LayerNorm, scorer stay fp32
        const size_t nb = 2 * (size_t)L * (3 * DD + DD) + 2 * (2 * DD) + (size_t)K * 2 * DD;
        if (!h->arena_s) HIP_TRY(hipMalloc((void**)&h->arena_s, 3 * nb * sizeof(bf16_t)));
        bf16_t* q = h->arena_s;
        SplitMany sm(stream);                 // all 10 + 2 K matrices in one launch (two from K = 12 on)
        auto split = [&](bf16_t** dst, const float* src, size_t n) -> int {
            *dst = q;
            q += 3 * n;
            return sm.add(src, *dst, n, 0);
        };
        for (int m = 0; m < 2; ++m) {
            for (int l = 0; l < L; ++l) {
                if (int rc = split(&h->in_ws[m][l], h->in_w[m][l], 3 * DD)) return rc;
                if (int rc = split(&h->out_ws[m][l], h->out_w[m][l], DD)) return rc;
            }
            if (int rc = split(&h->head_ws[m], h->head_w[m], 2 * DD)) return rc;
        }
        for (int k = 0; k < K; ++k) {
            if (int rc = split(&h->ref_w1s[k], h->ref_w1[k], DD)) return rc;
            if (int rc = split(&h->ref_w2s[k], h->ref_w2[k], DD)) return rc;
        }
        if (int rc = sm.flush()) return rc;
    }
    if (h->cfg.compute == IEFVAD_COMPUTE_FP16X3) {
        // two fp16 planes per projection matrix, scaled by a power of two from the matrix's max |w| (gemm_split.h, F16)
        const size_t nb = 2 * (size_t)L * (3 * DD + DD) + 2 * (2 * DD) + (size_t)K * 2 * DD;
        if (!h->arena_h) HIP_TRY(hipMalloc((void**)&h->arena_h, 2 * nb * sizeof(_Float16)));
        HIP_TRY(hipMemsetAsync(h->amax_dev, 0, kAmaxActBase * IEF_AMAX_FLOATS * sizeof(float), stream));
        _Float16* q = h->arena_h;
        int widx = 0;
        auto split = [&](_Float16** dst, const float** amax, const float* src, size_t n) -> int {
            if (widx >= kAmaxActBase) return fail("iefvad_set_weights: too many projection matrices");
            *dst = q;
            q += 2 * n;
            float* word = h->amax_dev + IEF_AMAX_FLOATS * widx++;
            *amax = word;
            size_t blocks = (n / 4 + 255) / 256;
            if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(iefvad_amax_kernel, dim3((unsigned)blocks, 1), dim3(256), 0, stream, src, src, word, word, n);
            hipLaunchKernelGGL(iefvad_split_planes_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, *dst, n,
                               (const float*)word);
            HIP_TRY(hipGetLastError());
            return 0;
        };
        for (int m = 0; m < 2; ++m) {
            for (int l = 0; l < L; ++l) {
                if (int rc = split(&h->in_wh[m][l], &h->in_wa[m][l], h->in_w[m][l], 3 * DD)) return rc;
                if (int rc = split(&h->out_wh[m][l], &h->out_wa[m][l], h->out_w[m][l], DD)) return rc;
            }
            if (int rc = split(&h->head_wh[m], &h->head_wa[m], h->head_w[m], 2 * DD)) return rc;
        }
        for (int k = 0; k < K; ++k) {
            if (int rc = split(&h->ref_w1h[k], &h->ref_w1a[k], h->ref_w1[k], DD)) return rc;
            if (int rc = split(&h->ref_w2h[k], &h->ref_w2a[k], h->ref_w2[k], DD)) return rc;
        }
    }
    h->weights_set = true;
    h->tplanes_valid = false;
    return 0;
}

// workspace, in floats per row: xin(2) + qkv(6) + att(2) + y(2) + x(2) = 14 * 768, + 1 (logits scratch).
// mu_i, lv_i, mu_e, lv_e, z, h alias the qkv region (6 * 768), which is dead once the encoder is done.
static const size_t kWsFloatsPerRow = 14 * IEF_D + 4;

extern "C" size_t iefvad_workspace_bytes(const iefvad_handle* h, int32_t B) {
    if (!h || B <= 0) return 0;
    const int mb = micro_batch(h);
    const size_t rows = (size_t)(B < mb ? B : mb) * IEF_T;
    return rows * kWsFloatsPerRow * sizeof(float) + 256;
}

// ------------------------------------------------------------------------------------------------
// stage timing
// ------------------------------------------------------------------------------------------------
enum Stage { ST_QKV = 0, ST_ATT, ST_OUT, ST_LN, ST_HEAD, ST_FUSION, ST_REFINE, ST_SCORER, ST_CAST, ST_COUNT };

// hipEvents are pooled on the handle and reused by every timed call (a timed forward of one micro-batch brackets ~35
// launches; creating and destroying 70 events per call would sit inside bench.py's timed region).
struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipError_t err = hipSuccess;
    hipEvent_t take() {
        if (used == ev.size()) {
            hipEvent_t e = nullptr;
            hipError_t r = hipEventCreate(&e);
            if (r != hipSuccess) {
                if (err == hipSuccess) err = r;
                return nullptr;
            }
            ev.push_back(e);
        }
        return ev[used++];
    }
    void release_all() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear();
        used = 0;
    }
};

struct Timer {
    bool on = false;
    hipStream_t stream = nullptr;
    EventPool* pool = nullptr;
    struct Span { int stage; hipEvent_t a, b; };
    std::vector<Span> spans;
    int gemm_launches = 0;
    hipEvent_t begin(int stage) {
        if (!on) return nullptr;
        Span s;
        s.stage = stage;
        s.a = pool->take();
        s.b = pool->take();
        if (!s.a || !s.b) return nullptr;          // pool->err is reported by iefvad_forward_timed
        (void)hipEventRecord(s.a, stream);
        spans.push_back(s);
        return s.b;
    }
    void end(hipEvent_t b) {
        if (on && b) (void)hipEventRecord(b, stream);
    }
};

static void release_events(iefvad_handle* h) {
    if (h->events) {
        h->events->release_all();
        delete h->events;
        h->events = nullptr;
    }
}

// A/B switch for tests and tools (IEFVAD_NO_TINY_GEMM=1): route small problems to the 64x64 kernel as before round 2
static const bool g_force_no_tiny = [] { const char* v = getenv("IEFVAD_NO_TINY_GEMM"); return v && v[0] == '1'; }();
// grid-size rules of the fp32 tilings (tools/f32_threshold_probe.py): IEFVAD_F32_RULES="tiny_max_blocks64,t256_min_blocks,small_max_blocks128"
static int g_f32_rules[3] = {320, 1536, 1024};      // were {320, 256, 256} ("when the grid fills the chip") until measured: B = 48 forward 6.7 -> 5.5 ms
static const bool g_f32_rules_read = [] {
    const char* v = getenv("IEFVAD_F32_RULES");
    if (v) (void)sscanf(v, "%d,%d,%d", &g_f32_rules[0], &g_f32_rules[1], &g_f32_rules[2]);
    return true;
}();

static int launch_gemm(const GemmArgs& a, int nz, hipStream_t stream, Timer& tm, int stage) {
    if (a.M % GEMM_BM || a.N % GEMM_BN || a.K % GEMM_BK)
        return fail("gemm: shape M=%d N=%d K=%d not a multiple of the %dx%dx%d tile", a.M, a.N, a.K, GEMM_BM, GEMM_BN,
                    GEMM_BK);
    // Four tilings of the same contraction, bit-identical to each other (same k order per output element):
    //   128x256 / 3-slot ring (iefvad_gemm_f32_t256_kernel)  the throughput kernel, from 6 blocks per CU on (it wins 2 % at a full
    //                                                         micro-batch and loses 10-18 % between 1 and 4 blocks per CU);
    //   128x128 / double buffer (iefvad_gemm_f32_kernel)      mid-size grids or N not a multiple of 256;
    //   64x64 (iefvad_gemm_f32_small_kernel)                  up to 4 blocks of 128x128 per CU: 4x the blocks, a quarter of the MFMA chain;
    //   32x32 on 16x16x4 MFMAs (iefvad_gemm_f32_tiny_kernel)  the per-video pattern (B = 1 .. a few chunks): 16x the
    //                                                         blocks, a wave's chain is 3.2 us instead of 10 / 41 us.
    const int blocks128 = (a.M / GEMM_BM) * (a.N / GEMM_BN) * nz;
    const int blocks64 = (a.M / GEMS_BM) * (a.N / GEMS_BN) * nz;
    const bool t256_ok = (a.N % GB2_BN == 0) && (a.K % 16 == 0) && (a.K >= 32);
    const int blocks256 = t256_ok ? (a.M / GB2_BM) * (a.N / GB2_BN) * nz : 0;
    const bool tiny = blocks64 < g_f32_rules[0] && a.K % 64 == 0 && !g_force_no_tiny;      // < 1.25 blocks of 64x64 per CU: the chain, not the chip, bounds it
    hipEvent_t e = tm.begin(stage);
    if (tiny) {
        dim3 grid((a.M / GEMT_BM) * (a.N / GEMT_BN), 1, nz);
        hipLaunchKernelGGL(iefvad_gemm_f32_tiny_kernel, grid, dim3(256), 0, stream, a);
    } else if (blocks256 >= g_f32_rules[1]) {
        GemmBArgs b;
        memset(&b, 0, sizeof(b));
        b.M = a.M; b.N = a.N; b.K = a.K; b.lda = a.lda; b.ldc = a.ldc; b.epi = a.epi; b.alpha = a.alpha; b.qcols = a.qcols;
        for (int m = 0; m < nz; ++m) {
            b.p[m].A = (const bf16_t*)a.p[m].A; b.p[m].W = (const bf16_t*)a.p[m].W;     // fp32 data behind the typed pointer
            b.p[m].bias = a.p[m].bias; b.p[m].C = a.p[m].C; b.p[m].R = a.p[m].R; b.p[m].C2 = a.p[m].C2;
        }
        dim3 grid((a.M / GB2_BM) * (a.N / GB2_BN), 1, nz);
        hipLaunchKernelGGL(iefvad_gemm_f32_t256_kernel, grid, dim3(256), GB2_LDS_BYTES, stream, b);
    } else if (blocks128 < g_f32_rules[2]) {
        dim3 grid((a.M / GEMS_BM) * (a.N / GEMS_BN), 1, nz);
        hipLaunchKernelGGL(iefvad_gemm_f32_small_kernel, grid, dim3(256), 0, stream, a);
    } else {
        dim3 grid((a.M / GEMM_BM) * (a.N / GEMM_BN), 1, nz);
        hipLaunchKernelGGL(iefvad_gemm_f32_kernel, grid, dim3(256), 0, stream, a);
    }
    tm.end(e);
    tm.gemm_launches += 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

static int launch_gemm_b(const GemmBArgs& a, int nz, hipStream_t stream, Timer& tm, int stage) {
    hipEvent_t e;
    if (a.N % GB2_BN == 0 && a.M % GB2_BM == 0 && a.K % GB2_BK == 0 && a.K >= 2 * GB2_BK) {
        e = tm.begin(stage);
        // two bit-identical tilings: 256 x 256 / 8 waves / one workgroup per CU is 2-6 % faster with bias-type epilogues
        // (in_proj, out_proj, heads, the refinement's first projection), 128 x 256 / 4 waves / two per CU with the refinement
        // epilogue (residual read + two stores): tools/gemm_tune_bf16, profiles/r02_gemm_bf16_w256.log
        if (a.M % GB3_BM == 0 && a.epi != EPI_REFINE && (a.M / GB3_BM) * (a.N / GB2_BN) * nz >= 256) {
            dim3 grid((a.M / GB3_BM) * (a.N / GB2_BN), 1, nz);
            hipLaunchKernelGGL(iefvad_gemm_bf16_w256_kernel, grid, dim3(512), GB3_LDS_BYTES, stream, a);
        } else {
            dim3 grid((a.M / GB2_BM) * (a.N / GB2_BN), 1, nz);
            hipLaunchKernelGGL(iefvad_gemm_bf16_pipe_kernel, grid, dim3(256), GB2_LDS_BYTES, stream, a);   // pinned issue order: +1..3 %, same bits
        }
    } else {
        if (a.M % GEMM_BM || a.N % GEMM_BN || a.K % GEMMB_BK)
            return fail("gemm(bf16): shape M=%d N=%d K=%d not a multiple of the %dx%dx%d tile", a.M, a.N, a.K, GEMM_BM,
                        GEMM_BN, GEMMB_BK);
        dim3 grid((a.M / GEMM_BM) * (a.N / GEMM_BN), 1, nz);
        e = tm.begin(stage);
        hipLaunchKernelGGL(iefvad_gemm_bf16_v1_kernel, grid, dim3(256), 0, stream, a);
    }
    tm.end(e);
    tm.gemm_launches += 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

// BF16X6: the split kernel (128 x 128 tiles, two workgroups per CU; tools/gemm_tune_split: +5..8 % over the 128 x 256 /
// one-workgroup configuration, same bits) takes a micro-batch's projections from 72 workgroups of a 768-wide projection on
// (6 chunks: measured crossover, B = 24 forward 3.55 -> 1.97 ms; it was 512 workgroups, "fills the chip", until the end of
// round 3); smaller problems run on the fp32 kernels (launch_gemm)
static const int kSplitBN = GS_BN_OF(2);
static int g_split_min_wgs = [] { const char* v = getenv("IEFVAD_SPLIT_MIN_WGS"); return (v && atoi(v) > 0) ? atoi(v) : 72; }();      // 6 chunks; tools/split_threshold_probe.py
static bool split_eligible(int M, int N, int K, int nz) {
    return M % GS_BM == 0 && N % kSplitBN == 0 && K % 64 == 0 && K >= 64 && (M / GS_BM) * (N / kSplitBN) * nz >= g_split_min_wgs;
}

static int launch_gemm_split(const GemmBArgs& a, int nz, hipStream_t stream, Timer& tm, int stage, bool f16 = false) {
    if (a.M % GS_BM || a.N % kSplitBN || a.K % 64 || a.K < 64)
        return fail("gemm(bf16x6): shape M=%d N=%d K=%d not a multiple of the %dx%dx64 tile", a.M, a.N, a.K, GS_BM, kSplitBN);
    dim3 grid((a.M / GS_BM) * (a.N / kSplitBN), 1, nz);
    hipEvent_t e = tm.begin(stage);
    if (f16) hipLaunchKernelGGL(iefvad_gemm_split_f16_n128_kernel, grid, dim3(256), GS_LDS_BYTES_OF(2), stream, a);
    else hipLaunchKernelGGL(iefvad_gemm_split_n128_kernel, grid, dim3(256), GS_LDS_BYTES_OF(2), stream, a);
    tm.end(e);
    tm.gemm_launches += 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

static size_t in_elem_bytes(int in_dtype) { return in_dtype == IEFVAD_IN_F32 ? 4 : 2; }

// One projection in either arithmetic: fills the fp32 or the bf16 argument block from the same description.
struct Proj {
    const float* A32[2];      // fp32 A operand (IEFVAD_COMPUTE_F32)
    const bf16_t* A16[2];     // bf16 A operand (IEFVAD_COMPUTE_BF16)
    const float* W32[2];
    const bf16_t* W16[2];
    const bf16_t* Ws[2];      // three-plane split of W (IEFVAD_COMPUTE_BF16X6)
    const _Float16* Wh[2];    // two scaled fp16 planes of W + the running-max words (IEFVAD_COMPUTE_FP16X3)
    const float* amaxW[2];
    const float* amaxA[2];
    float* amaxC[2];
    const float* bias[2];
    float* C[2];              // fp32 result (nullable in bf16 mode)
    bf16_t* Cb[2];            // bf16 copy of the result (bf16 mode only, nullable)
    const float* R[2];
    float* C2[2];
    int N, ldc, epi, nz;
    float alpha;
    int qcols;
};

static int launch_proj(const Proj& p, int compute, bool use_split, int rows, hipStream_t stream, Timer& tm, int stage) {
    const bool bf16 = (compute == IEFVAD_COMPUTE_BF16);
    if (use_split) {   // one rule for every projection of a micro-batch: the 768-wide, single-problem grid fills the chip
        const bool f16 = (compute == IEFVAD_COMPUTE_FP16X3);
        GemmBArgs g;
        memset(&g, 0, sizeof(g));
        g.M = rows; g.N = p.N; g.K = IEF_D; g.lda = IEF_D; g.ldc = p.ldc; g.epi = p.epi; g.alpha = p.alpha; g.qcols = p.qcols;
        g.wplane = p.N * IEF_D * 2;
        for (int m = 0; m < p.nz; ++m) {
            g.p[m].A = (const bf16_t*)p.A32[m];          // fp32 data behind the typed pointer
            g.p[m].W = f16 ? (const bf16_t*)p.Wh[m] : p.Ws[m];
            g.p[m].bias = p.bias[m]; g.p[m].C = p.C[m]; g.p[m].R = p.R[m]; g.p[m].C2 = p.C2[m];
            if (f16) {
                if (!p.amaxA[m] || !p.amaxW[m]) return fail("gemm(fp16x3): missing running-max word");
                g.p[m].amaxA = p.amaxA[m]; g.p[m].amaxW = p.amaxW[m]; g.p[m].amaxC = p.amaxC[m];
            }
        }
        return launch_gemm_split(g, p.nz, stream, tm, stage, f16);
    }
    if (!bf16) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.M = rows; g.N = p.N; g.K = IEF_D; g.lda = IEF_D; g.ldc = p.ldc; g.epi = p.epi; g.alpha = p.alpha; g.qcols = p.qcols;
        for (int m = 0; m < p.nz; ++m) {
            g.p[m].A = p.A32[m]; g.p[m].W = p.W32[m]; g.p[m].bias = p.bias[m]; g.p[m].C = p.C[m]; g.p[m].R = p.R[m];
            g.p[m].C2 = p.C2[m];
        }
        return launch_gemm(g, p.nz, stream, tm, stage);
    }
    GemmBArgs g;
    memset(&g, 0, sizeof(g));
    g.M = rows; g.N = p.N; g.K = IEF_D; g.lda = IEF_D; g.ldc = p.ldc; g.epi = p.epi; g.alpha = p.alpha; g.qcols = p.qcols;
    for (int m = 0; m < p.nz; ++m) {
        g.p[m].A = p.A16[m]; g.p[m].W = p.W16[m]; g.p[m].bias = p.bias[m]; g.p[m].C = p.C[m]; g.p[m].Cb = p.Cb[m];
        g.p[m].R = p.R[m]; g.p[m].C2 = p.C2[m];
    }
    return launch_gemm_b(g, p.nz, stream, tm, stage);
}

// ---- the four row-block launches of the bf16 mode: shared by forward_pass and by the unit entry iefvad_rowblock_unit, so a unit test
// runs the very kernel symbol, grid and LDS size the forward uses at that row count
static int launch_inproj_chain(iefvad_handle* h, int l, const void* const A[2], bool a_fp32, bf16_t* const C[2], int rows, hipStream_t stream, Timer& tm) {
    InProjChainArgs ia;
    memset(&ia, 0, sizeof(ia));
    for (int m = 0; m < 2; ++m) {
        ia.p[m].A = A[m];
        ia.p[m].stream = h->iproj_stream[m][l]; ia.p[m].bias = h->in_b[m][l]; ia.p[m].C = C[m];
    }
    // q is pre-scaled for the softmax by log2(e)/sqrt(96): both attention kernels use exp2
    ia.M = rows; ia.alpha = (1.0f / sqrtf((float)IEF_DH)) * 1.4426950408889634f; ia.wave_stride = (unsigned)wstream_wave_stride_bytes(IC_NPASS);
#ifdef IC_DIAG
    { static unsigned long long* dg = [] { const char* v = getenv("IEFVAD_IC_DIAG_PTR"); return v ? (unsigned long long*)strtoull(v, nullptr, 0) : nullptr; }(); ia.diag = dg; }
#endif
    hipEvent_t e = tm.begin(ST_QKV);
    if (a_fp32) hipLaunchKernelGGL(iefvad_inproj_chain_f32in_kernel, dim3(rows / IC_BM, 2), dim3(512), IC_LDS_BYTES, stream, ia);
    else hipLaunchKernelGGL(iefvad_inproj_chain_bf16_kernel, dim3(rows / IC_BM, 2), dim3(512), IC_LDS_BYTES, stream, ia);
    tm.end(e);
    tm.gemm_launches += 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

static int launch_outproj_ln_chain(iefvad_handle* h, int l, bool whiten, const bf16_t* const A[2], const float* const R[2], float* const y[2],
                                   bf16_t* const yb[2], int rows, hipStream_t stream, Timer& tm) {
    OutLnChainArgs oa;
    memset(&oa, 0, sizeof(oa));
    for (int m = 0; m < 2; ++m) {
        OutLnChainProblem& q = oa.p[m];
        q.A = A[m]; q.stream = h->oproj_stream[m][l]; q.bias = h->out_b[m][l]; q.R = R[m];
        q.g1 = h->norm_w[m][l]; q.b1 = h->norm_b[m][l];
        if (whiten) { q.g2 = h->whiten_w[m]; q.b2 = h->whiten_b[m]; }
        q.y = y[m];
        q.yb = yb[m];
    }
    oa.M = rows; oa.eps = 1e-5f; oa.wave_stride = (unsigned)wstream_wave_stride_bytes();
    { static const int st = [] { const char* v = getenv("IEFVAD_OL_STAGGER"); return v ? atoi(v) : 0; }(); oa.stagger = st; }
#ifdef OC_DIAG
    { static unsigned long long* dg = [] { const char* v = getenv("IEFVAD_OC_DIAG_PTR"); return v ? (unsigned long long*)strtoull(v, nullptr, 0) : nullptr; }(); oa.diag = dg; }
#endif
    // From two blocks per workgroup on: the persistent kernel, one workgroup per CU, the next block's image fetched during the LayerNorm
    // epilogue (outproj_ln_pchain_bf16.h; same bits); IEFVAD_PERSIST=0 keeps the one-block-per-workgroup kernels (A/B).
    // IEFVAD_OUTLN=r (opt-in, round 5's experiment): the LayerNorm in the accumulator registers (outproj_ln_rchain_bf16.h: its own weight
    // streams, any number of blocks per workgroup; bit-identical; epilogue 30 k -> 13 k cycles per block, but its 32 - 64-byte row pieces
    // cost more in memory waits than the park-through-LDS epilogue costs in barriers: 19.8 vs 16.7 ms per step, TRIED.md)
    static const bool rchain = [] { const char* v = getenv("IEFVAD_OUTLN"); return v && v[0] == 'r'; }();
    static const bool persist = [] { const char* v = getenv("IEFVAD_PERSIST"); return !(v && v[0] == '0'); }();
    const int gx = h->num_cus / 2;
    const int nblk = rows / OC_BM;
    hipEvent_t e = tm.begin(ST_OUT);
    if (rchain && persist) {
        for (int m = 0; m < 2; ++m) oa.p[m].stream = h->oproj_stream_r[m][l];
        hipLaunchKernelGGL(iefvad_outproj_ln_rchain_bf16_kernel, dim3(nblk < gx ? nblk : gx, 2), dim3(512), OR_LDS_BYTES, stream, oa);
    } else if (persist && nblk >= 2 * gx && (y[0] != nullptr) == (y[1] != nullptr) && (yb[0] != nullptr) == (yb[1] != nullptr) && (y[0] || yb[0])) {
        if (y[0] && yb[0]) hipLaunchKernelGGL((iefvad_outproj_ln_pchain_bf16_kernel<true, true>), dim3(gx, 2), dim3(512), OP_LDS_BYTES, stream, oa);
        else if (y[0]) hipLaunchKernelGGL((iefvad_outproj_ln_pchain_bf16_kernel<true, false>), dim3(gx, 2), dim3(512), OP_LDS_BYTES, stream, oa);
        else hipLaunchKernelGGL((iefvad_outproj_ln_pchain_bf16_kernel<false, true>), dim3(gx, 2), dim3(512), OP_LDS_BYTES, stream, oa);
    } else
        hipLaunchKernelGGL(iefvad_outproj_ln_chain_bf16_kernel, dim3(rows / OC_BM, 2), dim3(512), OC_LDS_BYTES, stream, oa);
    tm.end(e);
    tm.gemm_launches += 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

// `ha` arrives with its tensors filled in; the stream, scalars and the kernel choice are set here
static int launch_heads_chain(iefvad_handle* h, HeadsChainArgs& ha, int rows, float factor, hipStream_t stream, Timer& tm) {
    for (int m = 0; m < 2; ++m) ha.bias[m] = h->head_b[m];
    ha.stream = h->heads_stream;
    ha.M = rows; ha.factor = factor; ha.eps = h->cfg.epsilon; ha.wave_stride = (unsigned)heads_stream_wave_stride_bytes();
#ifdef HC_DIAG
    { static unsigned long long* dg = [] { const char* v = getenv("IEFVAD_HC_DIAG_PTR"); return v ? (unsigned long long*)strtoull(v, nullptr, 0) : nullptr; }(); ha.diag = dg; }
#endif
    // from two blocks per workgroup on: the persistent kernel (heads_pchain_bf16.h: one workgroup per CU and column third, both
    // images by LDS-DMA under the previous block's epilogue / the first part of phase 2; same bits); IEFVAD_PERSIST=0: A/B
    static const bool persist = [] { const char* v = getenv("IEFVAD_PERSIST"); return !(v && v[0] == '0'); }();
    const int gx = h->num_cus / HC_THIRDS;
    hipEvent_t e = tm.begin(ST_HEAD);
    if (persist && rows / HC_BM >= 2 * gx)
        hipLaunchKernelGGL(iefvad_heads_pchain_bf16_kernel, dim3(gx, HC_THIRDS), dim3(512), HP_LDS_BYTES, stream, ha);
    else
        hipLaunchKernelGGL(iefvad_heads_chain_bf16_kernel, dim3(rows / HC_BM, HC_THIRDS), dim3(512), HC_LDS_BYTES, stream, ha);
    tm.end(e);
    tm.gemm_launches += 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

static int launch_refine_chain(iefvad_handle* h, const float* z_in, float* z_out, float* logits, int rows, hipStream_t stream, Timer& tm) {
    const int K = h->cfg.num_steps;
    ChainArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.z_in = z_in; ca.stream = h->chain_stream; ca.cls_w = h->cls_w; ca.cls_b = h->cls_b;
    ca.z_out = z_out;           // may be z_in (in place): a workgroup reads its 64 rows before it writes them
    ca.logits = logits;
    ca.M = rows; ca.K = K; ca.lambda = h->cfg.lambda_ref; ca.wave_stride = (unsigned)chain_wave_stride_bytes(K);
#ifdef RC_DIAG
    { static unsigned long long* dg = [] { const char* v = getenv("IEFVAD_RC_DIAG_PTR"); return v ? (unsigned long long*)strtoull(v, nullptr, 0) : nullptr; }(); ca.diag = dg; }
#endif
    hipEvent_t e = tm.begin(ST_REFINE);
    hipLaunchKernelGGL(iefvad_refine_chain_bf16_kernel, dim3(rows / RC_BM), dim3(64 * RC_NW), RC_LDS_BYTES, stream, ca);
    tm.end(e);
    tm.gemm_launches += 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

// One pass of the valid rows of whole videos (iefvad_forward_videos): where the packed rows of the pass's chunks come from and
// where its per-row results go.
struct RaggedPass {
    const void* img_rows;            // first packed row of the pass, element type in_dtype
    const void* ev_rows;
    const RaggedChunk* d_chunks;     // device: the pass's chunks (src_row relative to the pass's first packed row)
    const int* d_flags;              // device: NaN flag per (video, modality); nullptr = no nan_to_num
    int valid_rows;                  // packed rows of the pass
    int enc_used_rows;               // rows of the compressed set that belong to chunks
    int enc_rows;                    // > 0: the encoder's row set is row-compressed (RaggedChunk, common.h) and has this many rows (a multiple of 256)
    float* logits;                   // packed outputs at the pass's first row, nullable
    float* w_i_mean;
    float* w_e_mean;
};

// One micro-batch of `nb` chunks.  Dense (rg == nullptr): pi / pe are the pass's [nb, 256, 768] blocks of `in_dtype`, results go
// to `out` at row offset row0.  Ragged: the chunks are built on the device from rg's packed rows, everything behind the encoder
// runs on the valid rows only, results go to rg's packed vectors (`out` must be all-null).
static int forward_pass(iefvad_handle* h, const void* pi_, const void* pe_, int32_t in_dtype, int nb, size_t row0, void* workspace,
                        const iefvad_outputs* out, const RaggedPass* rg, hipStream_t stream, Timer& tm) {
    const iefvad_config& c = h->cfg;
    const bool bf = (c.compute == IEFVAD_COMPUTE_BF16);
    const int L = c.num_layers, K = c.num_steps;
    const int mb = micro_batch(h);
    const size_t D = IEF_D;
    const float factor = (c.noise_model == IEFVAD_NOISE_STUDENT_T) ? (c.nu + 1.0f) / c.nu : 1.0f;   // imf_vad.py:134
    const float qscale = 1.0f / sqrtf((float)IEF_DH);
    {
        const bool enc_rows_mode = rg && rg->enc_rows > 0;       // row-compressed chunks: valid rows + one pad row each
        int rows = enc_rows_mode ? rg->enc_rows : nb * IEF_T;
        const size_t R = (size_t)nb * IEF_T;                     // region stride: the dense capacity
        // workspace regions, in units of R*768 floats: xin 0..2 | qkv 2..8 | att 8..10 | y 10..12 | x 12..14 | logits
        float* ws = (float*)workspace;
        float* xin[2] = {ws, ws + R * D};
        float* qkv[2] = {ws + 2 * R * D, ws + 5 * R * D};
        bf16_t* qkvb[2] = {(bf16_t*)(ws + 2 * R * D), (bf16_t*)(ws + 2 * R * D) + 3 * R * D};   // bf16 mode: q|k|v as bf16
        float* att[2] = {ws + 8 * R * D, ws + 9 * R * D};                       // fp32 mode: attention output
        bf16_t* attb[2] = {(bf16_t*)(ws + 8 * R * D), (bf16_t*)(ws + 8 * R * D) + R * D};   // bf16 mode: the same region
        bf16_t* xb[2] = {(bf16_t*)(ws + 9 * R * D), (bf16_t*)(ws + 9 * R * D) + R * D};     // holds attb | xb
        float* ybuf[2] = {ws + 10 * R * D, ws + 11 * R * D};
        float* xbuf[2] = {ws + 12 * R * D, ws + 13 * R * D};
        float* lg_scratch = ws + 14 * R * D;          // R floats; the 3 R behind it (kWsFloatsPerRow) hold the ragged path's row means
        // tail buffers alias the (dead by then) qkv region: mu_i lv_i mu_e lv_e z | h  (bf16 mode: hb, zb in h's slot)
        float* t0 = ws + 2 * R * D;
        float* mu_i = out->image_mu ? out->image_mu + row0 * D : t0;
        float* lv_i = out->image_logvar ? out->image_logvar + row0 * D : t0 + R * D;
        float* mu_e = out->event_mu ? out->event_mu + row0 * D : t0 + 2 * R * D;
        float* lv_e = out->event_logvar ? out->event_logvar + row0 * D : t0 + 3 * R * D;
        float* z = out->fused ? out->fused + row0 * D : t0 + 4 * R * D;
        float* hbuf = t0 + 5 * R * D;
        bf16_t* hb = (bf16_t*)(t0 + 5 * R * D);
        bf16_t* zb = hb + R * D;
        float* logits = out->logits ? out->logits + row0 : lg_scratch;
        float* wim_out = out->w_i_mean ? out->w_i_mean + row0 : nullptr;
        float* wem_out = out->w_e_mean ? out->w_e_mean + row0 : nullptr;
        if (rg) {                                     // per-row results of the pass land in scratch, then in rg's packed vectors
            logits = lg_scratch;
            wim_out = rg->w_i_mean ? lg_scratch + R : nullptr;
            wem_out = rg->w_e_mean ? lg_scratch + 2 * R : nullptr;
        }

        // bf16 mode, full grids: in_proj on the row-block kernel (inproj_chain_bf16.h); its first layer reads the fp32 rows
        const bool ip_chain = bf && !h->no_inproj_chain && rows % IC_BM == 0 && (rows / IC_BM) * 2 >= h->rowblock_min_wgs;
        // 0. inputs: `.to(torch.float)` (imf_vad.py:41-42); bf16 mode also needs the bf16 operand copy (unless ip_chain)
        const bool need_xb0 = bf && !ip_chain;
        const float* cur[2];
        const char* pi = (const char*)pi_;
        const char* pe = (const char*)pe_;
        if (rg) {
            // the chunker (tools.py:100-114) and the conditional nan_to_num (test.py:90-95) on the device: ragged.h
            if (enc_rows_mode && rg->enc_used_rows < rows) {     // the zero rows that round the set up to whole 256-row tiles
                const size_t o = (size_t)rg->enc_used_rows * D, n = (size_t)(rows - rg->enc_used_rows) * D;
                for (int m = 0; m < 2; ++m) {
                    HIP_TRY(hipMemsetAsync(xin[m] + o, 0, n * sizeof(float), stream));
                    if (need_xb0) HIP_TRY(hipMemsetAsync(xb[m] + o, 0, n * sizeof(bf16_t), stream));
                }
            }
            hipEvent_t e = tm.begin(ST_CAST);
            // (the NaN flags of ALL the call's videos are already set: forward_videos_impl scans every pass's chunks before the
            // first pass runs, because test.py:90-95 decides per whole video and a video may straddle passes)
#define RAGGED_IN(T)                                                                                                        \
    do {                                                                                                                    \
        hipLaunchKernelGGL(iefvad_scatter_rows_kernel<T>, dim3(nb, 2, IEF_RAGGED_SLICES), dim3(256), 0, stream, (const T*)rg->img_rows,         \
                           (const T*)rg->ev_rows, rg->d_chunks, rg->d_flags, xin[0], xin[1], need_xb0 ? xb[0] : (bf16_t*)nullptr, \
                           need_xb0 ? xb[1] : (bf16_t*)nullptr, enc_rows_mode ? 0 : IEF_T);                                  \
    } while (0)
            if (in_dtype == IEFVAD_IN_F32) RAGGED_IN(float);
            else if (in_dtype == IEFVAD_IN_F16) RAGGED_IN(__half);
            else RAGGED_IN(__hip_bfloat16);
#undef RAGGED_IN
            tm.end(e);
            HIP_TRY(hipGetLastError());
            cur[0] = xin[0];
            cur[1] = xin[1];
        } else if (h->row_scale[0] || h->row_scale[1]) {
            // iefvad_forward_scaled: the rows pass through the cast kernel whatever their type, scaled on the way (rowops.h)
            hipEvent_t e = tm.begin(ST_CAST);
            bf16_t* b0p = need_xb0 ? xb[0] : nullptr;
            bf16_t* b1p = need_xb0 ? xb[1] : nullptr;
            const float* s0 = h->row_scale[0] ? h->row_scale[0] + row0 : nullptr;
            const float* s1 = h->row_scale[1] ? h->row_scale[1] + row0 : nullptr;
            int rc = (in_dtype == IEFVAD_IN_F32)   ? launch_cast_scaled<float>(pi, pe, xin[0], xin[1], b0p, b1p, R * D, s0, s1, stream)
                     : (in_dtype == IEFVAD_IN_F16) ? launch_cast_scaled<__half>(pi, pe, xin[0], xin[1], b0p, b1p, R * D, s0, s1, stream)
                                                   : launch_cast_scaled<__hip_bfloat16>(pi, pe, xin[0], xin[1], b0p, b1p, R * D, s0, s1, stream);
            tm.end(e);
            if (rc) return rc;
            cur[0] = xin[0];
            cur[1] = xin[1];
        } else if (in_dtype == IEFVAD_IN_F32) {
            cur[0] = (const float*)pi;
            cur[1] = (const float*)pe;
            if (need_xb0) {
                hipEvent_t e = tm.begin(ST_CAST);
                int rc = launch_cast<float>(pi, pe, nullptr, nullptr, xb[0], xb[1], R * D, 2, stream);
                tm.end(e);
                if (rc) return rc;
            }
        } else {
            hipEvent_t e = tm.begin(ST_CAST);
            bf16_t* b0p = need_xb0 ? xb[0] : nullptr;
            bf16_t* b1p = need_xb0 ? xb[1] : nullptr;
            int rc = (in_dtype == IEFVAD_IN_F16) ? launch_cast<__half>(pi, pe, xin[0], xin[1], b0p, b1p, R * D, 2, stream)
                                                 : launch_cast<__hip_bfloat16>(pi, pe, xin[0], xin[1], b0p, b1p, R * D, 2, stream);
            tm.end(e);
            if (rc) return rc;
            cur[0] = xin[0];
            cur[1] = xin[1];
        }

        // fp16x3: the running-max words of this micro-batch's activations (one word per tensor that feeds a projection)
        const bool splitmb = (c.compute == IEFVAD_COMPUTE_BF16X6 || c.compute == IEFVAD_COMPUTE_FP16X3) &&
                             split_eligible(rows, IEF_D, IEF_D, 1);
        const bool f16mb = splitmb && c.compute == IEFVAD_COMPUTE_FP16X3;
        // activation tensor t owns IEF_AMAX_PARTS words per chunk of the micro-batch
        float* am = h->amax_dev ? h->amax_dev + kAmaxActBase * IEF_AMAX_FLOATS : nullptr;
        const size_t mbs = (size_t)mb * IEF_AMAX_PARTS;
        auto am_t = [&](int t) { return f16mb ? am + mbs * t : nullptr; };
        auto am_in = [&](int m) { return am_t(m); };
        auto am_att = [&](int l, int m) { return am_t(2 + 6 * l + m); };
        auto am_x = [&](int l, int m) { return am_t(2 + 6 * l + 2 + m); };
        auto am_qkv = [&](int l, int m) { return am_t(2 + 6 * l + 4 + m); };
        auto am_z = [&](int k) { return am_t(2 + 6 * L + k); };
        auto am_h = [&](int k) { return am_t(2 + 6 * L + (K + 1) + k); };
        if (f16mb) {
            HIP_TRY(hipMemsetAsync(am, 0, (size_t)amax_act_tensors(L, K) * mbs * sizeof(float), stream));
            hipEvent_t e = tm.begin(ST_CAST);
            hipLaunchKernelGGL(iefvad_amax_chunk_kernel, dim3(nb, 2), dim3(256), 0, stream, cur[0], cur[1], am_in(0), am_in(1));
            tm.end(e);
            HIP_TRY(hipGetLastError());
        }

        // 1. temporal encoder (imf_vad.py:113-123): L x { in_proj, attention, out_proj + residual, LayerNorm }
        for (int l = 0; l < L; ++l) {
            Proj p;
            if (ip_chain) {
                const void* ipA[2] = {(l == 0) ? (const void*)cur[0] : (const void*)xb[0], (l == 0) ? (const void*)cur[1] : (const void*)xb[1]};
                if (int rc = launch_inproj_chain(h, l, ipA, l == 0, qkvb, rows, stream, tm)) return rc;
            }
            memset(&p, 0, sizeof(p));
            p.N = 3 * IEF_D; p.ldc = 3 * IEF_D; p.epi = EPI_QKV; p.qcols = IEF_D; p.nz = 2;
            // q is pre-scaled for the softmax by log2(e)/sqrt(96): both attention kernels use exp2
            p.alpha = qscale * 1.4426950408889634f;
            for (int m = 0; m < 2; ++m) {
                p.A32[m] = cur[m]; p.A16[m] = xb[m]; p.W32[m] = h->in_w[m][l]; p.W16[m] = h->in_wb[m][l]; p.Ws[m] = h->in_ws[m][l];
                p.Wh[m] = h->in_wh[m][l]; p.amaxW[m] = h->in_wa[m][l]; p.amaxA[m] = l == 0 ? am_in(m) : am_x(l - 1, m);
                p.amaxC[m] = am_qkv(l, m);
                p.bias[m] = h->in_b[m][l];
                if (bf) p.Cb[m] = qkvb[m]; else p.C[m] = qkv[m];
            }
            if (!ip_chain)
                if (int rc = launch_proj(p, c.compute, splitmb, rows, stream, tm, ST_QKV)) return rc;

            hipEvent_t e = tm.begin(ST_ATT);
            if (bf) {
                AttnBArgs ab;
                for (int m = 0; m < 2; ++m) { ab.qkv[m] = qkvb[m]; ab.out[m] = attb[m]; }
                ab.nchunks = nb;
                ab.chunks = enc_rows_mode ? rg->d_chunks : nullptr;
                ab.head_major = ip_chain ? 1 : 0;
                ab.nrows = rows;
                // from two items per CU on: the persistent kernel (attention_pbf16.h: one 8-wave workgroup per CU, K / V staged once for
                // both query halves by LDS-DMA, the next item's K in flight under the current item); bit-identical to the one below
                static const bool persist_att = [] { const char* v = getenv("IEFVAD_PERSIST"); return !(v && v[0] == '0'); }();      // one switch for the three persistent kernels
                const int items = 2 * nb * IEF_H;
                if (persist_att && items >= 2 * h->num_cus) {
                    if (enc_rows_mode) hipLaunchKernelGGL(iefvad_attention_pbf16_rows_kernel, dim3(h->num_cus), dim3(512), APB_LDS_BYTES, stream, ab);
                    else hipLaunchKernelGGL(iefvad_attention_pbf16_kernel, dim3(h->num_cus), dim3(512), APB_LDS_BYTES, stream, ab);
                } else if (enc_rows_mode) hipLaunchKernelGGL(iefvad_attention_bf16_rows_kernel, dim3(IEF_H, 2, 2 * nb), dim3(256), 0, stream, ab);
                else hipLaunchKernelGGL(iefvad_attention_bf16_kernel, dim3(IEF_H, 2, 2 * nb), dim3(256), 0, stream, ab);
            } else {
                AttnArgs aa;
                memset(&aa, 0, sizeof(aa));
                for (int m = 0; m < 2; ++m) { aa.qkv[m] = qkv[m]; aa.out[m] = att[m]; }
                aa.nchunks = nb;
                aa.chunks = enc_rows_mode ? rg->d_chunks : nullptr;
                // bf16x6: the split attention kernel goes with the split projections (same batch-size rule), so a small
                // batch is computed exactly as in the f32 mode
                for (int m = 0; m < 2; ++m) { aa.amax[m] = am_att(l, m); aa.amax_in[m] = am_qkv(l, m); }
                if (f16mb)
                    hipLaunchKernelGGL(iefvad_attention_split_f16_kernel, dim3(IEF_H, 2, 2 * nb), dim3(256), ATS_LDS_BYTES, stream, aa);
                else if (splitmb && enc_rows_mode)
                    hipLaunchKernelGGL(iefvad_attention_split_rows_kernel, dim3(IEF_H, 2, 2 * nb), dim3(256), ATS_LDS_BYTES, stream, aa);
                else if (splitmb)
                    hipLaunchKernelGGL(iefvad_attention_split_kernel, dim3(IEF_H, 2, 2 * nb), dim3(256), ATS_LDS_BYTES, stream, aa);
                else if (enc_rows_mode)
                    hipLaunchKernelGGL(iefvad_attention_f32_rows_kernel, dim3(IEF_H, 2, 2 * nb), dim3(256), 0, stream, aa);
                else
                    hipLaunchKernelGGL(iefvad_attention_f32_kernel, dim3(IEF_H, 2, 2 * nb), dim3(256), 0, stream, aa);
            }
            tm.end(e);
            HIP_TRY(hipGetLastError());

            // out_proj + residual + LayerNorm(s) in one row-owning kernel (bf16 mode, full grids): 64-row blocks on the refinement chain's
            // structure (outproj_ln_chain_bf16.h); same bits as the GEMM + LayerNorm kernels
            const bool ln_fused = bf && !h->no_ln_fusion && rows % OC_BM == 0 && (rows / OC_BM) * 2 >= h->rowblock_min_wgs;
            if (ln_fused) {
                const bf16_t* oA[2] = {attb[0], attb[1]};
                float* oy[2] = {(l < L - 1) ? xbuf[0] : nullptr, (l < L - 1) ? xbuf[1] : nullptr};      // fp32 rows are only the next layer's residual
                if (int rc = launch_outproj_ln_chain(h, l, l == L - 1, oA, cur, oy, xb, rows, stream, tm)) return rc;
                cur[0] = xbuf[0];
                cur[1] = xbuf[1];
                continue;
            }

            memset(&p, 0, sizeof(p));
            p.N = IEF_D; p.ldc = IEF_D; p.epi = EPI_BIAS_RESID; p.nz = 2;
            for (int m = 0; m < 2; ++m) {
                p.A32[m] = att[m]; p.A16[m] = attb[m]; p.W32[m] = h->out_w[m][l]; p.W16[m] = h->out_wb[m][l]; p.Ws[m] = h->out_ws[m][l];
                p.Wh[m] = h->out_wh[m][l]; p.amaxW[m] = h->out_wa[m][l]; p.amaxA[m] = am_att(l, m);
                p.bias[m] = h->out_b[m][l]; p.C[m] = ybuf[m]; p.R[m] = cur[m];
            }
            if (int rc = launch_proj(p, c.compute, splitmb, rows, stream, tm, ST_OUT)) return rc;

            LnArgs la;
            memset(&la, 0, sizeof(la));
            la.nrows = rows; la.eps = 1e-5f;
            for (int m = 0; m < 2; ++m) {
                la.x[m] = ybuf[m]; la.g1[m] = h->norm_w[m][l]; la.b1[m] = h->norm_b[m][l];
                if (l == L - 1) { la.g2[m] = h->whiten_w[m]; la.b2[m] = h->whiten_b[m]; }   // whitening LN, :117,:123
                // fp32 mode: x feeds both the next projection and the next residual; bf16 mode: the bf16 copy feeds the
                // projection, the fp32 tensor is only the next layer's residual (not needed after the last layer)
                la.y[m] = (!bf || l < L - 1) ? xbuf[m] : nullptr;
                la.yb[m] = bf ? xb[m] : nullptr;
                la.amax[m] = am_x(l, m);
            }
            e = tm.begin(ST_LN);
            hipLaunchKernelGGL(iefvad_layernorm_kernel, dim3((rows + ROW_WAVES - 1) / ROW_WAVES, 2), dim3(256), 0, stream, la);
            tm.end(e);
            HIP_TRY(hipGetLastError());
            cur[0] = xbuf[0];
            cur[1] = xbuf[1];
        }

        // Ragged pass with whole chunks in the encoder (IEFVAD_DENSE_ENCODER=1): everything behind the encoder is row-wise
        // (imf_vad.py:125-150) and the reference slices the pad rows away (test.py:121), so the valid rows of the last
        // LayerNorm's output are gathered (packed order, padded with zero rows to whole 256-row tiles) and `rows` shrinks to
        // that count from here on.  fp16x3 keeps whole chunks: its operand scales are per chunk.  The compact operands live in
        // the attention-output region, dead by now.  (Row-compressed chunks, the default: the tail runs on the encoder's row set.)
        const float* xt[2] = {xbuf[0], xbuf[1]};      // the tail's fp32 / bf16 A operands
        const bf16_t* xtb[2] = {xb[0], xb[1]};
        bool compacted = false;
        if (rg && !enc_rows_mode && c.compute != IEFVAD_COMPUTE_FP16X3) {
            const int mc = (rg->valid_rows + 255) / 256 * 256;
            if (mc < rows) {
                CompactArgs ca;
                memset(&ca, 0, sizeof(ca));
                for (int m = 0; m < 2; ++m) {
                    if (bf) { ca.xb[m] = xb[m]; ca.xcb[m] = attb[m]; xtb[m] = attb[m]; }
                    else { ca.x[m] = xbuf[m]; ca.xc[m] = att[m]; xt[m] = att[m]; }
                }
                ca.chunks = rg->d_chunks;
                const size_t tail0 = (size_t)rg->valid_rows * D, tailn = (size_t)(mc - rg->valid_rows) * D;
                if (tailn)
                    for (int m = 0; m < 2; ++m)
                        HIP_TRY(bf ? hipMemsetAsync(attb[m] + tail0, 0, tailn * sizeof(bf16_t), stream)
                                   : hipMemsetAsync(att[m] + tail0, 0, tailn * sizeof(float), stream));
                hipEvent_t e = tm.begin(ST_CAST);
                hipLaunchKernelGGL(iefvad_compact_rows_kernel, dim3(nb, 2), dim3(256), 0, stream, ca);
                tm.end(e);
                HIP_TRY(hipGetLastError());
                rows = mc;
                compacted = true;
            }
        }
        const bool tail_split = splitmb && (!compacted || split_eligible(rows, IEF_D, IEF_D, 1));   // bf16x6: a small compact set runs on the fp32 kernels

        // 2 + 3 in one kernel (bf16 mode, full grids): heads of both modalities + fusion on the row-block structure (heads_chain_bf16.h).
        // The four head tensors are stored only if the caller asked for them.
        const bool heads_rows = bf && !h->no_heads_fusion && h->heads_stream && rows % HC_BM == 0 && (rows / HC_BM) * HC_THIRDS >= h->rowblock_min_wgs;
        const bool heads_fused = heads_rows;
        // 4 + 5 in one kernel (bf16 mode): the K refinement steps and the scorer with the state on chip, refine_chain_bf16.h
        const bool chain = bf && K > 0 && !h->no_chain && h->chain_stream && rows % RC_BM == 0 && rows / RC_BM >= h->chain_min_blocks;
        if (heads_rows) {
            HeadsChainArgs ha;
            memset(&ha, 0, sizeof(ha));
            for (int m = 0; m < 2; ++m) ha.A[m] = xtb[m];
            ha.mu[0] = out->image_mu ? mu_i : nullptr;
            ha.lv[0] = out->image_logvar ? lv_i : nullptr;
            ha.mu[1] = out->event_mu ? mu_e : nullptr;
            ha.lv[1] = out->event_logvar ? lv_e : nullptr;
            ha.n[0] = out->w_i ? out->w_i + row0 * D : nullptr;
            ha.n[1] = out->w_e ? out->w_e + row0 * D : nullptr;
            ha.z = z;
            ha.zb = chain ? nullptr : zb;
            const bool means = wim_out || wem_out;
            ha.nsum_part = means ? ybuf[0] : nullptr;        // y is dead after the last LayerNorm: 48 of its 768 floats per row
            if (int rc = launch_heads_chain(h, ha, rows, factor, stream, tm)) return rc;
            hipEvent_t e;
            if (means) {
                e = tm.begin(ST_FUSION);
                hipLaunchKernelGGL(iefvad_rowmean_finish_kernel, dim3((2 * rows + 255) / 256), dim3(256), 0, stream, ha.nsum_part,
                                   wim_out, wem_out, rows, HC_NPART);
                tm.end(e);
                HIP_TRY(hipGetLastError());
            }
        }

        // 2. mu / logvar heads (imf_vad.py:125-128): one [768 -> 1536] projection per modality
        if (!heads_fused) {
            Proj p;
            memset(&p, 0, sizeof(p));
            p.N = 2 * IEF_D; p.ldc = IEF_D; p.epi = EPI_HEADS; p.nz = 2;
            for (int m = 0; m < 2; ++m) {
                p.A32[m] = xt[m]; p.A16[m] = xtb[m]; p.W32[m] = h->head_w[m]; p.W16[m] = h->head_wb[m]; p.Ws[m] = h->head_ws[m]; p.bias[m] = h->head_b[m];
                p.Wh[m] = h->head_wh[m]; p.amaxW[m] = h->head_wa[m]; p.amaxA[m] = am_x(L - 1, m);
            }
            p.C[0] = mu_i; p.C2[0] = lv_i; p.C[1] = mu_e; p.C2[1] = lv_e;
            if (int rc = launch_proj(p, c.compute, tail_split, rows, stream, tm, ST_HEAD)) return rc;
        }

        // 3. precision weights + fusion (imf_vad.py:130-144), fp32 in both modes
        if (!heads_fused) {
            FusionArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.mu_i = mu_i; fa.lv_i = lv_i; fa.mu_e = mu_e; fa.lv_e = lv_e;
            fa.n_i = out->w_i ? out->w_i + row0 * D : nullptr;
            fa.n_e = out->w_e ? out->w_e + row0 * D : nullptr;
            fa.z = z;
            fa.zb = (bf && !chain) ? zb : nullptr;
            fa.n_i_mean = wim_out;
            fa.n_e_mean = wem_out;
            fa.nrows = rows; fa.factor = factor; fa.eps = c.epsilon;
            fa.z_amax = am_z(0);
            hipEvent_t e = tm.begin(ST_FUSION);
            hipLaunchKernelGGL(iefvad_fusion_kernel, dim3((rows + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, stream, fa);
            tm.end(e);
            HIP_TRY(hipGetLastError());
        }

        if (chain)
            if (int rc = launch_refine_chain(h, z, out->fused ? z : nullptr, logits, rows, stream, tm)) return rc;

        // 4. K refinement steps z <- z - lambda * (W2 relu(W1 z + b1) + b2) (imf_vad.py:146-149); the state z stays fp32
        for (int k = 0; k < K && !chain; ++k) {
            Proj p;
            memset(&p, 0, sizeof(p));
            p.N = IEF_D; p.ldc = IEF_D; p.epi = EPI_BIAS_RELU; p.nz = 1;
            p.A32[0] = z; p.A16[0] = zb; p.W32[0] = h->ref_w1[k]; p.W16[0] = h->ref_w1b[k]; p.Ws[0] = h->ref_w1s[k]; p.bias[0] = h->ref_b1[k];
            p.Wh[0] = h->ref_w1h[k]; p.amaxW[0] = h->ref_w1a[k]; p.amaxA[0] = am_z(k); p.amaxC[0] = am_h(k);
            p.C[0] = bf ? nullptr : hbuf; p.Cb[0] = bf ? hb : nullptr;
            if (int rc = launch_proj(p, c.compute, tail_split, rows, stream, tm, ST_REFINE)) return rc;
            memset(&p, 0, sizeof(p));
            p.N = IEF_D; p.ldc = IEF_D; p.epi = EPI_REFINE; p.alpha = c.lambda_ref; p.nz = 1;
            p.A32[0] = hbuf; p.A16[0] = hb; p.W32[0] = h->ref_w2[k]; p.W16[0] = h->ref_w2b[k]; p.Ws[0] = h->ref_w2s[k]; p.bias[0] = h->ref_b2[k];
            p.Wh[0] = h->ref_w2h[k]; p.amaxW[0] = h->ref_w2a[k]; p.amaxA[0] = am_h(k); p.amaxC[0] = am_z(k + 1);
            p.C[0] = z; p.R[0] = z; p.Cb[0] = (bf && k + 1 < K) ? zb : nullptr;
            if (int rc = launch_proj(p, c.compute, tail_split, rows, stream, tm, ST_REFINE)) return rc;
        }

        // 5. scorer (imf_vad.py:150)
        if (!chain) {
            hipEvent_t e = tm.begin(ST_SCORER);
            hipLaunchKernelGGL(iefvad_scorer_kernel, dim3((rows + ROW_WAVES - 1) / ROW_WAVES), dim3(256), 0, stream, z,
                               h->cls_w, h->cls_b, logits, rows);
            tm.end(e);
            HIP_TRY(hipGetLastError());
        }

        // ragged pass: the valid rows' results -> the caller's packed vectors (test.py:119-121: logits1[0:len_cur])
        if (rg) {
            hipEvent_t e = tm.begin(ST_SCORER);
            if (compacted)
                hipLaunchKernelGGL(iefvad_rows_out_kernel, dim3((rg->valid_rows + 255) / 256), dim3(256), 0, stream, logits, wim_out, wem_out,
                                   rg->logits, rg->w_i_mean, rg->w_e_mean, (const RaggedChunk*)nullptr, rg->valid_rows);
            else
                hipLaunchKernelGGL(iefvad_rows_out_kernel, dim3(nb), dim3(256), 0, stream, logits, wim_out, wem_out, rg->logits,
                                   rg->w_i_mean, rg->w_e_mean, rg->d_chunks, rg->valid_rows);
            tm.end(e);
            HIP_TRY(hipGetLastError());
        }
    }
    return 0;
}

static int forward_impl(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B, void* workspace,
                        size_t workspace_bytes, const iefvad_outputs* out, hipStream_t stream, Timer& tm) {
    if (!h || !img || !ev || !out) return fail("iefvad_forward: null argument");
    if (!h->weights_set) return fail("iefvad_forward: weights not set");
    if (B <= 0) return fail("iefvad_forward: B must be positive (got %d)", B);
    if (in_dtype != IEFVAD_IN_F32 && in_dtype != IEFVAD_IN_F16 && in_dtype != IEFVAD_IN_BF16)
        return fail("iefvad_forward: unknown in_dtype %d", in_dtype);
    if (!workspace || workspace_bytes < iefvad_workspace_bytes(h, B))
        return fail("iefvad_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, iefvad_workspace_bytes(h, B));
    if (((uintptr_t)workspace & 15) || ((uintptr_t)img & 15) || ((uintptr_t)ev & 15))
        return fail("iefvad_forward: buffers must be 16-byte aligned");
    const int mb = micro_batch(h);
    for (int b0 = 0; b0 < B; b0 += mb) {
        const int nb = (B - b0 < mb) ? (B - b0) : mb;
        const size_t row0 = (size_t)b0 * IEF_T;
        const size_t in_off = row0 * IEF_D * in_elem_bytes(in_dtype);
        if (int rc = forward_pass(h, (const char*)img + in_off, (const char*)ev + in_off, in_dtype, nb, row0, workspace, out, nullptr,
                                  stream, tm))
            return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// whole videos: valid rows in, per-snippet results out (include/iefvad.h, csrc/ragged.h)
// ------------------------------------------------------------------------------------------------
// chunk count of a video as the evaluation loop needs it: process_split's len // 256 + 1 chunks (tools.py:105-112) minus the
// all-zero one of a len % 256 == 0 video, whose rows test.py:121 slices away
static int video_chunks(int n) { return n < IEF_T ? 1 : n / IEF_T + (n % IEF_T ? 1 : 0); }

// per-call metadata (chunk table, NaN flags) travels through a small ring of pinned host / device buffer pairs: the copy is
// asynchronous, so a slot is reused only after the event recorded behind its last use has completed
struct MetaRing {
    static const int kSlots = 4;
    void* host[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    void* dev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[kSlots] = {0, 0, 0, 0};
    hipEvent_t done[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    int turn = 0;
};

static void release_meta(iefvad_handle* h) {
    if (!h->meta) return;
    for (int i = 0; i < MetaRing::kSlots; ++i) {
        if (h->meta->host[i]) (void)hipHostFree(h->meta->host[i]);
        if (h->meta->dev[i]) (void)hipFree(h->meta->dev[i]);
        if (h->meta->done[i]) (void)hipEventDestroy(h->meta->done[i]);
    }
    delete h->meta;
    h->meta = nullptr;
}

static int videos_layout(const int32_t* lengths, int32_t nvideos, long long* total_rows, long long* total_chunks) {
    long long rows = 0, chunks = 0;
    for (int v = 0; v < nvideos; ++v) {
        if (lengths[v] <= 0) return fail("iefvad_forward_videos: lengths[%d] = %d", v, lengths[v]);
        rows += lengths[v];
        chunks += video_chunks(lengths[v]);
    }
    *total_rows = rows;
    *total_chunks = chunks;
    return 0;
}

extern "C" size_t iefvad_videos_workspace_bytes(const iefvad_handle* h, const int32_t* lengths, int32_t nvideos) {
    if (!h || !lengths || nvideos <= 0) return 0;
    long long rows, chunks;
    if (videos_layout(lengths, nvideos, &rows, &chunks) || chunks > 0x7fffffffLL) return 0;
    return iefvad_workspace_bytes(h, (int32_t)chunks);
}

static int forward_videos_impl(iefvad_handle* h, const void* img_rows, const void* ev_rows, int32_t in_dtype, const int32_t* lengths,
                               int32_t nvideos, int32_t nan_to_num, void* workspace, size_t workspace_bytes, float* logits,
                               float* w_i_mean, float* w_e_mean, hipStream_t stream, Timer& tm) {
    if (!h || !img_rows || !ev_rows || !lengths || !logits) return fail("iefvad_forward_videos: null argument");
    if (!h->weights_set) return fail("iefvad_forward_videos: weights not set");
    if (nvideos <= 0) return fail("iefvad_forward_videos: nvideos must be positive (got %d)", nvideos);
    if (in_dtype != IEFVAD_IN_F32 && in_dtype != IEFVAD_IN_F16 && in_dtype != IEFVAD_IN_BF16)
        return fail("iefvad_forward_videos: unknown in_dtype %d", in_dtype);
    long long total_rows, total_chunks;
    if (int rc = videos_layout(lengths, nvideos, &total_rows, &total_chunks)) return rc;
    if (total_chunks > 0x7fffffffLL / IEF_T) return fail("iefvad_forward_videos: too many chunks");
    const size_t need = iefvad_workspace_bytes(h, (int32_t)total_chunks);
    if (!workspace || workspace_bytes < need) return fail("iefvad_forward_videos: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    if (((uintptr_t)workspace & 15) || ((uintptr_t)img_rows & 15) || ((uintptr_t)ev_rows & 15))
        return fail("iefvad_forward_videos: buffers must be 16-byte aligned");

    // ---- metadata: the chunk table of the whole call (src_row relative to its pass, filled below) + the flag words
    const size_t chunk_bytes = (size_t)total_chunks * sizeof(RaggedChunk);
    const size_t flag_off = (chunk_bytes + 255) & ~(size_t)255;
    const size_t flag_bytes = nan_to_num ? (size_t)nvideos * 2 * sizeof(int) : 0;
    const size_t meta_bytes = flag_off + flag_bytes;
    if (!h->meta) {
        h->meta = new (std::nothrow) MetaRing();
        if (!h->meta) return fail("iefvad_forward_videos: out of host memory");
    }
    MetaRing& mr = *h->meta;
    const int slot = mr.turn;
    mr.turn = (mr.turn + 1) % MetaRing::kSlots;
    HIP_TRY(hipSetDevice(h->device));
    if (mr.done[slot]) HIP_TRY(hipEventSynchronize(mr.done[slot]));
    else HIP_TRY(hipEventCreateWithFlags(&mr.done[slot], hipEventDisableTiming));
    if (mr.cap[slot] < meta_bytes) {
        if (mr.host[slot]) (void)hipHostFree(mr.host[slot]);
        if (mr.dev[slot]) (void)hipFree(mr.dev[slot]);
        mr.host[slot] = mr.dev[slot] = nullptr;
        mr.cap[slot] = 0;
        const size_t cap = meta_bytes * 2 + 4096;
        HIP_TRY(hipHostMalloc(&mr.host[slot], cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&mr.dev[slot], cap));
        mr.cap[slot] = cap;
    }
    RaggedChunk* hc = (RaggedChunk*)mr.host[slot];
    const int mb = micro_batch(h);
    // fp16x3 carries one operand scale per 256-row chunk of the row set: it keeps whole chunks
    const bool enc_rows_mode = !h->dense_encoder && h->cfg.compute != IEFVAD_COMPUTE_FP16X3;
    {
        // chunk table; src_row and enc_row are relative to the chunk's PASS (passes are runs of <= mb chunks)
        long long row = 0, pass_row0 = 0;
        long long ci = 0;
        int enc = 0;
        for (int v = 0; v < nvideos; ++v) {
            const int n = lengths[v], nch = video_chunks(n);
            for (int j = 0; j < nch; ++j, ++ci) {
                if (ci % mb == 0) { pass_row0 = row; enc = 0; }
                const int valid = (n - j * IEF_T) < IEF_T ? (n - j * IEF_T) : IEF_T;
                hc[ci].src_row = (int)(row - pass_row0);
                hc[ci].valid = valid;
                hc[ci].video = v;
                hc[ci].enc_row = enc;
                enc += enc_rows_mode ? ragged_rows(valid) : IEF_T;
                row += valid;
            }
        }
        if (flag_bytes) memset((char*)mr.host[slot] + flag_off, 0, flag_bytes);
    }
    HIP_TRY(hipMemcpyAsync(mr.dev[slot], mr.host[slot], meta_bytes, hipMemcpyHostToDevice, stream));
    const RaggedChunk* dc = (const RaggedChunk*)mr.dev[slot];
    const int* dflags = flag_bytes ? (const int*)((const char*)mr.dev[slot] + flag_off) : nullptr;

    iefvad_outputs none;
    memset(&none, 0, sizeof(none));
    const size_t esz = in_elem_bytes(in_dtype);
    long long row0 = 0;
    int rc = 0;
    if (dflags) {
        // The conditional nan_to_num of test.py:90-95 is decided on the WHOLE video tensor, and a video may straddle micro-batch
        // passes: every chunk of the call is scanned before the first pass lays out (and fixes up) its rows.  src_row is
        // pass-relative, so the scan goes pass by pass too, with the pass's base pointers.
        hipEvent_t e = tm.begin(ST_CAST);
        long long r0 = 0;
        for (long long c0 = 0; c0 < total_chunks; c0 += mb) {
            const int nb = (int)((total_chunks - c0 < mb) ? (total_chunks - c0) : mb);
            const void* pi = (const char*)img_rows + (size_t)r0 * IEF_D * esz;
            const void* pe = (const char*)ev_rows + (size_t)r0 * IEF_D * esz;
            if (in_dtype == IEFVAD_IN_F32)
                hipLaunchKernelGGL(iefvad_nanflag_kernel<float>, dim3(nb, 2, IEF_RAGGED_SLICES), dim3(256), 0, stream, (const float*)pi, (const float*)pe, dc + c0, (int*)dflags);
            else if (in_dtype == IEFVAD_IN_F16)
                hipLaunchKernelGGL(iefvad_nanflag_kernel<__half>, dim3(nb, 2, IEF_RAGGED_SLICES), dim3(256), 0, stream, (const __half*)pi, (const __half*)pe, dc + c0, (int*)dflags);
            else
                hipLaunchKernelGGL(iefvad_nanflag_kernel<__hip_bfloat16>, dim3(nb, 2, IEF_RAGGED_SLICES), dim3(256), 0, stream, (const __hip_bfloat16*)pi, (const __hip_bfloat16*)pe, dc + c0, (int*)dflags);
            for (int j = 0; j < nb; ++j) r0 += hc[c0 + j].valid;
        }
        tm.end(e);
        HIP_TRY(hipGetLastError());
    }
    for (long long c0 = 0; c0 < total_chunks && !rc; c0 += mb) {
        const int nb = (int)((total_chunks - c0 < mb) ? (total_chunks - c0) : mb);
        long long vrows = 0;
        for (int j = 0; j < nb; ++j) vrows += hc[c0 + j].valid;
        RaggedPass rg;
        rg.img_rows = (const char*)img_rows + (size_t)row0 * IEF_D * esz;
        rg.ev_rows = (const char*)ev_rows + (size_t)row0 * IEF_D * esz;
        rg.d_chunks = dc + c0;
        rg.d_flags = dflags;
        rg.valid_rows = (int)vrows;
        rg.enc_used_rows = rg.enc_rows = 0;
        if (enc_rows_mode) {
            rg.enc_used_rows = hc[c0 + nb - 1].enc_row + ragged_rows(hc[c0 + nb - 1].valid);
            rg.enc_rows = (rg.enc_used_rows + IEF_T - 1) / IEF_T * IEF_T;        // <= nb * 256
        }
        rg.logits = logits + row0;
        rg.w_i_mean = w_i_mean ? w_i_mean + row0 : nullptr;
        rg.w_e_mean = w_e_mean ? w_e_mean + row0 : nullptr;
        rc = forward_pass(h, nullptr, nullptr, in_dtype, nb, 0, workspace, &none, &rg, stream, tm);
        row0 += vrows;
    }
    (void)hipEventRecord(mr.done[slot], stream);       // the slot's buffers are free once everything enqueued above has run
    return rc;
}

extern "C" int iefvad_forward_videos(iefvad_handle* h, const void* img_rows, const void* ev_rows, int32_t in_dtype,
                                     const int32_t* lengths, int32_t nvideos, int32_t nan_to_num, void* workspace,
                                     size_t workspace_bytes, float* logits, float* w_i_mean, float* w_e_mean, void* stream) {
    Timer tm;
    return forward_videos_impl(h, img_rows, ev_rows, in_dtype, lengths, nvideos, nan_to_num, workspace, workspace_bytes, logits, w_i_mean,
                               w_e_mean, (hipStream_t)stream, tm);
}


// ------------------------------------------------------------------------------------------------
// small batches: replay a captured hipGraph of the forward
// ------------------------------------------------------------------------------------------------
// The reference calls the model once per video (test.py:76-117): B = 1 .. a few chunks, ~31 kernels of a few microseconds
// each.  Launched one by one they are host-bound (3-5 us of enqueue per launch).  For B <= cfg.graph_chunks the library
// captures forward_impl once per (B, in_dtype, output set, workspace) on a private stream, with every caller-owned
// pointer replaced by a library-owned staging buffer, and a call becomes: two device-to-device copies of the inputs,
// ONE hipGraphLaunch on the caller's stream, and one copy per requested output.  Same kernels, same order: same bits.
static const int kGraphDefaultChunks = 8, kGraphMaxChunks = 32, kGraphMaxEntries = 48;

struct GraphEntry {
    int B, in_dtype;
    unsigned outmask;
    void* workspace;
    size_t workspace_bytes;
    hipGraphExec_t exec;
    unsigned long long last_use;
};

struct GraphCache {
    hipStream_t cap_stream = nullptr;
    char* in_buf = nullptr;      // img | ev staging, sized for max_chunks at 4 bytes per element
    float* small_buf = nullptr;  // logits | w_i_mean | w_e_mean, max_chunks * T floats each
    float* big_buf = nullptr;    // the seven [B*T, D] outputs (allocated on the first call that asks for one)
    int max_chunks = 0;
    unsigned long long clock = 0;
    bool disabled = false;       // capture or instantiation failed once: every later call launches directly
    std::vector<GraphEntry> entries;
};

static void release_graphs(iefvad_handle* h) {
    if (!h->graphs) return;
    for (auto& e : h->graphs->entries) (void)hipGraphExecDestroy(e.exec);
    if (h->graphs->cap_stream) (void)hipStreamDestroy(h->graphs->cap_stream);
    if (h->graphs->in_buf) (void)hipFree(h->graphs->in_buf);
    if (h->graphs->small_buf) (void)hipFree(h->graphs->small_buf);
    if (h->graphs->big_buf) (void)hipFree(h->graphs->big_buf);
    delete h->graphs;
    h->graphs = nullptr;
}

static int graph_limit(const iefvad_handle* h) {
    const int g = h->cfg.graph_chunks;
    if (g < 0) return 0;
    const int lim = g == 0 ? kGraphDefaultChunks : (g < kGraphMaxChunks ? g : kGraphMaxChunks);
    const int mb = micro_batch(h);
    return lim < mb ? lim : mb;          // a graphed call is a single micro-batch
}

static int forward_graphed(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B, void* workspace,
                           size_t workspace_bytes, const iefvad_outputs* out, hipStream_t stream) {
    static_assert(sizeof(iefvad_outputs) == 10 * sizeof(float*), "iefvad_outputs is ten pointers");
    const size_t D = IEF_D, T = IEF_T;
    if (!h->graphs) {
        h->graphs = new (std::nothrow) GraphCache();
        if (!h->graphs) return fail("iefvad_forward: out of host memory");
    }
    GraphCache& gc = *h->graphs;
    if (!gc.cap_stream) {
        HIP_TRY(hipSetDevice(h->device));
        gc.max_chunks = graph_limit(h);
        HIP_TRY(hipStreamCreateWithFlags(&gc.cap_stream, hipStreamNonBlocking));
        HIP_TRY(hipMalloc((void**)&gc.in_buf, 2 * (size_t)gc.max_chunks * T * D * 4));
        HIP_TRY(hipMalloc((void**)&gc.small_buf, 3 * (size_t)gc.max_chunks * T * sizeof(float)));
    }
    const size_t rows = (size_t)B * T, mrows = (size_t)gc.max_chunks * T;
    float* const* of = (float* const*)out;               // fused logits image_mu event_mu image_logvar event_logvar w_i w_e w_i_mean w_e_mean
    static const bool kBig[10] = {true, false, true, true, true, true, true, true, false, false};
    unsigned outmask = 0;
    bool any_big = false;
    for (int i = 0; i < 10; ++i)
        if (of[i]) { outmask |= 1u << i; any_big |= kBig[i]; }
    if (any_big && !gc.big_buf) HIP_TRY(hipMalloc((void**)&gc.big_buf, 7 * mrows * D * sizeof(float)));
    // staging addresses (fixed for the life of the handle, so every captured graph stays valid)
    char* s_img = gc.in_buf;
    char* s_ev = gc.in_buf + mrows * D * 4;
    float* s_out[10];
    {
        int big = 0, small = 0;
        for (int i = 0; i < 10; ++i)
            s_out[i] = kBig[i] ? gc.big_buf + (size_t)(big++) * mrows * D : gc.small_buf + (size_t)(small++) * mrows;
    }
    GraphEntry* hit = nullptr;
    for (auto& e : gc.entries)
        if (e.B == B && e.in_dtype == in_dtype && e.outmask == outmask && e.workspace == workspace && e.workspace_bytes == workspace_bytes) hit = &e;
    if (!hit) {
        iefvad_outputs so;
        float** sf = (float**)&so;
        for (int i = 0; i < 10; ++i) sf[i] = of[i] ? s_out[i] : nullptr;
        // The graph is an optimisation: if capture or instantiation is refused (a capture already active on this thread, an
        // exhausted graph pool ...), remember it and let the caller's direct path run instead.
        Timer tm;
        if (hipStreamBeginCapture(gc.cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            gc.disabled = true;
            return -1;
        }
        const int rc = forward_impl(h, s_img, s_ev, in_dtype, B, workspace, workspace_bytes, &so, gc.cap_stream, tm);
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(gc.cap_stream, &graph);      // always end the capture, also after a failed launch
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        hipGraphExec_t exec = nullptr;
        hipError_t ie = (ce == hipSuccess && graph) ? hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) : hipErrorUnknown;
        if (graph) (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) {
            (void)hipGetLastError();
            gc.disabled = true;
            return -1;
        }
        if ((int)gc.entries.size() >= kGraphMaxEntries) {                      // evict the least recently used graph
            size_t lru = 0;
            for (size_t i = 1; i < gc.entries.size(); ++i)
                if (gc.entries[i].last_use < gc.entries[lru].last_use) lru = i;
            (void)hipGraphExecDestroy(gc.entries[lru].exec);
            gc.entries.erase(gc.entries.begin() + (long)lru);
        }
        gc.entries.push_back(GraphEntry{B, in_dtype, outmask, workspace, workspace_bytes, exec, 0});
        hit = &gc.entries.back();
    }
    hit->last_use = ++gc.clock;
    const size_t in_bytes = rows * D * in_elem_bytes(in_dtype);
    HIP_TRY(hipMemcpyAsync(s_img, img, in_bytes, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(s_ev, ev, in_bytes, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipGraphLaunch(hit->exec, stream));
    for (int i = 0; i < 10; ++i)
        if (of[i]) HIP_TRY(hipMemcpyAsync(of[i], s_out[i], rows * (kBig[i] ? D : 1) * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return 0;
}

extern "C" int iefvad_forward(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B, void* workspace,
                              size_t workspace_bytes, const iefvad_outputs* out, void* stream) {
    if (h && out && img && ev && h->weights_set && B > 0 && B <= graph_limit(h) && workspace &&
        workspace_bytes >= iefvad_workspace_bytes(h, B) && !(((uintptr_t)workspace | (uintptr_t)img | (uintptr_t)ev) & 15) &&
        (in_dtype == IEFVAD_IN_F32 || in_dtype == IEFVAD_IN_F16 || in_dtype == IEFVAD_IN_BF16))
    {
        if (!(h->graphs && h->graphs->disabled)) {
            const int rc = forward_graphed(h, img, ev, in_dtype, B, workspace, workspace_bytes, out, (hipStream_t)stream);
            if (rc >= 0) return rc;            // -1: graphs unavailable, fall through to direct launches
        }
    }
    Timer tm;      // everything else, and every invalid argument (reported by forward_impl), takes the direct path
    return forward_impl(h, img, ev, in_dtype, B, workspace, workspace_bytes, out, (hipStream_t)stream, tm);
}

extern "C" int iefvad_forward_scaled(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B, const float* img_row_scale,
                                     const float* ev_row_scale, void* workspace, size_t workspace_bytes, const iefvad_outputs* out, void* stream) {
    if (!h) return fail("iefvad_forward_scaled: null argument");
    if (!img_row_scale && !ev_row_scale) return iefvad_forward(h, img, ev, in_dtype, B, workspace, workspace_bytes, out, stream);
    Timer tm;      // direct launches: the scale vectors are per call, a cached graph would pin their addresses
    h->row_scale[0] = img_row_scale;
    h->row_scale[1] = ev_row_scale;
    const int rc = forward_impl(h, img, ev, in_dtype, B, workspace, workspace_bytes, out, (hipStream_t)stream, tm);
    h->row_scale[0] = h->row_scale[1] = nullptr;
    return rc;
}

extern "C" int iefvad_forward_timed(iefvad_handle* h, const void* img, const void* ev, int32_t in_dtype, int32_t B,
                                    void* workspace, size_t workspace_bytes, const iefvad_outputs* out, void* stream_,
                                    iefvad_stage_times* times) {
    if (!times) return fail("iefvad_forward_timed: null times");
    hipStream_t stream = (hipStream_t)stream_;
    if (!h) return fail("iefvad_forward_timed: null handle");
    if (!h->events) h->events = new (std::nothrow) EventPool();
    if (!h->events) return fail("iefvad_forward_timed: out of host memory");
    EventPool& pool = *h->events;
    pool.used = 0;
    pool.err = hipSuccess;
    Timer tm;
    tm.on = true;
    tm.stream = stream;
    tm.pool = &pool;
    hipEvent_t t0 = pool.take(), t1 = pool.take();
    if (!t0 || !t1) return fail("iefvad_forward_timed: hipEventCreate: %s", hipGetErrorString(pool.err));
    HIP_TRY(hipEventRecord(t0, stream));
    int rc = forward_impl(h, img, ev, in_dtype, B, workspace, workspace_bytes, out, stream, tm);
    (void)hipEventRecord(t1, stream);
    hipError_t se = hipStreamSynchronize(stream);
    if (rc) return rc;
    if (pool.err != hipSuccess) return fail("iefvad_forward_timed: hipEventCreate: %s", hipGetErrorString(pool.err));
    if (se != hipSuccess) return fail("iefvad_forward_timed: %s", hipGetErrorString(se));
    float acc[ST_COUNT] = {0};
    for (auto& s : tm.spans) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
        acc[s.stage] += ms;
    }
    float total = 0.f;
    HIP_TRY(hipEventElapsedTime(&total, t0, t1));
    memset(times, 0, sizeof(*times));
    times->total_ms = total;
    times->qkv_gemm_ms = acc[ST_QKV];
    times->attention_ms = acc[ST_ATT];
    times->out_gemm_ms = acc[ST_OUT];
    times->layernorm_ms = acc[ST_LN];
    times->head_gemm_ms = acc[ST_HEAD];
    times->fusion_ms = acc[ST_FUSION];
    times->refine_gemm_ms = acc[ST_REFINE];
    times->scorer_ms = acc[ST_SCORER];
    times->cast_ms = acc[ST_CAST];
    times->gemm_launches = tm.gemm_launches;
    return 0;
}


// ------------------------------------------------------------------------------------------------
// training: train-mode forward and the model's backward pass
// ------------------------------------------------------------------------------------------------
#include "train.h"

// ------------------------------------------------------------------------------------------------
// multi-GPU score gather (include/iefvad.h; csrc/gather.h binds librccl at run time)
// ------------------------------------------------------------------------------------------------
#define RCCL_TRY(api, expr)                                                                                   \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) return fail("%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

extern "C" int iefvad_comm_unique_id(void* id_bytes) {
    if (!id_bytes) return fail("iefvad_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == IEFVAD_COMM_ID_BYTES, "ncclUniqueId size");
    const char* why = "";
    const RcclApi* api = rccl_api(&why);
    if (!api) return fail("iefvad_comm_unique_id: %s", why);
    ncclUniqueId id;
    RCCL_TRY(api, api->GetUniqueId(&id));
    memcpy(id_bytes, &id, sizeof(id));
    return 0;
}

extern "C" int iefvad_comm_create(const void* id_bytes, int32_t nranks, int32_t rank, iefvad_comm** out) {
    if (!id_bytes || !out) return fail("iefvad_comm_create: null argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("iefvad_comm_create: rank %d of %d", rank, nranks);
    const char* why = "";
    const RcclApi* api = rccl_api(&why);
    if (!api) return fail("iefvad_comm_create: %s", why);
    iefvad_comm* c = new (std::nothrow) iefvad_comm();
    if (!c) return fail("iefvad_comm_create: out of host memory");
    memset(c, 0, sizeof(*c));
    c->rank = rank;
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) {
        delete c;
        return fail("iefvad_comm_create: %s", hipGetErrorString(e));
    }
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclResult_t r = api->CommInitRank(&c->comm, nranks, id, rank);
    if (r == ncclSuccess) r = api->CommCount(c->comm, &c->nranks);
    if (r != ncclSuccess) {
        if (c->comm) (void)api->CommDestroy(c->comm);
        delete c;
        return fail("iefvad_comm_create: %s", api->GetErrorString(r));
    }
    if (c->nranks != nranks) {
        const int got = c->nranks;
        (void)api->CommDestroy(c->comm);
        delete c;
        return fail("iefvad_comm_create: RCCL reports %d ranks, caller said %d", got, nranks);
    }
    *out = c;
    return 0;
}

extern "C" int32_t iefvad_comm_nranks(const iefvad_comm* c) { return c ? c->nranks : 0; }

extern "C" void iefvad_comm_destroy(iefvad_comm* c) {
    if (!c) return;
    const RcclApi* api = rccl_api(nullptr);
    if (api && c->comm) (void)api->CommDestroy(c->comm);
    delete c;
}

extern "C" int32_t iefvad_rccl_version(void) {
    const RcclApi* api = rccl_api(nullptr);
    int v = 0;
    if (!api || api->GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

extern "C" int iefvad_gather_plan(int32_t nranks, int32_t rank, const int64_t* counts, int64_t count, int64_t* summary,
                                  int64_t* steps) {
    if (!summary) return fail("iefvad_gather_plan: null summary");
    if (!counts && count < 0) return fail("iefvad_gather_plan: count = %lld", (long long)count);
    GatherPlan plan;
    if (const char* why = gather_plan(nranks, rank, counts, (size_t)count, &plan)) return fail("iefvad_gather_plan: %s", why);
    summary[0] = plan.equal ? 1 : 0;
    summary[1] = (int64_t)plan.my_offset;
    summary[2] = (int64_t)plan.my_count;
    summary[3] = (int64_t)plan.total;
    summary[4] = (int64_t)plan.steps.size();
    if (steps)
        for (size_t i = 0; i < plan.steps.size(); ++i) {
            steps[4 * i + 0] = plan.steps[i].peer;
            steps[4 * i + 1] = (int64_t)plan.steps[i].send_count;
            steps[4 * i + 2] = (int64_t)plan.steps[i].recv_offset;
            steps[4 * i + 3] = (int64_t)plan.steps[i].recv_count;
        }
    return 0;
}

extern "C" int iefvad_gather_scores(iefvad_comm* c, const float* local, size_t count, const int64_t* counts, float* gathered,
                                    size_t gathered_capacity, void* stream_) {
    if (!c || !gathered) return fail("iefvad_gather_scores: null argument");
    const RcclApi* api = rccl_api(nullptr);
    if (!api) return fail("iefvad_gather_scores: librccl not bound");
    hipStream_t stream = (hipStream_t)stream_;
    GatherPlan plan;
    if (const char* why = gather_plan(c->nranks, c->rank, counts, count, &plan)) return fail("iefvad_gather_scores: %s", why);
    if (plan.total > gathered_capacity)
        return fail("iefvad_gather_scores: the ranks contribute %zu elements, `gathered` holds %zu", plan.total, gathered_capacity);
    if (plan.my_count > 0 && !local) return fail("iefvad_gather_scores: null local buffer");
    // `local` may BE its own slot of `gathered` (in place, as ncclAllGather allows); any other overlap would be overwritten
    // by a peer's slice while it is still being sent
    if (plan.my_count > 0 && local != gathered + plan.my_offset && local < gathered + plan.total && gathered < local + plan.my_count)
        return fail("iefvad_gather_scores: `local` overlaps `gathered` outside its own slot");
    if (plan.total == 0) return 0;
    if (plan.equal) {   // the common case (bench, balanced shards): ONE collective
        RCCL_TRY(api, api->AllGather(local, gathered, plan.my_count, ncclFloat32, c->comm, stream));
        return 0;
    }
    // unequal shards: one grouped point-to-point exchange -- rank r's slice lands at its running offset on every peer
    RCCL_TRY(api, api->GroupStart());
    ncclResult_t first_bad = ncclSuccess;
    for (const GatherStep& s : plan.steps) {
        ncclResult_t a = s.send_count ? api->Send(local, s.send_count, ncclFloat32, s.peer, c->comm, stream) : ncclSuccess;
        ncclResult_t b = s.recv_count ? api->Recv(gathered + s.recv_offset, s.recv_count, ncclFloat32, s.peer, c->comm, stream) : ncclSuccess;
        if (first_bad == ncclSuccess) first_bad = (a != ncclSuccess) ? a : b;
    }
    ncclResult_t ge = api->GroupEnd();      // always close the group, even after a failed enqueue
    if (first_bad != ncclSuccess) return fail("iefvad_gather_scores: %s", api->GetErrorString(first_bad));
    if (ge != ncclSuccess) return fail("iefvad_gather_scores: ncclGroupEnd: %s", api->GetErrorString(ge));
    if (plan.my_count && gathered + plan.my_offset != local)
        HIP_TRY(hipMemcpyAsync(gathered + plan.my_offset, local, plan.my_count * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// training-side loss head: forward and the gradients with respect to the model's outputs (csrc/loss.h)
// ------------------------------------------------------------------------------------------------
extern "C" size_t iefvad_loss_workspace_bytes(int32_t B, int32_t T) {
    if (B <= 0 || T <= 0) return 0;
    return ((size_t)B * T * 4 + (size_t)B) * sizeof(float) + 256;
}

extern "C" int iefvad_loss_forward(const float* logits, const float* image_mu, const float* event_mu, const float* image_logvar,
                                   const float* event_logvar, const int32_t* lengths, const float* targets, int32_t B, int32_t T,
                                   int32_t noise_model, float nu, float lambda_reg, float lambda_kl, float* out, void* workspace,
                                   size_t workspace_bytes, void* stream_) {
    if (!logits || !lengths || !targets || !out || !workspace) return fail("iefvad_loss_forward: null argument");
    const bool heads = image_mu || event_mu || image_logvar || event_logvar;      // all four or none (none: CLAS2 alone)
    if (heads && !(image_mu && event_mu && image_logvar && event_logvar))
        return fail("iefvad_loss_forward: image_mu, event_mu, image_logvar, event_logvar must be given together");
    if (B <= 0) return fail("iefvad_loss_forward: B must be positive (got %d)", B);
    if (T != IEF_T) return fail("iefvad_loss_forward: kernels are built for T = %d (got %d)", IEF_T, T);
    if (noise_model != IEFVAD_NOISE_GAUSSIAN && noise_model != IEFVAD_NOISE_STUDENT_T)
        return fail("Unsupported noise_model. Choose 'Gaussian' or 'StudentT'.");
    if (noise_model == IEFVAD_NOISE_STUDENT_T && !(nu > 0.f)) return fail("iefvad_loss_forward: nu must be positive for StudentT");
    if (workspace_bytes < iefvad_loss_workspace_bytes(B, T)) return fail("iefvad_loss_forward: workspace too small");
    if ((uintptr_t)workspace & 15) return fail("iefvad_loss_forward: the workspace must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const int rows = heads ? B * T : 0;
    float* part = (float*)workspace;
    float* inst = part + (size_t)B * T * 4;
    hipLaunchKernelGGL(iefvad_mil_topk_kernel, dim3(B), dim3(256), 0, stream, logits, (const int*)lengths, inst, T);
    LossRowArgs ra;
    ra.mu_i = image_mu; ra.mu_e = event_mu; ra.lv_i = image_logvar; ra.lv_e = event_logvar; ra.part = part; ra.rows = rows;
    ra.lv_shift = noise_model == IEFVAD_NOISE_STUDENT_T ? logf(nu / (nu + 1.0f)) : 0.f;       // ucf_train.py:94-95
    if (heads) hipLaunchKernelGGL(iefvad_loss_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, ra);
    LossFinishArgs fa;
    fa.inst = inst; fa.targets = targets; fa.part = part; fa.out = out; fa.B = B; fa.rows = rows; fa.lambda_reg = lambda_reg;
    fa.lambda_kl = lambda_kl;
    hipLaunchKernelGGL(iefvad_loss_finish_kernel, dim3(1), dim3(256), 0, stream, fa);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int iefvad_loss_backward(const float* logits, const float* image_mu, const float* event_mu, const float* image_logvar,
                                    const float* event_logvar, const int32_t* lengths, const float* targets, int32_t B, int32_t T,
                                    int32_t noise_model, float nu, float lambda_reg, float lambda_kl, float grad_scale,
                                    float* d_logits, float* d_image_mu, float* d_event_mu, float* d_image_logvar,
                                    float* d_event_logvar, const float* grad_scale_dev, void* stream_) {
    if (!logits || !lengths || !targets) return fail("iefvad_loss_backward: null argument");
    const bool heads = d_image_mu || d_event_mu || d_image_logvar || d_event_logvar;
    if (heads && !(image_mu && event_mu && image_logvar && event_logvar))
        return fail("iefvad_loss_backward: image_mu, event_mu, image_logvar, event_logvar are needed for their gradients");
    if (B <= 0) return fail("iefvad_loss_backward: B must be positive (got %d)", B);
    if (T != IEF_T) return fail("iefvad_loss_backward: kernels are built for T = %d (got %d)", IEF_T, T);
    if (noise_model != IEFVAD_NOISE_GAUSSIAN && noise_model != IEFVAD_NOISE_STUDENT_T)
        return fail("Unsupported noise_model. Choose 'Gaussian' or 'StudentT'.");
    if (noise_model == IEFVAD_NOISE_STUDENT_T && !(nu > 0.f)) return fail("iefvad_loss_backward: nu must be positive for StudentT");
    hipStream_t stream = (hipStream_t)stream_;
    if (d_logits)
        hipLaunchKernelGGL(iefvad_mil_topk_grad_kernel, dim3(B), dim3(256), 0, stream, logits, (const int*)lengths, targets, d_logits, T,
                           grad_scale / (float)B, grad_scale_dev);
    if (heads) {
        LossRowGradArgs ga;
        ga.mu_i = image_mu; ga.mu_e = event_mu; ga.lv_i = image_logvar; ga.lv_e = event_logvar;
        ga.d_mu_i = d_image_mu; ga.d_mu_e = d_event_mu; ga.d_lv_i = d_image_logvar; ga.d_lv_e = d_event_logvar;
        ga.rows = B * T;
        ga.lv_shift = noise_model == IEFVAD_NOISE_STUDENT_T ? logf(nu / (nu + 1.0f)) : 0.f;
        ga.lambda_reg = lambda_reg; ga.lambda_kl = lambda_kl; ga.scale = grad_scale; ga.scale_dev = grad_scale_dev;
        hipLaunchKernelGGL(iefvad_loss_rows_grad_kernel, dim3((ga.rows + 3) / 4), dim3(256), 0, stream, ga);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int iefvad_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                                 double beta2, double eps, double weight_decay, int32_t step, void* stream_) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail("iefvad_adamw_step: null argument");
    if (step < 1) return fail("iefvad_adamw_step: step counts from 1 (got %d)", step);
    if (n == 0) return 0;
    AdamWArgs a;
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
    // torch forms these in Python floats (doubles) and rounds once when they meet the fp32 tensors
    a.decay = (float)(1.0 - lr * weight_decay);
    a.w1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    a.eps = (float)eps;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(iefvad_adamw_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int iefvad_adamw_step_multi(const iefvad_adamw_tensor* table_dev, int32_t count, uint64_t total_chunks, double lr, double beta1, double beta2,
                                       double eps, double weight_decay, int32_t step, void* stream_) {
    if (!table_dev) return fail("iefvad_adamw_step_multi: null argument");
    if (step < 1) return fail("iefvad_adamw_step_multi: step counts from 1 (got %d)", step);
    if (count <= 0 || total_chunks == 0) return 0;
    if (total_chunks > 0x7fffffffull) return fail("iefvad_adamw_step_multi: too many chunks");
    AdamWMultiArgs a;
    a.table = table_dev; a.count = count;
    a.decay = (float)(1.0 - lr * weight_decay);       // the scalars of iefvad_adamw_step, formed the same way
    a.w1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    a.eps = (float)eps;
    hipLaunchKernelGGL(iefvad_adamw_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream_, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// metric tail of the evaluation loop (metrics.h)
// ------------------------------------------------------------------------------------------------
static size_t mt_align(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t metric_layout(int64_t n, char* base, MetricWs* w) {
    const size_t tiles = (size_t)((n + MT_TILE - 1) / MT_TILE);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += mt_align(bytes); return p; };
    char* a = take((size_t)n * 8);
    char* b = take((size_t)n * 8);
    char* hist = take(tiles * MT_RADIX * 4);
    char* bsum = take(tiles * 4);
    char* bstart = take(tiles * 4);
    char* ap_part = take(tiles * 8);
    char* tail = take(16);
    char* dtotal = take(MT_RADIX * 4);
    if (w) {
        w->dtotal = (unsigned*)dtotal;
        w->a = (unsigned long long*)a; w->b = (unsigned long long*)b; w->hist = (unsigned*)hist; w->bsum = (unsigned*)bsum;
        w->bstart = (unsigned*)bstart; w->ap_part = (double*)ap_part; w->auc_num = (unsigned long long*)tail; w->flags = (unsigned*)(tail + 8);
    }
    return off;
}

extern "C" size_t iefvad_auc_ap_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return metric_layout(n, nullptr, nullptr);
}

extern "C" int iefvad_auc_ap(const float* scores, const uint8_t* gt_frames, int64_t n, int32_t repeat, double* auc, double* ap,
                             void* workspace, size_t workspace_bytes, void* stream_) {
    if (!scores || !gt_frames || !workspace || (!auc && !ap)) return fail("iefvad_auc_ap: null argument");
    if (n <= 0 || repeat <= 0) return fail("iefvad_auc_ap: n = %lld, repeat = %d", (long long)n, repeat);
    if ((unsigned long long)n * (unsigned long long)repeat >= (1ull << 32))
        return fail("iefvad_auc_ap: n * repeat = %llu frames do not fit the 32-bit frame counters", (unsigned long long)n * (unsigned long long)repeat);
    if (((uintptr_t)workspace & 255) != 0) return fail("iefvad_auc_ap: the workspace must be 256-byte aligned");
    MetricWs w;
    const size_t need = metric_layout(n, (char*)workspace, &w);
    if (workspace_bytes < need) return fail("iefvad_auc_ap: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    hipStream_t stream = (hipStream_t)stream_;
    const int tiles = (int)((n + MT_TILE - 1) / MT_TILE);
    HIP_TRY(hipMemsetAsync(w.auc_num, 0, 16, stream));
    hipLaunchKernelGGL(iefvad_metric_pairs_kernel, dim3((unsigned)((n + MT_THREADS - 1) / MT_THREADS)), dim3(MT_THREADS), 0, stream, scores,
                       (const unsigned char*)gt_frames, (long long)n, (int)repeat, w.a, w.flags);
    unsigned long long* src = w.a;
    unsigned long long* dst = w.b;
    for (int pass = 0; pass < 4; ++pass) {          // the key is the upper word of a pair
        const int shift = 32 + 8 * pass;
        hipLaunchKernelGGL(iefvad_metric_hist_kernel, dim3(tiles), dim3(MT_THREADS), 0, stream, src, (long long)n, shift, w.hist, tiles);
        hipLaunchKernelGGL(iefvad_metric_digit_scan_kernel, dim3(MT_RADIX), dim3(256), 0, stream, w.hist, tiles, w.dtotal);
        hipLaunchKernelGGL(iefvad_metric_scatter_kernel, dim3(tiles), dim3(MT_THREADS), 0, stream, src, dst, (long long)n, shift, w.hist, tiles, w.dtotal);
        unsigned long long* t = src; src = dst; dst = t;
    }
    // four passes: the sorted pairs are back in w.a, w.b is free for the two scanned columns
    unsigned* tp_incl = (unsigned*)w.b;
    unsigned* gstart = tp_incl + n;
    hipLaunchKernelGGL(iefvad_metric_tile_sums_kernel, dim3(tiles), dim3(MT_THREADS), 0, stream, src, (long long)n, w.bsum, w.bstart);
    hipLaunchKernelGGL(iefvad_metric_tile_scan_kernel, dim3(1), dim3(1024), 0, stream, w.bsum, w.bstart, tiles);
    hipLaunchKernelGGL(iefvad_metric_tile_apply_kernel, dim3(tiles), dim3(MT_THREADS), 0, stream, src, (long long)n, w.bsum, w.bstart, tp_incl, gstart);
    hipLaunchKernelGGL(iefvad_metric_groups_kernel, dim3(tiles), dim3(MT_THREADS), 0, stream, src, (long long)n, (int)repeat, tp_incl, gstart, w.auc_num,
                       w.ap_part);
    hipLaunchKernelGGL(iefvad_metric_finish_kernel, dim3(1), dim3(256), 0, stream, tp_incl, (long long)n, (int)repeat, w.auc_num, w.ap_part, tiles, w.flags,
                       auc, ap);
    HIP_TRY(hipGetLastError());
    return 0;
}

#include "vadclip.h"
#include "hostgather.h"
#include "hostpipe.h"

// One production row-block kernel of the bf16 mode on caller-supplied rows (include/iefvad.h): the launch helpers forward_pass uses.
extern "C" int iefvad_rowblock_unit(iefvad_handle* h, int32_t stage, int32_t layer, int32_t rows, const iefvad_unit_io* io, void* stream_) {
    if (!h || !io) return fail("iefvad_rowblock_unit: null argument");
    if (!h->weights_set) return fail("iefvad_rowblock_unit: weights not set");
    if (h->cfg.compute != IEFVAD_COMPUTE_BF16) return fail("iefvad_rowblock_unit: the row-block kernels belong to compute = BF16 (got %d)", h->cfg.compute);
    if (rows <= 0 || rows % 64) return fail("iefvad_rowblock_unit: rows = %d must be a positive multiple of 64", rows);
    hipStream_t stream = (hipStream_t)stream_;
    const int L = h->cfg.num_layers, K = h->cfg.num_steps;
    Timer tm;
    auto aligned = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    switch (stage) {
    case IEFVAD_UNIT_INPROJ: {
        if (layer < 0 || layer >= L) return fail("iefvad_rowblock_unit: layer %d of %d", layer, L);
        if (!h->iproj_stream[0][layer]) return fail("iefvad_rowblock_unit: in_proj streams are not packed on this handle");
        const void* A[2] = {io->x[0], io->x[1]};
        bf16_t* C[2] = {(bf16_t*)io->y[0], (bf16_t*)io->y[1]};
        for (int m = 0; m < 2; ++m)
            if (!A[m] || !C[m] || !aligned(A[m]) || !aligned(C[m])) return fail("iefvad_rowblock_unit: INPROJ needs x[m] and y[m], 16-byte aligned");
        return launch_inproj_chain(h, layer, A, layer == 0, C, rows, stream, tm);
    }
    case IEFVAD_UNIT_OUTPROJ_LN: {
        if (layer < 0 || layer >= L) return fail("iefvad_rowblock_unit: layer %d of %d", layer, L);
        if (!h->oproj_stream[0][layer]) return fail("iefvad_rowblock_unit: out_proj streams are not packed on this handle");
        const bf16_t* A[2] = {(const bf16_t*)io->x[0], (const bf16_t*)io->x[1]};
        const float* R[2] = {io->resid[0], io->resid[1]};
        float* y[2] = {(float*)io->y[0], (float*)io->y[1]};
        bf16_t* yb[2] = {(bf16_t*)io->yb[0], (bf16_t*)io->yb[1]};
        for (int m = 0; m < 2; ++m)
            if (!A[m] || !R[m] || (!y[m] && !yb[m]) || !aligned(A[m]) || !aligned(R[m]) || !aligned(y[m]) || !aligned(yb[m]))
                return fail("iefvad_rowblock_unit: OUTPROJ_LN needs x[m], resid[m] and y[m] or yb[m], 16-byte aligned");
        return launch_outproj_ln_chain(h, layer, layer == L - 1, A, R, y, yb, rows, stream, tm);
    }
    case IEFVAD_UNIT_HEADS: {
        if (!h->heads_stream) return fail("iefvad_rowblock_unit: the heads stream is not packed on this handle");
        if (!io->x[0] || !io->x[1] || !io->z || !aligned(io->x[0]) || !aligned(io->x[1]) || !aligned(io->z))
            return fail("iefvad_rowblock_unit: HEADS needs x[0], x[1] and z, 16-byte aligned");
        HeadsChainArgs ha;
        memset(&ha, 0, sizeof(ha));
        for (int m = 0; m < 2; ++m) {
            ha.A[m] = (const bf16_t*)io->x[m];
            ha.mu[m] = io->mu[m]; ha.lv[m] = io->logvar[m]; ha.n[m] = io->w[m];
        }
        ha.z = io->z;
        const float factor = (h->cfg.noise_model == IEFVAD_NOISE_STUDENT_T) ? (h->cfg.nu + 1.0f) / h->cfg.nu : 1.0f;
        return launch_heads_chain(h, ha, rows, factor, stream, tm);
    }
    case IEFVAD_UNIT_REFINE: {
        if (K < 1 || !h->chain_stream) return fail("iefvad_rowblock_unit: the refinement chain needs K >= 1 (K = %d)", K);
        if (!io->x[0] || !io->logits || !aligned(io->x[0]) || !aligned(io->z)) return fail("iefvad_rowblock_unit: REFINE needs x[0] (z_0) and logits");
        return launch_refine_chain(h, (const float*)io->x[0], io->z, io->logits, rows, stream, tm);
    }
    default:
        return fail("iefvad_rowblock_unit: unknown stage %d", stage);
    }
}

extern "C" int iefvad_gemm_bias(const void* A, const void* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                                int32_t compute, void* stream) {
    if (!A || !W || !bias || !C) return fail("iefvad_gemm_bias: null argument");
    Timer tm;
    if (compute == IEFVAD_COMPUTE_F32) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.epi = EPI_BIAS;
        g.p[0].A = (const float*)A; g.p[0].W = (const float*)W; g.p[0].bias = bias; g.p[0].C = C;
        return launch_gemm(g, 1, (hipStream_t)stream, tm, ST_QKV);
    }
    if (compute == IEFVAD_COMPUTE_BF16) {
        GemmBArgs g;
        memset(&g, 0, sizeof(g));
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.epi = EPI_BIAS;
        g.p[0].A = (const bf16_t*)A; g.p[0].W = (const bf16_t*)W; g.p[0].bias = bias; g.p[0].C = C;
        return launch_gemm_b(g, 1, (hipStream_t)stream, tm, ST_QKV);
    }
    if (compute == IEFVAD_COMPUTE_BF16X6) {
        GemmBArgs g;
        memset(&g, 0, sizeof(g));
        g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.epi = EPI_BIAS; g.wplane = N * K * 2;
        g.p[0].A = (const bf16_t*)A; g.p[0].W = (const bf16_t*)W; g.p[0].bias = bias; g.p[0].C = C;
        hipError_t e = hipFuncSetAttribute((const void*)iefvad_gemm_split_n128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           GS_LDS_BYTES_OF(2));
        if (e != hipSuccess) return fail("iefvad_gemm_bias: %s", hipGetErrorString(e));
        return launch_gemm_split(g, 1, (hipStream_t)stream, tm, ST_QKV);
    }
    return fail("iefvad_gemm_bias: unknown compute mode %d", compute);
}
