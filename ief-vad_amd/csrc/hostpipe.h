// iefvad_forward_videos_host: a whole evaluation list in ONE library call (include/iefvad.h).  Included by iefvad.hip behind
// forward_videos_impl and stream_copy.
//
// The reference's evaluation loop (/root/reference/test.py:76-121; data/dataset.py:34-52) loads, pads, uploads and scores one video
// per Python iteration.  iefvad_forward_videos already takes only the valid rows of a packed batch of videos, but the loop around it
// was Python: per batch it built pointer tables, staged the rows into pinned memory, issued the copies and the forward -- on short
// videos (ShanghaiTech / MSAD: 40 snippets on average) three times the device time.  Here the caller hands over the HOST row
// pointers and lengths of every video and the library walks the list:
//   * a worker thread gathers the rows of the next passes into a ring of four pinned staging slots (non-temporal stores, a
//     persistent pool of copy threads, byte ranges balanced across videos);
//   * the calling thread sends pass k on an internal copy stream (one asynchronous copy per modality) and enqueues its forward
//     (forward_videos_impl) behind an event on one of two internal compute streams, alternating: copies run under forwards, and two
//     partly filled passes run side by side;
//   * per-snippet results land in the caller's DEVICE vectors in list order; the caller's stream waits for both compute streams at
//     the end of the call (nothing is read back here).
// Staging slots, device input slots and the forward's workspace belong to the handle and are reused across calls (pinning memory
// costs milliseconds per 100 MB).  A slot is rewritten only after the event behind its last use has completed.
#pragma once
#include <chrono>

struct HostPipe {
    GatherPool* pool = nullptr;
    hipStream_t copy_stream = nullptr;
    static const int kSlots = 4;                    // passes in flight: one being gathered, one or two being sent, one being computed
    void* pinned[kSlots][2] = {};                   // [slot][modality]
    void* dev_in[kSlots][2] = {};
    size_t slot_bytes = 0;
    hipEvent_t sent[kSlots] = {};                   // the copies out of pinned[slot] have completed
    hipEvent_t used[kSlots] = {};                   // the forward that read dev_in[slot] has completed
    hipEvent_t ready = nullptr;                     // caller's stream -> internal streams ordering at entry
    // passes alternate between two compute streams (each with a workspace of its own): the kernels of a 60 - 130 chunk pass fill a
    // fraction of the chip and are latency-bound one behind the other; two passes side by side overlap (as harness lanes = 2 did)
    static const int kLanes = 2;
    hipStream_t lane[kLanes] = {};
    hipEvent_t lane_done[kLanes] = {};
    void* workspace[kLanes] = {};
    size_t workspace_bytes = 0;
};

static void release_hostpipe(iefvad_handle* h) {
    HostPipe* p = h->hostpipe;
    if (!p) return;
    // nothing of an earlier call may still be copying out of the pinned slots or computing in the workspaces
    if (p->copy_stream) (void)hipStreamSynchronize(p->copy_stream);
    for (int l = 0; l < HostPipe::kLanes; ++l)
        if (p->lane[l]) (void)hipStreamSynchronize(p->lane[l]);
    for (int s = 0; s < HostPipe::kSlots; ++s) {
        for (int m = 0; m < 2; ++m) {
            if (p->pinned[s][m]) (void)hipHostFree(p->pinned[s][m]);
            if (p->dev_in[s][m]) (void)hipFree(p->dev_in[s][m]);
        }
        if (p->sent[s]) (void)hipEventDestroy(p->sent[s]);
        if (p->used[s]) (void)hipEventDestroy(p->used[s]);
    }
    if (p->ready) (void)hipEventDestroy(p->ready);
    for (int l = 0; l < HostPipe::kLanes; ++l) {
        if (p->workspace[l]) (void)hipFree(p->workspace[l]);
        if (p->lane[l]) (void)hipStreamDestroy(p->lane[l]);
        if (p->lane_done[l]) (void)hipEventDestroy(p->lane_done[l]);
    }
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    delete p->pool;
    delete p;
    h->hostpipe = nullptr;
}

extern "C" int iefvad_forward_videos_host(iefvad_handle* h, const void* const* img_rows, const void* const* ev_rows, int32_t in_dtype,
                                          int32_t wire_dtype, const int32_t* lengths, int32_t nvideos, int32_t nan_to_num, int32_t batch_chunks,
                                          int32_t host_threads, float* logits, float* w_i_mean, float* w_e_mean, void* stream_) {
    if (!h || !img_rows || !ev_rows || !lengths || !logits) return fail("iefvad_forward_videos_host: null argument");
    if (!h->weights_set) return fail("iefvad_forward_videos_host: weights not set");
    if (nvideos <= 0) return fail("iefvad_forward_videos_host: nvideos must be positive (got %d)", nvideos);
    if (in_dtype != IEFVAD_IN_F32 && in_dtype != IEFVAD_IN_F16 && in_dtype != IEFVAD_IN_BF16)
        return fail("iefvad_forward_videos_host: unknown in_dtype %d", in_dtype);
    if (wire_dtype != in_dtype && !(in_dtype == IEFVAD_IN_F32 && wire_dtype == IEFVAD_IN_BF16))
        return fail("iefvad_forward_videos_host: wire_dtype %d with in_dtype %d (the wire type is in_dtype, or BF16 for F32 rows)", wire_dtype, in_dtype);
    if (wire_dtype != in_dtype && h->cfg.compute != IEFVAD_COMPUTE_BF16)
        return fail("iefvad_forward_videos_host: a narrowed wire type belongs to the bf16 mode (compute = %d)", h->cfg.compute);
    for (int v = 0; v < nvideos; ++v) {
        if (lengths[v] <= 0) return fail("iefvad_forward_videos_host: lengths[%d] = %d", v, lengths[v]);
        if (!img_rows[v] || !ev_rows[v]) return fail("iefvad_forward_videos_host: null row pointer (video %d)", v);
    }
    hipStream_t stream = (hipStream_t)stream_;
    const bool narrow = wire_dtype != in_dtype;
    const size_t row_bytes = (size_t)IEF_D * in_elem_bytes(in_dtype);            // host rows
    const size_t wire_row_bytes = (size_t)IEF_D * in_elem_bytes(wire_dtype);     // staging slots, copies, device input slots
    const int want = batch_chunks > 0 ? batch_chunks : 128;
    try {

    // passes: runs of whole videos with >= `want` chunks; the FIRST pass is half that (the first forward starts after a short
    // gather + copy) and the last chunks of the list are cut into two halves (what is left when the call returns is the copy
    // and the forward of a half pass).  Finer tapering loses: every pass costs ~0.3 ms of copy / event latency.
    // Row-compressed passes (valid rows + one pad row per chunk) are also closed BEFORE their row set outgrows one round of the
    // row-block kernels -- one 64-row block per CU (num_cus x 64 rows per 128 chunks asked for): a 129th .. 257th block costs a
    // second round on 256 CUs, i.e. as much as the first 256.
    struct Batch { int v0, v1; long long rows, chunks, enc; };
    std::vector<Batch> batches;
    long long max_rows = 0, max_chunks = 0, total_chunks = 0;
    for (int v = 0; v < nvideos; ++v) total_chunks += video_chunks(lengths[v]);
    const bool compressed = !h->dense_encoder && h->cfg.compute != IEFVAD_COMPUTE_FP16X3;
    static const bool row_cap_off = [] { const char* v = getenv("IEFVAD_HOSTPIPE_ROWCAP"); return v && v[0] == '0'; }();
    const long long round_rows = (long long)(h->num_cus > 0 ? h->num_cus : 256) * 64;
    const long long row_cap = (compressed && !row_cap_off) ? round_rows * ((want + 64) / 128 > 1 ? (want + 64) / 128 : 1) : (1LL << 60);
    auto enc_rows_of = [](int n) -> long long { return (long long)(n / IEF_T) * IEF_T + (n % IEF_T ? n % IEF_T + 1 : 0); };
    {
        Batch b = {0, 0, 0, 0, 0};
        long long done_chunks = 0;
        long long target = want / 2 > 16 ? want / 2 : want;
        auto close = [&](int next_v) {
            batches.push_back(b);
            if (b.rows > max_rows) max_rows = b.rows;
            if (b.chunks > max_chunks) max_chunks = b.chunks;
            done_chunks += b.chunks;
            b = Batch{next_v, next_v, 0, 0, 0};
            const long long left = total_chunks - done_chunks;
            target = want;
            if (left < want + want / 2 && left > want / 2) target = (left + 1) / 2;
        };
        for (int v = 0; v < nvideos; ++v) {
            const long long e = enc_rows_of(lengths[v]);
            if (b.chunks > 0 && b.enc + e > row_cap && b.enc >= row_cap / 2) close(v);      // the video would start a second round
            b.rows += lengths[v];
            b.chunks += video_chunks(lengths[v]);
            b.enc += e;
            b.v1 = v + 1;
            if (b.chunks >= target || v == nvideos - 1) close(v + 1);
        }
    }
    if (max_chunks > 0x7fffffffLL / IEF_T) return fail("iefvad_forward_videos_host: batch too large");

    HIP_TRY(hipSetDevice(h->device));
    if (!h->hostpipe) {
        // built completely or not at all: a handle never keeps a HostPipe with missing streams / events
        h->hostpipe = new (std::nothrow) HostPipe();
        if (!h->hostpipe) return fail("iefvad_forward_videos_host: out of host memory");
        HostPipe& np = *h->hostpipe;
        hipError_t ce = hipStreamCreateWithFlags(&np.copy_stream, hipStreamNonBlocking);
        for (int s = 0; s < HostPipe::kSlots && ce == hipSuccess; ++s) {
            ce = hipEventCreateWithFlags(&np.sent[s], hipEventDisableTiming);
            if (ce == hipSuccess) ce = hipEventCreateWithFlags(&np.used[s], hipEventDisableTiming);
        }
        if (ce == hipSuccess) ce = hipEventCreateWithFlags(&np.ready, hipEventDisableTiming);
        for (int l = 0; l < HostPipe::kLanes && ce == hipSuccess; ++l) {
            ce = hipStreamCreateWithFlags(&np.lane[l], hipStreamNonBlocking);
            if (ce == hipSuccess) ce = hipEventCreateWithFlags(&np.lane_done[l], hipEventDisableTiming);
        }
        if (ce != hipSuccess) {
            release_hostpipe(h);
            return fail("iefvad_forward_videos_host: creating the internal streams / events failed: %s", hipGetErrorString(ce));
        }
    }
    HostPipe& p = *h->hostpipe;
    const size_t need_slot = (size_t)max_rows * wire_row_bytes;
    if (p.slot_bytes < need_slot) {
        for (int s = 0; s < HostPipe::kSlots; ++s) {       // nothing may still be reading the old slots
            HIP_TRY(hipEventSynchronize(p.sent[s]));
            HIP_TRY(hipEventSynchronize(p.used[s]));
            for (int m = 0; m < 2; ++m) {
                if (p.pinned[s][m]) (void)hipHostFree(p.pinned[s][m]);
                if (p.dev_in[s][m]) (void)hipFree(p.dev_in[s][m]);
                p.pinned[s][m] = p.dev_in[s][m] = nullptr;
            }
        }
        p.slot_bytes = 0;
        const size_t cap = need_slot + need_slot / 4 + 4096;
        for (int s = 0; s < HostPipe::kSlots; ++s)
            for (int m = 0; m < 2; ++m) {
                HIP_TRY(hipHostMalloc(&p.pinned[s][m], cap, hipHostMallocDefault));
                HIP_TRY(hipMalloc(&p.dev_in[s][m], cap));
            }
        p.slot_bytes = cap;
    }
    const size_t need_ws = iefvad_workspace_bytes(h, (int32_t)max_chunks);
    if (p.workspace_bytes < need_ws) {
        HIP_TRY(hipDeviceSynchronize());                   // a forward of an earlier call may still use the old workspaces
        p.workspace_bytes = 0;
        for (int l = 0; l < HostPipe::kLanes; ++l) {
            if (p.workspace[l]) (void)hipFree(p.workspace[l]);
            p.workspace[l] = nullptr;
            HIP_TRY(hipMalloc(&p.workspace[l], need_ws + need_ws / 8));
        }
        p.workspace_bytes = need_ws + need_ws / 8;
    }

    // ---- worker: stage batch k into pinned slot k % kSlots as soon as the copies of batch k - kSlots have left it
    const int nb = (int)batches.size();
    std::mutex mu;
    std::condition_variable cv;
    int staged = 0;                 // batches whose rows are in pinned memory
    int issued = 0;                 // batches whose copies have been enqueued (their `sent` event recorded)
    bool abort_all = false;
    const int nthreads = host_threads > 0 ? (host_threads > 16 ? 16 : host_threads) : 8;
    if (!p.pool) p.pool = new (std::nothrow) GatherPool();
    if (!p.pool) return fail("iefvad_forward_videos_host: out of host memory");
    p.pool->start(nthreads);
    std::vector<const void*> srcs;
    std::vector<size_t> offs;
    srcs.reserve((size_t)nvideos);
    offs.reserve((size_t)nvideos + 1);
    auto stage = [&](int k) {       // the slot is free: gather the rows of batch k into it
        const int s = k % HostPipe::kSlots;
        (void)hipEventSynchronize(p.sent[s]);
        const Batch& b = batches[(size_t)k];
        const int n = b.v1 - b.v0;
        srcs.resize((size_t)n);
        offs.resize((size_t)n + 1);
        offs[0] = 0;
        for (int i = 0; i < n; ++i) offs[(size_t)i + 1] = offs[(size_t)i] + (size_t)lengths[b.v0 + i] * row_bytes;
        for (int m = 0; m < 2; ++m) {
            for (int i = 0; i < n; ++i) srcs[(size_t)i] = (m ? ev_rows : img_rows)[b.v0 + i];
            p.pool->run((char*)p.pinned[s][m], srcs.data(), offs.data(), n, narrow);
        }
    };
    std::thread worker;
    // whatever leaves this scope -- a return, an exception of the host-side tables -- releases and joins the worker first
    // (a joinable std::thread that is destroyed calls std::terminate: an abort across the ABI)
    struct WorkerGuard {
        std::thread& t; std::mutex& mu; std::condition_variable& cv; bool& abort_all; int& issued; int release;
        ~WorkerGuard() {
            if (!t.joinable()) return;
            {
                std::lock_guard<std::mutex> lk(mu);
                abort_all = true;
                issued = release;
            }
            cv.notify_all();
            t.join();
        }
    } guard{worker, mu, cv, abort_all, issued, nb + HostPipe::kSlots};
    bool threaded = nb > 1;
    if (threaded) {
        try {
            worker = std::thread([&] {
                for (int k = 0; k < nb; ++k) {
                    {   // the slot was last used by pass k - kSlots: its `sent` event must have been RECORDED (issued > k - kSlots)
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return abort_all || issued > k - HostPipe::kSlots; });
                        if (abort_all) return;
                    }
                    stage(k);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        staged = k + 1;
                    }
                    cv.notify_all();
                }
            });
        } catch (...) {
            threaded = false;       // thread creation refused: stage inline, batch by batch
        }
    }

    // ---- this thread: copies on the copy stream, forwards on the two internal lane streams
    int rc = 0;
    hipError_t he = hipEventRecord(p.ready, stream);             // inputs of this call must not overtake what the caller enqueued before
    if (he == hipSuccess) he = hipStreamWaitEvent(p.copy_stream, p.ready, 0);
    for (int l = 0; l < HostPipe::kLanes && he == hipSuccess; ++l) he = hipStreamWaitEvent(p.lane[l], p.ready, 0);
    long long row0 = 0;
    Timer tm;
    static const bool trace = [] { const char* v = getenv("IEFVAD_HOSTPIPE_TRACE"); return v && v[0] == '1'; }();
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = trace ? now_us() : 0.0;
    for (int k = 0; k < nb && rc == 0 && he == hipSuccess; ++k) {
        const int s = k % HostPipe::kSlots;
        const Batch& b = batches[(size_t)k];
        const double t0 = trace ? now_us() : 0.0;
        if (threaded) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return staged > k; });
        } else {
            stage(k);
        }
        const double t1 = trace ? now_us() : 0.0;
        const size_t bytes = (size_t)b.rows * wire_row_bytes;
        he = hipStreamWaitEvent(p.copy_stream, p.used[s], 0);    // the forward of batch k - kSlots has read dev_in[s]
        for (int m = 0; m < 2 && he == hipSuccess; ++m)
            he = hipMemcpyAsync(p.dev_in[s][m], p.pinned[s][m], bytes, hipMemcpyHostToDevice, p.copy_stream);
        if (he == hipSuccess) he = hipEventRecord(p.sent[s], p.copy_stream);
        {
            std::lock_guard<std::mutex> lk(mu);
            issued = k + 1;
        }
        cv.notify_all();
        if (he != hipSuccess) break;
        // fp16x3 keeps its running-max words on the handle: one pass at a time there
        const int ln = h->cfg.compute == IEFVAD_COMPUTE_FP16X3 ? 0 : k % HostPipe::kLanes;
        hipStream_t cs = p.lane[ln];
        he = hipStreamWaitEvent(cs, p.sent[s], 0);
        if (he != hipSuccess) break;
        rc = forward_videos_impl(h, p.dev_in[s][0], p.dev_in[s][1], wire_dtype, lengths + b.v0, b.v1 - b.v0, nan_to_num,
                                 p.workspace[ln], p.workspace_bytes, logits + row0, w_i_mean ? w_i_mean + row0 : nullptr,
                                 w_e_mean ? w_e_mean + row0 : nullptr, cs, tm);
        if (rc == 0) he = hipEventRecord(p.used[s], cs);
        if (trace)
            fprintf(stderr, "[hostpipe] pass %d: %lld chunks %lld rows | waited for staging %.0f us | enqueue copies + forward %.0f us | t = %.0f us\n", k,
                    b.chunks, b.rows, t1 - t0, now_us() - t1, now_us() - t_begin);
        row0 += b.rows;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc != 0 || he != hipSuccess) abort_all = true;
        issued = nb + HostPipe::kSlots;
    }
    cv.notify_all();
    if (worker.joinable()) worker.join();
    // the caller's stream continues behind both lanes
    for (int l = 0; l < HostPipe::kLanes; ++l) {
        hipError_t e2 = hipEventRecord(p.lane_done[l], p.lane[l]);
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(stream, p.lane_done[l], 0);
        if (he == hipSuccess) he = e2;
    }
    if (rc) return rc;
    if (he != hipSuccess) return fail("iefvad_forward_videos_host: %s", hipGetErrorString(he));
    return 0;
    } catch (const std::exception& e) {      // nothing throws across the ABI (allocation failures of the host-side tables)
        return fail("iefvad_forward_videos_host: %s", e.what());
    }
}
