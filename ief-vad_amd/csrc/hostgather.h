// Host side of the whole-video path: gather the videos' rows into one (pinned) staging buffer -- iefvad_host_gather,
// iefvad_host_gather_bf16 (include/iefvad.h) and the persistent copy-thread pool the list walk (hostpipe.h) stages with.
// Pure host C++: no HIP type or call, so that tests/cabi/hostgather_san.cpp can build exactly this code with gcc under
// -fsanitize=thread and -fsanitize=address,undefined (the GPU box has no sanitizer runs; the CPU build does).
// The includer provides `static int fail(const char* fmt, ...)` (thread-local message, returns non-zero).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

// hipcc (clang) has the non-temporal store builtin; gcc (the sanitizer build of tests/cabi) takes a plain store: same bytes
#if defined(__clang__)
#define IEF_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define IEF_NT_STORE(v, p) (*(p) = (v))
#endif

// Copy with non-temporal stores: the destination is a pinned staging buffer that only the DMA engine reads next, so its lines need
// neither be fetched for ownership nor stay in the CPU caches (a plain memcpy of the 890 MB of an XD-sized list moves 2.7 GB
// through the memory controllers, this 1.8 GB -- and the H2D copy of the previous batch is reading the same DRAM meanwhile).
static void stream_copy(char* d, const char* s, size_t n) {
    typedef long long v4 __attribute__((vector_size(32)));
    typedef long long v4u __attribute__((vector_size(32), aligned(1)));
    if (n < 4096) { memcpy(d, s, n); return; }
    size_t head = (32 - ((uintptr_t)d & 31)) & 31;
    memcpy(d, s, head);
    d += head; s += head; n -= head;
    const size_t body = n & ~(size_t)127;
    for (size_t i = 0; i < body; i += 128) {
        const v4 a = *(const v4u*)(s + i), b = *(const v4u*)(s + i + 32), c = *(const v4u*)(s + i + 64), e = *(const v4u*)(s + i + 96);
        IEF_NT_STORE(a, (v4*)(d + i));
        IEF_NT_STORE(b, (v4*)(d + i + 32));
        IEF_NT_STORE(c, (v4*)(d + i + 64));
        IEF_NT_STORE(e, (v4*)(d + i + 96));
    }
    memcpy(d + body, s + body, n - body);
}

// fp32 -> bf16 (round to nearest even, NaN -> quiet NaN with its sign: what the device's v_cvt_pk_bf16_f32 gives) with non-temporal
// stores: the staging form of wire_dtype = BF16 (include/iefvad.h).  n_src bytes of fp32 in, n_src / 2 bytes out; the caller keeps
// ranges at multiples of 64 source bytes.
#define IEF_CONVERT_BODY                                                                                                     \
    typedef unsigned u8v __attribute__((vector_size(32)));                                                                   \
    typedef unsigned u8vu __attribute__((vector_size(32), aligned(1)));                                                      \
    typedef int i8v __attribute__((vector_size(32)));                                                                        \
    typedef unsigned short h8v __attribute__((vector_size(16)));                                                             \
    typedef unsigned short h16v __attribute__((vector_size(32)));                                                            \
    size_t i = 0;                                                                                                            \
    if (((uintptr_t)d & 31) == 0) {                                                                                          \
        for (; i + 64 <= n_src; i += 64) {                                                                                   \
            h8v half[2];                                                                                                     \
            for (int q = 0; q < 2; ++q) {                                                                                    \
                const u8v u = *(const u8vu*)(s + i + 32 * q);                                                                \
                const u8v r = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;                                                        \
                const i8v isnan = (i8v)((u & 0x7FFFFFFFu) > 0x7F800000u);                                                    \
                const u8v o = ((u8v)isnan & ((u >> 16) | 0x40u)) | (~(u8v)isnan & r);                                        \
                half[q] = __builtin_convertvector(o, h8v);                                                                   \
            }                                                                                                                \
            h16v both;                                                                                                       \
            memcpy(&both, half, 32);                                                                                         \
            IEF_NT_STORE(both, (h16v*)(d + i / 2));                                                                          \
        }                                                                                                                    \
    }                                                                                                                        \
    for (; i + 4 <= n_src; i += 4) {                                                                                         \
        unsigned u;                                                                                                          \
        memcpy(&u, s + i, 4);                                                                                                \
        const unsigned short o = ((u & 0x7FFFFFFFu) > 0x7F800000u) ? (unsigned short)((u >> 16) | 0x40u)                      \
                                                                   : (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); \
        memcpy(d + i / 2, &o, 2);                                                                                            \
    }
__attribute__((target("avx2"))) static void stream_convert_bf16_avx2(char* d, const char* s, size_t n_src) { IEF_CONVERT_BODY }
static void stream_convert_bf16_base(char* d, const char* s, size_t n_src) { IEF_CONVERT_BODY }
#undef IEF_CONVERT_BODY
static void stream_convert_bf16(char* d, const char* s, size_t n_src) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) stream_convert_bf16_avx2(d, s, n_src);
    else stream_convert_bf16_base(d, s, n_src);
}

// below this many bytes a job is not worth waking the pool for (the sanitizer build lowers it so that small jobs exercise the pool)
#ifndef IEF_POOL_MIN_BYTES
#define IEF_POOL_MIN_BYTES ((size_t)1 << 20)
#endif

// A few persistent copy threads: a job is one byte stream (the concatenation of `count` pieces) cut into equal byte ranges, one per
// thread -- a range may start and end inside a piece, so short and long videos balance.  Threads sleep between jobs.
// Everything a worker reads about a job -- including HOW MANY ranges the job has -- is a job field published under `mu`;
// a thread added by a later start() begins at the generation current at its creation, so it never runs a job that was
// published (and whose tables may be gone) before it existed.
struct GatherPool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    unsigned long long generation = 0;
    int pending = 0;
    bool stop = false;
    // the job
    char* dst = nullptr;
    const void* const* srcs = nullptr;
    const size_t* offs = nullptr;        // count + 1 prefix sums of the piece sizes (SOURCE bytes)
    int64_t count = 0;
    bool to_bf16 = false;                // the pieces are fp32, the destination takes them as bf16 (half the bytes)
    int job_ranges = 1;                  // ranges of the current job: the caller's + one per thread that existed at run()

    struct Job { char* dst; const void* const* srcs; const size_t* offs; int64_t count; bool to_bf16; };
    static void work(const Job& j, int t, int nt) {
        const size_t total = j.offs[j.count];
        const size_t step = (total / nt) & ~(size_t)63;      // ranges start at multiples of 64 source bytes (16 fp32 -> one 32-byte store)
        const size_t lo = step * t, hi = (t == nt - 1) ? total : step * (t + 1);
        if (hi <= lo) return;
        int64_t i = (int64_t)(std::upper_bound(j.offs, j.offs + j.count + 1, lo) - j.offs) - 1;      // piece that holds byte lo
        size_t pos = lo;
        while (pos < hi) {
            const size_t end = j.offs[i + 1] < hi ? j.offs[i + 1] : hi;
            if (end > pos) {
                if (j.to_bf16) stream_convert_bf16(j.dst + pos / 2, (const char*)j.srcs[i] + (pos - j.offs[i]), end - pos);
                else stream_copy(j.dst + pos, (const char*)j.srcs[i] + (pos - j.offs[i]), end - pos);
            }
            pos = end;
            ++i;
        }
        __builtin_ia32_sfence();
    }
    void loop(int t, unsigned long long seen) {
        for (;;) {
            Job j;
            int nt;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                j = Job{dst, srcs, offs, count, to_bf16};
                nt = job_ranges;
            }
            if (t + 1 < nt) work(j, t + 1, nt);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    // called by ONE thread at a time; the caller copies range 0 itself
    void run(char* dst_, const void* const* srcs_, const size_t* offs_, int64_t count_, bool to_bf16_ = false) {
        const Job j{dst_, srcs_, offs_, count_, to_bf16_};
        const size_t total = offs_[count_];
        if (threads.empty() || total < IEF_POOL_MIN_BYTES) { work(j, 0, 1); return; }
        int nt;
        {
            std::lock_guard<std::mutex> lk(mu);
            dst = dst_; srcs = srcs_; offs = offs_; count = count_; to_bf16 = to_bf16_;
            nt = job_ranges = (int)threads.size() + 1;
            pending = (int)threads.size();
            ++generation;
        }
        cv_go.notify_all();
        work(j, 0, nt);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    // grow to n - 1 threads (never shrinks); called by the thread that calls run(), between jobs
    void start(int n) {
        unsigned long long gen;
        {
            std::lock_guard<std::mutex> lk(mu);
            gen = generation;
        }
        try {
            for (int t = (int)threads.size(); t < n - 1; ++t) threads.emplace_back([this, t, gen] { loop(t, gen); });
        } catch (...) {}            // fewer threads than asked for: the ranges adapt
    }
    ~GatherPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_go.notify_all();
        for (auto& th : threads) th.join();
    }
};

// contiguous-range copy of `count` pieces into dst by up to `threads` threads (the body of iefvad_host_gather)
static void host_gather_run(char* dst, const void* const* srcs, const size_t* nbytes, int64_t count, int threads) {
    std::vector<size_t> off((size_t)count + 1);
    off[0] = 0;
    for (int64_t i = 0; i < count; ++i) off[(size_t)i + 1] = off[(size_t)i] + nbytes[i];
    const size_t total = off[(size_t)count];
    int nt = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
    if (total < ((size_t)4 << 20)) nt = 1;
    auto run = [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i)
            if (nbytes[i]) stream_copy(dst + off[(size_t)i], (const char*)srcs[i], nbytes[i]);
        __builtin_ia32_sfence();
    };
    if (nt == 1) { run(0, count); return; }
    std::vector<int64_t> cut((size_t)nt + 1, count);
    cut[0] = 0;
    int k = 1;
    for (int64_t i = 0; i < count && k < nt; ++i)
        if (off[(size_t)i + 1] >= total / nt * k) cut[(size_t)k++] = i + 1;
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < nt; ++t)
            if (cut[(size_t)t + 1] > cut[(size_t)t]) pool.emplace_back(run, cut[(size_t)t], cut[(size_t)t + 1]);
    } catch (...) {
        for (auto& th : pool) th.join();
        run(cut[1], count);
        run(0, cut[1]);
        return;
    }
    run(0, cut[1]);
    for (auto& th : pool) th.join();
}

// the staging form of wire_dtype = BF16 as an entry of its own (what the copy threads of the list walk run): fp32 pieces in, one
// contiguous bf16 stream out
extern "C" int iefvad_host_gather_bf16(void* dst, const void* const* srcs, const size_t* nbytes, int64_t count, int32_t threads) {
    if (count < 0 || (count > 0 && (!dst || !srcs || !nbytes))) return fail("iefvad_host_gather_bf16: null argument");
    if (count == 0) return 0;
    for (int64_t i = 0; i < count; ++i) {
        if (nbytes[i] && !srcs[i]) return fail("iefvad_host_gather_bf16: srcs[%lld] is null", (long long)i);
        if (nbytes[i] % 64) return fail("iefvad_host_gather_bf16: nbytes[%lld] = %zu is not a multiple of 64 (16 fp32 values)", (long long)i, nbytes[i]);
    }
    try {
        std::vector<size_t> off((size_t)count + 1);
        off[0] = 0;
        for (int64_t i = 0; i < count; ++i) off[(size_t)i + 1] = off[(size_t)i] + nbytes[i];
        GatherPool pool;
        pool.start(threads < 1 ? 1 : (threads > 16 ? 16 : threads));
        pool.run((char*)dst, srcs, off.data(), count, true);
    } catch (const std::exception& e) {
        return fail("iefvad_host_gather_bf16: %s", e.what());
    }
    return 0;
}

extern "C" int iefvad_host_gather(void* dst, const void* const* srcs, const size_t* nbytes, int64_t count, int32_t threads) {
    if (count < 0 || (count > 0 && (!dst || !srcs || !nbytes))) return fail("iefvad_host_gather: null argument");
    if (count == 0) return 0;
    for (int64_t i = 0; i < count; ++i)
        if (nbytes[i] && !srcs[i]) return fail("iefvad_host_gather: srcs[%lld] is null", (long long)i);
    try {
        host_gather_run((char*)dst, srcs, nbytes, count, threads);
    } catch (const std::exception& e) {
        return fail("iefvad_host_gather: %s", e.what());
    }
    return 0;
}

