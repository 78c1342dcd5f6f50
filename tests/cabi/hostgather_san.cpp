// CPU-only stress driver for the host side of the whole-video path, built by tests/cabi/Makefile with gcc under
// -fsanitize=thread and again under -fsanitize=address,undefined (the GPU pool offers no sanitizer runs; this code never touches
// the GPU).  It compiles ief-vad_amd/csrc/hostgather.h itself -- the very GatherPool, iefvad_host_gather and
// iefvad_host_gather_bf16 that libiefvad.so ships -- and checks every result against a plain scalar evaluation:
//   1. the vectors of tests/test_cabi_cpu.py (ragged pieces, empty pieces, a piece larger than the threading threshold);
//   2. a 1,000-job loop on ONE persistent pool with 1..16 threads, sources at odd byte offsets, copy and bf16 jobs mixed, the
//      pool GROWING between jobs (start(2) ... start(16): a thread created after a job must not run that job's tables -- they
//      are freed here on purpose before the next start) ;
//   3. the bf16 conversion against a bit-level reference on NaN / inf / denormal / rounding-tie patterns.
// Exit code 0 = all equal; any sanitizer report makes the process exit non-zero by itself (halt_on_error).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

#define IEF_POOL_MIN_BYTES ((size_t)32768)
#include "../../ief-vad_amd/csrc/hostgather.h"

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned long long rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}

// the CHECKER's own loops run uninstrumented (they touch only memory the pool has finished with; instrumenting them is most of
// the run time under -fsanitize=thread); the code under test -- everything in hostgather.h -- stays fully instrumented
#if defined(__SANITIZE_THREAD__)
#define UNCHECKED __attribute__((no_sanitize_thread, noinline))
#else
#define UNCHECKED __attribute__((noinline))
#endif
UNCHECKED static void fill(char* p, size_t n) {
    for (size_t j = 0; j < n; j += 8) {
        const unsigned long long r = rnd();
        memcpy(p + j, &r, n - j < 8 ? n - j : 8);
    }
}
static unsigned short bf16_ref(unsigned u);
UNCHECKED static long first_bf16_mismatch(const char* dst, const char* src, size_t n_elems);

static unsigned short bf16_ref(unsigned u) {
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

UNCHECKED static long first_bf16_mismatch(const char* dst, const char* src, size_t n_elems) {
    for (size_t e = 0; e < n_elems; ++e) {
        unsigned u;
        unsigned short got;
        memcpy(&u, src + 4 * e, 4);
        memcpy(&got, dst + 2 * e, 2);
        const unsigned short want = ((u & 0x7FFFFFFFu) > 0x7F800000u) ? (unsigned short)((u >> 16) | 0x40u)
                                                                       : (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
        if (got != want) return (long)e;
    }
    return -1;
}

#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

static int check_copy(const std::vector<char>& dst, const std::vector<const void*>& srcs, const std::vector<size_t>& nb) {
    size_t pos = 0;
    for (size_t i = 0; i < srcs.size(); ++i) {
        if (nb[i] && memcmp(dst.data() + pos, srcs[i], nb[i])) return 1;
        pos += nb[i];
    }
    return 0;
}

int main() {
    // ---- 1. entry points, small and large, ragged
    {
        std::vector<std::vector<char>> bufs;
        std::vector<const void*> srcs;
        std::vector<size_t> nb;
        const size_t sizes[] = {0, 3072, 1, 6 << 20, 4095, 4096, 12288, 0, (2 << 20) + 17};
        size_t total = 0;
        for (size_t n : sizes) {
            bufs.emplace_back(n + 3);
            fill(bufs.back().data(), n + 3);
            srcs.push_back(n ? bufs.back().data() + 3 : nullptr);       // unaligned sources; a null pointer is fine for an empty piece
            nb.push_back(n);
            total += n;
        }
        for (int threads : {0, 1, 2, 5, 16, 99}) {
            std::vector<char> dst(total + 64, 0x5A);
            CHECK(iefvad_host_gather(dst.data() + 1, srcs.data(), nb.data(), (int64_t)srcs.size(), threads) == 0, "%s", g_err);
            std::vector<char> body(dst.begin() + 1, dst.begin() + 1 + (long)total);
            CHECK(check_copy(body, srcs, nb) == 0, "iefvad_host_gather differs (threads %d)", threads);
            CHECK(dst[0] == 0x5A && dst[1 + total] == 0x5A, "iefvad_host_gather wrote outside its range");
        }
        CHECK(iefvad_host_gather(nullptr, srcs.data(), nb.data(), 2, 1) != 0, "null dst accepted");
        const void* bad[] = {nullptr};
        const size_t one[] = {64};
        std::vector<char> d(64);
        CHECK(iefvad_host_gather(d.data(), bad, one, 1, 1) != 0, "null source of a non-empty piece accepted");
        CHECK(iefvad_host_gather_bf16(d.data(), srcs.data(), nb.data() + 1, 1, 1) != 0, "nbytes not a multiple of 64 accepted");
    }
    // ---- 3. bf16 conversion, bit level
    {
        const unsigned special[] = {0x00000000u, 0x80000000u, 0x7F800000u, 0xFF800000u, 0x7FC00000u, 0xFFC00001u, 0x7F800001u, 0x00000001u,
                                    0x00008000u, 0x00018000u, 0x3F808000u, 0x3F818000u, 0x3F807FFFu, 0x7F7FFFFFu, 0xFF7FFFFFu, 0x7F7F8000u};
        const size_t n = (3 << 20) / 4;         // 3 MB of fp32
        std::vector<unsigned> src(n + 1);
        for (size_t i = 0; i < n + 1; ++i) src[i] = i % 37 < 16 ? special[i % 16] : (unsigned)rnd();
        const void* srcs[2] = {src.data(), (const char*)src.data() + 1024 * 1024};
        const size_t nb[2] = {1024 * 1024, n * 4 - 1024 * 1024};
        for (int threads : {1, 3, 8, 16}) {
            std::vector<unsigned short> dst(n + 16, 0xABCD);
            CHECK(iefvad_host_gather_bf16(dst.data(), srcs, nb, 2, threads) == 0, "%s", g_err);
            const long bad = first_bf16_mismatch((const char*)dst.data(), (const char*)src.data(), n);
            CHECK(bad < 0, "bf16 of %08x: got %04x want %04x (threads %d)", src[(size_t)bad], dst[(size_t)bad], bf16_ref(src[(size_t)bad]), threads);
            CHECK(dst[n] == 0xABCD, "iefvad_host_gather_bf16 wrote past its range");
        }
    }
    // ---- 2. one persistent pool, growing, 1,000 jobs
    {
        GatherPool pool;
        int threads_now = 1;
        for (int job = 0; job < 1000; ++job) {
            if (job % 60 == 0 && threads_now < 16) {       // grow between jobs, as a later forward_videos_host call with more host_threads does
                threads_now = threads_now < 2 ? 2 : threads_now + 3 > 16 ? 16 : threads_now + 3;
                pool.start(threads_now);
            }
            const bool narrow = job % 3 == 1;
            const int pieces = 9 + (int)(rnd() % 16);
            // the tables of a job live on the heap and die with the job: a worker that touched them late would be a use-after-free
            auto* bufs = new std::vector<std::vector<char>>();
            auto* srcs = new std::vector<const void*>();
            auto* offs = new std::vector<size_t>(1, 0);
            const bool big = job % 4 != 3;                   // mostly above the pool's threshold (lowered to 32 KB for this build)
            for (int i = 0; i < pieces; ++i) {
                size_t nbytes = (size_t)(rnd() % (big ? 40000 : 2000));
                nbytes = narrow ? nbytes & ~(size_t)63 : nbytes;
                const size_t skew = narrow ? 4 * (rnd() % 4) : rnd() % 7;     // fp32 sources stay 4-byte aligned, byte sources need not be
                bufs->emplace_back(nbytes + skew + 1);
                fill(bufs->back().data(), nbytes + skew);
                srcs->push_back(bufs->back().data() + skew);
                offs->push_back(offs->back() + nbytes);
            }
            const size_t total = offs->back();
            std::vector<char> dst((narrow ? total / 2 : total) + 96, 0x33);
            char* d = dst.data() + (narrow ? 32 - ((uintptr_t)dst.data() & 31) : 1 + rnd() % 3);
            pool.run(d, srcs->data(), offs->data(), pieces, narrow);
            size_t pos = 0;
            for (int i = 0; i < pieces; ++i) {
                const size_t nbytes = (*offs)[(size_t)i + 1] - (*offs)[(size_t)i];
                if (!narrow) {
                    CHECK(!nbytes || !memcmp(d + pos, (*srcs)[(size_t)i], nbytes), "job %d piece %d differs (threads %d)", job, i, threads_now);
                    pos += nbytes;
                } else {
                    const long bad = first_bf16_mismatch(d + pos, (const char*)(*srcs)[(size_t)i], nbytes / 4);
                    CHECK(bad < 0, "job %d piece %d element %ld: bf16 differs (threads %d)", job, i, bad, threads_now);
                    pos += nbytes / 2;
                }
            }
            CHECK(d[pos] == 0x33, "job %d wrote past its range", job);
            delete offs; delete srcs; delete bufs;
        }
    }
    printf("hostgather_san: all checks passed\n");
    return 0;
}
