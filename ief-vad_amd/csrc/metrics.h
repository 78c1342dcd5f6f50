// iefvad_auc_ap: the metric tail of the evaluation loop on the device.
//
// The reference ends test() with sklearn on the host (/root/reference/test.py:158-159):
//     ROC1 = roc_auc_score(gt, np.repeat(ap1, 16));  AP1 = average_precision_score(gt, np.repeat(ap1, 16))
// -- a sort of 16 x n points (33.5 M at BASELINE config 4) for numbers that depend on n snippet scores and, per snippet, on HOW
// MANY of its 16 frames are anomalous.  Here: one LSD radix sort of n (score key, positives-of-the-snippet) pairs, one scan, one
// reduction over the tie groups; the x16 repeat is never materialised.
//
//   thresholds = the DISTINCT score values, in decreasing order (sklearn: _binary_clf_curve); a snippet contributes its `repeat`
//   frames at one threshold.  With, per tie group g, P_g / N_g positive / negative frames, tp_g / fp_g the cumulative counts up
//   to and including g, P / N the totals:
//       AUC = sum_g P_g (2 (N - fp_g) + N_g) / (2 P N)        (trapezoids of the ROC curve = Mann-Whitney with ties at 1/2)
//       AP  = sum_g (P_g / P) tp_g / (tp_g + fp_g)            (sklearn: -sum(diff(recall) * precision[:-1]))
//   The AUC numerator is an INTEGER (at most (n repeat)^2 / 2 < 2^63 for n repeat < 2^32 frames): it is accumulated exactly in 64-bit
//   integers, so the result does not depend on any summation order; AP terms are doubles reduced in a fixed order (per-block partials, one
//   final block).  Deterministic, no floating-point atomics.
//
// Kernels are HBM-bound integer / byte work on 8-byte pairs: coalesced tile loads, LDS histograms, wave-level ballots for the
// stable in-tile ranks (wave64: one 64-bit ballot per digit bit); nothing here is reshaped into a GEMM.
#pragma once
#include "common.h"

#define MT_THREADS 256
#define MT_ITEMS 16
#define MT_TILE (MT_THREADS * MT_ITEMS)      // 4096 pairs per workgroup
#define MT_WAVES (MT_THREADS / 64)
#define MT_RADIX 256

struct MetricWs {                  // carved out of the caller's workspace by metric_layout()
    unsigned long long* a;         // [n] pairs: key << 32 | positives
    unsigned long long* b;         // [n] the sort's other buffer; afterwards tp_incl (u32 [n]) | gstart (u32 [n])
    unsigned* hist;                // [MT_RADIX * tiles]
    unsigned* bsum;                // [tiles] positives per tile
    unsigned* bstart;              // [tiles] largest group-start index inside the tile (0 if none)
    double* ap_part;               // [tiles]
    unsigned long long* auc_num;   // [1] exact numerator
    unsigned* flags;               // [1] bit 0: a NaN score was seen
    unsigned* dtotal;              // [MT_RADIX] pairs per digit of the current sort pass
};

// descending order of the scores = ascending order of the keys.  -0.0 is folded into +0.0 (equal as numbers, so one threshold).
__device__ __forceinline__ unsigned metric_key(float s) {
    unsigned u = __float_as_uint(s);
    if ((u << 1) == 0) u = 0;
    const unsigned asc = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    return ~asc;
}

// pairs from scores + frame-level ground truth (the sort kernels take n and mask the tail of the last tile themselves: nothing is padded)
__global__ __launch_bounds__(MT_THREADS) void iefvad_metric_pairs_kernel(const float* scores, const unsigned char* gt, long long n, int repeat,
                                                                        unsigned long long* out, unsigned* flags) {
    const long long i = (long long)blockIdx.x * MT_THREADS + threadIdx.x;
    if (i >= n) return;
    const float s = scores[i];
    unsigned pos = 0;
    const unsigned char* g = gt + i * repeat;
    // any non-zero byte counts as a positive frame (gt is 0 / 1 in every list the reference ships)
    if (repeat == 16 && (((uintptr_t)gt) & 15) == 0) {          // the reference's 16 frames per snippet: one 16-byte load
        const uint4 v = *(const uint4*)g;
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) pos += ((w[k] >> (8 * b)) & 0xFFu) != 0;
    } else {
        for (int k = 0; k < repeat; ++k) pos += g[k] != 0;
    }
    if (s != s) atomicOr(flags, 1u);
    out[i] = ((unsigned long long)metric_key(s) << 32) | pos;
}

// ---- LSD radix sort on the key half, 8 bits per pass ---------------------------------------------------------------------------
// element order inside a tile: wave w owns items [w * 1024, (w + 1) * 1024) of the tile, round r of the wave is 64 consecutive items
__device__ __forceinline__ long long mt_index(long long tile0, int wave, int round, int lane) {
    return tile0 + (long long)wave * (MT_ITEMS * 64) + round * 64 + lane;
}

__global__ __launch_bounds__(MT_THREADS) void iefvad_metric_hist_kernel(const unsigned long long* in, long long n, int shift, unsigned* hist, int tiles) {
    __shared__ unsigned h[MT_RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long long tile0 = (long long)blockIdx.x * MT_TILE;
#pragma unroll
    for (int it = 0; it < MT_ITEMS; ++it) {
        const long long i = tile0 + it * MT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned)(in[i] >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * tiles + blockIdx.x] = h[threadIdx.x];      // digit-major: a digit's tile counts are one contiguous row
}

// The digit-major table [256 digits][tiles] becomes output bases in two levels: one workgroup per DIGIT scans its row of tile counts
// in place (exclusive) and leaves the digit's total; the scatter kernel turns the 256 totals into digit bases itself (a 256-entry scan
// in LDS per workgroup).  (The first build scanned the whole 131 k-word table with ONE workgroup: 72 us per pass, half of the call.)
__global__ __launch_bounds__(256) void iefvad_metric_digit_scan_kernel(unsigned* hist, int tiles, unsigned* dtotal) {
    __shared__ unsigned wsum[4];
    __shared__ unsigned carry_s;
    unsigned* row = hist + (size_t)blockIdx.x * tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < tiles; base += 1024) {
        const int i0 = base + 4 * threadIdx.x;
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i0 + k < tiles) ? row[i0 + k] : 0u;
        const unsigned mine = v[0] + v[1] + v[2] + v[3];
        unsigned inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned wbase = carry_s;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        unsigned run = wbase + inc - mine;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < tiles) row[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 255) carry_s = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) dtotal[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(MT_THREADS) void iefvad_metric_scatter_kernel(const unsigned long long* in, unsigned long long* out, long long n, int shift,
                                                                          const unsigned* hist, int tiles, const unsigned* dtotal) {
    __shared__ unsigned wcount[MT_WAVES][MT_RADIX];        // per wave: first its digit counts, then its running output offsets
    __shared__ unsigned dwave[MT_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile0 = (long long)blockIdx.x * MT_TILE;
#pragma unroll
    for (int w = 0; w < MT_WAVES; ++w) wcount[w][threadIdx.x] = 0;
    __syncthreads();
    unsigned long long e[MT_ITEMS];
#pragma unroll
    for (int r = 0; r < MT_ITEMS; ++r) {
        const long long i = mt_index(tile0, wave, r, lane);
        e[r] = (i < n) ? in[i] : 0ull;
        if (i < n) atomicAdd(&wcount[wave][(unsigned)(e[r] >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    // digit bases: exclusive scan of the 256 digit totals (thread d = digit d)
    const unsigned dt = dtotal[threadIdx.x];
    unsigned dinc = dt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(dinc, d);
        if (lane >= d) dinc += o;
    }
    if (lane == 63) dwave[wave] = dinc;
    __syncthreads();
    unsigned dbase = dinc - dt;
    for (int w = 0; w < wave; ++w) dbase += dwave[w];
    {   // thread d: global base of digit d for this tile, then one running offset per wave (waves in order: stable)
        unsigned run = dbase + hist[(size_t)threadIdx.x * tiles + blockIdx.x];
#pragma unroll
        for (int w = 0; w < MT_WAVES; ++w) {
            const unsigned c = wcount[w][threadIdx.x];
            wcount[w][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
    volatile unsigned* off = wcount[wave];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < MT_ITEMS; ++r) {
        const long long i = mt_index(tile0, wave, r, lane);
        const bool valid = i < n;
        const unsigned d = (unsigned)(e[r] >> shift) & 0xFFu;
        unsigned long long peers = __ballot(valid);        // lanes of this round holding the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        unsigned base = 0;
        const int leader = peers ? __ffsll((long long)peers) - 1 : 0;
        if (valid && lane == leader) {
            base = off[d];
            off[d] = base + (unsigned)__popcll(peers);
        }
        base = __shfl(base, leader);
        if (valid) out[(size_t)base + (unsigned)__popcll(peers & lt)] = e[r];
    }
}

// ---- behind the sort: inclusive positives and, per element, the first index of its tie group -------------------------------------
__global__ __launch_bounds__(MT_THREADS) void iefvad_metric_tile_sums_kernel(const unsigned long long* sorted, long long n, unsigned* bsum, unsigned* bstart) {
    __shared__ unsigned s_sum[MT_WAVES], s_start[MT_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i0 = (long long)blockIdx.x * MT_TILE + (long long)threadIdx.x * MT_ITEMS;
    unsigned sum = 0, start = 0;
    unsigned prev_key = (i0 > 0 && i0 - 1 < n) ? (unsigned)(sorted[i0 - 1] >> 32) : 0u;
#pragma unroll
    for (int k = 0; k < MT_ITEMS; ++k) {
        const long long i = i0 + k;
        if (i < n) {
            const unsigned long long p = sorted[i];
            sum += (unsigned)p;
            const unsigned key = (unsigned)(p >> 32);
            if (i == 0 || key != prev_key) start = (unsigned)i;
            prev_key = key;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d);
        start = max(start, (unsigned)__shfl_xor((int)start, d));
    }
    if (lane == 0) { s_sum[wave] = sum; s_start[wave] = start; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned a = 0, b = 0;
        for (int w = 0; w < MT_WAVES; ++w) { a += s_sum[w]; b = max(b, s_start[w]); }
        bsum[blockIdx.x] = a;
        bstart[blockIdx.x] = b;
    }
}

// one workgroup: bsum -> exclusive sums, bstart -> exclusive running maxima (a group start index grows with the tile, so "max" = "last")
__global__ __launch_bounds__(1024) void iefvad_metric_tile_scan_kernel(unsigned* bsum, unsigned* bstart, int tiles) {
    __shared__ unsigned ws[16], wm[16];
    __shared__ unsigned carry_sum, carry_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { carry_sum = 0; carry_max = 0; }
    __syncthreads();
    for (int base = 0; base < tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned v = i < tiles ? bsum[i] : 0u, m = i < tiles ? bstart[i] : 0u;
        unsigned inc = v, mx = m;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(inc, d), om = __shfl_up(mx, d);
            if (lane >= d) { inc += o; mx = max(mx, om); }
        }
        if (lane == 63) { ws[wave] = inc; wm[wave] = mx; }
        __syncthreads();
        unsigned bs = carry_sum, bm = carry_max;
        for (int w = 0; w < wave; ++w) { bs += ws[w]; bm = max(bm, wm[w]); }
        const unsigned up = (unsigned)__shfl_up(mx, 1);          // every lane takes part in the shuffle
        if (i < tiles) { bsum[i] = bs + inc - v; bstart[i] = lane ? max(bm, up) : bm; }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_sum = bs + inc; carry_max = max(bm, mx); }
        __syncthreads();
    }
}

__global__ __launch_bounds__(MT_THREADS) void iefvad_metric_tile_apply_kernel(const unsigned long long* sorted, long long n, const unsigned* bsum,
                                                                             const unsigned* bstart, unsigned* tp_incl, unsigned* gstart) {
    __shared__ unsigned s_sum[MT_WAVES], s_start[MT_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i0 = (long long)blockIdx.x * MT_TILE + (long long)threadIdx.x * MT_ITEMS;
    unsigned pos[MT_ITEMS], st[MT_ITEMS];
    unsigned sum = 0, start = 0;
    unsigned prev_key = (i0 > 0 && i0 - 1 < n) ? (unsigned)(sorted[i0 - 1] >> 32) : 0u;
#pragma unroll
    for (int k = 0; k < MT_ITEMS; ++k) {
        const long long i = i0 + k;
        pos[k] = 0; st[k] = 0;
        if (i < n) {
            const unsigned long long p = sorted[i];
            const unsigned key = (unsigned)(p >> 32);
            if (i == 0 || key != prev_key) start = (unsigned)i;
            prev_key = key;
            sum += (unsigned)p;
            pos[k] = sum;          // inclusive within the thread
            st[k] = start;         // running maximum within the thread (0 = "none yet in this thread")
        }
    }
    unsigned inc = sum, mx = start;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(inc, d), om = __shfl_up(mx, d);
        if (lane >= d) { inc += o; mx = max(mx, om); }
    }
    if (lane == 63) { s_sum[wave] = inc; s_start[wave] = mx; }
    __syncthreads();
    unsigned bs = bsum[blockIdx.x], bm = bstart[blockIdx.x];
    for (int w = 0; w < wave; ++w) { bs += s_sum[w]; bm = max(bm, s_start[w]); }
    const unsigned up = (unsigned)__shfl_up(mx, 1);
    const unsigned excl_sum = bs + inc - sum, excl_max = lane ? max(bm, up) : bm;
#pragma unroll
    for (int k = 0; k < MT_ITEMS; ++k) {
        const long long i = i0 + k;
        if (i < n) {
            tp_incl[i] = excl_sum + pos[k];
            gstart[i] = max(excl_max, st[k]);
        }
    }
}

// every LAST element of a tie group adds its group's terms
__global__ __launch_bounds__(MT_THREADS) void iefvad_metric_groups_kernel(const unsigned long long* sorted, long long n, int repeat, const unsigned* tp_incl,
                                                                         const unsigned* gstart, unsigned long long* auc_num, double* ap_part) {
    __shared__ double s_ap[MT_WAVES];
    __shared__ unsigned long long s_auc[MT_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long P = tp_incl[n - 1];
    const unsigned long long N = (unsigned long long)n * repeat - P;
    unsigned long long auc = 0;
    double ap = 0.0;
    const long long tile0 = (long long)blockIdx.x * MT_TILE;
#pragma unroll 4
    for (int it = 0; it < MT_ITEMS; ++it) {
        const long long i = tile0 + it * MT_THREADS + threadIdx.x;
        if (i >= n) continue;
        const unsigned key = (unsigned)(sorted[i] >> 32);
        if (i + 1 < n && (unsigned)(sorted[i + 1] >> 32) == key) continue;        // not the end of its group
        const unsigned long long s = gstart[i];
        const unsigned long long tp = tp_incl[i], tp_prev = s ? tp_incl[s - 1] : 0ull;
        const unsigned long long Pg = tp - tp_prev;
        const unsigned long long Ng = (unsigned long long)repeat * (unsigned long long)(i - (long long)s + 1) - Pg;
        const unsigned long long seen = (unsigned long long)repeat * (unsigned long long)(i + 1);      // tp + fp
        const unsigned long long fp = seen - tp;
        auc += Pg * (2ull * (N - fp) + Ng);
        if (Pg) ap += (double)Pg * ((double)tp / (double)seen);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        auc += __shfl_xor(auc, d);
        ap += __shfl_xor(ap, d);
    }
    if (lane == 0) { s_auc[wave] = auc; s_ap[wave] = ap; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0;
        double p = 0.0;
        for (int w = 0; w < MT_WAVES; ++w) { a += s_auc[w]; p += s_ap[w]; }
        if (a) atomicAdd(auc_num, a);          // integers: exact in any order
        ap_part[blockIdx.x] = p;
    }
}

__global__ __launch_bounds__(256) void iefvad_metric_finish_kernel(const unsigned* tp_incl, long long n, int repeat, const unsigned long long* auc_num,
                                                                  const double* ap_part, int tiles, const unsigned* flags, double* auc, double* ap) {
    __shared__ double s[256];
    double p = 0.0;
    for (int i = threadIdx.x; i < tiles; i += 256) p += ap_part[i];          // fixed assignment, fixed tree below
    s[threadIdx.x] = p;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double P = (double)tp_incl[n - 1];
        const double N = (double)n * repeat - P;
        const double nan = __builtin_nan("");
        const bool bad = (*flags & 1u) != 0;
        // one class only: sklearn's roc_auc_score raises; here NaN.  No positive frame: average_precision_score gives 0.
        if (auc) *auc = (bad || P == 0.0 || N == 0.0) ? nan : (double)*auc_num / (2.0 * P * N);
        if (ap) *ap = bad ? nan : (P == 0.0 ? 0.0 : s[0] / P);
    }
}
