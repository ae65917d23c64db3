// Host-side pieces of libpup_hip.so that never touch a kernel: pinned staging memory and the two array passes that turn
// a region's features into engine input.  The reference does these per snippet in Python (control shifts
// coolpuppy/coolpup.py:387-453, the bounds test :1105-1114, one dict per snippet); the vectorised numpy form of them was
// ~40 passes over 10^7 rows, which made the host the slowest part of a pile-up by two orders of magnitude.  Here each is
// one fused, multi-threaded pass that writes straight into pinned memory the engine's DMA copies read.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/pup_hip.h"

#define PUP_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

int n_workers(int64_t rows) {
    const unsigned hw = std::thread::hardware_concurrency();
    int64_t want = rows / 150000 + 1;                     // a thread is worth starting for ~0.15 M rows
    want = std::min<int64_t>(want, std::max(1u, std::min(hw, 16u)));
    return (int)want;
}

template <class F> void parallel_chunks(int64_t n, int workers, F&& f) {
    if (workers <= 1) { f(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)workers);
    for (int k = 0; k < workers; ++k) {
        const int64_t a = n * k / workers, b = n * (k + 1) / workers;
        th.emplace_back([&f, k, a, b] { f(k, a, b); });
    }
    for (auto& t : th) t.join();
}

}  // namespace

PUP_EXPORT int pup_host_alloc(void** ptr, size_t bytes) {
    if (!ptr) return PUP_EINVAL;
    *ptr = nullptr;
    if (bytes == 0) return PUP_OK;
    return hipHostMalloc(ptr, bytes, hipHostMallocDefault) == hipSuccess ? PUP_OK : PUP_ENOMEM;
}

PUP_EXPORT int pup_host_free(void* ptr) {
    if (!ptr) return PUP_OK;
    return hipHostFree(ptr) == hipSuccess ? PUP_OK : PUP_EHIP;
}

// Windows of one region (pair): the n ROI windows followed by nshifts randomly shifted copies of all of them, the copies
// in the reference's order (copy 0 of every window, copy 1 of every window, ...).  shift / sign are the reference's own RNG
// draws, in its order (np.random.randint, np.random.choice([-1, 1]), n * nshifts each); both sides of a control window move
// by round(shift * sign / resolution) bins (half-to-even, numpy's round).  A window is kept when it lies inside its
// region(s): lo1 <= r0 and r0 + h <= hi1, same for columns.  Outputs are compacted in order; returns the number kept and
// *n_roi_kept of them are ROI windows.  code (may be NULL) is carried along: code_out[k] = code of the window's ROI row.
PUP_EXPORT int64_t pup_host_windows(const int32_t* st1, const int32_t* st2, const int32_t* code, int64_t n,
                                    const int64_t* shift, const int64_t* sign, int32_t nshifts, double resolution,
                                    int64_t off1, int64_t off2, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2,
                                    int32_t h, int32_t w, int32_t* r0, int32_t* c0, int32_t* code_out, int64_t* n_roi_kept) {
    if (n < 0 || nshifts < 0 || (n > 0 && (!st1 || !st2 || !r0 || !c0)) || (nshifts > 0 && n > 0 && (!shift || !sign))) return -1;
    const int64_t total = n * (1 + (int64_t)nshifts);
    const int workers = n_workers(total);
    auto window = [&](int64_t k, int64_t& r, int64_t& c) -> bool {
        int64_t row = k, d = 0;
        if (k >= n) {
            const int64_t m = k - n;
            row = m % n;
            d = (int64_t)std::nearbyint((double)(shift[m] * sign[m]) / resolution);
        }
        r = (int64_t)(int32_t)(st1[row] + (int32_t)d) + off1;
        c = (int64_t)(int32_t)(st2[row] + (int32_t)d) + off2;
        return r >= lo1 && r + h <= hi1 && c >= lo2 && c + w <= hi2;
    };
    std::vector<int64_t> kept((size_t)workers + 1, 0), kept_roi((size_t)workers, 0);
    parallel_chunks(total, workers, [&](int k, int64_t a, int64_t b) {
        int64_t cnt = 0, roi = 0, r, c;
        for (int64_t i = a; i < b; ++i) if (window(i, r, c)) { ++cnt; roi += (i < n); }
        kept[(size_t)k + 1] = cnt; kept_roi[(size_t)k] = roi;
    });
    for (int k = 0; k < workers; ++k) kept[(size_t)k + 1] += kept[(size_t)k];
    parallel_chunks(total, workers, [&](int k, int64_t a, int64_t b) {
        int64_t o = kept[(size_t)k], r, c;
        for (int64_t i = a; i < b; ++i) {
            if (!window(i, r, c)) continue;
            r0[o] = (int32_t)r; c0[o] = (int32_t)c;
            if (code_out) code_out[o] = code ? code[i < n ? i : (i - n) % n] : -1;
            ++o;
        }
    });
    if (n_roi_kept) { int64_t s = 0; for (int64_t v : kept_roi) s += v; *n_roi_kept = s; }
    return kept[(size_t)workers];
}

// Gather the windows of several regions into one engine call: stable counting sort by tile id over the concatenation of
// the parts (part order, then order inside the part).  tile_ptr[T + 1] receives the tile boundaries.
PUP_EXPORT int pup_host_group_tiles(int32_t n_parts, const int32_t* const* r0, const int32_t* const* c0,
                                    const int32_t* const* tile, const int64_t* len, int32_t T,
                                    int32_t* r0_out, int32_t* c0_out, int64_t* tile_ptr) {
    if (n_parts < 0 || T <= 0 || !tile_ptr || (n_parts > 0 && (!r0 || !c0 || !tile || !len))) return PUP_EINVAL;
    std::vector<int64_t> counts((size_t)n_parts * T, 0);
    std::vector<int> bad((size_t)std::max(n_parts, 1), 0);
    auto count_part = [&](int p) {
        int64_t* cnt = counts.data() + (size_t)p * T;
        const int32_t* t = tile[p];
        for (int64_t i = 0; i < len[p]; ++i) { const int32_t v = t[i]; if (v < 0 || v >= T) { bad[(size_t)p] = 1; return; } ++cnt[v]; }
    };
    const int workers = std::min<int>(std::max(n_parts, 1), n_workers([&] { int64_t s = 0; for (int p = 0; p < n_parts; ++p) s += len[p]; return s; }()));
    parallel_chunks(n_parts, workers, [&](int, int64_t a, int64_t b) { for (int64_t p = a; p < b; ++p) count_part((int)p); });
    for (int p = 0; p < n_parts; ++p) if (bad[(size_t)p]) return PUP_EINVAL;
    // destination of part p's first window of tile t = windows of smaller tiles + windows of tile t in earlier parts
    int64_t run = 0;
    for (int t = 0; t < T; ++t) {
        tile_ptr[t] = run;
        for (int p = 0; p < n_parts; ++p) { const int64_t k = counts[(size_t)p * T + t]; counts[(size_t)p * T + t] = run; run += k; }
    }
    tile_ptr[T] = run;
    parallel_chunks(n_parts, workers, [&](int, int64_t a, int64_t b) {
        for (int64_t p = a; p < b; ++p) {
            int64_t* dst = counts.data() + (size_t)p * T;
            const int32_t *t = tile[p], *rr = r0[p], *cc = c0[p];
            for (int64_t i = 0; i < len[p]; ++i) { const int64_t o = dst[t[i]]++; r0_out[o] = rr[i]; c0_out[o] = cc[i]; }
        }
    });
    return PUP_OK;
}
