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
#include <condition_variable>
#include <functional>
#include <limits>
#include <mutex>
#include <new>
#include <thread>
#include <pthread.h>
#include <cstdlib>
#include <atomic>
#include <vector>

#include "../../include/pup_hip.h"

#define PUP_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

int n_workers(int64_t rows) {
    const unsigned hw = std::thread::hardware_concurrency();
    int64_t want = rows / 60000 + 1;                      // a thread is worth starting for ~60 k rows (~0.1 ms of work against ~20 us)
    want = std::min<int64_t>(want, std::max(1u, std::min(hw, 16u)));
    return (int)want;
}

struct PhasePool {                                   // n - 1 helper threads + the caller: run(f) = f(0) .. f(n - 1), back when all are done
    int n;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<void(int)> job;
    long gen = 0;
    int pending = 0;
    bool stop = false, failed = false;
    explicit PhasePool(int n_) : n(n_ < 1 ? 1 : n_) {
        try {
            for (int k = 1; k < n; ++k) th.emplace_back([this, k] { loop(k); });
        } catch (...) { n = 1 + (int)th.size(); }      // fewer helpers than asked for: the shares are by n
    }
    ~PhasePool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_go.notify_all();
        for (auto& t : th) t.join();
    }
    void loop(int k) {
        long seen = 0;
        for (;;) {
            std::function<void(int)> f;
            {
                std::unique_lock<std::mutex> g(mu);
                cv_go.wait(g, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = job;
            }
            bool bad = false;
            try { f(k); } catch (...) { bad = true; }
            {
                std::lock_guard<std::mutex> g(mu);
                if (bad) failed = true;
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    template <class F> void run(F&& f) {
        if (n > 1) {
            { std::lock_guard<std::mutex> g(mu); job = f; pending = n - 1; ++gen; }
            cv_go.notify_all();
        }
        bool bad = false;
        try { f(0); } catch (...) { bad = true; }
        if (n > 1) { std::unique_lock<std::mutex> g(mu); cv_done.wait(g, [&] { return pending == 0; }); }
        if (bad || failed) { failed = false; throw std::bad_alloc(); }
    }
};

// the process's helper threads for the array passes below (lazily started, min(cores, 16) - 1 of them, asleep between passes).
// Every pass used to start its own threads: ~15 us each, sixteen of them, twice per call — the 46 window passes of a 1e6-pair pile-up
// spent 11 of their 15 ms starting threads (r05 host profile).  One pass at a time owns the pool (a second caller — the draw helper
// runs beside the main thread — starts threads of its own as before); a forked child starts a pool of its own.
PhasePool* g_pool = nullptr;
std::mutex g_pool_busy, g_pool_make;
bool g_pool_atfork = false;
int pool_size() { const unsigned hw = std::thread::hardware_concurrency(); return (int)std::max(1u, std::min(hw ? hw : 1u, 16u)); }
PhasePool* the_pool() {
    std::lock_guard<std::mutex> g(g_pool_make);
    if (!g_pool) {
        if (!g_pool_atfork) {
            g_pool_atfork = true;
            // (the child of a fork has none of the helper threads: it leaves the parent's pool object alone and makes its own)
            pthread_atfork(nullptr, nullptr, [] { g_pool = nullptr; new (&g_pool_busy) std::mutex(); new (&g_pool_make) std::mutex(); });
        }
        g_pool = new PhasePool(pool_size());
    }
    return g_pool;
}

template <class F> void parallel_chunks(int64_t n, int workers, F&& f) {
    if (workers <= 1) { f(0, (int64_t)0, n); return; }
    if (!getenv("COOLPUPPY_AMD_NO_HOST_POOL")) {
        std::unique_lock<std::mutex> own(g_pool_busy, std::try_to_lock);
        if (own.owns_lock()) {
            PhasePool* pool = nullptr;
            try { pool = the_pool(); } catch (...) { pool = nullptr; }
            if (pool && pool->n >= 2) {
                const int w = std::min(workers, pool->n);
                pool->run([&f, w, n](int k) { if (k < w) f(k, n * k / w, n * (k + 1) / w); });
                return;
            }
        }
    }
    std::vector<std::thread> th;
    std::atomic<int> failed{0};
    th.reserve((size_t)workers);
    try {
        for (int k = 0; k < workers; ++k) {
            const int64_t a = n * k / workers, b = n * (k + 1) / workers;
            th.emplace_back([&f, &failed, k, a, b] { try { f(k, a, b); } catch (...) { failed.store(1); } });
        }
    } catch (...) {                       // a thread could not be started: let the started ones finish, then report (PUP_HOST_GUARD)
        for (auto& t : th) t.join();
        throw;
    }
    for (auto& t : th) t.join();
    if (failed.load()) throw std::bad_alloc();
}

// scratch vectors of the calling thread, kept between calls (a pile-up sorts its features once per call, always from the same thread)
std::vector<uint64_t>& scratch_u64(int slot) { static thread_local std::vector<uint64_t> v[3]; return v[slot]; }
std::vector<int64_t>& scratch_i64(int slot) { static thread_local std::vector<int64_t> v[3]; return v[slot]; }

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
// first_copy = 0: the ROI windows, then the copies; 1: the shifted copies only (pup_host_control_windows)
static int64_t host_windows_from(int first_copy, const int32_t* st1, const int32_t* st2, const int32_t* code, int64_t n,
                                 const int32_t* shift, const int32_t* sign, int32_t nshifts, double resolution,
                                 int64_t off1, int64_t off2, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2,
                                 int32_t h, int32_t w, int32_t* r0, int32_t* c0, int32_t* code_out, int64_t* n_roi_kept) {
    if (n < 0 || nshifts < 0 || (n > 0 && (!st1 || !st2 || !r0 || !c0)) || (nshifts > 0 && n > 0 && (!shift || !sign))) return -1;
    const int64_t skip = first_copy ? n : 0;             // windows of the full sequence left out in front
    const int64_t total = n * (1 + (int64_t)nshifts) - skip;
    const int workers = n_workers(total);
    // ONE pass: every worker writes the windows of its share at their uncompacted positions and counts the kept ones; nearly
    // always every window is kept (a control copy only falls off a chromosome's ends) and that is all.  Otherwise the shares
    // are closed up in order (a worker's kept windows move left, never past a share that has not moved yet).  The copies are
    // walked copy by copy, row by row: no division per window.  (Two passes with an `m % n` each: 26 of the 280 ms of a pile-up.)
    std::vector<int64_t> kept((size_t)workers + 1, 0), kept_roi((size_t)workers, 0), lo((size_t)workers + 1, 0);
    for (int k = 0; k <= workers; ++k) lo[(size_t)k] = workers <= 1 ? (k ? total : 0) : total * k / workers;
    parallel_chunks(total, workers, [&](int k, int64_t a, int64_t b) {
        int64_t o = a, roi = 0;
        int64_t i = a + skip;                                                 // (i: position in the full sequence; o: where it is written)
        const int64_t b_full = b + skip;
        while (i < b_full) {
            const int64_t copy = i / n, row0 = i - copy * n;                  // (once per run of rows)
            const int64_t run_end = std::min<int64_t>(b_full, (copy + 1) * n);
            for (int64_t row = row0; i < run_end; ++i, ++row) {
                int64_t d = 0;
                if (copy > 0) {
                    const int64_t m = i - n;
                    d = (int64_t)std::nearbyint((double)((int64_t)shift[m] * (int64_t)sign[m]) / resolution);
                }
                const int64_t r = (int64_t)(int32_t)(st1[row] + (int32_t)d) + off1;
                const int64_t c = (int64_t)(int32_t)(st2[row] + (int32_t)d) + off2;
                if (!(r >= lo1 && r + h <= hi1 && c >= lo2 && c + w <= hi2)) continue;
                r0[o] = (int32_t)r; c0[o] = (int32_t)c;
                if (code_out) code_out[o] = code ? code[row] : -1;
                ++o; roi += copy == 0;
            }
        }
        kept[(size_t)k + 1] = o - a; kept_roi[(size_t)k] = roi;
    });
    {
        int64_t at = kept[1];                                                 // share 0 is in place
        for (int k = 1; k < workers; ++k) {
            const int64_t a = lo[(size_t)k], cnt = kept[(size_t)k + 1];
            if (at != a && cnt > 0) {
                std::memmove(r0 + at, r0 + a, (size_t)cnt * sizeof(int32_t));
                std::memmove(c0 + at, c0 + a, (size_t)cnt * sizeof(int32_t));
                if (code_out) std::memmove(code_out + at, code_out + a, (size_t)cnt * sizeof(int32_t));
            }
            at += cnt;
        }
        kept[(size_t)workers] = at;
    }
    if (n_roi_kept) { int64_t s = 0; for (int64_t v : kept_roi) s += v; *n_roi_kept = s; }
    return kept[(size_t)workers];
}

static int64_t pup_host_windows_impl(const int32_t* st1, const int32_t* st2, const int32_t* code, int64_t n,
                                     const int32_t* shift, const int32_t* sign, int32_t nshifts, double resolution,
                                     int64_t off1, int64_t off2, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2,
                                     int32_t h, int32_t w, int32_t* r0, int32_t* c0, int32_t* code_out, int64_t* n_roi_kept) {
    return host_windows_from(0, st1, st2, code, n, shift, sign, nshifts, resolution, off1, off2, lo1, hi1, lo2, hi2, h, w, r0, c0, code_out, n_roi_kept);
}

// The shifted control copies alone (capacity n * nshifts): a whole pile-up's windows can then be written where the engine call
// wants them — every region's ROI windows first (no draw needed), then region after region its copies as the draws arrive —
// straight into page-locked memory, without a per-region intermediate and the pass that gathers those by tile.
static int64_t pup_host_control_windows_impl(const int32_t* st1, const int32_t* st2, const int32_t* code, int64_t n,
                                             const int32_t* shift, const int32_t* sign, int32_t nshifts, double resolution,
                                             int64_t off1, int64_t off2, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2,
                                             int32_t h, int32_t w, int32_t* r0, int32_t* c0, int32_t* code_out) {
    if (nshifts <= 0 || n <= 0) return n < 0 || nshifts < 0 ? -1 : 0;
    return host_windows_from(1, st1, st2, code, n, shift, sign, nshifts, resolution, off1, off2, lo1, hi1, lo2, hi2, h, w, r0, c0, code_out, nullptr);
}

// iv.take(order) of a frame's numeric columns: dst[c][i] = src[c][order[i]], elements of esize[c] = 1, 2, 4 or 8 bytes, all columns
// in one call, rows shared out to the workers.  (numpy gathers a column at a time on one thread: a random read per element —
// seven columns of 10^6 rows were 55 of the 280 ms of a pile-up.)
static int pup_host_take_rows_impl(int32_t ncols, const void* const* src, void* const* dst, const int32_t* esize,
                                  const int64_t* order, int64_t n, int64_t n_src) {
    if (ncols < 0 || n < 0 || (ncols > 0 && n > 0 && (!src || !dst || !esize || !order))) return PUP_EINVAL;
    for (int c = 0; c < ncols; ++c) if (esize[c] != 1 && esize[c] != 2 && esize[c] != 4 && esize[c] != 8) return PUP_EINVAL;
    std::atomic<int> bad{0};
    parallel_chunks(n, n_workers(n * std::max(ncols, 1) / 2), [&](int, int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) if (order[i] < 0 || order[i] >= n_src) { bad.store(1); return; }
        for (int c = 0; c < ncols; ++c) {
            switch (esize[c]) {
                case 8: { const uint64_t* s = (const uint64_t*)src[c]; uint64_t* d = (uint64_t*)dst[c]; for (int64_t i = a; i < b; ++i) d[i] = s[order[i]]; break; }
                case 4: { const uint32_t* s = (const uint32_t*)src[c]; uint32_t* d = (uint32_t*)dst[c]; for (int64_t i = a; i < b; ++i) d[i] = s[order[i]]; break; }
                case 2: { const uint16_t* s = (const uint16_t*)src[c]; uint16_t* d = (uint16_t*)dst[c]; for (int64_t i = a; i < b; ++i) d[i] = s[order[i]]; break; }
                default: { const uint8_t* s = (const uint8_t*)src[c]; uint8_t* d = (uint8_t*)dst[c]; for (int64_t i = a; i < b; ++i) d[i] = s[order[i]]; break; }
            }
        }
    });
    return bad.load() ? PUP_EINVAL : PUP_OK;
}

// Factorisation of an array of POINTERS by identity: codes[i] = number (in order of first appearance) of the distinct pointer
// values, first[j] = index of the first occurrence of value j.  The chromosome columns of a feature frame are object arrays whose
// million entries point at a few dozen string objects (pandas' readers box equal strings once; so does numpy's fancy indexing of
// a name table): hashing the pointers costs ~3 ns each, hashing the strings through pandas' object table 17 ms per million.  The
// caller merges distinct objects that compare equal (it factorises the `n_uniq` representatives themselves).  Returns n_uniq, or
// -1 when there are more than max_uniq distinct pointers (the caller then takes the general way), -2 for bad arguments.
namespace {

// open-addressing table pointer -> small code, in order of first appearance
struct PtrTable {
    std::vector<uintptr_t> keys; std::vector<int32_t> vals; std::vector<int64_t> first; std::vector<uintptr_t> uniq; size_t cap;
    // (starts small and grows: a chromosome column holds a few dozen names — sixteen tables sized for the 65 536 the interface allows
    // cost more to clear than a million pointers to hash)
    PtrTable() : keys(256, 0), vals(256, -1), cap(256) {}
    size_t slot(uintptr_t p) const {
        size_t h = (size_t)((p >> 4) * 0x9E3779B97F4A7C15ull) & (cap - 1);
        while (vals[h] >= 0 && keys[h] != p) h = (h + 1) & (cap - 1);
        return h;
    }
    void grow() {
        cap <<= 2;
        keys.assign(cap, 0); vals.assign(cap, -1);
        for (size_t j = 0; j < uniq.size(); ++j) { const size_t h = slot(uniq[j]); keys[h] = uniq[j]; vals[h] = (int32_t)j; }
    }
    // code of p (a new one when unseen); -1 when that would exceed max_uniq
    int32_t code(uintptr_t p, int64_t at, int64_t max_uniq) {
        size_t h = slot(p);
        if (vals[h] < 0) {
            if ((int64_t)uniq.size() >= max_uniq) return -1;
            if ((uniq.size() + 1) * 4 > cap) { grow(); h = slot(p); }
            keys[h] = p; vals[h] = (int32_t)uniq.size(); uniq.push_back(p); first.push_back(at);
        }
        return vals[h];
    }
};

}  // namespace

static int64_t pup_host_factorize_ptr_impl(const uintptr_t* ptrs, int64_t n, int32_t* codes, int64_t* first, int64_t max_uniq) {
    if (n < 0 || max_uniq < 1 || (n > 0 && (!ptrs || !codes || !first))) return -2;
    // every worker numbers the pointers of its share on its own (runs of one value — the usual case after a sort by chromosome —
    // skip the table); the shares' dictionaries are then merged in order, which keeps the numbering that of first appearance
    // over the whole array, and the shares' codes are renumbered.  (One thread: 7 ms per 10^6 pointers.)
    const int workers = n_workers(n);
    std::vector<PtrTable> tabs((size_t)workers);
    std::vector<int> over((size_t)workers, 0);
    parallel_chunks(n, workers, [&](int k, int64_t a, int64_t b) {
        PtrTable& t = tabs[(size_t)k];
        // (the last TWO distinct pointers skip the table: a strand column alternates between two string objects at random)
        uintptr_t last = 0, prev = 0; int32_t last_code = -1, prev_code = -1;
        for (int64_t i = a; i < b; ++i) {
            const uintptr_t p = ptrs[i];
            if (last_code >= 0 && p == last) { codes[i] = last_code; continue; }
            if (prev_code >= 0 && p == prev) { codes[i] = prev_code; std::swap(last, prev); std::swap(last_code, prev_code); continue; }
            const int32_t c = t.code(p, i, max_uniq);
            if (c < 0) { over[(size_t)k] = 1; return; }
            prev = last; prev_code = last_code;
            codes[i] = last_code = c; last = p;
        }
    });
    for (int k = 0; k < workers; ++k) if (over[(size_t)k]) return -1;
    PtrTable all;
    std::vector<std::vector<int32_t>> remap((size_t)workers);
    bool changed = false;
    for (int k = 0; k < workers; ++k) {
        const PtrTable& t = tabs[(size_t)k];
        remap[(size_t)k].resize(t.uniq.size());
        for (size_t j = 0; j < t.uniq.size(); ++j) {
            const int32_t c = all.code(t.uniq[j], t.first[j], max_uniq);
            if (c < 0) return -1;
            remap[(size_t)k][j] = c; changed |= c != (int32_t)j;
        }
    }
    if (changed)
        parallel_chunks(n, workers, [&](int k, int64_t a, int64_t b) {
            const int32_t* r = remap[(size_t)k].data();
            for (int64_t i = a; i < b; ++i) codes[i] = r[codes[i]];
        });
    for (size_t j = 0; j < all.uniq.size(); ++j) first[j] = all.first[j];
    return (int64_t)all.uniq.size();
}

// Stable argsort of n keys of `bits` significant bits: order[i] = index of the i-th smallest key, equal keys in index order
// (numpy's argsort(kind="stable")) — a least-significant-digit radix sort, 11 bits per pass, rows shared out to the workers
// (per-worker digit counts, one prefix over digits x workers, a stable scatter).  CoordCreator sorts 10^6 features by one packed
// key per row (chromosome pair | start1 | start2): numpy's single-threaded sort of those was 13 ms of a pile-up.
static int pup_host_argsort_impl(const uint64_t* keys, int64_t n, int32_t bits, int64_t* order) {
    if (n < 0 || bits < 0 || bits > 64 || (n > 0 && (!keys || !order))) return PUP_EINVAL;
    if (n == 0) return PUP_OK;
    constexpr int RB = 11, NB = 1 << RB;
    const int passes = std::max(1, (bits + RB - 1) / RB);
    const int workers = n_workers(n);
    // (scratch kept between calls: three fresh 8 MB vectors per million keys cost more in page faults than the passes themselves)
    std::vector<uint64_t>&ka = scratch_u64(0), &kb = scratch_u64(1);
    std::vector<int64_t>& ob = scratch_i64(0);
    if (ka.size() < (size_t)n) ka.resize((size_t)n);
    if (kb.size() < (size_t)n) kb.resize((size_t)n);
    if (ob.size() < (size_t)n) ob.resize((size_t)n);
    std::vector<int64_t> hist((size_t)workers * NB);
    const uint64_t* ksrc = keys; uint64_t* kdst = ka.data();
    int64_t* osrc = nullptr; int64_t* odst = (passes & 1) ? order : ob.data();     // the last pass must land in `order`
    for (int p = 0; p < passes; ++p) {
        const int sh = p * RB;
        std::fill(hist.begin(), hist.end(), 0);
        parallel_chunks(n, workers, [&](int k, int64_t a, int64_t b) {
            int64_t* h = hist.data() + (size_t)k * NB;
            for (int64_t i = a; i < b; ++i) ++h[(ksrc[i] >> sh) & (NB - 1)];
        });
        int64_t run = 0;
        for (int d = 0; d < NB; ++d)
            for (int k = 0; k < workers; ++k) { int64_t& x = hist[(size_t)k * NB + d]; const int64_t c = x; x = run; run += c; }
        const bool last = p + 1 == passes;
        parallel_chunks(n, workers, [&](int k, int64_t a, int64_t b) {
            int64_t* h = hist.data() + (size_t)k * NB;
            for (int64_t i = a; i < b; ++i) {
                const uint64_t key = ksrc[i];
                const int64_t at = h[(key >> sh) & (NB - 1)]++;
                odst[at] = osrc ? osrc[i] : i;
                if (!last) kdst[at] = key;
            }
        });
        ksrc = kdst; kdst = (kdst == ka.data()) ? kb.data() : ka.data();
        osrc = odst; odst = (odst == order) ? ob.data() : order;
    }
    return PUP_OK;
}

// How many BEDPE rows every view region will hand to the window pass — before anything is sorted: the rows that pass the distance
// filter (centres in double, as pup_host_sort_pairs) and lie, both anchors, inside the region (start >= region start, end < region
// end: the reference's region filter, coolpuppy/coolpup.py:549-562).  The random-shift draws of a pile-up (:420-436) only depend on
// these counts, so they can start while the table is still being sorted.  Regions may overlap; counts[k] for region k of
// chromosome code reg_code[k].
static int pup_host_pair_region_counts_impl(const int64_t* s1, const int64_t* e1, const int64_t* s2, const int64_t* e2, const int32_t* c1,
                                            const int32_t* c2, int64_t n, double mindist, double maxdist, const int32_t* reg_code,
                                            const int64_t* reg_start, const int64_t* reg_end, int32_t n_regions, int64_t* counts) {
    if (n < 0 || n_regions < 0 || (n > 0 && (!s1 || !e1 || !s2 || !e2 || !c1 || !c2)) || (n_regions > 0 && (!reg_code || !reg_start || !reg_end || !counts)))
        return PUP_EINVAL;
    for (int k = 0; k < n_regions; ++k) counts[k] = 0;
    if (n == 0 || n_regions == 0) return PUP_OK;
    int32_t max_code = 0;
    for (int k = 0; k < n_regions; ++k) { if (reg_code[k] < 0) return PUP_EINVAL; max_code = std::max(max_code, reg_code[k]); }
    // regions of a chromosome code: first[code] .. first[code + 1) in `by_code`
    std::vector<int> first((size_t)max_code + 2, 0), by_code((size_t)n_regions);
    for (int k = 0; k < n_regions; ++k) ++first[(size_t)reg_code[k] + 1];
    for (size_t k = 1; k < first.size(); ++k) first[k] += first[k - 1];
    { std::vector<int> at(first.begin(), first.end() - 1); for (int k = 0; k < n_regions; ++k) by_code[(size_t)at[(size_t)reg_code[k]]++] = k; }
    const int workers = n_workers(n);
    std::vector<int64_t> part((size_t)workers * (size_t)n_regions, 0);
    parallel_chunks(n, workers, [&](int w, int64_t a, int64_t b) {
        int64_t* cnt = part.data() + (size_t)w * (size_t)n_regions;
        for (int64_t i = a; i < b; ++i) {
            const int32_t code = c1[i];
            if (code != c2[i] || code < 0 || code > max_code) continue;
            const double ca = (double)(s1[i] + e1[i]) / 2.0, cb = (double)(s2[i] + e2[i]) / 2.0, d = std::fabs(cb - ca);
            if (!(mindist <= d && d <= maxdist)) continue;
            for (int j = first[(size_t)code]; j < first[(size_t)code + 1]; ++j) {
                const int k = by_code[(size_t)j];
                if (s1[i] >= reg_start[k] && e1[i] < reg_end[k] && s2[i] >= reg_start[k] && e2[i] < reg_end[k]) ++cnt[k];
            }
        }
    });
    for (int w = 0; w < workers; ++w) for (int k = 0; k < n_regions; ++k) counts[k] += part[(size_t)w * (size_t)n_regions + (size_t)k];
    return PUP_OK;
}

// CoordCreator.process for BEDPE features in one call (coolpuppy/coolpup.py:296-321 centres and the mindist / maxdist filter,
// :489-527 the sort by (chrom1, chrom2, start1, start2)): which rows stay, in which order, and the coordinate / chromosome-code
// columns in that order.  The numpy form was ~20 passes over the 10^6 rows (two gcd reductions among them); here: one pass
// for the filter and the key ranges, one for the keys, the radix sort above, one gather.
//   c = (start + end) / 2 in double, as numpy computes it; a row stays when mindist <= |c2 - c1| <= maxdist;
//   key = (rank[c1] * nu + rank[c2]) | start1 / g1 | start2 / g2   (g: the greatest common divisor of the kept starts — bin-aligned
//   anchors need fewer bits), sorted stably: rows[i] = source row of sorted row i.
// Returns the number of rows kept; -1 bad arguments; -6 (PUP_ENOTSUP) when a start is negative, a code is out of range or the
// key does not fit 63 bits — the caller then sorts the general way.  *flags: bit 0 = the filter dropped rows, bit 1 = the kept
// rows were not in order already.
static int64_t pup_host_sort_pairs_impl(const int64_t* s1, const int64_t* e1, const int64_t* s2, const int64_t* e2,
                                       const int32_t* c1, const int32_t* c2, int64_t n, const int64_t* rank, int32_t nu,
                                       double mindist, double maxdist, int64_t* rows, int64_t* s1o, int64_t* e1o, int64_t* s2o,
                                       int64_t* e2o, int32_t* c1o, int32_t* c2o, int32_t* flags) {
    if (n < 0 || nu < 1 || !rank || !flags || (n > 0 && (!s1 || !e1 || !s2 || !e2 || !c1 || !c2 || !rows || !s1o || !e1o || !s2o || !e2o || !c1o || !c2o)))
        return PUP_EINVAL;
    *flags = 0;
    if (n == 0) return 0;
    const int workers = n_workers(n);
    struct Part { int64_t kept = 0, max1 = 0, max2 = 0, g1 = 0, g2 = 0; int bad = 0; };
    std::vector<Part> part((size_t)workers);
    auto stays = [&](int64_t i) {
        const double a = (double)(s1[i] + e1[i]) / 2.0, b = (double)(s2[i] + e2[i]) / 2.0, d = std::fabs(b - a);
        return mindist <= d && d <= maxdist;
    };
    auto gcd64 = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
    parallel_chunks(n, workers, [&](int k, int64_t a, int64_t b) {
        Part p;
        for (int64_t i = a; i < b; ++i) {
            if (!stays(i)) continue;
            if (s1[i] < 0 || s2[i] < 0 || c1[i] < 0 || c1[i] >= nu || c2[i] < 0 || c2[i] >= nu) { p.bad = 1; break; }
            ++p.kept;
            p.max1 = std::max(p.max1, s1[i]); p.max2 = std::max(p.max2, s2[i]);
            if (p.g1 != 1) p.g1 = gcd64(s1[i], p.g1);
            if (p.g2 != 1) p.g2 = gcd64(s2[i], p.g2);
        }
        part[(size_t)k] = p;
    });
    int64_t kept = 0, max1 = 0, max2 = 0, g1 = 0, g2 = 0;
    std::vector<int64_t> at((size_t)workers + 1, 0);
    for (int k = 0; k < workers; ++k) {
        const Part& p = part[(size_t)k];
        if (p.bad) return -6;
        at[(size_t)k + 1] = at[(size_t)k] + p.kept;
        kept += p.kept; max1 = std::max(max1, p.max1); max2 = std::max(max2, p.max2);
        g1 = gcd64(p.g1, g1); g2 = gcd64(p.g2, g2);
    }
    if (kept == 0) { *flags = 1; return 0; }
    g1 = std::max<int64_t>(g1, 1); g2 = std::max<int64_t>(g2, 1);
    auto bits = [](uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; };
    const int w0 = bits((uint64_t)nu * (uint64_t)nu - 1), w1 = bits((uint64_t)(max1 / g1)), w2 = bits((uint64_t)(max2 / g2));
    if (w0 + w1 + w2 > 63) return -6;
    std::vector<uint64_t>& key = scratch_u64(2);
    std::vector<int64_t>& idx = scratch_i64(1);
    if (key.size() < (size_t)kept) key.resize((size_t)kept);
    if (idx.size() < (size_t)kept) idx.resize((size_t)kept);
    std::vector<int> unsorted((size_t)workers, 0);
    parallel_chunks(n, workers, [&](int k, int64_t a, int64_t b) {
        int64_t o = at[(size_t)k];
        uint64_t prev = 0; bool have = false;
        for (int64_t i = a; i < b; ++i) {
            if (!stays(i)) continue;
            const uint64_t kk = ((((uint64_t)rank[c1[i]] * (uint64_t)nu + (uint64_t)rank[c2[i]]) << w1 | (uint64_t)(s1[i] / g1)) << w2) | (uint64_t)(s2[i] / g2);
            if (have && kk < prev) unsorted[(size_t)k] = 1;
            prev = kk; have = true;
            key[(size_t)o] = kk; idx[(size_t)o] = i; ++o;
        }
    });
    bool permute = false;
    for (int k = 0; k < workers; ++k) permute |= unsorted[(size_t)k] != 0;
    for (int k = 1; k < workers && !permute; ++k) {                                  // across the shares' borders
        const int64_t a = at[(size_t)k];
        if (a > 0 && a < kept && part[(size_t)k].kept > 0 && key[(size_t)a] < key[(size_t)a - 1]) permute = true;
    }
    if (permute) {
        std::vector<int64_t>& ord = scratch_i64(2);
        if (ord.size() < (size_t)kept) ord.resize((size_t)kept);
        const int rc = pup_host_argsort(key.data(), kept, w0 + w1 + w2, ord.data());
        if (rc != PUP_OK) return rc;
        parallel_chunks(kept, workers, [&](int, int64_t a, int64_t b) { for (int64_t i = a; i < b; ++i) rows[i] = idx[(size_t)ord[(size_t)i]]; });
    } else {
        std::memcpy(rows, idx.data(), (size_t)kept * sizeof(int64_t));
    }
    parallel_chunks(kept, workers, [&](int, int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const int64_t r = rows[i];
            s1o[i] = s1[r]; e1o[i] = e1[r]; s2o[i] = s2[r]; e2o[i] = e2[r]; c1o[i] = c1[r]; c2o[i] = c2[r];
        }
    });
    *flags = (kept != n ? 1 : 0) | (permute ? 2 : 0);
    return kept;
}

// ---- the reference's random draws, at memory speed ----------------------------------------------------------------------
// CoordCreator._control_regions (coolpuppy/coolpup.py:420-436) draws the shifts of the control windows with the LEGACY numpy
// generator — np.random.randint(minshift, maxshift, m) and np.random.choice([-1, 1], m), m = windows x nshifts — and a seeded
// pile-up is only reproducible if exactly those numbers come out.  numpy produces them one call per number (~5 ns each:
// 0.1 s of a 0.37 s pile-up of 10^7 control windows).  The stream is fully specified: MT19937 (Matsumoto & Nishimura 1998,
// numpy/random/src/mt19937), one tempered 32-bit word per candidate, `& mask` with the smallest all-ones mask covering
// rng = high - low - 1, candidates above rng rejected (numpy/random/src/distributions: random_bounded_uint64_fill with
// use_masked, the legacy RandomState path).  Here the raw state words are generated block by block (the twist vectorises),
// tempered / masked / counted by several threads, and the accepted candidates written in order; the generator state
// (key[624], pos) is returned as numpy would have left it.
namespace {

constexpr int kMtN = 624, kMtM = 397;

// next block of raw state words from the previous one (out of place: every term of the recurrence is either an old word or
// a new word at least 227 places back, so the loops vectorise)
__attribute__((target_clones("avx512f", "avx2", "default")))
void mt_twist(const uint32_t* __restrict__ old, uint32_t* __restrict__ nw) {
    auto mix = [](uint32_t a, uint32_t b, uint32_t c) -> uint32_t {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        return c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    };
    // three runs without a dependence inside: [0, 227) from old words only, [227, 454) from old words and new [0, 227), [454, 623) from
    // old words and new [227, 396) — spelled as separate source / destination pointers so that the vectoriser sees it (one loop over
    // [227, 623) with a dependence distance of 227: half the speed under clang)
    constexpr int D = kMtN - kMtM;                    // 227
    for (int k = 0; k < D; ++k) nw[k] = mix(old[k], old[k + 1], old[k + kMtM]);
    { const uint32_t* __restrict__ src = nw; uint32_t* __restrict__ dst = nw + D;
      for (int k = 0; k < D; ++k) dst[k] = mix(old[k + D], old[k + D + 1], src[k]); }
    { const uint32_t* __restrict__ src = nw + D; uint32_t* __restrict__ dst = nw + 2 * D;
      for (int k = 0; k < kMtN - 1 - 2 * D; ++k) dst[k] = mix(old[k + 2 * D], old[k + 2 * D + 1], src[k]); }
    nw[kMtN - 1] = mix(old[kMtN - 1], nw[0], nw[kMtM - 1]);
}

inline uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

// how many of raw[a, b) give a candidate within the range (the raw words stay: they are the generator's state)
__attribute__((target_clones("avx512f", "avx2", "default")))
int64_t mt_temper_count(const uint32_t* __restrict__ raw, int64_t a, int64_t b, uint32_t mask, uint32_t rng) {
    int64_t c = 0;
    for (int64_t i = a; i < b; ++i) c += ((mt_temper(raw[i]) & mask) <= rng);
    return c;
}

// scratch for the raw words, kept between calls (a fresh 40 MB buffer costs its page faults every time)
// (process-wide: the draws of a pile-up come from a helper thread that lives for that call only — a thread_local buffer was a fresh
// 40 MB and its page faults every time.  The legacy generator is one sequence: its callers are serialised anyway; the mutex makes it so.)
std::vector<uint32_t>& mt_scratch() { static std::vector<uint32_t> v; return v; }
std::mutex& mt_mutex() { static std::mutex m; return m; }

}  // namespace

// out[i] = offset + scale * (low + d_i), d_i the i-th number np.random.randint(low, high, m) would draw from the legacy
// generator whose state is (key[624], *pos); out: int32 or int64 elements (out_bytes), or NULL (draw and discard).  The state is advanced exactly as numpy
// advances it.  Returns 0, or < 0 for bad arguments (needs 0 < high - low <= 2^32, 0 <= *pos <= 624).
static int pup_host_mt_randint_impl(uint32_t* key, int32_t* pos, int64_t low, int64_t high, int64_t m, int64_t scale,
                                   int64_t offset, void* out_any, int32_t out_bytes) {
    if (out_any && out_bytes != 4 && out_bytes != 8) return PUP_EINVAL;
    int64_t* out = (out_any && out_bytes == 8) ? static_cast<int64_t*>(out_any) : nullptr;
    int32_t* out32 = (out_any && out_bytes == 4) ? static_cast<int32_t*>(out_any) : nullptr;
    if (!key || !pos || *pos < 0 || *pos > kMtN || m < 0 || high <= low || (uint64_t)(high - low - 1) > 0xffffffffull) return PUP_EINVAL;
    if (m == 0) return PUP_OK;
    const uint64_t rng = (uint64_t)(high - low - 1);
    if (rng == 0) {                                       // numpy draws nothing for a single-valued range
        if (out) for (int64_t i = 0; i < m; ++i) out[i] = offset + scale * low;
        if (out32) for (int64_t i = 0; i < m; ++i) out32[i] = (int32_t)(offset + scale * low);
        return PUP_OK;
    }
    uint32_t mask = (uint32_t)rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    const bool all = rng == 0xffffffffull;               // (numpy takes the words as they come)
    // raw words: the unread tail of the current block, then freshly twisted blocks, in rounds until m candidates are accepted
    std::lock_guard<std::mutex> guard(mt_mutex());
    std::vector<uint32_t>& raw = mt_scratch();
    const double p_accept = ((double)rng + 1.0) / ((double)mask + 1.0);
    int64_t done = 0;                                     // accepted so far
    int64_t consumed = 0;                                 // raw words consumed so far (over all rounds)
    std::vector<uint32_t> cur(key, key + kMtN);           // the block the generator currently reads from
    int cur_pos = *pos;
    while (done < m) {
        const int64_t need = m - done;
        int64_t want = (int64_t)((double)need / p_accept * 1.002) + 4096;          // words to look at this round
        const int64_t tail = kMtN - cur_pos;
        const int64_t nblocks = want > tail ? (want - tail + kMtN - 1) / kMtN : 0;
        const int64_t words = tail + nblocks * kMtN;
        if (raw.size() < (size_t)words) raw.resize((size_t)words);
        std::memcpy(raw.data(), cur.data() + cur_pos, (size_t)tail * 4);
        {
            const uint32_t* prev = cur.data();
            for (int64_t b = 0; b < nblocks; ++b) {
                uint32_t* nw = raw.data() + tail + b * kMtN;
                mt_twist(prev, nw);
                prev = nw;
            }
        }
        // accepted candidates per chunk, then their places
        const int workers = n_workers(words);
        std::vector<int64_t> acc((size_t)workers + 1, 0);
        parallel_chunks(words, workers, [&](int k, int64_t a, int64_t b) {
            acc[(size_t)k + 1] = mt_temper_count(raw.data(), a, b, mask, (uint32_t)rng);   // (rng = 2^32 - 1: everything counts)
        });
        for (int k = 0; k < workers; ++k) acc[(size_t)k + 1] += acc[(size_t)k];
        // the chunk in which the last needed candidate falls ends the round; chunks behind it are not consumed
        std::vector<int64_t> last_word((size_t)workers, -1);     // per chunk: index of the word that completed the request
        parallel_chunks(words, workers, [&](int k, int64_t a, int64_t b) {
            int64_t o = acc[(size_t)k];
            if (o >= need) return;
            for (int64_t i = a; i < b; ++i) {
                const uint32_t v = mt_temper(raw[(size_t)i]) & mask;
                if (!all && (uint64_t)v > rng) continue;
                if (out) out[done + o] = offset + scale * (low + (int64_t)v);
                if (out32) out32[done + o] = (int32_t)(offset + scale * (low + (int64_t)v));
                if (++o == need) { last_word[(size_t)k] = i; return; }
            }
        });
        int64_t used = words;                              // all looked at, unless the request completed inside
        const int64_t got = std::min<int64_t>(acc[(size_t)workers], need);
        for (int k = 0; k < workers; ++k) if (last_word[(size_t)k] >= 0) used = last_word[(size_t)k] + 1;
        done += got; consumed += used;
        // the block the generator reads from after `used` words of this round
        if (used <= tail) cur_pos += (int)used;
        else {
            const int64_t u = used - tail, b = (u - 1) / kMtN;
            std::vector<uint32_t> nb(raw.begin() + tail + b * kMtN, raw.begin() + tail + (b + 1) * kMtN);
            cur.swap(nb);
            cur_pos = (int)(u - b * kMtN);
        }
    }
    (void)consumed;
    std::memcpy(key, cur.data(), (size_t)kMtN * 4);
    *pos = cur_pos;
    return PUP_OK;
}

// ---- a whole pile-up's draws in one call ----------------------------------------------------------------------------------
// The reference draws region after region (coolpuppy/coolpup.py:420-436): randint(minshift, maxshift, m_r), then choice([-1, 1], m_r)
// — 46 calls for a 23-chromosome pile-up, every one of which twisted its raw words on ONE thread and then started sixteen
// threads twice (count, place): of the 45 ms the 2 x 10^7 draws of the bench pile-up took, the sequential twist was ~14 and the
// thread starts ~20.  The sizes of all calls are known before the first draw (pup_host_pair_region_counts), so the sequence is
// drawn as ONE job: the raw stream is produced buffer by buffer (kPlanBlocks blocks of 624 words) by a twister thread that runs
// one buffer ahead; a pool of workers that lives for the call takes each buffer in two passes — accepted candidates per chunk
// under the rejection mask, then, once the calling thread has walked the calls over those counts (a call ends at its last
// ACCEPTED word: the exact end is found by a scan inside one chunk), the tempered / masked / scaled numbers written to where
// each call wants them.  Same stream, same numbers, same final state as the per-call form (tests/test_host_misc.py).
// Calls that reject (range not a power of two) must share one range — true of a pile-up's draws; else PUP_ENOTSUP and the
// caller draws call by call.
namespace {

constexpr int kPlanChunkBlocks = 16;                  // blocks per chunk: the unit of the counts and of the workers' shares
constexpr int64_t kPlanChunk = (int64_t)kPlanChunkBlocks * kMtN;
std::vector<uint32_t>& mt_plan_buffer(int which) { static std::vector<uint32_t> v[2]; return v[which]; }

struct PlanPiece { int32_t call; int64_t a, b, out_pos; };      // words [a, b) of the buffer belong to `call`; its accepted ones go to out[out_pos ...]

template <class T>
void plan_write(T* out, const uint32_t* __restrict__ raw, int64_t a, int64_t b, uint32_t mask, uint32_t rng, bool rej,
                int64_t low, int64_t scale, int64_t offset) {
    if (!rej) { for (int64_t i = a; i < b; ++i) *out++ = (T)(offset + scale * (low + (int64_t)(mt_temper(raw[i]) & mask))); return; }
    for (int64_t i = a; i < b; ++i) {
        const uint32_t v = mt_temper(raw[i]) & mask;
        if (v <= rng) *out++ = (T)(offset + scale * (low + (int64_t)v));
    }
}

}  // namespace

static int pup_host_mt_randint_plan_impl(uint32_t* key, int32_t* pos, int32_t n_calls, const int64_t* low, const int64_t* high,
                                        const int64_t* m, const int64_t* scale, const int64_t* offset, void* const* out,
                                        const int32_t* out_bytes) {
    if (!key || !pos || *pos < 0 || *pos > kMtN || n_calls < 0 || (n_calls > 0 && (!low || !high || !m || !scale || !offset || !out || !out_bytes)))
        return PUP_EINVAL;
    struct Call { uint32_t rng, mask; bool rej, none; };
    std::vector<Call> calls((size_t)n_calls);
    bool have_rej = false; uint32_t rng_rej = 0, mask_rej = 0;
    double words_est = 0.0;
    for (int k = 0; k < n_calls; ++k) {
        if (m[k] < 0 || high[k] <= low[k] || (uint64_t)(high[k] - low[k] - 1) > 0xffffffffull) return PUP_EINVAL;
        if (out[k] && out_bytes[k] != 4 && out_bytes[k] != 8) return PUP_EINVAL;
        Call c;
        c.rng = (uint32_t)(high[k] - low[k] - 1);
        uint32_t mk = c.rng;
        mk |= mk >> 1; mk |= mk >> 2; mk |= mk >> 4; mk |= mk >> 8; mk |= mk >> 16;
        c.mask = mk; c.none = c.rng == 0; c.rej = c.rng != mk;
        if (c.rej && m[k] > 0) {
            if (have_rej && rng_rej != c.rng) return PUP_ENOTSUP;
            have_rej = true; rng_rej = c.rng; mask_rej = mk;
        }
        if (!c.none) words_est += (double)m[k] * (((double)mk + 1.0) / ((double)c.rng + 1.0));
        calls[(size_t)k] = c;
    }
    std::lock_guard<std::mutex> guard(mt_mutex());
    // buffer size: an eighth of the job, between 256 and 4096 blocks (10 MB), whole chunks
    int64_t NB = (int64_t)(words_est / kMtN / 8.0) + 1;
    NB = std::max<int64_t>(256, std::min<int64_t>(4096, NB));
    NB = (NB + kPlanChunkBlocks - 1) / kPlanChunkBlocks * kPlanChunkBlocks;
    const int64_t NBW = NB * kMtN, n_chunks = NB / kPlanChunkBlocks;
    for (int b = 0; b < 2; ++b) if (mt_plan_buffer(b).size() < (size_t)NBW) mt_plan_buffer(b).resize((size_t)NBW);
    uint32_t* const buf[2] = {mt_plan_buffer(0).data(), mt_plan_buffer(1).data()};
    std::atomic<bool> stop{false};
    // buffer r = blocks [r NB, (r + 1) NB) of the stream whose block 0 is the incoming key
    auto fill = [&](int64_t r) {
        uint32_t* dst = buf[r & 1];
        const uint32_t* prev;
        int64_t b0 = 0;
        if (r == 0) { std::memcpy(dst, key, (size_t)kMtN * 4); prev = dst; b0 = 1; }
        else prev = buf[(r - 1) & 1] + (NB - 1) * kMtN;
        for (int64_t b = b0; b < NB; ++b) {
            if ((b & 63) == 0 && stop.load(std::memory_order_relaxed)) return;
            mt_twist(prev, dst + b * kMtN);
            prev = dst + b * kMtN;
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    PhasePool pool((int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(hw ? hw : 1, 16), (int64_t)(words_est / 200000.0) + 1)));
    const int NWK = pool.n;
    std::vector<int64_t> cnt((size_t)n_chunks + 1, 0);
    std::vector<PlanPiece> pieces;
    std::vector<uint32_t> carry((size_t)kMtN);           // last block of a buffer consumed to its end
    bool have_carry = false;
    int k = 0;
    int64_t donek = 0;
    int64_t i = *pos;                                     // next unread word of the current buffer
    const int64_t i_start = i;
    int64_t r = 0;
    bool consumed_any = false;
    fill(0);
    auto accepted = [&](const uint32_t* raw, int64_t j) -> bool { return (mt_temper(raw[j]) & mask_rej) <= rng_rej; };
    for (;; ++r) {
        const uint32_t* raw = buf[r & 1];
        std::thread tw;
        bool tw_started = false;
        try { tw = std::thread([&fill, r] { fill(r + 1); }); tw_started = true; } catch (...) {}
        struct Join { std::thread& t; std::atomic<bool>& s; bool on; ~Join() { if (on) { s.store(true); t.join(); s.store(false); } } } joiner{tw, stop, tw_started};
        // pass 1: accepted candidates per chunk under the rejection mask (only while a rejecting call is still to come)
        bool rej_left = false;
        for (int q = k; q < n_calls && !rej_left; ++q) rej_left = calls[(size_t)q].rej && m[q] > 0;
        if (rej_left)
            pool.run([&](int w) {
                for (int64_t c = n_chunks * w / NWK; c < n_chunks * (w + 1) / NWK; ++c)
                    cnt[(size_t)c] = mt_temper_count(raw, c * kPlanChunk, (c + 1) * kPlanChunk, mask_rej, rng_rej);
            });
        // the calls walked over the buffer
        pieces.clear();
        while (k < n_calls && i < NBW) {
            const Call& cl = calls[(size_t)k];
            if (cl.none || m[k] == donek) {
                if (cl.none && out[k]) {
                    const int64_t val = offset[k] + scale[k] * low[k];
                    if (out_bytes[k] == 8) { int64_t* o = static_cast<int64_t*>(out[k]); for (int64_t t = 0; t < m[k]; ++t) o[t] = val; }
                    else { int32_t* o = static_cast<int32_t*>(out[k]); for (int64_t t = 0; t < m[k]; ++t) o[t] = (int32_t)val; }
                }
                ++k; donek = 0; continue;
            }
            const int64_t need = m[k] - donek;
            const int64_t cend = std::min<int64_t>((i / kPlanChunk + 1) * kPlanChunk, NBW);
            if (!cl.rej) {
                const int64_t take = std::min<int64_t>(need, cend - i);
                pieces.push_back({k, i, i + take, donek});
                donek += take; i += take;
            } else if (i % kPlanChunk == 0 && cnt[(size_t)(i / kPlanChunk)] < need) {
                pieces.push_back({k, i, cend, donek});
                donek += cnt[(size_t)(i / kPlanChunk)]; i = cend;
            } else {                                      // a partial chunk, or the chunk the call ends in: scan
                int64_t got = 0, j = i;
                while (j < cend && got < need) { got += accepted(raw, j) ? 1 : 0; ++j; }
                pieces.push_back({k, i, j, donek});
                donek += got; i = j;
            }
            consumed_any = true;
            if (donek == m[k]) { ++k; donek = 0; }
        }
        while (k < n_calls && (calls[(size_t)k].none || m[k] == 0)) {      // trailing calls that take no words
            const Call& cl = calls[(size_t)k];
            if (cl.none && out[k]) {
                const int64_t val = offset[k] + scale[k] * low[k];
                if (out_bytes[k] == 8) { int64_t* o = static_cast<int64_t*>(out[k]); for (int64_t t = 0; t < m[k]; ++t) o[t] = val; }
                else { int32_t* o = static_cast<int32_t*>(out[k]); for (int64_t t = 0; t < m[k]; ++t) o[t] = (int32_t)val; }
            }
            ++k;
        }
        // pass 2: the numbers
        const int64_t np = (int64_t)pieces.size();
        if (np > 0)
            pool.run([&](int w) {
                for (int64_t q = np * w / NWK; q < np * (w + 1) / NWK; ++q) {
                    const PlanPiece& pc = pieces[(size_t)q];
                    const Call& cl = calls[(size_t)pc.call];
                    void* o = out[pc.call];
                    if (!o) continue;
                    if (out_bytes[pc.call] == 8) plan_write(static_cast<int64_t*>(o) + pc.out_pos, raw, pc.a, pc.b, cl.mask, cl.rng, cl.rej, low[pc.call], scale[pc.call], offset[pc.call]);
                    else plan_write(static_cast<int32_t*>(o) + pc.out_pos, raw, pc.a, pc.b, cl.mask, cl.rng, cl.rej, low[pc.call], scale[pc.call], offset[pc.call]);
                }
            });
        if (k >= n_calls) break;                          // (the twister is stopped and joined by `joiner`)
        // the buffer is used up: the next one must be complete; keep this one's last block (the state if nothing more is read)
        std::memcpy(carry.data(), raw + (NB - 1) * kMtN, (size_t)kMtN * 4); have_carry = true;
        if (!tw_started) fill(r + 1);
        else { joiner.on = false; tw.join(); }
        i = 0;
    }
    // the generator's state: the block holding the last word read, and the place behind it
    if (consumed_any) {
        if (i > 0) {
            const int64_t bl = (i - 1) / kMtN;
            std::memcpy(key, buf[r & 1] + bl * kMtN, (size_t)kMtN * 4);
            *pos = (int32_t)(i - bl * kMtN);
        } else if (have_carry) {
            std::memcpy(key, carry.data(), (size_t)kMtN * 4);
            *pos = kMtN;
        }
    } else *pos = (int32_t)i_start;
    return PUP_OK;
}

// Gather the windows of several regions into one engine call: stable counting sort by tile id over the concatenation of
// the parts (part order, then order inside the part).  tile_ptr[T + 1] receives the tile boundaries.
// Parts whose tile array is NULL are RUN-CODED: windows [0, split[p]) of part p belong to tile tile_a[p], the rest to tile_b[p] (an
// ungrouped region with controls: its ROI windows, then their shifted copies) — no per-window tile array is built or read for them,
// and their windows move as two block copies.  (np.zeros + an in-place add of 1.1e7 tile numbers was 30 ms of a pile-up's host time.)
static int pup_host_group_tiles_runs_impl(int32_t n_parts, const int32_t* const* r0, const int32_t* const* c0,
                                         const int32_t* const* tile, const int64_t* split, const int32_t* tile_a, const int32_t* tile_b,
                                         const int64_t* len, int32_t T, int32_t* r0_out, int32_t* c0_out, int64_t* tile_ptr) {
    if (n_parts < 0 || T <= 0 || !tile_ptr || (n_parts > 0 && (!r0 || !c0 || !tile || !len))) return PUP_EINVAL;
    std::vector<int64_t> counts((size_t)n_parts * T, 0);
    std::vector<int> bad((size_t)std::max(n_parts, 1), 0);
    auto count_part = [&](int p) {
        int64_t* cnt = counts.data() + (size_t)p * T;
        const int32_t* t = tile[p];
        if (!t) {
            if (!split || !tile_a || !tile_b || split[p] < 0 || split[p] > len[p] ||
                (split[p] > 0 && (tile_a[p] < 0 || tile_a[p] >= T)) || (split[p] < len[p] && (tile_b[p] < 0 || tile_b[p] >= T))) { bad[(size_t)p] = 1; return; }
            if (split[p] > 0) cnt[tile_a[p]] += split[p];
            if (split[p] < len[p]) cnt[tile_b[p]] += len[p] - split[p];
            return;
        }
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
            if (!t) {
                const int64_t sp = split[p], rest = len[p] - sp;
                if (sp > 0) { std::memcpy(r0_out + dst[tile_a[p]], rr, (size_t)sp * 4); std::memcpy(c0_out + dst[tile_a[p]], cc, (size_t)sp * 4); dst[tile_a[p]] += sp; }
                if (rest > 0) { std::memcpy(r0_out + dst[tile_b[p]], rr + sp, (size_t)rest * 4); std::memcpy(c0_out + dst[tile_b[p]], cc + sp, (size_t)rest * 4); dst[tile_b[p]] += rest; }
                continue;
            }
            for (int64_t i = 0; i < len[p]; ++i) { const int64_t o = dst[t[i]]++; r0_out[o] = rr[i]; c0_out[o] = cc[i]; }
        }
    });
    return PUP_OK;
}

PUP_EXPORT int pup_host_group_tiles(int32_t n_parts, const int32_t* const* r0, const int32_t* const* c0,
                                    const int32_t* const* tile, const int64_t* len, int32_t T,
                                    int32_t* r0_out, int32_t* c0_out, int64_t* tile_ptr) {
    for (int p = 0; p < n_parts; ++p) if (tile && !tile[p]) return PUP_EINVAL;
    return pup_host_group_tiles_runs(n_parts, r0, c0, tile, nullptr, nullptr, nullptr, len, T, r0_out, c0_out, tile_ptr);
}

// ---- the finaliser's arithmetic on whole tile arrays -----------------------------------------------------------------------
// data = (sum / num) [/ (control_sum / control_num)], +inf -> NaN (coolpuppy/coolpup.py:1533-1545), element by element in place of
// `sum`, in numpy's order of operations (so that the frame equals the per-row form bit for bit).  A by-window pile-up has a tile
// per feature — 1.6e7 cells for 37 k CTCF sites: numpy's four single-threaded passes were 20 ms of a 90 ms call.
static int pup_host_normalise_tiles_impl(double* sum, const int64_t* num, double* csum, const int64_t* cnum, int64_t count) {
    if (count < 0 || (count > 0 && (!sum || !num)) || ((csum == nullptr) != (cnum == nullptr))) return PUP_EINVAL;
    const double inf = std::numeric_limits<double>::infinity(), qnan = std::numeric_limits<double>::quiet_NaN();
    parallel_chunks(count, n_workers(count / 4), [&](int, int64_t a, int64_t b) {
        if (csum) {
            for (int64_t i = a; i < b; ++i) {
                const double c = csum[i] / (double)cnum[i];
                csum[i] = c;                                 // (the caller's control array holds the quotient afterwards, as numpy's out=)
                const double v = (sum[i] / (double)num[i]) / c;
                sum[i] = v == inf ? qnan : v;
            }
        } else {
            for (int64_t i = a; i < b; ++i) { const double v = sum[i] / (double)num[i]; sum[i] = v == inf ? qnan : v; }
        }
    });
    return PUP_OK;
}

// ---- two small passes of the grouped plan ------------------------------------------------------------------------------------
// tile numbers of a region's windows: out[i] = lut[codes[i]] (+ add from window add_from on: the controls' half of the tiles) — numpy's two
// single-threaded gathers over 10^7 codes were 14 ms of a 1e6-pair by-distance x by-strand call
static int pup_host_lut_i32_impl(const int32_t* lut, int64_t n_lut, const int32_t* codes, int64_t n, int64_t add_from, int32_t add, int32_t* out) {
    if (n < 0 || n_lut < 0 || (n > 0 && (!lut || !codes || !out || n_lut == 0))) return PUP_EINVAL;
    std::atomic<int> bad{0};
    parallel_chunks(n, n_workers(n / 2), [&](int, int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const int32_t c = codes[i];
            if (c < 0 || c >= n_lut) { bad.store(1); return; }
            out[i] = lut[c] + (i >= add_from ? add : 0);
        }
    });
    return bad.load() ? PUP_ERANGE : PUP_OK;
}

// np.searchsorted(edges, values, side="right") for a short sorted edge list (distance bands, coolpuppy/coolpup.py:28-51): the number of
// edges <= value, by counting (NaN compares false everywhere: 0... numpy sorts NaN last and answers n_edges — the caller never passes NaN)
static int pup_host_count_le_impl(const double* edges, int32_t n_edges, const double* values, int64_t n, int32_t* out) {
    if (n < 0 || n_edges < 0 || n_edges > 64 || (n > 0 && (!values || !out)) || (n_edges > 0 && !edges)) return PUP_EINVAL;
    for (int k = 1; k < n_edges; ++k) if (!(edges[k - 1] <= edges[k])) return PUP_EINVAL;
    parallel_chunks(n, n_workers(n / 2), [&](int, int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const double v = values[i];
            int32_t c = 0;
            for (int k = 0; k < n_edges; ++k) c += edges[k] <= v ? 1 : 0;
            out[i] = c;
        }
    });
    return PUP_OK;
}

// No exception may cross the C boundary (a std::bad_alloc of the scratch vectors, a std::system_error of a thread that could not
// be started inside a container's limits): the entry points catch everything and report an error code; the callers fall back to numpy.
#define PUP_HOST_GUARD(call, err) try { return call; } catch (...) { return err; }

PUP_EXPORT int64_t pup_host_windows(const int32_t* st1, const int32_t* st2, const int32_t* code, int64_t n, const int32_t* shift, const int32_t* sign, int32_t nshifts, double resolution, int64_t off1, int64_t off2, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2, int32_t h, int32_t w, int32_t* r0, int32_t* c0, int32_t* code_out, int64_t* n_roi_kept) {
    PUP_HOST_GUARD(pup_host_windows_impl(st1, st2, code, n, shift, sign, nshifts, resolution, off1, off2, lo1, hi1, lo2, hi2, h, w, r0, c0, code_out, n_roi_kept), PUP_ENOMEM);
}

PUP_EXPORT int pup_host_take_rows(int32_t ncols, const void* const* src, void* const* dst, const int32_t* esize, const int64_t* order, int64_t n, int64_t n_src) {
    PUP_HOST_GUARD(pup_host_take_rows_impl(ncols, src, dst, esize, order, n, n_src), PUP_ENOMEM);
}

PUP_EXPORT int64_t pup_host_factorize_ptr(const uintptr_t* ptrs, int64_t n, int32_t* codes, int64_t* first, int64_t max_uniq) {
    PUP_HOST_GUARD(pup_host_factorize_ptr_impl(ptrs, n, codes, first, max_uniq), -2);
}

PUP_EXPORT int pup_host_argsort(const uint64_t* keys, int64_t n, int32_t bits, int64_t* order) {
    PUP_HOST_GUARD(pup_host_argsort_impl(keys, n, bits, order), PUP_ENOMEM);
}

PUP_EXPORT int64_t pup_host_sort_pairs(const int64_t* s1, const int64_t* e1, const int64_t* s2, const int64_t* e2, const int32_t* c1, const int32_t* c2, int64_t n, const int64_t* rank, int32_t nu, double mindist, double maxdist, int64_t* rows, int64_t* s1o, int64_t* e1o, int64_t* s2o, int64_t* e2o, int32_t* c1o, int32_t* c2o, int32_t* flags) {
    PUP_HOST_GUARD(pup_host_sort_pairs_impl(s1, e1, s2, e2, c1, c2, n, rank, nu, mindist, maxdist, rows, s1o, e1o, s2o, e2o, c1o, c2o, flags), PUP_ENOMEM);
}

PUP_EXPORT int pup_host_mt_randint(uint32_t* key, int32_t* pos, int64_t low, int64_t high, int64_t m, int64_t scale, int64_t offset, void* out_any, int32_t out_bytes) {
    PUP_HOST_GUARD(pup_host_mt_randint_impl(key, pos, low, high, m, scale, offset, out_any, out_bytes), PUP_ENOMEM);
}

PUP_EXPORT int pup_host_mt_randint_plan(uint32_t* key, int32_t* pos, int32_t n_calls, const int64_t* low, const int64_t* high, const int64_t* m, const int64_t* scale, const int64_t* offset, void* const* out, const int32_t* out_bytes) {
    PUP_HOST_GUARD(pup_host_mt_randint_plan_impl(key, pos, n_calls, low, high, m, scale, offset, out, out_bytes), PUP_ENOMEM);
}

PUP_EXPORT int pup_host_lut_i32(const int32_t* lut, int64_t n_lut, const int32_t* codes, int64_t n, int64_t add_from, int32_t add, int32_t* out) {
    PUP_HOST_GUARD(pup_host_lut_i32_impl(lut, n_lut, codes, n, add_from, add, out), PUP_ENOMEM);
}

PUP_EXPORT int pup_host_count_le(const double* edges, int32_t n_edges, const double* values, int64_t n, int32_t* out) {
    PUP_HOST_GUARD(pup_host_count_le_impl(edges, n_edges, values, n, out), PUP_ENOMEM);
}

PUP_EXPORT int pup_host_normalise_tiles(double* sum, const int64_t* num, double* csum, const int64_t* cnum, int64_t count) {
    PUP_HOST_GUARD(pup_host_normalise_tiles_impl(sum, num, csum, cnum, count), PUP_ENOMEM);
}

PUP_EXPORT int pup_host_group_tiles_runs(int32_t n_parts, const int32_t* const* r0, const int32_t* const* c0, const int32_t* const* tile, const int64_t* split, const int32_t* tile_a, const int32_t* tile_b, const int64_t* len, int32_t T, int32_t* r0_out, int32_t* c0_out, int64_t* tile_ptr) {
    PUP_HOST_GUARD(pup_host_group_tiles_runs_impl(n_parts, r0, c0, tile, split, tile_a, tile_b, len, T, r0_out, c0_out, tile_ptr), PUP_ENOMEM);
}

PUP_EXPORT int64_t pup_host_control_windows(const int32_t* st1, const int32_t* st2, const int32_t* code, int64_t n, const int32_t* shift, const int32_t* sign, int32_t nshifts, double resolution, int64_t off1, int64_t off2, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2, int32_t h, int32_t w, int32_t* r0, int32_t* c0, int32_t* code_out) {
    PUP_HOST_GUARD(pup_host_control_windows_impl(st1, st2, code, n, shift, sign, nshifts, resolution, off1, off2, lo1, hi1, lo2, hi2, h, w, r0, c0, code_out), PUP_ENOMEM);
}

PUP_EXPORT int pup_host_pair_region_counts(const int64_t* s1, const int64_t* e1, const int64_t* s2, const int64_t* e2, const int32_t* c1, const int32_t* c2, int64_t n, double mindist, double maxdist, const int32_t* reg_code, const int64_t* reg_start, const int64_t* reg_end, int32_t n_regions, int64_t* counts) {
    PUP_HOST_GUARD(pup_host_pair_region_counts_impl(s1, e1, s2, e2, c1, c2, n, mindist, maxdist, reg_code, reg_start, reg_end, n_regions, counts), PUP_ENOMEM);
}
