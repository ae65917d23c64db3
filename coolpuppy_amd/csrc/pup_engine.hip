// pup_engine.hip — C ABI (include/pup_hip.h) over the gfx950 pile-up kernels.
//
// Host-side responsibilities: device residency of the pixel table / bin vectors / expected,
// chunking of tile-grouped snippets, kernel launches on one HIP stream, deterministic two-level
// reduction of per-chunk partial tiles into the running accumulators, HIP-event timing.
// Replaces the data handling of PileUpper.get_data / _stream_snips / accumulate_stream
// (reference coolpuppy/coolpup.py:1024-1057, 1059-1191, 1236-1283) — see the header for the map.
#include "../../include/pup_hip.h"
#include "pup_kernels.hpp"
#include "pup_staged.hpp"
#include "pup_staged_launch.hpp"
#include "pup_wide.hpp"
#include "pup_bin.hpp"

#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error = "";

template <typename T>
struct DevBuf {           // grow-only device buffer
    T* p = nullptr;
    size_t cap = 0;       // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = n + n / 4 + 16;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e != hipSuccess) { p = nullptr; cap = 0; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct pup_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // the plain kernel of a mixed (staged + plain) pile-up runs beside the staged one
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string err;
    // resident tables
    DevBuf<long long> indptr;
    DevBuf<int2> px;
    DevBuf<int> cnt32;
    DevBuf<double> cntf;                     // pixel values of a float pixels/count column (pup_load_pixel_values); float_values says whether in use
    bool float_values = false;
    DevBuf<double> bal;
    DevBuf<unsigned long long> badbits;
    DevBuf<unsigned long long> nf_keys;      // pixels with a non-finite balanced value (see collect_nonfinite_kernel), sorted
    DevBuf<int> xt_ids;                      // pup_pack_tiles / pup_unpack_tiles / pup_allgather_tiles: tile numbers on the device
    DevBuf<double> xt_f64;                   // pup_allgather_tiles: the ranks' tile blocks (f64 / i64 parts)
    DevBuf<long long> xt_i64;
    long long nf_count = 0;
    DevBuf<long long> nf_tp;                 // tile_ptr | flip_from of the current call, for the fix pass
    bool have_bal = false;
    DevBuf<pup::IdxBlock> idx;
    DevBuf<pup::IdxChrom> idx_chrom;
    std::vector<pup::IdxChrom> h_chroms;    // host copy of idx_chrom
    DevBuf<unsigned short> bin_chrom;       // [nbins] chromosome (index into idx_chrom) of every bin
    DevBuf<int> d_brow;                     // [n_chrom] block rows before each chromosome (block-order prepass)
    DevBuf<unsigned> rowseg;               // [nbins][n_chrom+1] search bounds per (row, chromosome), see K1Args
    DevBuf<uint2> rowabs;                  // [n_chrom][nbins] the same as absolute positions, chromosome-major (sparse trans kernel)
    DevBuf<unsigned long long> tbits;      // presence filter of the table as overlapping 32-bit words [column block][row] (sparse trans kernel, K1Args::tbits), built on first use
    int tbits_state = 0;                   // 0: not tried for this table, 1: built, -1: does not fit
    int tbits_shift = 0;                   // columns per bit of the bitmap = 1 << tbits_shift (a coarser filter when the exact one does not fit)
    DevBuf<int> band;                       // dense band of counts near the diagonal (staged kernel), [nbins][band_w] + zeros
    int band_w = 0;                         // 0: no band table
    int n_chrom = 0;
    bool have_idx = false, have_rowseg = false;
    long long idx_bytes = 0;
    DevBuf<double> weight, cov, expv, exp_pair;
    DevBuf<pup::ExpRegion> exp_regions;
    int n_exp_regions = 0; bool have_exp_pair = false;
    bool have_px = false, have_weight = false, have_cov = false;
    long long nbins = 0, nnz = 0, nexp = 0;
    // accumulators (packed layout of the header)
    DevBuf<double> acc_f64;
    struct { long long* p = nullptr; } acc_i64;   // view: the integer accumulators follow the f64 ones in acc_f64's allocation
    int T = 0, pad = 0, W = 0;
    // workspaces
    DevBuf<int> d_r0, d_c0, d_h, d_w;
    // block-ordered copy of the snippets for the workgroup-staged kernel (K1q, pup_staged.hpp)
    DevBuf<unsigned long long> d_keys, d_keys2;
    DevBuf<unsigned> d_k32, d_k32b, d_cnt32, d_starts;
    DevBuf<unsigned short> d_win, d_win2;    // window-in-block values before / after the block sort
    DevBuf<pup::StagedBlock> d_blocks;
    DevBuf<int> d_wgfirst;
    DevBuf<long long> d_timing; int timing_G = 0;
    DevBuf<unsigned char> d_teams;
    DevBuf<long long> d_segend;
    DevBuf<unsigned char> d_sorttmp;
    // K1w (pup_wide.hpp): partial records of the wide-window staged kernel
    DevBuf<double> wrec_f64; DevBuf<unsigned> wrec_num, wrec_seg;
    // hand-written block binning (pup_bin.hpp): digit counts / bucket bases / ticket / per-bucket block counts, tile descriptors of the
    // partition pass, the buckets' block lists, the packed block keys, the sorted low digits (sets of tile pairs)
    DevBuf<unsigned> d_binmeta, d_bindesc, d_blkkey, d_bkeys; DevBuf<unsigned short> d_low;
    DevBuf<double> rs_scratch;               // K5: one slab of window cells per workgroup (see pileup_rescale_kernel)
    DevBuf<double> rs_tile;                  // K5: output tiles too large for LDS, one per workgroup
    DevBuf<double> cov_rec; DevBuf<unsigned> cov_owner;     // coverage-vector pass beside the staged kernels (cov_vectors_kernel)
    long long wide_min = 20000;              // calls of at least this many wide cis windows take the staged wide kernel
    const char* last_kernel = "";            // which pile-up kernel the last pup_accumulate ran (diagnostics)
    const char* last_prepass = "";           // block order of the last staged call: "binning" (pup_bin.hpp) or "library_sort" (rocPRIM); "" when none ran
    unsigned warned = 0;                     // bit per reason: a large call left the staged kernels (said once per context, on stderr)
    void off_staged(unsigned bit, const char* why, long long n) {
        if ((warned & bit) || getenv("COOLPUPPY_AMD_QUIET")) return;
        warned |= bit;
        fprintf(stderr, "[coolpuppy_amd] a pile-up of %lld windows runs on the per-window kernels (several times slower than the staged ones): %s\n", n, why);
    }
    // the staged kernel takes calls from this many windows on (measured against the per-window kernel on 21-bin windows, round 3:
    // plain 3.3e5 windows 0.49 / 0.44 ms, 6.6e5 0.53 / 0.70; observed over expected 1.1e5 0.45 / 0.50, 3.3e5 0.61 / 1.24)
    long long tiled_min = 400000, tiled_min_ooe = 150000;
    // the key kernel's verdict (ineligible windows, windows a diagonal mask reaches) reaches the host through mapped
    // page-locked memory while the sort is already running; the block count of the last staged call comes back the same way
    volatile unsigned* h_flags = nullptr;    // [0] ineligible [1] unclear [2] outside the band [3] ticket   [4] blocks of the last staged call [5] its ticket
    unsigned* d_flags = nullptr;             // device address of h_flags
    hipEvent_t ev_key = nullptr;
    unsigned ticket = 0;
    // density of the last call with this signature: decides staged / per-window without a host round trip (see staged_run)
    std::vector<long long> hint_sig;
    long long hint_blocks = -1;              // -1: unknown
    unsigned hint_ticket = 0;                // ticket whose block count h_flags[4] will hold
    bool counts_added = false;               // this call's reduction kernel has added the windows-per-tile counts
    unsigned long long last_stagings = 0;   // diagnostics: regions staged by the last K1q launch (0: it did not run)
    bool last_staged = false;
    // launch geometry: ONE device blob (one H2D copy per new geometry), the typed views below point into it
    DevBuf<unsigned char> d_geom;
    struct GeomView {
        const long long *chunk_begin = nullptr, *chunk_end = nullptr, *seg1 = nullptr, *seg2 = nullptr, *dn = nullptr, *run_ptr = nullptr;
        const int *chunk_stride = nullptr, *block_chunk = nullptr, *block_band = nullptr, *block_chunk_t = nullptr;
        const unsigned char* chunk_flip = nullptr;
    } gv;
    DevBuf<double> part_f64, slice_f64;
    DevBuf<unsigned> part_num;
    DevBuf<long long> slice_num;
    DevBuf<unsigned long long> counters;   // [2]
    DevBuf<int> d_err;
    // stats / timing
    std::vector<int> brow_sent;              // what d_brow / d_segend hold (plan_block_order)
    std::vector<long long> htab_sent;
    std::vector<unsigned char> teams_sent;
    // by-diagonal expected as the host handed it over (a copy, with the regions' {offset, length}): when every diagonal a
    // window reaches has a usable expected (neither NaN nor 0), dividing by expected leaves cell validity a matter of row /
    // column masks only (factorised counts, see staged_run).  exp_far(igd) = the first unusable diagonal at or after igd
    // over all regions (the vectors' ends count as unusable), cached per igd
    std::vector<double> h_exp;
    std::vector<std::pair<long long, long long>> h_exp_reg;
    std::vector<std::pair<int, int>> h_exp_bounds;        // [start, end) global bins of the expected regions
    long long exp_far_igd = -1, exp_far_val = 0;
    bool hint_have_verdict = false, hint_fact = false, hint_band = false;   // the last staged call's kernel choice (same hint_sig)
    long long exp_far(long long igd) {
        if (exp_far_igd == igd) return exp_far_val;
        long long far = 0x7fffffffffffffffLL;
        for (const auto& g : h_exp_reg) {
            long long d = std::min(igd, g.second);
            for (; d < g.second; ++d) { const double e = h_exp[(size_t)(g.first + d)]; if (!(e == e) || e == 0.0) break; }
            far = std::min(far, d);
        }
        exp_far_igd = igd; exp_far_val = h_exp_reg.empty() ? 0 : far;
        return exp_far_val;
    }
    // the remembered verdict belongs to (table, index, expected, tuning): whatever replaces one of them forgets it — a
    // speculative launch on a stale verdict could stage from a band table that no longer exists
    bool no_rel_bc = false;                  // a call of this table had windows beyond the band: block columns keep the plain numbering
    void forget_hints() { hint_sig.clear(); hint_blocks = -1; hint_have_verdict = false; hint_sparse_calls = 0; no_rel_bc = false; }
    int hint_sparse_calls = 0;               // calls answered "too sparse" from memory since the block count was last measured
    bool profiling = false;      // HIP events around the kernels
    bool count_pixels = false;   // kernels also count the pixels inside the windows (statistics; costs a little)
    pup_stats stats{};
    struct EvTriple { hipEvent_t a, b, c, p; };   // K1 = a..b, reduction = b..c, block-order prepass = p..a (p may be null)
    std::vector<EvTriple> pending;             // awaiting a stream sync
    hipEvent_t slots[8] = {};
    int chunk_snippets = 0, variant = 0, group_waves = 0, debug_phases = 0;
    std::vector<long long> geom_key;            // launch-geometry cache (see pup_accumulate)
    long long g_nchunks = 0, g_nblocks = 0, g_nblocks_t = 0, g_nslices = 0, g_max_per_tile = 0;
    bool g_two_level = false;
    int max_lds = 0, n_cu = 0;
    float last_coverage_ms = 0.f;
};

namespace {

int fail(pup_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(ctx, expr)                                                                  \
    do { hipError_t e_ = (expr);                                                           \
         if (e_ != hipSuccess)                                                             \
             return fail((ctx), e_ == hipErrorOutOfMemory ? PUP_ENOMEM : PUP_EHIP,         \
                         "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

int bind(pup_ctx* c) {
    HIPCHK(c, hipSetDevice(c->device));
    return PUP_OK;
}

// drain finished event triples into the stats (after a stream sync)
void collect_events(pup_ctx* c) {
    for (auto& p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { c->stats.k1_ms += ms; c->stats.k1_launches += 1; }
        if (hipEventElapsedTime(&ms, p.b, p.c) == hipSuccess) c->stats.reduce_ms += ms;
        if (p.p && hipEventElapsedTime(&ms, p.p, p.a) == hipSuccess) c->stats.prepare_ms += ms;
        (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); (void)hipEventDestroy(p.c);
        if (p.p) (void)hipEventDestroy(p.p);
    }
    c->pending.clear();
}

int check_async_error(pup_ctx* c) {
    if (!c->d_err.p) return PUP_OK;
    int h = 0;
    HIPCHK(c, hipMemcpy(&h, c->d_err.p, sizeof(int), hipMemcpyDeviceToHost));
    if (h) {
        int zero = 0;
        HIPCHK(c, hipMemcpy(c->d_err.p, &zero, sizeof(int), hipMemcpyHostToDevice));
        if (h == 2) return fail(c, PUP_EHIP, "the staged kernel gave up waiting for one of its own waves (progressive staging): results of this call are void — please report; it only runs with tuning bit 22");
        return fail(c, PUP_ERANGE, "a snippet window leaves the bin table [0, %lld)", c->nbins);
    }
    return PUP_OK;
}

template <int WT>
void launch_k1(const pup::K1Args& a, int nchunks, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL(pup::pileup_chunk_kernel<WT>, dim3(nchunks), dim3(pup::kWave), lds, s, a);
}

template <int W>
void launch_k1r(const pup::K1Args& a, int nchunks, hipStream_t s) {
    if (a.mode & PUP_MODE_OOE)
        hipLaunchKernelGGL((pup::pileup_regtile_kernel<W, true>), dim3(nchunks), dim3(pup::kWave), 0, s, a);
    else
        hipLaunchKernelGGL((pup::pileup_regtile_kernel<W, false>), dim3(nchunks), dim3(pup::kWave), 0, s, a);
}

bool tiled_supported(int W) { return W >= 3 && W <= 31 && (W & 1); }

// workgroup-staged kernel (K1q, pup_staged.hpp): persistent workgroups over the device-built block table.  The region
// geometry is a property of the instantiation (StagedGeom) and only depends on facts the host knows BEFORE the prepass
// (window width, observed-over-expected, coverage / statistics riding along) — the block size of the prepass follows from it.
// The instantiations themselves live in their own translation units (pup_staged_tu.hip, compiled once per group of window
// widths, in parallel): pup::launch_staged picks the one for a call.
struct StagedGeo { int RSR, RSC, NW; };
StagedGeo staged_geometry(int W, bool ooe, bool extra, bool small21) {
    const bool two_buffers = small21 && W == 21 && !ooe && !extra;     // tuning bit 7: two 64-row buffers, sixteen waves (pup_staged.hpp: DB)
    const bool big = W <= 21 && !extra && !two_buffers;
    return StagedGeo{big ? 128 : (two_buffers ? pup::kSmallRows : 64), 128, (big || two_buffers) ? 16 : 8};
}

// banded register-tile kernel: NCH column chunks of 16 cells -> windows up to 16*NCH wide
int band_nch(int W) { return W <= 64 ? 4 : (W <= 128 ? 8 : 16); }

template <int NCH>
void launch_k1b(const pup::K1Args& a, int nblocks, hipStream_t s) {
    if (a.mode & PUP_MODE_OOE)
        hipLaunchKernelGGL((pup::pileup_band_kernel<NCH, true>), dim3(nblocks), dim3(pup::kWave), 0, s, a);
    else
        hipLaunchKernelGGL((pup::pileup_band_kernel<NCH, false>), dim3(nblocks), dim3(pup::kWave), 0, s, a);
}

// window widths the register-tile kernel is instantiated for (pad 1..15 -> W = 3..31)
bool launch_regtile(int W, const pup::K1Args& a, int nchunks, hipStream_t s) {
    switch (W) {
        case 3:  launch_k1r<3>(a, nchunks, s);  return true;
        case 5:  launch_k1r<5>(a, nchunks, s);  return true;
        case 7:  launch_k1r<7>(a, nchunks, s);  return true;
        case 9:  launch_k1r<9>(a, nchunks, s);  return true;
        case 11: launch_k1r<11>(a, nchunks, s); return true;
        case 13: launch_k1r<13>(a, nchunks, s); return true;
        case 15: launch_k1r<15>(a, nchunks, s); return true;
        case 17: launch_k1r<17>(a, nchunks, s); return true;
        case 19: launch_k1r<19>(a, nchunks, s); return true;
        case 21: launch_k1r<21>(a, nchunks, s); return true;
        case 23: launch_k1r<23>(a, nchunks, s); return true;
        case 25: launch_k1r<25>(a, nchunks, s); return true;
        case 27: launch_k1r<27>(a, nchunks, s); return true;
        case 29: launch_k1r<29>(a, nchunks, s); return true;
        case 31: launch_k1r<31>(a, nchunks, s); return true;
        default: return false;
    }
}

}  // namespace

// the library is built with -fvisibility=hidden: only the C ABI of include/pup_hip.h is exported
#pragma GCC visibility push(default)
extern "C" {

int pup_version(void) { return 100; }

int pup_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return PUP_EHIP; }
    return n;
}

const char* pup_last_error(const pup_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int pup_create(int device_id, pup_ctx** out) {
    if (!out) return fail(nullptr, PUP_EINVAL, "pup_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, PUP_EHIP, "pup_create: no HIP device available (%s)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, PUP_EINVAL, "pup_create: device_id %d out of range [0,%d)", device_id, n);
    pup_ctx* c = new (std::nothrow) pup_ctx();
    if (!c) return fail(nullptr, PUP_ENOMEM, "pup_create: out of host memory");
    c->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        delete c;
        return fail(nullptr, PUP_EHIP, "pup_create: %s", hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        c->max_lds = (int)prop.sharedMemPerBlock;
        c->n_cu = prop.multiProcessorCount;
    } else { c->max_lds = 64 * 1024; c->n_cu = 256; }
    for (auto& s : c->slots) (void)hipEventCreate(&s);
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) c->stream2 = nullptr;
    (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming); (void)hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (c->counters.reserve(2) != hipSuccess || c->d_err.reserve(1) != hipSuccess) {
        pup_destroy(c);
        return fail(nullptr, PUP_ENOMEM, "pup_create: device allocation failed");
    }
    {   // mapped page-locked flags: what the staged path's prepass tells the host without a blocking copy (staged_run)
        void* hp = nullptr; void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            std::memset(hp, 0, 64);
            c->h_flags = static_cast<volatile unsigned*>(hp); c->d_flags = static_cast<unsigned*>(dp);
            if (hipEventCreateWithFlags(&c->ev_key, hipEventDisableTiming) != hipSuccess) c->ev_key = nullptr;
        } else { if (hp) (void)hipHostFree(hp); (void)hipGetLastError(); }
    }
    (void)hipMemset(c->counters.p, 0, 2 * sizeof(unsigned long long));
    (void)hipMemset(c->d_err.p, 0, sizeof(int));
    *out = c;
    return PUP_OK;
}

void pup_destroy(pup_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    collect_events(c);
    c->indptr.release(); c->px.release(); c->cnt32.release(); c->bal.release(); c->badbits.release(); c->idx.release(); c->idx_chrom.release(); c->weight.release(); c->cov.release(); c->expv.release(); c->exp_pair.release(); c->exp_regions.release();
    c->bin_chrom.release(); c->d_brow.release(); c->brow_sent.clear(); c->band.release(); c->rowseg.release(); c->rowabs.release(); c->tbits.release(); c->tbits_state = 0;
    c->acc_f64.release(); c->acc_i64.p = nullptr;
    c->d_r0.release(); c->d_c0.release(); c->d_h.release(); c->d_w.release(); c->d_geom.release();
    c->d_keys.release(); c->d_keys2.release(); c->d_cnt32.release(); c->d_win.release(); c->d_win2.release();
    c->d_starts.release(); c->d_blocks.release();
    c->d_wgfirst.release(); c->d_segend.release(); c->htab_sent.clear(); c->d_sorttmp.release();
    c->wrec_f64.release(); c->wrec_num.release(); c->wrec_seg.release(); c->cov_rec.release(); c->cov_owner.release(); c->rs_scratch.release();
    c->d_binmeta.release(); c->d_bindesc.release(); c->d_blkkey.release(); c->d_bkeys.release(); c->d_low.release();
    if (c->ev_key) (void)hipEventDestroy(c->ev_key);
    if (c->h_flags) (void)hipHostFree(const_cast<unsigned*>(c->h_flags));
    c->d_k32.release(); c->d_k32b.release();
    c->part_f64.release(); c->slice_f64.release(); c->part_num.release(); c->slice_num.release();
    c->counters.release(); c->d_err.release();
    for (auto& s : c->slots) if (s) (void)hipEventDestroy(s);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pup_load_pixels(pup_ctx* c, const int64_t* bin1_offset, const void* bin2_id, int bin2_bytes,
                    const int32_t* count, int64_t nbins, int64_t nnz) {
    if (!c) return PUP_EINVAL;
    if (!bin1_offset || nbins <= 0 || nnz < 0 || (nnz > 0 && (!bin2_id || !count)))
        return fail(c, PUP_EINVAL, "pup_load_pixels: NULL table or bad sizes (nbins=%lld nnz=%lld)",
                    (long long)nbins, (long long)nnz);
    if (bin2_bytes != 4 && bin2_bytes != 8)
        return fail(c, PUP_EINVAL, "pup_load_pixels: bin2_bytes must be 4 or 8, got %d", bin2_bytes);
    if (nbins > 0x7fffffffLL - 4096)
        return fail(c, PUP_ENOTSUP, "pup_load_pixels: nbins %lld does not fit int32 bin ids", (long long)nbins);
    if (bin1_offset[0] != 0 || bin1_offset[nbins] != nnz)
        return fail(c, PUP_EINVAL, "pup_load_pixels: bin1_offset[0]=%lld, bin1_offset[nbins]=%lld, expected 0 and nnz=%lld",
                    (long long)bin1_offset[0], (long long)bin1_offset[nbins], (long long)nnz);
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_px = false; c->have_idx = false; c->forget_hints();
    c->tbits_state = 0;                                  // (the bitmap of the previous table: rebuilt on first use; its memory is kept)
    HIPCHK(c, c->indptr.reserve((size_t)nbins + 1));
    HIPCHK(c, c->px.reserve((size_t)nnz + 64));         // +64: K3 reads pixel pairs (16-byte loads) across a row's end
    HIPCHK(c, c->cnt32.reserve((size_t)nnz + 64));     // +64: the register-tile kernel loads counts unconditionally
    HIPCHK(c, hipMemset(c->cnt32.p + nnz, 0, 64 * sizeof(int)));
    HIPCHK(c, hipMemcpy(c->indptr.p, bin1_offset, ((size_t)nbins + 1) * sizeof(long long), hipMemcpyHostToDevice));
    // stage bin2/count through a bounded device staging area and interleave on device
    const size_t slab = (size_t)1 << 26;   // 64 Mi pixels per slab
    void* d_col = nullptr; int* d_cnt = nullptr;
    const size_t slab_n = (size_t)std::min<int64_t>(nnz, (int64_t)slab);
    if (nnz > 0) {
        HIPCHK(c, hipMalloc(&d_col, slab_n * (size_t)bin2_bytes));
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_cnt), slab_n * sizeof(int));
        if (e != hipSuccess) { (void)hipFree(d_col); return fail(c, PUP_ENOMEM, "pup_load_pixels: staging alloc failed"); }
    }
    int status = PUP_OK;
    for (int64_t off = 0; off < nnz && status == PUP_OK; off += (int64_t)slab) {
        const size_t m = (size_t)std::min<int64_t>((int64_t)slab, nnz - off);
        hipError_t e = hipMemcpy(d_col, static_cast<const char*>(bin2_id) + (size_t)off * bin2_bytes,
                                 m * (size_t)bin2_bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_cnt, count + off, m * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            const int blocks = (int)std::min<size_t>((m + 255) / 256, 65536);
            if (bin2_bytes == 8)
                hipLaunchKernelGGL(pup::pack_pixels_kernel<long long>, dim3(blocks), dim3(256), 0, c->stream,
                                   static_cast<const long long*>(d_col), d_cnt, c->px.p + off, c->cnt32.p + off, (long long)m);
            else
                hipLaunchKernelGGL(pup::pack_pixels_kernel<int>, dim3(blocks), dim3(256), 0, c->stream,
                                   static_cast<const int*>(d_col), d_cnt, c->px.p + off, c->cnt32.p + off, (long long)m);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        if (e != hipSuccess) status = fail(c, PUP_EHIP, "pup_load_pixels: upload failed: %s", hipGetErrorString(e));
    }
    if (d_col) (void)hipFree(d_col);
    if (d_cnt) (void)hipFree(d_cnt);
    if (status != PUP_OK) return status;
    c->nbins = nbins; c->nnz = nnz; c->have_px = true;
    c->have_weight = c->have_cov = false; c->have_bal = false;
    c->float_values = false; c->band_w = 0;              // (the band of the previous table, or none after float values: pup_build_index makes this table's)
    return PUP_OK;
}

// Float pixel values: cooler allows pixels/count to be a float column (merged / scaled / simulated maps) and the reference multiplies
// whatever it finds (coolpuppy/coolpup.py:1053-1057).  The engine's own tables are integer counts — the dense band, the count
// arrays the staged kernels balance at store time, the exact integer sums of the coverage kernel — so such a table is served by the
// kernels that read the balanced VALUE table (`bal`, float64, built from these values by pup_load_bins): the per-window register
// tile (K1r), the banded tile (K1b), the sparse trans kernel (K1s), rescaled windows, stripes and per-snippet windows.  Same
// results as for counts; the staged kernels and pup_coverage decline (the latter with PUP_ENOTSUP).
int pup_load_pixel_values(pup_ctx* c, const double* value, int64_t nnz) {
    if (!c) return PUP_EINVAL;
    if (!c->have_px) return fail(c, PUP_ESTATE, "pup_load_pixel_values: call pup_load_pixels first");
    if (nnz != c->nnz || (nnz > 0 && !value)) return fail(c, PUP_EINVAL, "pup_load_pixel_values: %lld values for a table of %lld pixels", (long long)nnz, (long long)c->nnz);
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, c->cntf.reserve((size_t)std::max<int64_t>(nnz, 1)));
    if (nnz > 0) HIPCHK(c, hipMemcpy(c->cntf.p, value, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice));
    c->float_values = true;
    c->band_w = 0;                                       // (a band built from the placeholder counts must not be staged from)
    c->have_bal = false; c->have_weight = c->have_cov = false;       // pup_load_bins rebuilds the value table
    c->forget_hints();
    return PUP_OK;
}


int pup_load_pixels_stream(pup_ctx* c, const int64_t* bin1_offset, int64_t nbins, int64_t nnz, int bin2_bytes,
                           int64_t slab_pixels, pup_fill_fn fill, void* user, double* h2d_ms, int64_t* h2d_bytes) {
    if (!c) return PUP_EINVAL;
    if (!bin1_offset || nbins <= 0 || nnz < 0 || !fill)
        return fail(c, PUP_EINVAL, "pup_load_pixels_stream: NULL table / reader or bad sizes (nbins=%lld nnz=%lld)", (long long)nbins, (long long)nnz);
    if (bin2_bytes != 4 && bin2_bytes != 8) return fail(c, PUP_EINVAL, "pup_load_pixels_stream: bin2_bytes must be 4 or 8, got %d", bin2_bytes);
    if (nbins > 0x7fffffffLL - 4096) return fail(c, PUP_ENOTSUP, "pup_load_pixels_stream: nbins %lld does not fit int32 bin ids", (long long)nbins);
    if (bin1_offset[0] != 0 || bin1_offset[nbins] != nnz)
        return fail(c, PUP_EINVAL, "pup_load_pixels_stream: bin1_offset[0]=%lld, bin1_offset[nbins]=%lld, expected 0 and nnz=%lld",
                    (long long)bin1_offset[0], (long long)bin1_offset[nbins], (long long)nnz);
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_px = false; c->have_idx = false; c->forget_hints();
    c->tbits_state = 0;
    HIPCHK(c, c->indptr.reserve((size_t)nbins + 1));
    HIPCHK(c, c->px.reserve((size_t)nnz + 64));
    HIPCHK(c, c->cnt32.reserve((size_t)nnz + 64));
    HIPCHK(c, hipMemset(c->cnt32.p + nnz, 0, 64 * sizeof(int)));
    HIPCHK(c, hipMemcpy(c->indptr.p, bin1_offset, ((size_t)nbins + 1) * sizeof(long long), hipMemcpyHostToDevice));
    const size_t slab = (size_t)std::max<int64_t>(1, std::min<int64_t>(slab_pixels > 0 ? slab_pixels : ((int64_t)1 << 25), std::max<int64_t>(nnz, 1)));
    hipStream_t cs = c->stream2 ? c->stream2 : c->stream;                 // the copy stream
    // two page-locked slabs {bin2_id | count} and two device staging slabs: slab k % 2 is refilled once the pack kernel of slab
    // k - 2 is done with it; the reader fills one while the other is in flight
    char* h_slab[2] = {nullptr, nullptr}; char* d_slab[2] = {nullptr, nullptr};
    hipEvent_t ev_done[2] = {nullptr, nullptr}, ev_t0[2] = {nullptr, nullptr}, ev_t1[2] = {nullptr, nullptr};
    const size_t slab_bytes = slab * ((size_t)bin2_bytes + sizeof(int));
    int status = PUP_OK;
    double ms_total = 0.0; long long bytes_total = 0;
    auto cleanup = [&]() {
        for (int s = 0; s < 2; ++s) {
            if (h_slab[s]) (void)hipHostFree(h_slab[s]);
            if (d_slab[s]) (void)hipFree(d_slab[s]);
            if (ev_done[s]) (void)hipEventDestroy(ev_done[s]);
            if (ev_t0[s]) (void)hipEventDestroy(ev_t0[s]);
            if (ev_t1[s]) (void)hipEventDestroy(ev_t1[s]);
        }
    };
    for (int s = 0; s < 2 && status == PUP_OK && nnz > 0; ++s) {
        if (hipHostMalloc(reinterpret_cast<void**>(&h_slab[s]), slab_bytes, hipHostMallocDefault) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d_slab[s]), slab_bytes) != hipSuccess ||
            hipEventCreate(&ev_done[s]) != hipSuccess || hipEventCreate(&ev_t0[s]) != hipSuccess || hipEventCreate(&ev_t1[s]) != hipSuccess)
            status = fail(c, PUP_ENOMEM, "pup_load_pixels_stream: staging allocation failed (%zu bytes per slab)", slab_bytes);
    }
    bool used[2] = {false, false};
    auto harvest = [&](int s) {                           // copy time of the slab's previous trip (its events are complete)
        float ms = 0.f;
        if (used[s] && hipEventElapsedTime(&ms, ev_t0[s], ev_t1[s]) == hipSuccess) ms_total += ms;
    };
    int k = 0;
    for (int64_t off = 0; off < nnz && status == PUP_OK; off += (int64_t)slab, ++k) {
        const int s = k & 1;
        const size_t m = (size_t)std::min<int64_t>((int64_t)slab, nnz - off);
        if (used[s]) {
            hipError_t e = hipEventSynchronize(ev_done[s]);
            if (e != hipSuccess) { status = fail(c, PUP_EHIP, "pup_load_pixels_stream: %s", hipGetErrorString(e)); break; }
            harvest(s);
        }
        void* hcol = h_slab[s]; int* hcnt = reinterpret_cast<int*>(h_slab[s] + slab * (size_t)bin2_bytes);
        if (fill(user, off, (int64_t)m, hcol, hcnt) != 0) { status = fail(c, PUP_EINVAL, "pup_load_pixels_stream: the reader failed at pixel %lld", (long long)off); break; }
        void* dcol = d_slab[s]; int* dcnt = reinterpret_cast<int*>(d_slab[s] + slab * (size_t)bin2_bytes);
        hipError_t e = hipEventRecord(ev_t0[s], cs);
        if (e == hipSuccess) e = hipMemcpyAsync(dcol, hcol, m * (size_t)bin2_bytes, hipMemcpyHostToDevice, cs);
        if (e == hipSuccess) e = hipMemcpyAsync(dcnt, hcnt, m * sizeof(int), hipMemcpyHostToDevice, cs);
        if (e == hipSuccess) e = hipEventRecord(ev_t1[s], cs);
        if (e == hipSuccess) {
            const int blocks = (int)std::min<size_t>((m + 255) / 256, 65536);
            if (bin2_bytes == 8)
                hipLaunchKernelGGL(pup::pack_pixels_kernel<long long>, dim3(blocks), dim3(256), 0, cs,
                                   static_cast<const long long*>(dcol), dcnt, c->px.p + off, c->cnt32.p + off, (long long)m);
            else
                hipLaunchKernelGGL(pup::pack_pixels_kernel<int>, dim3(blocks), dim3(256), 0, cs,
                                   static_cast<const int*>(dcol), dcnt, c->px.p + off, c->cnt32.p + off, (long long)m);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(ev_done[s], cs);
        if (e != hipSuccess) { status = fail(c, PUP_EHIP, "pup_load_pixels_stream: upload failed: %s", hipGetErrorString(e)); break; }
        used[s] = true;
        bytes_total += (long long)(m * ((size_t)bin2_bytes + sizeof(int)));
    }
    {
        hipError_t e = hipStreamSynchronize(cs);
        if (e != hipSuccess && status == PUP_OK) status = fail(c, PUP_EHIP, "pup_load_pixels_stream: %s", hipGetErrorString(e));
        for (int s = 0; s < 2; ++s) harvest(s);
    }
    cleanup();
    if (status != PUP_OK) return status;
    if (h2d_ms) *h2d_ms = ms_total;
    if (h2d_bytes) *h2d_bytes = bytes_total;
    c->nbins = nbins; c->nnz = nnz; c->have_px = true;
    c->have_weight = c->have_cov = false; c->have_bal = false;
    c->float_values = false; c->band_w = 0;              // (as pup_load_pixels: a new table is a table of counts until pup_load_pixel_values says otherwise)
    c->forget_hints();
    return PUP_OK;
}

int pup_build_index(pup_ctx* c, const int64_t* chrom_offset, int32_t n_chroms, int64_t max_bytes) {
    if (!c) return PUP_EINVAL;
    if (!c->have_px) return fail(c, PUP_ESTATE, "pup_build_index: call pup_load_pixels first");
    if (!chrom_offset || n_chroms <= 0) return fail(c, PUP_EINVAL, "pup_build_index: NULL chrom_offset or n_chroms <= 0");
    if (chrom_offset[0] != 0 || chrom_offset[n_chroms] != c->nbins)
        return fail(c, PUP_EINVAL, "pup_build_index: chrom_offset must run from 0 to nbins=%lld", c->nbins);
    std::vector<pup::IdxChrom> tab((size_t)n_chroms);
    long long nblocks = 0;
    for (int k = 0; k < n_chroms; ++k) {
        const long long lo = chrom_offset[k], hi = chrom_offset[k + 1];
        if (hi < lo) return fail(c, PUP_EINVAL, "pup_build_index: chrom_offset decreases at %d", k);
        const long long nb = hi - lo;
        tab[(size_t)k].start = (int)lo; tab[(size_t)k].end = (int)hi;
        tab[(size_t)k].nblk = (int)((nb + pup::kIdxCols - 1) / pup::kIdxCols);
        tab[(size_t)k].pad_ = 0;
        tab[(size_t)k].blk_base = nblocks;
        nblocks += nb * tab[(size_t)k].nblk;
    }
    const long long bytes = nblocks * (long long)sizeof(pup::IdxBlock);
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_idx = false; c->forget_hints();
    if (max_bytes > 0 && bytes > max_bytes)
        return fail(c, PUP_ENOMEM, "pup_build_index: index needs %lld bytes, limit is %lld", bytes, (long long)max_bytes);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)bytes + ((size_t)1 << 30) > free_b + c->idx.cap * sizeof(pup::IdxBlock))
        return fail(c, PUP_ENOMEM, "pup_build_index: index needs %lld bytes, only %zu free on the device", bytes, free_b);
    HIPCHK(c, c->idx.reserve((size_t)std::max<long long>(nblocks, 1)));
    HIPCHK(c, c->idx_chrom.reserve((size_t)n_chroms));
    HIPCHK(c, hipMemcpy(c->idx_chrom.p, tab.data(), tab.size() * sizeof(pup::IdxChrom), hipMemcpyHostToDevice));
    c->h_chroms = tab;
    c->bin_chrom.release();
    if (n_chroms <= 0xffff) {
        std::vector<unsigned short> bc((size_t)c->nbins);
        for (int k = 0; k < n_chroms; ++k) std::fill(bc.begin() + tab[(size_t)k].start, bc.begin() + tab[(size_t)k].end, (unsigned short)k);
        HIPCHK(c, c->bin_chrom.reserve((size_t)c->nbins));
        HIPCHK(c, hipMemcpy(c->bin_chrom.p, bc.data(), bc.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMemsetAsync(c->idx.p, 0, (size_t)nblocks * sizeof(pup::IdxBlock), c->stream));
    const int rows_per_block = 4;
    const unsigned g1 = (unsigned)std::min<long long>((c->nbins + rows_per_block - 1) / rows_per_block, 1 << 20);
    hipLaunchKernelGGL(pup::index_fill_kernel, dim3(g1), dim3(64 * rows_per_block), 0, c->stream,
                       c->indptr.p, c->px.p, c->idx_chrom.p, n_chroms, c->idx.p, c->nbins);
    hipLaunchKernelGGL(pup::index_rank_kernel, dim3((unsigned)((c->nbins + 255) / 256)), dim3(256), 0, c->stream,
                       c->indptr.p, c->idx_chrom.p, n_chroms, c->idx.p, c->nbins);
    // search bounds for windows the rank-bitmap index does not cover (inter-chromosomal): skipped when a row can
    // hold 2^32 pixels or the table would be out of proportion (many small contigs)
    c->have_rowseg = false;
    const long long seg_entries = c->nbins * (long long)(n_chroms + 1);
    if (n_chroms > 1 && c->nnz < 0xffffffffLL && seg_entries * 4 <= std::max<long long>(c->nnz, 1 << 20)) {
        HIPCHK(c, c->rowseg.reserve((size_t)seg_entries));
        hipLaunchKernelGGL(pup::rowseg_kernel, dim3((unsigned)((seg_entries + 255) / 256)), dim3(256), 0, c->stream,
                           c->indptr.p, c->px.p, c->idx_chrom.p, n_chroms, c->rowseg.p, c->nbins);
        c->have_rowseg = true;
        HIPCHK(c, c->rowabs.reserve((size_t)c->nbins * (size_t)n_chroms));
        hipLaunchKernelGGL(pup::rowabs_kernel, dim3((unsigned)((c->nbins * (long long)n_chroms + 255) / 256)), dim3(256), 0, c->stream,
                           c->indptr.p, c->px.p, c->idx_chrom.p, n_chroms, c->rowabs.p, c->nbins);
    }
    // dense band of counts for the staged kernel: band[row][j] = count(row, row + j), j < 1024 (10 Mb at 10 kb) — 4 KiB per
    // matrix row of the 288 GB; skipped when it would not fit 32-bit byte offsets or a quarter of the free memory
    c->band_w = 0;
    {
        // widest band of 1024 / 512 / 256 columns that fits (10 Mb at 10 kb; a 2 kb table still gets 1 Mb): a call whose windows
        // leave the band is staged through the index instead (the key kernel counts them)
        // pads: kBandFront cells before row 0 and 129 rows of zeros behind the last one — the staged kernel's factorised
        // variant reads whole region rows without masking (cells left of the diagonal, rows past the table)
        size_t fb = 0, tb = 0;
        const bool have_mem = hipMemGetInfo(&fb, &tb) == hipSuccess;
        long long widest = 1024;
        if (const char* e = getenv("COOLPUPPY_AMD_BAND_COLUMNS")) {          // tests: start from a narrower band
            const long long v = atoll(e);
            if (v == 256 || v == 512 || v == 1024 || v == 2048 || v == 4096) widest = v;
        }
        for (long long BWd = widest; BWd >= 256 && have_mem && !(c->variant & 256) && !c->float_values; BWd >>= 1) {
            const long long cells = pup::kBandFront + (c->nbins + 129) * BWd;
            if (!((size_t)cells * 4 <= fb / 4 + c->band.cap * sizeof(int))) continue;     // (64-bit addressed: no cell-count limit)
            HIPCHK(c, c->band.reserve((size_t)cells));
            HIPCHK(c, hipMemsetAsync(c->band.p, 0, (size_t)cells * sizeof(int), c->stream));
            const unsigned gb2 = (unsigned)std::min<long long>((c->nbins + 3) / 4, 1 << 20);
            hipLaunchKernelGGL(pup::band_fill_kernel, dim3(gb2), dim3(256), 0, c->stream, c->indptr.p, c->px.p, c->band.p + pup::kBandFront, (int)BWd, c->nbins);
            c->band_w = (int)BWd;
            break;
        }
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n_chrom = n_chroms; c->have_idx = true; c->idx_bytes = bytes;
    return PUP_OK;
}

int pup_coverage(pup_ctx* c, const int64_t* chrom_offset, int32_t n_chroms, int32_t ignore_diags,
                 double* cov_cis, double* cov_tot) {
    if (!c) return PUP_EINVAL;
    if (!c->have_px) return fail(c, PUP_ESTATE, "pup_coverage: call pup_load_pixels first");
    if (!chrom_offset || n_chroms <= 0) return fail(c, PUP_EINVAL, "pup_coverage: NULL chrom_offset or n_chroms <= 0");
    if (chrom_offset[0] != 0 || chrom_offset[n_chroms] != c->nbins)
        return fail(c, PUP_EINVAL, "pup_coverage: chrom_offset must run from 0 to nbins=%lld", c->nbins);
    if (ignore_diags < 0) return fail(c, PUP_EINVAL, "pup_coverage: ignore_diags must be >= 0");
    int rc = bind(c); if (rc) return rc;
    std::vector<pup::IdxChrom> tab((size_t)n_chroms);
    for (int k = 0; k < n_chroms; ++k) {
        if (chrom_offset[k + 1] < chrom_offset[k]) return fail(c, PUP_EINVAL, "pup_coverage: chrom_offset decreases at %d", k);
        tab[(size_t)k] = pup::IdxChrom{(int)chrom_offset[k], (int)chrom_offset[k + 1], 0, 0, 0};
    }
    // counts: u64 sums (exact, order-independent); float pixel values (pup_load_pixel_values): f64 sums of the same pass
    const bool flt = c->float_values;
    DevBuf<pup::IdxChrom> d_tab; DevBuf<unsigned long long> d_cov;          // (2 x nbins 8-byte accumulators either way)
    const size_t nb = (size_t)c->nbins;
    hipError_t e = d_tab.reserve((size_t)n_chroms);
    if (e == hipSuccess) e = d_cov.reserve(2 * nb);
    if (e != hipSuccess) { d_tab.release(); d_cov.release(); return fail(c, PUP_ENOMEM, "pup_coverage: device allocation failed"); }
    int status = PUP_OK;
    std::vector<unsigned long long> h(2 * nb);
    e = hipMemcpy(d_tab.p, tab.data(), tab.size() * sizeof(pup::IdxChrom), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(d_cov.p, 0, 2 * nb * sizeof(unsigned long long), c->stream);     // (all-zero bits are 0.0 too)
    if (e == hipSuccess) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, c->stream);
        const dim3 grid((unsigned)((c->nbins + pup::kCovRows - 1) / pup::kCovRows));
        if (flt)
            hipLaunchKernelGGL(pup::coverage_kernel<double>, grid, dim3(512), 0, c->stream, c->indptr.p, c->px.p, (const double*)c->cntf.p,
                               d_tab.p, n_chroms, ignore_diags, reinterpret_cast<double*>(d_cov.p), reinterpret_cast<double*>(d_cov.p + nb), c->nbins);
        else
            hipLaunchKernelGGL(pup::coverage_kernel<unsigned long long>, grid, dim3(512), 0, c->stream, c->indptr.p, c->px.p, (const double*)nullptr,
                               d_tab.p, n_chroms, ignore_diags, d_cov.p, d_cov.p + nb, c->nbins);
        (void)hipEventRecord(e1, c->stream);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        float ms = 0.f;
        if (e == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) c->last_coverage_ms = ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    if (e == hipSuccess) e = hipMemcpy(h.data(), d_cov.p, 2 * nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e != hipSuccess) status = fail(c, PUP_EHIP, "pup_coverage: %s", hipGetErrorString(e));
    d_tab.release(); d_cov.release();
    if (status != PUP_OK) return status;
    if (flt) {
        const double* hd = reinterpret_cast<const double*>(h.data());
        for (size_t i = 0; i < nb; ++i) {
            if (cov_cis) cov_cis[i] = hd[nb + i];
            if (cov_tot) cov_tot[i] = hd[nb + i] + hd[i];
        }
        return PUP_OK;
    }
    for (size_t i = 0; i < nb; ++i) {
        if (cov_cis) cov_cis[i] = (double)h[nb + i];
        if (cov_tot) cov_tot[i] = (double)(h[nb + i] + h[i]);      // intra- plus inter-chromosomal
    }
    return PUP_OK;
}

int pup_load_bins(pup_ctx* c, const double* weight, const double* cov) {
    if (!c) return PUP_EINVAL;
    if (!c->have_px) return fail(c, PUP_ESTATE, "pup_load_bins: call pup_load_pixels first");
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_weight = c->have_cov = false; c->have_bal = false; c->forget_hints();
    if (weight) {
        HIPCHK(c, c->weight.reserve((size_t)c->nbins));
        HIPCHK(c, hipMemcpy(c->weight.p, weight, (size_t)c->nbins * sizeof(double), hipMemcpyHostToDevice));
        c->have_weight = true;
    }
    if (cov) {
        HIPCHK(c, c->cov.reserve((size_t)c->nbins));
        HIPCHK(c, hipMemcpy(c->cov.p, cov, (size_t)c->nbins * sizeof(double), hipMemcpyHostToDevice));
        c->have_cov = true;
    }
    // per (table, weight column): balanced pixel values + masked-bin bitmap for the register-tile kernel
    const long long nwords = c->nbins / 64 + 3;
    HIPCHK(c, c->bal.reserve((size_t)c->nnz + 64));
    HIPCHK(c, c->badbits.reserve((size_t)nwords));
    HIPCHK(c, hipMemsetAsync(c->bal.p + c->nnz, 0, 64 * sizeof(double), c->stream));
    const double* dw = c->have_weight ? c->weight.p : nullptr;
    const unsigned gb = (unsigned)std::min<long long>((c->nbins + 3) / 4, 1 << 20);
    hipLaunchKernelGGL(pup::balance_pixels_kernel, dim3(gb), dim3(256), 0, c->stream,
                       c->indptr.p, c->px.p, dw, c->bal.p, c->nbins, c->float_values ? (const double*)c->cntf.p : (const double*)nullptr);
    hipLaunchKernelGGL(pup::badbits_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, c->stream,
                       dw, c->badbits.p, c->nbins, nwords);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // weights of +-inf (never produced by balancing, but legal values of the column): list the pixels they spoil
    c->nf_count = 0;
    bool any_inf = false;
    if (weight) for (long long i = 0; i < c->nbins && !any_inf; ++i) any_inf = std::isinf(weight[i]);
    if (any_inf) {
        DevBuf<unsigned long long> cnt;
        HIPCHK(c, cnt.reserve(1));
        unsigned long long total = 0;
        for (int pass = 0; pass < 2; ++pass) {
            HIPCHK(c, hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), c->stream));
            hipLaunchKernelGGL(pup::collect_nonfinite_kernel, dim3(gb), dim3(256), 0, c->stream, c->indptr.p, c->px.p, dw, c->nbins,
                               pass ? c->nf_keys.p : nullptr, total, cnt.p, c->float_values ? (const double*)c->cntf.p : (const double*)nullptr);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(&total, cnt.p, sizeof(total), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (total == 0) break;
            if (pass == 0) HIPCHK(c, c->nf_keys.reserve((size_t)total));
        }
        if (total > 0) {
            std::vector<unsigned long long> h((size_t)total);
            HIPCHK(c, hipMemcpy(h.data(), c->nf_keys.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            HIPCHK(c, hipMemcpy(c->nf_keys.p, h.data(), h.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
            c->nf_count = (long long)total;
        }
    }
    c->have_bal = true;
    return PUP_OK;
}

int pup_set_expected(pup_ctx* c, const double* expected, int64_t n) {
    if (!c) return PUP_EINVAL;
    if (n < 0 || (n > 0 && !expected)) return fail(c, PUP_EINVAL, "pup_set_expected: bad arguments");
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));   // earlier launches may still read the old vector
    c->forget_hints();
    c->nexp = 0; c->n_exp_regions = 0; c->have_exp_pair = false;
    c->h_exp.clear(); c->h_exp_reg.clear(); c->h_exp_bounds.clear(); c->exp_far_igd = -1;
    if (n > 0) {
        HIPCHK(c, c->expv.reserve((size_t)n));
        HIPCHK(c, hipMemcpy(c->expv.p, expected, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        c->nexp = n;
        c->h_exp.assign(expected, expected + n); c->h_exp_reg.emplace_back(0, n);
    }
    return PUP_OK;
}

int pup_set_expected_table(pup_ctx* c, const int32_t* start, const int32_t* end, const int64_t* offset,
                           const int64_t* length, int32_t n_regions, const double* values, int64_t n_values,
                           const double* pair) {
    if (!c) return PUP_EINVAL;
    if (n_regions <= 0 || !start || !end) return fail(c, PUP_EINVAL, "pup_set_expected_table: no regions");
    if (!pair && (!offset || !length || !values || n_values <= 0))
        return fail(c, PUP_EINVAL, "pup_set_expected_table: cis table needs offset/length/values");
    std::vector<pup::ExpRegion> tab((size_t)n_regions);
    for (int r = 0; r < n_regions; ++r) {
        if (end[r] < start[r] || (r > 0 && start[r] < end[r - 1]))
            return fail(c, PUP_EINVAL, "pup_set_expected_table: regions must be sorted by start and disjoint (region %d)", r);
        const long long off = pair ? 0 : offset[r], len = pair ? 0 : length[r];
        if (off < 0 || len < 0 || off + len > n_values)
            if (!pair) return fail(c, PUP_EINVAL, "pup_set_expected_table: vector of region %d leaves values[]", r);
        tab[(size_t)r] = pup::ExpRegion{start[r], end[r], off, len};
    }
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->forget_hints();
    c->nexp = 0; c->n_exp_regions = 0; c->have_exp_pair = false;
    c->h_exp.clear(); c->h_exp_reg.clear(); c->h_exp_bounds.clear(); c->exp_far_igd = -1;
    HIPCHK(c, c->exp_regions.reserve((size_t)n_regions));
    HIPCHK(c, hipMemcpy(c->exp_regions.p, tab.data(), tab.size() * sizeof(pup::ExpRegion), hipMemcpyHostToDevice));
    if (pair) {
        HIPCHK(c, c->exp_pair.reserve((size_t)n_regions * n_regions));
        HIPCHK(c, hipMemcpy(c->exp_pair.p, pair, (size_t)n_regions * n_regions * sizeof(double), hipMemcpyHostToDevice));
        c->have_exp_pair = true;
    } else {
        HIPCHK(c, c->expv.reserve((size_t)n_values));
        HIPCHK(c, hipMemcpy(c->expv.p, values, (size_t)n_values * sizeof(double), hipMemcpyHostToDevice));
        c->h_exp.assign(values, values + n_values);
        for (const pup::ExpRegion& g : tab) { c->h_exp_reg.emplace_back(g.off, g.len); c->h_exp_bounds.emplace_back(g.start, g.end); }
    }
    c->n_exp_regions = n_regions;
    return PUP_OK;
}

static hipError_t clear_async(pup_ctx* c, void* p, size_t bytes);      // (below: small clears by a plain kernel)
// zoom weights per output row / column the rescaling kernel can keep in LDS beside its S x S tile (0: per-sample zoom)
static int rescale_sep_k(const pup_ctx* c, size_t tile_lds, int S) {
    if ((c->variant & 2048) || S <= 0) return 0;
    const long long room = (long long)c->max_lds - (long long)tile_lds - 8 - 2LL * S * (long long)sizeof(int);
    const long long k = room / (2LL * S * (long long)sizeof(double));
    return k < 3 ? 0 : (int)std::min<long long>(k, pup::kRescaleSepK);
}
int pup_reset(pup_ctx* c, int32_t n_tiles, int32_t pad) {
    if (!c) return PUP_EINVAL;
    if (n_tiles <= 0 || pad < 0) return fail(c, PUP_EINVAL, "pup_reset: n_tiles=%d pad=%d", n_tiles, pad);
    const int W = 2 * pad + 1;
    if ((long long)W * W > 0x3fffffffLL) return fail(c, PUP_ENOTSUP, "pup_reset: window %dx%d does not fit 32-bit cell numbers", W, W);
    int rc = bind(c); if (rc) return rc;
    const size_t W2 = (size_t)W * W;
    const size_t nf = (size_t)n_tiles * (W2 + 2 * (size_t)W), ni = (size_t)n_tiles * (W2 + 1);
    // (no synchronisation unless the buffer must grow: the memset below is ordered behind whatever still uses the
    // accumulators on the stream, so back-to-back reset / accumulate loops keep the GPU's queue filled)
    if (nf + ni > c->acc_f64.cap) HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, c->acc_f64.reserve(nf + ni));             // one allocation, one memset: f64 [nf] | i64 [ni]
    c->acc_i64.p = reinterpret_cast<long long*>(c->acc_f64.p + nf);
    HIPCHK(c, clear_async(c, c->acc_f64.p, (nf + ni) * sizeof(double)));
    c->T = n_tiles; c->pad = pad; c->W = W;
    return PUP_OK;
}

// ---- the workgroup-staged path (K1q, pup_staged.hpp): block-order prepass + persistent kernel + reduction -----------
// Eligible calls: register-tile widths, plain / OOE modes, cis (ignore_diags >= 0), rank-bitmap index present and covering
// every window, at least `tiled_min` windows, and windows dense enough that a staged region serves many of them.
//   segment = the snippets that share a pass: (tile, flip) — or, when tiles are PAIRED (T even: tile t with tile
//   t + T/2, the ROI and the control tile of one group in the host layer's numbering), (pair, flip) with the tile's
//   half as a per-window slot bit, so that a sparse tile rides on the regions its dense partner stages anyway.
// Device pipeline, NO host round trip on its critical path:
//   keys (segment | expected region | block row | block col) + window-in-block values  -> publish the key kernel's verdict to
//   mapped host memory, record an event -> radix sort (rocPRIM) -> block starts (own ordered compaction) -> block table +
//   workgroup ranges (grid-stride over a block count only the device knows) -> [the host waits for the EVENT — the key
//   kernel is long done, the sort still runs — and picks the kernel: any ineligible window sends the call to the per-window
//   kernels, any window a diagonal mask reaches rules out the factorised count] -> K1q on its fixed persistent grid ->
//   reduction of the partial records.
// What the host cannot know without waiting for the sort is the number of blocks, i.e. whether staging pays (a region
// must serve enough windows).  It is learnt once per call SIGNATURE (window counts per tile, width, mode): the first call with
// a signature waits for the count, later ones reuse the verdict and refresh it from the count the previous call left in
// mapped memory — steady-state loops never synchronise.
// ten radix bits per onesweep pass (rocPRIM's default is eight): a pass costs the same here (measured: 8, 9 and 10 bits within
// 5 %, 11 bits 2.6x — the look-back state outgrows the cache), so 17..20-bit keys take two passes instead of three
using Radix10 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 10,
                                        rocprim::block_radix_rank_algorithm::match>>;

extern "C++" {
template <typename KeyT>
static void launch_key_kernel(pup_ctx* c, int BR, int BC, unsigned grid, const int* dr0, const int* dc0, long long n, int nseg2t, int H,
                              int set_pairs, const pup::ExpRegion* d_eregs, int n_eregs, int W, int sh_br, int sh_er, int sh_seg,
                              int seg_shift, int clear_gap, int far_gap, KeyT* keys, int rel_bc, unsigned* hi_hist = nullptr, int hi_shift = 0, int hi_bins = 0) {
    // with the binning prepass a workgroup keys one of its tiles of 8192 windows: 256 threads x 32 (measured: 68 us; 1024 threads x 8: 85 us;
    // round 3's 256 x 4 without the digit counts: 58 us)
    const int threads = 256;
    const int per_thread = hi_hist ? pup::kBinTile / 256 : 4;
#define PUP_KEY_ARGS dr0, dc0, n, (const long long*)c->d_segend.p, nseg2t, H, set_pairs, (const pup::IdxChrom*)c->idx_chrom.p, c->n_chrom, \
        (const unsigned short*)c->bin_chrom.p, (long long)c->nbins, (const int*)c->d_brow.p, d_eregs, n_eregs, W, BR, BC, sh_br, \
        sh_er, sh_seg, seg_shift, clear_gap, far_gap, (c->band_w > 0 && !(c->variant & 256)) ? c->band_w : 0, keys, c->d_win.p, c->d_cnt32.p, hi_hist, hi_shift, hi_bins, per_thread, rel_bc
    const size_t lds = (((size_t)nseg2t + (size_t)hi_bins + 3 * (size_t)c->n_chrom) * 4 + 15) & ~(size_t)15;      // run ends | digit counts | chromosome table
    if (BR == 108 && BC == 108)
        hipLaunchKernelGGL((pup::staged_key_kernel<KeyT, 108, 108>), dim3(grid), dim3(threads), lds, c->stream, PUP_KEY_ARGS);
    else if (BR == 44 && BC == 108)
        hipLaunchKernelGGL((pup::staged_key_kernel<KeyT, 44, 108>), dim3(grid), dim3(threads), lds, c->stream, PUP_KEY_ARGS);
    else if (BR == pup::kSmallRows - 20 && BC == 108)
        hipLaunchKernelGGL((pup::staged_key_kernel<KeyT, pup::kSmallRows - 20, 108>), dim3(grid), dim3(threads), lds, c->stream, PUP_KEY_ARGS);
    else
        hipLaunchKernelGGL((pup::staged_key_kernel<KeyT, 0, 0>), dim3(grid), dim3(threads), lds, c->stream, PUP_KEY_ARGS);
#undef PUP_KEY_ARGS
}
}   // extern "C++"

static void fill_k1_args(pup_ctx* c, pup::K1Args& a, int32_t ignore_diags, uint32_t mode) {
    a.indptr = c->indptr.p; a.px = c->px.p; a.cnt32 = c->cnt32.p; a.cntf = c->float_values ? c->cntf.p : nullptr;
    a.bal = c->bal.p; a.badbits = c->badbits.p;
    const bool use_idx = c->have_idx && !(c->variant & 1);
    a.idx = use_idx ? c->idx.p : nullptr; a.idx_chrom = use_idx ? c->idx_chrom.p : nullptr;
    a.n_chrom = use_idx ? c->n_chrom : 0;
    a.rowseg = (use_idx && c->have_rowseg) ? c->rowseg.p : nullptr;
    a.rowabs = (use_idx && c->have_rowseg) ? c->rowabs.p : nullptr;
    a.weight = c->have_weight ? c->weight.p : nullptr;
    a.cov = c->have_cov ? c->cov.p : nullptr;
    a.expv = (c->nexp > 0 || (c->n_exp_regions > 0 && !c->have_exp_pair)) ? c->expv.p : nullptr;
    a.nexp = c->nexp; a.nbins = c->nbins;
    a.exp_regions = c->n_exp_regions > 0 ? c->exp_regions.p : nullptr; a.n_exp_regions = c->n_exp_regions;
    a.exp_pair = c->have_exp_pair ? c->exp_pair.p : nullptr;
    a.part_f64 = c->part_f64.p; a.part_num = c->part_num.p;
    a.counters = c->count_pixels ? c->counters.p : nullptr; a.err = c->d_err.p;
    a.nf_pixels = c->nf_count > 0 ? 1 : 0; a.nnz = c->nnz;
    a.band = c->band_w > 0 ? c->band.p + pup::kBandFront : nullptr; a.band_w = c->band_w; a.band_zero = (unsigned)(c->nbins * (long long)c->band_w);
    a.W = c->W; a.ignore_diags = ignore_diags; a.mode = mode;
}



// clear `bytes` (a multiple of 4) at p on the engine's stream: small ranges by a plain kernel, large ones by the runtime's memset
static hipError_t clear_async(pup_ctx* c, void* p, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    if (bytes > (1u << 20) || (bytes & 3u) || (reinterpret_cast<uintptr_t>(p) & 3u)) return hipMemsetAsync(p, 0, bytes, c->stream);
    const long long nw = (long long)(bytes / 4);
    hipLaunchKernelGGL(pup::zero_words_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, c->stream, reinterpret_cast<unsigned*>(p), nw);
    return hipGetLastError();
}

// ---- the hand-written block binning of the staged kernels' prepass (pup_bin.hpp) -------------------------------------------
// Digit split of a key of `end_bit` bits whose lowest `slot_bits` hold the accumulator slot: false when the key is too wide
struct BinPlan { int DL, DH; long long ntiles; };
static bool bin_plan(int end_bit, int slot_bits, long long n_items, BinPlan& bp) {
    // high digit: 10 bits for keys of up to 20 (as many, as small buckets as the partition pass's per-wave counters take without slowing
    // down — the bucket pass's time is its biggest bucket; measured on the 16-bit keys of the headline call: 8 bits 0.206, 9 0.183,
    // 10 0.171, 11 0.187 ms of prepass), 11 beyond (22-bit keys of a grouped call: 0.264 against 0.346 with 10 + 12)
    int DL = std::min(std::max(end_bit - (end_bit <= 20 ? 10 : pup::kBinMaxDigit), slot_bits), pup::kBinMaxLow);
    if (const char* e = getenv("COOLPUPPY_AMD_BIN_DH")) {        // experiments: bits of the high digit
        const int dh = std::max(0, std::min(atoi(e), std::min(end_bit, pup::kBinMaxDigit)));
        DL = std::min(std::max(end_bit - dh, slot_bits), pup::kBinMaxLow);
    }
    const int DH = std::max(end_bit - DL, 0);
    if (DH > pup::kBinMaxDigit) return false;
    if (DH > pup::kBinMaxDigit || n_items >= 0x3fffffffLL) return false;
    bp.DL = DL; bp.DH = DH; bp.ntiles = (n_items + pup::kBinTile - 1) / pup::kBinTile;
    return true;
}
// buffers of a binning run (nothing to clear: every table is written in full by the kernel before its reader)
static int bin_prepare(pup_ctx* c, const BinPlan& bp, long long n_items, bool want_low) {
    const size_t nd = (size_t)1 << bp.DH;
    const size_t nchunks = (size_t)((bp.ntiles + pup::kBinChunk - 1) / pup::kBinChunk);
    HIPCHK(c, c->d_binmeta.reserve(2 * nd + 8 + nchunks * nd));      // base[nd + 1] | blk_count[nd] | chunksum[nchunks][nd]
    HIPCHK(c, c->d_bindesc.reserve((size_t)bp.ntiles * nd));         // tilehist[ntiles][nd]: the key kernel's workgroups write their rows
    HIPCHK(c, c->d_blkkey.reserve((size_t)n_items + 1)); HIPCHK(c, c->d_bkeys.reserve((size_t)n_items + 1));
    if (want_low) HIPCHK(c, c->d_low.reserve((size_t)n_items + 8));
    return PUP_OK;
}
// keys (u32) / vals (u16) of n_items windows -> vals in block order (vals_out), block starts + keys (d_starts / d_bkeys), block count
// (n_runs).  keys_scratch: n_items dwords the partition pass writes (low digit | value); the keys themselves are free afterwards
static int bin_run(pup_ctx* c, const BinPlan& bp, long long n_items, int slot_bits, unsigned* keys, const unsigned short* vals,
                   unsigned* keys_scratch, unsigned short* vals_out, bool want_low, unsigned* n_runs) {
    const int nd = 1 << bp.DH, nl = 1 << bp.DL;
    const int nchunks = (int)((bp.ntiles + pup::kBinChunk - 1) / pup::kBinChunk);
    unsigned* base = c->d_binmeta.p; unsigned* blk_count = base + nd + 1; unsigned* chunksum = blk_count + nd + 7;
    unsigned* tilehist = c->d_bindesc.p;
    hipLaunchKernelGGL(pup::bin_chunksum_kernel, dim3((unsigned)nchunks, (unsigned)((nd + 1023) / 1024)), dim3(nd < 1024 ? nd : 1024), 0, c->stream,
                       tilehist, bp.ntiles, nd, chunksum);
    hipLaunchKernelGGL(pup::bin_scan_kernel, dim3(1), dim3(1024), 0, c->stream, chunksum, nchunks, nd, base);
    const size_t lds1 = (size_t)pup::kBinWaves * (nd < 2 ? 2 : nd) * sizeof(unsigned short), lds2_unused = 0;
    (void)lds2_unused;
    // waves per bucket: what a bucket costs beside its windows is clearing and scanning waves x 2^DL counters — eight waves for low digits
    // of up to 10 bits (the headline workload: 51 us against 60 with four), four beyond (pad 25, DL = 11: 95 us with eight)
    int bwaves = nl >= 2048 ? 4 : pup::kBucketWaves;
    if (const char* e = getenv("COOLPUPPY_AMD_BUCKET_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= pup::kBucketWaves) bwaves = v; }
    const size_t lds2 = (size_t)(bwaves + 2) * nl * sizeof(unsigned);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pup::bin_bucket_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pup::bin_partition_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    hipLaunchKernelGGL(pup::bin_partition_kernel, dim3((unsigned)bp.ntiles), dim3(pup::kWave * pup::kBinWaves), lds1, c->stream,
                       (const unsigned*)keys, vals, n_items, bp.DL, bp.DH, (const unsigned*)base, (const unsigned*)chunksum, (const unsigned*)tilehist, keys_scratch);
    // (the keys are dead now: their buffer takes the buckets' block starts)
    hipLaunchKernelGGL(pup::bin_bucket_kernel, dim3((unsigned)nd), dim3(pup::kWave * bwaves), lds2,
                       c->stream, (const unsigned*)keys_scratch, (const unsigned*)base, bp.DL, bp.DH, slot_bits, vals_out,
                       want_low ? c->d_low.p : (unsigned short*)nullptr, keys, c->d_blkkey.p, blk_count);
    hipLaunchKernelGGL(pup::bin_compact_kernel, dim3((unsigned)nd), dim3(256), 0, c->stream, (const unsigned*)base, (const unsigned*)blk_count, nd,
                       (const unsigned*)keys, (const unsigned*)c->d_blkkey.p, c->d_starts.p, c->d_bkeys.p, n_runs);
    HIPCHK(c, hipGetLastError());
    return PUP_OK;
}

// coverage vectors of the call as a pass of their own (see cov_vectors_kernel): enqueued behind the staged pile-up, adds into
// the cov slots of the accumulators.  d_segend holds the call's (tile, flip) run ends.
static int cov_pass(pup_ctx* c, const int* dr0, const int* dc0, int64_t n, uint32_t mode) {
    const int W = c->W, T = c->T;
    const size_t L = 2 * (size_t)W, nchunks = (size_t)((n + pup::kCovChunk - 1) / pup::kCovChunk), nrec = nchunks + (size_t)T;
    HIPCHK(c, c->cov_rec.reserve(nrec * L)); HIPCHK(c, c->cov_owner.reserve(nrec));
    HIPCHK(c, hipMemsetAsync(c->cov_owner.p, 0, nrec * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(pup::cov_vectors_kernel, dim3((unsigned)nchunks), dim3(256), 4 * L * sizeof(double), c->stream, dr0, dc0, (long long)n,
                       (const long long*)c->d_segend.p, T, (const double*)c->cov.p, (long long)c->nbins, W, (mode & PUP_MODE_TRANSPOSE) ? 1 : 0,
                       c->cov_rec.p, c->cov_owner.p);
    hipLaunchKernelGGL(pup::cov_reduce_kernel, dim3((unsigned)((L + 63) / 64), (unsigned)T), dim3(64), 0, c->stream, (const double*)c->cov_rec.p,
                       (const unsigned*)c->cov_owner.p, (const long long*)c->d_segend.p, W, (int)((size_t)W * W + L), c->acc_f64.p);
    HIPCHK(c, hipGetLastError());
    return PUP_OK;
}

// returns PUP_OK when the call was piled up here (K1q + reduction enqueued), 1 when the per-window kernels must take it,
// a negative code on error.  ev[0..2]: optional timing events (prepass start, K1 start, K1 end).
// shares of the sixteen waves of K1q's workgroup (StagedArgs::wgt, 1/1024 of an average wave's): the SIMD favours some of its four
// waves whatever their priorities, the look-ahead work sits at a different place in every wave's slice, and the workgroup waits for its
// slowest wave at the barrier behind the window loop — so a wave's slice is sized by its measured pace.  Rounds 3-5 had one weight per
// AGE (wave >> 2: 0.89 / 1.02 / 1.0 / 0.83 = 911 / 1044 / 1024 / 850); the table below is the fixed point of tools/ab/tune_weights.py
// (four rounds of "phase clocks -> shares towards equal time"; window-loop clocks per control wave 832-882 k -> 852-866 k, wait at the
// barrier 111 k -> 104 k clocks: what is left of that wait is per-block scatter, not a standing imbalance; K1 -1.5 %,
// profiles/r05_k1q_weights.txt).  COOLPUPPY_AMD_K1Q_WEIGHTS="w0,...,w15" (in 1/1024) replaces it (experiments).
static void staged_wave_weights(unsigned short (&wgt)[16]) {
    static const unsigned short kDefault[16] = {900, 922, 895, 899, 1085, 1062, 1023, 1027, 1062, 1068, 1039, 1041, 799, 858, 825, 839};
    static unsigned short table[16];
    static bool have = false;
    if (!have) {
        for (int w = 0; w < 16; ++w) table[w] = kDefault[w];
        if (const char* e = getenv("COOLPUPPY_AMD_K1Q_WEIGHTS")) {
            int v[16], n = 0;
            const char* p = e;
            while (n < 16 && *p) { char* q; const long x = strtol(p, &q, 10); if (q == p) break; v[n++] = (int)x; p = *q == ',' ? q + 1 : q; }
            if (n == 16) { bool ok = true; for (int w = 0; w < 16; ++w) ok = ok && v[w] > 0 && v[w] < 65536; if (ok) for (int w = 0; w < 16; ++w) table[w] = (unsigned short)v[w]; }
        }
        have = true;
    }
    for (int w = 0; w < 16; ++w) wgt[w] = table[w];
}

static int staged_run(pup_ctx* c, const int* dr0, const int* dc0, int64_t n, const int64_t* tile_ptr, const int64_t* flip_from,
                      int32_t ignore_diags, uint32_t mode, bool rescale, hipEvent_t* ev) {
    const int W = c->W, T = c->T;
    c->last_stagings = 0; c->last_staged = false;
    const bool force = (c->variant & 8) != 0, forbid = (c->variant & 16) != 0 || c->float_values;     // (float pixel values: the kernels on the `bal` table)
    const bool use_idx_t = c->have_idx && !(c->variant & 1);
    if (c->n_chrom > pup::kKeyMaxChrom && !forbid && !rescale && ignore_diags >= 0 && n >= c->tiled_min)
        c->off_staged(8u, "the table has more chromosomes / scaffolds than the staged kernels' key pass keeps in LDS (3072)", (long long)n);
    if (forbid || rescale || (mode & (PUP_MODE_EXPECTED | PUP_MODE_TRANSPOSE)) || (c->variant & 2) || !use_idx_t ||
        ignore_diags < 0 || !tiled_supported(W) || n >= 0x7fff0000LL || c->n_chrom > pup::kKeyMaxChrom || !(force || n >= ((mode & PUP_MODE_OOE) ? c->tiled_min_ooe : c->tiled_min)) ||
        2 * T > pup::kMaxSegCount || T > pup::kMaxStagedTiles || !c->bin_chrom.p || !c->h_flags || !c->ev_key ||
        (int)c->h_chroms.size() != c->n_chrom)
        return 1;
    // coverage vectors do not ride inside the staged kernel any more (they forced its fat geometry): a pass of their own, below
    const bool cov_sep = (mode & PUP_MODE_COV) && c->have_cov;
    const bool extra = c->count_pixels;
    const bool small21 = (c->variant & 128) != 0;
    const StagedGeo geo = staged_geometry(W, (mode & PUP_MODE_OOE) != 0, extra, small21);
    const int BR = geo.RSR - W + 1, BC = geo.RSC - W + 1;
    const int G = c->n_cu * ((small21 || geo.RSR * geo.RSC > pup::kSmallRows * 128) ? 1 : 2);   // persistent workgroups (one / two per CU by LDS; the two-buffer geometry: one)
    // a staged region must serve this many windows on average to pay for its staging
    // (per-window division by expected makes the per-window kernel three times dearer: staging pays much earlier there)
    const long long min_per_block = ((mode & PUP_MODE_OOE) ? 2LL : 8LL) * (geo.RSR * geo.RSC) / (64 * 64);
    const int n_eregs = ((mode & PUP_MODE_OOE) && c->n_exp_regions > 0 && !c->have_exp_pair) ? c->n_exp_regions : 0;
    auto nbits = [](unsigned long long v) { int b = 1; while ((v >> b) != 0) ++b; return b; };

    // ---- what the last call with this signature found: staged or not, without waiting ---------------------------------
    std::vector<long long> sig;
    sig.reserve(8 + 2 * (size_t)T);
    sig.push_back(n); sig.push_back(T); sig.push_back(W); sig.push_back((long long)(mode & (PUP_MODE_OOE | PUP_MODE_COV)));
    sig.push_back(ignore_diags); sig.push_back(flip_from ? 1 : 0); sig.push_back(c->variant & (4 | 64 | 128 | 256 | 512)); sig.push_back(extra ? 1 : 0);
    mode &= ~(uint32_t)(cov_sep ? PUP_MODE_COV : 0);     // (the kernels below never see the bit)
    for (int t = 0; t <= T; ++t) sig.push_back(tile_ptr[t]);
    if (flip_from) for (int t = 0; t < T; ++t) sig.push_back(flip_from[t]);
    bool known = (sig == c->hint_sig) && c->hint_blocks >= 0;
    if (known && c->h_flags[5] == c->hint_ticket) c->hint_blocks = (long long)c->h_flags[4];   // the previous call's count has landed
    if (known && !force && c->hint_blocks * min_per_block > n) {
        // too sparse last time: per-window kernels — but window POSITIONS are not part of the signature, so every 16th such
        // call measures the block count again (as a first call does) instead of trusting the memory for ever
        if (++c->hint_sparse_calls % 16 != 0) return 1;
        known = false;
    }

    // block rows are numbered compactly over the genome (fewer key bits = fewer radix passes)
    long long max_len = 1, n_brows = 0;
    std::vector<int> brow_base((size_t)c->n_chrom);
    for (int k = 0; k < c->n_chrom; ++k) {
        const long long len = c->h_chroms[(size_t)k].end - c->h_chroms[(size_t)k].start;
        max_len = std::max<long long>(max_len, len);
        brow_base[(size_t)k] = (int)n_brows;
        n_brows += (len + BR - 1) / BR;
    }
    // small host tables are re-sent only when they differ from what the device holds (steady-state loops: never)
    if (brow_base != c->brow_sent) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_brow.cap < (size_t)c->n_chrom) c->brow_sent.clear();
        HIPCHK(c, c->d_brow.reserve((size_t)c->n_chrom));
        HIPCHK(c, hipMemcpy(c->d_brow.p, brow_base.data(), brow_base.size() * sizeof(int), hipMemcpyHostToDevice));
        c->brow_sent = brow_base;
    }
    const bool paired = T >= 2 && (T % 2) == 0 && !(c->variant & 64);
    const int H = paired ? T / 2 : 0;
    // several tile pairs (grouped pile-ups: by strand, by distance, ...): four pairs share a pass — a staged region then serves
    // the windows of eight tiles, each piled up by its own team of waves — instead of every pair staging the matrix again
    const bool sets = paired && H >= 2 && W <= 21 && !extra && !small21 && !(c->variant & 512);
    const int ACC = sets ? 8 : (paired ? 2 : 1), set_pairs = sets ? pup::kSetPairs : 1;
    const int U = sets ? (H + set_pairs - 1) / set_pairs : (paired ? H : T);
    const int slot_bits = sets ? pup::kSetSlotBits : 0;
    const int nseg = 2 * U;
    const int seg_shift = flip_from ? 0 : 1;            // no flipped windows: the flip bit is left out of the key
    // (tile, flip) runs of the caller's order, for the key kernel
    std::vector<long long> htab;
    for (int t = 0; t < T; ++t) { htab.push_back(flip_from ? flip_from[t] : tile_ptr[t + 1]); htab.push_back(tile_ptr[t + 1]); }
    const int nseg2t = (int)htab.size();
    // Column blocks: numbered from the chromosome's start (a chromosome's width / block side: 8 bits at 10 kb) — or, while the table has a
    // dense band and no call has said otherwise, from the one under the block row's first bin (+ a bias of two blocks to the left): a
    // window inside the band then needs (band + block) / block + 4 numbers, 4 bits, which brings grouped calls (tile sets: 3 slot +
    // 3 segment bits on top of 12 row bits) within the 23 bits of the hand-written binning.  The key kernel reports a window that does
    // not fit together with those that leave the band; the call is then redone with the plain numbering (below) and the table
    // remembers it (no_rel_bc: a workload of far-apart windows pays the second prepass once).
    const int rel_bias = (W + BC - 1) / BC + 1;
    const int rel_bc = (c->band_w > 0 && !(c->variant & 256) && !extra && !c->no_rel_bc) ? rel_bias + 1 : 0;
    const int bits_bc = rel_bc ? nbits((unsigned long long)((c->band_w + BR) / BC + 2 + rel_bias)) : nbits((unsigned long long)(max_len / BC + 1));
    const int bits_br = nbits((unsigned long long)n_brows + 1);
    // the expected region of a window is part of its key — unless every chromosome lies inside ONE region (or none): a block
    // never leaves its chromosome, so its region follows from its origin and the key stays short (one radix pass fewer)
    bool er_in_key = n_eregs > 0;
    if (er_in_key && (int)c->h_exp_bounds.size() == n_eregs) {
        bool whole = true;
        for (const auto& ch : c->h_chroms) {
            bool inside = false, touched = false;
            for (const auto& rg : c->h_exp_bounds) {
                if (rg.first <= ch.start && rg.second >= ch.end) inside = true;
                else if (rg.first < ch.end && rg.second > ch.start) touched = true;
            }
            if (touched && !inside) { whole = false; break; }
            if (touched && inside) { whole = false; break; }          // (overlapping regions: keep the general way)
        }
        if (whole) er_in_key = false;
    }
    const int bits_er = er_in_key ? nbits((unsigned long long)n_eregs) : 0;
    const int sh_br = bits_bc, sh_er = sh_br + bits_br, sh_seg = sh_er + bits_er;
    const int nseg_key = nseg >> seg_shift;
    const int end_bit = slot_bits + sh_seg + (nseg_key > 1 ? nbits((unsigned long long)(nseg_key - 1)) : 0);
    if (end_bit > 64) return 1;
    const bool k32 = end_bit <= 32;

    // ---- buffers (grow-only: steady-state calls allocate nothing) -----------------------------------------------------
    const int n_spans = (int)((n + pup::kSpan - 1) / pup::kSpan);
    const size_t ncnt = 4;                               // [0] ineligible [1] unclear [2] outside the band [3] blocks, then the span counters
    const size_t W2 = (size_t)W * W, Lf = W2 + 2 * (size_t)W;
    const size_t nrec = 8 * (2 * (size_t)T + (size_t)G);  // record slot * (2T + G) + tile * 2 + flip + workgroup (StagedArgs::rec_owner), slots <= 8
    HIPCHK(c, c->d_win.reserve((size_t)n + 8)); HIPCHK(c, c->d_win2.reserve((size_t)n + 8));   // +8: K1q fetches four window values at a time
    if (c->d_segend.cap < htab.size()) c->htab_sent.clear();          // the buffer is about to move
    HIPCHK(c, c->d_segend.reserve(htab.size()));
    // (the records' valid flags live behind the counters: one fill clears both)
    HIPCHK(c, c->d_cnt32.reserve(ncnt + (size_t)n_spans + (nrec + 1) / 2));
    HIPCHK(c, c->d_starts.reserve((size_t)n + 1));
    HIPCHK(c, c->d_blocks.reserve((size_t)std::min<long long>(n, (n_brows + 1) * (max_len / BC + 2) * (long long)std::max(nseg_key, 1) * (n_eregs + 1))));   // distinct keys at most
    HIPCHK(c, c->d_wgfirst.reserve((size_t)G + 1));
    HIPCHK(c, c->part_f64.reserve(nrec * Lf)); HIPCHK(c, c->part_num.reserve(nrec * W2));
    if (k32) { HIPCHK(c, c->d_k32.reserve((size_t)n)); HIPCHK(c, c->d_k32b.reserve((size_t)n)); }
    else     { HIPCHK(c, c->d_keys.reserve((size_t)n)); HIPCHK(c, c->d_keys2.reserve((size_t)n)); }
    if (htab != c->htab_sent) {
        HIPCHK(c, hipStreamSynchronize(c->stream));     // kernels of an earlier call may still read the old table
        HIPCHK(c, hipMemcpy(c->d_segend.p, htab.data(), htab.size() * 8, hipMemcpyHostToDevice));
        c->htab_sent = htab;
    }
    // wave teams (StagedArgs::teams): per pass unit, the waves of a workgroup are dealt to the unit's tiles in proportion to
    // the tiles' windows (every tile with windows gets a wave; then whoever has the most windows per wave gets the next).
    // Two tables: for the 16-wave kernels (factorised counts) and for the 8-wave ones — which it will be is only known
    // after the key kernel.
    const int nw_fact = geo.NW, nw_full = geo.NW == 16 ? 8 : geo.NW;
    std::vector<unsigned char> teams((size_t)2 * U * 16, 0);
    if (ACC > 1)
        for (int v = 0; v < 2; ++v) {
            const int nw = v ? nw_full : nw_fact;
            for (int u = 0; u < U; ++u) {
                long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int nwv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, used = 0;
                for (int q = 0; q < ACC; ++q) {
                    const int t = ACC == 8 ? pup::staged_tile<8>(u, q, H) : pup::staged_tile<2>(u, q, H);
                    cnt[q] = t >= 0 ? tile_ptr[t + 1] - tile_ptr[t] : 0;
                    if (cnt[q] > 0) { nwv[q] = 1; ++used; }
                }
                for (; used > 0 && used < nw; ++used) {
                    int best = -1;
                    for (int q = 0; q < ACC; ++q)
                        if (nwv[q] > 0 && (best < 0 || cnt[q] * nwv[best] > cnt[best] * nwv[q])) best = q;
                    ++nwv[best];
                }
                unsigned char* row = teams.data() + ((size_t)v * U + u) * 16;
                int at = 0;
                for (int q = 0; q <= ACC; ++q) { row[q] = (unsigned char)at; if (q < ACC) at += nwv[q]; }
                for (int q = ACC + 1; q < 16; ++q) row[q] = (unsigned char)at;
            }
        }
    if (ACC > 1 && teams != c->teams_sent) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_teams.cap < teams.size()) c->teams_sent.clear();
        HIPCHK(c, c->d_teams.reserve(teams.size()));
        HIPCHK(c, hipMemcpy(c->d_teams.p, teams.data(), teams.size(), hipMemcpyHostToDevice));
        c->teams_sent = teams;
    }
    // block order: the hand-written binning (pup_bin.hpp) for keys of up to 23 bits, else (or with variant bit 29) the library's radix sort
    BinPlan bp{};
    const bool use_bin = k32 && !(c->variant & 1024) && bin_plan(end_bit, slot_bits, (long long)n, bp);
    c->last_prepass = use_bin ? "binning" : "library_sort";
    size_t tmp_bytes = 0;
    hipError_t se = hipSuccess;
    if (!use_bin) {
        se = k32 ? rocprim::radix_sort_pairs<Radix10>(nullptr, tmp_bytes, c->d_k32.p, c->d_k32b.p, c->d_win.p, c->d_win2.p, (size_t)n, 0, end_bit, c->stream)
                 : rocprim::radix_sort_pairs<Radix10>(nullptr, tmp_bytes, c->d_keys.p, c->d_keys2.p, c->d_win.p, c->d_win2.p, (size_t)n, 0, end_bit, c->stream);
        if (se == hipSuccess) se = c->d_sorttmp.reserve(tmp_bytes + 16);
        if (se != hipSuccess) return fail(c, PUP_EHIP, "pup_accumulate: block sort: %s", hipGetErrorString(se));
    }

    int block_cost = pup::kBlockCost;
    if (const char* e = getenv("COOLPUPPY_AMD_K1Q_COST")) { const int v = atoi(e); if (v > 0) block_cost = v; }     // experiments
    // ---- prepass, all on the stream -------------------------------------------------------------------------------------
    if (ev) HIPCHK(c, hipEventRecord(ev[0], c->stream));
    if (use_bin) { const int brc = bin_prepare(c, bp, (long long)n, slot_bits > 0); if (brc != PUP_OK) return brc; }
    HIPCHK(c, clear_async(c, c->d_cnt32.p, (ncnt + (size_t)n_spans + (nrec + 1) / 2) * sizeof(unsigned)));
    unsigned short* const d_recvalid = reinterpret_cast<unsigned short*>(c->d_cnt32.p + ncnt + (size_t)n_spans);
    const unsigned gk4 = use_bin ? (unsigned)bp.ntiles : (unsigned)((n + 1023) / 1024);      // key kernel: four windows per thread, or a binning tile per workgroup
    const pup::ExpRegion* d_eregs = n_eregs > 0 ? c->exp_regions.p : nullptr;
    const unsigned ticket = ++c->ticket;
    // observed over expected with a by-diagonal expected whose unusable diagonals are all ignored ones: a cell's validity is
    // then its row / column masks, as without expected — provided no window reaches past the end of its expected vector
    // (counted by the key kernel together with the windows a diagonal mask reaches)
    const bool ooe = (mode & PUP_MODE_OOE) != 0;
    const bool ooe_vec = ooe && !c->have_exp_pair && (c->nexp > 1 || c->n_exp_regions > 0);
    bool ooe_clean = false;
    int far_gap = 0x7fffffff;                            // a window reaching this diagonal rules the factorised count out
    if (ooe && !c->have_exp_pair && !extra && !c->h_exp.empty()) {
        if (c->nexp == 1) ooe_clean = c->h_exp[0] == c->h_exp[0] && c->h_exp[0] != 0.0;     // one scalar for every cell
        else if (ooe_vec) {
            const long long far = c->exp_far(ignore_diags);
            ooe_clean = far > (long long)ignore_diags + W;                   // (some room for windows at all)
            if (ooe_clean) far_gap = (int)std::min<long long>(far, 0x7fffffff);
        }
    }
    const int n_eregs_key = er_in_key ? n_eregs : 0;
    if (k32) launch_key_kernel<unsigned>(c, BR, BC, gk4, dr0, dc0, (long long)n, nseg2t, H, set_pairs, d_eregs, n_eregs_key, W, sh_br, sh_er,
                                         sh_seg, seg_shift, ignore_diags + W - 1, far_gap, c->d_k32.p, rel_bc, use_bin ? c->d_bindesc.p : nullptr, bp.DL, 1 << bp.DH);
    else launch_key_kernel<unsigned long long>(c, BR, BC, gk4, dr0, dc0, (long long)n, nseg2t, H, set_pairs, d_eregs, n_eregs_key, W, sh_br,
                                               sh_er, sh_seg, seg_shift, ignore_diags + W - 1, far_gap, c->d_keys.p, rel_bc);
    hipLaunchKernelGGL(pup::staged_publish_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned*)c->d_cnt32.p, 3,
                       (volatile unsigned*)c->d_flags, ticket);
    HIPCHK(c, hipEventRecord(c->ev_key, c->stream));
    unsigned* d_spans = c->d_cnt32.p + ncnt;
    const unsigned gt = (unsigned)std::min<long long>(4096, (n + 255) / 256);
    if (use_bin) {
        const int brc = bin_run(c, bp, (long long)n, slot_bits, c->d_k32.p, c->d_win.p, c->d_k32b.p, c->d_win2.p, slot_bits > 0, c->d_cnt32.p + 3);
        if (brc != PUP_OK) return brc;
        hipLaunchKernelGGL((pup::staged_table_kernel<unsigned>), dim3(gt), dim3(256), 0, c->stream, (const unsigned*)c->d_starts.p,
                           (const unsigned*)(c->d_cnt32.p + 3), (long long)n, (const unsigned*)nullptr,
                           (const unsigned short*)c->d_win2.p, (const int*)c->d_brow.p, (const pup::IdxChrom*)c->idx_chrom.p,
                           c->n_chrom, W, W, geo.RSR, geo.RSC, 1, block_cost, sh_br, sh_er, sh_seg, seg_shift, slot_bits, n_eregs, er_in_key ? 1 : 0, d_eregs,
                           (const unsigned long long*)c->badbits.p, c->d_blocks.p, c->d_wgfirst.p, G, (const unsigned*)c->d_bkeys.p,
                           slot_bits > 0 ? (const unsigned short*)c->d_low.p : (const unsigned short*)nullptr,
                           (volatile unsigned*)(c->d_flags + 4), ticket, rel_bc);
    } else if (k32) {
        se = rocprim::radix_sort_pairs<Radix10>(c->d_sorttmp.p, tmp_bytes, c->d_k32.p, c->d_k32b.p, c->d_win.p, c->d_win2.p,
                                                (size_t)n, 0, end_bit, c->stream);
        if (se != hipSuccess) return fail(c, PUP_EHIP, "pup_accumulate: block sort: %s", hipGetErrorString(se));
        hipLaunchKernelGGL((pup::count_heads_kernel<unsigned>), dim3((unsigned)n_spans), dim3(256), 0, c->stream,
                           (const unsigned*)c->d_k32b.p, (long long)n, slot_bits, d_spans);
        hipLaunchKernelGGL((pup::block_starts_kernel<unsigned>), dim3((unsigned)n_spans), dim3(256), 0, c->stream,
                           (const unsigned*)c->d_k32b.p, (long long)n, slot_bits, (const unsigned*)d_spans, c->d_starts.p, c->d_cnt32.p + 3);
        hipLaunchKernelGGL((pup::staged_table_kernel<unsigned>), dim3(gt), dim3(256), 0, c->stream, (const unsigned*)c->d_starts.p,
                           (const unsigned*)(c->d_cnt32.p + 3), (long long)n, (const unsigned*)c->d_k32b.p,
                           (const unsigned short*)c->d_win2.p, (const int*)c->d_brow.p, (const pup::IdxChrom*)c->idx_chrom.p,
                           c->n_chrom, W, W, geo.RSR, geo.RSC, 1, block_cost, sh_br, sh_er, sh_seg, seg_shift, slot_bits, n_eregs, er_in_key ? 1 : 0, d_eregs,
                           (const unsigned long long*)c->badbits.p, c->d_blocks.p, c->d_wgfirst.p, G, (const unsigned*)nullptr, (const unsigned short*)nullptr,
                           (volatile unsigned*)(c->d_flags + 4), ticket, rel_bc);
    } else {
        se = rocprim::radix_sort_pairs<Radix10>(c->d_sorttmp.p, tmp_bytes, c->d_keys.p, c->d_keys2.p, c->d_win.p, c->d_win2.p,
                                                (size_t)n, 0, end_bit, c->stream);
        if (se != hipSuccess) return fail(c, PUP_EHIP, "pup_accumulate: block sort: %s", hipGetErrorString(se));
        hipLaunchKernelGGL((pup::count_heads_kernel<unsigned long long>), dim3((unsigned)n_spans), dim3(256), 0, c->stream,
                           (const unsigned long long*)c->d_keys2.p, (long long)n, slot_bits, d_spans);
        hipLaunchKernelGGL((pup::block_starts_kernel<unsigned long long>), dim3((unsigned)n_spans), dim3(256), 0, c->stream,
                           (const unsigned long long*)c->d_keys2.p, (long long)n, slot_bits, (const unsigned*)d_spans, c->d_starts.p, c->d_cnt32.p + 3);
        hipLaunchKernelGGL((pup::staged_table_kernel<unsigned long long>), dim3(gt), dim3(256), 0, c->stream, (const unsigned*)c->d_starts.p,
                           (const unsigned*)(c->d_cnt32.p + 3), (long long)n, (const unsigned long long*)c->d_keys2.p,
                           (const unsigned short*)c->d_win2.p, (const int*)c->d_brow.p, (const pup::IdxChrom*)c->idx_chrom.p,
                           c->n_chrom, W, W, geo.RSR, geo.RSC, 1, block_cost, sh_br, sh_er, sh_seg, seg_shift, slot_bits, n_eregs, er_in_key ? 1 : 0, d_eregs,
                           (const unsigned long long*)c->badbits.p, c->d_blocks.p, c->d_wgfirst.p, G, (const unsigned*)nullptr, (const unsigned short*)nullptr,
                           (volatile unsigned*)(c->d_flags + 4), ticket, rel_bc);
    }
    HIPCHK(c, hipGetLastError());

    // ---- K1q (+ reduction below) ---------------------------------------------------------------------------------------------
    const size_t Li0 = W2;
    auto launch_k1 = [&](bool fact, bool band) -> int {
        pup::K1Args a{};
        fill_k1_args(c, a, ignore_diags, mode);
        pup::StagedArgs sa{};
        sa.blocks = c->d_blocks.p; sa.win = c->d_win2.p; sa.wg_first = c->d_wgfirst.p; sa.U = U; sa.PH = H; sa.rec_owner = d_recvalid; sa.T = T;
        sa.teams = ACC > 1 ? c->d_teams.p + (fact ? 0 : (size_t)U * 16) : nullptr;      // (StagedGeom: 16 waves only with factorised counts)
        sa.debug = c->debug_phases & 0xb;
        staged_wave_weights(sa.wgt);
        sa.timing = nullptr;
        if (c->debug_phases & 4) {                       // phase clocks (diagnostics): [G][16][8] long long, read by pup_debug_timing
            HIPCHK(c, c->d_timing.reserve((size_t)G * 16 * 8));
            HIPCHK(c, hipMemsetAsync(c->d_timing.p, 0, (size_t)G * 16 * 8 * sizeof(long long), c->stream));
            sa.timing = c->d_timing.p; c->timing_G = G;
        }
        if (ev) HIPCHK(c, hipEventRecord(ev[1], c->stream));
        const pup::StagedLaunch sl{W, G, ACC, fact, extra, small21, band, (c->variant & 8192) != 0};
        if (!pup::launch_staged(sl, a, sa, c->stream))
            return fail(c, PUP_ENOTSUP, "pup_accumulate: staged kernel not built for W=%d", W);
        HIPCHK(c, hipGetLastError());
        if (ev) HIPCHK(c, hipEventRecord(ev[2], c->stream));
        return PUP_OK;
    };
    // A call shape seen before is launched on the LAST call's verdict, right behind the prepass — before the host has this
    // call's: the stream then never waits for the host (on a busy host the wait below could outlast the sort, 0.1 ms of idle
    // GPU per call were measured on one box).  The verdict is checked afterwards; the reduction — the only thing that touches
    // the accumulators — is launched only once it is known that the right kernel ran (else the records are dropped and the
    // right one runs)
    bool speculated = false;
    const bool spec_fact = c->hint_fact, spec_band = c->hint_band && c->band_w > 0;
    if (known && c->hint_have_verdict && (!c->hint_band || c->band_w > 0)) {
        const int rc1 = launch_k1(spec_fact, spec_band);
        if (rc1 != PUP_OK) return rc1;
        speculated = true;
    }

    // ---- the host's part: the key kernel's verdict (an event long reached: the sort is still running) -----------------
    HIPCHK(c, hipEventSynchronize(c->ev_key));
    if (c->h_flags[3] != ticket) return fail(c, PUP_EHIP, "pup_accumulate: the key kernel's verdict did not arrive");
    if (rel_bc && c->h_flags[2] != 0 && c->h_flags[0] == 0) {
        // a window beyond the band (or below the diagonal): its column block did not fit the diagonal-relative field — the block order of
        // this prepass is void.  Once more with the plain numbering; nothing of this attempt has touched the accumulators (a kernel
        // launched on the last call's verdict wrote records nobody will own: the next prepass clears the marks).
        c->no_rel_bc = true;
        c->hint_have_verdict = false;
        return staged_run(c, dr0, dc0, n, tile_ptr, flip_from, ignore_diags, mode | (cov_sep ? PUP_MODE_COV : 0u), rescale, ev);
    }
    const bool band = c->band_w > 0 && !(c->variant & 256) && !extra && c->h_flags[2] == 0;    // every window inside the dense band
    if (c->h_flags[0] != 0) {                                              // a window the index does not cover: the per-window kernels take the call
        c->hint_have_verdict = false;
        c->off_staged(1u, "some windows are not inside one chromosome (pass inter-chromosomal windows with ignore_diags < 0)", (long long)n);
        return 1;
    }
    const bool fact = (!ooe || ooe_clean) && c->h_flags[1] == 0 && !(c->variant & 4);
    if (!known) {
        // first call with this signature: wait for the block count once and decide whether staging pays
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->hint_sig = sig;
        c->hint_blocks = (long long)c->h_flags[4];
        c->hint_have_verdict = false; c->hint_sparse_calls = 0;
        if (!force && c->hint_blocks * min_per_block > n) { c->hint_ticket = ticket; return 1; }
    }
    c->hint_ticket = ticket;
    if (speculated && (fact != spec_fact || band != spec_band)) {
        HIPCHK(c, hipMemsetAsync(d_recvalid, 0, nrec * sizeof(unsigned short), c->stream));     // the wrong kernel's records: dropped
        speculated = false;
    }
    if (!speculated) {
        const int rc1 = launch_k1(fact, band);
        if (rc1 != PUP_OK) return rc1;
    }
    c->hint_fact = fact; c->hint_band = band; c->hint_have_verdict = true;
    const int Li = (int)Li0;
    hipLaunchKernelGGL(pup::reduce_staged_kernel, dim3((unsigned)((Lf + Li + 63) / 64), (unsigned)T), dim3(64, pup::kRedParts), 0,
                       c->stream, (const double*)c->part_f64.p, (const unsigned*)c->part_num.p,
                       (const unsigned short*)d_recvalid, G, T, ACC, H, (int)Lf, Li, c->acc_f64.p, c->acc_i64.p,
                       c->acc_i64.p + (size_t)T * W2, (const long long*)c->d_segend.p);
    c->counts_added = true;
    HIPCHK(c, hipGetLastError());
    if (cov_sep) { const int crc = cov_pass(c, dr0, dc0, n, mode); if (crc != PUP_OK) return crc; }
    c->last_staged = true;
    return PUP_OK;
}


// ---- the staged path for WIDE windows (K1w, pup_wide.hpp): sub-window items, block-order prepass, persistent kernel ----
// Eligible calls: W >= 32 (no upper limit), plain / observed-over-expected, cis with a diagonal mask (ignore_diags >= 0), the
// dense band table present and holding every window, every window inside one chromosome.  Coverage vectors, pixel
// statistics, inter-chromosomal and expected-only passes stay with the per-window kernels (any width, see accumulate_impl).
// returns PUP_OK when the call was piled up here, 1 when the per-window kernels must take it, a negative code on error.
static int wide_run(pup_ctx* c, const int* dr0, const int* dc0, int64_t n, const int64_t* tile_ptr, const int64_t* flip_from,
                    int32_t ignore_diags, uint32_t mode, bool rescale, hipEvent_t* ev) {
    const int W = c->W, T = c->T;
    const bool force = (c->variant & 8) != 0, forbid = (c->variant & 16) != 0 || c->float_values;
    const bool ooe = (mode & PUP_MODE_OOE) != 0;
    const bool cov_sep = (mode & PUP_MODE_COV) && c->have_cov;
    if (forbid || rescale || (mode & (PUP_MODE_EXPECTED | PUP_MODE_TRANSPOSE)) || (c->variant & (1 | 2 | 256)) || c->count_pixels ||
        !c->have_idx || c->band_w <= 0 || ignore_diags < 0 || W < 32 || !(force || n >= c->wide_min) || 2 * T > pup::kMaxSegCount ||
        !c->bin_chrom.p || !c->h_flags || !c->ev_key || (int)c->h_chroms.size() != c->n_chrom || (ooe && c->have_exp_pair))
        return 1;
    const pup::WideGeom geo = pup::wide_geometry(W);
    const int NG = geo.NGr * geo.NGc;
    const long long n_items = (long long)n * NG;
    if (n_items >= 0x7fff0000LL || c->n_chrom > pup::kKeyMaxChrom) return 1;
    const int RS = 128;
    const int BR = RS - geo.SH + 1, BC = RS - geo.SW + 1;
    const int G = c->n_cu;                               // persistent workgroups: one per CU (the region takes the LDS)
    auto nbits = [](unsigned long long v) { int b = 1; while ((v >> b) != 0) ++b; return b; };

    // observed over expected: the by-diagonal vector of a block's windows is that of the region its origin lies in — valid
    // when no chromosome is split between expected regions (the usual per-chromosome / per-arm-free table); else K1b
    const int n_eregs = (ooe && c->n_exp_regions > 0 && !c->have_exp_pair) ? c->n_exp_regions : 0;
    if (n_eregs > 0) {
        if ((int)c->h_exp_bounds.size() != n_eregs) return 1;
        for (const auto& ch : c->h_chroms) {
            bool inside = false, touched = false;
            for (const auto& rg : c->h_exp_bounds) {
                if (rg.first <= ch.start && rg.second >= ch.end) inside = true;
                else if (rg.first < ch.end && rg.second > ch.start) touched = true;
            }
            if (touched) return 1;
            (void)inside;
        }
    }
    long long max_len = 1, n_brows = 0;
    std::vector<int> brow_base((size_t)c->n_chrom);
    for (int k = 0; k < c->n_chrom; ++k) {
        const long long len = c->h_chroms[(size_t)k].end - c->h_chroms[(size_t)k].start;
        max_len = std::max<long long>(max_len, len);
        brow_base[(size_t)k] = (int)n_brows;
        n_brows += (len + BR - 1) / BR;
    }
    if (brow_base != c->brow_sent) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_brow.cap < (size_t)c->n_chrom) c->brow_sent.clear();
        HIPCHK(c, c->d_brow.reserve((size_t)c->n_chrom));
        HIPCHK(c, hipMemcpy(c->d_brow.p, brow_base.data(), brow_base.size() * sizeof(int), hipMemcpyHostToDevice));
        c->brow_sent = brow_base;
    }
    const int seg_shift = flip_from ? 0 : 1;
    std::vector<long long> htab;
    for (int t = 0; t < T; ++t) { htab.push_back(flip_from ? flip_from[t] : tile_ptr[t + 1]); htab.push_back(tile_ptr[t + 1]); }
    const int nseg2t = (int)htab.size();
    // column blocks are numbered from the one under the block row's first bin (+ a bias for sub-windows left of it): every window K1w
    // serves lies inside the band, so the field is (band + block) / block + bias wide instead of a chromosome's width — 5 bits
    // instead of 8 at 10 kb, which keeps 16-group calls (201-bin windows) within the hand-written binning's 23 bits.  A window that does
    // not fit is counted with those that leave the band: the call goes to the per-window kernel, as it would anyway.
    const int rel_bias = (W + BC - 1) / BC + 2, rel_bc = rel_bias + 1;
    const int bits_bc = nbits((unsigned long long)((c->band_w + BR) / BC + 2 + rel_bias)), bits_br = nbits((unsigned long long)n_brows + 1);
    const int sh_br = bits_bc, sh_seg = sh_br + bits_br;
    const long long nseg_key = ((long long)(2 * T) >> seg_shift) * NG;
    const int end_bit = sh_seg + (nseg_key > 1 ? nbits((unsigned long long)(nseg_key - 1)) : 0);
    if (end_bit > 64) return 1;
    const bool k32 = end_bit <= 32;
    const long long nseg_full = 2LL * T * NG;
    const long long nrec = nseg_full + G;
    if ((unsigned long long)nrec * pup::kWideRec * 12ull > (8ull << 30)) return 1;      // (by-window sized tile counts: per-window kernels)

    const int n_spans = (int)((n_items + pup::kSpan - 1) / pup::kSpan);
    const size_t ncnt = 4;
    HIPCHK(c, c->d_win.reserve((size_t)n_items + 8)); HIPCHK(c, c->d_win2.reserve((size_t)n_items + 8));
    if (c->d_segend.cap < htab.size()) c->htab_sent.clear();
    HIPCHK(c, c->d_segend.reserve(htab.size()));
    HIPCHK(c, c->d_cnt32.reserve(ncnt + (size_t)n_spans));
    HIPCHK(c, c->d_starts.reserve((size_t)n_items + 1));
    HIPCHK(c, c->d_blocks.reserve((size_t)std::min<long long>(n_items, (n_brows + 1) * (max_len / BC + 2) * std::max<long long>(nseg_key, 1))));
    HIPCHK(c, c->d_wgfirst.reserve((size_t)G + 1));
    HIPCHK(c, c->wrec_f64.reserve((size_t)nrec * pup::kWideRec)); HIPCHK(c, c->wrec_num.reserve((size_t)nrec * pup::kWideRec));
    HIPCHK(c, c->wrec_seg.reserve((size_t)nrec));
    if (k32) { HIPCHK(c, c->d_k32.reserve((size_t)n_items)); HIPCHK(c, c->d_k32b.reserve((size_t)n_items)); }
    else     { HIPCHK(c, c->d_keys.reserve((size_t)n_items)); HIPCHK(c, c->d_keys2.reserve((size_t)n_items)); }
    if (htab != c->htab_sent) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(c->d_segend.p, htab.data(), htab.size() * 8, hipMemcpyHostToDevice));
        c->htab_sent = htab;
    }
    BinPlan bp{};
    const bool use_bin = k32 && !(c->variant & 1024) && bin_plan(end_bit, 0, n_items, bp);      // (pup_bin.hpp; else the library's radix sort)
    c->last_prepass = use_bin ? "binning" : "library_sort";
    size_t tmp_bytes = 0;
    hipError_t se = hipSuccess;
    if (!use_bin) {
        se = k32 ? rocprim::radix_sort_pairs<Radix10>(nullptr, tmp_bytes, c->d_k32.p, c->d_k32b.p, c->d_win.p, c->d_win2.p, (size_t)n_items, 0, end_bit, c->stream)
                 : rocprim::radix_sort_pairs<Radix10>(nullptr, tmp_bytes, c->d_keys.p, c->d_keys2.p, c->d_win.p, c->d_win2.p, (size_t)n_items, 0, end_bit, c->stream);
        if (se == hipSuccess) se = c->d_sorttmp.reserve(tmp_bytes + 16);
        if (se != hipSuccess) return fail(c, PUP_EHIP, "pup_accumulate: block sort: %s", hipGetErrorString(se));
    }

    // ---- prepass, all on the stream -------------------------------------------------------------------------------------
    if (ev) HIPCHK(c, hipEventRecord(ev[0], c->stream));
    HIPCHK(c, clear_async(c, c->d_cnt32.p, (ncnt + (size_t)n_spans) * sizeof(unsigned)));
    HIPCHK(c, clear_async(c, c->wrec_seg.p, (size_t)nrec * sizeof(unsigned)));
    if (use_bin) { const int brc = bin_prepare(c, bp, n_items, false); if (brc != PUP_OK) return brc; }
    const unsigned ticket = ++c->ticket;
    const bool ooe_vec = ooe && !c->have_exp_pair && (c->nexp > 1 || c->n_exp_regions > 0);
    bool ooe_clean = false;
    int far_gap = 0x7fffffff;
    if (ooe && !c->have_exp_pair && !c->h_exp.empty()) {
        if (c->nexp == 1) ooe_clean = c->h_exp[0] == c->h_exp[0] && c->h_exp[0] != 0.0;
        else if (ooe_vec) {
            const long long far = c->exp_far(ignore_diags);
            ooe_clean = far > (long long)ignore_diags + W;
            if (ooe_clean) far_gap = (int)std::min<long long>(far, 0x7fffffff);
        }
    }
    const unsigned gk4 = use_bin ? (unsigned)bp.ntiles : (unsigned)((n_items + 1023) / 1024);
    const int wkey_threads = 256, wkey_per = use_bin ? pup::kBinTile / 256 : 4;
    // what staging a block costs, in items' worth of time (the workgroups' block ranges are balanced on items + cost x blocks): an item is
    // 4 x CH LDS reads, a block ~9 k clocks of store burst, barriers and look-ahead whatever the lane shape — measured (tools/ab/wide_cost.sh,
    // K1 ms at cost 100 / 130 / 160 / 200 / 260): 51-bin windows (CH 11) 2.71 / 2.70 / 2.78 / 2.87 / 2.93, 81-bin (CH 7, four groups) 7.67 / 6.98 /
    // 6.47 / 6.34 / 6.58, 201-bin (CH 11, sixteen groups) 22.2 / 20.5 / 20.9 / 21.5 / 22.5 — the fixed 100 of round 4 was right for the
    // shape it was measured on only
    int wide_cost = (pup::kWideBlockCostCells + geo.CH / 2) / geo.CH;
    if (const char* e = getenv("COOLPUPPY_AMD_WIDE_COST")) { const int v = atoi(e); if (v > 0) wide_cost = v; }     // experiments
#define PUP_WKEY_ARGS dr0, dc0, (long long)n, n_items, (const long long*)c->d_segend.p, nseg2t, (const pup::IdxChrom*)c->idx_chrom.p, c->n_chrom, \
        (const unsigned short*)c->bin_chrom.p, (long long)c->nbins, (const int*)c->d_brow.p, W, NG, geo.NGc, geo.SH, geo.SW, BR, BC, sh_br, sh_seg, \
        seg_shift, ignore_diags + W - 1, far_gap, c->band_w
    const size_t wkey_lds = (((size_t)nseg2t + (use_bin ? ((size_t)1 << bp.DH) : 0) + 3 * (size_t)c->n_chrom) * 4 + 15) & ~(size_t)15;
    if (k32) hipLaunchKernelGGL((pup::wide_key_kernel<unsigned>), dim3(gk4), dim3(wkey_threads), wkey_lds, c->stream, PUP_WKEY_ARGS, c->d_k32.p, c->d_win.p, c->d_cnt32.p, use_bin ? c->d_bindesc.p : (unsigned*)nullptr, bp.DL, 1 << bp.DH, wkey_per, rel_bc);
    else hipLaunchKernelGGL((pup::wide_key_kernel<unsigned long long>), dim3(gk4), dim3(wkey_threads), wkey_lds, c->stream, PUP_WKEY_ARGS, c->d_keys.p, c->d_win.p, c->d_cnt32.p, (unsigned*)nullptr, 0, 0, 4, rel_bc);
#undef PUP_WKEY_ARGS
    hipLaunchKernelGGL(pup::staged_publish_kernel, dim3(1), dim3(64), 0, c->stream, (const unsigned*)c->d_cnt32.p, 3,
                       (volatile unsigned*)c->d_flags, ticket);
    HIPCHK(c, hipEventRecord(c->ev_key, c->stream));
    unsigned* d_spans = c->d_cnt32.p + ncnt;
    const unsigned gt = (unsigned)std::min<long long>(4096, (n_items + 255) / 256);
    const pup::ExpRegion* d_eregs = n_eregs > 0 ? c->exp_regions.p : nullptr;
    if (use_bin) {
        const int brc = bin_run(c, bp, n_items, 0, c->d_k32.p, c->d_win.p, c->d_k32b.p, c->d_win2.p, false, c->d_cnt32.p + 3);
        if (brc != PUP_OK) return brc;
        hipLaunchKernelGGL((pup::staged_table_kernel<unsigned>), dim3(gt), dim3(256), 0, c->stream, (const unsigned*)c->d_starts.p,
                           (const unsigned*)(c->d_cnt32.p + 3), n_items, (const unsigned*)nullptr,
                           (const unsigned short*)c->d_win2.p, (const int*)c->d_brow.p, (const pup::IdxChrom*)c->idx_chrom.p,
                           c->n_chrom, geo.SH, geo.SW, RS, RS, NG, wide_cost, sh_br, sh_seg, sh_seg, seg_shift, 0, n_eregs, 0, d_eregs,
                           (const unsigned long long*)c->badbits.p, c->d_blocks.p, c->d_wgfirst.p, G, (const unsigned*)c->d_bkeys.p,
                           (const unsigned short*)nullptr,
                           (volatile unsigned*)(c->d_flags + 4), ticket, rel_bc);
    } else if (k32) {
        se = rocprim::radix_sort_pairs<Radix10>(c->d_sorttmp.p, tmp_bytes, c->d_k32.p, c->d_k32b.p, c->d_win.p, c->d_win2.p,
                                                (size_t)n_items, 0, end_bit, c->stream);
        if (se != hipSuccess) return fail(c, PUP_EHIP, "pup_accumulate: block sort: %s", hipGetErrorString(se));
        hipLaunchKernelGGL((pup::count_heads_kernel<unsigned>), dim3((unsigned)n_spans), dim3(256), 0, c->stream,
                           (const unsigned*)c->d_k32b.p, n_items, 0, d_spans);
        hipLaunchKernelGGL((pup::block_starts_kernel<unsigned>), dim3((unsigned)n_spans), dim3(256), 0, c->stream,
                           (const unsigned*)c->d_k32b.p, n_items, 0, (const unsigned*)d_spans, c->d_starts.p, c->d_cnt32.p + 3);
        hipLaunchKernelGGL((pup::staged_table_kernel<unsigned>), dim3(gt), dim3(256), 0, c->stream, (const unsigned*)c->d_starts.p,
                           (const unsigned*)(c->d_cnt32.p + 3), n_items, (const unsigned*)c->d_k32b.p,
                           (const unsigned short*)c->d_win2.p, (const int*)c->d_brow.p, (const pup::IdxChrom*)c->idx_chrom.p,
                           c->n_chrom, geo.SH, geo.SW, RS, RS, NG, wide_cost, sh_br, sh_seg, sh_seg, seg_shift, 0, n_eregs, 0, d_eregs,
                           (const unsigned long long*)c->badbits.p, c->d_blocks.p, c->d_wgfirst.p, G, (const unsigned*)nullptr, (const unsigned short*)nullptr,
                           (volatile unsigned*)(c->d_flags + 4), ticket, rel_bc);
    } else {
        se = rocprim::radix_sort_pairs<Radix10>(c->d_sorttmp.p, tmp_bytes, c->d_keys.p, c->d_keys2.p, c->d_win.p, c->d_win2.p,
                                                (size_t)n_items, 0, end_bit, c->stream);
        if (se != hipSuccess) return fail(c, PUP_EHIP, "pup_accumulate: block sort: %s", hipGetErrorString(se));
        hipLaunchKernelGGL((pup::count_heads_kernel<unsigned long long>), dim3((unsigned)n_spans), dim3(256), 0, c->stream,
                           (const unsigned long long*)c->d_keys2.p, n_items, 0, d_spans);
        hipLaunchKernelGGL((pup::block_starts_kernel<unsigned long long>), dim3((unsigned)n_spans), dim3(256), 0, c->stream,
                           (const unsigned long long*)c->d_keys2.p, n_items, 0, (const unsigned*)d_spans, c->d_starts.p, c->d_cnt32.p + 3);
        hipLaunchKernelGGL((pup::staged_table_kernel<unsigned long long>), dim3(gt), dim3(256), 0, c->stream, (const unsigned*)c->d_starts.p,
                           (const unsigned*)(c->d_cnt32.p + 3), n_items, (const unsigned long long*)c->d_keys2.p,
                           (const unsigned short*)c->d_win2.p, (const int*)c->d_brow.p, (const pup::IdxChrom*)c->idx_chrom.p,
                           c->n_chrom, geo.SH, geo.SW, RS, RS, NG, wide_cost, sh_br, sh_seg, sh_seg, seg_shift, 0, n_eregs, 0, d_eregs,
                           (const unsigned long long*)c->badbits.p, c->d_blocks.p, c->d_wgfirst.p, G, (const unsigned*)nullptr, (const unsigned short*)nullptr,
                           (volatile unsigned*)(c->d_flags + 4), ticket, rel_bc);
    }
    HIPCHK(c, hipGetLastError());

    // ---- the key kernel's verdict ---------------------------------------------------------------------------------------
    HIPCHK(c, hipEventSynchronize(c->ev_key));
    if (c->h_flags[3] != ticket) return fail(c, PUP_EHIP, "pup_accumulate: the key kernel's verdict did not arrive");
    c->hint_ticket = ticket;
    if (c->h_flags[0] != 0 || c->h_flags[2] != 0) {              // a window outside one chromosome / outside the band: per-window kernels
        c->off_staged(c->h_flags[0] ? 1u : 2u, c->h_flags[0] ? "some windows are not inside one chromosome" :
                      "some windows reach beyond the dense band of counts (pup_build_index: its width is what memory allowed)", (long long)n);
        return 1;
    }
    const bool fact = (!ooe || ooe_clean) && c->h_flags[1] == 0 && !(c->variant & 4);

    pup::K1Args a{};
    fill_k1_args(c, a, ignore_diags, mode);
    pup::WideArgs wa{};
    wa.blocks = c->d_blocks.p; wa.win = c->d_win2.p; wa.wg_first = c->d_wgfirst.p;
    wa.WF = W; wa.NGc = geo.NGc; wa.NG = NG; wa.SH = geo.SH; wa.SW = geo.SW; wa.NPC = geo.NPC;
    wa.rec_seg = c->wrec_seg.p; wa.rec_f64 = c->wrec_f64.p; wa.rec_num = c->wrec_num.p;
    wa.timing = nullptr;
    {   // shares of the four waves of a panel, oldest first (see slice_of); COOLPUPPY_AMD_WIDE_SHARES="a,b,c" (cumulative, of 256) for experiments
        int sh3[3] = {88, 161, 219};               // measured, pad 25: equal shares 3.30 ms, 73/140/202 3.15, 78/148/208 2.92, 80/151/211 2.87, 82/154/213 2.90;
                                                   // round 5 (bookkeeping panel rotating): pad 25 80/151/211 2.71, 88/161/219 2.71, 92/166/222 2.81; pad 100 22.2 / 21.9 / 21.7
        if (const char* e = getenv("COOLPUPPY_AMD_WIDE_SHARES")) { int x, y, z; if (sscanf(e, "%d,%d,%d", &x, &y, &z) == 3 && 0 <= x && x <= y && y <= z && z <= 256) { sh3[0] = x; sh3[1] = y; sh3[2] = z; } }
        wa.share[0] = sh3[0]; wa.share[1] = sh3[1]; wa.share[2] = sh3[2];
    }
    if (c->debug_phases & 4) {
        HIPCHK(c, c->d_timing.reserve((size_t)G * 16 * 8));
        HIPCHK(c, hipMemsetAsync(c->d_timing.p, 0, (size_t)G * 16 * 8 * sizeof(long long), c->stream));
        wa.timing = c->d_timing.p; c->timing_G = G;
    }
    if (ev) HIPCHK(c, hipEventRecord(ev[1], c->stream));
    if (!pup::launch_wide(geo.shape, a, wa, G, ooe, fact, c->stream))
        return fail(c, PUP_ENOTSUP, "pup_accumulate: wide staged kernel not built for lane shape %d", geo.shape);
    HIPCHK(c, hipGetLastError());
    if (ev) HIPCHK(c, hipEventRecord(ev[2], c->stream));
    const size_t W2 = (size_t)W * W, Lf = W2 + 2 * (size_t)W;
    hipLaunchKernelGGL(pup::reduce_wide_kernel, dim3((unsigned)((W2 + 63) / 64), (unsigned)T), dim3(64, pup::kRedParts), 0, c->stream,
                       (const double*)c->wrec_f64.p, (const unsigned*)c->wrec_num.p, (const unsigned*)c->wrec_seg.p, G, W, NG, geo.NGc,
                       geo.SH, geo.SW, flip_from ? 2 : 1, (int)Lf, c->acc_f64.p, c->acc_i64.p,
                       c->acc_i64.p + (size_t)T * W2, (const long long*)c->d_segend.p);
    c->counts_added = true;
    HIPCHK(c, hipGetLastError());
    if (cov_sep) { const int crc = cov_pass(c, dr0, dc0, n, mode); if (crc != PUP_OK) return crc; }
    c->last_staged = true;
    c->last_kernel = fact ? "wide_fact" : "wide";
    return PUP_OK;
}

static int accumulate_impl(pup_ctx* c, const int32_t* r0, const int32_t* c0, const int32_t* hgt, const int32_t* wid,
                           int64_t n, const int64_t* tile_ptr, const int64_t* flip_from, int32_t ignore_diags,
                           uint32_t mode);

static bool is_page_locked(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

static hipError_t upload_snippets(pup_ctx* c, void* dst, const void* src, size_t bytes) {
    if (is_page_locked(src)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return e;
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}

int pup_accumulate(pup_ctx* c, const int32_t* r0, const int32_t* c0, int64_t n, const int64_t* tile_ptr,
                   const int64_t* flip_from, int32_t ignore_diags, uint32_t mode) {
    if (mode & PUP_MODE_LOCAL) return c ? fail(c, PUP_EINVAL, "pup_accumulate: PUP_MODE_LOCAL only applies to rescaled pile-ups") : PUP_EINVAL;
    return accumulate_impl(c, r0, c0, nullptr, nullptr, n, tile_ptr, flip_from, ignore_diags, mode);
}

int pup_accumulate_rescaled(pup_ctx* c, const int32_t* r0, const int32_t* c0, const int32_t* height, const int32_t* width,
                            int64_t n, const int64_t* tile_ptr, const int64_t* flip_from, int32_t ignore_diags,
                            uint32_t mode) {
    if (!c) return PUP_EINVAL;
    if (n > 0 && (!height || !width)) return fail(c, PUP_EINVAL, "pup_accumulate_rescaled: NULL window sizes");
    if (mode & PUP_MODE_DEVPTR) return fail(c, PUP_EINVAL, "pup_accumulate_rescaled: host arrays only");
    return accumulate_impl(c, r0, c0, height, width, n, tile_ptr, flip_from, ignore_diags, mode);
}

static int accumulate_impl(pup_ctx* c, const int32_t* r0, const int32_t* c0, const int32_t* hgt, const int32_t* wid,
                           int64_t n, const int64_t* tile_ptr, const int64_t* flip_from, int32_t ignore_diags,
                           uint32_t mode) {
    if (!c) return PUP_EINVAL;
    const bool rescale = hgt != nullptr;
    if (!c->have_px) return fail(c, PUP_ESTATE, "pup_accumulate: no pixel table loaded");
    if (!c->have_bal) return fail(c, PUP_ESTATE, "pup_accumulate: call pup_load_bins first (weights or NULL for raw)");
    if (c->T <= 0) return fail(c, PUP_ESTATE, "pup_accumulate: call pup_reset first");
    if (n < 0 || !tile_ptr || (n > 0 && (!r0 || !c0)))
        return fail(c, PUP_EINVAL, "pup_accumulate: NULL snippet arrays or negative n");
    if (tile_ptr[0] != 0 || tile_ptr[c->T] != n)
        return fail(c, PUP_EINVAL, "pup_accumulate: tile_ptr must run from 0 to n=%lld (got %lld..%lld)",
                    (long long)n, (long long)tile_ptr[0], (long long)tile_ptr[c->T]);
    for (int t = 0; t < c->T; ++t) {
        if (tile_ptr[t + 1] < tile_ptr[t])
            return fail(c, PUP_EINVAL, "pup_accumulate: tile_ptr decreases at tile %d", t);
        if (flip_from && (flip_from[t] < tile_ptr[t] || flip_from[t] > tile_ptr[t + 1]))
            return fail(c, PUP_EINVAL, "pup_accumulate: flip_from[%d]=%lld outside its tile [%lld, %lld]", t,
                        (long long)flip_from[t], (long long)tile_ptr[t], (long long)tile_ptr[t + 1]);
    }
    const bool m_ooe = mode & PUP_MODE_OOE, m_exp = mode & PUP_MODE_EXPECTED;
    if ((m_ooe || m_exp) && c->nexp == 0 && c->n_exp_regions == 0)
        return fail(c, PUP_ESTATE, "pup_accumulate: OOE/EXPECTED mode without pup_set_expected");
    if (m_ooe && m_exp) return fail(c, PUP_EINVAL, "pup_accumulate: OOE and EXPECTED are exclusive");
    if ((mode & PUP_MODE_COV) && !c->have_cov)
        return fail(c, PUP_ESTATE, "pup_accumulate: COV mode without a coverage vector");
    if (n == 0) return PUP_OK;
    int rc = bind(c); if (rc) return rc;

    const int W = c->W;
    const size_t W2 = (size_t)W * W, Lf = W2 + 2 * (size_t)W;
    // rescaled windows are heavy (10^5 input cells each) and their S^2 output tile lives in LDS: as many workgroups per CU
    // as that allows, ~4 rounds of them, 1024 threads each when only one or two fit (latency hiding comes from waves)
    const size_t rs_tile = (size_t)c->W * c->W * 12 + 16 * (size_t)c->W;
    // (the kernel needs its 128 registers: 16 waves per CU whatever the tile size — one workgroup of 1024 threads per CU; with
    // 512-thread workgroups for small tiles a 51 x 51 output took 15.8 ms where 99 x 99 took 10)
    const int rs_wg_per_cu = 1;
    const int rs_threads = 1024;
    (void)rs_tile;

    // ---- snippets to device ----------------------------------------------------------------------
    const int *dr0, *dc0;
    if (mode & PUP_MODE_DEVPTR) { dr0 = r0; dc0 = c0; }
    else {
        // page-locked source (pup_host_alloc): DMA queued on the context's stream — it starts when the launches that still
        // read the staging buffers are done, and the caller is not held up.  Pageable source: wait for those launches,
        // then a blocking copy (the caller may release such an array as soon as this returns).
        HIPCHK(c, c->d_r0.reserve((size_t)n)); HIPCHK(c, c->d_c0.reserve((size_t)n));
        HIPCHK(c, upload_snippets(c, c->d_r0.p, r0, (size_t)n * sizeof(int)));
        HIPCHK(c, upload_snippets(c, c->d_c0.p, c0, (size_t)n * sizeof(int)));
        dr0 = c->d_r0.p; dc0 = c->d_c0.p;
    }
    if (rescale) {
        HIPCHK(c, c->d_h.reserve((size_t)n)); HIPCHK(c, c->d_w.reserve((size_t)n));
        HIPCHK(c, upload_snippets(c, c->d_h.p, hgt, (size_t)n * sizeof(int)));
        HIPCHK(c, upload_snippets(c, c->d_w.p, wid, (size_t)n * sizeof(int)));
        // (an output tile that does not fit LDS — rescale_size beyond ~115 — lives in global memory: pileup_rescale_kernel, tile_g)
    }

    const int T = c->T;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, ep = nullptr;
    if (c->profiling) {
        HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1)); HIPCHK(c, hipEventCreate(&e2)); HIPCHK(c, hipEventCreate(&ep));
    }
    // ---- many overlapping cis windows: block order + workgroup-staged kernel (K1q), see staged_run --------------------
    bool staged = false;
    c->counts_added = false;
    {
        hipEvent_t evs[3] = {ep, e0, e1};
        const int src = staged_run(c, dr0, dc0, n, tile_ptr, flip_from, ignore_diags, mode, rescale, c->profiling ? evs : nullptr);
        if (src < 0) return src;
        staged = src == PUP_OK;
        if (staged) c->last_kernel = "staged";
        else {
            const int wrc = wide_run(c, dr0, dc0, n, tile_ptr, flip_from, ignore_diags, mode, rescale, c->profiling ? evs : nullptr);
            if (wrc < 0) return wrc;
            staged = wrc == PUP_OK;
        }
    }
    const int *kr0 = dr0, *kc0 = dc0;
    if (!staged) {

    // launch geometry (chunk / group / reduction tables) depends only on the snippet COUNTS per tile, the blocks per
    // segment and the tuning: when it repeats (steady-state loops, benchmarks) the device tables of the last call are reused
    std::vector<long long> gkey;
    gkey.reserve(16 + 4 * (size_t)T);
    gkey.push_back(n); gkey.push_back(T); gkey.push_back(c->W); gkey.push_back(c->chunk_snippets);
    gkey.push_back(c->group_waves); gkey.push_back(c->variant & 2); gkey.push_back((mode & PUP_MODE_EXPECTED) ? 1 : 0);
    gkey.push_back(flip_from ? 1 : 0); gkey.push_back(rescale ? 1 : 0); gkey.push_back((ignore_diags < 0 ? 1 : 0) | (c->variant & 32) | ((c->nexp == 1 || c->have_exp_pair) ? 2 : 0) | ((mode & PUP_MODE_OOE) ? 4 : 0));
    for (int t = 0; t <= T; ++t) gkey.push_back(tile_ptr[t]);
    if (flip_from) for (int t = 0; t < T; ++t) gkey.push_back(flip_from[t]);
    const bool geom_hit = (gkey == c->geom_key);
    if (!geom_hit) {
    c->geom_key.clear();
    // ---- chunk table -------------------------------------------------------------------------------------
    // A chunk = what one wave accumulates: one tile and one flip state; it writes one partial record.
    // Plain chunks come in GROUPS: a group owns a contiguous range of the (position-sorted) snippets and its S chunks
    // interleave over it (chunk j takes range[j], range[j+S], ...), so the waves of a group walk the same few matrix
    // rows together and the rows stay in the XCD's L2.  Groups are dealt round-robin to the 8 XCDs; workgroup b runs on
    // XCD b % 8 (observed dispatch rule, used for speed only), so a group's chunks get ids b = xcd + 8*i.
    long long C = c->chunk_snippets;
    if (C <= 0) {
        const long long target = (long long)std::max(c->n_cu, 64) * 32 * 4;   // ~4 chunks per wave slot
        C = std::max<long long>(16, (n + target - 1) / target);
    }
    if (rescale) C = std::max<long long>(1, (n + (long long)c->n_cu * rs_wg_per_cu * 4 - 1) / ((long long)c->n_cu * rs_wg_per_cu * 4));
    const bool sparse_geom = !((mode & PUP_MODE_EXPECTED) || (c->variant & 2) || rescale) && ignore_diags < 0 && c->W <= 63 &&
                             !(c->variant & 32) && pup::k1s_lds_bytes(c->W) <= (size_t)c->max_lds &&
                             (!(mode & PUP_MODE_OOE) || c->nexp == 1 || c->have_exp_pair);
    if (sparse_geom && c->chunk_snippets <= 0) {
        // a sparse-kernel window is cheap, a chunk is not (zeroing and finishing a W^2 record, one more record for K2)
        // (measured on 4.9e5 51 x 51 windows, K1s + K2 ms: 100 per chunk 0.90, 200: 0.78, 300: 1.01 — fewer chunks than ~10
        // per CU leave the memory system idle, more pay their fixed cost and load the reduction)
        // (round 6, the queued form: 4.9e5 windows, K1s ms at 96 / 120 / 160 / 191 / 250 per chunk: 0.284 / 0.267 / 0.264-0.272 / 0.272-0.295 /
        // 0.276-0.299 — twelve chunks per CU, all resident at the kernel's four waves per SIMD)
        const bool queued = !(c->variant & 4096) && pup::k1sq_lds_bytes(c->W) <= (size_t)c->max_lds;
        const long long slots = (long long)c->n_cu * (queued ? 12 : 10);
        C = std::max<long long>(96, (n + slots - 1) / slots);
    }
    const int S_plain = c->group_waves > 0 ? c->group_waves : 128;
    const int n_xcd = 8;
    // kernel family: register tile (W <= 31), banded register tile (W <= 255), LDS tile (EXPECTED pass, variant&2)
    const bool lds_kernel = (mode & PUP_MODE_EXPECTED) || (c->variant & 2) || rescale;
    // inter-chromosomal windows (no diagonal mask): sparse kernel, O(W) per window
    // (OOE needs ONE expected value per window there: the trans scalar or the region-pair table)
    const bool sparse_kernel = sparse_geom;
    const bool band_kernel = !lds_kernel && !sparse_kernel && c->W > 31;
    // banded kernel: a wave owns 64 / NCH window rows x 16 NCH columns; windows wider than 256 bins take several column
    // panels (a "band" entry of the launch table = row band | column panel << 16)
    const int nrowbands = band_kernel ? (c->W + (pup::kWave / band_nch(c->W)) - 1) / (pup::kWave / band_nch(c->W)) : 1;
    const int ncpanels = band_kernel ? (c->W + 16 * band_nch(c->W) - 1) / (16 * band_nch(c->W)) : 1;
    const int nbands = nrowbands * ncpanels;
    const int U = T;
    std::vector<long long> run_ptr;                                  // chunk ranges of the (tile, flip) runs, [2T + 1]
    std::vector<long long> cb, ce, unit_chunk_ptr((size_t)U + 1, 0), dn((size_t)T);
    std::vector<unsigned char> cf;
    std::vector<int> cs;
    std::vector<std::vector<int>> xcd_list((size_t)n_xcd);       // entries: chunk * nbands + band
    struct Group { long long key; int first_chunk, waves; };
    std::vector<Group> groups;
    const bool host_pos = !(mode & PUP_MODE_DEVPTR) && kr0 == dr0;   // host order == launch order
    std::vector<long long> group_start;                             // DEVPTR: first snippet of every group
    auto add_run = [&](long long b, long long e, unsigned char flip) {          // plain chunks over snippets [b, e)
        for (long long g0 = b; g0 < e; g0 += (long long)S_plain * C) {
            const long long g1 = std::min(e, g0 + (long long)S_plain * C);
            // (chunks of >= 16 windows — except rescaled pile-ups, whose windows are 10^5 cells each: C of them per chunk, ~4 rounds of
            // workgroups per CU; with 16 per chunk 5 000 windows made 320 workgroups on 256 CUs: a quarter of the second round's time was idle CUs)
            const long long per_chunk = rescale ? C : 16;
            const int waves = (int)std::min<long long>(S_plain, std::max<long long>(1, (g1 - g0 + per_chunk - 1) / per_chunk));
            if (!host_pos) group_start.push_back(g0);
            groups.push_back(Group{host_pos ? (long long)r0[g0] : (long long)groups.size(), (int)cb.size(), waves});
            for (int j = 0; j < waves; ++j) {
                cb.push_back(g0 + j); ce.push_back(g1); cs.push_back(waves); cf.push_back(flip);
            }
        }
    };
    for (int t = 0; t < T; ++t) dn[(size_t)t] = tile_ptr[t + 1] - tile_ptr[t];
    for (int u = 0; u < U; ++u) {
        for (int f = 0; f < 2; ++f) {
            const long long b = tile_ptr[u], e = tile_ptr[u + 1], m = flip_from ? flip_from[u] : e;
            run_ptr.push_back((long long)cb.size());
            if (f == 0) add_run(b, m, 0); else add_run(m, e, 1);
        }
        unit_chunk_ptr[(size_t)u + 1] = (long long)cb.size();
    }
    run_ptr.push_back((long long)cb.size());
    const long long nchunks = (long long)cb.size();
    // records: chunk ck writes record ck: tile t's records are contiguous
    const long long nrec = nchunks;
    const std::vector<long long>& tile_rec_ptr = unit_chunk_ptr;
    // Launch order = matrix position, across tiles: with many tiles (by-distance x by-strand ...) every tile walks
    // the whole genome, so running the tiles one after the other re-reads every matrix row once per tile from HBM;
    // dealing the groups out by the row of their first snippet lets the groups that are in flight together — of
    // whatever tile — share rows in L2 / MALL.  (Chunk numbering, hence the reduction, stays tile-contiguous.)
    if (!host_pos && T > 1 && !groups.empty()) {
        // snippets are device-resident: fetch just the first row of every group
        const int ng = (int)groups.size();
        DevBuf<long long> d_pos; DevBuf<int> d_key;
        std::vector<int> keys((size_t)ng);
        hipError_t ge = d_pos.reserve((size_t)ng);
        if (ge == hipSuccess) ge = d_key.reserve((size_t)ng);
        if (ge == hipSuccess) ge = hipMemcpy(d_pos.p, group_start.data(), (size_t)ng * 8, hipMemcpyHostToDevice);
        if (ge == hipSuccess) {
            hipLaunchKernelGGL(pup::gather_int_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, c->stream, kr0,
                               (const long long*)d_pos.p, d_key.p, ng);
            ge = hipStreamSynchronize(c->stream);
        }
        if (ge == hipSuccess) ge = hipMemcpy(keys.data(), d_key.p, (size_t)ng * 4, hipMemcpyDeviceToHost);
        d_pos.release(); d_key.release();
        if (ge != hipSuccess) return fail(c, PUP_EHIP, "pup_accumulate: %s", hipGetErrorString(ge));
        for (int g = 0; g < ng; ++g) groups[(size_t)g].key = keys[(size_t)g];
    }
    if (T > 1)
        std::stable_sort(groups.begin(), groups.end(), [](const Group& x, const Group& y) { return x.key < y.key; });
    {
        size_t gp = 0;
        for (size_t g = 0; g < groups.size(); ++g) {
            auto& lst = xcd_list[gp++ % (size_t)n_xcd];
            for (int j = 0; j < groups[g].waves; ++j)
                for (int b = 0; b < nbands; ++b) lst.push_back((groups[g].first_chunk + j) * nbands + b);
        }
    }
    auto deal = [&](const std::vector<std::vector<int>>& lists, std::vector<int>& bc, std::vector<int>& bb) {
        size_t per_xcd = 0;
        for (auto& l : lists) per_xcd = std::max(per_xcd, l.size());
        bc.assign(per_xcd * (size_t)n_xcd, -1); bb.assign(per_xcd * (size_t)n_xcd, 0);
        for (int x = 0; x < n_xcd; ++x)
            for (size_t i = 0; i < lists[(size_t)x].size(); ++i) {
                const int e = lists[(size_t)x][i];
                bc[i * (size_t)n_xcd + (size_t)x] = e / nbands;
                bb[i * (size_t)n_xcd + (size_t)x] = ((e % nbands) % nrowbands) | (((e % nbands) / nrowbands) << 16);
            }
    };
    std::vector<int> block_chunk, block_band;
    deal(xcd_list, block_chunk, block_band);
    const long long nblocks = (long long)block_chunk.size();
    if (nrec > 0x7fffffffLL || nblocks > 0x7fffffffLL) return fail(c, PUP_ENOTSUP, "pup_accumulate: too many chunks");
    if (C > 0xffffffffLL) return fail(c, PUP_ENOTSUP, "pup_accumulate: chunk too long for 32-bit num partials");

    // two-level reduction plan: records -> slices of <= SL records (within a tile) -> tiles
    const long long SL = 256;
    long long max_per_tile = 0;
    for (int t = 0; t < T; ++t)
        max_per_tile = std::max(max_per_tile, tile_rec_ptr[(size_t)t + 1] - tile_rec_ptr[(size_t)t]);
    const bool two_level = max_per_tile > 2 * SL;
    std::vector<long long> seg1, seg2;   // seg1: slice -> record range; seg2: tile -> slice (or record) range
    if (two_level) {
        seg2.assign((size_t)T + 1, 0);
        seg1.push_back(0);
        for (int t = 0; t < T; ++t) {
            const long long b = tile_rec_ptr[(size_t)t], e = tile_rec_ptr[(size_t)t + 1];
            for (long long k = b; k < e; k += SL) seg1.push_back(std::min(e, k + SL));
            seg2[(size_t)t + 1] = (long long)seg1.size() - 1;
        }
    } else seg2 = tile_rec_ptr;
    const long long nslices = two_level ? (long long)seg1.size() - 1 : 0;

    HIPCHK(c, hipStreamSynchronize(c->stream));   // chunk/segment tables of the previous call are free now
    HIPCHK(c, c->part_f64.reserve((size_t)nrec * Lf)); HIPCHK(c, c->part_num.reserve((size_t)nrec * W2));
    if (two_level) { HIPCHK(c, c->slice_f64.reserve((size_t)nslices * Lf)); HIPCHK(c, c->slice_num.reserve((size_t)nslices * W2)); }
    {
        // pack every table into one host blob (8-byte aligned sections) -> one H2D copy
        std::vector<unsigned char> blob;
        auto put = [&](const void* src, size_t bytes) {
            const size_t off = (blob.size() + 7) & ~(size_t)7;
            blob.resize(off + bytes);
            if (bytes) std::memcpy(blob.data() + off, src, bytes);
            return off;
        };
        const size_t o_cb = put(cb.data(), (size_t)nchunks * 8), o_ce = put(ce.data(), (size_t)nchunks * 8);
        const size_t o_s2 = put(seg2.data(), seg2.size() * 8), o_dn = put(dn.data(), (size_t)T * 8);
        const size_t o_s1 = put(seg1.data(), seg1.size() * 8);
        const size_t o_cs = put(cs.data(), (size_t)nchunks * 4);
        const size_t o_bc = put(block_chunk.data(), (size_t)nblocks * 4), o_bb = put(block_band.data(), (size_t)nblocks * 4);
        const size_t o_cf = put(cf.data(), (size_t)nchunks);
        const size_t o_rp = put(run_ptr.data(), run_ptr.size() * 8);
        HIPCHK(c, c->d_geom.reserve(blob.size() + 8));
        HIPCHK(c, hipMemcpy(c->d_geom.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
        const unsigned char* g = c->d_geom.p;
        c->gv.chunk_begin = reinterpret_cast<const long long*>(g + o_cb);
        c->gv.chunk_end = reinterpret_cast<const long long*>(g + o_ce);
        c->gv.seg2 = reinterpret_cast<const long long*>(g + o_s2);
        c->gv.dn = reinterpret_cast<const long long*>(g + o_dn);
        c->gv.seg1 = reinterpret_cast<const long long*>(g + o_s1);
        c->gv.chunk_stride = reinterpret_cast<const int*>(g + o_cs);
        c->gv.block_chunk = reinterpret_cast<const int*>(g + o_bc);
        c->gv.block_band = reinterpret_cast<const int*>(g + o_bb);
        c->gv.chunk_flip = g + o_cf;
        c->gv.run_ptr = reinterpret_cast<const long long*>(g + o_rp);
    }
    c->g_nchunks = nchunks; c->g_nblocks = nblocks; c->g_two_level = two_level; c->g_nslices = nslices; c->g_max_per_tile = max_per_tile;
    c->geom_key = gkey;
    }   // !geom_hit
    const long long nblocks = c->g_nblocks, nslices = c->g_nslices;
    const bool two_level = c->g_two_level;

    // ---- K1 ---------------------------------------------------------------------------------------
    pup::K1Args a{};
    fill_k1_args(c, a, ignore_diags, mode);
    a.r0 = kr0; a.c0 = kc0;
    a.chunk_begin = c->gv.chunk_begin; a.chunk_end = c->gv.chunk_end; a.chunk_flip = c->gv.chunk_flip;
    a.chunk_stride = c->gv.chunk_stride; a.block_chunk = c->gv.block_chunk; a.block_band = c->gv.block_band;
    const size_t lds = pup::k1_lds_bytes(W);

    if (c->profiling) HIPCHK(c, hipEventRecord(e0, c->stream));
    // small windows: register-tile kernel; wide windows: banded register-tile kernel; EXPECTED-only passes and
    // variant&2: LDS-tile kernel (needs the whole tile in LDS)
    const bool lds_kernel2 = m_exp || (c->variant & 2);
    bool launched = false;
    bool diag_pass = false;
    if (m_exp && !(c->variant & 2) && !rescale) {
        // expected-as-control: O(W) per snippet on the 2W - 1 diagonals of the Toeplitz window, any width (pup_wide.hpp)
        const int ND = 2 * W - 1;
        for (int d0 = 0; d0 < ND; d0 += pup::kDiagPer * pup::kWave)
            hipLaunchKernelGGL(pup::expected_diag_kernel, dim3((unsigned)nblocks), dim3(pup::kWave), 0, c->stream, a, d0,
                               c->part_f64.p, c->part_num.p, ND);
        launched = true; diag_pass = true;
        c->last_kernel = "expected_diag";
    }
    if (rescale) {
        size_t rs_lds = (size_t)W * W * 12 + 16 * (size_t)W;
        const bool tile_global = rs_lds > (size_t)c->max_lds;                 // the S x S tile does not fit LDS: one per workgroup in HBM / L2
        if (tile_global) {
            rs_lds = 16 * (size_t)W;
            HIPCHK(c, c->rs_tile.reserve((size_t)nblocks * ((size_t)W * W + ((size_t)W * W + 1) / 2)));
        }
        // room for the separable zoom's weights: as many per output row and column as LDS has left, up to kRescaleSepK (a window
        // h times the output size needs h + 2; 99 x 99 outputs: 27, small outputs: 64)
        const int sep_k = rescale_sep_k(c, rs_lds, W);
        if (sep_k) rs_lds += 8 + 2 * (size_t)W * ((size_t)sep_k * sizeof(double) + sizeof(int));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pup::pileup_rescale_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs_lds);
        // a slab per workgroup for the gathered window: the largest window of the call, as long as the slabs stay below 8 GB
        long long max_cells = 0;
        for (int64_t i = 0; i < n; ++i) max_cells = std::max(max_cells, (long long)hgt[i] * (long long)wid[i]);
        long long slab_cells = (max_cells + 63) & ~63LL;
        if (slab_cells <= 0 || (unsigned long long)slab_cells * 8ull * (unsigned long long)nblocks > (8ull << 30) ||
            c->rs_scratch.reserve((size_t)slab_cells * (size_t)nblocks) != hipSuccess) { slab_cells = 0; (void)hipGetLastError(); }
        hipLaunchKernelGGL(pup::pileup_rescale_kernel, dim3((unsigned)nblocks), dim3(rs_threads), rs_lds, c->stream, a,
                           (const int*)c->d_h.p, (const int*)c->d_w.p, (double*)nullptr, (double*)nullptr, 0LL,
                           slab_cells ? c->rs_scratch.p : (double*)nullptr, slab_cells, sep_k, tile_global ? c->rs_tile.p : (double*)nullptr);
        launched = true; c->last_kernel = "rescale";
    }
    const bool sparse_launch = !lds_kernel2 && !rescale && ignore_diags < 0 && W <= 63 && !(c->variant & 32) &&
                               pup::k1s_lds_bytes(W) <= (size_t)c->max_lds &&
                               (!(mode & PUP_MODE_OOE) || c->nexp == 1 || c->have_exp_pair);
    if (!launched && sparse_launch) {
        // presence bitmap of the table (K1Args::tbits), built the first time a table is used with the sparse kernel: nbins^2 / 8
        // bytes, taken only when that is at most a quarter of the free memory
        if (c->tbits_state == 0) {
            c->tbits_state = -1;
            // the exact bitmap (a bit per column) when it fits a quarter of the free memory, else the coarsest-needed filter: a bit per
            // 2, 4, ... columns (COOLPUPPY_AMD_TBITS_SHIFT forces a coarseness: tests)
            size_t fb = 0, tb = 0;
            const bool have_mem = !(c->variant & 1) && hipMemGetInfo(&fb, &tb) == hipSuccess;
            // Measured (tools/probe_trans_bins.py, bench.py --config 4): the kernel is as fast or faster with 4 ... 64 columns per bit as
            // with the exact bitmap (0.767 against 0.793 ms: fewer bitmap bytes, and at Hi-C's inter-chromosomal densities a window row
            // with no pixel has none in the 16-column blocks around it either), so 16 columns per bit is where the search starts:
            // 0.7 GB instead of 11.5 GB for a human 10 kb table, and a first call of ~3 ms instead of 22.
            int sh0 = 4;
            // (round 6: overlapping 32-bit filter words, a window of up to 63 bins inside one of them: at least 4 columns per bit)
            if (const char* e = getenv("COOLPUPPY_AMD_TBITS_SHIFT")) sh0 = std::max(2, std::min(16, atoi(e)));
            for (int sh = sh0; have_mem && sh <= 16; ++sh) {
                const unsigned long long w32 = (unsigned long long)((((c->nbins + (1LL << sh) - 1) >> sh) + 15) / 16 + 1) * (unsigned long long)c->nbins;
                const unsigned long long words = (w32 + 1) / 2 + 1;      // 8-byte units of the buffer
                if (words * 8ull > fb / 4 + c->tbits.cap * 8ull) continue;
                if (c->tbits.reserve((size_t)words) != hipSuccess) { (void)hipGetLastError(); break; }
                HIPCHK(c, hipMemsetAsync(c->tbits.p, 0, (size_t)words * 8, c->stream));
                const unsigned gb3 = (unsigned)std::min<long long>((c->nbins + 3) / 4, 1 << 20);
                hipLaunchKernelGGL(pup::tbits_fill_kernel, dim3(gb3), dim3(256), 0, c->stream, c->indptr.p, c->px.p, reinterpret_cast<unsigned*>(c->tbits.p), c->nbins, sh);
                c->tbits_state = 1; c->tbits_shift = sh;
                if (sh > sh0 && !(c->warned & 4u) && !getenv("COOLPUPPY_AMD_QUIET")) {
                    c->warned |= 4u;
                    fprintf(stderr, "[coolpuppy_amd] inter-chromosomal pile-up: the presence bitmap of %lld bins is kept at %d columns per bit (%.1f GB) "
                                    "instead of %d: what fits a quarter of the free device memory\n", c->nbins, 1 << sh, (double)words * 8e-9, 1 << sh0);
                }
                break;
            }
            if (c->tbits_state != 1 && !(c->warned & 4u) && !getenv("COOLPUPPY_AMD_QUIET")) {
                c->warned |= 4u;
                fprintf(stderr, "[coolpuppy_amd] inter-chromosomal pile-up without a presence bitmap of the table (%lld bins): the sparse kernel bisects every "
                                "window row instead (1.5x to 5x slower)\n", c->nbins);
            }
        }
        a.tbits = c->tbits_state == 1 ? reinterpret_cast<const unsigned*>(c->tbits.p) : nullptr; a.tshift = c->tbits_state == 1 ? c->tbits_shift : 0;
        // round 6: per-lane hit queues (pileup_sparse_queue_kernel); tuning bit 21 keeps the first form (same results bit for bit: tests)
        const bool queued = !(c->variant & 4096) && pup::k1sq_lds_bytes(W) <= (size_t)c->max_lds;
        const size_t sl = queued ? pup::k1sq_lds_bytes(W) : pup::k1s_lds_bytes(W);
        auto launch_sparse = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl);
            hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(pup::kWave), sl, c->stream, a);
        };
        if (mode & PUP_MODE_OOE) { if (queued) launch_sparse(pup::pileup_sparse_queue_kernel<true>); else launch_sparse(pup::pileup_sparse_kernel<true>); }
        else { if (queued) launch_sparse(pup::pileup_sparse_queue_kernel<false>); else launch_sparse(pup::pileup_sparse_kernel<false>); }
        launched = true; c->last_kernel = "sparse";
    }
    if (!launched && !lds_kernel2 && W <= 31) { launched = launch_regtile(W, a, (int)nblocks, c->stream); if (launched) c->last_kernel = "regtile"; }
    if (!launched && !lds_kernel2 && W > 31) {
        switch (band_nch(W)) {
            case 4:  launch_k1b<4>(a, (int)nblocks, c->stream); break;
            case 8:  launch_k1b<8>(a, (int)nblocks, c->stream); break;
            default: launch_k1b<16>(a, (int)nblocks, c->stream); break;
        }
        launched = true; c->last_kernel = "band";
    }
    if (!launched) {
        if (pup::k1_lds_bytes(W) > (size_t)c->max_lds)
            return fail(c, PUP_ENOTSUP, "pup_accumulate: window %dx%d needs %zu B of LDS for this pass, device offers %d",
                        W, W, pup::k1_lds_bytes(W), c->max_lds);
        switch (W) {
            case 21: launch_k1<21>(a, (int)nblocks, lds, c->stream); break;
            case 51: launch_k1<51>(a, (int)nblocks, lds, c->stream); break;
            default: launch_k1<0>(a, (int)nblocks, lds, c->stream); break;
        }
        c->last_kernel = "lds_tile";
    }
    HIPCHK(c, hipGetLastError());
    if (c->profiling) HIPCHK(c, hipEventRecord(e1, c->stream));

    if (diag_pass) {
        // records of 2W - 1 diagonals: summed per (tile, flip) run in fixed order, then spread over the W x W accumulators
        const int ND = 2 * W - 1;
        HIPCHK(c, c->slice_f64.reserve((size_t)2 * T * ND)); HIPCHK(c, c->slice_num.reserve((size_t)2 * T * ND));
        hipLaunchKernelGGL((pup::reduce_partials_kernel<unsigned, false>), dim3((unsigned)((2 * ND + 63) / 64), (unsigned)(2 * T)),
                           dim3(64, pup::kRedParts), 0, c->stream, c->part_f64.p, c->part_num.p, c->gv.run_ptr, ND, ND,
                           c->slice_f64.p, c->slice_num.p);
        hipLaunchKernelGGL(pup::expand_diag_kernel, dim3((unsigned)((W2 + 255) / 256), (unsigned)T), dim3(256), 0, c->stream,
                           (const double*)c->slice_f64.p, (const long long*)c->slice_num.p, W, ND, (mode & PUP_MODE_TRANSPOSE) ? 1 : 0,
                           (int)Lf, c->acc_f64.p, c->acc_i64.p);
    } else {
    // ---- K2: partials -> (slices ->) accumulators ---------------------------------------------------
    const int Li = (int)W2;
    const dim3 rb(64, pup::kRedParts), rg1((unsigned)((Lf + Li + 63) / 64), (unsigned)std::max<long long>(nslices, 1));
    const dim3 rg2((unsigned)((Lf + Li + 63) / 64), (unsigned)c->T);
    if (two_level) {
        hipLaunchKernelGGL((pup::reduce_partials_kernel<unsigned, false>), rg1, rb, 0, c->stream,
                           c->part_f64.p, c->part_num.p, c->gv.seg1, (int)Lf, Li, c->slice_f64.p, c->slice_num.p);
        hipLaunchKernelGGL((pup::reduce_partials_kernel<long long, true>), rg2, rb, 0, c->stream,
                           c->slice_f64.p, c->slice_num.p, c->gv.seg2, (int)Lf, Li, c->acc_f64.p, c->acc_i64.p);
    } else if (c->T >= 1024 && c->g_nchunks <= 4LL * c->T && c->g_max_per_tile <= 512) {
        // many tiles with a few records each (by-window): thread per element, records in order
        hipLaunchKernelGGL((pup::reduce_partials_small_kernel<unsigned>), dim3((unsigned)((Lf + Li + 255) / 256), (unsigned)c->T), dim3(256), 0,
                           c->stream, c->part_f64.p, c->part_num.p, c->gv.seg2, (int)Lf, Li, c->acc_f64.p, c->acc_i64.p);
    } else {
        hipLaunchKernelGGL((pup::reduce_partials_kernel<unsigned, true>), rg2, rb, 0, c->stream,
                           c->part_f64.p, c->part_num.p, c->gv.seg2, (int)Lf, Li, c->acc_f64.p, c->acc_i64.p);
    }
    }   // !diag_pass
    hipLaunchKernelGGL(pup::add_counts_kernel, dim3((unsigned)((c->T + 255) / 256)), dim3(256), 0, c->stream,
                       c->acc_i64.p + (size_t)c->T * W2, c->gv.dn, c->T);
    HIPCHK(c, hipGetLastError());
    }   // !staged
    else if (!c->counts_added) {
        // windows per tile straight from the (tile, flip) boundaries the key kernel used (the staged kernels' reductions add them)
        hipLaunchKernelGGL(pup::add_counts_from_ends_kernel, dim3((unsigned)((c->T + 255) / 256)), dim3(256), 0, c->stream,
                           c->acc_i64.p + (size_t)c->T * W2, (const long long*)c->d_segend.p, c->T);
        HIPCHK(c, hipGetLastError());
    }
    if (c->nf_count > 0 && c->have_weight && !rescale && !(mode & PUP_MODE_EXPECTED)) {
        // take the pixels whose balanced value is inf / NaN out of `num` (see collect_nonfinite_kernel); the rescaled
        // path counts from the zoomed values themselves
        HIPCHK(c, c->nf_tp.reserve((size_t)(2 * T + 1)));
        HIPCHK(c, hipMemcpyAsync(c->nf_tp.p, tile_ptr, (size_t)(T + 1) * sizeof(long long), hipMemcpyHostToDevice, c->stream));
        if (flip_from) HIPCHK(c, hipMemcpyAsync(c->nf_tp.p + T + 1, flip_from, (size_t)T * sizeof(long long), hipMemcpyHostToDevice, c->stream));
        pup::K1Args f{};
        fill_k1_args(c, f, ignore_diags, mode);
        f.r0 = dr0; f.c0 = dc0;
        const long long threads = (long long)n * W;
        hipLaunchKernelGGL(pup::nonfinite_fix_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream, f,
                           c->nf_keys.p, c->nf_count, (long long)n, c->nf_tp.p, flip_from ? c->nf_tp.p + T + 1 : nullptr, T,
                           c->acc_i64.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));      // tile_ptr / flip_from are the caller's
    }
    if (c->profiling) {
        HIPCHK(c, hipEventRecord(e2, c->stream));
        if (!staged) { (void)hipEventDestroy(ep); ep = nullptr; }
        c->pending.push_back({e0, e1, e2, ep});
    }
    c->stats.snippets += n;
    return PUP_OK;
}

int pup_stripes(pup_ctx* c, const int32_t* r0, const int32_t* c0, int64_t n, int32_t pad, int32_t ignore_diags,
                uint32_t mode, double* horizontal, double* vertical) {
    if (!c) return PUP_EINVAL;
    if (!c->have_px) return fail(c, PUP_ESTATE, "pup_stripes: no pixel table loaded");
    if (!c->have_bal) return fail(c, PUP_ESTATE, "pup_stripes: call pup_load_bins first (weights or NULL for raw)");
    if (n < 0 || pad < 0 || (n > 0 && (!r0 || !c0 || !horizontal || !vertical)))
        return fail(c, PUP_EINVAL, "pup_stripes: NULL arrays or negative sizes");
    if ((mode & PUP_MODE_OOE) && c->nexp == 0 && c->n_exp_regions == 0)
        return fail(c, PUP_ESTATE, "pup_stripes: OOE mode without expected");
    if (mode & (PUP_MODE_EXPECTED | PUP_MODE_DEVPTR)) return fail(c, PUP_EINVAL, "pup_stripes: unsupported mode bits");
    if (n == 0) return PUP_OK;
    int rc = bind(c); if (rc) return rc;
    const int W = 2 * pad + 1;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    DevBuf<double> d_out;
    HIPCHK(c, c->d_r0.reserve((size_t)n)); HIPCHK(c, c->d_c0.reserve((size_t)n));
    HIPCHK(c, d_out.reserve((size_t)n * W * 2));
    hipError_t e = hipMemcpy(c->d_r0.p, r0, (size_t)n * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_c0.p, c0, (size_t)n * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        pup::K1Args a{};
        a.indptr = c->indptr.p; a.px = c->px.p; a.cntf = c->float_values ? c->cntf.p : nullptr;
        a.weight = c->have_weight ? c->weight.p : nullptr;
        a.expv = (c->nexp > 0 || (c->n_exp_regions > 0 && !c->have_exp_pair)) ? c->expv.p : nullptr;
        a.nexp = c->nexp; a.nbins = c->nbins;
        a.exp_regions = c->n_exp_regions > 0 ? c->exp_regions.p : nullptr; a.n_exp_regions = c->n_exp_regions;
        a.exp_pair = c->have_exp_pair ? c->exp_pair.p : nullptr;
        a.r0 = c->d_r0.p; a.c0 = c->d_c0.p; a.err = c->d_err.p;
        a.nf_pixels = c->nf_count > 0 ? 1 : 0;
        a.W = W; a.ignore_diags = ignore_diags; a.mode = mode;
        const unsigned grid = (unsigned)std::min<int64_t>(n, 65536);
        hipLaunchKernelGGL(pup::stripes_kernel, dim3(grid), dim3(pup::kWave), 0, c->stream, a, (long long)n,
                           d_out.p, d_out.p + (size_t)n * W);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(horizontal, d_out.p, (size_t)n * W * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(vertical, d_out.p + (size_t)n * W, (size_t)n * W * sizeof(double), hipMemcpyDeviceToHost);
    d_out.release();
    if (e != hipSuccess) return fail(c, PUP_EHIP, "pup_stripes: %s", hipGetErrorString(e));
    return check_async_error(c);
}

int pup_extract(pup_ctx* c, const int32_t* r0, const int32_t* c0, const int32_t* height, const int32_t* width,
                int64_t n, int32_t pad, int32_t ignore_diags, uint32_t mode, double* data, double* cov_start,
                double* cov_end) {
    if (!c) return PUP_EINVAL;
    const bool rescale = height != nullptr || width != nullptr;
    if (!c->have_px) return fail(c, PUP_ESTATE, "pup_extract: no pixel table loaded");
    if (!c->have_bal) return fail(c, PUP_ESTATE, "pup_extract: call pup_load_bins first (weights or NULL for raw)");
    if (n < 0 || pad < 0 || (n > 0 && (!r0 || !c0 || !data)) || (rescale && (!height || !width)))
        return fail(c, PUP_EINVAL, "pup_extract: NULL arrays or negative sizes");
    if ((cov_start == nullptr) != (cov_end == nullptr))
        return fail(c, PUP_EINVAL, "pup_extract: cov_start and cov_end must be given together");
    const bool m_ooe = mode & PUP_MODE_OOE, m_exp = mode & PUP_MODE_EXPECTED;
    if ((m_ooe || m_exp) && c->nexp == 0 && c->n_exp_regions == 0)
        return fail(c, PUP_ESTATE, "pup_extract: OOE/EXPECTED mode without expected");
    if (m_ooe && m_exp) return fail(c, PUP_EINVAL, "pup_extract: OOE and EXPECTED are exclusive");
    if (mode & PUP_MODE_DEVPTR) return fail(c, PUP_EINVAL, "pup_extract: DEVPTR is not supported");
    if ((mode & PUP_MODE_COV) && (!c->have_cov || !cov_start))
        return fail(c, PUP_ESTATE, "pup_extract: COV mode needs a coverage vector and cov_start / cov_end outputs");
    if (n == 0) return PUP_OK;
    int rc = bind(c); if (rc) return rc;
    const int W = 2 * pad + 1;
    const size_t W2 = (size_t)W * W;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, c->d_r0.reserve((size_t)n)); HIPCHK(c, c->d_c0.reserve((size_t)n));
    HIPCHK(c, hipMemcpy(c->d_r0.p, r0, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_c0.p, c0, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    if (rescale) {
        HIPCHK(c, c->d_h.reserve((size_t)n)); HIPCHK(c, c->d_w.reserve((size_t)n));
        HIPCHK(c, hipMemcpy(c->d_h.p, height, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_w.p, width, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    }
    DevBuf<double> d_out, d_cov;
    hipError_t e = d_out.reserve((size_t)n * W2);
    if (e == hipSuccess && cov_start) e = d_cov.reserve((size_t)n * 2 * W);
    if (e == hipSuccess) {
        pup::K1Args a{};
        a.indptr = c->indptr.p; a.px = c->px.p; a.cnt32 = c->cnt32.p; a.cntf = c->float_values ? c->cntf.p : nullptr;
        a.bal = c->bal.p; a.badbits = c->badbits.p;
        const bool use_idx = c->have_idx && !(c->variant & 1);
        a.idx = use_idx ? c->idx.p : nullptr; a.idx_chrom = use_idx ? c->idx_chrom.p : nullptr;
        a.n_chrom = use_idx ? c->n_chrom : 0;
    a.rowseg = (use_idx && c->have_rowseg) ? c->rowseg.p : nullptr;
    a.rowabs = (use_idx && c->have_rowseg) ? c->rowabs.p : nullptr;
        a.weight = c->have_weight ? c->weight.p : nullptr;
        a.cov = c->have_cov ? c->cov.p : nullptr;
        a.expv = (c->nexp > 0 || (c->n_exp_regions > 0 && !c->have_exp_pair)) ? c->expv.p : nullptr;
        a.nexp = c->nexp; a.nbins = c->nbins;
        a.exp_regions = c->n_exp_regions > 0 ? c->exp_regions.p : nullptr; a.n_exp_regions = c->n_exp_regions;
        a.exp_pair = c->have_exp_pair ? c->exp_pair.p : nullptr;
        a.r0 = c->d_r0.p; a.c0 = c->d_c0.p; a.err = c->d_err.p; a.counters = c->counters.p;
        a.nf_pixels = c->nf_count > 0 ? 1 : 0;
        a.W = W; a.ignore_diags = ignore_diags; a.mode = mode;
        const unsigned grid = (unsigned)std::min<int64_t>(n, 16384);
        if (rescale) {
            size_t rs_lds = W2 * 12 + 16 * (size_t)W;
            const bool tile_global = rs_lds > (size_t)c->max_lds;
            if (tile_global) {
                rs_lds = 16 * (size_t)W;
                if (c->rs_tile.reserve((size_t)grid * (W2 + (W2 + 1) / 2)) != hipSuccess) return fail(c, PUP_ENOMEM, "pup_extract: tile scratch");
            }
            const int sep_k = rescale_sep_k(c, rs_lds, W);
            if (sep_k) rs_lds += 8 + 2 * (size_t)W * ((size_t)sep_k * sizeof(double) + sizeof(int));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pup::pileup_rescale_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs_lds);
            if (cov_start && !(mode & PUP_MODE_COV)) e = hipMemsetAsync(d_cov.p, 0xff, (size_t)n * 2 * W * 8, c->stream);  // NaN
            long long max_cells = 0;
            for (int64_t i = 0; i < n; ++i) max_cells = std::max(max_cells, (long long)height[i] * (long long)width[i]);
            long long slab_cells = (max_cells + 63) & ~63LL;
            if (slab_cells <= 0 || (unsigned long long)slab_cells * 8ull * grid > (8ull << 30) ||
                c->rs_scratch.reserve((size_t)slab_cells * grid) != hipSuccess) { slab_cells = 0; (void)hipGetLastError(); }
            hipLaunchKernelGGL(pup::pileup_rescale_kernel, dim3(grid), dim3(256), rs_lds, c->stream, a,
                               (const int*)c->d_h.p, (const int*)c->d_w.p, d_out.p, cov_start ? d_cov.p : (double*)nullptr,
                               (long long)n, slab_cells ? c->rs_scratch.p : (double*)nullptr, slab_cells, sep_k, tile_global ? c->rs_tile.p : (double*)nullptr);
        } else {
            hipLaunchKernelGGL(pup::extract_windows_kernel, dim3(grid), dim3(256), 0, c->stream, a, (long long)n, d_out.p,
                               cov_start ? d_cov.p : (double*)nullptr);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(data, d_out.p, (size_t)n * W2 * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && cov_start) {
        // device layout per snippet: {cov_start[W], cov_end[W]}
        e = hipMemcpy2D(cov_start, (size_t)W * 8, d_cov.p, (size_t)2 * W * 8, (size_t)W * 8, (size_t)n, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            e = hipMemcpy2D(cov_end, (size_t)W * 8, d_cov.p + W, (size_t)2 * W * 8, (size_t)W * 8, (size_t)n, hipMemcpyDeviceToHost);
    }
    d_out.release(); d_cov.release();
    if (e != hipSuccess) return fail(c, PUP_EHIP, "pup_extract: %s", hipGetErrorString(e));
    return check_async_error(c);
}

int pup_sync(pup_ctx* c) {
    if (!c) return PUP_EINVAL;
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    collect_events(c);
    return check_async_error(c);
}

int pup_fetch(pup_ctx* c, double* sum, int64_t* num, int64_t* n, double* cov_start, double* cov_end) {
    if (!c) return PUP_EINVAL;
    if (c->T <= 0) return fail(c, PUP_ESTATE, "pup_fetch: call pup_reset first");
    int rc = pup_sync(c); if (rc) return rc;
    const int W = c->W; const size_t W2 = (size_t)W * W, Lf = W2 + 2 * (size_t)W, T = (size_t)c->T;
    // the three parts of a tile's record straight to their arrays (strided device rows -> dense host rows): a by-window pile-up
    // fetches a tile per feature — a zero-filled 140 MB staging vector and a second pass over it were half of its fetch
    const size_t pitch = Lf * sizeof(double);
    if (sum) HIPCHK(c, hipMemcpy2D(sum, W2 * sizeof(double), c->acc_f64.p, pitch, W2 * sizeof(double), T, hipMemcpyDeviceToHost));
    if (cov_start) HIPCHK(c, hipMemcpy2D(cov_start, (size_t)W * sizeof(double), c->acc_f64.p + W2, pitch, (size_t)W * sizeof(double), T, hipMemcpyDeviceToHost));
    if (cov_end) HIPCHK(c, hipMemcpy2D(cov_end, (size_t)W * sizeof(double), c->acc_f64.p + W2 + W, pitch, (size_t)W * sizeof(double), T, hipMemcpyDeviceToHost));
    if (num) HIPCHK(c, hipMemcpy(num, c->acc_i64.p, T * W2 * sizeof(long long), hipMemcpyDeviceToHost));
    if (n) HIPCHK(c, hipMemcpy(n, c->acc_i64.p + T * W2, T * sizeof(long long), hipMemcpyDeviceToHost));
    return PUP_OK;
}

int pup_packed_sizes(pup_ctx* c, int64_t* n_f64, int64_t* n_i64) {
    if (!c) return PUP_EINVAL;
    if (c->T <= 0) return fail(c, PUP_ESTATE, "pup_packed_sizes: call pup_reset first");
    const int64_t W2 = (int64_t)c->W * c->W;
    if (n_f64) *n_f64 = (int64_t)c->T * (W2 + 2 * c->W);
    if (n_i64) *n_i64 = (int64_t)c->T * (W2 + 1);
    return PUP_OK;
}

int pup_export(pup_ctx* c, void* dev_f64, void* dev_i64) {
    if (!c) return PUP_EINVAL;
    if (c->T <= 0) return fail(c, PUP_ESTATE, "pup_export: call pup_reset first");
    if (!dev_f64 || !dev_i64) return fail(c, PUP_EINVAL, "pup_export: NULL destination");
    int rc = bind(c); if (rc) return rc;
    int64_t nf, ni; pup_packed_sizes(c, &nf, &ni);
    // ordered behind the pile-up kernels on the context's stream; ONE synchronisation makes the copies visible to
    // whatever stream the caller's collective runs on
    HIPCHK(c, hipMemcpyAsync(dev_f64, c->acc_f64.p, (size_t)nf * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dev_i64, c->acc_i64.p, (size_t)ni * 8, hipMemcpyDeviceToDevice, c->stream));
    return pup_sync(c);
}

int pup_import(pup_ctx* c, const void* dev_f64, const void* dev_i64) {
    if (!c) return PUP_EINVAL;
    if (c->T <= 0) return fail(c, PUP_ESTATE, "pup_import: call pup_reset first");
    if (!dev_f64 || !dev_i64) return fail(c, PUP_EINVAL, "pup_import: NULL source");
    int rc = bind(c); if (rc) return rc;
    int64_t nf, ni; pup_packed_sizes(c, &nf, &ni);
    // the caller has synchronised its collective; the sources may be released as soon as this returns
    HIPCHK(c, hipMemcpyAsync(c->acc_f64.p, dev_f64, (size_t)nf * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->acc_i64.p, dev_i64, (size_t)ni * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PUP_OK;
}

// ---- native collective: RCCL all-reduce of the packed accumulators, in place, on the context's stream -----------
// librccl is opened on first use (no link-time dependency: single-GPU users never load it).  Only the handful of
// declarations needed are restated here (rccl.h: ncclResult_t 0 = success, ncclInt64 = 4, ncclFloat64 = 8, ncclSum = 0).
namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool tried = false;
};
RcclApi g_rccl;
// ONE ROCm stack per process: the RCCL to use is the one that sits beside the HIP runtime this library is running on
// (torch ships its own libamdhip64 + librccl; the system ROCm has another pair).  Mixing them — a system librccl inside
// a process whose libamdhip64 is torch's — corrupts the heap at exit (GPUTEST_r02: rc 134).  dladdr of a HIP entry point
// names the runtime that is actually mapped; its directory is searched first, the bare sonames only after it.
static std::string rccl_near_hip_runtime() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) || !info.dli_fname) return "";
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash == std::string::npos) return "";
    dir.resize(slash + 1);
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        const std::string cand = dir + name;
        if (FILE* f = std::fopen(cand.c_str(), "rb")) { std::fclose(f); return cand; }
    }
    return "";
}
static bool load_rccl() {
    if (g_rccl.tried) return g_rccl.AllReduce != nullptr;
    g_rccl.tried = true;
    const std::string near = rccl_near_hip_runtime();
    if (!near.empty()) g_rccl.lib = dlopen(near.c_str(), RTLD_NOW | RTLD_GLOBAL);
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        if (g_rccl.lib) break;
        g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!g_rccl.lib) return false;
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(g_rccl.lib, "ncclAllReduce"));
    g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(dlsym(g_rccl.lib, "ncclBroadcast"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(dlsym(g_rccl.lib, "ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(dlsym(g_rccl.lib, "ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.lib, "ncclGetErrorString"));
    if (!g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd) { g_rccl.AllReduce = nullptr; return false; }
    return true;
}
}  // namespace

int pup_rccl_path(char* buf, size_t cap) {
    if (!buf || cap == 0) return PUP_EINVAL;
    std::string path = rccl_near_hip_runtime();
    if (path.empty()) path = "librccl.so.1";             // no copy beside the runtime: the loader's search path decides
    if (path.size() + 1 > cap) return PUP_ERANGE;
    std::memcpy(buf, path.c_str(), path.size() + 1);
    return (int)path.size();
}

int pup_allreduce(pup_ctx* c, void* rccl_comm) {
    if (!c) return PUP_EINVAL;
    if (!rccl_comm) return fail(c, PUP_EINVAL, "pup_allreduce: NULL communicator");
    if (c->T <= 0) return fail(c, PUP_ESTATE, "pup_allreduce: call pup_reset first");
    if (!load_rccl()) return fail(c, PUP_ENOTSUP, "pup_allreduce: librccl.so could not be loaded");
    int rc = bind(c); if (rc) return rc;
    int64_t nf, ni; pup_packed_sizes(c, &nf, &ni);
    auto err = [&](int r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"; };
    int r = g_rccl.GroupStart();
    if (r == 0) r = g_rccl.AllReduce(c->acc_f64.p, c->acc_f64.p, (size_t)nf, /*ncclFloat64*/ 8, /*ncclSum*/ 0, rccl_comm, c->stream);
    if (r == 0) r = g_rccl.AllReduce(c->acc_i64.p, c->acc_i64.p, (size_t)ni, /*ncclInt64*/ 4, /*ncclSum*/ 0, rccl_comm, c->stream);
    const int r2 = g_rccl.GroupEnd();
    if (r != 0 || r2 != 0) return fail(c, PUP_EHIP, "pup_allreduce: %s", err(r != 0 ? r : r2));
    return PUP_OK;                       // asynchronous: ordered on the context's stream like every other call
}

// ---- exchange of SOME tiles (by-window pile-ups: a tile per feature, non-zero on the rank that owns the feature's region) ---------
// The reference merges only what a region produced (coolpuppy/coolpup.py:1511-1531: reduce(sum_pups) over the per-region dicts,
// lib/puputils.py:218-223 puts a snippet under its features' keys).  A flat all-reduce of every accumulator moves T tiles per rank
// whatever they hold — 550 MB for a genome's CTCF sites, nearly all zeros; here a rank sends the tiles it piled windows into, once.
namespace pup {
// block k of `out` = the packed accumulators of tile ids[k]: f64 [W2 sum | W cov_start | W cov_end], i64 [W2 num | n]
static __global__ __launch_bounds__(256) void pack_tiles_kernel(const double* __restrict__ acc_f64, const long long* __restrict__ acc_i64, int T,
                                                         int W2, int Lf, const int* __restrict__ ids, double* __restrict__ out_f64,
                                                         long long* __restrict__ out_i64) {
    const int k = blockIdx.y, t = ids[k], e = blockIdx.x * 256 + threadIdx.x;
    if (e < Lf) out_f64[(size_t)k * Lf + e] = acc_f64[(size_t)t * Lf + e];
    if (e < W2) out_i64[(size_t)k * (W2 + 1) + e] = acc_i64[(size_t)t * W2 + e];
    if (e == W2) out_i64[(size_t)k * (W2 + 1) + W2] = acc_i64[(size_t)T * W2 + t];
}
// mode 0: overwrite, 1: add, 2: clear (the sources are not read)
static __global__ __launch_bounds__(256) void unpack_tiles_kernel(double* __restrict__ acc_f64, long long* __restrict__ acc_i64, int T, int W2, int Lf,
                                                           const int* __restrict__ ids, const double* __restrict__ in_f64,
                                                           const long long* __restrict__ in_i64, int mode) {
    const int k = blockIdx.y, t = ids[k], e = blockIdx.x * 256 + threadIdx.x;
    if (e < Lf) { double& d = acc_f64[(size_t)t * Lf + e]; d = mode == 2 ? 0.0 : (mode == 1 ? d + in_f64[(size_t)k * Lf + e] : in_f64[(size_t)k * Lf + e]); }
    if (e <= W2) {
        long long& d = e < W2 ? acc_i64[(size_t)t * W2 + e] : acc_i64[(size_t)T * W2 + t];
        const long long v = mode == 2 ? 0 : in_i64[(size_t)k * (W2 + 1) + e];
        d = mode == 1 ? d + v : v;
    }
}
}  // namespace pup

// tile numbers to the device (validated: every id in [0, T)); at = offset into xt_ids
static int send_tile_ids(pup_ctx* c, const int32_t* ids, int64_t n, const char* who) {
    for (int64_t i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= c->T) return fail(c, PUP_ERANGE, "%s: tile %d outside [0, %d)", who, (int)ids[i], c->T);
    HIPCHK(c, c->xt_ids.reserve((size_t)std::max<int64_t>(n, 1)));
    if (n > 0) HIPCHK(c, hipMemcpyAsync(c->xt_ids.p, ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // (the caller's array may be pageable)
    return PUP_OK;
}
static void launch_unpack(pup_ctx* c, const int* d_ids, int64_t n, const double* f, const long long* i, int mode) {
    const int W2 = c->W * c->W, Lf = W2 + 2 * c->W;
    for (int64_t at = 0; at < n; at += 65535) {          // (grid.y limit)
        const unsigned ny = (unsigned)std::min<int64_t>(65535, n - at);
        hipLaunchKernelGGL(pup::unpack_tiles_kernel, dim3((unsigned)((Lf + 255) / 256), ny), dim3(256), 0, c->stream, c->acc_f64.p, c->acc_i64.p,
                           c->T, W2, Lf, d_ids + at, f ? f + (size_t)at * Lf : nullptr, i ? i + (size_t)at * (W2 + 1) : nullptr, mode);
    }
}
static void launch_pack(pup_ctx* c, const int* d_ids, int64_t n, double* f, long long* i) {
    const int W2 = c->W * c->W, Lf = W2 + 2 * c->W;
    for (int64_t at = 0; at < n; at += 65535) {
        const unsigned ny = (unsigned)std::min<int64_t>(65535, n - at);
        hipLaunchKernelGGL(pup::pack_tiles_kernel, dim3((unsigned)((Lf + 255) / 256), ny), dim3(256), 0, c->stream, (const double*)c->acc_f64.p,
                           (const long long*)c->acc_i64.p, c->T, W2, Lf, d_ids + at, f + (size_t)at * Lf, i + (size_t)at * (W2 + 1));
    }
}

int pup_pack_tiles(pup_ctx* c, const int32_t* tile_ids, int64_t n, void* dev_f64, void* dev_i64) {
    if (!c) return PUP_EINVAL;
    if (c->T <= 0) return fail(c, PUP_ESTATE, "pup_pack_tiles: call pup_reset first");
    if (n < 0 || (n > 0 && (!tile_ids || !dev_f64 || !dev_i64))) return fail(c, PUP_EINVAL, "pup_pack_tiles: bad arguments");
    int rc = bind(c); if (rc) return rc;
    if (n == 0) return pup_sync(c);
    rc = send_tile_ids(c, tile_ids, n, "pup_pack_tiles"); if (rc) return rc;
    launch_pack(c, c->xt_ids.p, n, static_cast<double*>(dev_f64), static_cast<long long*>(dev_i64));
    HIPCHK(c, hipGetLastError());
    return pup_sync(c);
}

int pup_unpack_tiles(pup_ctx* c, const int32_t* tile_ids, int64_t n, const void* dev_f64, const void* dev_i64, int32_t mode) {
    if (!c) return PUP_EINVAL;
    if (c->T <= 0) return fail(c, PUP_ESTATE, "pup_unpack_tiles: call pup_reset first");
    if (n < 0 || mode < 0 || mode > 2 || (n > 0 && (!tile_ids || (mode != 2 && (!dev_f64 || !dev_i64))))) return fail(c, PUP_EINVAL, "pup_unpack_tiles: bad arguments");
    int rc = bind(c); if (rc) return rc;
    if (n == 0) return PUP_OK;
    rc = send_tile_ids(c, tile_ids, n, "pup_unpack_tiles"); if (rc) return rc;
    launch_unpack(c, c->xt_ids.p, n, static_cast<const double*>(dev_f64), static_cast<const long long*>(dev_i64), mode);
    HIPCHK(c, hipGetLastError());
    return pup_sync(c);                                   // the sources may be released as soon as this returns
}

int pup_allgather_tiles(pup_ctx* c, void* rccl_comm, const int32_t* tile_ids, const int64_t* rank_ptr, int32_t n_ranks, int32_t my_rank) {
    if (!c) return PUP_EINVAL;
    if (!rccl_comm || !rank_ptr || n_ranks < 1 || my_rank < 0 || my_rank >= n_ranks) return fail(c, PUP_EINVAL, "pup_allgather_tiles: bad arguments");
    if (c->T <= 0) return fail(c, PUP_ESTATE, "pup_allgather_tiles: call pup_reset first");
    const int64_t total = rank_ptr[n_ranks];
    if (rank_ptr[0] != 0 || total < 0 || (total > 0 && !tile_ids)) return fail(c, PUP_EINVAL, "pup_allgather_tiles: bad tile lists");
    for (int r = 0; r < n_ranks; ++r) if (rank_ptr[r + 1] < rank_ptr[r]) return fail(c, PUP_EINVAL, "pup_allgather_tiles: bad tile lists");
    if (!load_rccl() || !g_rccl.Broadcast) return fail(c, PUP_ENOTSUP, "pup_allgather_tiles: librccl.so could not be loaded");
    int rc = bind(c); if (rc) return rc;
    if (total == 0) return PUP_OK;
    rc = send_tile_ids(c, tile_ids, total, "pup_allgather_tiles"); if (rc) return rc;
    const size_t W2 = (size_t)c->W * c->W, Lf = W2 + 2 * (size_t)c->W, Li = W2 + 1;
    HIPCHK(c, c->xt_f64.reserve((size_t)total * Lf)); HIPCHK(c, c->xt_i64.reserve((size_t)total * Li));
    const int64_t a = rank_ptr[my_rank], mine = rank_ptr[my_rank + 1] - a;
    // own tiles into this rank's block, then cleared: every listed tile becomes the sum of the blocks that list it, added in
    // RANK order on every rank — the same doubles everywhere, also where regions (hence owners) of a feature overlap
    if (mine > 0) {
        launch_pack(c, c->xt_ids.p + a, mine, c->xt_f64.p + (size_t)a * Lf, c->xt_i64.p + (size_t)a * Li);
        launch_unpack(c, c->xt_ids.p + a, mine, nullptr, nullptr, 2);
    }
    auto err = [&](int r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"; };
    int r = g_rccl.GroupStart();
    for (int k = 0; k < n_ranks && r == 0; ++k) {
        const int64_t b = rank_ptr[k], cnt = rank_ptr[k + 1] - b;
        if (cnt == 0) continue;
        r = g_rccl.Broadcast(c->xt_f64.p + (size_t)b * Lf, c->xt_f64.p + (size_t)b * Lf, (size_t)cnt * Lf, /*ncclFloat64*/ 8, k, rccl_comm, c->stream);
        if (r == 0) r = g_rccl.Broadcast(c->xt_i64.p + (size_t)b * Li, c->xt_i64.p + (size_t)b * Li, (size_t)cnt * Li, /*ncclInt64*/ 4, k, rccl_comm, c->stream);
    }
    const int r2 = g_rccl.GroupEnd();
    if (r != 0 || r2 != 0) return fail(c, PUP_EHIP, "pup_allgather_tiles: %s", err(r != 0 ? r : r2));
    for (int k = 0; k < n_ranks; ++k) {
        const int64_t b = rank_ptr[k], cnt = rank_ptr[k + 1] - b;
        if (cnt > 0) launch_unpack(c, c->xt_ids.p + b, cnt, c->xt_f64.p + (size_t)b * Lf, c->xt_i64.p + (size_t)b * Li, 1);
    }
    HIPCHK(c, hipGetLastError());
    return PUP_OK;                       // asynchronous, like pup_allreduce
}

int pup_set_profiling(pup_ctx* c, int enabled) {
    if (!c) return PUP_EINVAL;
    c->profiling = (enabled & 1) != 0;
    c->count_pixels = (enabled & 1) != 0 && (enabled & 2) == 0;
    return PUP_OK;
}

int pup_get_stats(pup_ctx* c, pup_stats* out) {
    if (!c || !out) return PUP_EINVAL;
    int rc = pup_sync(c); if (rc) return rc;
    unsigned long long h[2] = {0, 0};
    HIPCHK(c, hipMemcpy(h, c->counters.p, sizeof h, hipMemcpyDeviceToHost));
    c->stats.pixels_in_windows = (int64_t)h[0];
    c->stats.probe_loads = (int64_t)h[1];
    c->stats.coverage_ms = c->last_coverage_ms;
    // regions the last staged call piled up from: the count its prepass left in mapped memory (the stream is idle now)
    c->stats.staged_regions = (c->last_staged && c->h_flags && c->h_flags[5] == c->hint_ticket) ? (int64_t)c->h_flags[4] : 0;
    *out = c->stats;
    return PUP_OK;
}

int pup_clear_stats(pup_ctx* c) {
    if (!c) return PUP_EINVAL;
    int rc = pup_sync(c); if (rc) return rc;
    c->stats = pup_stats{};
    HIPCHK(c, hipMemset(c->counters.p, 0, 2 * sizeof(unsigned long long)));
    return PUP_OK;
}

int pup_event_record(pup_ctx* c, int slot) {
    if (!c) return PUP_EINVAL;
    if (slot < 0 || slot >= 8) return fail(c, PUP_EINVAL, "pup_event_record: slot %d out of [0,8)", slot);
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->slots[slot], c->stream));
    return PUP_OK;
}

int pup_debug_timing(pup_ctx* c, int64_t* out, int64_t cap) {
    if (!c || !out) return PUP_EINVAL;
    int rc = pup_sync(c); if (rc) return rc;
    const int64_t n = (int64_t)c->timing_G * 16 * 8;
    if (n == 0 || !c->d_timing.p) return 0;
    if (cap < n) return fail(c, PUP_ERANGE, "pup_debug_timing: needs room for %lld values", (long long)n);
    HIPCHK(c, hipMemcpy(out, c->d_timing.p, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
    return (int)c->timing_G;
}

const char* pup_last_kernel(const pup_ctx* c) { return c ? c->last_kernel : ""; }
const char* pup_last_prepass(const pup_ctx* c) { return (c && c->last_staged) ? c->last_prepass : ""; }

int pup_event_elapsed_ms(pup_ctx* c, int a, int b, float* ms) {
    if (!c || !ms) return PUP_EINVAL;
    if (a < 0 || a >= 8 || b < 0 || b >= 8) return fail(c, PUP_EINVAL, "pup_event_elapsed_ms: bad slot");
    int rc = bind(c); if (rc) return rc;
    HIPCHK(c, hipEventSynchronize(c->slots[b]));
    HIPCHK(c, hipEventElapsedTime(ms, c->slots[a], c->slots[b]));
    return PUP_OK;
}

int pup_set_tuning(pup_ctx* c, int32_t chunk_snippets, int32_t variant) {
    if (!c) return PUP_EINVAL;
    if (chunk_snippets < 0) return fail(c, PUP_EINVAL, "pup_set_tuning: negative chunk size");
    c->chunk_snippets = chunk_snippets;
    c->forget_hints();
    c->variant = (variant & 0xff) | ((variant >> 19) & 0x700);   // bit 27 -> 256: never stage from the dense band; bit 28 -> 512: tile pairs one by one; bit 29 -> 1024: library sort in the prepass
    c->variant |= ((variant >> 20) & 0x7) << 11;                 // bit 20 -> 2048: rescaled windows zoomed sample by sample; bit 21 -> 4096: the sparse trans kernel without hit queues; bit 22 -> 8192: the staged kernel with progressive staging (no barrier between blocks: measured slower, round 6; 21-bin windows only)
    c->group_waves = (variant >> 8) & 0xfff;
    c->debug_phases = ((variant >> 24) & 0x7) | ((variant >> 27) & 0x8);     // (bit 30 -> 8: K1q without its factorised-count bookkeeping, timing only)
    return PUP_OK;
}

}  // extern "C"
#pragma GCC visibility pop
