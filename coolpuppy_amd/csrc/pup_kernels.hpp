// pup_kernels.hpp — CDNA4 (gfx950) kernels of the pile-up engine.
//
// K1  pileup_chunk_kernel   one wavefront (64 lanes) per chunk of same-tile snippets:
//                           per snippet, lanes = window rows run a lower_bound on the row's column
//                           segment (HBM/L2 probes), then lanes = window slots gather the contiguous
//                           {col,count} run of each row, apply weight outer product / diagonal mask /
//                           expected, and add into a per-wave LDS tile (sum f64, num u32, cov f64).
//                           The tile is written out once per chunk as a partial.
//                           Two ways to find a row's pixels: (a) rank-bitmap index (one 64-B block holds
//                           the absolute position of its first pixel + 448 presence bits: one cache line
//                           replaces a ~10-probe binary search, popcounts give every pixel's position),
//                           built once per table for cis windows; (b) binary search, any window.
// K2  reduce_partials_kernel deterministic segmented reduction of partial tiles (fixed order), used
//                           twice (chunks -> slices -> running accumulators).
//
// What the arithmetic restates (reference = open2c/coolpuppy, paths relative to its tree):
//   window extraction, NaN rows/cols, diagonal mask, expected, coverage   coolpuppy/coolpup.py:1104-1157
//   nansum / isfinite / n accumulation                                    coolpuppy/lib/puputils.py:12-41
//   anti-transpose for flipped snippets                                    coolpuppy/coolpup.py:128-131
// No MFMA: this is a gather/reduce bounded by memory latency and bandwidth, not a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pup {

constexpr int kWave = 64;

// ---- rank-bitmap index ---------------------------------------------------------------------------------
// One 64-byte block covers kIdxCols consecutive columns of one row: pos = absolute index (into the pixel
// arrays) of the first pixel of the row at or after the block's first column; bits[w] bit b set <=> the
// row has a pixel at column (block_first_col + 64*w + b).  Row r of chromosome k owns nblk[k] blocks
// covering that chromosome's columns; only cis pixels are indexed.
constexpr int kIdxCols = 448;
struct __attribute__((aligned(64))) IdxBlock {
    unsigned long long pos;
    unsigned long long bits[7];
};
struct IdxChrom {             // per chromosome (device table, sorted by start)
    int       start;          // first global bin
    int       end;            // one past the last global bin
    int       nblk;           // blocks per row
    int       pad_;
    long long blk_base;       // index of the first block of the chromosome's first row
};

struct K1Args {
    // resident tables
    const long long* indptr;   // [nbins+1]
    const int2*      px;       // [nnz] {col, count}
    const int*       cnt32;    // [nnz] counts only (indexed path reads 4 B per pixel)
    const IdxBlock*  idx;      // rank-bitmap index or nullptr
    const IdxChrom*  idx_chrom;
    int              n_chrom;
    const double*    weight;   // [nbins] or nullptr (raw)
    const double*    cov;      // [nbins] or nullptr
    const double*    expv;     // [nexp] or nullptr
    long long        nexp;
    long long        nbins;
    // snippets (device)
    const int*           r0;
    const int*           c0;
    // chunk table (device): a chunk holds snippets of ONE tile and ONE flip state
    const long long* chunk_begin;  // [nchunks]
    const long long* chunk_end;    // [nchunks]
    const unsigned char* chunk_flip;  // [nchunks] non-zero: anti-transpose these snippets (flip_snip_func)
    // per-chunk partial outputs
    double*   part_f64;   // [nchunks][W2 + 2W]   (sum | cov_start | cov_end)
    unsigned* part_num;   // [nchunks][W2]
    // diagnostics
    unsigned long long* counters;  // [0] pixels inside windows, [1] search probes
    int*                err;       // set to 1 when a window leaves the bin table
    int      W;
    int      ignore_diags;         // < 0: no diagonal mask
    unsigned mode;                 // PUP_MODE_* bits
};

// LDS bytes one wave needs for window width W
__host__ __device__ inline size_t k1_lds_bytes(int W) {
    size_t W2 = (size_t)W * W;
    // f64: tile sum W2 | cov_s W | cov_e W | wr W | wc W | ex 2W ; i64: st W | hi W ; u32: num W2
    return 8 * (W2 + 6 * (size_t)W) + 16 * (size_t)W + 4 * W2;
}

__device__ __forceinline__ void lds_add_f64(double* p, double v) {
    // fire-and-forget LDS f64 add (ds_add_f64); cells within one snippet are distinct,
    // the atomic form is used because it is a single no-return instruction.
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// cell of the accumulator a window cell (p, q) lands in: TRANSPOSE first (back to the reference's frame),
// then the reference's flip = anti-transpose (coolpup.py:128-131)
__device__ __forceinline__ int map_cell(int p, int q, int W, bool tr, int fl) {
    int pp = tr ? q : p, qq = tr ? p : q;
    if (fl) { const int tp = W - 1 - qq; qq = W - 1 - pp; pp = tp; }
    return pp * W + qq;
}

// rank-bitmap lookup for one row: position of the first pixel with column >= the window start, and the
// presence bits of the W (<= 64) window columns.  rel_c = window start column relative to the chromosome.
__device__ __forceinline__ void idx_lookup(const IdxBlock* __restrict__ rowblk, int nblk, int rel_c, int W,
                                           long long& pos, unsigned long long& wbits) {
    const int b  = rel_c / kIdxCols;
    const int o  = rel_c - b * kIdxCols;
    const int ws = o >> 6, sh = o & 63;
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(rowblk + b);
    const ulonglong2 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];     // one 64-byte line
    const unsigned long long w[7] = {v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
    unsigned long long rank = 0, cur = 0, nxt = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        rank += (k < ws) ? (unsigned long long)__popcll(w[k]) : 0ull;
        cur = (k == ws) ? w[k] : cur;
        nxt = (k == ws + 1) ? w[k] : nxt;
    }
    if (ws == 6 && sh + W > 64 && b + 1 < nblk) nxt = rowblk[b + 1].bits[0];   // window crosses into the next block
    rank += (unsigned long long)__popcll(cur & ((1ull << sh) - 1ull));
    unsigned long long bits = cur >> sh;
    if (sh) bits |= nxt << (64 - sh);
    wbits = (W < 64) ? (bits & ((1ull << W) - 1ull)) : bits;
    pos = (long long)(v0.x + rank);
}

template <int WT>
__global__ __launch_bounds__(kWave) void pileup_chunk_kernel(K1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int W  = WT ? WT : a.W;
    const int W2 = W * W;
    const int lane = threadIdx.x;

    double*    tsum = reinterpret_cast<double*>(smem_raw);
    double*    covs = tsum + W2;
    double*    cove = covs + W;
    double*    wr   = cove + W;
    double*    wc   = wr + W;
    double*    ex   = wc + W;                       // 2W slots (2W-1 used)
    long long* st   = reinterpret_cast<long long*>(ex + 2 * W);
    long long* hi   = st + W;                       // search path: row end; indexed path: window bits
    unsigned long long* wbv = reinterpret_cast<unsigned long long*>(hi);
    unsigned*  tnum = reinterpret_cast<unsigned*>(hi + W);

    const bool m_ooe   = a.mode & 0x01u;
    const bool m_exp   = a.mode & 0x02u;
    const bool m_cov   = (a.mode & 0x04u) && a.cov != nullptr;
    const bool m_tr    = a.mode & 0x08u;
    const bool use_exp = (m_ooe || m_exp) && a.expv != nullptr && a.nexp > 0;
    const int  igd     = a.ignore_diags;
    const bool have_idx = a.idx != nullptr && W <= 64;

    for (int t = lane; t < W2; t += kWave) { tsum[t] = 0.0; tnum[t] = 0u; }
    for (int t = lane; t < 2 * W; t += kWave) covs[t] = 0.0;   // covs and cove are contiguous
    __syncthreads();

    const long long cb = a.chunk_begin[blockIdx.x];
    const long long ce = a.chunk_end[blockIdx.x];
    const int fl = a.chunk_flip[blockIdx.x];
    unsigned long long npix = 0, nprobe = 0;
    // chromosome of the previous snippet (snippets arrive sorted: the lookup is almost always a hit)
    int ch_start = 0, ch_end = -1, ch_nblk = 0; long long ch_base = 0;

    for (long long s = cb; s < ce; ++s) {
        const int r0s = __builtin_amdgcn_readfirstlane(a.r0[s]);
        const int c0s = __builtin_amdgcn_readfirstlane(a.c0[s]);
        if (r0s < 0 || c0s < 0 || (long long)r0s + W > a.nbins || (long long)c0s + W > a.nbins) {
            if (lane == 0) atomicExch(a.err, 1);
            continue;   // wave-uniform
        }
        // indexed path only when rows and columns of the window lie in ONE chromosome (cis window)
        bool indexed = false;
        if (have_idx && !m_exp) {
            if (!(r0s >= ch_start && r0s < ch_end)) {          // wave-uniform re-lookup
                int lo = 0, hi_k = a.n_chrom;
                while (lo < hi_k) { const int m = (lo + hi_k) >> 1; if (a.idx_chrom[m].end <= r0s) lo = m + 1; else hi_k = m; }
                if (lo < a.n_chrom) {
                    const IdxChrom c = a.idx_chrom[lo];
                    ch_start = c.start; ch_end = c.end; ch_nblk = c.nblk; ch_base = c.blk_base;
                } else { ch_start = 0; ch_end = -1; }
            }
            indexed = r0s >= ch_start && r0s + W <= ch_end && c0s >= ch_start && c0s + W <= ch_end;
        }
        // ---- phase A: locate each row's pixels + per-snippet vectors -----------------------------
        if (!m_exp) {
            for (int p = lane; p < W; p += kWave) {
                const int r = r0s + p;
                if (indexed) {
                    const IdxBlock* rowblk = a.idx + ch_base + (long long)(r - ch_start) * ch_nblk;
                    long long pos; unsigned long long bits;
                    idx_lookup(rowblk, ch_nblk, c0s - ch_start, W, pos, bits);
                    st[p] = pos; wbv[p] = bits;
                } else {
                    long long lo = a.indptr[r];
                    const long long h = a.indptr[r + 1];
                    long long b = h;
                    while (lo < b) {                       // lower_bound(col >= c0s)
                        const long long m = (lo + b) >> 1;
                        if (a.px[m].x < c0s) lo = m + 1; else b = m;
                        ++nprobe;
                    }
                    st[p] = lo; hi[p] = h;
                }
                wr[p] = a.weight ? a.weight[r] : 1.0;
                wc[p] = a.weight ? a.weight[c0s + p] : 1.0;
                if (m_cov) {
                    const double cr = a.cov[r], cc = a.cov[c0s + p];
                    // reference: cov_start follows the snippet's rows, cov_end its columns
                    // (coolpup.py:1151-1153); under TRANSPOSE the kernel's rows are the reference's columns
                    const double vs = m_tr ? cc : cr, ve = m_tr ? cr : cc;
                    if (vs == vs) covs[p] += vs;       // nansum: NaN coverage adds 0
                    if (ve == ve) cove[p] += ve;
                }
            }
        }
        if (use_exp) {
            const int dmin = (c0s - r0s) - (W - 1);
            for (int i = lane; i < 2 * W - 1; i += kWave) {
                long long ad = dmin + i; if (ad < 0) ad = -ad;
                double e;
                if (a.nexp == 1) e = a.expv[0];        // trans scalar
                else e = ad < a.nexp ? a.expv[ad] : __builtin_nan("");
                ex[i] = e;
            }
        }
        __syncthreads();
        // ---- phase B: lanes = window cells --------------------------------------------------------
        for (int t = lane; t < W2; t += kWave) {
            const int p = t / W;
            const int j = t - p * W;
            if (m_exp) {
                // expected-as-control: the Toeplitz window itself, no masks (coolpup.py:1135-1139)
                const double e = use_exp ? ex[j - p + W - 1] : __builtin_nan("");
                const int cell = map_cell(p, j, W, m_tr, fl);
                if (e == e) {
                    lds_add_f64(&tsum[cell], e);
                    if (!__builtin_isinf(e)) atomicAdd(&tnum[cell], 1u);
                }
                continue;
            }
            const double wrp = wr[p];
            const double wcj = wc[j];
            const int dj = (c0s + j) - (r0s + p);
            // validity of cell (p, j): masks only, independent of the pixel table
            bool ok = (wrp == wrp) && (wcj == wcj) && (igd < 0 || dj >= igd);
            double ej = 1.0;
            if (m_ooe) { ej = use_exp ? ex[j - p + W - 1] : __builtin_nan(""); ok = ok && (ej == ej) && (ej != 0.0); }
            const int cell = map_cell(p, j, W, m_tr, fl);
            if (ok) atomicAdd(&tnum[cell], 1u);
            if (indexed) {
                // cell (p, j) holds a pixel iff bit j of the row's window bits is set; its position in the
                // pixel arrays is the row's start + the number of set bits below j
                const unsigned long long bits = wbv[p];
                if ((bits >> j) & 1ull) {
                    ++npix;
                    const long long pos = st[p] + __popcll(bits & ((1ull << j) - 1ull));
                    double val = (double)a.cnt32[pos] * wrp * wcj;
                    bool okv = (val == val) && (igd < 0 || dj >= igd);
                    if (m_ooe) { val = val / ej; okv = okv && (val == val); }
                    if (okv) lds_add_f64(&tsum[cell], val);
                }
            } else {
                // j-th pixel of row p at or after the window start
                const long long pos = st[p] + j;
                if (pos < hi[p]) {
                    const int2 e2 = a.px[pos];
                    const int q = e2.x - c0s;              // >= 0 by lower_bound
                    if (q < W) {
                        ++npix;
                        double val = (double)e2.y * wrp * wc[q];
                        const int d = (c0s + q) - (r0s + p);
                        bool okv = (val == val) && (igd < 0 || d >= igd);
                        if (m_ooe) { val = val / (use_exp ? ex[q - p + W - 1] : __builtin_nan("")); okv = okv && (val == val); }
                        if (okv) lds_add_f64(&tsum[map_cell(p, q, W, m_tr, fl)], val);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- flush the chunk's partial tile ---------------------------------------------------------
    const size_t L = (size_t)W2 + 2 * (size_t)W;
    double*   of = a.part_f64 + (size_t)blockIdx.x * L;
    unsigned* on = a.part_num + (size_t)blockIdx.x * W2;
    for (int t = lane; t < W2; t += kWave) { of[t] = tsum[t]; on[t] = tnum[t]; }
    for (int t = lane; t < 2 * W; t += kWave) of[W2 + t] = covs[t];
    // wave-reduce diagnostics, one atomic per chunk
    for (int off = 32; off > 0; off >>= 1) {
        npix   += __shfl_down(npix, off);
        nprobe += __shfl_down(nprobe, off);
    }
    if (lane == 0 && a.counters) {
        atomicAdd(&a.counters[0], npix);
        atomicAdd(&a.counters[1], nprobe);
    }
}

// ---- K1r: register-tile variant for small windows (W <= 32) ----------------------------------------------
// Lane (p, k) owns the CH = ceil(W / NCH) cells of window row p, columns [k*CH, (k+1)*CH), NCH = 64 / W, for
// EVERY snippet of the chunk: sum (f64) and num (u32) of those cells live in registers, so the hot loop has no
// LDS traffic, no atomics and no barrier.  Per snippet a lane (1) reads ONE 64-byte index block of its row
// (or binary-searches the row when the window is not cis / no index) to get the position of the first pixel
// of its column chunk and the chunk's presence bits, (2) issues one 4-byte count load per set bit, all
// independent, (3) applies weights / masks / expected in registers.  The next snippet's index block and
// weights are requested before the current snippet's arithmetic (software pipelining across snippets).
// The flip / transpose cell mapping is applied once, when the chunk's partial tile is written.
struct RowLoc { long long pos; unsigned bits; };

template <int CHW>
__device__ __forceinline__ RowLoc search_row_chunk(const K1Args& a, int r, int c_first, unsigned long long& nprobe) {
    // binary search for the first pixel with column >= c_first, then presence bits of the CHW columns
    long long lo = a.indptr[r];
    const long long h = a.indptr[r + 1];
    long long b = h;
    while (lo < b) {
        const long long m = (lo + b) >> 1;
        if (a.px[m].x < c_first) lo = m + 1; else b = m;
        ++nprobe;
    }
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < CHW; ++i) {
        if (lo + i < h) {
            const int d = a.px[lo + i].x - c_first;
            if (d < CHW) bits |= 1u << d;
        }
    }
    return {lo, bits};
}

template <int W>
__global__ __launch_bounds__(kWave) void pileup_regtile_kernel(K1Args a) {
    static_assert(W >= 1 && W <= 32, "register-tile kernel serves windows up to 32 bins");
    constexpr int NCH = kWave / W;
    constexpr int CH  = (W + NCH - 1) / NCH;
    constexpr int W2  = W * W;
    __shared__ double cov_lds[2 * W];
    const int lane = threadIdx.x;
    const int p  = lane / NCH;
    const int k  = lane - p * NCH;
    const int q0 = k * CH;
    const bool active = (p < W) && (q0 < W);
    const int chw = active ? ((W - q0) < CH ? (W - q0) : CH) : 0;     // columns this lane really owns

    const bool m_ooe   = a.mode & 0x01u;
    const bool m_cov   = (a.mode & 0x04u) && a.cov != nullptr;
    const bool m_tr    = a.mode & 0x08u;
    const bool use_exp = m_ooe && a.expv != nullptr && a.nexp > 0;
    const int  igd     = a.ignore_diags;
    const bool have_idx = a.idx != nullptr;

    double   sum[CH];
    unsigned num[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { sum[i] = 0.0; num[i] = 0u; }
    if (m_cov) {
        for (int t = lane; t < 2 * W; t += kWave) cov_lds[t] = 0.0;
        __syncthreads();
    }

    const long long cb = a.chunk_begin[blockIdx.x];
    const long long ce = a.chunk_end[blockIdx.x];
    const int fl = a.chunk_flip[blockIdx.x];
    unsigned long long npix = 0, nprobe = 0;
    int ch_start = 0, ch_end = -1, ch_nblk = 0; long long ch_base = 0;

    // ---- software pipeline state: raw index line + weights of the snippet about to be processed ----------
    ulonglong2 nb0 = {0, 0}, nb1 = {0, 0}, nb2 = {0, 0}, nb3 = {0, 0};
    double n_wr = 1.0; double n_wc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) n_wc[i] = 1.0;
    int n_r0 = 0, n_c0 = 0; bool n_valid = false, n_indexed = false; int n_o = 0, n_b = 0;
    const IdxBlock* n_rowblk = nullptr;

    auto issue = [&](long long s) {
        n_valid = false; n_indexed = false;
        if (s >= ce) return;
        n_r0 = __builtin_amdgcn_readfirstlane(a.r0[s]);
        n_c0 = __builtin_amdgcn_readfirstlane(a.c0[s]);
        if (n_r0 < 0 || n_c0 < 0 || (long long)n_r0 + W > a.nbins || (long long)n_c0 + W > a.nbins) {
            if (lane == 0) atomicExch(a.err, 1);
            return;
        }
        n_valid = true;
        if (have_idx) {
            if (!(n_r0 >= ch_start && n_r0 < ch_end)) {
                int lo = 0, hi_k = a.n_chrom;
                while (lo < hi_k) { const int m = (lo + hi_k) >> 1; if (a.idx_chrom[m].end <= n_r0) lo = m + 1; else hi_k = m; }
                if (lo < a.n_chrom) {
                    const IdxChrom c = a.idx_chrom[lo];
                    ch_start = c.start; ch_end = c.end; ch_nblk = c.nblk; ch_base = c.blk_base;
                } else { ch_start = 0; ch_end = -1; }
            }
            n_indexed = n_r0 >= ch_start && n_r0 + W <= ch_end && n_c0 >= ch_start && n_c0 + W <= ch_end;
        }
        if (active) {
            const int r = n_r0 + p;
            if (n_indexed) {
                const int rel = (n_c0 - ch_start) + q0;
                n_b = rel / kIdxCols; n_o = rel - n_b * kIdxCols;
                n_rowblk = a.idx + ch_base + (long long)(r - ch_start) * ch_nblk;
                const ulonglong2* q = reinterpret_cast<const ulonglong2*>(n_rowblk + n_b);
                nb0 = q[0]; nb1 = q[1]; nb2 = q[2]; nb3 = q[3];
            }
            if (a.weight) {
                n_wr = a.weight[r];
#pragma unroll
                for (int i = 0; i < CH; ++i) if (i < chw) n_wc[i] = a.weight[n_c0 + q0 + i];
            }
        }
    };

    issue(cb);
    for (long long s = cb; s < ce; ++s) {
        // ---- take over the prefetched snippet -------------------------------------------------------------
        const bool valid = n_valid, indexed = n_indexed;
        const int r0s = n_r0, c0s = n_c0;
        const double wrp = n_wr;
        double wc[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) wc[i] = n_wc[i];
        RowLoc loc = {0, 0u};
        if (valid && active) {
            if (indexed) {
                const unsigned long long w[7] = {nb0.y, nb1.x, nb1.y, nb2.x, nb2.y, nb3.x, nb3.y};
                const int ws = n_o >> 6, sh = n_o & 63;
                unsigned long long rank = 0, cur = 0, nxt = 0;
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    rank += (i < ws) ? (unsigned long long)__popcll(w[i]) : 0ull;
                    cur = (i == ws) ? w[i] : cur;
                    nxt = (i == ws + 1) ? w[i] : nxt;
                }
                if (ws == 6 && sh + chw > 64 && n_b + 1 < ch_nblk) nxt = n_rowblk[n_b + 1].bits[0];
                rank += (unsigned long long)__popcll(cur & ((1ull << sh) - 1ull));
                unsigned long long bits = cur >> sh;
                if (sh) bits |= nxt << (64 - sh);
                loc.bits = (unsigned)(bits & ((1ull << chw) - 1ull));
                loc.pos = (long long)(nb0.x + rank);
            } else {
                loc = search_row_chunk<CH>(a, r0s + p, c0s + q0, nprobe);
                loc.bits &= (chw < 32) ? ((1u << chw) - 1u) : 0xffffffffu;
            }
        }
        // ---- request the next snippet's index line and weights before doing this one's arithmetic ---------
        issue(s + 1);
        if (!valid) continue;                                        // wave-uniform
        if (m_cov) {
            if (active && k == 0) {
                const double cr = a.cov[r0s + p], cc = a.cov[c0s + p];
                const double vs = m_tr ? cc : cr, ve = m_tr ? cr : cc;
                if (vs == vs) cov_lds[p] += vs;                      // one lane per element: no race
                if (ve == ve) cov_lds[W + p] += ve;
            }
        }
        if (active) {
            // all count loads of this lane are independent: issue them together
            int cnt[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                cnt[i] = 0;
                if ((loc.bits >> i) & 1u) cnt[i] = a.cnt32[loc.pos + __popc(loc.bits & ((1u << i) - 1u))];
            }
            double ev[CH];
            if (m_ooe) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    ev[i] = __builtin_nan("");
                    if (use_exp && i < chw) {
                        long long ad = (long long)(c0s + q0 + i) - (r0s + p); if (ad < 0) ad = -ad;
                        ev[i] = (a.nexp == 1) ? a.expv[0] : (ad < a.nexp ? a.expv[ad] : __builtin_nan(""));
                    }
                }
            }
            npix += (unsigned long long)__popc(loc.bits);
            const bool rowok = (wrp == wrp);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (i < chw) {
                    const int dj = (c0s + q0 + i) - (r0s + p);
                    bool ok = rowok && (wc[i] == wc[i]) && (igd < 0 || dj >= igd);
                    if (m_ooe) ok = ok && (ev[i] == ev[i]) && (ev[i] != 0.0);
                    num[i] += ok ? 1u : 0u;
                    if ((loc.bits >> i) & 1u) {
                        double val = (double)cnt[i] * wrp * wc[i];
                        bool okv = (val == val) && (igd < 0 || dj >= igd);
                        if (m_ooe) { val = val / ev[i]; okv = okv && (val == val); }
                        if (okv) sum[i] += val;
                    }
                }
            }
        }
    }

    // ---- flush: window frame -> accumulator frame (transpose, then anti-transpose when flipped) ----------
    if (m_cov) __syncthreads();
    const size_t L = (size_t)W2 + 2 * (size_t)W;
    double*   of = a.part_f64 + (size_t)blockIdx.x * L;
    unsigned* on = a.part_num + (size_t)blockIdx.x * W2;
    if (active) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (i < chw) {
                const int cell = map_cell(p, q0 + i, W, m_tr, fl);
                of[cell] = sum[i];
                on[cell] = num[i];
            }
        }
    }
    for (int t = lane; t < 2 * W; t += kWave) of[W2 + t] = m_cov ? cov_lds[t] : 0.0;
    for (int off = 32; off > 0; off >>= 1) {
        npix   += __shfl_down(npix, off);
        nprobe += __shfl_down(nprobe, off);
    }
    if (lane == 0 && a.counters) {
        atomicAdd(&a.counters[0], npix);
        atomicAdd(&a.counters[1], nprobe);
    }
}

// ---- index construction (once per pixel table) ---------------------------------------------------------
// one wave per row: set the presence bit of every cis pixel of the row
__global__ __launch_bounds__(256) void index_fill_kernel(const long long* __restrict__ indptr, const int2* __restrict__ px,
                                                         const IdxChrom* __restrict__ chroms, int n_chrom,
                                                         IdxBlock* __restrict__ idx, long long nbins) {
    const int lane = threadIdx.x & 63;
    long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; r < nbins; r += stride) {
        int lo = 0, hi = n_chrom;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (chroms[m].end <= r) lo = m + 1; else hi = m; }
        if (lo >= n_chrom) continue;
        const IdxChrom c = chroms[lo];
        if (r < c.start) continue;
        unsigned long long* rowbits = reinterpret_cast<unsigned long long*>(idx + c.blk_base + (r - c.start) * c.nblk);
        const long long b = indptr[r], e = indptr[r + 1];
        for (long long k = b + lane; k < e; k += 64) {
            const int col = px[k].x;
            if (col >= c.end) continue;              // trans pixel: not indexed
            const int rel = col - c.start;
            const int blk = rel / kIdxCols, o = rel - blk * kIdxCols;
            atomicOr(rowbits + (size_t)blk * 8 + 1 + (o >> 6), 1ull << (o & 63));
        }
    }
}

// one thread per row: absolute position of each block's first pixel (running popcount)
__global__ __launch_bounds__(256) void index_rank_kernel(const long long* __restrict__ indptr,
                                                         const IdxChrom* __restrict__ chroms, int n_chrom,
                                                         IdxBlock* __restrict__ idx, long long nbins) {
    long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nbins) return;
    int lo = 0, hi = n_chrom;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (chroms[m].end <= r) lo = m + 1; else hi = m; }
    if (lo >= n_chrom) return;
    const IdxChrom c = chroms[lo];
    if (r < c.start) return;
    IdxBlock* row = idx + c.blk_base + (r - c.start) * c.nblk;
    unsigned long long pos = (unsigned long long)indptr[r];
    for (int b = 0; b < c.nblk; ++b) {
        row[b].pos = pos;
#pragma unroll
        for (int k = 0; k < 7; ++k) pos += (unsigned long long)__popcll(row[b].bits[k]);
    }
}

// Segmented, order-fixed reduction of partial records.
//   f64 record length Lf, integer record length Li.  Output segment g sums input records
//   [seg_ptr[g], seg_ptr[g+1]).  ACCUM: add into the output instead of overwriting, and route
//   record g to output record out_index[g] (running accumulators, one per tile).
template <typename NumIn, bool ACCUM>
__global__ __launch_bounds__(256) void reduce_partials_kernel(
        const double* __restrict__ in_f64, const NumIn* __restrict__ in_num,
        const long long* __restrict__ seg_ptr, int Lf, int Li,
        double* out_f64, long long* out_num) {
    const int g = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Lf + Li) return;
    const long long b = seg_ptr[g], e = seg_ptr[g + 1];
    if (idx < Lf) {
        double acc = 0.0;
        for (long long c = b; c < e; ++c) acc += in_f64[(size_t)c * Lf + idx];
        double* o = out_f64 + (size_t)g * Lf + idx;
        if (ACCUM) *o += acc; else *o = acc;
    } else {
        const int k = idx - Lf;
        long long acc = 0;
        for (long long c = b; c < e; ++c) acc += (long long)in_num[(size_t)c * Li + k];
        long long* o = out_num + (size_t)g * Li + k;
        if (ACCUM) *o += acc; else *o = acc;
    }
}

// n[t] += dn[t]
__global__ void add_counts_kernel(long long* n, const long long* dn, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) n[t] += dn[t];
}

// interleave bin2/count into {col,count} pairs (upload helper), 64-bit or 32-bit column ids
template <typename ColT>
__global__ void pack_pixels_kernel(const ColT* __restrict__ col, const int* __restrict__ cnt,
                                   int2* __restrict__ out, int* __restrict__ out_cnt, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { const int c = cnt[i]; out[i] = make_int2((int)col[i], c); out_cnt[i] = c; }
}

}  // namespace pup
