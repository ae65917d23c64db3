// pup_kernels.hpp — CDNA4 (gfx950) kernels of the pile-up engine.
//
// K1  pileup_chunk_kernel   one wavefront (64 lanes) per chunk of same-tile snippets:
//                           per snippet, lanes = window rows run a lower_bound on the row's column
//                           segment (HBM/L2 probes), then lanes = window slots gather the contiguous
//                           {col,count} run of each row, apply weight outer product / diagonal mask /
//                           expected, and add into a per-wave LDS tile (sum f64, num u32, cov f64).
//                           The tile is written out once per chunk as a partial.
//                           Two ways to find a row's pixels: (a) rank-bitmap index (one 64-B block holds
//                           the absolute position of its first pixel + 320 presence bits: one cache line
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
// kernels that are not templates are defined in every translation unit that includes this header: the units holding only
// instantiations of the staged kernel (pup_staged_tu.hip) define PUP_KERNEL as `static __global__` and never reference them
#ifndef PUP_KERNEL
#define PUP_KERNEL __global__
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace pup {

constexpr int kWave = 64;

// ---- rank-bitmap index ---------------------------------------------------------------------------------
// One 64-byte block (= one cache line) covers kIdxCols consecutive columns of one row:
//   pos      absolute index (into the pixel arrays) of the row's first pixel at or after the block's first
//            column (< 2^48: its top 16 bits double as cum[0] == 0)
//   cum[w-1] number of pixels in words 0..w-1 of this block (w = 1..4)
//   bits[w]  bit b set <=> the row has a pixel at column block_first_col + 64*w + b
//   next0    copy of the NEXT block's bits[0] (0 for the row's last block): a window that starts in word 4
//            never needs a second line
// A lookup is three loads from one line: pos, cum[word], and the 16 bytes {word, word+1}.
// Row r of chromosome k owns nblk[k] blocks covering that chromosome's columns; only cis pixels are indexed.
constexpr int kIdxCols = 320;
struct __attribute__((aligned(64))) IdxBlock {
    unsigned long long pos;
    unsigned short     cum[4];
    unsigned long long bits[5];
    unsigned long long next0;
};
static_assert(sizeof(IdxBlock) == 64, "index block must be one 64-byte line");
struct IdxChrom {             // per chromosome (device table, sorted by start)
    int       start;          // first global bin
    int       end;            // one past the last global bin
    int       nblk;           // blocks per row
    int       pad_;
    long long blk_base;       // index of the first block of the chromosome's first row
};

// expected for many regions in one call: region = global bins [start, end), by-diagonal vector at expv[off .. off+len)
struct ExpRegion { int start, end; long long off, len; };

// what one snippet divides by: a by-diagonal vector (value = base[|col-row|], NaN beyond len) or one scalar
struct ExpSel {
    const double* base; long long len; double scalar; bool is_scalar;
    __device__ __forceinline__ double at(long long ad) const {
        if (is_scalar) return scalar;
        return ad < len ? base[ad] : __builtin_nan("");
    }
};

struct K1Args {
    // resident tables
    const long long* indptr;   // [nbins+1]
    const int2*      px;       // [nnz] {col, count}
    const int*       cnt32;    // [nnz+64] counts only (LDS-tile kernel, indexed path)
    const double*    cntf;     // [nnz] pixel VALUES as float64 when the cooler's pixels/count is a float column (pup_load_pixel_values),
                               //       else nullptr: the integer counts above are then placeholders and only kernels reading `bal` / this run
    const double*    bal;      // [nnz+64] balanced value of every pixel, count*w[row]*w[col] (0 where a weight is
                               //          NaN; plain count when raw) — what get_data() yields, computed once per
                               //          (table, weight column); read by the register-tile kernel
    const unsigned long long* badbits;  // [nbins/64 + 3] bit b of word k set <=> bin 64k+b has a NaN weight
    const IdxBlock*  idx;      // rank-bitmap index or nullptr
    const IdxChrom*  idx_chrom;
    int              n_chrom;
    const unsigned*  rowseg;   // [nbins][n_chrom+1] or nullptr: offset, from the row's first pixel, of its first pixel
                               //          in chromosome k (entry n_chrom = row length): bounds the binary search
                               //          of a window that the rank-bitmap index does not cover (trans)
    const uint2*     rowabs;   // [n_chrom][nbins] or nullptr: {first, end} of the row's pixels in chromosome k as ABSOLUTE positions in
                               //          the pixel table, chromosome-major (the sparse trans kernel: one load per window row)
    int              tshift;   // columns per bit of tbits = 1 << tshift (>= 2)
    const unsigned* tbits;     // [ceil(ceil(nbins >> tshift) / 16) + 1][nbins] or nullptr: bit j of the 32-bit word [cb][row] set <=> the
                               //          table holds a pixel of row `row` in columns [(16 cb + j) << tshift, (16 cb + j + 1) << tshift):
                               //          the words OVERLAP by half.  COLUMN-block major: the words of consecutive rows for one block
                               //          of columns are contiguous (the sparse trans kernel, see tbits_fill_kernel)
    const double*    weight;   // [nbins] or nullptr (raw)
    const double*    cov;      // [nbins] or nullptr
    const double*    expv;     // [nexp] or nullptr: ONE by-diagonal vector (nexp >= 2) or ONE scalar (nexp == 1) ...
    long long        nexp;
    const ExpRegion* exp_regions;  // ... or, when n_exp_regions > 0, a table of regions: expv holds their vectors
    int              n_exp_regions;
    const double*    exp_pair;     // [n_exp_regions^2] trans: scalar expected of block (region of r0, region of c0), or nullptr
    long long        nbins;
    // snippets (device)
    const int*           r0;
    const int*           c0;
    // chunk table (device): a chunk holds snippets of ONE tile and ONE flip state
    const long long* chunk_begin;  // [nchunks]
    const long long* chunk_end;    // [nchunks]
    const unsigned char* chunk_flip;  // [nchunks] non-zero: anti-transpose these snippets (flip_snip_func)
    const int*       chunk_stride; // [nchunks] the chunk takes snippets begin, begin+stride, ... < end: the waves of
                                   //           one group interleave over a contiguous snippet range so that what an
                                   //           XCD works on at any moment is a few consecutive rows (L2-resident)
    const int*       block_chunk;  // [gridDim.x] chunk executed by each workgroup (-1: none); groups are laid out
                                   //           so that workgroups b, b+8, b+16, ... (one XCD) share a group
    const int*       block_band;   // [gridDim.x] row band of the window the workgroup owns (banded kernel; else 0)
    // per-chunk partial outputs
    double*   part_f64;   // [nrecords][W2 + 2W]   (sum | cov_start | cov_end)
    unsigned* part_num;   // [nrecords][W2]
    // diagnostics
    unsigned long long* counters;  // [0] pixels inside windows, [1] search probes
    int*                err;       // set to 1 when a window leaves the bin table
    long long nnz;                 // pixels in the table (cnt32 / bal are padded by 64 zeros from here on)
    const int* band;               // dense band of counts near the diagonal, band[row * band_w + j] = count(row, row + j), or nullptr
    int      band_w;               // columns of the band (0: none)
    unsigned band_zero;            // index of >= 64 zeros behind the band
    int      nf_pixels;            // != 0: some pixels have a non-finite balanced value although both their weights are numbers
                                   //       (weights of +-inf): per-snippet outputs then multiply count * w[row] * w[col] out
                                   //       themselves — `bal` stores NaN products as 0 (see lookup_bal)
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

// region lookup for the expected table (regions sorted by start, disjoint); -1 when the bin is in no region
__device__ __forceinline__ int find_exp_region(const ExpRegion* __restrict__ regs, int n, int bin) {
    int lo = 0, hi = n;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (regs[m].end <= bin) lo = m + 1; else hi = m; }
    return (lo < n && regs[lo].start <= bin) ? lo : -1;
}

struct ExpCache { int r_idx = -1, r_start = 0, r_end = -1; long long r_off = 0, r_len = 0; int c_idx = -1, c_start = 0, c_end = -1; };

// expected of the snippet at (r0, c0): legacy single vector / scalar, or the region table (cis: the vector of the
// region holding r0; trans: the pair scalar of (region of r0, region of c0)).  All wave-uniform.
__device__ __forceinline__ ExpSel select_expected(const K1Args& a, ExpCache& ec, int r0, int c0) {
    ExpSel e;
    const double qn = __builtin_nan("");
    if (a.n_exp_regions <= 0) {
        e.base = a.expv; e.len = a.nexp; e.is_scalar = (a.nexp == 1);
        e.scalar = (a.nexp == 1 && a.expv) ? a.expv[0] : qn;
        if (!a.expv || a.nexp <= 0) { e.is_scalar = true; e.scalar = qn; }
        return e;
    }
    if (!(r0 >= ec.r_start && r0 < ec.r_end)) {
        ec.r_idx = find_exp_region(a.exp_regions, a.n_exp_regions, r0);
        if (ec.r_idx >= 0) { const ExpRegion g = a.exp_regions[ec.r_idx]; ec.r_start = g.start; ec.r_end = g.end; ec.r_off = g.off; ec.r_len = g.len; }
        else { ec.r_start = 0; ec.r_end = -1; }
    }
    if (a.exp_pair) {
        if (!(c0 >= ec.c_start && c0 < ec.c_end)) {
            ec.c_idx = find_exp_region(a.exp_regions, a.n_exp_regions, c0);
            if (ec.c_idx >= 0) { const ExpRegion g = a.exp_regions[ec.c_idx]; ec.c_start = g.start; ec.c_end = g.end; }
            else { ec.c_start = 0; ec.c_end = -1; }
        }
        e.is_scalar = true; e.base = nullptr; e.len = 0;
        e.scalar = (ec.r_idx >= 0 && ec.r_end > r0 && ec.c_idx >= 0 && ec.c_end > c0)
                       ? a.exp_pair[(long long)ec.r_idx * a.n_exp_regions + ec.c_idx] : qn;
        return e;
    }
    e.is_scalar = false; e.scalar = qn;
    if (ec.r_idx >= 0 && r0 < ec.r_end) { e.base = a.expv + ec.r_off; e.len = ec.r_len; }
    else { e.base = a.expv; e.len = 0; }
    return e;
}

// cell of the accumulator a window cell (p, q) lands in: TRANSPOSE first (back to the reference's frame),
// then the reference's flip = anti-transpose (coolpup.py:128-131)
__device__ __forceinline__ int map_cell(int p, int q, int W, bool tr, int fl) {
    int pp = tr ? q : p, qq = tr ? p : q;
    if (fl) { const int tp = W - 1 - qq; qq = W - 1 - pp; pp = tp; }
    return pp * W + qq;
}

// rank-bitmap lookup: position of the row's first pixel with column >= the window start and the presence
// bits of the next 64 columns.  rel_c = window start column relative to the chromosome.
__device__ __forceinline__ void idx_lookup64(const IdxBlock* __restrict__ rowblk, int rel_c,
                                             long long& pos, unsigned long long& bits) {
    const int b  = rel_c / kIdxCols;
    const int o  = rel_c - b * kIdxCols;
    const int ws = o >> 6, sh = o & 63;
    const char* base = reinterpret_cast<const char*>(rowblk + b);
    const unsigned long long p0 = *reinterpret_cast<const unsigned long long*>(base);
    const unsigned cum = *reinterpret_cast<const unsigned short*>(base + 6 + 2 * ws);   // ws == 0 reads pos[63:48] == 0
    unsigned long long cur, nxt;
    {   // {bits[ws], bits[ws+1] or next0}: 16 bytes at an 8-byte aligned address of the same line
        const unsigned long long* w = reinterpret_cast<const unsigned long long*>(base + 16 + 8 * ws);
        cur = w[0]; nxt = w[1];
    }
    const unsigned long long below = cur & ((1ull << sh) - 1ull);
    bits = cur >> sh;
    if (sh) bits |= nxt << (64 - sh);
    pos = (long long)(p0 + cum + (unsigned long long)__popcll(below));
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
    const bool use_exp = (m_ooe || m_exp) && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const int  igd     = a.ignore_diags;
    const bool have_idx = a.idx != nullptr && W <= 64;
    ExpCache ecache;

    const int ck = a.block_chunk[blockIdx.x];
    if (ck < 0) return;                                          // padding workgroup (wave-uniform)
    for (int t = lane; t < W2; t += kWave) { tsum[t] = 0.0; tnum[t] = 0u; }
    for (int t = lane; t < 2 * W; t += kWave) covs[t] = 0.0;   // covs and cove are contiguous
    __syncthreads();

    const long long cb = a.chunk_begin[ck];
    const long long ce = a.chunk_end[ck];
    const long long cstep = a.chunk_stride[ck];
    const int fl = a.chunk_flip[ck];
    unsigned long long npix = 0, nprobe = 0;
    // chromosome of the previous snippet (snippets arrive sorted: the lookup is almost always a hit)
    int ch_start = 0, ch_end = -1, ch_nblk = 0; long long ch_base = 0;

    for (long long s = cb; s < ce; s += cstep) {
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
                    idx_lookup64(rowblk, c0s - ch_start, pos, bits);
                    st[p] = pos; wbv[p] = (W < 64) ? (bits & ((1ull << W) - 1ull)) : bits;
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
            const ExpSel es = select_expected(a, ecache, r0s, c0s);
            const int dmin = (c0s - r0s) - (W - 1);
            for (int i = lane; i < 2 * W - 1; i += kWave) {
                long long ad = dmin + i; if (ad < 0) ad = -ad;
                ex[i] = es.at(ad);
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
                    double val = (a.cntf ? a.cntf[pos] : (double)a.cnt32[pos]) * wrp * wcj;
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
                        double val = (a.cntf ? a.cntf[pos] : (double)e2.y) * wrp * wc[q];
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
    double*   of = a.part_f64 + (size_t)ck * L;
    unsigned* on = a.part_num + (size_t)ck * W2;
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

struct RowLoc { long long pos; unsigned bits; };
// 16-byte loads at 8-byte aligned addresses (global_load_dwordx4 needs dword alignment only)
struct __attribute__((packed, aligned(8))) U64x2 { unsigned long long a, b; };
struct __attribute__((packed, aligned(8))) F64x2 { double a, b; };

// binary search for a row's first pixel with column >= c_first, then the presence bits of CHW columns.
// seg (nullable): the row's {first, one-past-last} pixel offsets of the chromosome holding c_first.
template <int CHW>
__device__ __forceinline__ RowLoc search_row_chunk(const K1Args& a, int r, int c_first, unsigned long long& nprobe,
                                                   const unsigned* __restrict__ seg = nullptr) {
    const long long base = a.indptr[r];
    const long long h = a.indptr[r + 1];
    long long lo = base, b = h, hend = h;
    if (seg) { lo = base + seg[0]; b = hend = base + seg[1]; }     // nothing past the chromosome's segment can match
    while (lo < b) {
        const long long m = (lo + b) >> 1;
        if (a.px[m].x < c_first) lo = m + 1; else b = m;
        ++nprobe;
    }
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < CHW; ++i) {
        if (lo + i < hend) {
            const int d = a.px[lo + i].x - c_first;
            if (d < CHW) bits |= 1u << d;
        }
    }
    return {lo, bits};
}

// chromosome (index into idx_chrom) holding bin c, cached across snippets; wave-uniform
struct ChromOf { int start = 0, end = -1, k = 0; };
__device__ __forceinline__ int chrom_of(const K1Args& a, ChromOf& cc, int c) {
    if (!(c >= cc.start && c < cc.end)) {
        int lo = 0, hi = a.n_chrom;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (a.idx_chrom[m].end <= c) lo = m + 1; else hi = m; }
        if (lo >= a.n_chrom) lo = a.n_chrom - 1;
        cc.k = lo; cc.start = a.idx_chrom[lo].start; cc.end = a.idx_chrom[lo].end;
    }
    return cc.k;
}

// one thread per (row, chromosome boundary): rowseg[row][k] = pixels of the row with column < start of chromosome k
PUP_KERNEL __launch_bounds__(256) void rowseg_kernel(const long long* __restrict__ indptr, const int2* __restrict__ px,
                                                     const IdxChrom* __restrict__ chroms, int n_chrom,
                                                     unsigned* __restrict__ rowseg, long long nbins) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = n_chrom + 1;
    if (t >= nbins * stride) return;
    const long long r = t / stride;
    const int k = (int)(t - r * stride);
    const long long base = indptr[r], end = indptr[r + 1];
    if (k == n_chrom) { rowseg[t] = (unsigned)(end - base); return; }
    const int c_first = chroms[k].start;
    long long lo = base, hi = end;
    while (lo < hi) { const long long m = (lo + hi) >> 1; if (px[m].x < c_first) lo = m + 1; else hi = m; }
    rowseg[t] = (unsigned)(lo - base);
}

// one thread per (chromosome, row): rowabs[k][row] = absolute positions {first pixel of the row in chromosome k, one past its
// last}.  Chromosome-major: the lanes of a wave look up consecutive rows of one chromosome's column range, so their entries
// share cache lines; absolute: the row's offset (indptr) is not needed on top
PUP_KERNEL __launch_bounds__(256) void rowabs_kernel(const long long* __restrict__ indptr, const int2* __restrict__ px,
                                                     const IdxChrom* __restrict__ chroms, int n_chrom,
                                                     uint2* __restrict__ rowabs, long long nbins) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbins * n_chrom) return;
    const int k = (int)(t / nbins);
    const long long r = t - (long long)k * nbins;
    const long long base = indptr[r], end = indptr[r + 1];
    auto first_at = [&](int col) {
        long long lo = base, hi = end;
        while (lo < hi) { const long long m = (lo + hi) >> 1; if (px[m].x < col) lo = m + 1; else hi = m; }
        return (unsigned)lo;
    };
    rowabs[t] = make_uint2(first_at(chroms[k].start), k + 1 < n_chrom ? first_at(chroms[k + 1].start) : (unsigned)end);
}

// Presence FILTER of the whole table for the sparse trans kernel: bit j of the 32-bit word tbits[cb][row] says that row `row` holds a
// pixel in columns [(16 cb + j) << tshift, (16 cb + j + 1) << tshift).  Without it every (window, row) pair of a trans pile-up
// fetched a cache line of the pixel table — 2.5e7 random lines, 3 GB, per 4.9e5 windows of 51 x 51 — to find, 97 times out of 100,
// nothing; with it the W consecutive words of a window's rows are one coalesced load, and only the rows whose bits meet the window's
// columns look the table up.  History: round 3 an exact bitmap of 64-bit words (nbins^2 / 8 bytes: 11.5 GB for a human 10 kb table);
// round 4 a bit per 2^tshift columns (a set bit only SENDS the row to the real lookup, and at trans densities a handful of extra
// columns changes the hit rate from 0.36 % to 0.4 %: 0.7 GB at 16 columns per bit; tables whose filter does not fit get a coarser one).
// Round 6: OVERLAPPING 32-bit words — word cb covers the filter columns [16 cb, 16 cb + 32), so the window whose first filter column
// is 16 cb + i, i < 16, lies inside word cb (tshift >= 2: a window of up to 63 bins spans at most 17 filter columns): ONE 4-byte load
// per window row, never a second word for windows across a boundary, one register per window in flight, 4 W bytes = 2-3 lines (the
// 8-byte words: 4.5 lines + the second load).  Twice the bits of a plain bitmap: 1.4 GB for a human 10 kb table at 16 columns per bit.
// Built on the first call that uses the sparse kernel (pup_engine.hip), one wave per matrix row: a row's pixels are sorted by column,
// so the lanes of a batch of 64 pixels that fall into one column block are neighbours — their bits are ORed along the run (six shuffle
// steps) and the run's last lane sends TWO atomics.  An atomic per pixel serialised on the few words a row's thousand cis pixels
// share: 24 ms per 3.8e8-pixel table (44 with the overlap); now 4.
PUP_KERNEL __launch_bounds__(256) void tbits_fill_kernel(const long long* __restrict__ indptr, const int2* __restrict__ px,
                                                         unsigned* __restrict__ tbits, long long nbins, int tshift) {
    const int lane = threadIdx.x & 63;
    long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; r < nbins; r += stride) {
        const long long b = indptr[r], e = indptr[r + 1];
        for (long long k0 = b; k0 < e; k0 += 64) {
            const long long k = k0 + lane;
            const bool live = k < e;
            const int cc = live ? (px[k].x >> tshift) : -1;
            int cb = live ? (cc >> 4) : -2 - lane;                  // (idle lanes: blocks of their own, never written)
            unsigned bits = live ? (1u << (cc & 15)) : 0u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned ob = __shfl_up(bits, d);
                const int ocb = __shfl_up(cb, d);
                if (lane >= d && ocb == cb) bits |= ob;
            }
            const int ncb = __shfl_down(cb, 1);
            if (live && (lane == 63 || ncb != cb)) {                // the run's last lane holds the OR of the whole run
                atomicOr(&tbits[(long long)cb * nbins + r], bits);
                if (cb > 0) atomicOr(&tbits[(long long)(cb - 1) * nbins + r], bits << 16);
            }
        }
    }
}
// the word of a window's row: index of its column block
__device__ __forceinline__ long long tbits_block(int c0, int tshift) { return (long long)((c0 >> tshift) >> 4); }
// filter bits of a window's columns from its row's word: bit i <=> filter column (c0 >> tshift) + i may hold a pixel
__device__ __forceinline__ unsigned tbits_window(unsigned w, int c0, int W, int tshift) {
    const int cc0 = c0 >> tshift, span = ((c0 + W - 1) >> tshift) - cc0 + 1;      // span <= 17 for W <= 63, tshift >= 2
    return (w >> (cc0 & 15)) & ((1u << span) - 1u);
}

// ---- K1r: register-tile variant for small windows (W <= 32) ----------------------------------------------
// Lane (p, k) owns the CH = ceil(W / NCH) cells of window row p, columns [k*CH, (k+1)*CH), NCH = 64 / W, for
// EVERY snippet of the chunk: sum (f64) and num (u32) of those cells live in registers, so the hot loop has no
// LDS traffic, no atomics and no barrier.  Per snippet a lane
//   (1) reads ONE 64-byte index line of its row (or binary-searches the row when the window is not cis / there
//       is no index) -> position of the first pixel of its column chunk + the chunk's presence bits,
//   (2) reads the masked-bin bits of its row and of its columns (one word + one word pair),
//   (3) issues one 8-byte load per owned cell of the pre-balanced pixel value (count*w_row*w_col, exactly the
//       product the reference forms once per region in get_data), all independent and unconditional,
//   (4) adds with selects: the loop body is straight-line, no per-lane branches.
// The next snippet's index line and mask words are requested before the current snippet's arithmetic; the loop
// is unrolled x2 over two register sets so the pipeline needs no register copies.  The flip / transpose cell
// mapping is applied once, at the chunk's flush.
struct RtStage {
    unsigned long long p0, cums, cur, nxt;   // raw index words: position, packed cum[4], {word, word+1}
    unsigned           rowbad, colbad; // wave-uniform: masked-bin bits of the window's W rows / W columns
    int                ws, sh;         // word and bit offset of the lane's chunk inside the index line
    long long          spos;           // search path: position / bits already resolved
    unsigned           sbits;
    int                r0, c0;         // wave-uniform
    bool               valid, indexed; // wave-uniform
};

// occupancy target (waves per SIMD) the register allocator is held to: measured on W=21, 5 waves (<= 96 VGPRs,
// 8 B/lane of scratch) beats 4 (100 VGPRs) by 8 % and 6 (68 B/lane spilled) loses 40 %
constexpr int rt_min_waves(int W) { return ((W + (kWave / W) - 1) / (kWave / W)) <= 8 ? 5 : 2; }

template <int W, bool OOE>
__global__ __launch_bounds__(kWave, rt_min_waves(W)) void pileup_regtile_kernel(K1Args a) {
    static_assert(W >= 1 && W <= 32, "register-tile kernel serves windows up to 32 bins");
    constexpr int NCH = kWave / W;
    constexpr int CH  = (W + NCH - 1) / NCH;
    constexpr int W2  = W * W;
    __shared__ double cov_lds[2 * W];
    const int lane = threadIdx.x;
    const int p_raw = lane / NCH;
    const int k  = lane - p_raw * NCH;
    const int q0 = k * CH;
    const bool lane_ok = (p_raw < W) && (q0 < W);
    const int p  = p_raw < W ? p_raw : W - 1;                         // idle lanes shadow the last row, add nothing
    const int chw = lane_ok ? ((W - q0) < CH ? (W - q0) : CH) : 0;    // cells this lane really owns
    const unsigned chmask = chw >= 32 ? 0xffffffffu : ((1u << chw) - 1u);
    const int qs = q0 < W ? q0 : 0;                                   // column chunk actually addressed

    const bool m_cov   = (a.mode & 0x04u) && a.cov != nullptr;
    const bool m_tr    = a.mode & 0x08u;
    const bool use_exp = OOE && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const int  igd     = a.ignore_diags;
    const bool have_idx = a.idx != nullptr;
    const double qnan = __builtin_nan("");
    ExpCache ecache;
    ChromOf colchrom;

    const int ck = a.block_chunk[blockIdx.x];
    if (ck < 0) return;                                               // padding workgroup (wave-uniform)
    double   sum[CH];
    unsigned num[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { sum[i] = 0.0; num[i] = 0u; }
    if (m_cov) {
        for (int t = lane; t < 2 * W; t += kWave) cov_lds[t] = 0.0;
        __syncthreads();
    }

    const long long cb = a.chunk_begin[ck];
    const long long ce = a.chunk_end[ck];
    const long long cstep = a.chunk_stride[ck];
    const int fl = a.chunk_flip[ck];
    unsigned long long npix = 0, nprobe = 0;
    int ch_start = 0, ch_end = -1, ch_nblk = 0; long long ch_base = 0;

    // request everything snippet s needs that does not depend on other loads
    auto issue = [&](RtStage& g, long long s) __attribute__((always_inline)) {
        g.valid = false; g.indexed = false;
        if (s >= ce) return;
        g.r0 = __builtin_amdgcn_readfirstlane(a.r0[s]);
        g.c0 = __builtin_amdgcn_readfirstlane(a.c0[s]);
        if (g.r0 < 0 || g.c0 < 0 || (long long)g.r0 + W > a.nbins || (long long)g.c0 + W > a.nbins) {
            if (lane == 0) atomicExch(a.err, 1);
            return;
        }
        g.valid = true;
        if (have_idx) {
            if (!(g.r0 >= ch_start && g.r0 < ch_end)) {               // snippets arrive sorted: rare
                int lo = 0, hi_k = a.n_chrom;
                while (lo < hi_k) { const int m = (lo + hi_k) >> 1; if (a.idx_chrom[m].end <= g.r0) lo = m + 1; else hi_k = m; }
                if (lo < a.n_chrom) {
                    const IdxChrom c = a.idx_chrom[lo];
                    ch_start = c.start; ch_end = c.end; ch_nblk = c.nblk; ch_base = c.blk_base;
                } else { ch_start = 0; ch_end = -1; }
            }
            g.indexed = g.r0 >= ch_start && g.r0 + W <= ch_end && g.c0 >= ch_start && g.c0 + W <= ch_end;
        }
        const int r = g.r0 + p;
        if (g.indexed) {
            const int rel = (g.c0 - ch_start) + qs;
            const int b = rel / kIdxCols, o = rel - b * kIdxCols;
            g.ws = o >> 6;
            g.sh = o & 63;
            // two 16-byte loads from one 64-byte line: {pos, cum[4]} and {bits[ws], bits[ws+1] | next0}
            const U64x2* base = reinterpret_cast<const U64x2*>(a.idx + ch_base + (long long)(r - ch_start) * ch_nblk + b);
            const U64x2 h = base[0];
            g.p0 = h.a; g.cums = h.b;
            const U64x2 w = *reinterpret_cast<const U64x2*>(reinterpret_cast<const char*>(base) + 16 + 8 * g.ws);
            g.cur = w.a; g.nxt = w.b;
        } else {
            const unsigned* seg = nullptr;
            if (a.rowseg != nullptr) seg = a.rowseg + (long long)r * (a.n_chrom + 1) + chrom_of(a, colchrom, g.c0);
            const RowLoc loc = search_row_chunk<CH>(a, r, g.c0 + qs, nprobe, seg);
            g.spos = loc.pos; g.sbits = loc.bits;
        }
        // masked-bin bits of the window's rows and columns: wave-uniform -> scalar loads and scalar shifts
        {
            const unsigned long long* rw = a.badbits + (g.r0 >> 6);
            const unsigned long long* cw = a.badbits + (g.c0 >> 6);
            const int rs = g.r0 & 63, cs = g.c0 & 63;
            unsigned long long rb = rw[0] >> rs, cbm = cw[0] >> cs;
            if (rs) rb |= rw[1] << (64 - rs);
            if (cs) cbm |= cw[1] << (64 - cs);
            g.rowbad = (unsigned)rb; g.colbad = (unsigned)cbm;       // W <= 32 bits are used
        }
    };

    auto process = [&](const RtStage& g) __attribute__((always_inline)) {
        if (!g.valid) return;                                          // wave-uniform
        long long pos; unsigned bits;
        if (g.indexed) {
            unsigned long long b64 = g.cur >> g.sh;
            if (g.sh) b64 |= g.nxt << (64 - g.sh);
            bits = (unsigned)b64 & chmask;
            const unsigned cum = g.ws ? (unsigned)(g.cums >> ((g.ws - 1) * 16)) & 0xffffu : 0u;
            pos = (long long)(g.p0 + cum + (unsigned long long)__popcll(g.cur & ((1ull << g.sh) - 1ull)));
        } else { pos = g.spos; bits = g.sbits & chmask; }
        // values of the lane's cells: one 16-byte load per PAIR of cells.  The pixel of cell i sits at
        // pos + popc(bits below i); the pair load at that address returns it and its successor, which is cell
        // i+1's pixel when cell i holds one (otherwise cell i+1's pixel is the first value itself).  bal is
        // padded, so a cell without a pixel harmlessly reads a neighbour that is then discarded.
        double v[CH];
#pragma unroll
        for (int i = 0; i < CH; i += 2) {
            const F64x2 pr = *reinterpret_cast<const F64x2*>(a.bal + pos + __popc(bits & ((1u << i) - 1u)));
            v[i] = pr.a;
            if (i + 1 < CH) v[i + 1] = ((bits >> i) & 1u) ? pr.b : pr.a;
        }
        // validity of the lane's cells as a bit mask: bin masks, diagonal mask
        const int r = g.r0 + p, cc = g.c0 + qs;
        unsigned ok = chmask & ~(g.colbad >> qs);
        if ((g.rowbad >> p) & 1u) ok = 0u;
        if (igd >= 0) {
            // cell i is on or above the first kept diagonal iff (c0+qs+i) - (r0+p) >= igd  <=>  i >= t0
            const int t0 = igd - (cc - r);
            ok &= t0 <= 0 ? 0xffffffffu : (t0 >= 32 ? 0u : ~((1u << t0) - 1u));
        }
        double ev[CH];
        if (OOE) {
            ExpSel es; es.base = a.expv; es.len = 0; es.scalar = qnan; es.is_scalar = true;
            if (use_exp) es = select_expected(a, ecache, g.r0, g.c0);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                long long ad = (long long)(cc + i) - r; if (ad < 0) ad = -ad;
                const double e = es.is_scalar ? es.scalar : es.base[ad < es.len ? ad : 0];   // unconditional load
                ev[i] = (es.is_scalar || ad < es.len) ? e : qnan;
            }
        }
        if (m_cov) {
            if (lane_ok && k == 0) {
                const double cr = a.cov[g.r0 + p], cv = a.cov[g.c0 + p];
                const double vs = m_tr ? cv : cr, ve = m_tr ? cr : cv;
                if (vs == vs) cov_lds[p] += vs;                        // one lane per element: no race
                if (ve == ve) cov_lds[W + p] += ve;
            }
        }
        npix += (unsigned long long)__popc(bits);
        const unsigned addm = ok & bits;                                // cells that hold a pixel and are unmasked
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (OOE) {
                // reference: data / exp; NaN results (exp NaN, 0/0) are skipped, inf is kept by nansum
                const double q = v[i] / ev[i];
                const bool okn = ((ok >> i) & 1u) && (ev[i] == ev[i]) && (ev[i] != 0.0);
                num[i] += okn ? 1u : 0u;
                sum[i] += (((addm >> i) & 1u) && (q == q)) ? q : 0.0;
            } else {
                num[i] += (ok >> i) & 1u;
                sum[i] += ((addm >> i) & 1u) ? v[i] : 0.0;
            }
        }
    };

    RtStage A, B;
    issue(A, cb);
    for (long long s = cb; s < ce; s += 2 * cstep) {
        issue(B, s + cstep);
        process(A);
        issue(A, s + 2 * cstep);
        process(B);
    }

    // ---- flush: window frame -> accumulator frame (transpose, then anti-transpose when flipped) ----------
    if (m_cov) __syncthreads();
    const size_t L = (size_t)W2 + 2 * (size_t)W;
    double*   of = a.part_f64 + (size_t)ck * L;
    unsigned* on = a.part_num + (size_t)ck * W2;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if ((chmask >> i) & 1u) {
            const int cell = map_cell(p, q0 + i, W, m_tr, fl);
            of[cell] = sum[i];
            on[cell] = num[i];
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

// LDS reads the compiler must not merge: two ds_read_b64 at one base become ds_read2_b64, which the LDS serves at half
// the bytes per clock (128 vs 256 B/clk/CU on gfx950).  Issued through inline asm (the compiler does not track them):
// lds_wait_all() + lds_pin() must stand between a read and the first use of its value.
// The destination is EARLY-CLOBBER: it must not share a register with the address.  A burst of reads off one address
// register whose last read returned into that register (`ds_read_b64 v[128:129], v128 offset:0x90`) gave wrong data
// as soon as several workgroups shared a CU (found by tests/test_properties_gpu.py::test_staged_kernel_many_workgroups);
// lds_wait_all(addresses) additionally keeps the address registers untouched until the data has arrived.
template <int OFF>
__device__ __forceinline__ void lds_read_b64(double& dst, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF));
}
template <int I, int N, int STRIDE_BYTES>
struct LdsReadRow {
    static __device__ __forceinline__ void go(double* v, unsigned addr) {
        lds_read_b64<I * STRIDE_BYTES>(v[I], addr);
        LdsReadRow<I + 1, N, STRIDE_BYTES>::go(v, addr);
    }
};
template <int N, int STRIDE_BYTES>
struct LdsReadRow<N, N, STRIDE_BYTES> { static __device__ __forceinline__ void go(double*, unsigned) {} };
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// the same, and the address registers of the reads stay untouched until here (the compiler does not know the asm reads
// are still in flight and would otherwise recycle their address VGPRs right behind them)
__device__ __forceinline__ void lds_wait_all(unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}
__device__ __forceinline__ void lds_pin1(double& v) { asm volatile("" : "+v"(v)); }
template <int N>
__device__ __forceinline__ void lds_pin(double (&v)[N]) {     // later uses of v[] are ordered after the preceding asm
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// (K1q, the workgroup-staged kernel for many overlapping cis windows, and its block-order prepass: pup_staged.hpp)

// ---- K1s: sparse kernel for inter-chromosomal (trans) windows, W <= 63 ---------------------------------------
// A trans window holds a handful of pixels (5e7 trans pixels under 4.6e10 cells: ~3 per 51 x 51 window), yet a dense
// kernel touches all W^2 cells of every window — to count num.  Without a diagonal mask the validity of cell (p, q) of
// a window factorises: valid = e_ok * !rowbad[p] * !colbad[q], so over a chunk
//     num[p][q] = N_e  -  R[p]  -  C[q]  +  RC[p][q]
// with N_e the windows whose expected is usable, R / C how often window row p / column q was a masked bin, and RC how
// often both were (sparse: ~1 bad row x ~1 bad column per window).  Per window the wave therefore does O(W) work: one
// lane per window row searches its matrix row (bounded by the per-chromosome segment table) and adds the few pixels
// it finds to the chunk's partial tile.  Same chunk records and reduction as the other K1 kernels.
// Round 3.  What bounds this kernel is the texture-address unit: a vector load whose lanes go to different places costs
// about a cycle per lane whatever it fetches (counters: TA busy 73 %, 8.3 M load instructions x 64 lanes = the busy cycles),
// and round 2's kernel — a W^2 f64 tile per wave in LDS: seven waves per CU — sat behind that with too few windows in
// flight to fill it.  Now: (1) the tile lives in the chunk's OUTPUT RECORD in global memory; a lane adds the few pixels of
// its row with hardware f64 atomics that return nothing (20+ waves per CU; per cell and window there is one adder);
// (2) loads per window row cut from ~17 to ~4: masked-bin words by scalar loads, the row's pixel range in one load
// (`rowabs`), a bisection probe only while the range is longer than a LEAF of four pixels, the leaf in two 16-byte loads,
// a value load only for a pixel inside the window (3 % of the rows).
// LDS per wave: 24 W bytes (masked-row / masked-column counts, coverage).
__host__ __device__ inline size_t k1s_lds_bytes(int W) { return (size_t)W * 24; }

template <bool OOE>
__global__ __launch_bounds__(kWave) void pileup_sparse_kernel(K1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int W = a.W, W2 = W * W;
    double*   tcov = reinterpret_cast<double*>(smem_raw);            // [2W]
    unsigned* trb  = reinterpret_cast<unsigned*>(tcov + 2 * W);      // [W]   R
    unsigned* tcb  = trb + W;                                        // [W]   C
    const int lane = threadIdx.x;
    const int ck = a.block_chunk[blockIdx.x];
    if (ck < 0) return;
    unsigned* trc = a.part_num + (size_t)ck * W2;                    // [W2]  RC, already in the ACCUMULATOR frame, in the chunk's own record
    double*   tsum = a.part_f64 + (size_t)ck * ((size_t)W2 + 2 * (size_t)W);   // [W2] the chunk's sums, ACCUMULATOR frame (map_cell)
    for (int t = lane; t < W2; t += kWave) { tsum[t] = 0.0; trc[t] = 0u; }
    for (int t = lane; t < 2 * W; t += kWave) tcov[t] = 0.0;
    for (int t = lane; t < W; t += kWave) { trb[t] = 0u; tcb[t] = 0u; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");               // the zeros have reached L2 before any atomic of this wave gets there
    __syncthreads();

    const bool m_cov = (a.mode & 0x04u) && a.cov != nullptr;
    const bool m_tr  = a.mode & 0x08u;
    const bool use_exp = OOE && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const double qnan = __builtin_nan("");
    const long long cb = a.chunk_begin[ck], ce = a.chunk_end[ck], cstep = a.chunk_stride[ck];
    const int fl = a.chunk_flip[ck];
    ExpCache ecache;
    unsigned n_e = 0;                                                // wave-uniform
    unsigned long long npix = 0, nprobe = 0;
    const bool rowlane = lane < W;
    const unsigned long long wmask = W >= 64 ? ~0ull : ((1ull << W) - 1ull);

    // one window's state between its phases; four windows are in flight per wave (their load chains interleave)
    constexpr int LEAF = 4;                                          // pixels of a row looked at in registers (two 16-byte loads; 8: the same time)
    struct Win { int r0, c0; bool valid, e_ok; unsigned long long rowmask, colmask; unsigned wa, wb; double e; long long lo, b, hi; int x[LEAF];
                 int first, first_q; double first_v; };              // first pixel of the leaf inside the window (-1: none), its column, its value
    // masked-bin bits of bins [bin, bin + 64).  Worked out per BATCH, a lane per window (vector loads, all in flight together),
    // and handed to the window's turn by readlane: as scalar loads inside the window's turn they were sixteen dependent
    // round trips per four windows — 0.4 ms of the kernel (phases switched off one by one, round 3)
    auto bits64 = [&](int bin) __attribute__((always_inline)) -> unsigned long long {
        const unsigned long long* wd = a.badbits + (bin >> 6);
        const int sh = bin & 63;
        unsigned long long v = wd[0] >> sh;
        if (sh) v |= wd[1] << (64 - sh);
        return v;
    };
    auto lane64 = [&](unsigned long long v, int j) __attribute__((always_inline)) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), j) << 32) |
               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, j);
    };
    auto begin = [&](Win& w, int r0, int c0, unsigned long long rowmask, unsigned long long colmask, int kc, bool have) __attribute__((always_inline)) {
        w.valid = false; w.lo = 0; w.b = 0; w.hi = 0; w.rowmask = 0ull; w.colmask = 0ull;
        if (!have) return;
        if (r0 < 0 || c0 < 0 || (long long)r0 + W > a.nbins || (long long)c0 + W > a.nbins) {
            if (lane == 0) atomicExch(a.err, 1);
            return;
        }
        w.valid = true; w.r0 = r0; w.c0 = c0; w.e = 1.0; w.e_ok = true;
        if (OOE) {
            ExpSel es; es.base = a.expv; es.len = 0; es.scalar = qnan; es.is_scalar = true;
            if (use_exp) es = select_expected(a, ecache, r0, c0);
            // trans expected is one scalar per region pair (the engine sends by-diagonal vectors to the dense kernels)
            w.e = es.is_scalar ? es.scalar : qnan;
            w.e_ok = (w.e == w.e) && (w.e != 0.0);
        }
        w.rowmask = rowmask; w.colmask = colmask; w.wa = ~0u; w.wb = 0u;
        if (rowlane && a.tbits != nullptr)
            // which columns of this lane's matrix row hold a pixel: the word of the column block under the window
            // (consecutive lanes = consecutive rows = consecutive words: one coalesced load per window)
            w.wa = a.tbits[tbits_block(c0, a.tshift) * a.nbins + (long long)r0 + lane];
    };
    // phase 2 (the bitmap words of every window in flight have been requested): the pixel range of the rows that hold a pixel
    // inside the window — three rows in a hundred; the others are done
    auto ranges = [&](Win& w, int kc) __attribute__((always_inline)) {
        if (!w.valid || !rowlane) return;
        if (a.tbits != nullptr) {
            if (tbits_window(w.wa, w.c0, W, a.tshift) == 0u) return;     // no pixel of this row inside the window
        }
        const long long myrow = (long long)w.r0 + lane;
        if (a.rowabs != nullptr) {
            const uint2 sg = a.rowabs[(long long)kc * a.nbins + myrow];
            w.lo = sg.x; w.hi = sg.y;
        } else { w.lo = a.indptr[myrow]; w.hi = a.indptr[myrow + 1]; }
        w.b = w.hi;
    };
    auto finish = [&](Win& w) __attribute__((always_inline)) {
        if (!w.valid) return;                                         // wave-uniform
        const bool rbad = (w.rowmask >> lane) & 1ull, cbad = (w.colmask >> lane) & 1ull;     // (bits past W are clear)
        if (w.e_ok) {
            ++n_e;
            if (rbad) trb[lane] += 1u;
            if (cbad) tcb[lane] += 1u;
            if (w.rowmask && w.colmask) {
                unsigned long long rm = w.rowmask;
                while (rm) {                                          // wave-uniform loop over the (rare) masked rows
                    const int p = __ffsll((long long)rm) - 1; rm &= rm - 1;
                    if (cbad) atomicAdd(&trc[map_cell(p, lane, W, m_tr, fl)], 1u);     // (L2 atomic: rare, and coherent for the flush)
                }
            }
        }
        if (m_cov && rowlane) {
            const double cr = a.cov[w.r0 + lane], cv = a.cov[w.c0 + lane];
            const double vs = m_tr ? cv : cr, ve = m_tr ? cr : cv;
            if (vs == vs) tcov[lane] += vs;
            if (ve == ve) tcov[W + lane] += ve;
        }
        if (rowlane && w.first >= 0) {
            // pixels of this lane's matrix row inside [c0, c0 + W).  The bisection leaves the first pixel at or after the
            // window's first column INSIDE the leaf, so a row without a pixel in the window (97 % of them) is done here without a
            // load; the value of the first pixel inside was requested for all windows in flight together.  This path must not
            // contain a load: the wait for it would also wait for the ATOMICS of the windows before (no-return atomics count
            // in vmcnt on gfx9: a round trip to L2 per window — a third of the kernel, phase clocks)
            auto add_value = [&](int q, double v) __attribute__((always_inline)) {
                ++npix;
                if (rbad || ((w.colmask >> q) & 1ull)) return;        // masked bin: contributes nothing
                const double x = OOE ? v / w.e : v;
                // lane owns row `lane` of the record, a row's pixels have distinct columns: one adder per cell, and nothing
                // comes back to wait for
                if (x == x) unsafeAtomicAdd(&tsum[map_cell(lane, q, W, m_tr, fl)], x);
            };
            add_value(w.first_q, w.first_v);
            // further pixels of the row inside the window (rare): the rest of the leaf, then the table
            bool more = true;
#pragma unroll
            for (int i = 1; i < LEAF; ++i)
                if (i > w.first && more) {
                    const int q = w.x[i] - w.c0;
                    if (w.lo + i < w.hi && q < W) add_value(q, a.bal[w.lo + i]); else more = false;
                }
            if (more)
                for (long long k = w.lo + LEAF; k < w.hi; ++k) {
                    const int q = a.px[k].x - w.c0;
                    if (q >= W) break;
                    add_value(q, a.bal[k]);
                }
        }
    };

    // coordinates of 64 snippets per batch (one per lane, next batch in flight), handed out by readlane
    auto coord = [&](long long s0, const int* __restrict__ src) { const long long sl = s0 + (long long)lane * cstep; return sl < ce ? src[sl] : 0; };
    int r0n = coord(cb, a.r0), c0n = coord(cb, a.c0);
    for (long long s0 = cb; s0 < ce; s0 += (long long)kWave * cstep) {
        const int r0v = r0n, c0v = c0n;
        r0n = coord(s0 + (long long)kWave * cstep, a.r0); c0n = coord(s0 + (long long)kWave * cstep, a.c0);
        const long long left = (ce - s0 + cstep - 1) / cstep;
        const int nb = (int)(left < kWave ? left : kWave);
        // per batch, a lane per window: masked-bin words of its rows / columns, the chromosome of its columns
        unsigned long long rmv = 0ull, cmv = 0ull;
        int kcv = 0;
        if (lane < nb && r0v >= 0 && c0v >= 0 && (long long)r0v + W <= a.nbins && (long long)c0v + W <= a.nbins) {
            rmv = bits64(r0v) & wmask; cmv = bits64(c0v) & wmask;
            if (a.rowabs != nullptr) {
                int lo = 0, hi = a.n_chrom;
                while (lo < hi) { const int m = (lo + hi) >> 1; if (a.idx_chrom[m].end <= c0v) lo = m + 1; else hi = m; }
                kcv = lo < a.n_chrom ? lo : a.n_chrom - 1;
            }
        }
        constexpr int NWIN = 8;                                       // windows in flight per wave (4: 0.80 ms against 0.75)
        for (int j = 0; j < nb; j += NWIN) {
            Win w[NWIN];
#pragma unroll
            for (int u = 0; u < NWIN; ++u) {
                const int ju = j + u < nb ? j + u : j;
                begin(w[u], __builtin_amdgcn_readlane(r0v, ju), __builtin_amdgcn_readlane(c0v, ju), lane64(rmv, ju), lane64(cmv, ju),
                      __builtin_amdgcn_readlane(kcv, ju), j + u < nb);
            }
#pragma unroll
            for (int u = 0; u < NWIN; ++u) ranges(w[u], __builtin_amdgcn_readlane(kcv, j + u < nb ? j + u : j));
            // the bisections in lockstep, until the lower bound is known to within LEAF - 1 pixels — the leaf [lo, lo + LEAF) then
            // holds the first pixel at or after the window's first column; the probes of a step are independent loads
            for (;;) {
                bool any = false;
#pragma unroll
                for (int u = 0; u < NWIN; ++u) any = any || (w[u].b - w[u].lo >= LEAF);
                if (!__ballot(any)) break;
                int x[NWIN]; long long m[NWIN];
#pragma unroll
                for (int u = 0; u < NWIN; ++u) { m[u] = (w[u].lo + w[u].b) >> 1; x[u] = a.px[m[u]].x; }   // padded table: reading at a row's end is harmless
#pragma unroll
                for (int u = 0; u < NWIN; ++u)
                    if (w[u].b - w[u].lo >= LEAF) { if (x[u] < w[u].c0) w[u].lo = m[u] + 1; else w[u].b = m[u]; ++nprobe; }
            }
#pragma unroll
            for (int u = 0; u < NWIN; ++u) {
                // the leaf: LEAF pixels {col, count} from lo on, 16 bytes per load at an 8-byte aligned address (global loads
                // need dword alignment only); the table is padded: reading past a row's end is harmless
                const bool has = rowlane && w[u].valid && w[u].lo < w[u].hi;
                const int2* src = a.px + (has ? w[u].lo : 0);
#pragma unroll
                for (int i = 0; i < LEAF; i += 2) {
                    const int4 two = *reinterpret_cast<const int4*>(src + i);
                    w[u].x[i] = has ? two.x : 0x7fffffff; w[u].x[i + 1] = has ? two.z : 0x7fffffff;
                }
            }
#pragma unroll
            for (int u = 0; u < NWIN; ++u) {                          // values of the first pixels inside the windows: requested together
                w[u].first = -1; w[u].first_q = 0;
#pragma unroll
                for (int i = LEAF - 1; i >= 0; --i) {
                    const int q = w[u].x[i] - w[u].c0;
                    if (w[u].lo + i < w[u].hi && q >= 0 && q < W) { w[u].first = i; w[u].first_q = q; }
                }
                w[u].first_v = w[u].first >= 0 ? a.bal[w[u].lo + w[u].first] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NWIN; ++u) finish(w[u]);
        }
    }
    __syncthreads();
    // ---- flush: num from the factorised counts; window frame -> accumulator frame ---------------------------
    const size_t L = (size_t)W2 + 2 * (size_t)W;
    double*   of = a.part_f64 + (size_t)ck * L;
    unsigned* on = a.part_num + (size_t)ck * W2;
    for (int t = lane; t < W2; t += kWave) {
        const int p = t / W, q = t - p * W;
        const int cell = map_cell(p, q, W, m_tr, fl);
        // (the sums are in the record already)  RC was counted in place (accumulator frame); read it where the atomics put it, past the L1
        const unsigned rc = __hip_atomic_load(&on[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        on[cell] = n_e - trb[p] - tcb[q] + rc;
    }
    for (int t = lane; t < 2 * W; t += kWave) of[W2 + t] = m_cov ? tcov[t] : 0.0;
    for (int off = 32; off > 0; off >>= 1) { npix += __shfl_down(npix, off); nprobe += __shfl_down(nprobe, off); }
    if (lane == 0 && a.counters) { atomicAdd(&a.counters[0], npix); atomicAdd(&a.counters[1], nprobe); }
}

// ---- K1s, round 6: the same kernel around per-lane HIT QUEUES -------------------------------------------------------------
// In pileup_sparse_kernel a window's turn — pixel range, bisection, leaf, value, add — is executed by the 64 lanes of the wave for
// the three rows in a hundred that hold a pixel under the window: ~300 vector and ~160 scalar instructions per window, 97 % of
// their lanes idle (counters, round 3 / 5: waves waiting 66 %, 0.11 of the HBM peak on the traffic).  Here the batch of 64
// windows is walked TWICE:
//   phase A: the factorised-count bookkeeping with a lane per WINDOW (its masked-row / masked-column words -> integer atomics), then
//            per window (uniform) one coalesced load of the filter words of its rows and the hit test — a lane whose row MAY hold
//            a pixel appends the window's number to ITS OWN queue (LDS, 64 entries x 64 lanes, slot-major: q[slot][lane]);
//   phase B, per queue slot: lane p takes the slot-th window of its queue — its OWN window, coordinates by ds_bpermute from the
//            lane that holds them — and runs the lookup chain for (that window, row p): every lane with a queued hit is busy.
// A batch holds ~1.5 hits per window = ~2 per lane: phase B is one or two rounds of U = 4 chains instead of 64 / 8 rounds of 8.
// Lane p still owns row p of the chunk's record and meets its windows in batch order, its pixels in column order: the adds of a
// cell are the ones the first form makes, in the same order — the two forms agree bit for bit (tests run both: tuning bit 21).
constexpr int kSparseLdsChroms = 256;                    // chromosome ends kept in LDS for the batch set-up (more: bisection in global memory)
__host__ __device__ inline size_t k1sq_lds_bytes(int W) { return k1s_lds_bytes(W) + 64 * 64 + 8 + kSparseLdsChroms * sizeof(int); }

#ifndef PUP_K1S_CLOCKS
#define PUP_K1S_CLOCKS 0
#endif
#ifndef PUP_K1S_NA
#define PUP_K1S_NA 16
#endif
#ifndef PUP_K1S_WAVES
#define PUP_K1S_WAVES 4
#endif
#ifndef PUP_K1S_U
#define PUP_K1S_U 4
#endif
template <bool OOE>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(PUP_K1S_WAVES, PUP_K1S_WAVES))) void pileup_sparse_queue_kernel(K1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int W = a.W, W2 = W * W;
    double*   tcov = reinterpret_cast<double*>(smem_raw);            // [2W]
    unsigned* trb  = reinterpret_cast<unsigned*>(tcov + 2 * W);      // [W]   R
    unsigned* tcb  = trb + W;                                        // [W]   C
    unsigned char* queue = reinterpret_cast<unsigned char*>(tcb + W);               // [64 slots][64 lanes]: window numbers
    int* s_cend = reinterpret_cast<int*>(queue + 64 * 64);           // [n_chrom <= kSparseLdsChroms] chromosome ends (24 W + 4096 bytes in: aligned)
    const int lane = threadIdx.x;
    const int ck = a.block_chunk[blockIdx.x];
    if (ck < 0) return;
#if PUP_K1S_CLOCKS
    // (dev builds, -DPUP_K1S_CLOCKS=1: per-wave clocks of the phases leave through the two diagnostic counters, packed as
    // clocks / 16 in 32-bit halves: [0] = total | phase A << 32, [1] = phase B | batch set-up << 32 — tools/probe_trans.py PHASES=1)
    const long long tk0 = (long long)__builtin_readcyclecounter();
    long long tkA = 0, tkB = 0, tkS = 0;
#endif
    unsigned* trc = a.part_num + (size_t)ck * W2;                    // [W2]  RC, accumulator frame, in the chunk's own record
    double*   tsum = a.part_f64 + (size_t)ck * ((size_t)W2 + 2 * (size_t)W);   // [W2] the chunk's sums, accumulator frame (map_cell)
    for (int t = lane; t < W2; t += kWave) { tsum[t] = 0.0; trc[t] = 0u; }
    for (int t = lane; t < 2 * W; t += kWave) tcov[t] = 0.0;
    for (int t = lane; t < W; t += kWave) { trb[t] = 0u; tcb[t] = 0u; }
    const bool lds_chroms = a.rowabs != nullptr && a.n_chrom <= kSparseLdsChroms;
    if (lds_chroms) for (int t = lane; t < a.n_chrom; t += kWave) s_cend[t] = a.idx_chrom[t].end;
    // the zeros have reached L2 before any atomic of this wave gets there: the wave waits for its stores' acknowledgements
    // (workgroup scope = s_waitcnt vmcnt(0); the record is this wave's alone and its atomics execute in the same XCD's L2 — the
    // agent-scope release of the first form also wrote the whole L2 back, `buffer_wbl2 sc1`, once per wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();

    const bool m_cov = (a.mode & 0x04u) && a.cov != nullptr;
    const bool m_tr  = a.mode & 0x08u;
    const bool use_exp = OOE && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const double qnan = __builtin_nan("");
    const long long cb = a.chunk_begin[ck], ce = a.chunk_end[ck], cstep = a.chunk_stride[ck];
    const int fl = a.chunk_flip[ck];
    ExpCache ecache;
    unsigned n_e = 0;                                                // wave-uniform
    double cov_s = 0.0, cov_e = 0.0;                                 // lane p: coverage sums of window row / column p
    unsigned long long npix = 0, nprobe = 0;
    const bool rowlane = lane < W;
    const unsigned long long wmask = W >= 64 ? ~0ull : ((1ull << W) - 1ull);
    const bool filt = a.tbits != nullptr;
    const int tsh = a.tshift;
    const bool from_counts = a.cntf == nullptr;                      // (uniform) pixel values formed from the leaf's counts and the weight vector

    auto bits64 = [&](int bin) __attribute__((always_inline)) -> unsigned long long {
        const unsigned long long* wd = a.badbits + (bin >> 6);
        const int sh = bin & 63;
        unsigned long long v = wd[0] >> sh;
        if (sh) v |= wd[1] << (64 - sh);
        return v;
    };
    // value of `v` in lane `src` (per lane: ds_bpermute goes through the LDS crossbar, no memory)
    auto from32 = [&](int v, int src) __attribute__((always_inline)) -> int { return __builtin_amdgcn_ds_bpermute(src << 2, v); };
    auto from64 = [&](unsigned long long v, int src) __attribute__((always_inline)) -> unsigned long long {
        return ((unsigned long long)(unsigned)from32((int)(v >> 32), src) << 32) | (unsigned)from32((int)(unsigned)v, src);
    };

    constexpr int LEAF = 4;                                          // pixels of a row looked at in registers (two 16-byte loads)
    constexpr int NA = PUP_K1S_NA;                                   // windows of phase A whose filter words are in flight together
    constexpr int U = PUP_K1S_U;                                     // queue slots of phase B in flight per lane
#if PUP_K1S_CLOCKS
    const long long tkP = (long long)__builtin_readcyclecounter() - tk0;
#endif
    auto coord = [&](long long s0, const int* __restrict__ src) { const long long sl = s0 + (long long)lane * cstep; return sl < ce ? src[sl] : 0; };
    int r0n = coord(cb, a.r0), c0n = coord(cb, a.c0);
    for (long long s0 = cb; s0 < ce; s0 += (long long)kWave * cstep) {
#if PUP_K1S_CLOCKS
        const long long tkb0 = (long long)__builtin_readcyclecounter();
#endif
        const int r0v = r0n, c0v = c0n;
        r0n = coord(s0 + (long long)kWave * cstep, a.r0); c0n = coord(s0 + (long long)kWave * cstep, a.c0);
        const long long left = (ce - s0 + cstep - 1) / cstep;
        const int nb = (int)(left < kWave ? left : kWave);
        // per batch, a lane per window: is it inside the table, masked-bin words of its rows / columns, the chromosome of its columns
        unsigned long long rmv = 0ull, cmv = 0ull;
        int kcv = 0;
        const bool okv = lane < nb && r0v >= 0 && c0v >= 0 && (long long)r0v + W <= a.nbins && (long long)c0v + W <= a.nbins;
        if (okv) {
            rmv = bits64(r0v) & wmask; cmv = bits64(c0v) & wmask;
            if (a.rowabs != nullptr) {
                // (the chromosome table in LDS: five dependent global loads per batch were a tenth of the kernel — phase clocks)
                int lo = 0, hi = a.n_chrom;
                if (lds_chroms) { while (lo < hi) { const int m = (lo + hi) >> 1; if (s_cend[m] <= c0v) lo = m + 1; else hi = m; } }
                else { while (lo < hi) { const int m = (lo + hi) >> 1; if (a.idx_chrom[m].end <= c0v) lo = m + 1; else hi = m; } }
                kcv = lo < a.n_chrom ? lo : a.n_chrom - 1;
            }
        }
        const unsigned long long okm = __ballot(okv);                // (uniform) windows of the batch inside the table
        if (okm != (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull)) && lane == 0) atomicExch(a.err, 1);
        unsigned long long evv = 0ull;                               // OOE: lane j holds the expected of window j (bits of the double)
        int cnt = 0;                                                 // entries of this lane's queue

#if PUP_K1S_CLOCKS
        const long long tkb1 = (long long)__builtin_readcyclecounter();
        tkS += tkb1 - tkb0;
#endif
        // ---- phase A ----------------------------------------------------------------------------------------------------------
        // (1) the factorised counts, a LANE PER WINDOW (round 6, second pass: walked window by window with the whole wave — two
        // 64-bit mask tests, two counters and a branch per window — this bookkeeping was most of the kernel's ~100 vector and ~85
        // scalar instructions per window, and instruction issue is what the kernel is bound by once it no longer waits for
        // memory: counters, SIMDs busy > 90 %).  Lane j walks the set bits of ITS window's masked-row / masked-column words:
        // integer LDS atomics for R[p] and C[q], the record's own cells (L2 atomics) for the pairs — exact, order-free.
        unsigned long long eokm = okm;                               // (uniform) windows that count: inside the table, usable expected
        if (OOE) {
            for (int ju = 0; ju < nb; ++ju) {
                if (!((okm >> ju) & 1ull)) continue;
                const int r0 = __builtin_amdgcn_readlane(r0v, ju), c0 = __builtin_amdgcn_readlane(c0v, ju);
                ExpSel es; es.base = a.expv; es.len = 0; es.scalar = qnan; es.is_scalar = true;
                if (use_exp) es = select_expected(a, ecache, r0, c0);
                // trans expected is one scalar per region pair (the engine sends by-diagonal vectors to the dense kernels)
                const double e = es.is_scalar ? es.scalar : qnan;
                if (!((e == e) && (e != 0.0))) eokm &= ~(1ull << ju);
                if (lane == ju) evv = (unsigned long long)__double_as_longlong(e);
            }
        }
        n_e += (unsigned)__builtin_popcountll(eokm);
        {
            const bool mine = (eokm >> lane) & 1ull;
            const unsigned long long rm = mine ? rmv : 0ull, cmk = mine ? cmv : 0ull;
            for (unsigned long long x = rm; x; x &= x - 1) atomicAdd(&trb[__ffsll((long long)x) - 1], 1u);
            for (unsigned long long y = cmk; y; y &= y - 1) atomicAdd(&tcb[__ffsll((long long)y) - 1], 1u);
            if (cmk)
                for (unsigned long long x = rm; x; x &= x - 1) {
                    const int p = __ffsll((long long)x) - 1;
                    for (unsigned long long y = cmk; y; y &= y - 1) atomicAdd(&trc[map_cell(p, __ffsll((long long)y) - 1, W, m_tr, fl)], 1u);
                }
        }
        // (2) coverage vectors (coverage_norm): lane p adds its bins of every window, in batch order
        if (m_cov) {
            for (int j = 0; j < nb; j += 8) {
                double cr[8], cv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ju = j + u < nb ? j + u : j;
                    const int r0 = __builtin_amdgcn_readlane(r0v, ju), c0 = __builtin_amdgcn_readlane(c0v, ju);
                    const bool on = rowlane && j + u < nb && ((okm >> ju) & 1ull);
                    cr[u] = on ? a.cov[r0 + lane] : qnan; cv[u] = on ? a.cov[c0 + lane] : qnan;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double vs = m_tr ? cv[u] : cr[u], ve = m_tr ? cr[u] : cv[u];
                    if (vs == vs) cov_s += vs;
                    if (ve == ve) cov_e += ve;
                }
            }
        }
        // (3) the filter: which rows of which windows may hold a pixel under the window -> the lanes' queues
        for (int j = 0; j < nb; j += NA) {
            unsigned wa[NA];
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int ju = j + u < nb ? j + u : j;
                wa[u] = ~0u;
                if (filt && rowlane && ((okm >> ju) & 1ull)) {
                    // which columns of this lane's matrix row hold a pixel: the word of the column block under the window
                    // (consecutive lanes = consecutive rows = consecutive words: 4 W bytes, two or three lines)
                    const int r0 = __builtin_amdgcn_readlane(r0v, ju), c0 = __builtin_amdgcn_readlane(c0v, ju);
                    wa[u] = a.tbits[tbits_block(c0, tsh) * a.nbins + (long long)r0 + lane];
                }
            }
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int ju = j + u;
                if (ju >= nb || !((okm >> ju) & 1ull)) continue;      // (uniform)
                const int c0 = __builtin_amdgcn_readlane(c0v, ju);
                const bool hit = rowlane && (!filt || tbits_window(wa[u], c0, W, tsh) != 0u);
                if (hit) { queue[cnt * kWave + lane] = (unsigned char)ju; ++cnt; }
            }
        }

#if PUP_K1S_CLOCKS
        const long long tkb2 = (long long)__builtin_readcyclecounter();
        tkA += tkb2 - tkb1;
#endif
        // ---- phase B: lane p and the windows of ITS queue, U at a time ----------------------------------------------------------
        for (int t = 0; __ballot(t < cnt) != 0ull; t += U) {
            bool live[U], rbad[U];
            int r0[U], c0[U];
            unsigned long long cm[U];
            double ev[U];
            long long lo[U], bnd[U], hi[U];
            double wrow[U];                                           // weight of the lane's matrix row (values from counts: below)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                live[u] = t + u < cnt;
                const int jw = live[u] ? (int)queue[(t + u) * kWave + lane] : 0;
                rbad[u] = (from64(rmv, jw) >> lane) & 1ull;           // (bits past W are clear)
                r0[u] = from32(r0v, jw); c0[u] = from32(c0v, jw);
                cm[u] = from64(cmv, jw);
                ev[u] = OOE ? __longlong_as_double((long long)from64(evv, jw)) : 1.0;
                const int kc = from32(kcv, jw);
                lo[u] = 0; hi[u] = 0; wrow[u] = 1.0;
                if (live[u]) {
                    const long long myrow = (long long)r0[u] + lane;
                    if (a.rowabs != nullptr) { const uint2 sg = a.rowabs[(long long)kc * a.nbins + myrow]; lo[u] = sg.x; hi[u] = sg.y; }
                    else { lo[u] = a.indptr[myrow]; hi[u] = a.indptr[myrow + 1]; }
                    if (from_counts && a.weight) wrow[u] = a.weight[myrow];
                }
                bnd[u] = hi[u];
            }
            // the bisections in lockstep, until the lower bound is known to within LEAF - 1 pixels — the leaf [lo, lo + LEAF) then
            // holds the first pixel at or after the window's first column; the probes of a step are independent loads
            for (;;) {
                bool any = false;
#pragma unroll
                for (int u = 0; u < U; ++u) any = any || (bnd[u] - lo[u] >= LEAF);
                if (!__ballot(any)) break;
                int x[U]; long long m[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { m[u] = (lo[u] + bnd[u]) >> 1; x[u] = a.px[m[u]].x; }   // padded table: reading at a row's end is harmless
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (bnd[u] - lo[u] >= LEAF) { if (x[u] < c0[u]) lo[u] = m[u] + 1; else bnd[u] = m[u]; ++nprobe; }
            }
            int xs[U][LEAF], first[U], first_q[U], first_cnt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool has = live[u] && lo[u] < hi[u];
                const int2* src = a.px + (has ? lo[u] : 0);
                first[u] = -1; first_q[u] = 0; first_cnt[u] = 0;
#pragma unroll
                for (int i = LEAF - 2; i >= 0; i -= 2) {              // (backwards: the FIRST pixel inside the window is what is kept)
                    const int4 two = *reinterpret_cast<const int4*>(src + i);
                    xs[u][i] = has ? two.x : 0x7fffffff; xs[u][i + 1] = has ? two.z : 0x7fffffff;
                    const int q1 = xs[u][i + 1] - c0[u], q0 = xs[u][i] - c0[u];
                    if (lo[u] + i + 1 < hi[u] && q1 >= 0 && q1 < W) { first[u] = i + 1; first_q[u] = q1; first_cnt[u] = two.w; }
                    if (lo[u] + i < hi[u] && q0 >= 0 && q0 < W) { first[u] = i; first_q[u] = q0; first_cnt[u] = two.y; }
                }
            }
            // value of pixel k = (row of the lane, column col, count cnt).  The leaf holds the COUNT beside the column, and the balanced
            // value table is nothing but (count * w[row]) * w[col] with NaN products stored as 0 (balance_pixels_kernel): formed here, in
            // that order, it is the same double — and the weights are a 2 MB vector that lives in L2, where `bal[k]` was one more random
            // line from HBM per pixel found (a table of float pixel values keeps reading `bal`)
            auto value_of = [&](int u, long long k, int col, int cnt_k, double wcol) __attribute__((always_inline)) -> double {
                if (!from_counts) return a.bal[k];
                double v = (double)cnt_k;
                if (a.weight) { v = v * wrow[u] * wcol; if (!(v == v)) v = 0.0; }
                return v;
            };
            double first_w[U];                                        // from_counts: the weight of the first pixel's column, else its value
#pragma unroll
            for (int u = 0; u < U; ++u) {                             // requested together for the U windows
                first_w[u] = 0.0;
                if (first[u] >= 0) first_w[u] = from_counts ? (a.weight ? a.weight[c0[u] + first_q[u]] : 1.0) : a.bal[lo[u] + first[u]];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (first[u] < 0) continue;
                // (no load on this path before the first add: see pileup_sparse_kernel)
                auto add_value = [&](int q, double v) __attribute__((always_inline)) {
                    ++npix;
                    if (rbad[u] || ((cm[u] >> q) & 1ull)) return;     // masked bin: contributes nothing
                    const double x = OOE ? v / ev[u] : v;
                    if (x == x) unsafeAtomicAdd(&tsum[map_cell(lane, q, W, m_tr, fl)], x);
                };
                add_value(first_q[u], from_counts ? value_of(u, 0, 0, first_cnt[u], first_w[u]) : first_w[u]);
                bool more = true;
#pragma unroll
                for (int i = 1; i < LEAF; ++i)
                    if (i > first[u] && more) {
                        const int q = xs[u][i] - c0[u];
                        if (lo[u] + i < hi[u] && q < W)            // (a second pixel of the row inside the window: rare — its count comes from the table again)
                            add_value(q, value_of(u, lo[u] + i, xs[u][i], from_counts ? a.px[lo[u] + i].y : 0, (from_counts && a.weight) ? a.weight[xs[u][i]] : 1.0));
                        else more = false;
                    }
                if (more)
                    for (long long k = lo[u] + LEAF; k < hi[u]; ++k) {
                        const int2 pc = a.px[k];
                        const int q = pc.x - c0[u];
                        if (q >= W) break;
                        add_value(q, value_of(u, k, pc.x, pc.y, (from_counts && a.weight) ? a.weight[pc.x] : 1.0));
                    }
            }
        }
#if PUP_K1S_CLOCKS
        tkB += (long long)__builtin_readcyclecounter() - tkb2;
#endif
    }
#if PUP_K1S_CLOCKS
    const long long tkF0 = (long long)__builtin_readcyclecounter();
#endif
    if (m_cov && rowlane) { tcov[lane] = cov_s; tcov[W + lane] = cov_e; }
    __syncthreads();
    // ---- flush: num from the factorised counts; window frame -> accumulator frame ---------------------------
    const size_t L = (size_t)W2 + 2 * (size_t)W;
    double*   of = a.part_f64 + (size_t)ck * L;
    unsigned* on = a.part_num + (size_t)ck * W2;
    // RC was counted in place by L2 atomics; the rest of num joins it the same way — an atomic that returns nothing: reading RC
    // back (past the L1) was a dependent round trip per cell and lane, 41 of them for a 51 x 51 record, 4-5 us each under this
    // kernel's load (phase clocks, round 6).  u32 arithmetic wraps: the sum is what it would have been.
    for (int t = lane; t < W2; t += kWave) {
        const int p = t / W, q = t - p * W;
        atomicAdd(&on[map_cell(p, q, W, m_tr, fl)], n_e - trb[p] - tcb[q]);
    }
    for (int t = lane; t < 2 * W; t += kWave) of[W2 + t] = m_cov ? tcov[t] : 0.0;
#if PUP_K1S_CLOCKS
    const long long tkF1 = (long long)__builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0 && a.counters) {
        const long long tkE = (long long)__builtin_readcyclecounter();
        const unsigned long long tot = (unsigned long long)(tkE - tk0) >> 4;
#if PUP_K1S_CLOCKS == 2
        // (second set: prologue, flush issue, final drain)
        atomicAdd(&a.counters[0], tot | ((unsigned long long)tkP >> 4) << 32);
        atomicAdd(&a.counters[1], ((unsigned long long)(tkF1 - tkF0) >> 4) | ((unsigned long long)(tkE - tkF1) >> 4) << 32);
#else
        atomicAdd(&a.counters[0], tot | ((unsigned long long)tkA >> 4) << 32);
        atomicAdd(&a.counters[1], ((unsigned long long)tkB >> 4) | ((unsigned long long)tkS >> 4) << 32);
#endif
    }
    return;
#endif
    for (int off = 32; off > 0; off >>= 1) { npix += __shfl_down(npix, off); nprobe += __shfl_down(nprobe, off); }
    if (lane == 0 && a.counters) { atomicAdd(&a.counters[0], npix); atomicAdd(&a.counters[1], nprobe); }
}

// ---- K1b: banded register-tile kernel for wide windows (31 < W <= 16*NCH, up to 255) ----------------------------
// Same idea as K1r, but a wave owns only a BAND of H = 64/NCH consecutive window rows of every snippet of its chunk
// (lane (p,k): row band*H + p, the 16 columns [16k, 16k+16)), so the per-lane register tile stays 16 cells whatever
// the window width; the ceil(W/H) band-waves of a chunk write disjoint rows of the chunk's one partial tile.
// NCH = 4 / 8 / 16 serves W <= 64 / 128 / 255.  Row/column validity bits are fetched per lane (rows: one bit,
// columns: 16 bits) because a band's rows and a lane's columns are no longer wave-uniform 32-bit masks.
constexpr int kBandCH = 16;

struct BandStage {
    unsigned long long p0, cums, cur, nxt;
    unsigned long long rw, cw0, cw1;   // masked-bin words: the lane's row, the lane's 16 columns
    int                ws, sh;
    long long          spos;
    unsigned           sbits;
    int                r0, c0;
    bool               valid, indexed;
};

template <int NCH, bool OOE>
__global__ __launch_bounds__(kWave, 3) void pileup_band_kernel(K1Args a) {
    constexpr int CH = kBandCH;
    constexpr int H  = kWave / NCH;                  // rows per band
    const int W  = a.W;
    const int W2 = W * W;
    const int lane = threadIdx.x;
    const int ck = a.block_chunk[blockIdx.x];
    if (ck < 0) return;
    const int band = a.block_band[blockIdx.x] & 0xffff;
    const int cpanel = a.block_band[blockIdx.x] >> 16;   // windows wider than 16 NCH bins: column panel of this wave
    const int p_in = lane / NCH;
    const int k    = lane - p_in * NCH;
    const int pg_raw = band * H + p_in;              // window row of this lane
    const int q0 = (cpanel * NCH + k) * CH;
    const bool lane_ok = (pg_raw < W) && (q0 < W);
    const int pg = pg_raw < W ? pg_raw : W - 1;
    const int chw = lane_ok ? ((W - q0) < CH ? (W - q0) : CH) : 0;
    const unsigned chmask = chw >= 32 ? 0xffffffffu : ((1u << chw) - 1u);
    const int qs = q0 < W ? q0 : 0;

    const bool m_cov   = (a.mode & 0x04u) && a.cov != nullptr;
    const bool m_tr    = a.mode & 0x08u;
    const bool use_exp = OOE && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const int  igd     = a.ignore_diags;
    const bool have_idx = a.idx != nullptr;
    const double qnan = __builtin_nan("");
    ExpCache ecache;
    ChromOf colchrom;

    double   sum[CH];
    unsigned num[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { sum[i] = 0.0; num[i] = 0u; }
    double cov_s = 0.0, cov_e = 0.0;                 // lanes with k == 0: coverage of window row pg / column pg

    const long long cb = a.chunk_begin[ck];
    const long long ce = a.chunk_end[ck];
    const long long cstep = a.chunk_stride[ck];
    const int fl = a.chunk_flip[ck];
    unsigned long long npix = 0, nprobe = 0;
    int ch_start = 0, ch_end = -1, ch_nblk = 0; long long ch_base = 0;

    auto issue = [&](BandStage& g, long long s) __attribute__((always_inline)) {
        g.valid = false; g.indexed = false;
        if (s >= ce) return;
        g.r0 = __builtin_amdgcn_readfirstlane(a.r0[s]);
        g.c0 = __builtin_amdgcn_readfirstlane(a.c0[s]);
        if (g.r0 < 0 || g.c0 < 0 || (long long)g.r0 + W > a.nbins || (long long)g.c0 + W > a.nbins) {
            if (lane == 0 && band == 0 && cpanel == 0) atomicExch(a.err, 1);
            return;
        }
        g.valid = true;
        if (have_idx) {
            if (!(g.r0 >= ch_start && g.r0 < ch_end)) {
                int lo = 0, hi_k = a.n_chrom;
                while (lo < hi_k) { const int m = (lo + hi_k) >> 1; if (a.idx_chrom[m].end <= g.r0) lo = m + 1; else hi_k = m; }
                if (lo < a.n_chrom) {
                    const IdxChrom c = a.idx_chrom[lo];
                    ch_start = c.start; ch_end = c.end; ch_nblk = c.nblk; ch_base = c.blk_base;
                } else { ch_start = 0; ch_end = -1; }
            }
            g.indexed = g.r0 >= ch_start && g.r0 + W <= ch_end && g.c0 >= ch_start && g.c0 + W <= ch_end;
        }
        const int r = g.r0 + pg;
        if (g.indexed) {
            const int rel = (g.c0 - ch_start) + qs;
            const int b = rel / kIdxCols, o = rel - b * kIdxCols;
            g.ws = o >> 6;
            g.sh = o & 63;
            const U64x2* base = reinterpret_cast<const U64x2*>(a.idx + ch_base + (long long)(r - ch_start) * ch_nblk + b);
            const U64x2 h = base[0];
            g.p0 = h.a; g.cums = h.b;
            const U64x2 w = *reinterpret_cast<const U64x2*>(reinterpret_cast<const char*>(base) + 16 + 8 * g.ws);
            g.cur = w.a; g.nxt = w.b;
        } else {
            const unsigned* seg = nullptr;
            if (a.rowseg != nullptr) seg = a.rowseg + (long long)r * (a.n_chrom + 1) + chrom_of(a, colchrom, g.c0);
            const RowLoc loc = search_row_chunk<CH>(a, r, g.c0 + qs, nprobe, seg);
            g.spos = loc.pos; g.sbits = loc.bits;
        }
        g.rw = a.badbits[r >> 6];
        const U64x2 cw = *reinterpret_cast<const U64x2*>(a.badbits + ((g.c0 + qs) >> 6));
        g.cw0 = cw.a; g.cw1 = cw.b;
    };

    auto process = [&](const BandStage& g) __attribute__((always_inline)) {
        if (!g.valid) return;
        long long pos; unsigned bits;
        if (g.indexed) {
            unsigned long long b64 = g.cur >> g.sh;
            if (g.sh) b64 |= g.nxt << (64 - g.sh);
            bits = (unsigned)b64 & chmask;
            const unsigned cum = g.ws ? (unsigned)(g.cums >> ((g.ws - 1) * 16)) & 0xffffu : 0u;
            pos = (long long)(g.p0 + cum + (unsigned long long)__popcll(g.cur & ((1ull << g.sh) - 1ull)));
        } else { pos = g.spos; bits = g.sbits & chmask; }
        double v[CH];
        // sparse (inter-chromosomal) windows: most bands hold no pixel at all -> skip the value loads wave-wide
        if (g.indexed || __ballot(bits != 0u) != 0ull) {
#pragma unroll
            for (int i = 0; i < CH; i += 2) {
                const F64x2 pr = *reinterpret_cast<const F64x2*>(a.bal + pos + __popc(bits & ((1u << i) - 1u)));
                v[i] = pr.a;
                if (i + 1 < CH) v[i + 1] = ((bits >> i) & 1u) ? pr.b : pr.a;
            }
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] = 0.0;
        }
        const int r = g.r0 + pg, cc = g.c0 + qs;
        const int csh = cc & 63;
        unsigned long long cb64 = g.cw0 >> csh;
        if (csh) cb64 |= g.cw1 << (64 - csh);
        unsigned ok = chmask & ~(unsigned)cb64;
        if ((g.rw >> (r & 63)) & 1ull) ok = 0u;
        if (igd >= 0) {
            const int t0 = igd - (cc - r);
            ok &= t0 <= 0 ? 0xffffffffu : (t0 >= 32 ? 0u : ~((1u << t0) - 1u));
        }
        double ev[CH];
        if (OOE) {
            ExpSel es; es.base = a.expv; es.len = 0; es.scalar = qnan; es.is_scalar = true;
            if (use_exp) es = select_expected(a, ecache, g.r0, g.c0);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                long long ad = (long long)(cc + i) - r; if (ad < 0) ad = -ad;
                const double e = es.is_scalar ? es.scalar : es.base[ad < es.len ? ad : 0];
                ev[i] = (es.is_scalar || ad < es.len) ? e : qnan;
            }
        }
        if (m_cov && lane_ok && k == 0 && cpanel == 0) {
            const double cr = a.cov[g.r0 + pg], cv = a.cov[g.c0 + pg];
            const double vs = m_tr ? cv : cr, ve = m_tr ? cr : cv;
            if (vs == vs) cov_s += vs;
            if (ve == ve) cov_e += ve;
        }
        npix += (unsigned long long)__popc(bits);
        const unsigned addm = ok & bits;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (OOE) {
                const double q = v[i] / ev[i];
                const bool okn = ((ok >> i) & 1u) && (ev[i] == ev[i]) && (ev[i] != 0.0);
                num[i] += okn ? 1u : 0u;
                sum[i] += (((addm >> i) & 1u) && (q == q)) ? q : 0.0;
            } else {
                num[i] += (ok >> i) & 1u;
                sum[i] += ((addm >> i) & 1u) ? v[i] : 0.0;
            }
        }
    };

    BandStage A, B;
    issue(A, cb);
    for (long long s = cb; s < ce; s += 2 * cstep) {
        issue(B, s + cstep);
        process(A);
        issue(A, s + 2 * cstep);
        process(B);
    }

    const size_t L = (size_t)W2 + 2 * (size_t)W;
    double*   of = a.part_f64 + (size_t)ck * L;
    unsigned* on = a.part_num + (size_t)ck * W2;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if ((chmask >> i) & 1u) {
            const int cell = map_cell(pg, q0 + i, W, m_tr, fl);
            of[cell] = sum[i];
            on[cell] = num[i];
        }
    }
    if (lane_ok && k == 0 && cpanel == 0) { of[W2 + pg] = cov_s; of[W2 + W + pg] = cov_e; }
    for (int off = 32; off > 0; off >>= 1) {
        npix   += __shfl_down(npix, off);
        nprobe += __shfl_down(nprobe, off);
    }
    if (lane == 0 && a.counters) {
        atomicAdd(&a.counters[0], npix);
        atomicAdd(&a.counters[1], nprobe);
    }
}

// ---- K5: rescaled pile-up (variable h x w windows zoomed to S x S) -------------------------------------------
// PileUpper._rescale_snip (coolpup.py:1193-1234) with cooltools' zoom_array on the GPU: per snippet the masked /
// normalised window is (local pile-ups) symmetrised by nanmean, NaN -> 0 (+inf -> DBL_MAX, np.nan_to_num), blown up
// by bilinear interpolation (scipy.ndimage.zoom order=1: sample o of an axis reads coordinate o*(n_in-1)/(n_out-1);
// a coordinate that rounding pushes past n_in-1 yields 0) to the integer multiple (mh*S) x (mw*S) of the target and
// block-averaged; an output cell is NaN when any contributing sample touched a NaN.  An all-NaN window contributes
// zeros (counted in num), as the reference does.  One 256-thread workgroup per chunk; thread t owns output cells
// t, t+256, ... of the chunk's LDS tile for every snippet (no atomics, no barrier in the snippet loop); every input
// cell is fetched on demand (rank-bitmap index or binary search) — rescaled pile-ups are thousands of windows, not
// millions, so the gather is not staged.
struct RsGeom { int ch_start, ch_end, ch_nblk; long long ch_base; bool have; };
constexpr int kRescaleSepK = 64;                      // most zoom weights kept per output row / column (separable zoom): windows up to 62 x the output

// balanced value of the stored pixel at `pos` as a per-snippet output shows it: the `bal` table — except when the weight
// column holds +-inf: cooler multiplies such pixels out to +-inf or NaN (0 * inf) and the reference's windows carry exactly
// that (coolpuppy/coolpup.py:1115-1123), while `bal` keeps NaN products as 0 for the accumulating kernels
__device__ __forceinline__ double pixel_value(const K1Args& a, long long pos, int row, int col) {
    if (a.nf_pixels && a.weight) return (a.cntf ? a.cntf[pos] : (double)a.cnt32[pos]) * a.weight[row] * a.weight[col];
    return a.bal[pos];
}

__device__ __forceinline__ double lookup_bal(const K1Args& a, const RsGeom& g, int row, int col) {
    // near the diagonal the count comes from the dense band (one load, nothing depends on it but the product; the index path below
    // is a chain of two loads and a popcount): rescaled pile-ups are windows ON the diagonal.  Same double as the `bal` table:
    // (count * w[row]) * w[col], NaN -> 0 unless the weights hold infinities (pixel_value)
    if (a.band != nullptr && col >= row && col - row < a.band_w && g.have && row >= g.ch_start && col < g.ch_end) {
        const int cnt = a.band[(long long)row * a.band_w + (col - row)];
        double v = (double)cnt;
        if (a.weight) { v = v * a.weight[row] * a.weight[col]; if (!a.nf_pixels && !(v == v)) v = 0.0; }
        return cnt ? v : 0.0;
    }
    if (g.have && row >= g.ch_start && row < g.ch_end && col >= g.ch_start && col < g.ch_end) {
        const int rel = col - g.ch_start;
        const int b = rel / kIdxCols, o = rel - b * kIdxCols;
        const int ws = o >> 6, sh = o & 63;
        const IdxBlock* blk = a.idx + g.ch_base + (long long)(row - g.ch_start) * g.ch_nblk + b;
        const unsigned long long wbits = blk->bits[ws];
        if (!((wbits >> sh) & 1ull)) return 0.0;
        const unsigned cum = ws ? blk->cum[ws - 1] : 0u;
        const long long pos = (long long)(blk->pos + cum + (unsigned long long)__popcll(wbits & ((1ull << sh) - 1ull)));
        return pixel_value(a, pos, row, col);
    }
    long long lo = a.indptr[row], hi = a.indptr[row + 1];
    const long long end = hi;
    while (lo < hi) { const long long m = (lo + hi) >> 1; if (a.px[m].x < col) lo = m + 1; else hi = m; }
    return (lo < end && a.px[lo].x == col) ? pixel_value(a, lo, row, col) : 0.0;
}

__device__ __forceinline__ bool bin_bad(const K1Args& a, int bin) { return (a.badbits[bin >> 6] >> (bin & 63)) & 1ull; }

//
// Emission mode (emit != nullptr, pup_extract): nothing is accumulated; the zoomed S x S tile of snippet s is written
// to emit[s] (reference frame, i.e. TRANSPOSE undone; NaN where the reference's zoomed NaN mask is set) and its zoomed
// coverage vectors to emit_cov[s] = {cov_start[S], cov_end[S]}.  Blocks then stride over snippets 0..emit_n.
PUP_KERNEL __launch_bounds__(1024) void pileup_rescale_kernel(K1Args a, const int* __restrict__ hs, const int* __restrict__ wsz,
                                                             double* __restrict__ emit, double* __restrict__ emit_cov,
                                                             long long emit_n, double* __restrict__ scratch, long long scratch_cells,
                                                             int sep_k /* > 0: LDS holds room for sep_k zoom weights per output row / column */,
                                                             double* __restrict__ tile_g = nullptr /* output tiles too large for LDS (S > ~115): one
                                                                 per workgroup in global memory, S2 doubles + S2 counts, 8-byte aligned stride */) {
#pragma clang fp contract(off)      // zoom coordinates must be plain IEEE products (see below)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int S = a.W, S2 = S * S;
    // the S x S output tile: in LDS when it fits (thread-owned cells either way: no atomics) — the reference zooms to any odd size
    // (coolpuppy/coolpup.py:1193-1234), so a larger tile lives in the workgroup's stretch of a global scratch buffer (L2-resident)
    const bool tile_in_lds = tile_g == nullptr;
    const size_t tile_stride = (size_t)S2 + ((size_t)S2 + 1) / 2;        // doubles per global tile: sums, then counts
    double*   tsum = tile_in_lds ? reinterpret_cast<double*>(smem_raw) : tile_g + (size_t)blockIdx.x * tile_stride;
    double*   tcov = tile_in_lds ? tsum + S2 : reinterpret_cast<double*>(smem_raw);                               // [2S]
    unsigned* tnum = tile_in_lds ? reinterpret_cast<unsigned*>(tcov + 2 * S) : reinterpret_cast<unsigned*>(tsum + S2);
    // zoom weights of the current window (see below): Wy[S][sep_k] | Wx[S][sep_k] doubles, first input row / column of every output
    // row / column (2S ints)
    double* const wsep = reinterpret_cast<double*>(smem_raw + (tile_in_lds ? (((size_t)S2 * 12 + 16 * (size_t)S + 7) & ~(size_t)7) : 16 * (size_t)S));
    int* const wlo = reinterpret_cast<int*>(wsep + 2 * (size_t)S * (sep_k > 0 ? sep_k : 0));
    const bool emitting = emit != nullptr;
    const int ck = emitting ? (int)blockIdx.x : a.block_chunk[blockIdx.x];
    if (ck < 0) return;
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int t = tid; t < S2; t += nthr) { tsum[t] = 0.0; tnum[t] = 0u; }
    for (int t = tid; t < 2 * S; t += nthr) tcov[t] = 0.0;
    // every thread only ever touches its own cells: no barrier needed before the loop

    const bool m_ooe = a.mode & 0x01u, m_exp = a.mode & 0x02u, m_cov = (a.mode & 0x04u) && a.cov != nullptr;
    const bool m_tr = a.mode & 0x08u, m_local = a.mode & 0x20u;
    const bool use_exp = (m_ooe || m_exp) && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const int igd = a.ignore_diags;
    const double qn = __builtin_nan("");
    const double DBLMAX = 1.7976931348623157e308;
    const long long cb = emitting ? (long long)blockIdx.x : a.chunk_begin[ck];
    const long long ce = emitting ? emit_n : a.chunk_end[ck];
    const long long cstep = emitting ? (long long)gridDim.x : a.chunk_stride[ck];
    const int fl = emitting ? 0 : a.chunk_flip[ck];
    ExpCache ecache;
    RsGeom geo{0, -1, 0, 0, false};

    for (long long s = cb; s < ce; s += cstep) {
        const int rs = a.r0[s], cs = a.c0[s], h = hs[s], w = wsz[s];
        if (rs < 0 || cs < 0 || h < 1 || w < 1 || (long long)rs + h > a.nbins || (long long)cs + w > a.nbins) {
            if (tid == 0) atomicExch(a.err, 1);
            continue;
        }
        if (a.idx != nullptr && !(rs >= geo.ch_start && rs < geo.ch_end)) {
            int lo = 0, hi_k = a.n_chrom;
            while (lo < hi_k) { const int m = (lo + hi_k) >> 1; if (a.idx_chrom[m].end <= rs) lo = m + 1; else hi_k = m; }
            if (lo < a.n_chrom) { const IdxChrom c = a.idx_chrom[lo]; geo = RsGeom{c.start, c.end, c.nblk, c.blk_base, true}; }
            else geo = RsGeom{0, -1, 0, 0, false};
        }
        ExpSel es; es.base = a.expv; es.len = 0; es.scalar = qn; es.is_scalar = true;
        if (use_exp) es = select_expected(a, ecache, rs, cs);

        // masked, normalised window cell (i, j) before symmetrisation; NaN = masked
        auto cell = [&](int i, int j) -> double {
            const int row = rs + i, col = cs + j;
            long long ad = (long long)col - row; if (ad < 0) ad = -ad;
            if (m_exp) return es.at(ad);
            if (bin_bad(a, row) || bin_bad(a, col)) return qn;
            if (igd >= 0 && (col - row) < igd) return qn;
            double v = lookup_bal(a, geo, row, col);
            if (m_ooe) v = v / es.at(ad);
            return v;
        };
        auto cell_sym = [&](int i, int j) -> double {
            double v = cell(i, j);
            if (m_local && h == w) {
                const double u = cell(j, i);                  // square window (same feature on both sides)
                if (v != v) v = u; else if (u == u) v = 0.5 * (v + u);
            }
            return v;
        };
        // ---- round 4: the window is gathered ONCE.  Every thread fetches cells t, t + nthr, ... of the masked / normalised /
        // symmetrised window (independent index lookups: the memory system sees them all at once) into the workgroup's slab
        // of `scratch` (global memory, L2-resident: a window is 10^4 - 10^5 cells); the zoom below then reads its 4 mh mw
        // taps per output cell from there.  Round 3 fetched every tap through the index again — ~12 dependent lookups per
        // window cell, 51 ms per 5 000 TADs.  Same values, same arithmetic: results are bit-identical to the on-demand path,
        // which remains for windows larger than the slab (scratch_cells) or when no slab could be had.
        const bool staged_win = scratch != nullptr && (long long)h * w <= scratch_cells;
        double* const slab = staged_win ? scratch + (size_t)blockIdx.x * (size_t)scratch_cells : nullptr;
        if (staged_win) {
            __syncthreads();                                  // the previous window's taps have been read
            if (m_local && h == w) {
                // the symmetrised window is symmetric: its upper triangle is worked out, every value stored twice.  Rows a and
                // h - 1 - a together hold h + 1 cells of the triangle: the pairs of rows make a ceil(h / 2) x (h + 1) rectangle every
                // lane of which has a cell to fetch
                const int half = (h + 1) / 2, span = h + 1;
                for (int t = tid; t < half * span; t += nthr) {
                    const int ra = t / span, b = t - ra * span;
                    int i, j;
                    if (b < h - ra) { i = ra; j = ra + b; }
                    else { i = h - 1 - ra; j = i + (b - (h - ra)); if (i == ra) continue; }      // (odd h: the middle row pairs with itself)
                    const double v = cell_sym(i, j);
                    slab[i * w + j] = v; slab[j * w + i] = v;
                }
            } else {
                const int hw = h * w;
                for (int t = tid; t < hw; t += nthr) { const int i = t / w; slab[t] = cell_sym(i, t - i * w); }
            }
            __syncthreads();
        }
        auto tap = [&](int i, int j) -> double { return staged_win ? slab[i * w + j] : cell_sym(i, j); };
        // ---- does the window hold any non-NaN cell? ----
        bool all_nan;
        if (m_exp) {
            all_nan = true;
            for (int d = 0; d < h + w - 1 && all_nan; ++d) { long long ad = (long long)(cs - rs) - (h - 1) + d; if (ad < 0) ad = -ad;
                                                             const double e = es.at(ad); if (e == e) all_nan = false; }
        } else {
            int imin = -1, jmax = -1;
            for (int i = 0; i < h && imin < 0; ++i) if (!bin_bad(a, rs + i)) imin = i;
            for (int j = w - 1; j >= 0 && jmax < 0; --j) if (!bin_bad(a, cs + j)) jmax = j;
            all_nan = imin < 0 || jmax < 0 || (igd >= 0 && (cs + jmax) - (rs + imin) < igd);
            if (!all_nan && m_ooe) {
                // data / expected: a cell is NaN when masked, when its expected is NaN, and when it is 0 / 0.  If every
                // expected value on the window's diagonals is usable and non-zero, any unmasked cell is a number;
                // otherwise the workgroup looks at the cells themselves (reference :1213 tests np.all(np.isnan(data)))
                // (the window's h + w - 1 diagonals shared out to the threads: every thread walking all of them, one dependent load
                // after the other, was a quarter of a window's time)
                int e_bad = 0;
                const long long dlo = (long long)(cs - rs) - (h - 1), dhi = (long long)(cs - rs) + (w - 1);
                for (long long d = dlo + tid; d <= dhi; d += nthr) { const double e = es.at(d < 0 ? -d : d); if (!(e == e) || e == 0.0) e_bad = 1; }
                const bool e_good = !__syncthreads_or(e_bad);
                if (!e_good) {
                    int found = 0;
                    // (the symmetrised window holds a number exactly when the plain one does: nanmean of (v, u) is NaN iff both are)
                    for (int t = tid; t < h * w && !found; t += nthr) { const int i = t / w; const double v = staged_win ? slab[t] : cell(i, t - i * w); if (v == v) found = 1; }
                    all_nan = !__syncthreads_or(found);          // h, w, e_good are uniform over the workgroup
                }
            }
        }
        const int mh = S < h ? (h + S - 1) / S : 1, mw = S < w ? (w + S - 1) / S : 1;
        const int th = S * mh, tw = S * mw;
        // scipy's coordinates are plain IEEE products o * ((n_in-1)/(n_out-1)): keep the compiler from fusing the
        // product into the later subtraction (an FMA would make an exactly-integer coordinate look fractional)
        const double sy = th > 1 ? __ddiv_rn((double)(h - 1), (double)(th - 1)) : 0.0;
        const double sx = tw > 1 ? __ddiv_rn((double)(w - 1), (double)(tw - 1)) : 0.0;
        const double inv = 1.0 / ((double)mh * (double)mw);
        // ---- round 4: the zoom as TWO SETS OF WEIGHTS.  Bilinear interpolation and the block mean are both separable, so the value of
        // output cell (A, B) is  inv * sum_i sum_j Wy[A][i] Wx[B][j] v[i][j]  with Wy[A][i] = the summed interpolation weights of input
        // row i over the mh sample rows of output row A (at most mh + 2 input rows: the samples of one output row span < mh of
        // them), Wx likewise — 25-40 window cells per output cell instead of 4 mh mw taps (144 at a threefold reduction), the weights
        // worked out once per window by 2S threads.  A sample outside the input contributes nothing (it gets no weight on its
        // axis: the product form is exact); an input NaN taints the output cell exactly when both its weights are non-zero, which is
        // the per-tap rule below.  Same sums up to the order of the additions.  Windows whose mh + 2 exceeds the LDS room (sep_k)
        // take the per-sample loop.
        const bool sep = sep_k > 0 && !all_nan && mh + 2 <= sep_k && mw + 2 <= sep_k;
        if (sep) {
            __syncthreads();                                  // the previous window's weights have been read
            for (int t = tid; t < 2 * S; t += nthr) {
                const bool ax_y = t < S;
                const int A = ax_y ? t : t - S;
                const int n_in = ax_y ? h : w, m = ax_y ? mh : mw;
                const double sc = ax_y ? sy : sx;
                double* Wr = wsep + (size_t)t * sep_k;
                for (int k = 0; k < sep_k; ++k) Wr[k] = 0.0;
                int lo = -1;
                for (int d = 0; d < m; ++d) {
                    const double y = __dmul_rn((double)(A * m + d), sc);
                    if (y > (double)(n_in - 1)) continue;     // constant-mode sample outside the input: 0
                    const int i0 = (int)y;
                    const double tt = y - (double)i0;
                    if (lo < 0) lo = i0;                      // (coordinates ascend: the first sample's row is the lowest)
                    Wr[i0 - lo] += 1.0 - tt;
                    if (tt > 0.0) Wr[(i0 + 1 < n_in ? i0 + 1 : n_in - 1) - lo] += tt;
                }
                wlo[t] = lo < 0 ? 0 : lo;
            }
            __syncthreads();
        }
        for (int t = tid; t < S2; t += nthr) {
            const int A = t / S, B = t - A * S;
            double acc = 0.0; bool anynan = false;
            if (sep) {
                const double* Wy = wsep + (size_t)A * sep_k;
                const double* Wx = wsep + (size_t)(S + B) * sep_k;
                const int i_lo = wlo[A], j_lo = wlo[S + B];
                for (int ki = 0; ki < mh + 2; ++ki) {
                    const double wy = Wy[ki];
                    const int i = i_lo + ki;
                    if (!(wy > 0.0) || i >= h) continue;
                    double rowacc = 0.0;
                    for (int kj = 0; kj < mw + 2; ++kj) {
                        const double wx = Wx[kj];
                        const int j = j_lo + kj;
                        if (!(wx > 0.0) || j >= w) continue;
                        double v = tap(i, j);
                        if (v != v) { anynan = true; v = 0.0; }
                        v = v > DBLMAX ? DBLMAX : (v < -DBLMAX ? -DBLMAX : v);
                        rowacc += v * wx;
                    }
                    acc += rowacc * wy;
                }
                acc *= inv;
            } else if (!all_nan) {
                for (int da = 0; da < mh; ++da) {
                    const double ya = __dmul_rn((double)(A * mh + da), sy);
                    const bool oob_y = ya > (double)(h - 1);
                    const int i0 = oob_y ? h - 1 : (int)ya;
                    const double ty = ya - (double)i0;
                    const int i1 = i0 + 1 < h ? i0 + 1 : h - 1;
                    double rowacc = 0.0;
                    for (int db = 0; db < mw; ++db) {
                        const double xb = __dmul_rn((double)(B * mw + db), sx);
                        const bool oob = oob_y || xb > (double)(w - 1);
                        if (oob) continue;                       // constant-mode sample outside the input: 0, not NaN
                        const int j0 = (int)xb;
                        const double tx = xb - (double)j0;
                        const int j1 = j0 + 1 < w ? j0 + 1 : w - 1;
                        double v00 = tap(i0, j0), v01 = tx > 0.0 ? tap(i0, j1) : 0.0;
                        double v10 = ty > 0.0 ? tap(i1, j0) : 0.0, v11 = (ty > 0.0 && tx > 0.0) ? tap(i1, j1) : 0.0;
                        // a NaN input taints the sample only when its interpolation weight is non-zero
                        const bool n00 = v00 != v00, n01 = v01 != v01, n10 = v10 != v10, n11 = v11 != v11;
                        if (n00 || n01 || n10 || n11) anynan = true;
                        auto clean = [&](double v) { return v != v ? 0.0 : (v > DBLMAX ? DBLMAX : (v < -DBLMAX ? -DBLMAX : v)); };
                        v00 = clean(v00); v01 = clean(v01); v10 = clean(v10); v11 = clean(v11);
                        rowacc += (v00 * (1.0 - tx) + v01 * tx) * (1.0 - ty) + (v10 * (1.0 - tx) + v11 * tx) * ty;
                    }
                    acc += rowacc;
                }
                acc *= inv;
            }
            if (emitting) {
                emit[(size_t)s * S2 + (m_tr ? B * S + A : t)] = anynan ? qn : acc;
            } else if (!anynan) {
                const int cellidx = t;                          // window frame; flip / transpose applied at the flush
                if (acc == acc) tsum[cellidx] += acc;
                if (!(acc != acc) && !__builtin_isinf(acc)) tnum[cellidx] += 1u;
            }
        }
        if (m_cov && !m_exp) {
            // zoom_array of the coverage vectors (1-D): same sampling, same block mean; NaN coverage adds nothing
            for (int t = tid; t < 2 * S; t += nthr) {
                const bool start_side = t < S;
                const int A = start_side ? t : t - S;
                const bool rows = start_side != m_tr;             // cov_start follows the reference's rows
                const int n_in = rows ? h : w, m = rows ? mh : mw, base = rows ? rs : cs;
                const int n_t = S * m;
                const double sc = n_t > 1 ? __ddiv_rn((double)(n_in - 1), (double)(n_t - 1)) : 0.0;
                double accv = 0.0;
                for (int d = 0; d < m; ++d) {
                    const double y = __dmul_rn((double)(A * m + d), sc);
                    if (y > (double)(n_in - 1)) continue;
                    const int i0 = (int)y; const double tt = y - (double)i0;
                    // both taps are always evaluated (scipy: NaN * 0 = NaN); a tap past the end is the constant 0
                    accv += a.cov[base + i0] * (1.0 - tt) + (i0 + 1 < n_in ? a.cov[base + i0 + 1] * tt : 0.0);
                }
                accv /= (double)m;
                if (emitting) { if (emit_cov) emit_cov[(size_t)s * 2 * S + t] = accv; }
                else if (accv == accv) tcov[t] += accv;
            }
        }
    }
    if (emitting) return;
    // ---- flush (owner threads write their own cells) ----
    const size_t L = (size_t)S2 + 2 * (size_t)S;
    double*   of = a.part_f64 + (size_t)ck * L;
    unsigned* on = a.part_num + (size_t)ck * S2;
    for (int t = tid; t < S2; t += nthr) {
        const int A = t / S, B = t - A * S;
        const int cellidx = map_cell(A, B, S, m_tr, fl);
        of[cellidx] = tsum[t]; on[cellidx] = tnum[t];
    }
    for (int t = tid; t < 2 * S; t += nthr) of[S2 + t] = tcov[t];
}

// ---- per (table, weight column) precomputation ---------------------------------------------------------------
// balanced value of every pixel (one wave per row) — the product PileUpper.get_data() obtains from
// cooler's matrix(balance=w) once per region (coolpup.py:1053-1055), evaluated in the same order
// (count * w[row]) * w[col]; NaN (masked bin) is stored as 0 and masked through badbits instead
PUP_KERNEL __launch_bounds__(256) void balance_pixels_kernel(const long long* __restrict__ indptr, const int2* __restrict__ px,
                                                             const double* __restrict__ weight, double* __restrict__ bal,
                                                             long long nbins, const double* __restrict__ cntf = nullptr) {
    const int lane = threadIdx.x & 63;
    long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; r < nbins; r += stride) {
        const double wr = weight ? weight[r] : 1.0;
        const long long b = indptr[r], e = indptr[r + 1];
        for (long long k = b + lane; k < e; k += 64) {
            const int2 pc = px[k];
            double v = cntf ? cntf[k] : (double)pc.y;        // (float pixel values: cooler multiplies them through the same way)
            if (weight) { v = v * wr * weight[pc.x]; if (!(v == v)) v = 0.0; }
            bal[k] = v;
        }
    }
}

// ---- pixels whose balanced value is not a number although both weights are (weights of +-inf, and 0 * inf) -----------
// cooler multiplies them out all the same and the reference then leaves those cells out of `num` one by one
// (np.isfinite, lib/puputils.py:18-29) while the empty cells of the same rows still count.  The pile-up kernels decide
// validity from the bin masks alone, so such pixels — none at all in a normally balanced table — are listed once per
// weight column (both orientations, sorted by (row, col)) and a small pass after the reduction takes them out of `num`.
PUP_KERNEL __launch_bounds__(256) void collect_nonfinite_kernel(const long long* __restrict__ indptr, const int2* __restrict__ px,
                                                                const double* __restrict__ weight, long long nbins,
                                                                unsigned long long* __restrict__ keys, unsigned long long cap,
                                                                unsigned long long* __restrict__ count, const double* __restrict__ cntf = nullptr) {
    const int lane = threadIdx.x & 63;
    long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; r < nbins; r += stride) {
        const double wr = weight[r];
        if (wr != wr) continue;
        const long long b = indptr[r], e = indptr[r + 1];
        for (long long k = b + lane; k < e; k += 64) {
            const int2 pc = px[k];
            const double wc = weight[pc.x];
            if (wc != wc) continue;
            const double v = (cntf ? cntf[k] : (double)pc.y) * wr * wc;
            if (v == v && !__builtin_isinf(v)) continue;
            const int both = (pc.x != (int)r) ? 2 : 1;
            const unsigned long long at = atomicAdd(count, (unsigned long long)both);
            if (keys != nullptr && at + both <= cap) {
                keys[at] = ((unsigned long long)r << 32) | (unsigned)pc.x;
                if (both == 2) keys[at + 1] = ((unsigned long long)(unsigned)pc.x << 32) | (unsigned long long)r;
            }
        }
    }
}

// one thread per (snippet, window row): the listed pixels under that row are cells the kernels counted as valid
PUP_KERNEL __launch_bounds__(256) void nonfinite_fix_kernel(K1Args a, const unsigned long long* __restrict__ keys, long long nkeys,
                                                            long long n, const long long* __restrict__ tile_ptr,
                                                            const long long* __restrict__ flip_from, int T,
                                                            long long* __restrict__ acc_num) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int W = a.W;
    if (g >= n * W) return;
    const long long s = g / W;
    const int i = (int)(g - s * W);
    const int r0 = a.r0[s], c0 = a.c0[s];
    const unsigned long long klo = ((unsigned long long)(unsigned)(r0 + i) << 32) | (unsigned)c0;
    long long lo = 0, hi = nkeys;
    while (lo < hi) { const long long m = (lo + hi) >> 1; if (keys[m] < klo) lo = m + 1; else hi = m; }
    if (lo >= nkeys || (keys[lo] >> 32) != (unsigned long long)(r0 + i) || (int)(keys[lo] & 0xffffffffu) >= c0 + W) return;
    int t0 = 0, t1 = T;                                              // tile of snippet s
    while (t1 - t0 > 1) { const int m = (t0 + t1) >> 1; if (tile_ptr[m] <= s) t0 = m; else t1 = m; }
    const int fl = (flip_from != nullptr && s >= flip_from[t0]) ? 1 : 0;
    const bool m_ooe = a.mode & 0x01u, m_tr = a.mode & 0x08u;
    const int igd = a.ignore_diags;
    ExpCache ecache;
    ExpSel es; es.base = nullptr; es.len = 0; es.scalar = __builtin_nan(""); es.is_scalar = true;
    if (m_ooe) es = select_expected(a, ecache, r0, c0);
    for (; lo < nkeys; ++lo) {
        const unsigned long long k = keys[lo];
        const int col = (int)(k & 0xffffffffu);
        if ((k >> 32) != (unsigned long long)(r0 + i) || col >= c0 + W) break;
        const int dj = col - (r0 + i);
        if (igd >= 0 && dj < igd) continue;
        if (m_ooe) { const double e = es.at(dj < 0 ? -(long long)dj : (long long)dj); int e_ok = (e == e) && (e != 0.0); asm volatile("" : "+v"(e_ok)); if (!e_ok) continue; }
        const int cell = map_cell(i, col - c0, W, m_tr, fl);
        atomicAdd((unsigned long long*)&acc_num[(size_t)t0 * W * W + cell], ~0ull);      // -1
    }
}

PUP_KERNEL void badbits_kernel(const double* __restrict__ weight, unsigned long long* __restrict__ badbits,
                               long long nbins, long long nwords) {
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    unsigned long long m = 0;
    if (weight)
        for (int b = 0; b < 64; ++b) {
            const long long i = w * 64 + b;
            if (i < nbins && !(weight[i] == weight[i])) m |= 1ull << b;
        }
    badbits[w] = m;
}

// ---- index construction (once per pixel table) ---------------------------------------------------------
// one wave per row: set the presence bit of every cis pixel of the row
PUP_KERNEL __launch_bounds__(256) void index_fill_kernel(const long long* __restrict__ indptr, const int2* __restrict__ px,
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
        unsigned long long* rowwords = reinterpret_cast<unsigned long long*>(idx + c.blk_base + (r - c.start) * c.nblk);
        const long long b = indptr[r], e = indptr[r + 1];
        for (long long k = b + lane; k < e; k += 64) {
            const int col = px[k].x;
            if (col >= c.end) continue;              // trans pixel: not indexed
            const int rel = col - c.start;
            const int blk = rel / kIdxCols, o = rel - blk * kIdxCols;
            atomicOr(rowwords + (size_t)blk * 8 + 2 + (o >> 6), 1ull << (o & 63));   // words: pos, cum, bits[0..4], next0
        }
    }
}

// one thread per row: per block the absolute position of its first pixel, the in-block cumulative counts,
// and the copy of the following block's first word
PUP_KERNEL __launch_bounds__(256) void index_rank_kernel(const long long* __restrict__ indptr,
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
        unsigned run = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (k > 0) row[b].cum[k - 1] = (unsigned short)run;
            run += (unsigned)__popcll(row[b].bits[k]);
        }
        pos += run;
        row[b].next0 = (b + 1 < c.nblk) ? row[b + 1].bits[0] : 0ull;
    }
}

// Segmented, order-fixed reduction of partial records.
//   f64 record length Lf, integer record length Li.  Output record g sums input records
//   [seg_ptr[g], seg_ptr[g+1]).  A workgroup = 64 record elements x kRedParts interleaved partial sums
//   (part y adds records b+y, b+y+kRedParts, ...), combined in the fixed order y = 0..kRedParts-1:
//   parallel, yet the summation order never depends on timing.  ACCUM: add into the output (running accumulators).
constexpr int kRedParts = 16;
template <typename NumIn, bool ACCUM>
__global__ __launch_bounds__(64 * kRedParts) void reduce_partials_kernel(
        const double* __restrict__ in_f64, const NumIn* __restrict__ in_num,
        const long long* __restrict__ seg_ptr, int Lf, int Li,
        double* out_f64, long long* out_num) {
    __shared__ double    sf[kRedParts][64];
    __shared__ long long si[kRedParts][64];
    const int g = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y;
    const int idx = blockIdx.x * 64 + cx;
    const long long b = seg_ptr[g], e = seg_ptr[g + 1];
    double accf = 0.0; long long acci = 0;
    if (idx < Lf) {
        for (long long c = b + py; c < e; c += kRedParts) accf += in_f64[(size_t)c * Lf + idx];
    } else if (idx < Lf + Li) {
        const int k = idx - Lf;
        for (long long c = b + py; c < e; c += kRedParts) acci += (long long)in_num[(size_t)c * Li + k];
    }
    sf[py][cx] = accf; si[py][cx] = acci;
    __syncthreads();
    if (py != 0 || idx >= Lf + Li) return;
    if (idx < Lf) {
        double t = 0.0;
#pragma unroll
        for (int y = 0; y < kRedParts; ++y) t += sf[y][cx];
        double* o = out_f64 + (size_t)g * Lf + idx;
        if (ACCUM) *o += t; else *o = t;
    } else {
        long long t = 0;
#pragma unroll
        for (int y = 0; y < kRedParts; ++y) t += si[y][cx];
        long long* o = out_num + (size_t)g * Li + (idx - Lf);
        if (ACCUM) *o += t; else *o = t;
    }
}

// The same reduction for calls of MANY tiles with a handful of records each (by-window pile-ups: 7e4 tiles, one or two chunks per
// tile): a thread per record element walks the tile's records in order — no 16-way split, no LDS stage.  With the kernel above
// such a call launched 1.1e6 workgroups of 1024 threads to add up 1.5 records each (2.5 ms; this one: see DESIGN).
template <typename NumIn>
__global__ __launch_bounds__(256) void reduce_partials_small_kernel(
        const double* __restrict__ in_f64, const NumIn* __restrict__ in_num, const long long* __restrict__ seg_ptr, int Lf, int Li,
        double* out_f64, long long* out_num) {
    const int g = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const long long b = seg_ptr[g], e = seg_ptr[g + 1];
    if (b >= e) return;
    if (idx < Lf) {
        double acc = 0.0;
        for (long long c = b; c < e; ++c) acc += in_f64[(size_t)c * Lf + idx];
        out_f64[(size_t)g * Lf + idx] += acc;
    } else if (idx < Lf + Li) {
        const int k = idx - Lf;
        long long acc = 0;
        for (long long c = b; c < e; ++c) acc += (long long)in_num[(size_t)c * Li + k];
        out_num[(size_t)g * Li + k] += acc;
    }
}

// ---- K3: per-bin coverage (marginal sums of raw counts) ------------------------------------------------------
// What coolpuppy obtains from cooltools.api.coverage.coverage(clr, ignore_diags=..., store=True) when the bins
// table lacks cov_cis_raw / cov_tot_raw (coolpup.py:955-963): every pixel adds its count to BOTH of its bins
// (a main-diagonal pixel therefore twice), pixels with |bin2 - bin1| < ignore_diags count as 0, the cis variant
// only uses pixels whose bins share a chromosome.  One streaming pass over the pixel table (8 B per pixel):
// a workgroup owns kCovRows consecutive rows; row sums are wave-reduced, column sums go to an LDS histogram over
// the kCovCols columns following the block's first row (where almost all cis mass lies) and to global integer
// atomics beyond it.  Integer (u64) accumulation: exact and order-independent.
constexpr int kCovRows = 64;
constexpr int kCovCols = 4096;
// cov_cis accumulates intra-chromosomal pixels, cov_trans inter-chromosomal ones (one atomic per pixel either way);
// the host forms cov_tot = cov_cis + cov_trans.  A wave streams its row with 16-byte loads (two pixels per lane), two
// loads in flight per lane: the pass is bound by load latency, not by the LDS atomics (the distinct columns of one
// row never collide; four loads in flight per lane and a 2048-column window were measured: 0.86 ms and 2.1 ms against 0.78).  Columns beyond the block's LDS window and all trans columns take global atomics.
// Round 6: the same pass over FLOAT pixel values (a float pixels/count column, pup_load_pixel_values): Acc = double, the value of
// pixel k comes from cntf[k], the sums are f64 atomics (LDS and global, hardware adds on gfx950) — the row sums stay a fixed-order
// wave reduction, the column sums depend on arrival order in their last bits (cooltools sums floats in table order; 1e-6 is the bar).
template <typename Acc>
struct CovOps;
template <>
struct CovOps<unsigned long long> {
    static constexpr bool kFloat = false;
    static __device__ __forceinline__ void add(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
    static __device__ __forceinline__ bool nonzero(unsigned long long v) { return v != 0ull; }
};
template <>
struct CovOps<double> {
    static constexpr bool kFloat = true;
    static __device__ __forceinline__ void add(double* p, double v) { unsafeAtomicAdd(p, v); }
    static __device__ __forceinline__ bool nonzero(double v) { return v != 0.0; }
};

template <typename Acc>
PUP_KERNEL __launch_bounds__(512) void coverage_kernel(const long long* __restrict__ indptr, const int2* __restrict__ px,
                                                       const double* __restrict__ cntf,
                                                       const IdxChrom* __restrict__ chroms, int n_chrom, int ignore_diags,
                                                       Acc* cov_trans, Acc* cov_cis, long long nbins) {
    constexpr bool FLT = CovOps<Acc>::kFloat;
    __shared__ Acc h_cis[kCovCols];
    const long long row0 = (long long)blockIdx.x * kCovRows;
    for (int t = threadIdx.x; t < kCovCols; t += blockDim.x) h_cis[t] = Acc(0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (long long r = row0 + wave; r < row0 + kCovRows && r < nbins; r += nwave) {
        int lo = 0, hi = n_chrom;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (chroms[m].end <= r) lo = m + 1; else hi = m; }
        const long long chrom_end = lo < n_chrom ? chroms[lo].end : nbins;
        Acc s_trans = Acc(0), s_cis = Acc(0);
        const long long b = indptr[r], e = indptr[r + 1];
        auto add = [&](long long k, int col, int cnt) __attribute__((always_inline)) {
            if (k < b || k >= e) return;
            const long long d = (long long)col - r;
            Acc w;
            if constexpr (FLT) w = cntf[k]; else w = (Acc)(unsigned)cnt;
            if ((d < 0 ? -d : d) < ignore_diags) w = Acc(0);
            if (col >= chrom_end) { s_trans += w; CovOps<Acc>::add(&cov_trans[col], w); return; }
            s_cis += w;
            const long long rel = (long long)col - row0;
            if (rel < kCovCols) CovOps<Acc>::add(&h_cis[rel], w);
            else CovOps<Acc>::add(&cov_cis[col], w);
        };
        // pixel pairs at even offsets (16-byte aligned); the table is padded, so the pair straddling e is readable
        for (long long k = (b & ~1LL) + 2 * lane; k < e; k += 256) {
            const int4 v0 = *reinterpret_cast<const int4*>(px + k);
            const bool two = k + 128 < e;
            const int4 v1 = two ? *reinterpret_cast<const int4*>(px + k + 128) : int4{0, 0, 0, 0};
            add(k, v0.x, v0.y); add(k + 1, v0.z, v0.w);
            if (two) { add(k + 128, v1.x, v1.y); add(k + 129, v1.z, v1.w); }
        }
        for (int off = 32; off > 0; off >>= 1) { s_trans += __shfl_down(s_trans, off); s_cis += __shfl_down(s_cis, off); }
        if (lane == 0) {
            CovOps<Acc>::add(&h_cis[r - row0], s_cis);
            if (CovOps<Acc>::nonzero(s_trans)) CovOps<Acc>::add(&cov_trans[r], s_trans);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kCovCols; t += blockDim.x) {
        const long long c = row0 + t;
        if (c < nbins && CovOps<Acc>::nonzero(h_cis[t])) CovOps<Acc>::add(&cov_cis[c], h_cis[t]);
    }
}

// ---- K4: per-snippet stripes (store_stripes) ---------------------------------------------------------------
// The centre row and the centre column of every snippet, masked and normalised exactly like the window itself
// (coolpup.py:1164-1169): horizontal = data[pad, :], vertical = data[:, pad][::-1].  Output is O(n*W), not a
// reduction: one wave per snippet, lanes 0..W-1 look up the row cells, lanes W..2W-1 the column cells, each by a
// binary search of its matrix row (2W cells per snippet: the index would save nothing worth its code here).
__device__ __forceinline__ double find_count(const K1Args& a, int row, int col, bool& found) {
    long long lo = a.indptr[row], hi = a.indptr[row + 1];
    const long long end = hi;
    while (lo < hi) { const long long m = (lo + hi) >> 1; if (a.px[m].x < col) lo = m + 1; else hi = m; }
    found = lo < end && a.px[lo].x == col;
    return found ? (a.cntf ? a.cntf[lo] : (double)a.px[lo].y) : 0.0;      // (float pixel values: pup_load_pixel_values)
}

PUP_KERNEL __launch_bounds__(kWave) void stripes_kernel(K1Args a, long long n, double* __restrict__ h_out,
                                                       double* __restrict__ v_out) {
    const int W = a.W, pad = W / 2;
    const int lane = threadIdx.x;
    const bool m_ooe = a.mode & 0x01u, m_tr = a.mode & 0x08u;
    const bool use_exp = m_ooe && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const double qn = __builtin_nan("");
    ExpCache ecache;
    for (long long s = blockIdx.x; s < n; s += gridDim.x) {
        const int r0s = a.r0[s], c0s = a.c0[s];
        if (r0s < 0 || c0s < 0 || (long long)r0s + W > a.nbins || (long long)c0s + W > a.nbins) {
            if (lane == 0) atomicExch(a.err, 1);
            continue;
        }
        ExpSel es; es.base = a.expv; es.len = 0; es.scalar = qn; es.is_scalar = true;
        if (use_exp) es = select_expected(a, ecache, r0s, c0s);
        for (int t = lane; t < 2 * W; t += kWave) {
            // engine frame: rows from r0s, columns from c0s.  In the reference's frame (after undoing TRANSPOSE)
            // the horizontal stripe runs along its columns and the vertical one, reversed, along its rows.
            const bool horiz = t < W;
            const int i = horiz ? t : t - W;
            int p, q;
            if (!m_tr) { p = horiz ? pad : (W - 1 - i); q = horiz ? i : pad; }
            else       { p = horiz ? i : pad;           q = horiz ? pad : (W - 1 - i); }
            const int row = r0s + p, col = c0s + q;
            bool found;
            double v = find_count(a, row, col, found);
            if (a.weight) {
                const double wr = a.weight[row], wc = a.weight[col];
                // cooler multiplies stored pixels out (a stored 0 next to an infinite weight is NaN); cells without a
                // stored pixel are 0 whatever the weights
                v = (wr == wr && wc == wc) ? (found ? v * wr * wc : 0.0) : qn;
            }
            if (a.ignore_diags >= 0 && (col - row) < a.ignore_diags) v = qn;
            if (m_ooe) { long long ad = (long long)col - row; if (ad < 0) ad = -ad; v = v / es.at(ad); }
            (horiz ? h_out : v_out)[s * W + i] = v;
        }
    }
}

// ---- K6: per-snippet windows (pup_extract) ------------------------------------------------------------------------
// The W x W window of every snippet exactly as PileUpper._stream_snips yields it (coolpup.py:1104-1158): balanced
// values, NaN on masked bins and ignored diagonals, divided by expected (OOE), or the expected window itself
// (EXPECTED, the reference's exp_snip); written in the reference's frame (TRANSPOSE undone, no flip — the flip is a
// per-snippet post-processing step in the reference, coolpup.py:128-131).  Exists for the per-snippet Python
// callbacks (postprocess_func / extra_sum_funcs), whose cost per snippet dwarfs this gather: one 256-thread
// workgroup per snippet, every cell looked up on its own through the rank-bitmap index (or a binary search).
PUP_KERNEL __launch_bounds__(256) void extract_windows_kernel(K1Args a, long long n, double* __restrict__ out,
                                                              double* __restrict__ cov_out) {
    const int W = a.W, W2 = W * W;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const bool m_ooe = a.mode & 0x01u, m_exp = a.mode & 0x02u, m_cov = (a.mode & 0x04u) && a.cov != nullptr;
    const bool m_tr = a.mode & 0x08u;
    const bool use_exp = (m_ooe || m_exp) && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const int igd = a.ignore_diags;
    const double qn = __builtin_nan("");
    ExpCache ecache;
    RsGeom geo{0, -1, 0, 0, false};
    for (long long s = blockIdx.x; s < n; s += gridDim.x) {
        const int rs = a.r0[s], cs = a.c0[s];
        if (rs < 0 || cs < 0 || (long long)rs + W > a.nbins || (long long)cs + W > a.nbins) {
            if (tid == 0) atomicExch(a.err, 1);
            continue;
        }
        if (a.idx != nullptr && !(rs >= geo.ch_start && rs < geo.ch_end)) {
            int lo = 0, hi_k = a.n_chrom;
            while (lo < hi_k) { const int m = (lo + hi_k) >> 1; if (a.idx_chrom[m].end <= rs) lo = m + 1; else hi_k = m; }
            if (lo < a.n_chrom) { const IdxChrom c = a.idx_chrom[lo]; geo = RsGeom{c.start, c.end, c.nblk, c.blk_base, true}; }
            else geo = RsGeom{0, -1, 0, 0, false};
        }
        ExpSel es; es.base = a.expv; es.len = 0; es.scalar = qn; es.is_scalar = true;
        if (use_exp) es = select_expected(a, ecache, rs, cs);
        for (int t = tid; t < W2; t += nthr) {
            const int i = t / W, j = t - i * W;
            const int row = rs + i, col = cs + j;
            long long ad = (long long)col - row; if (ad < 0) ad = -ad;
            double v;
            if (m_exp) v = es.at(ad);
            else if (bin_bad(a, row) || bin_bad(a, col)) v = qn;
            else if (igd >= 0 && (col - row) < igd) v = qn;
            else {
                v = lookup_bal(a, geo, row, col);
                if (m_ooe) v = v / es.at(ad);
            }
            out[(size_t)s * W2 + (m_tr ? j * W + i : t)] = v;
        }
        if (cov_out != nullptr) {
            for (int t = tid; t < 2 * W; t += nthr) {
                const bool start_side = t < W;
                const int k = start_side ? t : t - W;
                const bool rows = start_side != m_tr;            // cov_start follows the reference's rows
                cov_out[(size_t)s * 2 * W + t] = m_cov ? a.cov[(rows ? rs : cs) + k] : qn;
            }
        }
    }
}

// out[i] = src[pos[i]] (first-snippet rows of the launch groups when the snippets already live on the device)
PUP_KERNEL void gather_int_kernel(const int* __restrict__ src, const long long* __restrict__ pos, int* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[pos[i]];
}

// small clears on the hot path (counters of a call, the accumulators of a reset: 10-20 KB): the runtime's fill kernel takes ~7 us per
// call for these, a plain kernel ~2
PUP_KERNEL __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ p, long long n_words) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) p[i] = 0u;
}

// n[t] += dn[t]
PUP_KERNEL void add_counts_kernel(long long* n, const long long* dn, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) n[t] += dn[t];
}

// interleave bin2/count into {col,count} pairs (upload helper), 64-bit or 32-bit column ids
// dense band of counts (staged kernel, see pup_staged.hpp): one wave per row copies the row's pixels with
// col - row < band_w (they are the first ones of the row: columns ascend) to band[row * band_w + (col - row)]
PUP_KERNEL __launch_bounds__(256) void band_fill_kernel(const long long* __restrict__ indptr, const int2* __restrict__ px,
                                                        int* __restrict__ band, int band_w, long long nbins) {
    const int lane = threadIdx.x & 63;
    long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (; r < nbins; r += stride) {
        const long long b = indptr[r], e = indptr[r + 1];
        for (long long k0 = b; k0 < e; k0 += 64) {
            const long long k = k0 + lane;
            bool more = false;
            if (k < e) {
                const int2 pc = px[k];
                const long long j = (long long)pc.x - r;
                if (j >= 0 && j < band_w) { band[r * band_w + j] = pc.y; more = true; }
            }
            if (__ballot(more) == 0ull) break;           // (uniform) the rest of the row lies right of the band
        }
    }
}

// the same from the (tile, flip) run boundaries {flip_from | tile end, tile end} per tile (the staged path's host table)
PUP_KERNEL void add_counts_from_ends_kernel(long long* n, const long long* seg_end, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) n[t] += seg_end[2 * t + 1] - (t ? seg_end[2 * t - 1] : 0);
}

template <typename ColT>
__global__ void pack_pixels_kernel(const ColT* __restrict__ col, const int* __restrict__ cnt,
                                   int2* __restrict__ out, int* __restrict__ out_cnt, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { const int c = cnt[i]; out[i] = make_int2((int)col[i], c); out_cnt[i] = c; }
}

}  // namespace pup
