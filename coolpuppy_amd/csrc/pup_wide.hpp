// pup_wide.hpp — K1w, the workgroup-staged pile-up kernel for WIDE cis windows (W >= 32, no upper limit).  gfx950 only.
//
// K1q (pup_staged.hpp) stops at 31-bin windows: its lanes own whole window rows side by side (64 / W rows per wave) and a
// 128 x 128 region holds too few corners of wider windows.  The per-window kernel that served wider windows (K1b) fetches
// every window on its own: 8e6 windows of 51 x 51 pulled 66 GB from HBM against 1.2 GB of compulsory bytes (round 3
// counters).  K1w applies the staging idea to windows of ANY width by cutting a W x W window into SUB-WINDOWS of at most
// 64 rows x 52 columns (a grid of NGr x NGc groups): (window, group) items are block-sorted on the device like K1q's
// windows — the key carries the group, the corner is the sub-window's — and a persistent 16-wave workgroup stages the
// 128 x 128 region under a block of sub-window corners ONCE, as final cell values (coolpuppy/coolpup.py:1104-1157 makes
// a cell's value a function of its absolute (row, col) only), from the dense band of counts (pup_build_index).  Inside the
// workgroup a LANE owns a sub-window ROW and a WAVE one of NPC column PANELS of CH <= 13 cells (wave w: panel w % NPC);
// the NW / NPC waves of a panel share the block's windows.  Per window and wave: one readlane, one address add, CH
// ds_read_b64 (row stride 129 doubles: the 64 rows of a wave hit 64 different bank pairs) and CH f64 adds.
// What the reference does per snippet (dense slice of any size, coolpuppy/coolpup.py:1115-1121; masks :1122-1149; division by
// expected :1154-1156; nansum / isfinite accumulation coolpuppy/lib/puputils.py:12-41) is the same arithmetic as in K1q:
//   FACT   every window of the call clears the masked diagonals (and the expected is usable wherever a window reaches):
//          validity of cell (p, q) = !rowbad[p] & !colbad[q], so num[p][q] = N - R[p] - C[q] + RC[p][q] from per-batch
//          mask words (integer LDS atomics, exact) and the window loop touches values only;
//   !FACT  one validity bit per staged cell rides along (vbits), read with the window's row.
// Partial records are 64 x 64 sub-window tiles: record id = (tile, flip, group) number + workgroup number — a workgroup's
// block range is contiguous and the blocks are sorted by (tile, flip, group), so that sum is unique — and
// reduce_wide_kernel sums the valid ones per accumulator cell in fixed order (bit-reproducible).
// Same integers as every other kernel of the engine; sums differ by the order of the f64 additions only.
#pragma once
#include "pup_staged.hpp"

namespace pup {

constexpr int kWideRec = 64 * 64;                     // cells of a partial record: sub-window row p, column q at p * 64 + q
constexpr int kWideBlockCostCells = 1400;             // staging one region, in cells per lane of items' worth of time: a block costs 1400 / CH items
                                                      // (workgroup ranges; wide_run: ~9 k clocks per block — store burst, two barriers, look-ahead)
constexpr int kWideMaxRows = 64;                      // sub-window rows: a lane each
constexpr int kWideMaxCH = 13;                        // cells per lane: register budget of 16 waves x 128 VGPRs

struct WideArgs {
    const StagedBlock*    blocks;     // block table (pad[0] = sub-window group of the block's items)
    const unsigned short* win;        // items in block order: corner of the sub-window inside its region, dr | dc << 7
    const int*            wg_first;   // [G + 1] first block of every workgroup's range
    int                   WF;         // full window width
    int                   NGc;        // groups per row of the group grid (group = gi * NGc + gj), NG = NGr * NGc
    int                   NG;
    int                   SH, SW;     // nominal sub-window height / width (the last row / column of groups may be smaller)
    int                   NPC;        // column panels (divides 16): wave w piles up columns [CH * (w % NPC), + CH)
    unsigned*             rec_seg;    // [nrec] 0, or 1 + (tile * 2 + flip) * NG + group of the record written there
    double*               rec_f64;    // [nrec][kWideRec]
    unsigned*             rec_num;    // [nrec][kWideRec]
    long long*            timing;     // phase clocks per wave, [G][16][8], or nullptr
    int                   share[3];   // cumulative shares (in 1/256) of a block's windows of the first three waves of a panel
};

// geometry of a call, shared by host and device: sub-window grid and lane mapping for window width W.
// A sub-window of sh x sw cells is dealt to the 4 x 64 lanes of the NPC = 4 waves that pile it up together as SLOTS: slot
// s = 64 * panel + lane owns row p = s / NCH and the CH cells (p, k + NCH * i), k = s % NCH, i < CH — NCH interleaved column chunks
// per row, as K1q deals a row to its lanes.  (Round 4 began with lane = row, wave = contiguous column panel — NCH = 4 with rows up to
// 64: a 41-bin window kept 41 of 64 lanes busy, a 51-bin window 51.  The shapes below fill 94 % and 92 % of the lanes for those:
// (CH 7, NCH 6) is 42 rows x 42 columns, (CH 11, NCH 5) 51 x 55, (CH 7..13, NCH 4) 64 x 28..52.)  The cost of a window in LDS
// instructions is groups x 4 x CH — but what a (window, group) item costs beside its cells (its share of the prepass and of the
// staged blocks) weighs like ~100 of them: the FEWEST GROUPS win, the cheapest shape among those.  (201-bin windows as 5 x 5 groups
// of 41 x 41 instead of 4 x 4 of 51 x 51 — 700 against 704 instructions — took 29 ms instead of 22.)
struct WideGeom { int NGr, NGc, SH, SW, NPC, CH, NCH, shape; };
constexpr int kWideShapes = 9;
__host__ __device__ constexpr int wide_shape_ch(int k) { return k < 7 ? 7 + k : (k == 7 ? 11 : 7); }
__host__ __device__ constexpr int wide_shape_nch(int k) { return k < 7 ? 4 : (k == 7 ? 5 : 6); }
__host__ __device__ inline WideGeom wide_geometry(int W) {
    WideGeom best{1, 1, W, W, 4, 13, 4, 6};
    long long best_cost = -1;
    for (int k = 0; k < kWideShapes; ++k) {
        const int CH = wide_shape_ch(k), NCH = wide_shape_nch(k);
        const int max_rows = (4 * 64) / NCH < kWideMaxRows ? (4 * 64) / NCH : kWideMaxRows, max_cols = CH * NCH;
        const int NGr = (W + max_rows - 1) / max_rows, NGc = (W + max_cols - 1) / max_cols;
        const long long cost = (long long)NGr * NGc * 4 * CH, ng = (long long)NGr * NGc;
        const long long best_ng = (long long)best.NGr * best.NGc;
        if (best_cost < 0 || ng < best_ng || (ng == best_ng && cost < best_cost)) {
            best_cost = cost;
            best.NGr = NGr; best.NGc = NGc; best.SH = (W + NGr - 1) / NGr; best.SW = (W + NGc - 1) / NGc; best.NPC = 4; best.CH = CH; best.NCH = NCH; best.shape = k;
        }
    }
    return best;
}

// one step of K1w's rolling window pipeline per cell: wait for the oldest outstanding LDS read (cell I of the current window),
// add it, and reissue the register as the destination of cell I of the next window
template <int I, int N, int NB, bool FACT, int NCH>
struct RollRow {
    static __device__ __forceinline__ void go(double (&v)[N], double (&sum)[N], unsigned (&num)[N], unsigned long long vw, unsigned ad, unsigned adn) {
        lds_wait_but<NB - 1>(ad, adn);
        lds_pin1(v[I]);
        sum[I] += v[I];
        if (!FACT) num[I] += (unsigned)(vw >> (NCH * I)) & 1u;
        lds_read_b64<8 * NCH * I>(v[I], adn);
        RollRow<I + 1, N, NB, FACT, NCH>::go(v, sum, num, vw, ad, adn);
    }
};
template <int N, int NB, bool FACT, int NCH>
struct RollRow<N, N, NB, FACT, NCH> {
    static __device__ __forceinline__ void go(double (&)[N], double (&)[N], unsigned (&)[N], unsigned long long, unsigned, unsigned) {}
};

// the instantiations (the kWideShapes lane shapes x observed-over-expected x factorised counts) live in their own translation
// units, one per shape (pup_wide_tu.hip with -DPUP_TU_PART=shape), compiled side by side like K1q's
#define PUP_WIDE_PART_DECL(k) bool launch_wide_part##k(const K1Args&, const WideArgs&, int G, bool ooe, bool fact, hipStream_t);
PUP_WIDE_PART_DECL(0) PUP_WIDE_PART_DECL(1) PUP_WIDE_PART_DECL(2) PUP_WIDE_PART_DECL(3) PUP_WIDE_PART_DECL(4) PUP_WIDE_PART_DECL(5)
PUP_WIDE_PART_DECL(6) PUP_WIDE_PART_DECL(7) PUP_WIDE_PART_DECL(8)
#undef PUP_WIDE_PART_DECL
inline bool launch_wide(int shape, const K1Args& a, const WideArgs& wa, int G, bool ooe, bool fact, hipStream_t s) {
    switch (shape) {
        case 0: return launch_wide_part0(a, wa, G, ooe, fact, s);
        case 1: return launch_wide_part1(a, wa, G, ooe, fact, s);
        case 2: return launch_wide_part2(a, wa, G, ooe, fact, s);
        case 3: return launch_wide_part3(a, wa, G, ooe, fact, s);
        case 4: return launch_wide_part4(a, wa, G, ooe, fact, s);
        case 5: return launch_wide_part5(a, wa, G, ooe, fact, s);
        case 6: return launch_wide_part6(a, wa, G, ooe, fact, s);
        case 7: return launch_wide_part7(a, wa, G, ooe, fact, s);
        case 8: return launch_wide_part8(a, wa, G, ooe, fact, s);
        default: return false;
    }
}

#ifndef PUP_WIDE_NW
#define PUP_WIDE_NW 16
#endif
constexpr int kWideWaves = PUP_WIDE_NW;               // waves of K1w's workgroup (experiments: 8 — twice the registers per wave)
template <int CH, int NCH, bool OOE, bool FACT>
__global__ __launch_bounds__(kWave * kWideWaves, 1)
void pileup_wide_kernel(K1Args a, WideArgs wa) {
    static_assert(CH >= 1 && CH <= kWideMaxCH, "cells per lane");
    static_assert(NCH >= 1 && NCH <= 8 && NCH * (CH - 1) < 64, "column chunks per row; a lane's validity bits fit one 64-bit word");
    // row stride LS = RSC + NCH doubles: slot s reads double p * LS + k + NCH * i = s + NCH * i (mod 32 bank pairs) — the 32 lanes of
    // a half wave hit 32 different bank pairs
    constexpr int RSR = 128, RSC = 128, NW = kWideWaves, LS = RSC + NCH, RPW = RSR / NW, NH = 2, NRH = RPW * NH, VBW = 3, NTHR = kWave * NW;
    static_assert((size_t)(NW / 2) * CH * kWave * 12 <= (size_t)RSR * LS * 8, "merge scratch must fit the region buffer");
    __shared__ double tile[RSR * LS + 16];                          // (+16: the last panel's unowned cells may run past the last row)
    __shared__ unsigned long long vbits[FACT ? 1 : RSR * VBW + 2];  // bit c of row r: cell (r, c) counts in num (+2: the four-dword read of the last row)
    __shared__ double exp_lds[OOE ? 256 : 1];                       // expected of the region's 255 diagonals (see K1q)
    __shared__ unsigned rc_lds[FACT ? kWideRec : 1];                // masked row meets masked column (rare)
    __shared__ unsigned fact_tot[FACT ? 2 * 64 + 4 : 1];            // R[64] | C[64] | N
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NPC = wa.NPC, nsub = NW / NPC;
    const int panel = wave % NPC, sub = wave / NPC;
    const int slot = panel * kWave + lane;               // the lane's slot of the sub-window: row slot / NCH, column chunk slot % NCH
    const int p_raw = slot / NCH, kk = slot - p_raw * NCH;

    const bool use_exp = OOE && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const int  igd     = a.ignore_diags;
    const double qnan = __builtin_nan("");
    const bool nf = a.nf_pixels != 0;                    // (uniform) weights of +-inf in the table: NaN products are stored as 0

    const int G = (int)gridDim.x, g_id = (int)blockIdx.x;
    const int bb = wa.wg_first[g_id], be = wa.wg_first[g_id + 1];
    if (bb >= be) return;                                // (uniform) more workgroups than blocks

    // geometry of the current sub-window group (changes with the segment)
    int grp = -1, pr0 = 0, pc0 = 0, sh = 1, sw = 1, p = 0;
    bool row_ok = false;
    unsigned chmask = 0u;                                // bit i: the lane owns sub-window cell (p, kk + NCH * i)
    unsigned lane_off8 = 0u;                             // LDS byte address of the lane's first cell at corner (0, 0)
    auto set_group = [&](int g) __attribute__((always_inline)) {
        grp = g;
        const int gi = g / wa.NGc, gj = g - gi * wa.NGc;
        pr0 = gi * wa.SH; pc0 = gj * wa.SW;
        sh = wa.WF - pr0 < wa.SH ? wa.WF - pr0 : wa.SH;
        sw = wa.WF - pc0 < wa.SW ? wa.WF - pc0 : wa.SW;
        row_ok = p_raw < sh;
        p = row_ok ? p_raw : sh - 1;                     // idle lanes shadow the last row, flush nothing
        chmask = 0u;
#pragma unroll
        for (int i = 0; i < CH; ++i) if (row_ok && kk + NCH * i < sw) chmask |= 1u << i;
        lane_off8 = (unsigned)(uintptr_t)tile + 8u * (unsigned)(p * LS + kk);
    };

    double   sum[CH];
    unsigned num[CH];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) { sum[i] = 0.0; num[i] = 0u; }
        if constexpr (FACT) {                            // visible after the next barrier
            for (int t = tid; t < kWideRec; t += NTHR) rc_lds[t] = 0u;
            for (int t = tid; t < 2 * 64 + 4; t += NTHR) fact_tot[t] = 0u;
        }
    };
    zero_acc();
    if constexpr (!FACT) for (int t = tid; t < RSR * VBW; t += NTHR) vbits[t] = 0ull;     // the pad words stay zero

    const StagedBlock* __restrict__ blocks = wa.blocks;
    auto entry_load = [&](int b) __attribute__((always_inline)) -> int {
        return reinterpret_cast<const int*>(blocks + b)[lane & 31];
    };
    auto fld = [&](int ev, int i) __attribute__((always_inline)) -> int { return __builtin_amdgcn_readlane(ev, i); };
    auto fld64 = [&](int ev, int i) __attribute__((always_inline)) -> unsigned long long {
        return ((unsigned long long)(unsigned)fld(ev, i + 1) << 32) | (unsigned)fld(ev, i);
    };
    auto bcast64 = [&](unsigned long long v, int i) __attribute__((always_inline)) -> unsigned long long {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, i), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), i);
        return ((unsigned long long)hi << 32) | lo;
    };
    const int my_rr = wave * RPW + (lane < RPW ? lane : 0);      // the region row whose weight / validity this lane looks after
    auto weight_of = [&](double w) __attribute__((always_inline)) -> double {
        return a.weight ? ((nf || w == w) ? w : 0.0) : 1.0;
    };
    auto exp_of = [&](int ev) -> ExpSel {
        ExpSel es; es.base = a.expv; es.len = 0; es.scalar = qnan; es.is_scalar = true;
        if (!use_exp) return es;
        if (a.n_exp_regions <= 0) {
            es.len = a.nexp; es.is_scalar = (a.nexp == 1);
            es.scalar = (a.nexp == 1 && a.expv) ? a.expv[0] : qnan;
            if (!a.expv || a.nexp <= 0) { es.is_scalar = true; es.scalar = qnan; }
            return es;
        }
        const int er = fld(ev, 5);                       // expected region of the block (that of its origin: see wide_run)
        es.is_scalar = false;
        if (er >= 0 && er < a.n_exp_regions) { const ExpRegion g = a.exp_regions[er]; es.base = a.expv + g.off; es.len = g.len; }
        return es;
    };
    auto exp_entry = [&](double e) __attribute__((always_inline)) -> double { return FACT ? 1.0 / e : e; };     // (the reciprocal: see K1q)
    auto exp_fetch = [&](int ev) __attribute__((always_inline)) -> double {
        if (!use_exp || tid >= RSR + RSC - 1) return qnan;
        const ExpSel es = exp_of(ev);
        long long d = (long long)fld(ev, 1) - (long long)fld(ev, 0) - (RSR - 1) + tid;
        if (d < 0) d = -d;
        return es.at(d);
    };
    // unmasked cells of (region row rr, half h): column mask of the table entry, masked-row bit, diagonal mask — uniform
    auto ok_mask = [&](int ev, int rr, int h, bool row_bad) __attribute__((always_inline)) -> unsigned long long {
        unsigned long long ok = row_bad ? 0ull : fld64(ev, 12 + 2 * h);
        if (igd >= 0) {
            const int t0 = igd - (fld(ev, 1) + 64 * h - (fld(ev, 0) + rr));   // column C + 64 h + l is on or above the first kept diagonal iff l >= t0
            ok &= t0 <= 0 ? ~0ull : (t0 >= 64 ? 0ull : ~((1ull << t0) - 1ull));
        }
        return ok;
    };

    // ---- staging from the dense band of counts: band[row][j] = count(row, row + j) ------------------------------------
    // A region row is 128 consecutive counts of the band: one coalesced load per 64-column half whose address is the row's
    // (scalar, 64-bit) base plus 4 * lane.  FACT: no cell a window reads needs masking (every window clears the masked
    // diagonals and lies inside the band; masked bins multiply to zero through their weight), so the loads carry no
    // predicate; cells left of the diagonal or past the band's width read the neighbouring rows / the pads around the
    // table: finite garbage nobody looks at.  !FACT: lanes whose cell is masked, below the first kept diagonal or outside
    // the staged rows read the zeros behind the band.  Nothing here looks at a loaded value (see K1q).
    const int* const zeros = a.band + (long long)a.nbins * a.band_w;      // >= 129 rows of zeros behind the last row
    auto band_issue = [&](int ev, int (&v)[NRH], double (&wc)[NH], double& wrv) __attribute__((always_inline)) {
        const int R = fld(ev, 0), C = fld(ev, 1), ch_end = fld(ev, 6), row_lo = fld(ev, 21), row_hi = fld(ev, 22);
        const int hi2 = (ch_end - R) < row_hi ? (ch_end - R) : row_hi;             // live rows of the region: [row_lo, hi2)
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int rr = wave * RPW + i;
            const int row = R + rr;
            const bool live = rr >= row_lo && rr < hi2;                           // (uniform)
            const int* rowp = live ? a.band + (long long)row * a.band_w + (C - row) : zeros;
            if constexpr (FACT) {
                // a lane takes the NEIGHBOURING columns 2 l and 2 l + 1 of the row — one 8-byte load per row and, at store time, one
                // ds_write2_b64 — half the memory and LDS instructions of the staging (as K1q since round 5)
                int2 two;
                __builtin_memcpy(&two, rowp + 2 * lane, sizeof(two));      // (4-byte aligned: global_load_dwordx2)
                v[i * NH] = two.x; v[i * NH + 1] = two.y;
            } else {
                const bool row_bad = (fld64(ev, 16 + 2 * (rr >> 6)) >> (rr & 63)) & 1ull;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    // cells past the band's width are in no window (the engine checked); the band's own row ends at band_w
                    unsigned long long keep = live ? ok_mask(ev, rr, h, row_bad) : 0ull;
                    const int over = (C + 64 * h + 64) - (row + a.band_w);          // columns at / past the band's end
                    if (over > 0) keep &= over >= 64 ? 0ull : (~0ull >> over);
                    const bool has = __builtin_amdgcn_inverse_ballot_w64(keep);
                    const int* ptr = has ? rowp + 64 * h + lane : zeros + lane;
                    v[i * NH + h] = *ptr;
                }
            }
        }
        const double* wsrc = a.weight ? a.weight : reinterpret_cast<const double*>(a.indptr);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const long long col = FACT ? (long long)C + 2 * lane + h : (long long)C + 64 * h + lane;      // (FACT: columns 2 l, 2 l + 1)
            wc[h] = wsrc[col < a.nbins ? col : a.nbins - 1];      // raw: reads the row offsets, a table of the same length — never looked at
        }
        {   // row weights: lane i < RPW holds its row's, broadcast at store time
            const long long row = (long long)R + my_rr;
            wrv = wsrc[row < a.nbins ? row : a.nbins - 1];
        }
    };
    auto band_store = [&](int ev, const int (&v)[NRH], const double (&wc)[NH], double wrv) __attribute__((always_inline)) {
        const int R = fld(ev, 0), ch_end = fld(ev, 6), row_lo = fld(ev, 21), row_hi = fld(ev, 22);
        double wcs[NH];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) wcs[hh] = weight_of(wc[hh]);
        const double wrs = weight_of(wrv);
        unsigned long long okn[NH];                      // lane i < RPW: validity bits of its row (!FACT)
        if constexpr (!FACT) {
            const int row = R + my_rr;
            const bool live = lane < RPW && row < ch_end && my_rr >= row_lo && my_rr < row_hi;
            const bool row_bad = (a.badbits[(row < a.nbins ? row : 0) >> 6] >> (row & 63)) & 1ull;
#pragma unroll
            for (int h = 0; h < NH; ++h) okn[h] = live ? ok_mask(ev, my_rr, h, row_bad) : 0ull;
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int rr = wave * RPW + i;
            if (rr < row_lo || rr >= row_hi) continue;   // (uniform) no window of the block reads this row
            const double wr = __longlong_as_double((long long)bcast64((unsigned long long)__double_as_longlong(wrs), i));
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                const int colx = FACT ? 2 * lane + hh : 64 * hh + lane;      // (band_issue: FACT lanes hold columns 2 l and 2 l + 1)
                double val = (double)v[i * NH + hh] * wr * wcs[hh];      // (count_to_f64, K1q's way round v_cvt_f64_i32: 1 % slower here, A/B)
                if (nf) val = (val == val) ? val : 0.0;
                if (OOE) {
                    const double e = exp_lds[colx - rr + (RSR - 1)];      // expected of |col - row|
                    val = FACT ? val * e : val / e;         // (FACT: the entry is the reciprocal — exp_entry)
                    val = (val == val) ? val : 0.0;         // NaN quotients are skipped, inf is kept
                    if constexpr (!FACT) {
                        int e_ok = (e == e && e != 0.0) ? 1 : 0;     // (through a register the compiler cannot fold: see K1q's store_region)
                        asm volatile("" : "+v"(e_ok));
                        const unsigned long long eok = __ballot(e_ok);
                        if (lane == i) okn[hh] &= eok;
                    }
                }
                tile[rr * LS + colx] = val;
            }
        }
        if constexpr (!FACT) {
            if (lane < RPW && my_rr >= row_lo && my_rr < row_hi) {
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) vbits[my_rr * VBW + hh] = okn[hh];
            }
        }
    };

    // ---- the windows of the staged block ---------------------------------------------------------------------------
    struct Cur { int start, n; unsigned long long rowbad[2], colbad[2]; };
    const unsigned vb_base = (unsigned)(uintptr_t)vbits;
    auto load_batch = [&](int start, int at, int end) __attribute__((always_inline)) -> int {
        return at + lane < end ? (int)wa.win[start + at + lane] : 0;
    };
    // windows jb <= j < je of the wave's current batch (window j sits in lane j).  ROLLING pipeline over ONE set of value
    // registers: the CH (+1) LDS reads of a window are in flight; cell i of window j is waited for (`s_waitcnt lgkmcnt(NB-1)`:
    // LDS operations return in order, so all but the NB-1 youngest have landed = the oldest one has), added, and its register
    // is at once the destination of the same cell of window j + 1.  (K1q double-buffers two windows: at CH = 13 that is 52
    // value registers and the kernel spilled.)  Past the run's last window the pipeline reads that window once more, never added.
    auto run = [&](int offv, int drv, int dcv, int jb, int je) __attribute__((always_inline)) {
        if (jb >= je) return;
        constexpr int NB = CH + (FACT ? 0 : 2);            // LDS operations of one window (<= 15: what the counter holds)
        double v[CH];
        unsigned long long vraw = 0ull, vraw2 = 0ull;
        auto addr_of = [&](int jj) __attribute__((always_inline)) -> unsigned { return lane_off8 + (unsigned)__builtin_amdgcn_readlane(offv, jj); };
        // validity bits of the sub-window's row p from column dc + kk on: the FOUR dwords from the one holding bit dc + kk (the lane's
        // cells sit NCH columns apart: up to NCH * (CH - 1) + 31 < 96 bits behind the first dword's bit 0)
        auto vaddr_of = [&](int jj) __attribute__((always_inline)) -> unsigned {
            const int dr = __builtin_amdgcn_readlane(drv, jj), dc = __builtin_amdgcn_readlane(dcv, jj);
            return vb_base + (unsigned)((dr + p) * (VBW * 8)) + 4u * ((unsigned)(dc + kk) >> 5);
        };
        auto bits_of = [&](int jj) __attribute__((always_inline)) -> unsigned long long {
            const int sft = (__builtin_amdgcn_readlane(dcv, jj) + kk) & 31;
            return (vraw >> sft) | (sft ? vraw2 << (64 - sft) : 0ull);
        };
        unsigned ad = addr_of(jb), av = 0u;
        if constexpr (!FACT) { av = vaddr_of(jb); lds_read2_b32(vraw, av); lds_read2_b32_23(vraw2, av); }
        LdsReadRow<0, CH, 8 * NCH>::go(v, ad);
        for (int jj = jb; jj < je; ++jj) {
            const int nxt = jj + 1 < je ? jj + 1 : jj;
            const unsigned adn = addr_of(nxt);
            unsigned long long vw = 0ull;
            unsigned avn = av;
            if constexpr (!FACT) {
                avn = vaddr_of(nxt);
                lds_wait_but<NB - 2>(ad, av); lds_pin_u64(vraw); lds_pin_u64(vraw2);
                vw = bits_of(jj);
                lds_read2_b32(vraw, avn); lds_read2_b32_23(vraw2, avn);
            }
            RollRow<0, CH, NB, FACT, NCH>::go(v, sum, num, vw, ad, adn);
            ad = adn; av = avn;
        }
        lds_wait_all(ad, av, ad, av); lds_pin(v);
        if constexpr (!FACT) { lds_pin_u64(vraw); lds_pin_u64(vraw2); }
    };
    // bits [s, s + 64) of the 128-bit mask hi:lo, s in [0, 127]
    auto mask_at = [&](const unsigned long long (&m)[2], int s) __attribute__((always_inline)) -> unsigned long long {
        const int t = s & 63;
        unsigned long long v;
        if (s < 64) { v = m[0] >> t; if (t) v |= m[1] << (64 - t); } else v = m[1] >> t;
        return v;
    };
    // FACT bookkeeping of a batch, a lane per window (see K1q's fact_batch): fact_tot = {R[64], C[64], N}, rc_lds = RC
    auto fact_batch = [&](const Cur& g, int drv, int dcv, int nb) __attribute__((always_inline)) {
      if constexpr (FACT) {
        if (lane == 0) atomicAdd(&fact_tot[128], (unsigned)nb);
        if ((g.rowbad[0] | g.rowbad[1] | g.colbad[0] | g.colbad[1]) == 0ull) return;     // (uniform) no masked bin in the region
        const bool live = lane < nb;
        const unsigned long long hmask = sh >= 64 ? ~0ull : ((1ull << sh) - 1ull), wmask = sw >= 64 ? ~0ull : ((1ull << sw) - 1ull);
        unsigned long long rb = live ? mask_at(g.rowbad, drv) & hmask : 0ull;
        const unsigned long long cbm = live ? mask_at(g.colbad, dcv) & wmask : 0ull;
        unsigned long long cc = cbm;
        while (cc) { const int q = __ffsll((long long)cc) - 1; cc &= cc - 1ull; atomicAdd(&fact_tot[64 + q], 1u); }
        while (rb) {
            const int pp = __ffsll((long long)rb) - 1; rb &= rb - 1ull;
            atomicAdd(&fact_tot[pp], 1u);
            unsigned long long c2 = cbm;
            while (c2) { const int q = __ffsll((long long)c2) - 1; c2 &= c2 - 1ull; atomicAdd(&rc_lds[pp * 64 + q], 1u); }
        }
      }
    };
    // the NW / NPC waves of a panel share the block's windows in equal contiguous slices; the NPC waves that own slice `sub`
    // walk the same batches, and batch t's bookkeeping falls to panel t % NPC
    // Not equal shares: the SIMD favours its oldest wave, and the four waves of a panel sit on one SIMD (wave w: SIMD w % 4 =
    // panel) — with equal slices the youngest finished 35 % after the oldest and everybody waited at the barrier (phase
    // clocks, pad 25).  Shares in 1/256 of the block's windows, cumulative, for nsub = 4
    auto slice_of = [&](int n, int& lo, int& hi) __attribute__((always_inline)) {
        if (nsub == 4) {
            const int cum_lo = sub == 0 ? 0 : (sub == 1 ? wa.share[0] : (sub == 2 ? wa.share[1] : wa.share[2]));
            const int cum_hi = sub == 0 ? wa.share[0] : (sub == 1 ? wa.share[1] : (sub == 2 ? wa.share[2] : 256));
            lo = (int)(((long long)n * cum_lo + 128) >> 8); hi = sub == 3 ? n : (int)(((long long)n * cum_hi + 128) >> 8);
        } else { lo = (int)(((long long)n * sub) / nsub); hi = (int)(((long long)n * (sub + 1)) / nsub); }
    };
    auto windows = [&](const Cur& g, int wf, int rot, auto&& mid) __attribute__((always_inline)) {
        int lo, hi;
        slice_of(g.n, lo, hi);
        const bool cols_live = (panel * kWave) / NCH < sh;     // (uniform) a panel whose first slot lies past the group's last row has nothing to pile up
        // batch t's bookkeeping falls to panel (t + rot) % NPC, rot = the block's number: many-group calls (201-bin windows: ~80 items
        // per block, ~20 per slice) have ONE batch per slice and block — without the rotation panel 0 did all of it and its four
        // waves were the workgroup's slowest by 8-10 % (phase clocks, profiles/r05_k1w_phases.txt)
        int t = rot;
        {
            const int drv = wf & ((1 << kWinShift) - 1), dcv = (wf >> kWinShift) & ((1 << kWinShift) - 1);
            const int offv = 8 * (drv * LS + dcv);
            if (lo + kWave < hi) wf = load_batch(g.start, lo + kWave, hi);
            const int nb = (hi - lo) < kWave ? (hi - lo) : kWave;
            if (t % NPC == panel && nb > 0) fact_batch(g, drv, dcv, nb);
            const int cut = cols_live ? (nb * sub) / (2 * nsub) : 0;      // the look-ahead work sits at a different place in every wave of a SIMD
            run(offv, drv, dcv, 0, cut);
            mid();
            if (cols_live) run(offv, drv, dcv, cut, nb);
            ++t;
        }
        for (int s0 = lo + kWave; s0 < hi; s0 += kWave, ++t) {
            const int drv = wf & ((1 << kWinShift) - 1), dcv = (wf >> kWinShift) & ((1 << kWinShift) - 1);
            const int offv = 8 * (drv * LS + dcv);
            if (s0 + kWave < hi) wf = load_batch(g.start, s0 + kWave, hi);
            const int nb = (hi - s0) < kWave ? (hi - s0) : kWave;
            if (t % NPC == panel) fact_batch(g, drv, dcv, nb);
            if (cols_live) run(offv, drv, dcv, 0, nb);
        }
    };
    auto first_coords = [&](int ev, int& wf) __attribute__((always_inline)) {
        int lo, hi;
        slice_of(fld(ev, 3), lo, hi);
        wf = load_batch(fld(ev, 2), lo, hi);
    };
    auto cur_of = [&](int ev) __attribute__((always_inline)) -> Cur {
        Cur c;
        c.start = fld(ev, 2); c.n = fld(ev, 3);
        c.rowbad[0] = fld64(ev, 16); c.rowbad[1] = fld64(ev, 18);
        c.colbad[0] = ~fld64(ev, 12); c.colbad[1] = ~fld64(ev, 14);
        return c;
    };

    // ---- flush of a segment (tile, flip, group): merge the register tiles of the waves that share a panel (fixed binary
    // tree), write the partial record, clear the accumulators.  The region buffer is scratch here: every wave is done
    // reading the staged region, and the next region is stored after it.
    auto flush = [&](int seg) __attribute__((always_inline)) {
        double*   mf = tile;                                              // [NW/2][CH][64] doubles, then the same in u32
        unsigned* mn = reinterpret_cast<unsigned*>(tile + (NW / 2) * CH * kWave);
        __syncthreads();
        for (int step = 1; step < nsub; step <<= 1) {
            const int slot_w = panel + NPC * (sub >> 1);                 // writers of one step differ in (panel, sub >> 1)
            if ((sub & (2 * step - 1)) == step) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    mf[(slot_w * CH + i) * kWave + lane] = sum[i];
                    if (!FACT) mn[(slot_w * CH + i) * kWave + lane] = num[i];
                }
            }
            __syncthreads();
            if ((sub & (2 * step - 1)) == 0 && sub + step < nsub) {
                const int from = panel + NPC * ((sub + step) >> 1);
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    sum[i] += mf[(from * CH + i) * kWave + lane];
                    if (!FACT) num[i] += mn[(from * CH + i) * kWave + lane];
                }
            }
            __syncthreads();
        }
        const size_t rec = (size_t)seg * (size_t)wa.NG + (size_t)grp + (size_t)g_id;
        double*   of = wa.rec_f64 + rec * kWideRec;
        unsigned* on = wa.rec_num + rec * kWideRec;
        if (sub == 0) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if ((chmask >> i) & 1u) {
                    of[p * 64 + kk + NCH * i] = sum[i];
                    if (!FACT) on[p * 64 + kk + NCH * i] = num[i];
                }
        }
        if constexpr (FACT) {
            for (int t = tid; t < sh * 64; t += NTHR) {
                const int pp = t >> 6, qq = t & 63;
                if (qq < sw) on[t] = fact_tot[128] - fact_tot[pp] - fact_tot[64 + qq] + rc_lds[t];
            }
        }
        if (tid == 0) wa.rec_seg[rec] = (unsigned)(seg * wa.NG + grp) + 1u;
        __syncthreads();                                 // fact_tot / rc_lds have been read
        zero_acc();
        __syncthreads();
    };

    // ---- the block loop: region b is piled up while b+1's counts and b+2's table entry are on their way ----------------
    int ev0 = entry_load(bb), ev1 = ev0, evn = ev0;
    int v[NRH];
    double wc[NH], wrv = 1.0;
    int w0f, w1f = 0;
    {   // prologue: stage block bb without overlap
        set_group(fld(ev0, 30));
        band_issue(ev0, v, wc, wrv);
        first_coords(ev0, w0f);
        if (bb + 1 < be) ev1 = entry_load(bb + 1);
        if constexpr (OOE) { if (tid < 256) exp_lds[tid] = exp_entry(exp_fetch(ev0)); }
        __syncthreads();
        band_store(ev0, v, wc, wrv);
        __syncthreads();
    }
    long long tk[6] = {0, 0, 0, 0, 0, 0}, tmid = 0;
    double e_next = 0.0;
    const bool timed = wa.timing != nullptr;
    auto tick = [&]() __attribute__((always_inline)) -> long long { return timed ? (long long)__builtin_readcyclecounter() : 0; };
    for (int b = bb; b < be; ++b) {
        const bool has1 = b + 1 < be, has2 = b + 2 < be;
        const long long t0 = tick();
        auto lookahead = [&]() __attribute__((always_inline)) {
            const long long m0 = tick();
            if constexpr (OOE) { if (has1) e_next = exp_fetch(ev1); }
            if (has1) { band_issue(ev1, v, wc, wrv); first_coords(ev1, w1f); }
            if (has2) evn = entry_load(b + 2);
            tmid += tick() - m0;
        };
        const Cur c0 = cur_of(ev0);
        const long long t1 = tick();
        windows(c0, w0f, b & 3, lookahead);
        if constexpr (OOE) { if (has1 && tid < 256) exp_lds[tid] = exp_entry(e_next); }     // (last read when region b was stored; visible after the barrier below)
        const long long t2 = tick();
        const int seg0 = fld(ev0, 20), grp0 = fld(ev0, 30);
        if (!has1) { flush(seg0); break; }
        const int seg1 = fld(ev1, 20), grp1 = fld(ev1, 30);
        if (seg1 != seg0 || grp1 != grp0) {              // (uniform) the next block belongs to another (tile, flip, group)
            flush(seg0);
            if (grp1 != grp0) set_group(grp1);
        } else __syncthreads();                          // every wave is done reading region b
        const long long t3 = tick();
        band_store(ev1, v, wc, wrv);
        const long long t4 = tick();
        __syncthreads();
        const long long t5 = tick();
        ev0 = ev1; w0f = w1f; ev1 = evn;
        if (timed) { tk[0] += t1 - t0; tk[1] += t2 - t1; tk[2] += t3 - t2; tk[3] += t4 - t3; tk[4] += t5 - t4; tk[5] += tick() - t5; }
    }
    if (timed && lane == 0) {
        long long* o = wa.timing + ((size_t)g_id * NW + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = tk[i];
        o[6] = be - bb; o[7] = tmid;
    }
    (void)G;
}

// ---- block-order prepass of K1w: keys of the (window, group) items ---------------------------------------------------
// item ii = group * n + window (group-major: consecutive threads read consecutive windows).  key = ((tile, flip) run, group)
// | block row | block column of the SUB-window's corner; value = that corner inside its block.  The verdict counters are
// per WINDOW (group 0 counts): [0] windows the band cannot serve (not inside one chromosome), [1] windows a masked
// diagonal reaches (or that reach past the usable expected), [2] windows leaving the dense band.
template <typename KeyT>
__global__ __launch_bounds__(256) void wide_key_kernel(const int* __restrict__ r0, const int* __restrict__ c0, long long n, long long n_items,
                                                       const long long* __restrict__ seg_end, int nseg2t,
                                                       const IdxChrom* __restrict__ chroms, int n_chrom,
                                                       const unsigned short* __restrict__ bin_chrom, long long nbins,
                                                       const int* __restrict__ brow_base, int W, int NG, int NGc, int SH, int SW,
                                                       int BR, int BC, int sh_br, int sh_seg, int seg_shift,
                                                       int clear_gap, int far_gap, int band_w,
                                                       KeyT* __restrict__ keys, unsigned short* __restrict__ vals,
                                                       unsigned* __restrict__ counters,
                                                       unsigned* __restrict__ hi_hist /* nullable: tilehist[workgroup][hi_bins] (pup_bin.hpp) */,
                                                       int hi_shift, int hi_bins, int per_thread, int rel_bc /* see staged_key_kernel */) {
    // (32-bit index arithmetic — n_items < 2^31: wide_run —, tables in dynamic LDS, loads of a batch first: see staged_key_kernel)
    const int kPer = per_thread;
    extern __shared__ long long s_dyn[];
    int* const s_seg = reinterpret_cast<int*>(s_dyn);                 // [nseg2t] (tile, flip) run ends
    unsigned* const s_hh = reinterpret_cast<unsigned*>(s_seg + nseg2t);           // [hi_bins] high-digit counts of this workgroup's items
    int* const s_cs = reinterpret_cast<int*>(s_hh + hi_bins);         // [n_chrom] first bin | end | block rows before
    int* const s_ce = s_cs + n_chrom;
    int* const s_bb = s_ce + n_chrom;
    for (int k = threadIdx.x; k < n_chrom; k += blockDim.x) { s_cs[k] = chroms[k].start; s_ce[k] = chroms[k].end; s_bb[k] = brow_base[k]; }
    for (int k = threadIdx.x; k < nseg2t; k += blockDim.x) s_seg[k] = (int)seg_end[k];
    if (hi_hist) for (int k = threadIdx.x; k < hi_bins; k += blockDim.x) s_hh[k] = 0u;
    __syncthreads();
    unsigned bad = 0u, near_c = 0u, far_c = 0u;
    const int lane = threadIdx.x & 63;
    const int n32 = (int)n, ni32 = (int)n_items, ni_last = ni32 - 1, nb_last = (int)nbins - 1;
    // group and (tile, flip) run of an item depend on its NUMBER only; a workgroup's 8192 consecutive items nearly always share
    // them: worked out once per workgroup (per item: a division by n and a bisection)
    auto run_of = [&](int i) -> int {
        int lo = 0, hi = nseg2t;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (s_seg[m] <= i) lo = m + 1; else hi = m; }
        return lo;
    };
    const int tile_first = blockIdx.x * kPer * (int)blockDim.x;
    int tile_last = tile_first + kPer * (int)blockDim.x - 1;
    tile_last = tile_last < ni_last ? tile_last : ni_last;
    const int grp_a = tile_first / n32, grp_b = tile_last / n32;
    const int lo_a = __builtin_amdgcn_readfirstlane(run_of(tile_first - grp_a * n32)), lo_b = __builtin_amdgcn_readfirstlane(run_of(tile_last - grp_b * n32));
    const bool one_run = grp_a == grp_b && lo_a == lo_b;     // (uniform)
    constexpr int kKeyBatch = 8;
    for (int u0 = 0; u0 < kPer; u0 += kKeyBatch) {
    int rb[kKeyBatch], cb[kKeyBatch], cab[kKeyBatch], gb[kKeyBatch], ib[kKeyBatch];
    const int ii_base = (blockIdx.x * kPer + u0) * (int)blockDim.x + (int)threadIdx.x;
#pragma unroll
    for (int k = 0; k < kKeyBatch; ++k) {
        int ii = ii_base + k * (int)blockDim.x;
        ii = ii < ni_last ? ii : ni_last;
        gb[k] = one_run ? grp_a : ii / n32;
        ib[k] = ii - gb[k] * n32;
        rb[k] = r0[ib[k]]; cb[k] = c0[ib[k]];
    }
#pragma unroll
    for (int k = 0; k < kKeyBatch; ++k) {
        int rc = rb[k] > 0 ? rb[k] : 0;
        rc = rc < nb_last ? rc : nb_last;
        cab[k] = (int)bin_chrom[rc];
    }
#pragma unroll
    for (int k = 0; k < kKeyBatch; ++k) {
        if (u0 + k >= kPer) break;                           // (uniform)
        const int ii = ii_base + k * (int)blockDim.x;
        const bool live = ii < ni32;
        const int grp = live ? gb[k] : 0;
        const int i = live ? ib[k] : 0;
        const int r = live ? rb[k] : 0, c = live ? cb[k] : 0;
        const int lo = one_run ? lo_a : run_of(i);
        const unsigned seg = (unsigned)(lo >> seg_shift) * (unsigned)NG + (unsigned)grp;
        bool ok = r >= 0 && c >= 0 && r <= nb_last;
        bool rel_over = false;
        unsigned br = 0, bc = 0;
        unsigned inside = 0u;
        if (ok) {
            const int ca = cab[k];
            const int cs = s_cs[ca], ce = s_ce[ca];
            ok = r >= cs && r + W <= ce && c >= cs && c + W <= ce;
            if (ok) {
                const int gi = grp / NGc, gj = grp - gi * NGc;
                const int rs = r + gi * SH - cs, cc = c + gj * SW - cs;       // the sub-window's corner, chromosome-relative
                const int qr = rs / BR, qc = cc / BC;
                br = (unsigned)(s_bb[ca] + qr);
                bc = (unsigned)qc;
                if (rel_bc) {                                // (a window whose number does not fit leaves the band: the whole call goes to K1b)
                    const int rel = qc - (qr * BR) / BC + (rel_bc - 1);
                    if ((unsigned)rel >= (1u << sh_br)) { rel_over = true; bc = 0u; } else bc = (unsigned)rel;
                }
                inside = (unsigned)(rs - qr * BR) | ((unsigned)(cc - qc * BC) << kWinShift);
            }
        }
        const bool first = live && grp == 0;
        if (first && !ok) ++bad;
        near_c += (first && (c - r < clear_gap || (c + W - 1) - r >= far_gap)) ? 1u : 0u;
        far_c += ((first && (c + W - 1) - r >= band_w) || (live && rel_over)) ? 1u : 0u;
        unsigned key_hi = 0u; bool counted = false;
        if (live) {
            if constexpr (sizeof(KeyT) == 4) {
                const unsigned key = (seg << sh_seg) | (br << sh_br) | bc;
                keys[ii] = (KeyT)key;
                key_hi = key >> hi_shift;
            } else {
                const unsigned long long key = ((unsigned long long)seg << sh_seg) | ((unsigned long long)br << sh_br) | bc;
                keys[ii] = (KeyT)key;
                key_hi = (unsigned)(key >> hi_shift);
            }
            counted = hi_hist != nullptr;
            vals[ii] = (unsigned short)inside;
        }
        {   // the workgroup's high-digit counts: ONE LDS atomic per distinct digit of the wave (the stream is nearly sorted by block row:
            // 64 lanes adding to one counter cost more than the whole key computation)
            unsigned long long todo = __ballot(counted);
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)key_hi, l);
                const unsigned long long m = __ballot(counted && key_hi == d0);
                if (lane == l) atomicAdd(&s_hh[d0], (unsigned)__popcll(m));
                todo &= ~m;
            }
        }
    }
    }   // batches
    for (int off = 32; off > 0; off >>= 1) { near_c += __shfl_down(near_c, off); far_c += __shfl_down(far_c, off); }
    if (lane == 0 && near_c) atomicAdd(&counters[1], near_c);
    if (lane == 0 && far_c) atomicAdd(&counters[2], far_c);
    if (bad) atomicAdd(&counters[0], bad);
    if (hi_hist) {
        __syncthreads();
        for (int k = threadIdx.x; k < hi_bins; k += blockDim.x) hi_hist[(size_t)blockIdx.x * hi_bins + k] = s_hh[k];
    }
}

// fixed-order reduction of K1w's partial records into the running accumulators.  Workgroup = (64 accumulator cells, tile),
// 64 cells x kRedParts interleaved partial sums over the workgroups' records (as reduce_staged_kernel).  An accumulator cell
// (P, Q) receives sub-window cell (P, Q) of the unflipped records and, anti-transposed (flip_snip_func,
// coolpuppy/coolpup.py:128-131), cell (W-1-Q, W-1-P) of the flipped ones: flip 0 first, then flip 1, records in workgroup order.
PUP_KERNEL __launch_bounds__(64 * kRedParts) void reduce_wide_kernel(
        const double* __restrict__ rec_f64, const unsigned* __restrict__ rec_num, const unsigned* __restrict__ rec_seg,
        int G, int W, int NG, int NGc, int SH, int SW, int n_flip /* 1: no flipped windows in the call */,
        int Lf, double* out_f64, long long* out_num, long long* out_n /* nullable: windows per tile, from the run ends */, const long long* __restrict__ seg_end) {
    __shared__ double    sf[kRedParts][64];
    __shared__ long long si[kRedParts][64];
    const int t = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y;
    if (out_n && blockIdx.x == 0 && cx == 0 && py == 0) out_n[t] += seg_end[2 * t + 1] - (t ? seg_end[2 * t - 1] : 0);
    const int cell = blockIdx.x * 64 + cx;
    const int W2 = W * W;
    double accf = 0.0; long long acci = 0;
    if (cell < W2) {
        const int P = cell / W, Q = cell - P * W;
        for (int fl = 0; fl < n_flip; ++fl) {
            const int ps = fl ? W - 1 - Q : P, qs = fl ? W - 1 - P : Q;          // the window cell that lands here
            const int gi = ps / SH, gj = qs / SW;
            const int seg = (t * 2 + fl) * NG + gi * NGc + gj;
            const int at = (ps - gi * SH) * 64 + (qs - gj * SW);
            constexpr int kUn = 4;
            for (int g0 = py; g0 < G; g0 += kUn * kRedParts) {
                bool ok[kUn]; double vf[kUn]; unsigned vn[kUn];
#pragma unroll
                for (int u = 0; u < kUn; ++u) { const int g = g0 + u * kRedParts; ok[u] = g < G && rec_seg[seg + g] == (unsigned)seg + 1u; }
#pragma unroll
                for (int u = 0; u < kUn; ++u) {
                    const size_t rec = (size_t)(seg + g0 + u * kRedParts);
                    vf[u] = ok[u] ? rec_f64[rec * kWideRec + at] : 0.0;
                    vn[u] = ok[u] ? rec_num[rec * kWideRec + at] : 0u;
                }
#pragma unroll
                for (int u = 0; u < kUn; ++u) if (ok[u]) { accf += vf[u]; acci += (long long)vn[u]; }
            }
        }
    }
    sf[py][cx] = accf; si[py][cx] = acci;
    __syncthreads();
    if (py != 0 || cell >= W2) return;
    double tf = 0.0; long long ti = 0;
#pragma unroll
    for (int y = 0; y < kRedParts; ++y) { tf += sf[y][cx]; ti += si[y][cx]; }
    out_f64[(size_t)t * Lf + cell] += tf;
    out_num[(size_t)t * W2 + cell] += ti;
}


// ---- coverage vectors of a pile-up, on their own (PUP_MODE_COV beside the staged kernels) ---------------------------------------
// cov_start / cov_end of a tile are plain sums over its windows of cov[r0 .. r0 + W) and cov[c0 .. c0 + W) (NaN adds nothing;
// the flip leaves them alone; under TRANSPOSE the two swap: coolpuppy/coolpup.py:1151-1153, lib/puputils.py:30-38) — nothing in
// them depends on the pixels.  Riding inside the staged kernel they forced its fat 8-wave geometry (EXTRA); as a pass of their
// own — O(W) per window from a 2.4 MB vector that lives in L2 — the pile-up keeps the lean kernel.  Deterministic: fixed chunks
// of kCovChunk windows of the caller's order; a (chunk, tile) piece is summed by four waves (wave w: windows w, w + 4, ...; a
// lane per vector entry), the waves in order, into record chunk + tile (unique: both grow along the stream);
// cov_reduce_kernel adds a tile's records in chunk order.
constexpr int kCovChunk = 2048;
PUP_KERNEL __launch_bounds__(256) void cov_vectors_kernel(const int* __restrict__ r0, const int* __restrict__ c0, long long n,
                                                          const long long* __restrict__ seg_end /* [2T]: entry 2t + 1 = end of tile t */, int T,
                                                          const double* __restrict__ cov, long long nbins, int W, int transpose,
                                                          double* __restrict__ rec /* [nchunks + T][2W] */, unsigned* __restrict__ rec_owner) {
    extern __shared__ double sacc[];                      // [4][2W]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = 2 * W;
    const long long cb = (long long)blockIdx.x * kCovChunk, ce = cb + kCovChunk < n ? cb + kCovChunk : n;
    // first tile that ends behind the chunk's first window
    int lo = 0, hi = T;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (seg_end[2 * m + 1] <= cb) lo = m + 1; else hi = m; }
    for (int t = lo; t < T; ++t) {
        const long long tb = t ? seg_end[2 * t - 1] : 0, te = seg_end[2 * t + 1];
        if (tb >= ce) break;
        const long long b = tb > cb ? tb : cb, e = te < ce ? te : ce;
        if (b >= e) continue;
        for (int k = threadIdx.x; k < 4 * L; k += 256) sacc[k] = 0.0;
        __syncthreads();
        for (long long j = b + wave; j < e; j += 4) {
            const int rs = __builtin_amdgcn_readfirstlane(transpose ? c0[j] : r0[j]);
            const int cs = __builtin_amdgcn_readfirstlane(transpose ? r0[j] : c0[j]);
            if (rs < 0 || cs < 0 || (long long)rs + W > nbins || (long long)cs + W > nbins) continue;      // (reported by the pile-up kernel)
            for (int k = lane; k < L; k += kWave) {
                const double v = cov[k < W ? rs + k : cs + (k - W)];
                if (v == v) sacc[wave * L + k] += v;     // one lane per entry: no race
            }
        }
        __syncthreads();
        const size_t id = (size_t)blockIdx.x + (size_t)t;
        for (int k = threadIdx.x; k < L; k += 256) rec[id * L + k] = ((sacc[k] + sacc[L + k]) + sacc[2 * L + k]) + sacc[3 * L + k];
        if (threadIdx.x == 0) rec_owner[id] = (unsigned)t + 1u;
        __syncthreads();
    }
}

PUP_KERNEL __launch_bounds__(64) void cov_reduce_kernel(const double* __restrict__ rec, const unsigned* __restrict__ rec_owner,
                                                        const long long* __restrict__ seg_end, int W, int Lf, double* out_f64) {
    const int t = blockIdx.y, L = 2 * W;
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= L) return;
    const long long tb = t ? seg_end[2 * t - 1] : 0, te = seg_end[2 * t + 1];
    if (te <= tb) return;
    double acc = 0.0;
    for (long long ch = tb / kCovChunk; ch <= (te - 1) / kCovChunk; ++ch) {
        const size_t id = (size_t)ch + (size_t)t;
        if (rec_owner[id] == (unsigned)t + 1u) acc += rec[id * L + k];
    }
    out_f64[(size_t)t * Lf + (size_t)W * W + k] += acc;
}

// ---- the expected-as-control pass for windows of any width (PUP_MODE_EXPECTED) ------------------------------------------
// expected & !ooe: the "control" of a snippet is the unmasked window of its expected matrix (coolpuppy/coolpup.py:1135-1139),
// a Toeplitz slice: cell (p, q) = E[|c0 + q - r0 - p|] of the region holding the snippet's first row (or the trans scalar of
// the region pair).  It only depends on q - p: per snippet the 2W - 1 values S[d] = E[|c0 - r0 + d|], d = q - p in (-W, W),
// are ALL the arithmetic there is — a lane per diagonal instead of a lane per cell, O(W) per snippet, no tile in LDS, no
// limit on W (the LDS-tile kernel this replaces for the mode stopped at W = 115).  A chunk (one wave, one (tile, flip) run
// of snippets) keeps sum / count per diagonal in registers and writes one record of 2W - 1 diagonals; expand_diag_kernel
// sums a tile's records in fixed order and spreads them over the W x W accumulators (transposed / flipped cells keep their
// diagonal up to sign: map_cell).  nansum semantics as in the LDS-tile kernel: NaN adds nothing, +-inf is summed but not
// counted in num (coolpuppy/lib/puputils.py:18-29).
constexpr int kDiagPer = 8;                           // diagonals per lane: windows up to (64 * 8 + 1) / 2 = 256 bins per pass of a chunk
PUP_KERNEL __launch_bounds__(kWave) void expected_diag_kernel(K1Args a, int d_first /* first diagonal slot of this pass */,
                                                              double* __restrict__ part_f64, unsigned* __restrict__ part_num, int ND) {
    const int lane = threadIdx.x;
    const int ck = a.block_chunk[blockIdx.x];
    if (ck < 0) return;
    const int W = a.W;
    const long long cb = a.chunk_begin[ck], ce = a.chunk_end[ck], cstep = a.chunk_stride[ck];
    const bool use_exp = (a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0;
    ExpCache ecache;
    double s[kDiagPer]; unsigned m[kDiagPer];
#pragma unroll
    for (int i = 0; i < kDiagPer; ++i) { s[i] = 0.0; m[i] = 0u; }
    for (long long sn = cb; sn < ce; sn += cstep) {
        const int r0s = __builtin_amdgcn_readfirstlane(a.r0[sn]);
        const int c0s = __builtin_amdgcn_readfirstlane(a.c0[sn]);
        if (r0s < 0 || c0s < 0 || (long long)r0s + W > a.nbins || (long long)c0s + W > a.nbins) {
            if (lane == 0 && d_first == 0) atomicExch(a.err, 1);
            continue;
        }
        if (!use_exp) continue;
        const ExpSel es = select_expected(a, ecache, r0s, c0s);
#pragma unroll
        for (int i = 0; i < kDiagPer; ++i) {
            const int slot = d_first + i * kWave + lane;                 // diagonal d = slot - (W - 1)
            if (slot < ND) {
                long long ad = (long long)(c0s - r0s) + slot - (W - 1); if (ad < 0) ad = -ad;
                const double e = es.at(ad);
                if (e == e) { s[i] += e; if (!__builtin_isinf(e)) m[i] += 1u; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kDiagPer; ++i) {
        const int slot = d_first + i * kWave + lane;
        if (slot < ND) { part_f64[(size_t)ck * ND + slot] = s[i]; part_num[(size_t)ck * ND + slot] = m[i]; }
    }
}

// accumulator cell <- diagonal slot of the (tile, flip) run records (the chunk records summed in fixed order by
// reduce_partials_kernel): run 2t holds tile t's unflipped snippets, run 2t + 1 its flipped ones.
PUP_KERNEL __launch_bounds__(256) void expand_diag_kernel(const double* __restrict__ run_f64, const long long* __restrict__ run_num,
                                                          int W, int ND, int transpose, int Lf, double* out_f64, long long* out_num) {
    const int t = blockIdx.y;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    const int W2 = W * W;
    if (cell >= W2) return;
    const int P = cell / W, Q = cell - P * W;
    double accf = 0.0; long long acci = 0;
    for (int fl = 0; fl < 2; ++fl) {
        // invert map_cell: which window cell (p, q) lands in accumulator cell (P, Q)
        int pp = P, qq = Q;
        if (fl) { const int tp = W - 1 - qq; qq = W - 1 - pp; pp = tp; }
        const int p = transpose ? qq : pp, q = transpose ? pp : qq;
        const int slot = (q - p) + (W - 1);
        accf += run_f64[(size_t)(2 * t + fl) * ND + slot];
        acci += run_num[(size_t)(2 * t + fl) * ND + slot];
    }
    out_f64[(size_t)t * Lf + cell] += accf;
    out_num[(size_t)t * W2 + cell] += acci;
}

}  // namespace pup
