// pup_staged.hpp — K1q, the workgroup-staged pile-up kernel for many OVERLAPPING cis windows, and its device-side
// block-order prepass.  gfx950 (CDNA4) only.
//
// When a pile-up is large its windows overlap: 1.1e7 windows (1e6 pairs + 10 shifted controls each) on a human 10 kb map put
// ~750 top-left corners into every 108 x 108 block of the matrix near the diagonal.  The per-window kernel (K1r) fetches every
// window on its own (~75 L2 lines).  K1q has the engine sort the windows by BLOCK of corners on the device; a PERSISTENT
// workgroup then walks a contiguous range of blocks and for each one STAGES the RSR x RSC region of bins the block's windows
// live in ONCE into LDS, as final cell values — everything the reference does to a cell depends on its absolute (row, col)
// only: balanced value, masked bins, ignored diagonals, expected of |col-row| (coolpuppy/coolpup.py:1104-1157) — so the
// staged cell already is what gets summed (0 where nothing is to be added) and one validity bit per cell says whether it
// counts in num (coolpuppy/lib/puputils.py:18-29).  Per window a wave then does CH LDS reads + CH f64 adds per lane.
//
// Round 3 geometry (round 2: 64 x 64 regions, 4 waves, 4 workgroups per CU, one launch-time chunk table):
//   * region = RSR x RSC bins, up to 128 x 128 f64 = 128 KiB of the CU's 160 KiB LDS, block = (RSR-W+1) x (RSC-W+1)
//     corners (W = 21: 108 x 108).  Every matrix cell is staged (128/108)^2 = 1.4 times instead of (64/44)^2 = 2.1: the halo
//     re-reads that made the round-2 kernel fetch 2.3 x its compulsory HBM bytes shrink accordingly, and one staging (two
//     barriers, one table entry, one pipeline step) now serves ~750 windows instead of ~125;
//   * ONE workgroup of NW = 16 waves per CU (the big region leaves room for nothing else), persistent: workgroup g walks the
//     blocks [wg_first[g], wg_first[g+1]) — a contiguous range of the block table holding 1/G of the call's windows, worked
//     out on the device by the prepass — so there is no launch-time chunk table, no host round trip between sort and pile-up,
//     and one prologue / epilogue per workgroup instead of one per 4 regions;
//   * a range may cross SEGMENTS (tile pair x flip state): the accumulators are flushed into the partial record
//     (segment, slot, workgroup) at every change of segment; records carry a valid flag, the reduction skips the others;
//   * staging is wave64-shaped as before: a WAVE per region row, a LANE per column of a 64-column half; the row's presence
//     bits come from its rank-bitmap index line, the pixels under them are contiguous in `bal`, lane l's being the
//     mbcnt(bits, l)-th of the run: one coalesced value load per row half, masks as 64-bit lane predicates, conflict-free LDS
//     store; software-pipelined over the block table (values of block b+1, index lines of b+2, table entry of b+3 in flight
//     while block b is piled up);
//   * blocks with few windows stage only the rows their windows touch (row hull in the table entry);
//   * cells of a lane are INTERLEAVED (lane (p, k) owns columns k, k + NCH, ... of window row p) and the row stride is
//     LS = RSC + NCH doubles: ds_read_b64 at its conflict-free rate of 256 B/clk for every window offset and width.
// Same integers as K1r; sums differ by the order of the f64 additions only (fixed: bit-reproducible run to run).
#pragma once
#include "pup_kernels.hpp"

namespace pup {

// block table entry: 128 bytes, fetched with ONE vector load (dword lane & 31 per lane) that can stay in flight across the
// window loop; fields are pulled out with readlane when a pipeline stage needs them
struct __attribute__((aligned(128))) StagedBlock {
    int R, C;                    //  0  1  region origin (global bins)
    int start, count;            //  2  3  its windows [start, start + count) in the sorted copy
    int count0;                  //  4     (= bnd[0]) the first count0 of them go to accumulator slot 0
    int ereg;                    //  5     expected region of the block's windows (-1: none)
    int ch_end;                  //  6     end of the chromosome (global bin)
    int nblk;                    //  7     index lines per matrix row
    unsigned line0[2];           //  8  9  index line of (region row 0, region column 0 / 64)
    unsigned ws_sh[2];           // 10 11  word | bit << 8 of that column inside its line
    unsigned colok[4];           // 12-15  unmasked-column bits of columns 0..63 (lo, hi), 64..127 (lo, hi)
    unsigned rowbad[4];          // 16-19  masked-row bits of rows 0..63, 64..127
    int seg;                     // 20     segment = unit * 2 + flip (unit: tile pair or tile)
    int row_lo, row_hi;          // 21 22  region rows any window of the block touches: only these are staged
    int bnd[7];                  // 23-29  bnd[s] = end of the slot-s windows, s < 7 (slot 7 ends at count): inside a block the windows
                                 //        are in slot order (stable sort of a tile-major stream; with > 2 slots the slot is the key's
                                 //        lowest digit as well, so that the table kernel can find the boundaries)
    int pad[2];
};
static_assert(sizeof(StagedBlock) == 128, "block table entry must be two 64-byte lines");

struct StagedArgs {
    const StagedBlock*    blocks;
    const unsigned short* win;        // windows in block order, each as its corner inside its region: dr | dc << 7 | slot << 14
    const int*            wg_first;   // [G + 1] first block of every workgroup's range
    int                   U;          // pass units: tiles (ACC = 1), tile pairs (2), or sets of 4 pairs (8) — see staged_tile
    int                   PH;         // paired tiles: tile t < PH shares its pass with tile t + PH (its control); 0 when unpaired
    unsigned short*       rec_owner;  // [ACC * (2 T + G)] 1 + tile * 2 + flip of the record written there, 0: none.  Record of (tile, flip, slot,
                                      // workgroup g) = slot * (2 T + G) + tile * 2 + flip + g: a workgroup's block range is contiguous and the
                                      // blocks are sorted by (unit, flip), so within one slot that sum is unique — (2 T + G) records per slot
                                      // instead of 2 T G (round 3's table capped the staged kernel at 64 tiles)
    int                   T;          // tiles of the call
    const unsigned char*  teams;      // [U][16]: waves [teams[u][s], teams[u][s+1]) pile up the slot-s windows of unit u's blocks —
                                      // teams sized by the host in proportion to the tiles' window counts (none for an empty tile)
    int                   debug;      // timing experiments only (results are wrong): 1 = skip the window loop, 2 = skip the staging, 8 = skip the factorised-count bookkeeping
    unsigned short        wgt[16];    // sixteen-wave kernels: wave w's share of its team's windows, in 1/1024 of an average wave's (see wave_weight)
    long long*            timing;     // phase clocks per wave, [G][NW][8] (tools/k1_probe.py --phases), or nullptr
};

constexpr int kWinShift = 7;                          // bits of dr / dc in a window value
constexpr int kWinSlotBit = 14;                      // slot of a window of a tile PAIR (sets of pairs: lowest bits of the key)
constexpr int kSetPairs = 4;                          // tile pairs piled up together in one pass of an ACC = 8 kernel
constexpr int kSetSlotBits = 3;
constexpr int kBandFront = 256;                       // cells in front of the band table's row 0 (see band_issue)
constexpr int kMaxSegCount = 1024;                    // (tile, flip) runs one block-ordered call may have (key kernel LDS table)
constexpr int kProgSpinLimit = 1 << 18;                 // polls (s_sleep 1 each: ~10 ms in all) a wave of the progressive kernel waits for another before it gives up
// a waiting wave of the progressive kernel sleeps this long between polls (units of 64 clocks).  With s_sleep 1 a waiting wave — typically
// the OLDEST of its SIMD, which the arbiter favours — spent most of its SIMD's issue slots polling: the waves still piling up ran 2.5 x slower
#ifndef PUP_PROG_SLEEP
#define PUP_PROG_SLEEP 8
#endif
constexpr int kProgSleep = PUP_PROG_SLEEP;
constexpr int kBlockCost = 400;                       // staging one region, in windows' worth of time (workgroup ranges)
constexpr int kMaxStagedTiles = kMaxSegCount / 2;     // (tile, flip) runs fit the key kernel's LDS table
constexpr int kKeyMaxChrom = 3072;                    // chromosomes whose table (12 bytes each) the key kernels keep in LDS; assemblies of more
                                                      // scaffolds take the per-window kernels

__device__ __forceinline__ void lds_read2_b32(unsigned long long& dst, unsigned addr) {     // dwords at addr, addr + 4
    asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=&v"(dst) : "v"(addr));
}
__device__ __forceinline__ void lds_read2_b32_23(unsigned long long& dst, unsigned addr) {  // dwords at addr + 8, addr + 12
    asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=&v"(dst) : "v"(addr));
}
__device__ __forceinline__ void lds_pin_u64(unsigned long long& v) { asm volatile("" : "+v"(v)); }
// (double)v for a 32-bit count, exactly, without v_cvt_f64_i32: the dword v ^ 2^31 under the exponent of 2^52 IS the double
// 2^52 + 2^31 + v; one full-rate subtraction takes the bias off.  Same-box A/B against the conversion instruction (-DPUP_CVT_I32=1),
// three alternations: K1q 0.582 / 0.583 / 0.574 against 0.590 / 0.595 / 0.588 ms (the store phase's clocks do not move: it is bound
// by the LDS write path, the instruction mix of the look-ahead around it changes); K1w +1 %: kept here, not there.
#ifndef PUP_CVT_I32
#define PUP_CVT_I32 0
#endif
__device__ __forceinline__ double count_to_f64(int v) {
#if PUP_CVT_I32
    return (double)v;
#else
    return __hiloint2double(0x43300000, (int)((unsigned)v ^ 0x80000000u)) - 4503601774854144.0;      // 2^52 + 2^31
#endif
}
// wait until at most N LDS operations of this wave are outstanding (they return in order: everything older is complete);
// the address registers of the reads waited for stay untouched until here (see lds_wait_all in pup_kernels.hpp)
template <int N>
__device__ __forceinline__ void lds_wait_but(unsigned a0, unsigned a1) {
    asm volatile("s_waitcnt lgkmcnt(%2)" :: "v"(a0), "v"(a1), "n"(N) : "memory");
}

// accumulator slots of a pass unit -> tiles.  ACC = 1: the unit is the tile.  Paired (tile t with t + PH): ACC = 2 — unit u =
// pair u, slot = which of the two; ACC = 8 — unit u = pairs [4u, 4u + 4), slots 0..3 their first tiles, 4..7 their second.
template <int ACC>
__host__ __device__ inline int staged_tile(int unit, int slot, int PH) {
    if (ACC == 1) return unit;
    constexpr int SP = ACC / 2;
    const int g = unit * SP + slot % SP;
    return g < PH ? (slot / SP) * PH + g : -1;
}

// geometry of an instantiation (host and device agree through these)
template <int W> constexpr bool staged_big() { return W <= 21; }      // 128 x 128 regions, 16 waves: register budget of CH <= 7 cells
#ifndef PUP_SMALL_ROWS
#define PUP_SMALL_ROWS 64
#endif
constexpr int kSmallRows = PUP_SMALL_ROWS;            // rows of the DOUBLE-BUFFERED geometry (tuning bit 7): two regions of 64 x 131 doubles = 134 KB
template <int W, bool OOE, bool EXTRA, bool SMALL = false, bool FACT = true> struct StagedGeom {
    static constexpr bool big = staged_big<W>() && !EXTRA && !SMALL;
    static constexpr int RSR = big ? 128 : (SMALL ? kSmallRows : 64);
    static constexpr int RSC = 128;
    static constexpr int NW  = ((big || SMALL) && FACT) ? 16 : 8;   // 16 waves of 128 registers where the kernel fits them (factorised
                                                         // counts, one accumulator set per wave); the other instantiations
                                                         // need up to 256 registers: 8 waves on the same regions
};

// DB (round 5): TWO region buffers of RSR = 64 rows.  Block b is piled up from one buffer while every wave — at a staggered point
// inside its slice of windows — stores its rows of block b + 1 into the other (their counts were requested a block earlier) and
// requests those of block b + 2: the store burst is hidden behind the other waves' LDS reads and ONE barrier per block is left
// (with one buffer: barrier, store burst, barrier — 14 + 10 + 5 % of the kernel by the phase clocks of round 3).
#ifndef PUP_K1Q_DUAL
#define PUP_K1Q_DUAL 0
#endif
template <int W, bool OOE, int RSR, int RSC, int NW, int ACC, bool FACT, bool EXTRA, bool BAND = false, bool DB = false, bool PROG = false>
__global__ __launch_bounds__(kWave * NW, 1)
void pileup_staged_kernel(K1Args a, StagedArgs sa) {
    static_assert(!PROG || (BAND && FACT && !OOE && !EXTRA && !DB && NW == 16 && RSR == 128 && RSC == 128), "progressive staging: band, factorised counts, 128 x 128 regions on sixteen waves");
    static_assert(!(BAND && EXTRA), "pixel statistics need the presence bits of the index: sparse staging");
    static_assert(!DB || (BAND && FACT && !OOE && !EXTRA), "double-buffered regions: band staging, factorised counts, no expected");
    static_assert(W >= 3 && W <= 31, "workgroup-staged kernel serves windows of 3..31 bins");
    // (FACT && OOE: the engine has checked that every diagonal a window reaches has a usable expected — staged_run)
    static_assert(ACC == 1 || ACC == 2 || ACC == 8, "accumulator slots of a pass: a tile, a tile pair, four pairs");
    static_assert(RSC == 64 || RSC == 128, "a lane per column of a 64-column half");
    static_assert(RSR % NW == 0 && RSR <= 128 && RSC <= 128, "rows are dealt out evenly to the waves; corners need 7 bits");
    constexpr int NCH = kWave / W;
    constexpr int CH  = (W + NCH - 1) / NCH;
    constexpr int W2  = W * W;
    // DUAL (experiment, -DPUP_K1Q_DUAL=1): a tile PAIR with BOTH register tiles in every wave — no teams: every wave takes its share of
    // ALL the block's windows (the pair's first tile holds a tenth of them: two waves for it are 12.5 % of the workgroup for 9 % of
    // the windows, the fourteen others carry 6.5 % each instead of 6.25)
    constexpr bool DUAL = PUP_K1Q_DUAL && ACC == 2 && FACT && !EXTRA && !OOE && BAND && NW == 16 && !DB;
    constexpr int NH  = RSC / 64;                        // 64-column halves of a region row
    constexpr int LS  = RSC + ((NCH % 32) ? (NCH % 32) : 32);   // row stride in doubles, LS % 32 == NCH % 32 (see above)
    constexpr int RPW = RSR / NW;                        // region rows staged by each wave
    constexpr int NRH = RPW * NH;                        // row halves staged by each wave: one lane each in the lookup phase
    constexpr int NTHR = kWave * NW;
    constexpr int VBW = NH + 1;                          // validity words per row (+1: the dword-pair read may run one dword over)
    static_assert(NRH <= 32 && RPW <= 32, "row halves of a wave must fit the value registers");
    static_assert((size_t)NW / 2 * CH * kWave * 12 <= (size_t)RSR * LS * 8, "merge scratch must fit the region buffer");
    __shared__ double tile[(DB ? 2 : 1) * RSR * LS];
    __shared__ unsigned long long vbits[FACT ? 1 : RSR * VBW];      // bit c of row r: cell (r, c) counts in num
    __shared__ unsigned long long pbits[EXTRA ? RSR * VBW : 1];     // bit c: cell holds a pixel (statistics only)
    __shared__ double cov_lds[EXTRA ? NW : 1][2 * W];
    // observed over expected: the expected of the staged region's RSR + RSC - 1 diagonals, entry t = expected(|C - R - (RSR - 1)
    // + t|) — fetched by the first 256 threads during the window phase of the block before (one coalesced load), read from
    // here when the region is stored.  (Fetched per cell at store time, sixteen dependent-latency loads per lane: the store
    // phase was a third of the observed-over-expected kernel.)
    __shared__ double exp_lds[OOE ? 256 : 1];
    static_assert(RSR + RSC - 1 <= 256, "one expected value per diagonal of the region");
    // FACT: num[p][q] = N - R[p] - C[q] + RC[p][q] (see fact_batch): the sparse both-masked pairs, and the totals
    __shared__ unsigned rc_lds[ACC][FACT ? W2 : 1];
    __shared__ unsigned fact_tot[FACT ? ACC * (2 * W + 1) : 1];     // per slot: R[W] | C[W] | N
    // PROG: prog_p[w] = block * 32 + first 8-row bucket wave w may still read in that block (16: done with it) — monotone;
    //       prog_s[w] = newest block whose rows [8 w, 8 w + 8) wave w has stored
    __shared__ unsigned prog_p[PROG ? 16 : 1], prog_s[PROG ? 16 : 1];
    const int tid  = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p_raw = lane / NCH;
    const int k  = lane - p_raw * NCH;
    const bool row_ok = p_raw < W;
    const int p  = row_ok ? p_raw : W - 1;               // idle lanes shadow the last row, flush nothing
    unsigned chmask = 0u;                                // bit i: the lane owns window cell (p, k + NCH * i)
#pragma unroll
    for (int i = 0; i < CH; ++i) if (row_ok && k + NCH * i < W) chmask |= 1u << i;

    const bool m_cov   = EXTRA && (a.mode & 0x04u) && a.cov != nullptr;
    const bool use_exp = OOE && ((a.expv != nullptr && a.nexp > 0) || a.n_exp_regions > 0);
    const int  igd     = a.ignore_diags;
    const bool stats   = EXTRA && a.counters != nullptr;
    const double qnan = __builtin_nan("");

    const int G = (int)gridDim.x, g_id = (int)blockIdx.x;
    const int bb = sa.wg_first[g_id], be = sa.wg_first[g_id + 1];   // blocks [bb, be) of the block table
    if (bb >= be) return;                                // (uniform) more workgroups than blocks
    // ONE accumulator set per wave.  With paired tiles (ACC == 2) the waves split into two teams — waves [0, n0) take a
    // block's slot-0 windows (the pair's first tile: they come first in a block, stable sort), the others its slot-1
    // windows — sized by the host to the tiles' shares of the call's windows.  (Two sets per wave, round 2's way, cost
    // 2 * CH more double registers: the difference between 16 and 8 waves per CU, and a wave may have at most 15 LDS
    // reads in flight — it takes 16 waves to keep the LDS busy.)
    int my_slot = 0, team_lo = 0, team_n = NW;
    // a wave's share of its team's windows: [f_lo, f_hi) of them.  Not equal shares: with sixteen waves the oldest and the
    // youngest wave of a SIMD get through a window ~8 % slower than the two in between whatever the priorities (phase clocks,
    // round 3: the workgroup waited 14 % of its time for them at the barrier behind the window loop) — they get 7 % fewer
    auto wave_weight = [&](int w) __attribute__((always_inline)) -> float {
        return NW != 16 ? 1.0f : (float)sa.wgt[w & 15] * (1.0f / 1024.0f);     // (the engine's table: staged_wave_weights, from the phase clocks of profiles/r05_k1q_phases.txt)
    };
    float f_lo = 0.0f, f_hi = 1.0f;
    auto set_shares = [&]() __attribute__((always_inline)) {
        float before = 0.0f, mine = 0.0f, total = 0.0f;
        for (int w = team_lo; w < team_lo + team_n; ++w) {
            const float x = wave_weight(w);
            total += x; if (w < wave) before += x; if (w == wave) mine = x;
        }
        f_lo = total > 0.0f ? before / total : 0.0f;
        f_hi = (wave == team_lo + team_n - 1 || total <= 0.0f) ? 1.0f : (before + mine) / total;
    };
    set_shares();
    unsigned team_w[4] = {0u, 0u, 0u, 0u};               // the unit's team table, 16 bytes
    auto team_at = [&](int s) __attribute__((always_inline)) -> int { return (int)((team_w[s >> 2] >> ((s & 3) * 8)) & 0xffu); };
    auto set_team = [&](int unit) __attribute__((always_inline)) {
        if constexpr (ACC > 1) {
            const uint4 t = *reinterpret_cast<const uint4*>(sa.teams + 16 * (size_t)unit);
            team_w[0] = __builtin_amdgcn_readfirstlane(t.x); team_w[1] = __builtin_amdgcn_readfirstlane(t.y);
            team_w[2] = __builtin_amdgcn_readfirstlane(t.z); team_w[3] = __builtin_amdgcn_readfirstlane(t.w);
            my_slot = 0;
            if constexpr (DUAL) { team_lo = 0; team_n = NW; set_shares(); return; }    // (the table still says which tiles have windows: flush)
#pragma unroll
            for (int s = 1; s < ACC; ++s) my_slot += wave >= team_at(s) ? 1 : 0;     // (teams are contiguous, empty ones have equal ends)
            team_lo = team_at(my_slot); team_n = team_at(my_slot + 1) - team_lo;
            set_shares();
        }
    };
    double   sum[CH];
    unsigned num[CH];
    double   sum1[DUAL ? CH : 1];                        // DUAL: the pair's second tile (slot 1); `sum` is its first
    // the SIMD favours its oldest wave; left alone the youngest of the four finishes its (equal) share of a block's windows
    // 40 % later than the oldest and everybody waits for it at the barrier (phase clocks, round 3): priorities against age
    // (uniform) 0 = the oldest wave of its SIMD.  The priorities are swapped half way through every block's windows (see
    // `windows`): favoured first, then disfavoured, or the other way round — the four waves of a SIMD finish together
    const int age = NW == 16 ? (wave >> 2) : (NW == 8 ? 2 * (wave >> 2) : 0);
    auto set_prio = [&](int pr) __attribute__((always_inline)) {
        if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CH; ++i) { sum[i] = 0.0; num[i] = 0u; }
        if constexpr (DUAL) {
#pragma unroll
            for (int i = 0; i < CH; ++i) sum1[i] = 0.0;
        }
        if (m_cov) for (int t = lane; t < 2 * W; t += kWave) cov_lds[wave][t] = 0.0;
        if constexpr (FACT) {                            // visible after the next barrier
            for (int t = tid; t < ACC * W2; t += NTHR) (&rc_lds[0][0])[t] = 0u;
            for (int t = tid; t < ACC * (2 * W + 1); t += NTHR) fact_tot[t] = 0u;
        }
    };
    zero_acc();
    if constexpr (!FACT) for (int t = tid; t < RSR * VBW; t += NTHR) vbits[t] = 0ull;     // the pad words stay zero
    if (EXTRA) for (int t = tid; t < RSR * VBW; t += NTHR) pbits[t] = 0ull;

    const StagedBlock* __restrict__ blocks = sa.blocks;
    unsigned long long npix = 0;

    // ---- staging, in pieces (see the pipeline in the block loop) ----------------------------------------------------
    // A lane i < RPW looks up row i of the wave's RPW region rows (both 64-column halves); what the row amounts to is then
    // handed to all lanes with readlanes, row by row.  VALU instructions are what the staging costs (counters, round 3:
    // half of the kernel's VALU work was staging), so everything that is uniform per (row, half) is worked out on the
    // SCALAR unit from the broadcast words: the position of half 1's first pixel (half 0's + popcount of its bits), the
    // keep mask (column mask of the table entry, masked-row bit, diagonal mask) — and the mask is applied to the LOAD
    // ADDRESS: a lane without a pixel to keep fetches the zero that pads the count table, so the stored value needs no
    // select at all.
    struct Raw { U64x2 h, w[NH]; unsigned long long rw; double wr; };      // lane i < RPW: index words of the wave's i-th row, its weight
    struct Row { unsigned long long bits[NH], okn[NH]; long long pos; double wr; };    // lane i < RPW: what they amount to
    auto entry_load = [&](int b) __attribute__((always_inline)) -> int {
        return reinterpret_cast<const int*>(blocks + b)[lane & 31];
    };
    auto fld = [&](int ev, int i) __attribute__((always_inline)) -> int { return __builtin_amdgcn_readlane(ev, i); };
    auto fld64 = [&](int ev, int i) __attribute__((always_inline)) -> unsigned long long {
        return ((unsigned long long)(unsigned)fld(ev, i + 1) << 32) | (unsigned)fld(ev, i);
    };
    const bool nf = a.nf_pixels != 0;                    // (uniform) weights of +-inf in the table: products may be NaN
    const int my_rr = wave * RPW + (lane < RPW ? lane : 0);      // the region row this lane looks up
    auto load_raw = [&](int ev, Raw& x) __attribute__((always_inline)) {
        x.h.a = 0ull; x.h.b = 0ull; x.rw = ~0ull; x.wr = 1.0;
#pragma unroll
        for (int h = 0; h < NH; ++h) { x.w[h].a = 0ull; x.w[h].b = 0ull; }
        const int R = fld(ev, 0), ch_end = fld(ev, 6), nblk = fld(ev, 7), row_lo = fld(ev, 21), row_hi = fld(ev, 22);
        const int row = R + my_rr;
        if (lane < RPW && row < ch_end && my_rr >= row_lo && my_rr < row_hi) {
            const char* line = reinterpret_cast<const char*>(a.idx + (unsigned)fld(ev, 8) + (long long)my_rr * nblk);
            x.h = *reinterpret_cast<const U64x2*>(line);                                   // {pos, cum[4]}
            x.w[0] = *reinterpret_cast<const U64x2*>(line + 16 + 8 * (int)((unsigned)fld(ev, 10) & 0xffu));   // {bits[ws], bits[ws+1] | next0}
            if constexpr (NH == 2) {
                const char* line1 = reinterpret_cast<const char*>(a.idx + (unsigned)fld(ev, 9) + (long long)my_rr * nblk);
                x.w[1] = *reinterpret_cast<const U64x2*>(line1 + 16 + 8 * (int)((unsigned)fld(ev, 11) & 0xffu));
            }
            if (!FACT) x.rw = a.badbits[row >> 6];
            if (a.weight) x.wr = a.weight[row];
        }
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
    auto finish_rows = [&](int ev, const Raw& x) __attribute__((always_inline)) -> Row {
        Row r;
        const int R = fld(ev, 0), ch_end = fld(ev, 6), row_lo = fld(ev, 21), row_hi = fld(ev, 22);
        const int row = R + my_rr;
        const bool live = lane < RPW && row < ch_end && my_rr >= row_lo && my_rr < row_hi;
        const bool row_bad = (x.rw >> (row & 63)) & 1ull;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const unsigned wsh = (unsigned)fld(ev, 10 + h);
            const int sh = (int)(wsh >> 8);
            r.bits[h] = x.w[h].a >> sh;
            if (sh) r.bits[h] |= x.w[h].b << (64 - sh);
            r.okn[h] = ok_mask(ev, my_rr, h, row_bad);   // (per lane here: my_rr differs by lane; kept for the validity words)
            if (!live) { r.bits[h] = 0ull; r.okn[h] = 0ull; }
        }
        const unsigned wsh0 = (unsigned)fld(ev, 10);
        const int ws = (int)(wsh0 & 0xffu), sh0 = (int)(wsh0 >> 8);
        const unsigned cum = ws ? (unsigned)(x.h.b >> ((ws - 1) * 16)) & 0xffffu : 0u;
        r.pos = live ? (long long)(x.h.a + cum + (unsigned long long)__popcll(x.w[0].a & ((1ull << sh0) - 1ull))) : 0;
        // a NaN weight (masked bin) contributes 0, as in the `bal` table — unless the table holds infinite weights, whose
        // products cooler leaves as inf / NaN: those are sorted out value by value (store_region)
        r.wr = (nf || x.wr == x.wr) ? x.wr : 0.0;
        return r;
    };
    auto bcast64 = [&](unsigned long long v, int i) __attribute__((always_inline)) -> unsigned long long {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, i), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), i);
        return ((unsigned long long)hi << 32) | lo;
    };
    // the pixels of the region are fetched as their 4-byte COUNTS and balanced at store time, (count * w[row]) * w[col] in
    // the order balance_pixels_kernel (and cooler) multiplies: the same doubles as the resident `bal` table at half the HBM
    // bytes — with 128 x 128 regions the kernel is bound by what it pulls from HBM (round 2, at 64 x 64, it was not)
    auto issue_values = [&](int ev, const Row& r, int (&v)[NRH], double (&wc)[NH]) __attribute__((always_inline)) {
        // pixel positions are 64-bit (tables of 2^30 pixels and more — a deep 10 kb human map — are staged too): the row's first
        // position is a SCALAR, so the row base is a scalar pointer and a lane adds its 32-bit rank; a lane with nothing to
        // keep reads one of the 64 zeros behind the table
        const int* const zeros = a.cnt32 + a.nnz;        // cnt32[nnz .. nnz + 63] are zeros
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int rr = wave * RPW + i;
            const int* rowp = a.cnt32 + (long long)bcast64((unsigned long long)r.pos, i);
            const bool row_bad = !FACT && ((fld64(ev, 16 + 2 * (rr >> 6)) >> (rr & 63)) & 1ull);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const unsigned long long bits = bcast64(r.bits[h], i);
                // FACT: validity is counted elsewhere and masked bins multiply to 0 — only the columns past the chromosome's
                // end (whose "bits" belong to another row) must go; else the full mask
                const unsigned long long keep = bits & (FACT ? fld64(ev, 12 + 2 * h) : ok_mask(ev, rr, h, row_bad));
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bits >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bits, 0u));
                const bool has = __builtin_amdgcn_inverse_ballot_w64(keep);
                const int* ptr = has ? rowp + rank : zeros + lane;
                v[i * NH + h] = *ptr;
                rowp += __builtin_popcountll(bits);      // (scalar) half 1's pixels follow half 0's
            }
        }
        // column weights (1.0 when raw — branch-free: the load then reads the row offsets, a table of the same length)
        const int C = fld(ev, 1);
        const double* wsrc = a.weight ? a.weight : reinterpret_cast<const double*>(a.indptr);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const long long col = (long long)C + 64 * h + lane;
            wc[h] = wsrc[col < a.nbins ? col : a.nbins - 1];                          // (columns past the table are in no window)
        }
        // NOTHING here may look at a loaded value: the memory counter is in-order, so one compare of a weight would park the
        // wave — in the middle of its windows — until every count load above has come back from HBM (round 3: a third of the
        // kernel).  NaN weights are turned into 0 at store time (weight_of)
    };
    // a loaded weight as the factor it stands for: 1.0 when raw, 0 for a masked bin (NaN) unless the table holds infinite
    // weights, whose NaN products are sorted out value by value
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
        const int er = fld(ev, 5);                       // expected region of the block's windows (part of the sort key)
        es.is_scalar = false;
        if (er >= 0 && er < a.n_exp_regions) { const ExpRegion g = a.exp_regions[er]; es.base = a.expv + g.off; es.len = g.len; }
        return es;
    };
    // what exp_lds holds for an expected value e.  Factorised counts (FACT: every diagonal a window reaches has a usable expected, nobody
    // looks at e itself): its RECIPROCAL — the 16 384 cells of a region are then multiplied instead of divided: an f64 division is a
    // dozen instructions, sixteen per lane and block made the store phase of observed-over-expected pile-ups three times the plain one
    // (0.77 against 0.55 ms per 1.1e7 windows).  x * (1 / e) is x / e to within an ulp (the tests' 1e-12 / the reference's 1e-6);
    // e = 0 gives inf or NaN as the quotient does, e = NaN NaN, e = inf 0.  One division per diagonal and block, behind the window loop.
    auto exp_entry = [&](double e) __attribute__((always_inline)) -> double { return FACT ? 1.0 / e : e; };
    // thread t's entry of exp_lds for the region of table entry `ev`
    auto exp_fetch = [&](int ev) __attribute__((always_inline)) -> double {
        if (!use_exp || tid >= RSR + RSC - 1) return qnan;
        const ExpSel es = exp_of(ev);
        long long d = (long long)fld(ev, 1) - (long long)fld(ev, 0) - (RSR - 1) + tid;
        if (d < 0) d = -d;
        return es.at(d);
    };
    auto store_region = [&](auto nf_tag, int ev, Row& r, const int (&v)[NRH], const double (&wc)[NH], const ExpSel& es) __attribute__((always_inline)) {
        constexpr bool NFP = decltype(nf_tag)::value;    // the table holds infinite weights: NaN products are stored as 0
        const int R = fld(ev, 0), C = fld(ev, 1), row_lo = fld(ev, 21), row_hi = fld(ev, 22);
        double wcs[NH];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) wcs[hh] = weight_of(wc[hh]);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int rr = wave * RPW + i;
            if (rr < row_lo || rr >= row_hi) continue;   // (uniform) no window of the block reads this row
            const double wr = __longlong_as_double((long long)bcast64((unsigned long long)__double_as_longlong(r.wr), i));
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                // balanced value (raw: both weights are 1.0; lanes with nothing to keep loaded a zero count)
                double val = count_to_f64(v[i * NH + hh]) * wr * wcs[hh];
                if (NFP) val = (val == val) ? val : 0.0;
                if (OOE) {
                    const double e = exp_lds[64 * hh + lane - rr + (RSR - 1)];      // expected of |col - row|
                    val = FACT ? val * e : val / e;         // (FACT: the entry is the reciprocal — exp_entry)
                    val = (val == val) ? val : 0.0;         // NaN quotients are skipped, inf is kept
                    // usable expected = neither NaN nor zero.  The predicate goes through a register the compiler cannot see
                    // through: hipcc (ROCm 7.2) folds __ballot(e == e && e != 0.0) — and the equivalent v_cmp_class test —
                    // into v_cmp_neq_f64, the UNORDERED not-equal, which lets NaN pass
                    if constexpr (!FACT) {
                        int e_ok = (e == e && e != 0.0) ? 1 : 0;
                        asm volatile("" : "+v"(e_ok));
                        const unsigned long long eok = __ballot(e_ok);
                        if (lane == i) r.okn[hh] &= eok;
                    }
                }
                tile[rr * LS + 64 * hh + lane] = val;
            }
        }
        if (lane < RPW && my_rr >= row_lo && my_rr < row_hi) {
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                if constexpr (!FACT) vbits[my_rr * VBW + hh] = r.okn[hh];
                if (stats) pbits[my_rr * VBW + hh] = r.bits[hh];
            }
        }
    };

    // ---- staging from the dense BAND of counts (K1Args::band: band[row][j] = count(row, row + j), j < band_w) -----------
    // The sparse-to-dense conversion above — index words, presence bits, ranks: ~20 VALU instructions per row half — is
    // done once per TABLE instead of once per staged region: a region row is then 128 consecutive counts of the band, one
    // coalesced load per half with an address that is a scalar plus the lane number.  (Counters, round 3: with 128 x 128
    // regions the staged kernel is bound by VALU issue, half of it the per-region lookups.)  Lanes whose cell is masked
    // (non-FACT), below the diagonal or outside the staged rows read the zero behind the band.
    auto band_issue = [&](int ev, int (&v)[NRH], double (&wc)[NH], double& wrv) __attribute__((always_inline)) {
        const int R = fld(ev, 0), C = fld(ev, 1), ch_end = fld(ev, 6), row_lo = fld(ev, 21), row_hi = fld(ev, 22);
        // (64-bit row bases, all scalar: tables of a million bins and more have bands beyond 2^30 cells)
        const int* const zeros = a.band + (long long)a.nbins * a.band_w;      // >= 129 rows of zeros behind the last row
        if constexpr (FACT) {
            // Factorised counts = every window of the call is clear of the diagonal mask, so NO cell a window reads lies left
            // of the first kept diagonal (a window at (r, c) has c - r >= igd + W - 1), and the engine only stages from the band
            // when no window leaves it: the cells that would need masking are never read.  (Masked bins multiply to zero
            // through their weight.)  A region row is then a plain run of the band: scalar row base + 4 * lane, no vector
            // arithmetic, no predicate.  Cells outside the band's row (left of the diagonal, past its width) read neighbouring
            // rows or the pads around the table (pup_build_index): finite garbage nobody looks at.
            const int row0 = R + wave * RPW;
            const int hi2 = (ch_end - R) < row_hi ? (ch_end - R) : row_hi;             // live rows of the region: [row_lo, hi2)
            const unsigned i_lo = (unsigned)(row_lo - wave * RPW), n_live = (unsigned)(hi2 > row_lo ? hi2 - row_lo : 0);
            const int* rowp = a.band + (long long)row0 * a.band_w + (C - row0);        // (row + 1, C) sits band_w - 1 cells further
            // round 5: a lane takes the NEIGHBOURING columns 2 l and 2 l + 1 of a row — one 8-byte load per row instead of a
            // 4-byte load per half, and at store time one ds_write2_b64: half the memory and LDS instructions of the staging,
            // whose store burst is bound by vector-instruction issue (sixteen waves at once: phase clocks)
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const bool live = (unsigned)i - i_lo < n_live;                          // (uniform) else: zeros, all rows the same lines
                const int* src = live ? rowp : zeros;
                if constexpr (NH == 2) {
                    int2 two;
                    __builtin_memcpy(&two, src + 2 * lane, sizeof(two));                // (4-byte aligned: global_load_dwordx2)
                    v[i * NH] = two.x; v[i * NH + 1] = two.y;
                } else {
#pragma unroll
                    for (int h = 0; h < NH; ++h) v[i * NH + h] = src[64 * h + lane];
                }
                rowp += a.band_w - 1;
            }
        } else
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int rr = wave * RPW + i;
            const int row = R + rr;
            const bool live = row < ch_end && rr >= row_lo && rr < row_hi;
            const bool row_bad = !FACT && ((fld64(ev, 16 + 2 * (rr >> 6)) >> (rr & 63)) & 1ull);
            const int* rowp = a.band + (long long)row * a.band_w + (C - row);         // (scalar) cell (row, C)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                unsigned long long keep = live ? ok_mask(ev, rr, h, row_bad) : 0ull;
                const bool has = __builtin_amdgcn_inverse_ballot_w64(keep);
                const int* ptr = has ? rowp + 64 * h + lane : zeros + lane;
                v[i * NH + h] = *ptr;
            }
        }
        const double* wsrc = a.weight ? a.weight : reinterpret_cast<const double*>(a.indptr);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const long long col = (FACT && NH == 2) ? (long long)C + 2 * lane + h : (long long)C + 64 * h + lane;      // (FACT: columns 2 l, 2 l + 1)
            wc[h] = wsrc[col < a.nbins ? col : a.nbins - 1];      // raw: see issue_values — no loaded value is looked at before the store
        }
        {   // row weights: lane i < RPW holds its row's, broadcast at store time
            const long long row = (long long)R + my_rr;
            wrv = wsrc[row < a.nbins ? row : a.nbins - 1];
        }
    };
    auto band_store = [&](auto nf_tag, int ev, const int (&v)[NRH], const double (&wc)[NH], double wrv, const ExpSel& es, double* dst = nullptr) __attribute__((always_inline)) {
        constexpr bool NFP = decltype(nf_tag)::value;
        double* const out = DB ? dst : tile;
        const int R = fld(ev, 0), C = fld(ev, 1), ch_end = fld(ev, 6), row_lo = fld(ev, 21), row_hi = fld(ev, 22);
        double wcs[NH];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) wcs[hh] = weight_of(wc[hh]);
        const double wrs = weight_of(wrv);
        unsigned long long okn[NH];                      // lane i < RPW: validity bits of its row (non-FACT)
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
                constexpr bool PAIRS = FACT && NH == 2;      // (band_issue: the lane's two values are columns 2 l and 2 l + 1)
                const int colx = PAIRS ? 2 * lane + hh : 64 * hh + lane;
                double val = count_to_f64(v[i * NH + hh]) * wr * wcs[hh];
                if (NFP) val = (val == val) ? val : 0.0;
                if (OOE) {
                    const double e = exp_lds[colx - rr + (RSR - 1)];      // expected of |col - row|
                    val = FACT ? val * e : val / e;         // (FACT: the entry is the reciprocal — exp_entry)
                    val = (val == val) ? val : 0.0;         // NaN quotients are skipped, inf is kept
                    if constexpr (!FACT) {
                        int e_ok = (e == e && e != 0.0) ? 1 : 0;     // (through a register the compiler cannot fold: see store_region)
                        asm volatile("" : "+v"(e_ok));
                        const unsigned long long eok = __ballot(e_ok);
                        if (lane == i) okn[hh] &= eok;
                    }
                }
                out[rr * LS + colx] = val;
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
    // Every wave owns a contiguous SLICE of the block's windows — wave w the windows [w M, (w + 1) M), M = ceil(count / NW) —
    // and walks it in batches of 64, a lane per window: one 2-byte load per lane fetches a batch (the first batch of a block
    // while the previous block is still being piled up, the next batch of a long slice while the current one is), the LDS
    // byte offset of each window's corner is worked out once per batch in vector form, and per window the wave then needs
    // one readlane, one address add, CH LDS reads and CH f64 adds (+ the validity bits, or nothing at all when validity
    // factorises).  Slices are equally long, so the waves reach the block's closing barrier together; the slot-0 windows
    // of a pair come first in a block (stable sort), so at most one wave sees both slots.  (Round 2 dealt windows out
    // round robin from batches every wave held; with one workgroup per CU the per-batch fetch then stalled the whole CU,
    // and the factorised-count bookkeeping of a batch fell on one wave.)
    struct Cur { int R, C, start, first, n, split; unsigned long long rowbad[2], colbad[2]; };   // first, n: the windows of this wave's slot; split (DUAL): the block's first slot-1 window
    unsigned lane_off8 = (unsigned)(uintptr_t)tile + 8u * (unsigned)(p * LS + k);   // LDS byte address of the lane's first cell (DB: in the buffer being read)
    const unsigned vb_base = (unsigned)(uintptr_t)vbits;
    // window `at + lane` of the block, for the lanes below `end`
    auto load_batch = [&](int start, int at, int end) __attribute__((always_inline)) -> int {
        return at + lane < end ? (int)sa.win[start + at + lane] : 0;
    };
    // windows jb <= j < je of the wave's current batch (window j sits in lane j), all of accumulator slot S
    auto run = [&](double (&acc)[CH], const Cur& g, int offv, int drv, int dcv, int jb, int je) __attribute__((always_inline)) {
        auto gather = [&](int jj, double (&v)[CH], unsigned long long& vraw, unsigned& ad0, unsigned& ad1) __attribute__((always_inline)) {
            // single ds_read_b64 each (see lds_read_b64); cells the lane does not own read padding, never flushed
            ad0 = lane_off8 + (unsigned)__builtin_amdgcn_readlane(offv, jj);
            LdsReadRow<0, CH, 8 * NCH>::go(v, ad0);
            vraw = 0ull; ad1 = ad0;
            if constexpr (!FACT) {
                // validity bits of the window's row p from column dc + k on: the dword pair holding bit dc + k
                const int dr = __builtin_amdgcn_readlane(drv, jj), dc = __builtin_amdgcn_readlane(dcv, jj);
                ad1 = vb_base + (unsigned)((dr + p) * (VBW * 8)) + 4u * ((unsigned)(dc + k) >> 5);
                lds_read2_b32(vraw, ad1);
            }
        };
        auto bits_of = [&](int jj, unsigned long long vraw) __attribute__((always_inline)) -> unsigned {
            if (FACT) return 0u;
            return (unsigned)(vraw >> ((__builtin_amdgcn_readlane(dcv, jj) + k) & 31));
        };
        auto extra = [&](int jj) __attribute__((always_inline)) {          // coverage vectors, pixel statistics
            const int dr = __builtin_amdgcn_readlane(drv, jj), dc = __builtin_amdgcn_readlane(dcv, jj);
            if (m_cov && row_ok && k == 0) {
                const double vs = a.cov[g.R + dr + p], ve = a.cov[g.C + dc + p];
                if (vs == vs) cov_lds[wave][p] += vs;
                if (ve == ve) cov_lds[wave][W + p] += ve;
            }
            if (stats) {
                const unsigned c = (unsigned)(dc + k);
                const unsigned* pb = reinterpret_cast<const unsigned*>(pbits + (dr + p) * VBW) + (c >> 5);
                const unsigned long long two = ((unsigned long long)pb[1] << 32) | pb[0];
                const unsigned pw = (unsigned)(two >> (c & 31));
                unsigned m = 0u;
#pragma unroll
                for (int i = 0; i < CH; ++i) if ((chmask >> i) & 1u) m += (pw >> (NCH * i)) & 1u;
                npix += m;
            }
        };
        auto add = [&](const double (&v)[CH], unsigned vw) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < CH; ++i) { acc[i] += v[i]; if (!FACT) num[i] += (vw >> (NCH * i)) & 1u; }
        };
        constexpr int NB = CH + (FACT ? 0 : 1);            // LDS operations of one window's gather
        if constexpr (!EXTRA && NB <= 15) {
            // software pipeline over the windows: the reads of window j + 1 are in flight while window j is added, so the
            // LDS always has a window's worth of reads queued per wave (8 waves per CU: there is no other wave on the SIMD
            // to fill the gap).  `s_waitcnt lgkmcnt(NB)` = "all but the NB youngest LDS operations have returned" — LDS
            // operations return in order, so the older window is complete.  Past the run's last window the pipeline reads
            // that window once more (never added): no branch around the register arrays.
            int jj = jb;
            if (jj < je) {
                double va[CH], vb[CH]; unsigned long long wa, wb; unsigned a0, a1, b0, b1;
                gather(jj, va, wa, a0, a1);
                for (;;) {
                    gather(jj + 1 < je ? jj + 1 : jj, vb, wb, b0, b1);
                    lds_wait_but<NB>(a0, a1); lds_pin(va);
                    if constexpr (!FACT) lds_pin_u64(wa);
                    add(va, bits_of(jj, wa));
                    if (jj + 1 >= je) break;
                    gather(jj + 2 < je ? jj + 2 : jj + 1, va, wa, a0, a1);
                    lds_wait_but<NB>(b0, b1); lds_pin(vb);
                    if constexpr (!FACT) lds_pin_u64(wb);
                    add(vb, bits_of(jj + 1, wb));
                    if (jj + 2 >= je) break;
                    jj += 2;
                }
                // drain the trailing read: its destination registers stay reserved until the data has landed
                lds_wait_all(a0, a1, b0, b1); lds_pin(va); lds_pin(vb);
                if constexpr (!FACT) { lds_pin_u64(wa); lds_pin_u64(wb); }
            }
        } else {
            int jj = jb;
            for (; jj + 1 < je; jj += 2) {                // two windows in flight: both gathered before either is added
                double va[CH], vb[CH]; unsigned long long wa, wb; unsigned a0, a1, a2, a3;
                gather(jj, va, wa, a0, a1);
                gather(jj + 1, vb, wb, a2, a3);
                lds_wait_all(a0, a1, a2, a3); lds_pin(va); lds_pin(vb);
                if constexpr (!FACT) { lds_pin_u64(wa); lds_pin_u64(wb); }   // (FACT: no validity word was read — pinning would materialise a zero)
                add(va, bits_of(jj, wa));
                add(vb, bits_of(jj + 1, wb));
                if (EXTRA) { extra(jj); extra(jj + 1); }
            }
            if (jj < je) {
                double va[CH]; unsigned long long wa; unsigned a0, a1;
                gather(jj, va, wa, a0, a1);
                lds_wait_all(a0, a1, a0, a1); lds_pin(va);
                if constexpr (!FACT) lds_pin_u64(wa);
                add(va, bits_of(jj, wa));
                if (EXTRA) extra(jj);
            }
        }
    };
    // bits [s, s + 32) of the 128-bit mask hi:lo, s in [0, 127]
    auto mask_at = [&](const unsigned long long (&m)[2], int s) __attribute__((always_inline)) -> unsigned {
        const int t = s & 63;
        unsigned long long v;
        if (s < 64) { v = m[0] >> t; if (t) v |= m[1] << (64 - t); } else v = m[1] >> t;
        return (unsigned)v;
    };
    // FACT bookkeeping of a batch, a lane per window: validity of cell (p, q) of a window factorises (no diagonal mask
    // reaches it): valid = !rowbad[p] & !colbad[q], so over the segment num[p][q] = N - R[p] - C[q] + RC[p][q].
    // fact_tot[slot] = {R[W], C[W], N}; rc_lds[slot] = RC (masked row meets masked column: rare).  Integer LDS atomics:
    // exact and order-independent.
    auto fact_batch = [&](const Cur& g, int drv, int dcv, int nb, int js = 0) __attribute__((always_inline)) {
      if constexpr (FACT) {
        if (sa.debug & 8) return;                        // (timing experiments: what the bookkeeping costs; the counts are wrong then)
        constexpr unsigned WMASK = (1u << W) - 1u;
        // (DUAL: the batch's windows [0, js) belong to slot 0, the others to slot 1)
        const int slot = DUAL ? (lane >= js ? 1 : 0) : my_slot;
        const int tb = slot * (2 * W + 1);
        if constexpr (DUAL) {
            if (lane == 0 && js > 0) atomicAdd(&fact_tot[2 * W], (unsigned)js);
            if (lane == 0 && nb > js) atomicAdd(&fact_tot[(2 * W + 1) + 2 * W], (unsigned)(nb - js));
        } else if (lane == 0) atomicAdd(&fact_tot[tb + 2 * W], (unsigned)nb);
        if ((g.rowbad[0] | g.rowbad[1] | g.colbad[0] | g.colbad[1]) == 0ull) return;     // (uniform) no masked bin in the region
        const bool live = lane < nb;
        unsigned rb = live ? mask_at(g.rowbad, drv) & WMASK : 0u;
        const unsigned cbm = live ? mask_at(g.colbad, dcv) & WMASK : 0u;
        unsigned cc = cbm;
        while (cc) { const int q = __ffs((int)cc) - 1; cc &= cc - 1u; atomicAdd(&fact_tot[tb + W + q], 1u); }
        while (rb) {
            const int pp = __ffs((int)rb) - 1; rb &= rb - 1u;
            atomicAdd(&fact_tot[tb + pp], 1u);
            unsigned c2 = cbm;
            while (c2) { const int q = __ffs((int)c2) - 1; c2 &= c2 - 1u; atomicAdd(&rc_lds[slot][pp * W + q], 1u); }
        }
      }
    };
    // this wave's slice [lo, hi) of the block's windows: an equal share of its team's windows (the whole block, or the
    // slot-0 / slot-1 part of it); wf = its first batch, one window per lane, each as its corner inside the region (the
    // value the block sort carried)
    auto slot_range = [&](int ev, int& first, int& n) __attribute__((always_inline)) {
        if constexpr (DUAL) { first = 0; n = fld(ev, 3); return; }
        first = (ACC > 1 && my_slot > 0) ? fld(ev, 22 + my_slot) : 0;
        n = ((ACC > 1 && my_slot < ACC - 1) ? fld(ev, 23 + my_slot) : fld(ev, 3)) - first;
    };
    auto slice_of = [&](int first, int n, int& lo, int& hi) __attribute__((always_inline)) {
        // (both ends by the same formula on neighbouring waves: the slices tile the team's windows exactly)
        lo = (int)((float)n * f_lo + 0.5f); hi = f_hi >= 1.0f ? n : (int)((float)n * f_hi + 0.5f);
        lo = lo < n ? lo : n; hi = hi < n ? hi : n; hi = hi > lo ? hi : lo;
        lo += first; hi += first;
    };
    // `mid` = the look-ahead work of the pipeline (lookups and loads for the next blocks: VALU and address arithmetic, no
    // LDS): every wave does it once, INSIDE its first batch, after 0, 1/6, 2/6 or 3/6 of the batch's windows by its number
    // — so that of the four waves of a SIMD one at a time is busy with it while the others keep the LDS reading.  (All
    // sixteen doing it back to back before the window loop, as in round 2, cost 28 % of the kernel: phase clocks.)
    // windows [jb, je) of a batch whose first js windows belong to slot 0 (DUAL; else everything goes to the wave's one tile)
    auto run2 = [&](const Cur& g, int offv, int drv, int dcv, int jb, int je, int js) __attribute__((always_inline)) {
        if constexpr (DUAL) {
            run(sum, g, offv, drv, dcv, jb, je < js ? je : js);
            run(sum1, g, offv, drv, dcv, jb > js ? jb : js, je);
        } else run(sum, g, offv, drv, dcv, jb, je);
    };
    auto windows = [&](const Cur& g, int wf, auto&& mid) __attribute__((always_inline)) {
        int lo, hi;
        slice_of(g.first, g.n, lo, hi);
        {   // the first batch (possibly empty), with the look-ahead work inside it: `mid` is spelled ONCE — two copies joined
            // by a branch would make hipcc shuttle the register arrays it fills through scratch
            const int drv = wf & ((1 << kWinShift) - 1), dcv = (wf >> kWinShift) & ((1 << kWinShift) - 1);
            const int offv = 8 * (drv * LS + dcv);
            if (lo + kWave < hi) wf = load_batch(g.start, lo + kWave, hi);
            const int nb = (hi - lo) < kWave ? (hi - lo) : kWave;
            int js = g.split - lo; js = js < 0 ? 0 : (js > nb ? nb : js);
            fact_batch(g, drv, dcv, nb, js);
            // (measured: without the split and the priority flip the kernel is 5 % slower — tools/k1_probe.py history in DESIGN §4)
            const int cut = (nb * (wave & 3) * 43) >> 8;                           // ~ nb * (wave & 3) / 6
            const int half = nb >> 1;
            set_prio(age);
            run2(g, offv, drv, dcv, 0, cut, js);
            mid();
            run2(g, offv, drv, dcv, cut, cut > half ? cut : half, js);
            set_prio(3 - age);
            run2(g, offv, drv, dcv, cut > half ? cut : half, nb, js);
        }
        for (int s0 = lo + kWave; s0 < hi; s0 += kWave) {
            const int drv = wf & ((1 << kWinShift) - 1), dcv = (wf >> kWinShift) & ((1 << kWinShift) - 1);
            const int offv = 8 * (drv * LS + dcv);
            if (s0 + kWave < hi) wf = load_batch(g.start, s0 + kWave, hi);         // next batch of a long slice
            const int nb = (hi - s0) < kWave ? (hi - s0) : kWave;
            int js = g.split - s0; js = js < 0 ? 0 : (js > nb ? nb : js);
            fact_batch(g, drv, dcv, nb, js);
            run2(g, offv, drv, dcv, 0, nb, js);
        }
    };
    auto first_coords = [&](int ev, int& wf) __attribute__((always_inline)) {
        int first, n, lo, hi;
        slot_range(ev, first, n);
        slice_of(first, n, lo, hi);
        wf = load_batch(fld(ev, 2), lo, hi);
    };
    auto cur_of = [&](int ev) __attribute__((always_inline)) -> Cur {
        Cur c;
        c.R = fld(ev, 0); c.C = fld(ev, 1); c.start = fld(ev, 2); slot_range(ev, c.first, c.n);
        c.split = DUAL ? fld(ev, 4) : 0;
        c.rowbad[0] = fld64(ev, 16); c.rowbad[1] = fld64(ev, 18);
        c.colbad[0] = ~fld64(ev, 12); c.colbad[1] = ~fld64(ev, 14);     // (also set past the chromosome's end: no eligible window reaches there)
        return c;
    };

    // ---- flush of a segment: merge the waves' register tiles (fixed binary tree: bit-reproducible), write the partial
    // records (segment, slot, workgroup), clear the accumulators.  Uses the region buffer as scratch: every wave is done
    // reading the staged region when this is called, and the next region is stored after it.
    const size_t L = (size_t)W2 + 2 * (size_t)W;
    auto flush = [&](int seg, double* scratch = nullptr) __attribute__((always_inline)) {
        const int fl = seg & 1, unit = seg >> 1;
        double*   mf = DB ? scratch : tile;                               // [NW/2][CH][64] doubles, then the same in u32
        unsigned* mn = reinterpret_cast<unsigned*>(mf + (NW / 2) * CH * kWave);
        __syncthreads();
        // every team merges its waves' tiles into its first wave: binary tree over the position inside the team (both teams
        // at once: their scratch slots are disjoint — the absolute wave number picks the slot)
        auto merge = [&](double (&acc)[CH]) __attribute__((always_inline)) {
            const int r = wave - team_lo;
            for (int step = 1; step < NW; step <<= 1) {
                const int slot_w = wave >> 1;
                if ((r & (2 * step - 1)) == step) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        mf[(slot_w * CH + i) * kWave + lane] = acc[i];
                        if (!FACT) mn[(slot_w * CH + i) * kWave + lane] = num[i];
                    }
                }
                __syncthreads();
                if ((r & (2 * step - 1)) == 0 && r + step < team_n) {
                    const int from = (wave + step) >> 1;
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        acc[i] += mf[(from * CH + i) * kWave + lane];
                        if (!FACT) num[i] += mn[(from * CH + i) * kWave + lane];
                    }
                }
                __syncthreads();
            }
        };
        merge(sum);
        if constexpr (DUAL) merge(sum1);                 // (one team of sixteen: wave 0 ends up with both tiles)
        auto write_rec = [&](int s, const double (&acc)[CH]) __attribute__((always_inline)) {
            const int tl = staged_tile<ACC>(unit, s, sa.PH);
            const int lead = ACC > 1 ? team_at(s) : 0, w_hi = ACC > 1 ? team_at(s + 1) : NW;   // the team's first wave holds its merged tile
            if (tl < 0 || lead >= w_hi) return;                           // (uniform) no such tile / none of its windows in this call
            const size_t rec = (size_t)s * (size_t)(2 * sa.T + G) + (size_t)tl * 2 + (size_t)fl + (size_t)g_id;
            double*   of = a.part_f64 + rec * L;
            unsigned* on = a.part_num + rec * W2;
            if (wave == (DUAL ? 0 : lead)) {
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    if ((chmask >> i) & 1u) {
                        const int cell = map_cell(p, k + NCH * i, W, false, fl);
                        of[cell] = acc[i];
                        if (!FACT) on[cell] = num[i];
                    }
            }
            if constexpr (FACT) {
                // num of every cell from the factorised counts, in the accumulator frame
                const unsigned* tot = fact_tot + s * (2 * W + 1);
                for (int t = tid; t < W2; t += NTHR) {
                    const int pp = t / W, qq = t - pp * W;
                    on[map_cell(pp, qq, W, false, fl)] = tot[2 * W] - tot[pp] - tot[W + qq] + rc_lds[s][t];
                }
            }
            for (int t = tid; t < 2 * W; t += NTHR) {
                double cacc = 0.0;
                if (m_cov) for (int w = lead; w < w_hi; ++w) cacc += cov_lds[w][t];
                of[W2 + t] = cacc;
            }
            if (tid == 0) sa.rec_owner[rec] = (unsigned short)(tl * 2 + fl + 1);
        };
        if constexpr (DUAL) { write_rec(0, sum); write_rec(1, sum1); }
        else {
#pragma unroll
            for (int s = 0; s < ACC; ++s) write_rec(s, sum);
        }
        __syncthreads();                                 // fact_tot / rc_lds / cov_lds have been read
        zero_acc();
        __syncthreads();
    };

    // ---- the block loop, PROGRESSIVE staging (round 6): no barrier between blocks ------------------------------------------------
    // One buffer, one region — but a region row is dead as soon as no window that is still to come reads it.  Every wave puts the
    // windows of its slice into ROW order (buckets of eight rows: a counting sort over the lanes of a batch, ballots and one
    // ds_permute), publishes the bucket of its next window, and owns eight region rows as before: wave w stores rows [8 w, 8 w + 8)
    // of block b + 1 as soon as EVERY wave's next window of block b lies below them, and a window of block b + 1 is read once the
    // waves owning its rows have stored them.  The store burst and both barriers of the barrier form (26 % of its clocks: phase
    // clocks, round 5) dissolve into the window phase of the neighbouring blocks; a wave may run most of a block ahead of the
    // slowest one.  A wave enters block b + 1 only after storing its own rows of it, so whoever waits for rows of b + 1 waits for
    // waves still in block b, which wait for nobody ahead of them: no cycle.  A change of segment (the accumulators are flushed
    // through the region buffer) falls back to the barrier form for that one block.  Same integers as every other kernel; the sums
    // are added in (bucket, arrival) order — fixed, bit-reproducible.
    if constexpr (PROG) {
        constexpr int KMAX = (RSR - W) >> 3;                            // last bucket that holds a corner (W = 21: 13; W <= 7: 15)
        constexpr int IDLE = 16;                                        // bucket of the lanes behind a batch's windows
        constexpr int DONE = 17;                                        // progress value "no window of this block left"
        if (tid < 16) { prog_p[tid] = 0u; prog_s[tid] = 0u; }
        int ev0 = entry_load(bb), ev1 = ev0, evn = ev0;
        int v[NRH];
        double wc[NH], wrv = 1.0;
        int w0f, w1f = 0;
        const ExpSel es_none = exp_of(ev0);                            // (no expected in this instantiation)
        {   // prologue: stage block bb without overlap, request the counts of bb + 1
            band_issue(ev0, v, wc, wrv);
            set_team(fld(ev0, 20) >> 1);
            first_coords(ev0, w0f);
            if (bb + 1 < be) ev1 = entry_load(bb + 1);
            __syncthreads();
            if (nf) band_store(std::true_type{}, ev0, v, wc, wrv, es_none); else band_store(std::false_type{}, ev0, v, wc, wrv, es_none);
            if (bb + 1 < be) band_issue(ev1, v, wc, wrv);
            __syncthreads();
        }
        long long tk[6] = {0, 0, 0, 0, 0, 0}, tmid = 0;
        const bool timed = sa.timing != nullptr;
        auto tick = [&]() __attribute__((always_inline)) -> long long { return timed ? (long long)__builtin_readcyclecounter() : 0; };
        // all sixteen waves at or past `target` = block * 32 + bucket ?
        auto all_past = [&](unsigned target) __attribute__((always_inline)) -> bool {
            const unsigned pv = prog_p[lane & 15];
            return (__ballot(pv >= target) & 0xffffull) == 0xffffull;
        };
        // waves 0 .. n-1 have stored their rows of block lb: n
        auto stored_upto = [&](unsigned lb) __attribute__((always_inline)) -> int {
            const unsigned sv = prog_s[lane & 15];
            const unsigned long long okb = __ballot(sv >= lb) & 0xffffull;
            return (int)__builtin_ctzll(~okb);
        };
        for (int b = bb; b < be; ++b) {
            const unsigned lb = (unsigned)(b - bb);
            const bool has1 = b + 1 < be, has2 = b + 2 < be;
            const long long t0 = tick();
            const int seg0 = fld(ev0, 20), seg1 = has1 ? fld(ev1, 20) : seg0;
            const bool barrier_mode = has1 && seg1 != seg0;            // (uniform) another segment follows: the accumulators go out through
                                                                       // the region buffer — the barrier form for this one block
            if (has1) first_coords(ev1, w1f);                          // (same unit: same teams — a new unit reloads them below)
            if (has2) evn = entry_load(b + 2);
            const Cur g = cur_of(ev0);
            bool todo_store = has1, force = false;
            int spins = 0;
            int avail = __builtin_amdgcn_readfirstlane(lb == 0 ? 16 : 0);                              // waves whose rows of THIS block are known to be stored
            unsigned published = lb * 32u;
            const unsigned my_target = lb * 32u + (unsigned)(wave + 1 <= KMAX ? wave + 1 : DONE);     // everybody past this: my rows are dead
            auto publish = [&](int bucket) __attribute__((always_inline)) {
                const unsigned val = (unsigned)__builtin_amdgcn_readfirstlane((int)(lb * 32u + (unsigned)bucket));
                if (val != published) { published = val; if (lane == 0) prog_p[wave] = val; }
            };
            int lo, hi;
            slice_of(g.first, g.n, lo, hi);
            lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);     // (float arithmetic: they come out of vector registers)
            const long long t1 = tick();
            long long t2 = t1;
            int wf = w0f;
            for (int s0 = lo; ; s0 += kWave) {
                const int left = hi - s0;
                const int nb = __builtin_amdgcn_readfirstlane(left < 0 ? 0 : (left < kWave ? left : kWave));
                const bool last_batch = s0 + kWave >= hi;
                // the batch in row-bucket order: rank = windows in lower buckets + lanes before this one in its bucket; lane k of
                // `bstart` keeps the number of windows in buckets below k = where bucket k begins in the sorted batch
                const long long ts0 = tick();
                const int dr_u = wf & ((1 << kWinShift) - 1), dc_u = (wf >> kWinShift) & ((1 << kWinShift) - 1);
                if (!last_batch) wf = load_batch(g.start, s0 + kWave, hi);             // next batch of a long slice
                if (nb > 0) fact_batch(g, dr_u, dc_u, nb);
                const int bk_u = lane < nb ? (dr_u >> 3) : IDLE;       // (idle lanes behind every window)
                int rank = 0, base = 0, bstart = 0;
#pragma unroll
                for (int kb = 0; kb <= IDLE; ++kb) {
                    const unsigned long long mk = __ballot(bk_u == kb);
                    const int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                    rank = bk_u == kb ? base + before : rank;
                    bstart = lane == kb ? base : bstart;
                    base += (int)__builtin_popcountll(mk);
                }
                const int packed = __builtin_amdgcn_ds_permute(rank << 2, dr_u | (dc_u << 8));     // lane `rank` receives this window
                const int drv = packed & 0xff;
                int offv = 8 * (drv * LS + ((packed >> 8) & 0xff));
                if (sa.debug & 8) offv = 8 * (dr_u * LS + dc_u);       // (timing experiment, wrong sums: the reads in the unsorted order)
                tk[3] += tick() - ts0;
                int next_check = 0;                                    // first window of the next bucket: where the slow path below runs again
                // before the reads of window j, the first of its bucket, are issued: what this wave may still read is published (the
                // window before it is still in flight: ITS bucket), the rows of the new bucket must have been stored
                auto boundary = [&](int j) __attribute__((always_inline)) {
                    const int kb = __builtin_amdgcn_readlane(drv, j) >> 3;
                    if (last_batch) publish(j > 0 ? (__builtin_amdgcn_readlane(drv, j - 1) >> 3) : kb);
                    int need = (8 * kb + 7 + W - 1) >> 3;              // owner of the last row a window of this bucket may read
                    need = need < NW - 1 ? need : NW - 1;              // (corners stop at row RSR - W: the region's last row is wave 15's)
                    if (need >= avail) {
                        const long long e0 = tick();
                        while (need >= avail) {
                            avail = __builtin_amdgcn_readfirstlane(stored_upto(lb));
                            if (need >= avail) {
                                __builtin_amdgcn_s_sleep(kProgSleep);
                                if (++spins > kProgSpinLimit) { if (lane == 0) atomicExch(a.err, 2); avail = 16; }     // (never seen: a wedged workgroup ends with an error instead of hanging the queue)
                            }
                        }
                    }
                    next_check = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_readlane(bstart, kb + 1));
                };
                auto gather = [&](int j, double (&vv)[CH], unsigned& ad) __attribute__((always_inline)) {
                    ad = lane_off8 + (unsigned)__builtin_amdgcn_readlane(offv, j);
                    LdsReadRow<0, CH, 8 * NCH>::go(vv, ad);
                };
                if (!last_batch) publish(0);
                // The batch goes through in SEGMENTS: a segment ends where this wave may first be allowed to store its rows of the next
                // block (the start of the bucket behind its rows; one bucket at a time while the others are not there yet) — the ONE
                // place where block b + 1 is stored and b + 2 requested — else at the batch's end.  Inside a segment the window loop is
                // K1q's (two windows in flight, no check but the bucket boundary).
                int j0 = 0;
                for (;;) {
                    const bool tail = j0 >= nb && last_batch;
                    if (tail) {
                        publish(DONE);
                        if (t2 == t1) t2 = tick();
                        if (barrier_mode && !force) {
                            flush(seg0);
                            if (ACC > 1 && (seg1 >> 1) != (seg0 >> 1)) { set_team(seg1 >> 1); first_coords(ev1, w1f); }
                            force = true;
                        }
                    }
                    if (todo_store) {
                        bool go = force;
                        if (!force && !barrier_mode && published >= my_target) {
                            const long long e0 = tick();
                            go = all_past(my_target);
                        }
                        if (go) {
                            const long long m0 = tick();
                            if (!(sa.debug & 2)) { if (nf) band_store(std::true_type{}, ev1, v, wc, wrv, es_none); else band_store(std::false_type{}, ev1, v, wc, wrv, es_none); }
                            if (lane == 0) prog_s[wave] = lb + 1u;     // (behind the row stores: LDS operations of a wave stay in order)
                            if (force && barrier_mode) __syncthreads();
                            if (has2 && !(sa.debug & 2)) band_issue(evn, v, wc, wrv);
                            todo_store = false;
                            tmid += tick() - m0;
                        }
                    }
                    if (j0 >= nb) {
                        if (!tail || !todo_store) break;
                        __builtin_amdgcn_s_sleep(kProgSleep);
                        if (++spins > kProgSpinLimit) { if (lane == 0) atomicExch(a.err, 2); force = true; }
                        continue;
                    }
                    int j1 = nb;
                    if (todo_store && !barrier_mode && last_batch && !(sa.debug & 1)) {      // (debug 1, timing experiments: one segment, the store behind the last window)
                        const int kb0 = __builtin_amdgcn_readlane(drv, j0) >> 3;
                        const int kt = kb0 + 1 > wave + 1 ? kb0 + 1 : wave + 1;
                        j1 = kt <= KMAX ? __builtin_amdgcn_readlane(bstart, kt) : nb;
                    }
                    j1 = __builtin_amdgcn_readfirstlane(j1);
                    {   // windows [j0, j1)
                        const long long tl0 = tick();
                        tk[5] += timed ? (j1 - j0) : 0;
                        double va[CH], vb[CH]; unsigned a0, b0;
                        int jj = __builtin_amdgcn_readfirstlane(j0);
                        if (sa.debug & 1) next_check = nb;              // (timing experiment: no bucket boundaries)
                        if (jj >= next_check) boundary(jj);
                        gather(jj, va, a0);
                        for (;;) {
                            const bool n1 = jj + 1 < j1;
                            if (n1 && jj + 1 >= next_check) boundary(jj + 1);
                            gather(n1 ? jj + 1 : jj, vb, b0);
                            lds_wait_but<CH>(a0, a0); lds_pin(va);
#pragma unroll
                            for (int i = 0; i < CH; ++i) sum[i] += va[i];
                            if (!n1) break;
                            const bool n2 = jj + 2 < j1;
                            if (n2 && jj + 2 >= next_check) boundary(jj + 2);
                            gather(n2 ? jj + 2 : jj + 1, va, a0);
                            lds_wait_but<CH>(b0, b0); lds_pin(vb);
#pragma unroll
                            for (int i = 0; i < CH; ++i) sum[i] += vb[i];
                            if (!n2) break;
                            jj = __builtin_amdgcn_readfirstlane(jj + 2);
                        }
                        lds_wait_all(a0, b0, a0, b0); lds_pin(va); lds_pin(vb);      // the trailing read
                        tk[4] += tick() - tl0;
                    }
                    j0 = j1;
                    if (last_batch && j0 < nb) publish(__builtin_amdgcn_readlane(drv, j0) >> 3);   // (everything before j0 has landed)
                }
                if (last_batch) break;
            }
            if (!has1) { flush(seg0); if (timed) { tk[0] += t1 - t0; tk[1] += t2 - t1; } break; }
            const long long t3 = tick();
            ev0 = ev1; w0f = w1f; ev1 = evn;
            if (timed) { tk[0] += t1 - t0; tk[1] += t2 - t1; tk[2] += t3 - t2; }
        }
        if (timed && lane == 0) {
            long long* o = sa.timing + ((size_t)g_id * NW + wave) * 8;
            for (int i = 0; i < 6; ++i) o[i] = tk[i];
            o[6] = be - bb; o[7] = tmid;
        }
    } else
    // ---- the block loop, two buffers: region b is piled up from one buffer while b+1 is stored into the other (its counts were
    // requested while b-1 was piled up), b+2's counts are requested and b+3's table entry is on its way -------------------------
    if constexpr (DB) {
        constexpr int BUF = RSR * LS;                    // doubles per buffer
        int ev0 = entry_load(bb), ev1 = ev0, ev2 = ev0, evn = ev0;
        int v[NRH];
        double wc[NH], wrv = 1.0;
        int w0f, w1f = 0;
        int cur = 0;
        const ExpSel es_none = exp_of(ev0);              // (no expected in this instantiation)
        {   // prologue: stage block bb into buffer 0 without overlap, request the counts of bb + 1
            band_issue(ev0, v, wc, wrv);
            set_team(fld(ev0, 20) >> 1);
            first_coords(ev0, w0f);
            if (bb + 1 < be) ev1 = entry_load(bb + 1);
            if (bb + 2 < be) evn = entry_load(bb + 2);
            __syncthreads();
            if (nf) band_store(std::true_type{}, ev0, v, wc, wrv, es_none, tile); else band_store(std::false_type{}, ev0, v, wc, wrv, es_none, tile);
            if (bb + 1 < be) band_issue(ev1, v, wc, wrv);
            __syncthreads();
        }
        long long tk[6] = {0, 0, 0, 0, 0, 0}, tmid = 0;
        const bool timed = sa.timing != nullptr;
        auto tick = [&]() __attribute__((always_inline)) -> long long { return timed ? (long long)__builtin_readcyclecounter() : 0; };
        for (int b = bb; b < be; ++b) {
            const bool has1 = b + 1 < be, has2 = b + 2 < be;
            const long long t0 = tick();
            double* const other = tile + (cur ^ 1) * BUF;
            auto lookahead = [&]() __attribute__((always_inline)) {
                const long long m0 = tick();
                if (has1) {
                    first_coords(ev1, w1f);
                    if (!(sa.debug & 2)) { if (nf) band_store(std::true_type{}, ev1, v, wc, wrv, es_none, other); else band_store(std::false_type{}, ev1, v, wc, wrv, es_none, other); }
                }
                if (has2) { ev2 = evn; if (!(sa.debug & 2)) band_issue(ev2, v, wc, wrv); if (b + 3 < be) evn = entry_load(b + 3); }
                tmid += tick() - m0;
            };
            const Cur c0 = cur_of(ev0);
            const long long t1 = tick();
            windows(c0, w0f, lookahead);
            const long long t2 = tick();
            const int seg0 = fld(ev0, 20);
            if (!has1) { flush(seg0, tile + cur * BUF); break; }
            const int seg1 = fld(ev1, 20);
            if (seg1 != seg0) {                          // (uniform) the next block belongs to another segment
                flush(seg0, tile + cur * BUF);           // (its barriers also close block b: region b + 1 sits complete in the other buffer)
                if (ACC > 1 && (seg1 >> 1) != (seg0 >> 1)) { set_team(seg1 >> 1); first_coords(ev1, w1f); }   // other unit: other teams
            } else __syncthreads();                      // every wave is done reading region b and has stored its rows of b + 1
            const long long t3 = tick();
            cur ^= 1;
            lane_off8 = (unsigned)(uintptr_t)(tile + cur * BUF) + 8u * (unsigned)(p * LS + k);
            ev0 = ev1; w0f = w1f; ev1 = ev2;
            if (timed) { tk[0] += t1 - t0; tk[1] += t2 - t1; tk[2] += t3 - t2; tk[5] += tick() - t3; }
        }
        if (timed && lane == 0) {
            long long* o = sa.timing + ((size_t)g_id * NW + wave) * 8;
            for (int i = 0; i < 6; ++i) o[i] = tk[i];
            o[6] = be - bb; o[7] = tmid;
        }
    } else
    // ---- the block loop, band staging: region b is piled up while b+1's counts and b+2's table entry are on their way --------
    if constexpr (BAND) {
        int ev0 = entry_load(bb), ev1 = ev0, evn = ev0;
        int v[NRH];
        double wc[NH], wrv = 1.0;
        int w0f, w1f = 0;
        {   // prologue: stage block bb without overlap
            band_issue(ev0, v, wc, wrv);
            set_team(fld(ev0, 20) >> 1);
            first_coords(ev0, w0f);
            if (bb + 1 < be) ev1 = entry_load(bb + 1);
            const ExpSel es0 = exp_of(ev0);
            if constexpr (OOE) { if (tid < 256) exp_lds[tid] = exp_entry(exp_fetch(ev0)); }
            __syncthreads();
            if (nf) band_store(std::true_type{}, ev0, v, wc, wrv, es0); else band_store(std::false_type{}, ev0, v, wc, wrv, es0);
            __syncthreads();
        }
        long long tk[6] = {0, 0, 0, 0, 0, 0}, tmid = 0;
        double e_next = 0.0;
        const bool timed = sa.timing != nullptr;
        auto tick = [&]() __attribute__((always_inline)) -> long long { return timed ? (long long)__builtin_readcyclecounter() : 0; };
        for (int b = bb; b < be; ++b) {
            const bool has1 = b + 1 < be, has2 = b + 2 < be;
            const long long t0 = tick();
            auto lookahead = [&]() __attribute__((always_inline)) {
                const long long m0 = tick();
                if constexpr (OOE) { if (has1) e_next = exp_fetch(ev1); }
                if (has1) { if (!(sa.debug & 2)) band_issue(ev1, v, wc, wrv); first_coords(ev1, w1f); }
                if (has2) evn = entry_load(b + 2);
                tmid += tick() - m0;
            };
            const Cur c0 = cur_of(ev0);
            const long long t1 = tick();
            windows(c0, w0f, lookahead);
            if constexpr (OOE) { if (has1 && tid < 256) exp_lds[tid] = exp_entry(e_next); }     // (last read when region b was stored; visible after the barrier below)
            const long long t2 = tick();
            const int seg0 = fld(ev0, 20);
            if (!has1) { flush(seg0); break; }
            const ExpSel es1 = exp_of(ev1);
            const int seg1 = fld(ev1, 20);
            if (seg1 != seg0) {                          // (uniform) the next block belongs to another segment
                flush(seg0);
                if (ACC > 1 && (seg1 >> 1) != (seg0 >> 1)) { set_team(seg1 >> 1); first_coords(ev1, w1f); }   // other unit: other teams
            } else __syncthreads();                      // every wave is done reading region b
            const long long t3 = tick();
            if (!(sa.debug & 2)) { if (nf) band_store(std::true_type{}, ev1, v, wc, wrv, es1); else band_store(std::false_type{}, ev1, v, wc, wrv, es1); }
            const long long t4 = tick();
            __syncthreads();
            const long long t5 = tick();
            ev0 = ev1; w0f = w1f; ev1 = evn;
            if (timed) { tk[0] += t1 - t0; tk[1] += t2 - t1; tk[2] += t3 - t2; tk[3] += t4 - t3; tk[4] += t5 - t4; tk[5] += tick() - t5; }
        }
        if (timed && lane == 0) {
            long long* o = sa.timing + ((size_t)g_id * NW + wave) * 8;
            for (int i = 0; i < 6; ++i) o[i] = tk[i];
            o[6] = be - bb; o[7] = tmid;
        }
    } else
    // ---- the block loop: region b is piled up while b+1's values, b+2's index lines and b+3's table entry are on their way
    {
        int ev0 = entry_load(bb), ev1 = ev0, ev2 = ev0, evn = ev0;   // entries are consumed one stage after their load was issued
        Raw x1, x2;
        Row rw0, rw1;
        int v[NRH];
        double wc[NH];
        int w0f, w1f = 0;
        {   // prologue: stage block bb without overlap, start the lookups of bb+1
            Raw x0;
            load_raw(ev0, x0);
            set_team(fld(ev0, 20) >> 1);
            first_coords(ev0, w0f);
            if (bb + 1 < be) { ev1 = entry_load(bb + 1); load_raw(ev1, x1); }
            if (bb + 2 < be) evn = entry_load(bb + 2);
            rw0 = finish_rows(ev0, x0);
            issue_values(ev0, rw0, v, wc);
            const ExpSel es0 = exp_of(ev0);
            if constexpr (OOE) { if (tid < 256) exp_lds[tid] = exp_entry(exp_fetch(ev0)); }
            __syncthreads();
            if (nf) store_region(std::true_type{}, ev0, rw0, v, wc, es0); else store_region(std::false_type{}, ev0, rw0, v, wc, es0);
            __syncthreads();
        }
        // (diagnostics: per-wave clocks of the phases of the loop below, only when sa.timing is set)
        long long tk[6] = {0, 0, 0, 0, 0, 0}, tmid = 0;
        double e_next = 0.0;
        const bool timed = sa.timing != nullptr;
        auto tick = [&]() __attribute__((always_inline)) -> long long { return timed ? (long long)__builtin_readcyclecounter() : 0; };
        for (int b = bb; b < be; ++b) {
            const bool has1 = b + 1 < be, has2 = b + 2 < be;
            const long long t0 = tick();
            auto lookahead = [&]() __attribute__((always_inline)) {
                const long long m0 = tick();
                if constexpr (OOE) { if (has1) e_next = exp_fetch(ev1); }
                if (has1) { rw1 = finish_rows(ev1, x1); if (!(sa.debug & 2)) issue_values(ev1, rw1, v, wc); first_coords(ev1, w1f); }
                if (has2) { ev2 = evn; load_raw(ev2, x2); if (b + 3 < be) evn = entry_load(b + 3); }
                tmid += tick() - m0;
            };
            const Cur c0 = cur_of(ev0);
            const long long t1 = tick();
            windows(c0, w0f, lookahead);
            if constexpr (OOE) { if (has1 && tid < 256) exp_lds[tid] = exp_entry(e_next); }     // (last read when region b was stored; visible after the barrier below)
            const long long t2 = tick();
            const int seg0 = fld(ev0, 20);
            if (!has1) { flush(seg0); break; }
            const ExpSel es1 = exp_of(ev1);
            const int seg1 = fld(ev1, 20);
            if (seg1 != seg0) {                          // (uniform) the next block belongs to another segment
                flush(seg0);
                if (ACC > 1 && (seg1 >> 1) != (seg0 >> 1)) { set_team(seg1 >> 1); first_coords(ev1, w1f); }   // other unit: other teams
            } else __syncthreads();                      // every wave is done reading region b
            const long long t3 = tick();
            if (!(sa.debug & 2)) { if (nf) store_region(std::true_type{}, ev1, rw1, v, wc, es1); else store_region(std::false_type{}, ev1, rw1, v, wc, es1); }
            const long long t4 = tick();
            __syncthreads();
            const long long t5 = tick();
            ev0 = ev1; w0f = w1f;
            if (has2) { ev1 = ev2; x1 = x2; }
            if (timed) { tk[0] += t1 - t0; tk[1] += t2 - t1; tk[2] += t3 - t2; tk[3] += t4 - t3; tk[4] += t5 - t4; tk[5] += tick() - t5; }
        }
        if (timed && lane == 0) {
            long long* o = sa.timing + ((size_t)g_id * NW + wave) * 8;
            for (int i = 0; i < 6; ++i) o[i] = tk[i];
            o[6] = be - bb; o[7] = tmid;
        }
    }
    if (stats) {
        for (int off = 32; off > 0; off >>= 1) npix += __shfl_down(npix, off);
        if (lane == 0) atomicAdd(&a.counters[0], npix);
    }
}

// ---- block-order prepass of K1q ------------------------------------------------------------------------------------
// key of a snippet: (segment, expected region, block row, block col).  Segment = the (tile pair | tile, flip) run the
// snippet is piled up with; `pair_half` = T/2 when tile t shares its pass with tile t + T/2, 0 when every tile has its own
// pass.  Pairs go through the kernel one by one (`set_pairs` = 1: slot = t / (T/2) in bit kWinSlotBit of the value) or in
// sets of kSetPairs (the slot, 0..7, is then the key's lowest digit: see staged_tile).  Also checks that the window is one the
// rank-bitmap index covers (cis, inside one chromosome), counts the others, and counts the windows a diagonal mask reaches.
template <typename KeyT, int SIDE_R, int SIDE_C /* block sides when known at compile time (division by a constant), else 0 */>
__global__ __launch_bounds__(256) void staged_key_kernel(const int* __restrict__ r0, const int* __restrict__ c0, long long n,
                                                         const long long* __restrict__ seg_end, int nseg2t, int pair_half, int set_pairs,
                                                         const IdxChrom* __restrict__ chroms, int n_chrom,
                                                         const unsigned short* __restrict__ bin_chrom, long long nbins,
                                                         const int* __restrict__ brow_base /* [n_chrom] block rows before the chromosome */,
                                                         const ExpRegion* __restrict__ eregs, int n_eregs,
                                                         int W, int BR, int BC, int sh_br, int sh_er, int sh_seg,
                                                         int seg_shift /* 1: no flipped windows, the flip bit is left out */,
                                                         int clear_gap /* igd + W - 1 */, int far_gap /* shortest expected vector */,
                                                         int band_w /* 0: no band table */,
                                                         KeyT* __restrict__ keys, unsigned short* __restrict__ vals,
                                                         unsigned* __restrict__ counters /* [0] ineligible, [1] windows a diagonal mask
                                                                                            reaches, [2] windows leaving the dense band */,
                                                         unsigned* __restrict__ hi_hist /* nullable: tilehist[workgroup][hi_bins], counts of key >> hi_shift (pup_bin.hpp) */,
                                                         int hi_shift, int hi_bins, int per_thread /* windows per thread: 4, or 32 = a binning tile per workgroup */,
                                                         int rel_bc /* > 0: the key's column block is counted from the block row's own diagonal, biased by rel_bc - 1 (see staged_run) */) {
    // Small tables go to LDS once per workgroup (the chromosome table in chunks of the call's dynamic LDS: 12 bytes per chromosome,
    // the launch sizes it — see launch_key_kernel), so per window the chain of dependent global loads is r0 -> bin_chrom only.
    // What the kernel costs is VALU issue (counters, round 4: 156 vector instructions per 64 windows, 4 clocks each on a SIMD, made
    // 45 of its 68 us; a select between an LDS and a global table had turned every table read into a FLAT load behind
    // `s_waitcnt vmcnt(0)` — a wait for the previous window's stores): all index arithmetic is 32-bit (n < 2^31: staged_run),
    // the key is put together in 32 bits when it is 32 bits wide, broadcasts are readlanes.
    const int kPer = per_thread;
    extern __shared__ long long s_dyn[];
    int* const s_seg = reinterpret_cast<int*>(s_dyn);                 // [nseg2t] (tile, flip) run ends
    unsigned* const s_hh = reinterpret_cast<unsigned*>(s_seg + nseg2t);           // [hi_bins] high-digit counts of this workgroup's windows
    int* const s_cs = reinterpret_cast<int*>(s_hh + hi_bins);         // [n_chrom] first bin | [n_chrom] end | [n_chrom] block rows before
    int* const s_ce = s_cs + n_chrom;
    int* const s_bb = s_ce + n_chrom;
    for (int k = threadIdx.x; k < n_chrom; k += blockDim.x) { s_cs[k] = chroms[k].start; s_ce[k] = chroms[k].end; s_bb[k] = brow_base[k]; }
    for (int k = threadIdx.x; k < nseg2t; k += blockDim.x) s_seg[k] = (int)seg_end[k];
    if (hi_hist) for (int k = threadIdx.x; k < hi_bins; k += blockDim.x) s_hh[k] = 0u;
    __syncthreads();
    unsigned bad = 0u, near_c = 0u, far_c = 0u;
    const int lane = threadIdx.x & 63;
    const int n32 = (int)n, n_last = n32 - 1, nb_last = (int)nbins - 1;
    // the (tile, flip) run of a window and what follows from it — pass segment, accumulator slot, slot digit of the key — depend on
    // the window's NUMBER only, and a workgroup's 8192 consecutive windows nearly always lie in one run: found once per workgroup
    // (per window, the bisection of the run ends and two integer divisions were a third of the kernel's vector instructions)
    struct SegInfo { unsigned seg, slot, kslot, kbits; };
    auto run_of = [&](int i) -> int {
        int lo = 0, hi = nseg2t;
        while (lo < hi) { const int m = (lo + hi) >> 1; if (s_seg[m] <= i) lo = m + 1; else hi = m; }
        return lo;
    };
    auto seg_of = [&](int lo) -> SegInfo {
        const int t = lo >> 1, f = lo & 1;                   // tile, flip state of the snippet
        SegInfo si; si.seg = (unsigned)lo; si.slot = 0u; si.kslot = 0u; si.kbits = 0u;
        if (pair_half > 0) {
            const int kind = t / pair_half, g = t - kind * pair_half;
            if (set_pairs > 1) { si.kslot = (unsigned)(kind * set_pairs + g % set_pairs); si.kbits = kSetSlotBits; si.seg = (unsigned)((g / set_pairs) * 2 + f); }
            else { si.slot = (unsigned)kind; si.seg = (unsigned)(g * 2 + f); }
        }
        return si;
    };
    const int tile_first = blockIdx.x * kPer * (int)blockDim.x;
    int tile_last = tile_first + kPer * (int)blockDim.x - 1;
    tile_last = tile_last < n_last ? tile_last : n_last;
    const int lo_a = __builtin_amdgcn_readfirstlane(run_of(tile_first)), lo_b = __builtin_amdgcn_readfirstlane(run_of(tile_last));
    const bool one_run = lo_a == lo_b;                       // (uniform)
    const SegInfo si_a = seg_of(lo_a);
    // Windows are taken kKeyBatch at a time: first ALL their coordinate loads, then ALL their chromosome lookups, then the
    // arithmetic — and no loaded value is looked at (not even selected against a default) before the last load of its kind is
    // issued: indices are clamped instead of predicated.
    constexpr int kKeyBatch = 8;
    for (int u0 = 0; u0 < kPer; u0 += kKeyBatch) {
    int rb[kKeyBatch], cb[kKeyBatch], cab[kKeyBatch];
    const int i_base = (blockIdx.x * kPer + u0) * (int)blockDim.x + (int)threadIdx.x;
#pragma unroll
    for (int k = 0; k < kKeyBatch; ++k) {
        int i = i_base + k * (int)blockDim.x;
        i = i < n_last ? i : n_last;
        rb[k] = r0[i]; cb[k] = c0[i];
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
        const int i = i_base + k * (int)blockDim.x;
        const bool live = i < n32;
        const int r = live ? rb[k] : 0, c = live ? cb[k] : 0;
        SegInfo si = si_a;
        if (!one_run) si = seg_of(run_of(i));
        const unsigned seg = si.seg, slot = si.slot, kslot = si.kslot, kbits = si.kbits;
        bool ok = r >= 0 && c >= 0 && r <= nb_last;
        bool rel_over = false;
        unsigned br = 0, bc = 0, er = 0;
        unsigned inside = 0u;                                // the window's corner inside its block: all the staged kernel needs
        if (ok) {
            const int ca = cab[k];
            const int cs = s_cs[ca], ce = s_ce[ca];
            ok = r >= cs && r + W <= ce && c >= cs && c + W <= ce;
            if (ok) {
                constexpr int kR = SIDE_R ? SIDE_R : 1, kC = SIDE_C ? SIDE_C : 1;   // (a zero divisor must not even be spelled)
                const int qr = SIDE_R ? (r - cs) / kR : (r - cs) / BR, qc = SIDE_C ? (c - cs) / kC : (c - cs) / BC;
                br = (unsigned)(s_bb[ca] + qr);             // increasing over the genome, compact
                bc = (unsigned)qc;
                if (rel_bc) {
                    // column blocks counted from the one under the block row's first bin: a window inside the band is at most
                    // (band + block side) / block side of them away — 4 bits instead of the 8 a chromosome's width takes; a window
                    // whose number does not fit the field (beyond the band, or below the diagonal) is reported with the windows
                    // that leave the band: the host then redoes the call with the plain numbering
                    const int rel = qc - (SIDE_R ? (qr * kR) / kC : (qr * BR) / BC) + (rel_bc - 1);      // (rel_bc - 1 = blocks a window may lie LEFT of that one)
                    if ((unsigned)rel >= (1u << sh_br)) { rel_over = true; bc = 0u; } else bc = (unsigned)rel;
                }
                inside = (unsigned)((r - cs) - qr * (SIDE_R ? SIDE_R : BR)) | ((unsigned)((c - cs) - qc * (SIDE_C ? SIDE_C : BC)) << kWinShift);
            }
        }
        if (n_eregs > 0) {                                   // the region whose expected the snippet divides by (that of its first row)
            const int e = find_exp_region(eregs, n_eregs, r);
            er = (unsigned)(e < 0 ? n_eregs : e);
        }
        if (live && !ok) ++bad;
        // (counted per thread, summed per wave behind the loop: one atomic per wave and counter — a call of near-diagonal windows
        // would serialise on the counter)
        near_c += (live && (c - r < clear_gap || (c + W - 1) - r >= far_gap)) ? 1u : 0u;
        far_c += (live && ((band_w > 0 && (c + W - 1) - r >= band_w) || rel_over)) ? 1u : 0u;
        unsigned key_hi = 0u; bool counted = false;
        if (live) {
            if constexpr (sizeof(KeyT) == 4) {
                const unsigned key = (((((seg >> seg_shift) << sh_seg) | (er << sh_er) | (br << sh_br) | bc) << kbits) | kslot);
                keys[i] = (KeyT)key;
                key_hi = key >> hi_shift;
            } else {
                const unsigned long long key = ((((unsigned long long)(seg >> seg_shift) << sh_seg) | ((unsigned long long)er << sh_er) |
                                                 ((unsigned long long)br << sh_br) | bc) << kbits) | kslot;
                keys[i] = (KeyT)key;
                key_hi = (unsigned)(key >> hi_shift);
            }
            counted = hi_hist != nullptr;
            // the value that rides the sort is the window itself as the staged kernel wants it (the sort is stable, so windows
            // of a block keep the caller's order): no index to gather through afterwards
            vals[i] = (unsigned short)(inside | (slot << kWinSlotBit));
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
    if (bad) atomicAdd(&counters[0], bad);
    for (int off = 32; off > 0; off >>= 1) { near_c += __shfl_down(near_c, off); far_c += __shfl_down(far_c, off); }
    if (lane == 0 && near_c) atomicAdd(&counters[1], near_c);
    if (lane == 0 && far_c) atomicAdd(&counters[2], far_c);
    if (hi_hist) {
        __syncthreads();
        for (int k = threadIdx.x; k < hi_bins; k += blockDim.x) hi_hist[(size_t)blockIdx.x * hi_bins + k] = s_hh[k];
    }
}

// hand the key kernel's verdict to the host without stalling the stream: one thread copies the two counters into mapped
// page-locked host memory; the host waits on an event recorded right behind this kernel while the sort is already running
PUP_KERNEL void staged_publish_kernel(const unsigned* __restrict__ counters, int n, volatile unsigned* host_flags, unsigned ticket) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < n; ++i) host_flags[i] = counters[i];
        __threadfence_system();
        host_flags[n] = ticket;                          // behind the values: the host trusts them only under their ticket
        __threadfence_system();
    }
}

// the windows that start a block (key differs from the previous one), counted per span of kSpan windows —
// block_starts_kernel turns the counts into the ordered list
constexpr int kSpan = 4096;
template <typename KeyT>
__global__ __launch_bounds__(256) void count_heads_kernel(const KeyT* __restrict__ sorted_keys, long long n, int slot_bits,
                                                          unsigned* __restrict__ span_heads) {
    __shared__ unsigned red[4];
    const long long i0 = (long long)blockIdx.x * kSpan;                 // one workgroup per span
    unsigned cnt = 0;
#pragma unroll 4
    for (int t = threadIdx.x; t < kSpan; t += 256) {
        const long long i = i0 + t;
        if (i < n && (i == 0 || (sorted_keys[i] >> slot_bits) != (sorted_keys[i - 1] >> slot_bits))) ++cnt;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) span_heads[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ordered list of block starts: workgroup g owns windows [g*kSpan, (g+1)*kSpan); its output offset is the number of
// heads in the spans before it (a few thousand counters: summed by the workgroup itself, no separate scan pass).  The
// last workgroup also publishes the total (n_runs).
template <typename KeyT>
__global__ __launch_bounds__(256) void block_starts_kernel(const KeyT* __restrict__ sorted_keys, long long n, int slot_bits,
                                                           const unsigned* __restrict__ span_heads,
                                                           unsigned* __restrict__ starts, unsigned* __restrict__ n_runs) {
    __shared__ unsigned red[4];
    __shared__ unsigned wave_cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned part = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += blockDim.x) part += span_heads[k];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    unsigned base = red[0] + red[1] + red[2] + red[3];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) n_runs[0] = base + span_heads[blockIdx.x];
    if (span_heads[blockIdx.x] == 0) return;               // (uniform) nothing starts in this span
    const long long i0 = (long long)blockIdx.x * kSpan;
    for (int t = 0; t < kSpan; t += 256) {
        const long long i = i0 + t + threadIdx.x;
        const bool head = i < n && (i == 0 || (sorted_keys[i] >> slot_bits) != (sorted_keys[i - 1] >> slot_bits));
        const unsigned long long m = __ballot(head);
        __syncthreads();                                   // wave_cnt of the previous round has been read
        if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w];
        if (head) starts[base + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned)i;
        base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    }
}

// block table from the compacted block starts (grid-stride: the number of blocks is only known on the device): entry b =
// region origin, its windows, slot-0 windows, segment and expected region decoded from the key, the staging geometry of the
// region (see StagedBlock), the row hull of sparse blocks — and the first block of every workgroup's range: workgroup g of
// the G persistent ones takes an equal share of (windows + kBlockCost per block).
template <typename KeyT>
__global__ __launch_bounds__(256) void staged_table_kernel(const unsigned* __restrict__ starts, const unsigned* __restrict__ n_runs,
                                                           long long n, const KeyT* __restrict__ sorted_keys,
                                                           const unsigned short* __restrict__ win, const int* __restrict__ brow_base,
                                                           const IdxChrom* __restrict__ chroms, int n_chrom, int WR, int WC /* (sub-)window rows / columns */,
                                                           int RSR, int RSC, int NG /* sub-window groups in the key's segment field (K1w), else 1 */,
                                                           int block_cost /* staging one region, in windows' worth of time */,
                                                           int sh_br, int sh_er, int sh_seg, int seg_shift, int slot_bits, int n_eregs,
                                                           int er_in_key, const ExpRegion* __restrict__ eregs,
                                                           const unsigned long long* __restrict__ badbits,
                                                           StagedBlock* __restrict__ blocks, int* __restrict__ wg_first, int G,
                                                           const unsigned* __restrict__ block_keys /* nullable: key >> slot_bits of block b (pup_bin.hpp) */,
                                                           const unsigned short* __restrict__ sorted_low /* with block_keys and slot_bits > 0: low digits in block order */,
                                                           volatile unsigned* host_flags /* nullable: the block count goes there under `ticket` (was a launch of its own) */,
                                                           unsigned ticket, int rel_bc /* > 0: the key's column block is relative to the block row's diagonal, biased by rel_bc - 1 */) {
    const long long nr = (long long)n_runs[0];
    if (host_flags && blockIdx.x == 0 && threadIdx.x == 0) {
        // leave the block count where the NEXT call with this signature finds it without waiting
        host_flags[0] = n_runs[0];
        __threadfence_system();
        host_flags[1] = ticket;
        __threadfence_system();
    }
    const int BR = RSR - WR + 1, BC = RSC - WC + 1;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nr; b += (long long)gridDim.x * blockDim.x) {
        const unsigned s = starts[b];
        const long long e = (b + 1 < nr) ? (long long)starts[b + 1] : n;
        const unsigned long long key = block_keys ? (unsigned long long)block_keys[b] : (unsigned long long)sorted_keys[s] >> slot_bits;
        StagedBlock be;
        const int br = (int)((key >> sh_br) & ((1ull << (sh_er - sh_br)) - 1ull));
        int bc = (int)(key & ((1ull << sh_br) - 1ull));
        int lo = 0, hi = n_chrom;                              // last chromosome whose first block row is <= br
        while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (brow_base[m] <= br) lo = m; else hi = m; }
        const IdxChrom ch = chroms[lo];
        const int cs = ch.start;
        if (rel_bc) bc += ((br - brow_base[lo]) * BR) / BC - (rel_bc - 1);
        be.R = cs + (br - brow_base[lo]) * BR;
        be.C = cs + bc * BC;
        be.start = (int)s; be.count = (int)(e - (long long)s);
        {   // the windows of a block are in slot order (see StagedBlock::bnd): end of slot q = first window of a higher slot
            const unsigned smask = (1u << slot_bits) - 1u;
            auto slot_of = [&](long long m) -> unsigned {
                return slot_bits ? (block_keys ? (unsigned)sorted_low[m] : (unsigned)sorted_keys[m]) & smask : ((unsigned)win[m] >> kWinSlotBit) & 1u;
            };
            long long from = (long long)s;
            for (int q = 0; q < 7; ++q) {
                long long l2 = from, h2 = e;
                if (q < (slot_bits ? (1 << slot_bits) - 1 : 1))
                    while (l2 < h2) { const long long m = (l2 + h2) >> 1; if (slot_of(m) <= (unsigned)q) l2 = m + 1; else h2 = m; }
                else l2 = e;
                be.bnd[q] = (int)(l2 - (long long)s);
                from = l2;
            }
            be.count0 = be.bnd[0];
        }
        be.ereg = -1;
        if (n_eregs > 0) {
            // the expected region of the block's windows: part of the key, or — when no chromosome is split between regions —
            // the region its origin lies in
            const int er = er_in_key ? (int)((key >> sh_er) & ((1ull << (sh_seg - sh_er)) - 1ull))
                                     : find_exp_region(eregs, n_eregs, be.R);
            be.ereg = (er >= 0 && er < n_eregs) ? er : -1;
        }
        be.seg = (int)((key >> sh_seg) / (unsigned long long)NG) << seg_shift;
        be.ch_end = ch.end; be.nblk = ch.nblk;
        auto bits64 = [&](int bin) {                        // masked-bin bits of bins [bin, bin + 64)
            const unsigned long long* w = badbits + (bin >> 6);
            const int sh = bin & 63;
            unsigned long long v = w[0] >> sh;
            if (sh) v |= w[1] << (64 - sh);
            return v;
        };
        for (int h = 0; h < 2; ++h) {
            const int c_h = be.C + 64 * h;
            unsigned long long colok = 0ull;
            unsigned line = be.line0[0], wssh = be.ws_sh[0];
            if (h == 0 || (64 * h < RSC && c_h < ch.end)) {
                // staging geometry: the half's 64 columns start at bit `sh` of word `ws` of index line `bi` of every region row
                const int rel = c_h - cs;
                const int bi = rel / kIdxCols, o = rel - bi * kIdxCols;
                wssh = (unsigned)(o >> 6) | ((unsigned)(o & 63) << 8);
                line = (unsigned)(ch.blk_base + (long long)(be.R - cs) * ch.nblk + bi);
                colok = ~bits64(c_h);
                const int over = c_h + 64 - ch.end;         // columns at / past the chromosome's end are in no eligible window
                if (over > 0) colok &= over >= 64 ? 0ull : (~0ull >> over);
            }
            be.line0[h] = line; be.ws_sh[h] = wssh;
            be.colok[2 * h] = (unsigned)colok; be.colok[2 * h + 1] = (unsigned)(colok >> 32);
            const unsigned long long rb = (64 * h < RSR && be.R + 64 * h < ch.end) ? bits64(be.R + 64 * h) : 0ull;
            be.rowbad[2 * h] = (unsigned)rb; be.rowbad[2 * h + 1] = (unsigned)(rb >> 32);
        }
        be.row_lo = 0; be.row_hi = RSR;
        if (be.count <= 32) {                               // sparse block: stage only the rows its windows touch
            int mn = RSR, mx = 0;
            for (long long m = (long long)s; m < e; ++m) {
                const int dr = (int)(win[m] & ((1u << kWinShift) - 1u));
                mn = dr < mn ? dr : mn; mx = dr > mx ? dr : mx;
            }
            be.row_lo = mn; be.row_hi = mx + WR < RSR ? mx + WR : RSR;
        }
        be.pad[0] = (int)((key >> sh_seg) % (unsigned long long)NG); be.pad[1] = 0;      // K1w: the sub-window group of the block's items
        blocks[b] = be;
        // ranges of the persistent workgroups: equal shares of the call's COST, a block costing its windows plus kBlockCost
        // window-equivalents for its staging (a range of many sparse blocks would otherwise take far longer than one of a
        // few dense ones: staging a region is an HBM round trip plus ~2000 clocks of LDS stores, a window ~15 clocks)
        const unsigned long long total = (unsigned long long)n + (unsigned long long)block_cost * (unsigned long long)nr;
        const int g_cur = (int)((((unsigned long long)s + (unsigned long long)block_cost * (unsigned long long)b) * (unsigned long long)G) / total);
        const int g_prev = b == 0 ? -1 : (int)((((unsigned long long)starts[b - 1] + (unsigned long long)block_cost * (unsigned long long)(b - 1)) *
                                                (unsigned long long)G) / total);
        for (int g = g_prev + 1; g <= g_cur; ++g) wg_first[g] = (int)b;
        if (b + 1 == nr) for (int g = g_cur + 1; g <= G; ++g) wg_first[g] = (int)nr;
    }
    if (nr == 0 && blockIdx.x == 0) for (int g = threadIdx.x; g <= G; g += blockDim.x) wg_first[g] = 0;
}

// fixed-order reduction of the staged kernel's partial records into the running accumulators: tile t gathers the records its
// owner mark names (see StagedArgs::rec_owner) — flip 0 of every workgroup in order, then flip 1.
// Same shape as reduce_partials_kernel: 64 record elements x kRedParts interleaved partial sums in fixed order.
PUP_KERNEL __launch_bounds__(64 * kRedParts) void reduce_staged_kernel(
        const double* __restrict__ in_f64, const unsigned* __restrict__ in_num, const unsigned short* __restrict__ owner,
        int G, int T, int ACC, int PH, int Lf, int Li, double* out_f64, long long* out_num,
        long long* out_n /* nullable: windows per tile, added from the (tile, flip) run ends (was a launch of its own) */, const long long* __restrict__ seg_end) {
    __shared__ double    sf[kRedParts][64];
    __shared__ long long si[kRedParts][64];
    const int t = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y;
    if (out_n && blockIdx.x == 0 && cx == 0 && py == 0) out_n[t] += seg_end[2 * t + 1] - (t ? seg_end[2 * t - 1] : 0);
    const int idx = blockIdx.x * 64 + cx;
    // the accumulator slot tile t is piled up in (inverse of staged_tile)
    const int slot = ACC == 1 ? 0 : (ACC == 2 ? t / PH : (t / PH) * (ACC / 2) + (t % PH) % (ACC / 2));
    const size_t base = (size_t)slot * (size_t)(2 * T + G) + (size_t)t * 2;
    const long long e = 2LL * G;                          // virtual records c = flip * G + workgroup
    double accf = 0.0; long long acci = 0;
    // sixteen records per round: their marks, then their values, are independent loads (one at a time, a thread waited out a
    // memory round trip per record; 256 workgroups x 2 flips = 512 records are two rounds of the 16 parts); the additions keep
    // the order of the plain loop
    constexpr int kUn = 16;
    auto rec_of = [&](long long c) -> size_t { const int fl = c >= G ? 1 : 0; return base + (size_t)fl + (size_t)(c - (long long)fl * G); };
    auto mine = [&](long long c) -> bool { return c < e && owner[rec_of(c)] == (unsigned short)(t * 2 + (c >= G ? 1 : 0) + 1); };
    if (idx < Lf) {
        for (long long c0 = py; c0 < e; c0 += (long long)kUn * kRedParts) {
            bool ok[kUn]; double v[kUn];
#pragma unroll
            for (int u = 0; u < kUn; ++u) ok[u] = mine(c0 + (long long)u * kRedParts);
#pragma unroll
            for (int u = 0; u < kUn; ++u) { const long long c = c0 + (long long)u * kRedParts; v[u] = in_f64[rec_of(c < e ? c : 0) * Lf + idx]; }   // (not waiting for the marks: a record nobody owns holds old numbers, never added)
#pragma unroll
            for (int u = 0; u < kUn; ++u) if (ok[u]) accf += v[u];
        }
    } else if (idx < Lf + Li) {
        const int k = idx - Lf;
        for (long long c0 = py; c0 < e; c0 += (long long)kUn * kRedParts) {
            bool ok[kUn]; unsigned v[kUn];
#pragma unroll
            for (int u = 0; u < kUn; ++u) ok[u] = mine(c0 + (long long)u * kRedParts);
#pragma unroll
            for (int u = 0; u < kUn; ++u) { const long long c = c0 + (long long)u * kRedParts; v[u] = in_num[rec_of(c < e ? c : 0) * Li + k]; }
#pragma unroll
            for (int u = 0; u < kUn; ++u) if (ok[u]) acci += (long long)v[u];
        }
    }
    sf[py][cx] = accf; si[py][cx] = acci;
    __syncthreads();
    if (py != 0 || idx >= Lf + Li) return;
    if (idx < Lf) {
        double tt = 0.0;
#pragma unroll
        for (int y = 0; y < kRedParts; ++y) tt += sf[y][cx];
        out_f64[(size_t)t * Lf + idx] += tt;
    } else {
        long long tt = 0;
#pragma unroll
        for (int y = 0; y < kRedParts; ++y) tt += si[y][cx];
        out_num[(size_t)t * Li + (idx - Lf)] += tt;
    }
}

}  // namespace pup
