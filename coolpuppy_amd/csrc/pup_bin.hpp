// pup_bin.hpp — the block-order prepass of the staged kernels (K1q, K1w), hand-written for gfx950: no library sort on the hot path.
//
// Round 3 sorted the (block key, window) pairs with rocPRIM's radix sort: histogram + two onesweep passes (0.18 of the 0.31 ms
// prepass of the headline step), then two more kernels over the sorted keys to find the block starts.  What the staged kernels
// need is less than a sort: the windows GROUPED by block key, blocks in key order, the order inside a block deterministic
// (the caller's order: sums then do not depend on timing).  This file does exactly that for keys of up to 23 bits
// (hg38 at 10 kb: 20), split into a high digit of DH and a low digit of DL bits (both <= 11):
//   0. the key kernels (staged_key_kernel / wide_key_kernel) also count the high digits of their workgroup's 8192 windows (LDS
//      histogram, one add per distinct digit of a wave) and write the counts as the tile's row of tilehist;
//   1. bin_chunksum_kernel + bin_scan_kernel: the per-tile digit counts summed per chunk of 64 tiles, scanned over the chunks and
//      over the digits = where every bucket starts and where a tile's share of it goes;
//   2. bin_partition_kernel: ONE pass that moves every window to its bucket, stably.  Tiles of 8192 windows (= the key kernel's
//      workgroups, whose digit counts are the tile's); inside a tile the stable rank is ballots per wave (who else in my wave
//      has my digit?) plus per-wave counters in LDS.  What travels is ONE dword per window: low digit << 16 | the 16-bit window
//      value the staged kernel consumes;
//   3. bin_bucket_kernel: every bucket (a few thousand windows: it lives in L2) is ordered by the low digit by ONE small workgroup
//      (four waves: eight such workgroups per CU hide the latency) — counting sort in two sweeps over the bucket: per-wave counts
//      per digit, then every wave places its contiguous share; reads are sequential, writes land in the bucket's own span.  The same kernel emits the bucket's blocks (start, key): a block
//      IS a non-empty run of low digits, no second look at sorted keys;
//   4. bin_compact_kernel: the buckets' block lists packed into one (a scan over <= 2048 bucket counts).
// Keys wider than 23 bits (hundreds of tiles x expected regions) keep the library sort (staged_run / wide_run decide).
#pragma once
#include "pup_kernels.hpp"

namespace pup {

constexpr int kBinMaxDigit = 11;                      // bits of the HIGH digit: 2048 counters per wave of the partition pass
constexpr int kBinMaxLow = 12;                        // bits of the LOW digit: (waves + 2) x 4096 counters of a bucket's workgroup (96 KB with four waves)
constexpr int kBinTile = 8192;                        // windows per tile of the partition pass: 16 waves x 8 rounds x 64 lanes
constexpr int kBinWaves = 16;

// who else in my wave holds my digit: 64-bit lane mask, by ballots over the digit's bits (lanes that are not `live` match nobody)
__device__ __forceinline__ unsigned long long match_digit(unsigned d, int bits, bool live) {
    unsigned long long m = __ballot(live);
    for (int b = 0; b < bits; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return live ? m : 0ull;
}

// the same for digits that come in RUNS (the partition pass: consecutive windows of the caller's stream mostly share their block
// row): one ballot per distinct digit of the wave instead of one per bit; after kMatchTries distinct digits the bitwise form takes over
constexpr int kMatchTries = 6;
__device__ __forceinline__ unsigned long long match_runs(unsigned d, int bits, bool live) {
    unsigned long long mine = 0ull, todo = __ballot(live);
    for (int t = 0; t < kMatchTries && todo; ++t) {
        const int l = __ffsll((long long)todo) - 1;
        const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)d, l);
        const unsigned long long m = __ballot(live && d == d0);
        if (live && d == d0) mine = m;
        todo &= ~m;
    }
    if (todo) return match_digit(d, bits, live);             // (uniform) many distinct digits in this round
    return mine;
}

// ---- where a tile's windows of a digit go: counts -> bases, without look-back -------------------------------------------------
// tilehist[tile][digit] (written by the key kernel, dense rows) -> chunk sums over kBinChunk tiles -> exclusive over the chunks
// and over the digits.  A tile then needs base[d] + chunkexcl[chunk][d] + the rows of its own chunk before it — for the handful
// of digits it holds.  (A decoupled look-back over all 2^DH digits — the onesweep scheme, tried first — spent 33 of the
// partition pass's 77 us walking descriptors: with 512 tiles in flight a tile's predecessors are mostly unfinished.)
constexpr int kBinChunk = 64;
// (tilehist is rewritten in place as the EXCLUSIVE prefix over the tiles of the chunk: the partition pass then needs one load per
// digit a tile holds.  At first it summed the rows of its chunk before it itself: up to 63 loads one after the other in a handful
// of threads — half of that kernel's 65 us)
PUP_KERNEL __launch_bounds__(1024) void bin_chunksum_kernel(unsigned* __restrict__ tilehist, long long ntiles, int nd, unsigned* __restrict__ chunksum) {
    const int c = blockIdx.x;
    const int d = blockIdx.y * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    const long long t0 = (long long)c * kBinChunk, t1 = t0 + kBinChunk < ntiles ? t0 + kBinChunk : ntiles;
    unsigned s = 0;
    for (long long tb = t0; tb < t1; tb += 16) {
        unsigned v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = tb + k < t1 ? tilehist[(size_t)(tb + k) * nd + d] : 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) { if (tb + k < t1) tilehist[(size_t)(tb + k) * nd + d] = s; s += v[k]; }
    }
    chunksum[(size_t)c * nd + d] = s;
}
// one workgroup: per digit the exclusive prefix over the chunks (in place) and the digit's total; then the digits' bases
PUP_KERNEL __launch_bounds__(1024) void bin_scan_kernel(unsigned* __restrict__ chunksum, int nchunks, int nd, unsigned* __restrict__ base) {
    __shared__ unsigned part[1024];
    const int t = threadIdx.x;
    const int per = (nd + 1023) / 1024;                  // digits per thread (1 or 2), blocked: thread t owns [t per, (t + 1) per)
    unsigned tot[2] = {0u, 0u};
    for (int k = 0; k < per; ++k) {
        const int d = t * per + k;
        if (d >= nd) break;
        unsigned run = 0;
        for (int c = 0; c < nchunks; ++c) { const unsigned v = chunksum[(size_t)c * nd + d]; chunksum[(size_t)c * nd + d] = run; run += v; }
        tot[k] = run;
    }
    const unsigned s = tot[0] + tot[1];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {            // Hillis-Steele inclusive scan of the 1024 partial sums
        const unsigned v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned run = part[t] - s;
    for (int k = 0; k < per; ++k) { const int d = t * per + k; if (d < nd) { base[d] = run; run += tot[k]; } }
    if (t == 1023) base[nd] = part[1023];
}

// ---- pass 1: stable partition by the high digit --------------------------------------------------------------------------
PUP_KERNEL __launch_bounds__(kWave * kBinWaves) void bin_partition_kernel(
        const unsigned* __restrict__ keys, const unsigned short* __restrict__ vals, long long n, int DL, int DH,
        const unsigned* __restrict__ base, const unsigned* __restrict__ chunkexcl, const unsigned* __restrict__ tilehist,
        unsigned* __restrict__ out) {
    extern __shared__ unsigned short whist[];              // [kBinWaves][2^DH] per-wave digit counts, then exclusive over the waves
    __shared__ unsigned gbase[1 << kBinMaxDigit];          // where the tile's windows of a digit start in the output
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nd = 1 << DH;
    for (int k = tid; k < kBinWaves * nd; k += blockDim.x) whist[k] = 0;
    const unsigned tile = blockIdx.x;
    const long long t0 = (long long)tile * kBinTile;
    constexpr int R = kBinTile / (kWave * kBinWaves);      // rounds per wave: a wave owns R * 64 consecutive windows of the tile
    unsigned item[R]; unsigned short rnk[R], dig[R];
    unsigned short* mine = whist + wave * nd;
    // all loads of the tile first (independent: one memory latency), the ranking — a chain through the wave's LDS counters — after
    unsigned kreg[R]; unsigned short vreg[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = t0 + (long long)(wave * R + r) * kWave + lane;
        kreg[r] = i < n ? keys[i] : 0u;
        vreg[r] = i < n ? vals[i] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long i = t0 + (long long)(wave * R + r) * kWave + lane;
        const bool live = i < n;
        const unsigned key = kreg[r];
        const unsigned d = key >> DL;
        item[r] = live ? (((key & ((1u << DL) - 1u)) << 16) | (unsigned)vreg[r]) : 0u;
        dig[r] = live ? (unsigned short)d : (unsigned short)0xffff;
        const unsigned long long m = match_runs(d, DH, live);
        const int lead = m ? __ffsll((long long)m) - 1 : 0;
        unsigned prev = 0;
        if (live && lane == lead) { prev = mine[d]; mine[d] = (unsigned short)(prev + (unsigned)__popcll(m)); }
        prev = __shfl(prev, lead);
        rnk[r] = (unsigned short)(prev + (unsigned)__popcll(m & ((1ull << lane) - 1ull)));
    }
    __syncthreads();
    // per digit: exclusive over the waves; for the digits the tile holds, where they start: the digit's base + the chunks before
    // + the tiles of this chunk before
    for (int d = tid; d < nd; d += blockDim.x) {
        unsigned run = 0;
        for (int w = 0; w < kBinWaves; ++w) { const unsigned t = whist[w * nd + d]; whist[w * nd + d] = (unsigned short)run; run += t; }
        if (run == 0u) continue;
        gbase[d] = base[d] + chunkexcl[(size_t)(tile / kBinChunk) * nd + d] + tilehist[(size_t)tile * nd + d];    // (tilehist: exclusive within the chunk by now)
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (dig[r] == 0xffff) continue;
        const unsigned d = dig[r];
        out[gbase[d] + (unsigned)mine[d] + (unsigned)rnk[r]] = item[r];
    }
}

// exclusive prefixes of two per-thread values over a workgroup (scratch: [2][kBinWaves] in LDS)
__device__ __forceinline__ void block_excl_scan2(unsigned a, unsigned b, unsigned* scratch, unsigned& ea, unsigned& eb, unsigned& ta, unsigned& tb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (int)(blockDim.x >> 6);
    unsigned ia = a, ib = b;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned va = __shfl_up(ia, off), vb = __shfl_up(ib, off);
        if (lane >= off) { ia += va; ib += vb; }
    }
    __syncthreads();                                        // scratch is free
    if (lane == 63) { scratch[wave] = ia; scratch[kBinWaves + wave] = ib; }
    __syncthreads();
    unsigned pa = 0, pb = 0; ta = 0; tb = 0;
    for (int w = 0; w < nwaves; ++w) { const unsigned xa = scratch[w], xb = scratch[kBinWaves + w]; if (w < wave) { pa += xa; pb += xb; } ta += xa; tb += xb; }
    ea = pa + ia - a; eb = pb + ib - b;
}

// ---- pass 2: every bucket ordered by the low digit, blocks emitted ---------------------------------------------------------
// in[i] = low digit << 16 | value, bucket h = in[base[h] .. base[h+1]).  out_val: the values in final order.  Blocks: a block is
// a maximal run of low digits that agree above `slot_bits` (sets of tile pairs keep the slot in the key's lowest digit); the
// bucket writes its blocks' (start, key >> slot_bits) to blk_start / blk_key AT ITS OWN SPAN (a bucket of c windows has at most
// c blocks) and their number to blk_count[h].  out_low (nullable): the sorted low digits (the table kernel reads the slot there).
// A workgroup of kBucketWaves = 8 waves per bucket: every wave owns a contiguous share, counts its digits (sweep 1: plain LDS adds,
// integer counts are order-free), the waves' counts are scanned into starts / blocks / dense numbers of the digits that occur, and
// every wave places its windows in order (sweep 2: who else in my round has my digit? — ballots over the bits of the DENSE number:
// a few dozen of the 2^DL digits occur in a bucket).  Eight rounds of loads are in flight at a time.  Measured on the headline
// workload (1.1e7 windows, 702 buckets): 8 waves 51 us, 4 waves 60 us, 16 waves 65 us, one wave per bucket (2048 buckets) 93 us — the
// buckets are uneven and the biggest one's workgroup is the kernel's tail.  (Measured and dropped, round 4: the next eight rounds' loads
// issued before this eight's LDS work; the leaders' read-modify-write as LDS atomics with return, batched per eight rounds: no change.)
constexpr int kBucketWaves = 8;                      // most waves of a bucket's workgroup (the launch picks: bin_run)
PUP_KERNEL __launch_bounds__(kWave * kBucketWaves) void bin_bucket_kernel(
        const unsigned* __restrict__ in, const unsigned* __restrict__ base, int DL, int DH, int slot_bits,
        unsigned short* __restrict__ out_val, unsigned short* __restrict__ out_low,
        unsigned* __restrict__ blk_start, unsigned* __restrict__ blk_key, unsigned* __restrict__ blk_count) {
    extern __shared__ unsigned wcnt[];                     // [kBucketWaves][2^DL] per-wave digit counts, then exclusive over the waves | tot[2^DL] | dstart[2^DL]
    __shared__ unsigned scratch[2 * kBinWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = 1 << DL;
    const int NWV = (int)(blockDim.x >> 6);                // waves of the workgroup (<= kBucketWaves)
    unsigned* const tot = wcnt + NWV * nl;        // windows per low digit, later the digit's dense number
    unsigned* const dstart = tot + nl;                     // where the digit's run starts (bucket-relative)
    const int per = (nl + (int)blockDim.x - 1) / (int)blockDim.x;          // digits per thread, blocked: thread t owns [t per, (t + 1) per)
    unsigned* mine = wcnt + wave * nl;
    constexpr int U = 8;
    for (int h = blockIdx.x; h < (1 << DH); h += gridDim.x) {
        const unsigned b0 = base[h], b1 = base[h + 1];
        if (b0 == b1) { if (tid == 0) blk_count[h] = 0u; continue; }        // (uniform)
        const unsigned cnt = b1 - b0;
        const unsigned share = ((cnt + (unsigned)NWV - 1u) / (unsigned)NWV + 63u) & ~63u;        // whole rounds of 64
        const unsigned w0 = (unsigned)wave * share < cnt ? (unsigned)wave * share : cnt;
        const unsigned w1 = w0 + share < cnt ? w0 + share : cnt;
        __syncthreads();
        for (int k = tid; k < NWV * nl; k += blockDim.x) wcnt[k] = 0u;
        __syncthreads();
        for (unsigned i0 = w0; i0 < w1; i0 += U * 64) {       // sweep 1
            unsigned it[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const unsigned i = i0 + (unsigned)u * 64 + lane; it[u] = i < w1 ? in[b0 + i] : 0xffffffffu; }
#pragma unroll
            for (int u = 0; u < U; ++u) if (it[u] != 0xffffffffu) atomicAdd(&mine[it[u] >> 16], 1u);
        }
        __syncthreads();
        for (int d = tid; d < nl; d += blockDim.x) {          // exclusive over the waves; the bucket's count of the digit
            unsigned run = 0;
            for (int w = 0; w < NWV; ++w) { const unsigned t = wcnt[w * nl + d]; wcnt[w * nl + d] = run; run += t; }
            tot[d] = run;
        }
        __syncthreads();
        {   // digit starts (exclusive prefix of the counts), the bucket's block list, dense numbers of the digits that occur
            unsigned csum = 0, heads = 0, nz = 0, hmask = 0u;
            for (int k = 0; k < per; ++k) {
                const int d = tid * per + k;
                if (d >= nl || !tot[d]) continue;
                csum += tot[d]; ++nz;
                bool head = true;                           // no non-empty digit before it shares its bits above slot_bits
                for (int e = (d >> slot_bits) << slot_bits; e < d; ++e) if (tot[e]) { head = false; break; }
                if (head) { ++heads; hmask |= 1u << k; }
            }
            unsigned run, at, tc, th, id, dz, nd_, dz2;
            block_excl_scan2(csum, heads, scratch, run, at, tc, th);
            block_excl_scan2(nz, 0u, scratch, id, dz, nd_, dz2);
            __syncthreads();                                // every thread has read the neighbours' counts it needs
            for (int k = 0; k < per; ++k) {
                const int d = tid * per + k;
                if (d >= nl) break;
                const unsigned c = tot[d];
                if (c) {
                    if ((hmask >> k) & 1u) { blk_start[b0 + at] = b0 + run; blk_key[b0 + at] = ((unsigned)h << (DL - slot_bits)) | ((unsigned)d >> slot_bits); ++at; }
                    tot[d] = id++;
                }
                dstart[d] = run;
                run += c;
            }
            if (tid == 0) { blk_count[h] = th; scratch[2 * kBinWaves - 1] = nd_; }
        }
        __syncthreads();
        int idbits = 0;
        { const unsigned ndist = scratch[2 * kBinWaves - 1]; while ((1u << idbits) < ndist) ++idbits; }
        for (unsigned i0 = w0; i0 < w1; i0 += U * 64) {       // sweep 2
            unsigned itv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const unsigned i = i0 + (unsigned)u * 64 + lane; itv[u] = i < w1 ? in[b0 + i] : 0xffffffffu; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + (unsigned)u * 64 >= w1) break;      // (uniform)
                const unsigned it = itv[u];
                const bool live = it != 0xffffffffu;
                const unsigned d = live ? it >> 16 : 0u;
                const unsigned long long m = match_digit(live ? tot[d] : 0u, idbits, live);
                const int lead = m ? __ffsll((long long)m) - 1 : 0;
                unsigned prev = 0;
                if (live && lane == lead) { prev = mine[d]; mine[d] = prev + (unsigned)__popcll(m); }
                prev = __shfl(prev, lead);
                if (live) {
                    const unsigned pos = b0 + dstart[d] + prev + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                    out_val[pos] = (unsigned short)(it & 0xffffu);
                    if (out_low) out_low[pos] = (unsigned short)d;
                }
            }
        }
    }
}

// ---- the buckets' block lists packed into one: starts[b], bkeys[b], b in key order; n_runs[0] = their number -------------------
PUP_KERNEL __launch_bounds__(256) void bin_compact_kernel(const unsigned* __restrict__ base, const unsigned* __restrict__ blk_count, int nd,
                                                          const unsigned* __restrict__ blk_start, const unsigned* __restrict__ blk_key,
                                                          unsigned* __restrict__ starts, unsigned* __restrict__ bkeys, unsigned* __restrict__ n_runs) {
    __shared__ unsigned red[4];
    const int h = blockIdx.x;
    unsigned part = 0;
    for (int k = threadIdx.x; k < h; k += blockDim.x) part += blk_count[k];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const unsigned off = red[0] + red[1] + red[2] + red[3];
    const unsigned nb = blk_count[h], b0 = base[h];
    if (h == nd - 1 && threadIdx.x == 0) n_runs[0] = off + nb;
    for (unsigned j = threadIdx.x; j < nb; j += blockDim.x) { starts[off + j] = blk_start[b0 + j]; bkeys[off + j] = blk_key[b0 + j]; }
}

}  // namespace pup
