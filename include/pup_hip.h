/*
 * pup_hip.h — C ABI of the MI355X (gfx950) pile-up engine, libpup_hip.so.
 *
 * This is the drop-in boundary for ONE hot path of open2c/coolpuppy: the
 * per-region "slice a (2*pad+1)^2 window out of the chromosome CSR -> mask ->
 * normalise -> accumulate sum/num" loop.  The reference has no FFI of its own
 * (it is pure Python); each entry point below names the reference code it
 * replaces (paths are relative to the coolpuppy source tree):
 *
 *   pup_load_pixels / pup_load_pixels_stream / pup_load_bins
 *                                     <- PileUpper.get_data            coolpuppy/coolpup.py:1024-1057
 *                                        + weight / coverage columns   coolpuppy/coolpup.py:1081-1098
 *   pup_build_index                   <- (no counterpart: device-side search structure over the same table;
 *                                        chromosome partition = cooler's indexes/chrom_offset, the extents the
 *                                        reference takes from clr.extent / clr.offset,     coolpuppy/coolpup.py:922-925)
 *   pup_set_expected[_table]          <- expected_selections / get_expected_trans
 *                                                                      coolpuppy/coolpup.py:907-916, 999-1005
 *   pup_reset                         <- make_outmap / empty_pup       coolpuppy/coolpup.py:986-997, 1007-1022
 *   pup_accumulate                    <- _stream_snips + accumulate_stream + _add_snip
 *                                                                      coolpuppy/coolpup.py:1059-1191, 1236-1283
 *                                                                      coolpuppy/lib/puputils.py:12-41
 *                                        (flip bit: flip_snip_func     coolpuppy/coolpup.py:128-147)
 *   pup_load_pixel_values             <- the same table when pixels/count is a float column   coolpuppy/coolpup.py:1053-1057
 *   pup_coverage                      <- cooltools.api.coverage.coverage(clr, ignore_diags=..., store=True), called by
 *                                        PileUpper.__init__ when cov_*_raw is missing     coolpuppy/coolpup.py:955-963
 *   pup_accumulate_rescaled           <- the same loop with _rescale_snip               coolpuppy/coolpup.py:1159-1162, 1193-1234
 *   pup_export / pup_import / pup_allreduce <- reduce(sum_pups) over worker processes  coolpuppy/coolpup.py:1495-1531
 *   pup_allgather_tiles / pup_pack_tiles / pup_unpack_tiles <- the same merge for by-window pile-ups, whose per-feature
 *                                        pile-ups exist on one worker each        coolpuppy/coolpup.py:1696-1755
 *   pup_stripes                       <- the store_stripes branch of _stream_snips   coolpuppy/coolpup.py:1164-1182
 *   pup_extract                       <- _stream_snips as a producer of per-snippet windows for the Python callbacks
 *                                        (postprocess_func / extra_sum_funcs)          coolpuppy/coolpup.py:1104-1162, 1261-1262
 *   pup_fetch / pup_export / pup_import
 *                                     <- sum_pups cross-region merge   coolpuppy/lib/puputils.py:88-113
 *                                        and the reduce at             coolpuppy/coolpup.py:1511-1531
 *   pup_host_lut_i32 / _count_le      <- group -> tile numbers and distance bands of a grouped pile-up's windows
 *                                                                      coolpuppy/coolpup.py:28-51, 1757-1919
 *   pup_host_normalise_tiles          <- the ratio to the control and the inf -> NaN step of pileupsWithControl
 *                                                                      coolpuppy/coolpup.py:1533-1545
 *   pup_host_mt_randint / _plan       <- np.random.randint / np.random.choice draws of CoordCreator._control_regions
 *                                                                      coolpuppy/coolpup.py:420-436
 *   pup_host_windows                  <- CoordCreator._control_regions (shifted control copies) + the bounds test of
 *   pup_host_control_windows             _stream_snips, as one host pass  coolpuppy/coolpup.py:387-453, 1105-1114
 *   pup_host_factorize_ptr            <- the same sort's chromosome codes (object columns factorised by identity)
 *   pup_host_argsort                  <- the same sort's order (one packed key per row)
 *   pup_host_pair_region_counts       <- the rows per view region of the same table, before the sort (the draws' sizes)
 *   pup_host_sort_pairs               <- CoordCreator.process for BEDPE features: centres, mindist / maxdist filter and the sort
 *                                        by (chrom1, chrom2, start1, start2)          coolpuppy/coolpup.py:296-321, 489-527
 *   pup_host_take_rows                <- the sort of the feature frame in CoordCreator._binnify  coolpuppy/coolpup.py:489-527
 *   pup_host_group_tiles(_runs)       <- the per-group dicts of accumulate_stream as a grouping of windows by tile
 *                                                                      coolpuppy/coolpup.py:1263-1283
 *   pup_host_alloc / pup_host_free    <- (no counterpart: page-locked staging for asynchronous host-to-device copies)
 *   pup_rccl_path                     <- (no counterpart: which librccl the communicator of pup_allreduce must come from)
 *   pup_debug_timing                  <- (no counterpart: phase clocks of the staged kernel, development aid)
 *   pup_last_kernel / pup_last_prepass <- (no counterpart: which kernel family served the last call and how its windows were
 *                                        put in block order, for tests / benchmarks)
 *
 * Conventions
 *   - plain C: pointers + sizes only; no C++/torch types cross this line.
 *   - every function returns 0 on success or a negative PUP_E* code; the text
 *     of the last failure is available from pup_last_error().  Nothing throws.
 *   - all host buffers are caller-owned; the library copies what it needs (snippet arrays from pup_host_alloc memory are
 *     copied asynchronously: leave them untouched until the next pup_sync / pup_fetch).
 *   - one context per GPU; calls on one context must be serialised by the
 *     caller; different contexts are independent.
 *   - "bin" always means a GLOBAL bin id of the cooler bin table (chromosome
 *     offset already added).  "tile" = one (kind, group) accumulator: a WxW
 *     sum (f64), a WxW num (i64), cov_start/cov_end (f64[W]) and n (i64).
 */
#ifndef PUP_HIP_H
#define PUP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pup_ctx pup_ctx;

/* error codes */
#define PUP_OK          0
#define PUP_EINVAL     -1   /* bad argument */
#define PUP_ENOMEM     -2   /* host or device allocation failed */
#define PUP_EHIP       -3   /* a HIP runtime call failed */
#define PUP_ESTATE     -4   /* call order violated (e.g. accumulate before load) */
#define PUP_ERANGE     -5   /* a snippet window leaves the bin table (caught on device) */
#define PUP_ENOTSUP    -6   /* valid request the engine cannot serve (e.g. a rescaled output tile too large for LDS) */

/* mode bits for pup_accumulate */
#define PUP_MODE_OOE        0x01u  /* divide each value by expected before summing (ooe=True) */
#define PUP_MODE_EXPECTED   0x02u  /* accumulate the EXPECTED window itself, unmasked (the
                                      "control"-kind snippet of expected & !ooe, coolpup.py:1135-1139);
                                      the pixel table is not read */
#define PUP_MODE_COV        0x04u  /* accumulate cov_start / cov_end (coverage_norm) */
#define PUP_MODE_TRANSPOSE  0x08u  /* (r0,c0) were swapped by the caller so that r0's region precedes
                                      c0's in the upper-triangular table; cells are stored transposed */
#define PUP_MODE_DEVPTR     0x10u  /* r0 / c0 are DEVICE pointers (already resident in HBM) */
#define PUP_MODE_LOCAL      0x20u  /* rescaled pile-ups only: symmetrise each (square, on-diagonal) window by nanmean
                                      with its transpose before zooming (local=True, coolpup.py:1215-1220) */

/* lifetime ---------------------------------------------------------------------------------------- */
int  pup_create(int device_id, pup_ctx** out);
void pup_destroy(pup_ctx* ctx);
/* last error text of this context (or of pup_create when ctx == NULL); never NULL */
const char* pup_last_error(const pup_ctx* ctx);
/* library/ABI version: major*10000 + minor*100 + patch */
int  pup_version(void);
/* number of HIP devices visible, or a negative error */
int  pup_device_count(void);

/* inputs ------------------------------------------------------------------------------------------ */
/*
 * Upper-triangular pixel table exactly as a .cool stores it: bin1_offset = indexes/bin1_offset
 * (CSR row pointer, int64[nbins+1]), bin2_id = pixels/bin2_id (int32 when bin2_bytes == 4, int64 when 8),
 * count = pixels/count (int32).  Rows sorted, columns sorted within a row, bin2 >= bin1.
 * Stored on device as int64 indptr + interleaved {int32 col, int32 count} pairs (8 B per pixel).
 */
int pup_load_pixels(pup_ctx* ctx, const int64_t* bin1_offset, const void* bin2_id, int bin2_bytes,
                    const int32_t* count, int64_t nbins, int64_t nnz);

/*
 * pup_load_pixel_values: the pixel VALUES as float64, for coolers whose pixels/count is a float column (cooler allows it —
 * merged, scaled or simulated maps — and the reference multiplies whatever matrix(balance=...) hands it,
 * coolpuppy/coolpup.py:1053-1057).  Call after pup_load_pixels (whose integer counts are then placeholders; pass zeros) and
 * before pup_load_bins.  The table is then piled up by the kernels that read the balanced value table (per-window register /
 * banded tiles, the sparse trans kernel, rescaled windows, stripes, per-snippet windows): same results as for counts.  The
 * workgroup-staged kernels (integer count tables, dense band) do not run on it; pup_coverage sums the values in float64
 * (round 6; counts: exact integer sums).  The next pup_load_pixels / pup_load_pixels_stream returns the context to integer
 * counts.
 */
int pup_load_pixel_values(pup_ctx* ctx, const double* value, int64_t nnz);
/*
 * The same table STREAMED in: the caller hands over a reader instead of whole arrays.  The library owns two page-locked slabs and
 * calls fill(user, first, m, bin2_id, count) to have pixels [first, first + m) written into one of them (bin2_id: m values of
 * bin2_bytes bytes; count: m x int32; return 0, anything else aborts the load with PUP_EINVAL) while the previous slab is on
 * its way: hipMemcpyAsync from page-locked memory on a copy stream, interleaved on the device by the same kernel as
 * pup_load_pixels.  fill is called from the calling thread, in order, with m <= slab_pixels (0 = library default, 32 Mi).
 * This is the reader side of "cooler HDF5 I/O stays on the host with pinned hipMemcpyAsync overlap": cool_io.py fills the slabs
 * straight from pixels/bin2_id and pixels/count hyperslabs of the .cool file, so the pixel table never exists in pageable
 * host memory (the reference reaches the same bytes through cooler's matrix().fetch() per region, coolpuppy/coolpup.py:1053-1057).
 * h2d_ms (nullable): device time of the host-to-device copies alone; h2d_bytes (nullable): bytes they moved.
 */
typedef int (*pup_fill_fn)(void* user, int64_t first, int64_t m, void* bin2_id, int32_t* count);
int pup_load_pixels_stream(pup_ctx* ctx, const int64_t* bin1_offset, int64_t nbins, int64_t nnz, int bin2_bytes,
                           int64_t slab_pixels, pup_fill_fn fill, void* user, double* h2d_ms, int64_t* h2d_bytes);
/*
 * Per-bin vectors (float64[nbins]).  weight: balancing weights, NaN = masked bin; NULL = raw counts
 * (clr_weight_name falsy: no bin is masked).  cov: coverage column for coverage_norm; NULL = none.
 */
int pup_load_bins(pup_ctx* ctx, const double* weight, const double* cov);
/*
 * Optional accelerator: build a rank-bitmap index over the cis part of the loaded table (one 64-byte block
 * per 320 columns per row: first-pixel position + presence bits).  chrom_offset = indexes/chrom_offset
 * (int64[n_chroms+1], chrom_offset[0] == 0, chrom_offset[n_chroms] == nbins).  Windows whose rows and columns
 * lie inside one chromosome then cost one cache line per row instead of a binary search; all other windows
 * (trans) keep using the binary search.  Memory: sum_k nb_k * ceil(nb_k/320) * 64 bytes; returns PUP_ENOMEM
 * (and leaves the engine usable without index) when that exceeds max_bytes (0 = no limit).
 * Results never depend on whether the index exists.
 */
int pup_build_index(pup_ctx* ctx, const int64_t* chrom_offset, int32_t n_chroms, int64_t max_bytes);

/*
 * Expected for the region (pair) the next pup_accumulate calls belong to.
 *   n >= 2 : cis, by-diagonal vector, value for a cell = expected[|col - row|]
 *   n == 1 : trans, one scalar for the whole block
 *   n == 0 : none
 */
int pup_set_expected(pup_ctx* ctx, const double* expected, int64_t n);
/*
 * Expected for MANY regions at once, so that one pup_accumulate call can span regions (replaces the state set
 * by pup_set_expected).  Regions r = 0..n_regions-1 cover global bins [start[r], end[r]), sorted and disjoint.
 *   cis   (pair == NULL): the by-diagonal vector of region r is values[offset[r] .. offset[r]+length[r]); a
 *         snippet uses the vector of the region holding its first row (expected_selections, coolpup.py:913-916)
 *   trans (pair != NULL): pair[r1*n_regions + r2] is the scalar expected of the block with rows in region r1
 *         and columns in region r2 (get_expected_trans, coolpup.py:999-1005); values/offset/length are ignored
 * A snippet outside every region sees NaN expected (all of its cells are invalid).
 */
int pup_set_expected_table(pup_ctx* ctx, const int32_t* start, const int32_t* end, const int64_t* offset,
                           const int64_t* length, int32_t n_regions, const double* values, int64_t n_values,
                           const double* pair);

/*
 * Per-bin coverage of the loaded table (K3), cooltools semantics: every pixel adds its raw count to BOTH of its
 * bins (a main-diagonal pixel twice); pixels with |bin2 - bin1| < ignore_diags count as 0; cov_cis only uses
 * pixels whose two bins share a chromosome (chrom_offset = indexes/chrom_offset, int64[n_chroms+1]).
 * Outputs: host float64[nbins] (integers, exact; after pup_load_pixel_values: float64 sums of the pixel values, the column
 * sums' last bits depend on the order the atomics arrive in); either may be NULL.  Synchronous.
 */
int pup_coverage(pup_ctx* ctx, const int64_t* chrom_offset, int32_t n_chroms, int32_t ignore_diags,
                 double* cov_cis, double* cov_tot);

/* accumulators ------------------------------------------------------------------------------------ */
/* (re)allocate and zero n_tiles accumulators for windows of W = 2*pad+1 bins */
int pup_reset(pup_ctx* ctx, int32_t n_tiles, int32_t pad);

/*
 * Accumulate n snippets.  Snippet s covers rows [r0[s], r0[s]+W) and columns [c0[s], c0[s]+W) of the
 * global bin table; the caller has already dropped windows that leave their region
 * (coolpup.py:1111-1114).  Snippets MUST be grouped by tile: tile_ptr[t]..tile_ptr[t+1] (host array,
 * int64[n_tiles+1], non-decreasing, tile_ptr[0] == 0, tile_ptr[n_tiles] == n) are the snippets of tile t.
 * flip_from (nullable, host int64[n_tiles]): inside tile t the snippets [tile_ptr[t], flip_from[t]) are added as
 * they are and [flip_from[t], tile_ptr[t+1]) are anti-transposed first (flip_snip_func); NULL = no flips.
 * ignore_diags: cis: cells with (col - row) < ignore_diags are masked (must be >= 0: the table is
 * upper-triangular); pass a negative value for trans (no diagonal mask).
 * Asynchronous with respect to the host; results are ordered on the context's stream.
 */
int pup_accumulate(pup_ctx* ctx, const int32_t* r0, const int32_t* c0, int64_t n, const int64_t* tile_ptr,
                   const int64_t* flip_from, int32_t ignore_diags, uint32_t mode);

/*
 * Per-snippet stripes (store_stripes, coolpup.py:1164-1169): for each of the n snippets the centre row
 * (horizontal[s][0..W)) and the centre column reversed (vertical[s][i] = data[W-1-i][pad]) of the masked /
 * normalised window — NaN where the window is masked.  Uses the current weights and expected state; mode accepts
 * PUP_MODE_OOE and PUP_MODE_TRANSPOSE.  Host arrays in, host arrays out ([n][W] float64 each); synchronous.
 */
int pup_stripes(pup_ctx* ctx, const int32_t* r0, const int32_t* c0, int64_t n, int32_t pad, int32_t ignore_diags,
                uint32_t mode, double* horizontal, double* vertical);

/*
 * Per-snippet windows for host-side callbacks: data[s] ([n][W][W] float64, W = 2*pad+1, caller-allocated host array)
 * receives the window of snippet s exactly as the reference's _stream_snips yields snip["data"]
 * (coolpup.py:1104-1158): balanced values, NaN on masked bins / ignored diagonals, divided by expected with
 * PUP_MODE_OOE, or the expected window itself with PUP_MODE_EXPECTED (the reference's exp_snip); reference frame
 * (PUP_MODE_TRANSPOSE undone), not flipped.  height / width (nullable, both or neither): variable-size windows
 * rescaled to W x W as in pup_accumulate_rescaled (_rescale_snip, coolpup.py:1193-1234; PUP_MODE_LOCAL applies).
 * cov_start / cov_end (nullable, [n][W] each): the coverage slices of the window's rows / columns (zoomed when
 * rescaling) with PUP_MODE_COV, NaN otherwise.  Nothing is accumulated; synchronous.
 */
int pup_extract(pup_ctx* ctx, const int32_t* r0, const int32_t* c0, const int32_t* height, const int32_t* width,
                int64_t n, int32_t pad, int32_t ignore_diags, uint32_t mode, double* data, double* cov_start,
                double* cov_end);

/*
 * Rescaled pile-up (rescale=True, PileUpper._rescale_snip, coolpup.py:1193-1234): snippet s is the height[s] x width[s]
 * window at (r0[s], c0[s]); it is masked / normalised like any snippet, zoomed to W x W (W = 2*pad+1 of pup_reset =
 * rescale_size) exactly like cooltools' zoom_array (bilinear blow-up to an integer multiple, block mean, NaN where
 * a NaN input contributed; an all-NaN window adds zeros) and added to its tile.  Coverage vectors are zoomed too
 * (PUP_MODE_COV).  Host arrays only.  Same grouping arguments as pup_accumulate.
 */
int pup_accumulate_rescaled(pup_ctx* ctx, const int32_t* r0, const int32_t* c0, const int32_t* height,
                            const int32_t* width, int64_t n, const int64_t* tile_ptr, const int64_t* flip_from,
                            int32_t ignore_diags, uint32_t mode);

/* wait for all queued work; surfaces asynchronous errors (PUP_ERANGE, PUP_EHIP) */
int pup_sync(pup_ctx* ctx);

/*
 * Copy the accumulators to host memory (any pointer may be NULL to skip it):
 *   sum [n_tiles][W][W] f64, num [n_tiles][W][W] i64, n [n_tiles] i64,
 *   cov_start [n_tiles][W] f64, cov_end [n_tiles][W] f64.     Implies pup_sync.
 */
int pup_fetch(pup_ctx* ctx, double* sum, int64_t* num, int64_t* n, double* cov_start, double* cov_end);

/*
 * Packed accumulator image for cross-GPU reduction (the caller all-reduces it with RCCL, op = sum):
 *   f64 part: n_tiles records of [sum W*W | cov_start W | cov_end W]   = n_tiles*(W*W + 2*W) doubles
 *   i64 part: num [n_tiles][W*W] followed by n [n_tiles]               = n_tiles*(W*W + 1)   int64s
 * pup_packed_sizes reports the two element counts; pup_export copies device->device into
 * caller-provided DEVICE buffers, pup_import overwrites the accumulators from them.
 */
int pup_packed_sizes(pup_ctx* ctx, int64_t* n_f64, int64_t* n_i64);
int pup_export(pup_ctx* ctx, void* dev_f64, void* dev_i64);
int pup_import(pup_ctx* ctx, const void* dev_f64, const void* dev_i64);

/*
 * The same exchange without leaving the library: in-place RCCL all-reduce (sum) of the packed accumulators over the
 * communicator `rccl_comm` (an ncclComm_t created by the caller for this context's device; every rank calls with
 * identical n_tiles / pad), enqueued on the context's stream — no staging copy, no host synchronisation.  librccl is
 * opened with dlopen on first use.  Replaces reduce(sum_pups) across the reference's worker processes
 * (coolpuppy/coolpup.py:1495-1531).  PUP_ENOTSUP when librccl cannot be loaded.
 */
int pup_allreduce(pup_ctx* ctx, void* rccl_comm);

/*
 * Exchange of SOME tiles — by-window pile-ups (coolpuppy/coolpup.py:1696-1755: one pile-up per feature; lib/puputils.py:218-223
 * files a snippet under its two features) keep a tile per feature, and a feature's tile is non-zero only on the rank(s) that
 * piled up its region: a flat all-reduce would move every tile of every rank.
 *   pup_allgather_tiles: tile_ids[rank_ptr[r] .. rank_ptr[r+1]) = the tiles rank r piled windows into (the same lists on every
 *     rank).  Afterwards every listed tile holds, on every rank, the sum of the listing ranks' accumulators for it, added in rank
 *     order (identical doubles everywhere); tiles nobody lists are left alone.  A rank sends its own tiles once (ncclBroadcast
 *     per rank inside one group, on the context's stream, asynchronous like pup_allreduce): message = own tiles, not all tiles.
 *   pup_pack_tiles / pup_unpack_tiles: the two halves for callers that carry the blocks themselves (torch.distributed on
 *     exported buffers, tests): block k of the DEVICE buffers = tile tile_ids[k] in the packed layout (f64: W*W sum, W cov_start,
 *     W cov_end; i64: W*W num, n).  unpack mode 0 overwrites the tiles, 1 adds to them, 2 clears them (buffers may be NULL).
 * PUP_ERANGE for a tile number outside [0, n_tiles).
 */
int pup_pack_tiles(pup_ctx* ctx, const int32_t* tile_ids, int64_t n, void* dev_f64, void* dev_i64);
int pup_unpack_tiles(pup_ctx* ctx, const int32_t* tile_ids, int64_t n, const void* dev_f64, const void* dev_i64, int32_t mode);
int pup_allgather_tiles(pup_ctx* ctx, void* rccl_comm, const int32_t* tile_ids, const int64_t* rank_ptr, int32_t n_ranks,
                        int32_t my_rank);

/*
 * Path of the librccl that pup_allreduce opens: the copy that sits beside the HIP runtime mapped into this process
 * (one ROCm stack per process: a PyTorch-ROCm wheel bundles its own libamdhip64 + librccl, the system ROCm another
 * pair, and mixing them corrupts the heap at exit), or the bare soname "librccl.so.1" when there is none there.  The
 * caller creates its ncclComm_t with THIS library (ncclGetUniqueId / ncclCommInitRank).  Writes a NUL-terminated string,
 * returns its length; PUP_ERANGE when `cap` is too small.  No counterpart in the reference (it merges on the host).
 */
int pup_rccl_path(char* buf, size_t cap);

/* measurement ------------------------------------------------------------------------------------- */
typedef struct pup_stats {
    double  k1_ms;          /* total device time of the pile-up kernel (HIP events), ms */
    double  reduce_ms;      /* total device time of the partial-tile reduction kernel, ms */
    int64_t k1_launches;
    int64_t snippets;       /* snippets accumulated */
    int64_t pixels_in_windows; /* sum over snippets of nnz inside the window (counted on device) */
    int64_t probe_loads;    /* binary-search probes issued (counted on device) */
    double  coverage_ms;    /* device time of the last pup_coverage kernel, ms */
    int64_t staged_regions; /* last pup_accumulate: regions staged by the workgroup-staged kernel (0: it did not run) */
    double  prepare_ms;     /* total device time of the block-key / sort / permute prepass of that kernel, ms */
} pup_stats;
/* profiling: bit 0 = every kernel launch is bracketed by HIP events on the context's stream and the pile-up kernels
 * count the pixels inside the windows (pixels_in_windows, probe_loads); bit 1 (with bit 0) = events only, no counting
 * inside the kernels (what a timed benchmark loop wants).  0 = off. */
int pup_set_profiling(pup_ctx* ctx, int enabled);
int pup_get_stats(pup_ctx* ctx, pup_stats* out);   /* implies pup_sync */
int pup_clear_stats(pup_ctx* ctx);
/* diagnostics of the staged kernel (variant bit 26 of pup_set_tuning switches the collection on): per-wave clock totals of
 * its phases in the last pup_accumulate — out[workgroup][16 waves][8] = {issue, windows, barrier, store, barrier, rows, blocks,
 * look-ahead inside the window phase}.  Returns the number of workgroups (0: nothing collected).  Development aid (tools/k1_probe.py --phases); no counterpart
 * in the reference. */
int pup_debug_timing(pup_ctx* ctx, int64_t* out, int64_t cap);
/* name of the pile-up kernel family the last pup_accumulate / pup_accumulate_rescaled of this context ran ("staged" = K1q, "wide" /
 * "wide_fact" = K1w, "regtile" = K1r, "band" = K1b, "sparse" = K1s, "lds_tile", "expected_diag", "rescale"; "" before the first call).
 * Results never depend on it; tests and benchmarks assert through it that the kernel they mean to measure is the one that ran.  Never NULL. */
const char* pup_last_kernel(const pup_ctx* ctx);
/* How the last pile-up call served by a workgroup-staged kernel got its block order: "binning" (the hand-written passes of
 * csrc/pup_bin.hpp) or "library_sort" (rocPRIM's radix sort: keys beyond 23 bits, or tuning bit 29); "" when the last call was not
 * staged.  Tests pin the shapes that must stay on the binning. */
const char* pup_last_prepass(const pup_ctx* ctx);
/* generic stream timer: record slot (0..7) on the context's stream; elapsed between two slots */
int pup_event_record(pup_ctx* ctx, int slot);
int pup_event_elapsed_ms(pup_ctx* ctx, int slot_begin, int slot_end, float* ms);
/* tuning knobs (0 = library default): chunk_snippets = snippets per chunk (per wave) of the per-window kernels;
 * variant bits: 1 = ignore the index (binary search only), 2 = LDS-tile kernel for every width, 4 = no factorised `num`
 * in the staged kernel, 8 = always use the staged kernel where it is eligible (tests), 16 = never use it, 32 = never use
 * the sparse trans kernel, 64 = no tile pairing in the staged kernel, 128 = the 21-bin staged kernel on 64 x 128 regions
 * (tuning probe), bits 8..19 = waves per interleaved group, bit 20 = rescaled windows zoomed sample by sample instead of by separable
 * weights, bit 21 = the sparse trans kernel without its per-lane hit queues (round 5's form), bit 22 = the 21-bin staged kernel with
 * progressive staging instead of a barrier between blocks (round 6 experiment, slower: DESIGN.md), bit 26 = collect the staged kernel's phase clocks
 * (pup_debug_timing), bit 27 = never stage from the dense band of counts, bit 28 = pile tile pairs up one by one instead of
 * four pairs per staging.
 * Staged kernel: a call of >= 4e5 cis windows (1.5e5 with observed over expected; W <= 31, every window inside one chromosome, index built, at most 64 tiles) is
 * keyed on the device by (tile pair, block of top-left corners) and radix-sorted by block into a scratch copy; when a block
 * holds enough windows on average the call is piled up from LDS-staged regions — 128 x 128 bins for windows up to 21 bins
 * (blocks of 108 x 108 corners at W = 21), 64 x 128 otherwise — by persistent workgroups, tile t together with tile
 * t + T/2 (in coolpuppy's layout a group's ROI and control windows), four such pairs per staging when there are several
 * (grouped pile-ups: neighbouring tile numbers should belong to groups whose windows lie in the same part of the matrix).  No host synchronisation on the
 * way once a call shape has been seen (its block density is remembered).  Input order inside a tile is free.
 * Results do not depend on which kernel ran (integers exactly, sums up to the order of the f64 additions). */
int pup_set_tuning(pup_ctx* ctx, int32_t chunk_snippets, int32_t variant);

/*
 * ---- host-side helpers (no kernel involved) ---------------------------------------------------------------------------
 *
 * pup_host_alloc / pup_host_free: page-locked host memory.  Snippet arrays handed to pup_accumulate from such memory
 * reach the device by asynchronous DMA on the context's stream, ordered behind the kernels already queued — no staging
 * copy and no host synchronisation (north_star: "cooler HDF5 I/O stays on the host with pinned hipMemcpyAsync overlap").
 * Pageable arrays keep working: the runtime stages them and the call returns when the staging copy is done.  Either way
 * the arrays must stay untouched until the next pup_sync / pup_fetch.
 */
int pup_host_alloc(void** ptr, size_t bytes);
int pup_host_free(void* ptr);

/*
 * pup_host_windows: the windows of one region (pair) — its n ROI windows followed by nshifts shifted control copies
 * of all of them — as engine input.  Replaces the per-snippet loop of CoordCreator._control_regions
 * (coolpuppy/coolpup.py:387-453: both sides of copy m move by round(shift[m]*sign[m]/resolution) bins, half-to-even; shift
 * and sign are the caller's RNG draws, in the reference's order) and the bounds test of _stream_snips (:1105-1114: a window
 * is kept when lo1 <= r0, r0+h <= hi1, lo2 <= c0, c0+w <= hi2 with r0 = st1+off1, c0 = st2+off2).  Kept windows are written
 * in order to r0 / c0 (capacity n*(1+nshifts)); code (nullable) is a per-ROI-row value copied to code_out for every kept
 * window of that row.  Returns the number kept (< 0: bad arguments); *n_roi_kept of them come from the ROI rows.
 */
int64_t pup_host_windows(const int32_t* st1, const int32_t* st2, const int32_t* code, int64_t n,
                         const int32_t* shift, const int32_t* sign, int32_t nshifts, double resolution,
                         int64_t off1, int64_t off2, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2,
                         int32_t h, int32_t w, int32_t* r0, int32_t* c0, int32_t* code_out, int64_t* n_roi_kept);

/*
 * pup_host_control_windows: the nshifts shifted control copies of pup_host_windows ALONE (same arguments and rules, capacity
 * n * nshifts, no ROI windows in front).  With it the windows of a whole pile-up are written straight to where one
 * pup_accumulate call wants them — every region's ROI windows first (they need no draw), then, region after region as the
 * reference's draws arrive (coolpuppy/coolpup.py:420-436), its copies — in page-locked memory, without per-region arrays.
 */
int64_t pup_host_control_windows(const int32_t* st1, const int32_t* st2, const int32_t* code, int64_t n,
                                 const int32_t* shift, const int32_t* sign, int32_t nshifts, double resolution,
                                 int64_t off1, int64_t off2, int64_t lo1, int64_t hi1, int64_t lo2, int64_t hi2,
                                 int32_t h, int32_t w, int32_t* r0, int32_t* c0, int32_t* code_out);

/*
 * pup_host_mt_randint: the reference's random draws for the control windows — np.random.randint(low, high, m) of the LEGACY
 * numpy generator (coolpuppy/coolpup.py:420-436: randint(minshift, maxshift, m), and np.random.choice([-1, 1], m), which the
 * legacy generator implements as randint(0, 2, m)) — produced from the generator's state (key[624] and *pos as
 * np.random.get_state() returns them) and advancing it exactly as numpy does: MT19937 words, tempered, masked, rejected
 * above the range.  out[i] = offset + scale * draw_i, elements of out_bytes = 4 or 8 bytes (out may be NULL: draw and discard —
 * regions another rank piles up).
 * Returns PUP_OK or PUP_EINVAL (needs 0 < high - low <= 2^32).  tests/test_host_misc.py checks numbers and final state
 * against the installed numpy.
 */
int pup_host_mt_randint(uint32_t* key, int32_t* pos, int64_t low, int64_t high, int64_t m, int64_t scale, int64_t offset,
                        void* out, int32_t out_bytes);

/*
 * pup_host_lut_i32 / pup_host_count_le: two array passes of a GROUPED pile-up's plan (by distance band, strand, ...: coolpuppy/coolpup.py:
 * 28-51, 1757-1919) on several threads.  pup_host_lut_i32: out[i] = lut[codes[i]] (+ add for i >= add_from) — the tile numbers of a region's
 * windows from their group codes, the controls' half offset by the number of groups; PUP_ERANGE for a code outside [0, n_lut).
 * pup_host_count_le: out[i] = number of edges <= values[i] = np.searchsorted(edges, values, side="right") for up to 64 sorted edges (the
 * distance band of a pair).
 */
int pup_host_lut_i32(const int32_t* lut, int64_t n_lut, const int32_t* codes, int64_t n, int64_t add_from, int32_t add, int32_t* out);
int pup_host_count_le(const double* edges, int32_t n_edges, const double* values, int64_t n, int32_t* out);

/*
 * pup_host_normalise_tiles: the finaliser's arithmetic (coolpuppy/coolpup.py:1533-1545: data / num, the ratio to the control's
 * data / num, +inf -> NaN) on `count` cells at once, in place of `sum` (and of `csum`, which holds the control's quotient
 * afterwards), on several threads; csum / cnum NULL: no control.  Same operations in the same order as the reference's numpy
 * expressions.  Returns PUP_OK or PUP_EINVAL.
 */
int pup_host_normalise_tiles(double* sum, const int64_t* num, double* csum, const int64_t* cnum, int64_t count);

/*
 * pup_host_mt_randint_plan: n_calls consecutive pup_host_mt_randint calls as ONE job — the draws of every region of a pile-up
 * (coolpuppy/coolpup.py:420-436, region after region: randint(minshift, maxshift, m_r), then choice([-1, 1], m_r)), whose sizes
 * are known before the first of them (pup_host_pair_region_counts).  Call k draws m[k] numbers of [low[k], high[k]) into out[k]
 * (elements of out_bytes[k] = 4 or 8 bytes, as offset[k] + scale[k] * draw; NULL: draw and discard).  Numbers and final state are
 * those of the calls made one by one; the raw MT19937 stream is produced a buffer ahead by one thread while a pool of workers
 * counts / places the candidates of the buffer before (the per-call form starts its threads twice per call).
 * Returns PUP_OK, PUP_EINVAL, or PUP_ENOTSUP when two calls reject over different ranges (make the calls one by one then).
 */
int pup_host_mt_randint_plan(uint32_t* key, int32_t* pos, int32_t n_calls, const int64_t* low, const int64_t* high,
                             const int64_t* m, const int64_t* scale, const int64_t* offset, void* const* out,
                             const int32_t* out_bytes);

/*
 * pup_host_take_rows: frame.take(order) for the numeric columns of the feature frame CoordCreator sorts (the stable sort of
 * _binnify, coolpuppy/coolpup.py:489-527, permutes every column): dst[c][i] = src[c][order[i]] for ncols columns of n_src
 * elements of esize[c] (1, 2, 4 or 8) bytes, i < n, on several threads.  PUP_EINVAL for an index outside [0, n_src) or an
 * unsupported element size.
 */
int pup_host_take_rows(int32_t ncols, const void* const* src, void* const* dst, const int32_t* esize, const int64_t* order,
                       int64_t n, int64_t n_src);

/*
 * pup_host_factorize_ptr: codes of n pointer values by identity, in order of first appearance (first[j] = index of the first
 * occurrence of value j) — the chromosome columns of the feature frame CoordCreator sorts (coolpuppy/coolpup.py:489-527) are object
 * arrays pointing at a few dozen string objects.  Returns the number of distinct pointers, -1 when there are more than max_uniq
 * (nothing usable in the outputs), -2 for bad arguments.
 */
int64_t pup_host_factorize_ptr(const uintptr_t* ptrs, int64_t n, int32_t* codes, int64_t* first, int64_t max_uniq);

/*
 * pup_host_argsort: order[i] = index of the i-th smallest of n keys of `bits` significant bits, equal keys in index order —
 * the stable sort of the feature frame by (chrom1, chrom2, start1, start2) in CoordCreator._binnify (coolpuppy/coolpup.py:489-527)
 * on one packed key per row.  Multi-threaded LSD radix sort.  PUP_OK / PUP_EINVAL.
 */
int pup_host_argsort(const uint64_t* keys, int64_t n, int32_t bits, int64_t* order);

/*
 * pup_host_sort_pairs: what CoordCreator.process does to a BEDPE feature table before any window exists — centres
 * c = (start + end) / 2 (double), the distance filter mindist <= |c2 - c1| <= maxdist (coolpuppy/coolpup.py:296-321), the stable
 * sort by (chrom1, chrom2, start1, start2) of _binnify (:489-527; chromosome order given as rank[code], nu codes) — as one call:
 * rows[i] = source row of sorted row i, and the six columns in that order (s1o .. c2o; capacity n each).  Returns the number of
 * rows kept (0 .. n), PUP_EINVAL, or PUP_ENOTSUP when a start is negative, a code lies outside [0, nu) or the packed key
 * (chromosome pair | start1 / gcd | start2 / gcd) exceeds 63 bits — the caller then takes its general path.  *flags: bit 0 = rows
 * were dropped, bit 1 = the kept rows were out of order.
 */
int64_t pup_host_sort_pairs(const int64_t* s1, const int64_t* e1, const int64_t* s2, const int64_t* e2, const int32_t* c1,
                            const int32_t* c2, int64_t n, const int64_t* rank, int32_t nu, double mindist, double maxdist,
                            int64_t* rows, int64_t* s1o, int64_t* e1o, int64_t* s2o, int64_t* e2o, int32_t* c1o, int32_t* c2o,
                            int32_t* flags);

/*
 * pup_host_pair_region_counts: the number of BEDPE rows every view region will pile up — rows that pass the distance filter
 * (as pup_host_sort_pairs) with both anchors inside the region (start >= region start, end < region end: the reference's region
 * filter, coolpuppy/coolpup.py:549-562), region k lying on chromosome code reg_code[k] — from the UNSORTED table.  The control
 * shifts of a pile-up (coolpuppy/coolpup.py:420-436: rows x nshifts draws per region, in region order) depend on nothing else, so
 * their drawing can start while the table is still being sorted.  PUP_OK / PUP_EINVAL.
 */
int pup_host_pair_region_counts(const int64_t* s1, const int64_t* e1, const int64_t* s2, const int64_t* e2, const int32_t* c1,
                                const int32_t* c2, int64_t n, double mindist, double maxdist, const int32_t* reg_code,
                                const int64_t* reg_start, const int64_t* reg_end, int32_t n_regions, int64_t* counts);

/*
 * pup_host_group_tiles: the windows of several regions gathered into ONE pup_accumulate call — stable grouping by tile id
 * over the concatenation of the parts (what reduce(sum_pups) over per-region dicts amounts to for the tile layout,
 * coolpuppy/coolpup.py:1495-1531).  r0_out / c0_out hold sum(len) entries, tile_ptr T+1.
 */
int pup_host_group_tiles(int32_t n_parts, const int32_t* const* r0, const int32_t* const* c0, const int32_t* const* tile,
                         const int64_t* len, int32_t T, int32_t* r0_out, int32_t* c0_out, int64_t* tile_ptr);

/*
 * pup_host_group_tiles_runs: the same with RUN-CODED parts — tile[p] == NULL: windows [0, split[p]) of part p belong to tile
 * tile_a[p], the others to tile_b[p] (an ungrouped region with controls: ROI windows, then the shifted copies,
 * coolpuppy/coolpup.py:387-453) — for which no per-window tile array exists.  split / tile_a / tile_b may be NULL when no part is
 * run-coded.
 */
int pup_host_group_tiles_runs(int32_t n_parts, const int32_t* const* r0, const int32_t* const* c0, const int32_t* const* tile,
                              const int64_t* split, const int32_t* tile_a, const int32_t* tile_b, const int64_t* len, int32_t T,
                              int32_t* r0_out, int32_t* c0_out, int64_t* tile_ptr);

#ifdef __cplusplus
}
#endif
#endif /* PUP_HIP_H */
