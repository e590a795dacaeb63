/* genodsp_hip.h -- C ABI of libgenodsp_hip.so, the MI355X (gfx950) device side
 * of the genodsp hot path.
 *
 * Every entry point is what a genodsp operator's `apply` (genodsp_interface.h:76-93
 * in the reference) would call once its chromosome vector lives in HBM: plain
 * pointers and sizes, no C++ or torch types.  `d_` pointers are device memory
 * (hipMalloc, or any allocation a HIP stream can address, 16-byte aligned);
 * `h_` pointers are host memory.  `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Calls only enqueue work unless stated
 * otherwise.  Lengths are u32 like the reference's (genodsp_interface.h:45).
 *
 * Return value: 0 on success, a GDSP_E* code otherwise; gdsp_last_error()
 * gives the message.  The library never falls back to a CPU path.
 */
#ifndef GENODSP_HIP_H
#define GENODSP_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDSP_OK        0
#define GDSP_EINVAL    1   /* bad argument                         */
#define GDSP_EHIP      2   /* a HIP runtime call or launch failed  */
#define GDSP_ENOMEM    3

/* FIR arithmetic: EXACT rounds the product and the sum separately, ascending tap
 * order, and is bit-identical to the reference's loop (sum.c:659-662);
 * FMA fuses them (one rounding per tap instead of two). */
#define GDSP_FIR_EXACT 0
#define GDSP_FIR_FMA   1
/* HANN (gdsp_smooth only) uses what the window is -- tap k = c*(1 - cos(w(k+1))) -- and builds
 * each output from block sums of x and of x*exp(jwe): ~45 operations per base for any tap
 * count, additions only (no differences of running sums).  Within W * 2^-52 * sum|w_k v_k| of
 * the reference, like FMA, and no further from the exact value than the reference is.
 * Unlike EXACT and FMA it is not shift invariant (a flat input gives outputs that differ in
 * their last bits), so gdsp_smooth_local_extrema evaluates HANN as FMA.
 * Odd windows of 81..50001 taps have such a kernel (W=101 a dedicated one; beyond 4001 taps the block totals go through HBM,
 * gdsp_hann_far.hip); shorter ones are evaluated as FMA.  A tile that holds inf / NaN / a magnitude >= 2^1017 is evaluated tap by
 * tap (FMA's bits there), so the mode differs from the reference by rounding only on any input. */
#define GDSP_FIR_HANN  2

/* interval overlap operators, values as genodsp_interface.h:157-159 */
#define GDSP_OVERLAP_SUM 0
#define GDSP_OVERLAP_MIN 1
#define GDSP_OVERLAP_MAX 2

const char* gdsp_last_error (void);
const char* gdsp_version    (void);
/* debugging aid: 1 and *value = the pattern when the environment holds GDSP_POISON=<double|nan>, else 0.  Every gdsp_malloc
 * is then filled with it, and the driver refills a vector's partner after every flip: a kernel that reads memory nobody
 * wrote changes the output (tests/test_cli_hip.py::test_no_operator_reads_memory_nobody_wrote). */
int gdsp_poison (double* value);

/* ---- runtime plumbing (what genodsp.c:865-878 / :1890-2037 do with calloc) ---- */
int gdsp_device_count   (int* count);
int gdsp_set_device     (int device);
int gdsp_get_device     (int* device);
int gdsp_malloc         (void** d_ptr, size_t bytes);      /* (filled with GDSP_POISON when that is set, see gdsp_poison) */
int gdsp_free           (void* d_ptr);
int gdsp_host_alloc     (void** h_ptr, size_t bytes);          /* pinned staging  */
int gdsp_host_free      (void* h_ptr);
int gdsp_memcpy_h2d     (void* d_dst, const void* h_src, size_t bytes, void* stream);
int gdsp_memcpy_d2h     (void* h_dst, const void* d_src, size_t bytes, void* stream);
int gdsp_memcpy_d2d     (void* d_dst, const void* d_src, size_t bytes, void* stream);
int gdsp_memcpy_peer    (void* d_dst, int dstDevice, const void* d_src, int srcDevice, size_t bytes, void* stream); /* GPU to GPU (xGMI) */
int gdsp_memset         (void* d_dst, int byte, size_t bytes, void* stream);
int gdsp_stream_create  (void** stream);
int gdsp_stream_destroy (void* stream);
int gdsp_stream_sync    (void* stream);
int gdsp_device_sync    (void);                                /* every stream of the current device */
int gdsp_event_create   (void** event);
int gdsp_event_destroy  (void* event);
int gdsp_event_record   (void* event, void* stream);
int gdsp_stream_wait_event (void* stream, void* event);         /* work queued on stream from now on starts after the event (independent
                                                                 * chromosomes on alternating streams hide the drain between kernels: +3..5 %) */
int gdsp_event_elapsed_ms (void* start, void* stop, float* ms); /* syncs on stop   */
int gdsp_fill           (double* d_v, uint32_t n, double val, void* stream);

/* ---- sum.c ---------------------------------------------------------------------- */

/* Host: the reference's Hann taps (sum.c:632-645), W odd >= 3. */
int gdsp_hann_taps (uint32_t W, double* h_taps);

/* op_smooth_apply (sum.c:616-676): zero-padded W-tap FIR, out-of-place. */
typedef struct gdsp_fir_plan gdsp_fir_plan;
int gdsp_fir_plan_create  (gdsp_fir_plan** plan, const double* h_taps, uint32_t W);
int gdsp_fir_plan_destroy (gdsp_fir_plan* plan);
int gdsp_fir_apply        (const gdsp_fir_plan* plan, const double* d_in, double* d_out,
                           uint32_t n, int mode, void* stream);
/* convenience: Hann plan for W cached inside the library (per device) */
int gdsp_smooth           (const double* d_in, double* d_out, uint32_t n, uint32_t W,
                           int mode, void* stream);

/* `= smooth W = localmax|localmin N` in one pass (sum.c:616-676 feeding minmax.c:1183-1227 /
 * :981-1022): the smoothed tile is tested in LDS and only the peaks track is written.
 * Bit-identical to gdsp_smooth followed by gdsp_local_extrema.  Fusable for W=101, N<=129.
 * Synchronisation: for W=101 and N<=15 the call takes the filtered route (gdsp_peaks.hip), which reads a probe's counts
 * back to choose each vector's form: it WAITS for the stream once per table of <= 32 vectors (everything queued on the
 * stream before it included) and cannot be captured into a graph.  GDSP_PEAKS_FLAT=0 keeps the decision on the device
 * (no wait; vectors of flat stretches then take the direct kernel), GDSP_PEAKS_FILTER=0 the direct kernel throughout. */
int gdsp_smooth_local_extrema_fusable (uint32_t W, uint32_t N);
int gdsp_smooth_local_extrema (const double* d_in, double* d_out, uint32_t n, uint32_t W, int mode,
                               uint32_t N, int wantMax, double fill, void* stream);

/* op_sliding_sum_apply (sum.c:420-463): centred window sum / denom, out-of-place.
 * Bit-identical to the reference whenever every partial sum is exact (integer
 * or dyadic signals); otherwise within the running-sum rounding bound. */
int gdsp_sliding_sum      (const double* d_in, double* d_out, uint32_t n, uint32_t W,
                           double denom, void* stream);
/* op_window_sum_apply (sum.c:211-252), in place. */
int gdsp_window_sum       (double* d_v, uint32_t n, uint32_t W, double denom, int useActual,
                           double zeroVal, void* stream);
/* op_cumulative_sum_apply (sum.c:776-792), in place; d_work >= gdsp_cumulative_sum_work(n) bytes. */
size_t gdsp_cumulative_sum_work (uint32_t n);
int gdsp_cumulative_sum   (double* d_v, uint32_t n, void* d_work, void* stream);

/* ---- minmax.c ------------------------------------------------------------------- */

/* op_local_maxima_apply / op_local_minima_apply (minmax.c:1183-1227, :981-1022) */
int gdsp_local_extrema (const double* d_in, double* d_out, uint32_t n, uint32_t N,
                        int wantMax, double fill, void* stream);
/* op_best_local_max_apply / _min_ (minmax.c:1616-1721, :1369-1474): sliding max/min */
int gdsp_best_extrema  (const double* d_in, double* d_out, uint32_t n, uint32_t W,
                        int wantMax, void* stream);

/* Any window length (the tiled kernels above stop at what one LDS tile holds and return
 * GDSP_EINVAL beyond): the *_any forms use the tiled kernel when it applies and otherwise
 * whole-vector passes through d_work (>= gdsp_long_window_work(n) bytes of device memory). */
size_t gdsp_long_window_work (uint32_t n);
int gdsp_best_extrema_any  (const double* d_in, double* d_out, uint32_t n, uint32_t W, int wantMax,
                            void* d_work, size_t workBytes, void* stream);
int gdsp_local_extrema_any (const double* d_in, double* d_out, uint32_t n, uint32_t N, int wantMax, double fill,
                            void* d_work, size_t workBytes, void* stream);
int gdsp_sliding_sum_any   (const double* d_in, double* d_out, uint32_t n, uint32_t W, double denom,
                            void* d_work, size_t workBytes, void* stream);

/* ---- morphology.c --------------------------------------------------------------- */

/* All four binarise with v > T and write only one/zero.  Out-of-place
 * (d_out may not alias d_in). */
int gdsp_dilate (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right,
                 double T, double one, double zero, void* stream);   /* :882-1072  */
int gdsp_erode  (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right,
                 double T, double one, double zero, void* stream);   /* :1331-1454 */
/* `= dilate = erode [= binarize]` in one pass (morphology.c:882-1072 -> :1331-1454 ->
 * logical.c:216-268): the dilated set stays in LDS as a bit mask.  Bit-identical to the
 * three calls in sequence; each stage keeps its own threshold and output values. */
int gdsp_dilate_erode_fusable (uint32_t dLeft, uint32_t dRight, uint32_t eLeft, uint32_t eRight);   /* 1: fused for any vector length */
int gdsp_dilate_erode (const double* d_in, double* d_out, uint32_t n,
                       uint32_t dLeft, uint32_t dRight, double dT, double dOne, double dZero,
                       uint32_t eLeft, uint32_t eRight, double eT, double eOne, double eZero,
                       int binarize, double bT, int bTiesAbove, double bOne, double bZero, void* stream);
int gdsp_close  (const double* d_in, double* d_out, uint32_t n, double closingLength,
                 double T, double one, double zero, void* stream);   /* :231-319   */
int gdsp_open   (const double* d_in, double* d_out, uint32_t n, double openingLength,
                 double T, double one, double zero, void* stream);   /* :529-605   */

/* Any length (the reference accepts any, morphology.c:696-866, :1163-1315; the tiled kernels above return GDSP_EINVAL
 * once a window reaches beyond the 262 k bases one LDS tile can stage): the tiled kernel when it applies, otherwise
 * the set as bits plus per-word next / previous-member tables in d_work (>= gdsp_long_window_work(n) bytes). */
int gdsp_dilate_any (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right,
                     double T, double one, double zero, void* d_work, size_t workBytes, void* stream);
int gdsp_erode_any  (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right,
                     double T, double one, double zero, void* d_work, size_t workBytes, void* stream);
int gdsp_close_any  (const double* d_in, double* d_out, uint32_t n, double closingLength,
                     double T, double one, double zero, void* d_work, size_t workBytes, void* stream);
int gdsp_open_any   (const double* d_in, double* d_out, uint32_t n, double openingLength,
                     double T, double one, double zero, void* d_work, size_t workBytes, void* stream);

/* ---- one launch per operator per device (replaces the chromosome loop of genodsp.c:909-921) ----------------------
 * The reference applies an operator to one chromosome after the other.  On the GPU every launch ramps up and drains
 * (10-15 % of a short kernel's time), so a device that holds several chromosomes -- or several stretches of them --
 * is given ONE grid that covers all of its vectors: items[i] = (input, output, length) of vector i, all on the current
 * device, all 16-byte aligned, outputs distinct from inputs for the out-of-place operators (in-place operators use
 * d_out only).  Results are those of the single-vector calls bit for bit (the same kernels; only the block-to-tile
 * map differs).  Any number of items; tables of 32 vectors travel in the kernel arguments. */
typedef struct gdsp_batch_item { const double* d_in;  double* d_out;  uint32_t n; } gdsp_batch_item;
int gdsp_smooth_batch               (const gdsp_batch_item* items, int nitems, uint32_t W, int mode, void* stream);
int gdsp_smooth_local_extrema_batch (const gdsp_batch_item* items, int nitems, uint32_t W, int mode,
                                     uint32_t N, int wantMax, double fill, void* stream);
int gdsp_local_extrema_batch        (const gdsp_batch_item* items, int nitems, uint32_t N, int wantMax, double fill, void* stream);
int gdsp_best_extrema_batch         (const gdsp_batch_item* items, int nitems, uint32_t W, int wantMax, void* stream);
int gdsp_dilate_batch               (const gdsp_batch_item* items, int nitems, uint32_t left, uint32_t right,
                                     double T, double one, double zero, void* stream);
int gdsp_erode_batch                (const gdsp_batch_item* items, int nitems, uint32_t left, uint32_t right,
                                     double T, double one, double zero, void* stream);
int gdsp_dilate_erode_batch         (const gdsp_batch_item* items, int nitems,
                                     uint32_t dLeft, uint32_t dRight, double dT, double dOne, double dZero,
                                     uint32_t eLeft, uint32_t eRight, double eT, double eOne, double eZero,
                                     int binarize, double bT, int bTiesAbove, double bOne, double bZero, void* stream);
int gdsp_binarize_batch             (const gdsp_batch_item* items, int nitems, double T, int tiesAbove, double one, double zero, void* stream);
int gdsp_clip_batch                 (const gdsp_batch_item* items, int nitems, int haveMin, double minVal, int haveMax, double maxVal, void* stream);
int gdsp_erase_batch                (const gdsp_batch_item* items, int nitems, int haveMin, double minVal, int haveMax, double maxVal,
                                     int keepInside, double zero, void* stream);
int gdsp_add_constant_batch         (const gdsp_batch_item* items, int nitems, double c, void* stream);
int gdsp_abs_batch                  (const gdsp_batch_item* items, int nitems, void* stream);
/* GDSP_EINVAL from a *_batch call whose parameters the single-vector call would also refuse, or for which no tiled
 * kernel exists (the caller then loops over the vectors with the single-vector / *_any forms). */

/* ---- logical.c, mask.c, add.c (in place) ---------------------------------------- */
int gdsp_binarize     (double* d_v, uint32_t n, double T, int tiesAbove, double one, double zero,
                       void* stream);                                 /* logical.c:216-268 */
int gdsp_clip         (double* d_v, uint32_t n, int haveMin, double minVal, int haveMax,
                       double maxVal, void* stream);                  /* mask.c:850-924    */
int gdsp_erase        (double* d_v, uint32_t n, int haveMin, double minVal, int haveMax,
                       double maxVal, int keepInside, double zero, void* stream); /* mask.c:1147-1243 */
int gdsp_add_constant (double* d_v, uint32_t n, double c, void* stream);   /* add.c:726-741   */
int gdsp_abs          (double* d_v, uint32_t n, void* stream);             /* add.c:1038-1049 */
int gdsp_invert       (double* d_v, uint32_t n, double mid, void* stream); /* add.c:927-937   */
/* op_map_apply (map.c:194-381): piecewise-linear mapping through nknots (in,out) knots sorted by
 * `in` (device arrays).  Bit-identical to the reference for strictly increasing knots. */
int gdsp_map          (double* d_v, uint32_t n, const double* d_knotIn, const double* d_knotOut,
                       uint32_t nknots, void* stream);
/* clump_search (clump.c:494-736), op_clump_apply (above != 0) / op_skimp_apply: bases in stretches
 * of at least minLength whose average is >= (<=) `average` become `one`, the rest `zero`; each
 * merged run is trimmed to its first and last base on the right side of the threshold.  In place;
 * d_work >= gdsp_clump_work(n) bytes.  Whole-vector scans; bit-identical to the reference whenever
 * the running sum of (v - average) is exactly representable (depth against a dyadic threshold). */
size_t gdsp_clump_work (uint32_t n);
int gdsp_clump        (double* d_v, uint32_t n, double average, uint32_t minLength, int above,
                       double one, double zero, void* d_work, void* stream);
/* add.c:909-923 / percentile.c:434-530: d_minmax[0]=min(d_minmax[0], min over sample),
 * d_minmax[1]=max(...), d_minmax[2]+=count (as double); sample = every window-th value
 * with lo <= v <= hi.  Initialise with gdsp_minmax_init. */
int gdsp_minmax_init   (double* d_minmax, void* stream);
int gdsp_minmax_update (const double* d_v, uint32_t n, uint32_t window, double lo, double hi,
                        double* d_minmax, void* stream);

/* ---- percentile.c:392-751 -------------------------------------------------------- */

/* Exact order statistic by radix select on the order-preserving 64-bit image of
 * the doubles (-0.0 folded onto +0.0: the reference's comparator, genodsp.c:2262-2270,
 * cannot tell them apart).  The signal is left untouched (the reference scrambles
 * it, percentile.c:34-36).  One pass = gdsp_select_hist_init, then one
 * gdsp_select_histogram per chromosome on each GPU, then a SUM of the bins over
 * GPUs (min/max of the two trailing words) -- the only collective on the whole
 * path -- then gdsp_select_pick on the host to find the bucket holding rank k.
 *   sample  : every window-th value with !(v < lo) && !(v > hi)   (percentile.c:559-561)
 *   digit   : key bits [shift, shift+bits), bits <= GDSP_SELECT_MAX_BITS
 *   prefix  : only keys whose bits above the digit equal prefix's are counted
 *   d_hist  : (1<<bits) u64 counts, then the smallest and the largest matching key
 *             (when those two are equal every remaining candidate is that value). */
#define GDSP_SELECT_MAX_BITS 13
int gdsp_select_hist_init (uint64_t* d_hist, int bits, void* stream);
int gdsp_select_histogram (const double* d_v, uint32_t n, uint32_t window, double lo, double hi,
                           int shift, int bits, uint64_t prefix, uint64_t* d_hist, void* stream);
/* Host: bucket holding 0-based rank k among the counted keys, and the rank inside it. */
int gdsp_select_pick      (const uint64_t* h_hist, int bits, uint64_t k, uint32_t* bucket, uint64_t* kWithin);
/* Host: key image <-> double */
double   gdsp_key_to_double (uint64_t key);
uint64_t gdsp_double_to_key (double v);
/* Host: the reference's rank formula, percentile.c:587-589 and :688-710 */
uint32_t gdsp_percentile_rank (uint32_t numValues, uint32_t pThousandths);

/* op_percentile_apply (percentile.c:392-751) end to end: the exact percentiles of the sampled
 * genome -- every window-th value v with !(v < lo) && !(v > hi) of every source vector --
 * non-destructive.  values[i] = the order statistic of rank gdsp_percentile_rank(count, p[i]);
 * *count = the population size (0: nothing qualifies, values untouched).
 * Sources may sit on several devices of this process (the counts are added on the host); with
 * one process per GPU pass `reduce`, which must replace words[0..count) by their sum (op 0),
 * minimum (op 1) or maximum (op 2) over all ranks and return 0 -- every rank then takes the same
 * decisions and gets the same values.  That reduction (a few KiB per step) is the path's only
 * collective.
 * strategy: AUTO brackets the ranks with pivots from a strided subsample of `sampleTarget`
 * values (0 = one value in 1024, between 2^16 and 2^20) and settles all percentiles in one counting pass over the
 * population when it is larger than 2^20 (or than an explicit sampleTarget), RADIX is five histogram passes per percentile (the fallback of AUTO
 * whenever a bracket misses), BRACKET forces the first route (tests). */
typedef struct gdsp_select_source { const double* d_v; uint32_t n; int device; void* stream; } gdsp_select_source;
typedef int (*gdsp_reduce_fn) (void* ctx, uint64_t* words, size_t count, int op);
#define GDSP_SELECT_AUTO    0
#define GDSP_SELECT_RADIX   1
#define GDSP_SELECT_BRACKET 2
int gdsp_percentiles (const gdsp_select_source* sources, int nsources, uint32_t window, double lo, double hi,
                      const uint32_t* pThousandths, int npercentiles, int strategy, uint32_t sampleTarget,
                      gdsp_reduce_fn reduce, void* reduceCtx, double* values, uint64_t* count);
/* The path's only collective (gdsp_comm.hip): all-reduce of the counters across the GPUs of one node, RCCL over
 * xGMI.  One process drives several devices: gdsp_comm_create makes one RCCL communicator per device
 * (ncclCommInitAll; RCCL is dlopen'ed on first use, a one-GPU run never loads it) and the all-reduces run in place
 * on d_bufs[rank] -- rank r is devices[r] -- on streams[rank] (NULL: that device's default stream).  op: 0 sum,
 * 1 min, 2 max.  Replaces, for chromosomes dealt over GPUs, what the reference's single thread sees at once:
 * percentile.c:547-683 (the sort of the sampled genome), add.c:909-923 (invert's global min / max). */
typedef struct gdsp_comm gdsp_comm;
int gdsp_comm_create        (gdsp_comm** comm, const int* devices, int ndevices);
int gdsp_comm_destroy       (gdsp_comm* comm);
int gdsp_comm_size          (const gdsp_comm* comm);
int gdsp_comm_device        (const gdsp_comm* comm, int rank);
int gdsp_comm_rccl_version  (int* version);
int gdsp_comm_allreduce_u64 (gdsp_comm* comm, uint64_t* const* d_bufs, size_t count, int op, void* const* streams);
int gdsp_comm_allreduce_f64 (gdsp_comm* comm, double* const* d_bufs, size_t count, int op, void* const* streams);
/* gdsp_percentiles over the devices of THIS process: all-reduce its histograms and counters in HBM through `comm`
 * (whose devices must be the sources' devices) instead of adding host copies; NULL switches back. */
int gdsp_percentiles_use_comm (gdsp_comm* comm);
/* gdsp_percentiles with one process per GPU: hand every buffer to be reduced to the caller as DEVICE words on the
 * stream the counts were produced on (the caller runs its collective there, e.g. torch.distributed over RCCL, and
 * returns 0); NULL switches back to the host hook of gdsp_percentiles.  op as above. */
typedef int (*gdsp_device_reduce_fn) (void* ctx, uint64_t* d_words, size_t count, int op, void* stream);
int gdsp_percentiles_use_device_reduce (gdsp_device_reduce_fn fn, void* ctx);

/* `= percentile P = binarize --threshold=percentileP` (percentile.c:392-751 feeding logical.c:216-268) in ONE read of the
 * signal: the counting pass knows the bracket the percentile lies in before it knows the percentile, so it writes
 * one / zero for every base outside that bracket as it counts and queues the positions inside it (0.1 % of real-valued
 * coverage) for a fix-up once the value is known -- 16 B/base for the pair instead of 24.  d_out[i] receives
 * binarize(source i) against values[which] (out of place: the sources are left intact); *onePass = 1 when every source
 * went that way, 0 when some (or all) were binarized by a pass of their own -- strided sampling (window > 1), the radix
 * route, a percentile that fell outside its bracket, more than n/16 bases inside the bracket.  Same values, same
 * outputs either way.  *count = 0 (nothing qualifies): no output is written. */
typedef struct gdsp_percentile_binarize { int which;  int tiesAbove;  double one, zero;  double* const* d_out; } gdsp_percentile_binarize;
int gdsp_percentiles_binarize (const gdsp_select_source* sources, int nsources, uint32_t window, double lo, double hi,
                               const uint32_t* pThousandths, int npercentiles, int strategy, uint32_t sampleTarget,
                               gdsp_reduce_fn reduce, void* reduceCtx, double* values, uint64_t* count,
                               const gdsp_percentile_binarize* fuse, int* onePass);

/* what the last gdsp_percentiles call of this process did: [0] route taken (GDSP_SELECT_RADIX or
 * _BRACKET), [1] population, [2] subsample size, [3] candidates kept on this rank, [4] percentiles
 * that fell back to the radix route, [5] histogram passes over the population, [6] 1 when a fused binarize
 * was settled in the counting pass for every source, [7] 1 when the call was decided on the device with one read-back
 * (one device, nothing to reduce with; GDSP_PERCENTILE_RESIDENT_OFF forbids it) */
void gdsp_percentiles_stats (uint64_t out[8]);

/* ---- genodsp.c read_intervals / add.c / multiply.c ------------------------------ */

/* Interval-driven writes.  The host routes intervals to this chromosome, applies
 * origin and clipping rules, and bins their indices (file order kept) into tiles of
 * gdsp_interval_tile() bases with gdsp_bin_intervals; the device then applies, to
 * every base, the intervals covering it IN FILE ORDER -- bit-identical to the
 * reference's `for ix in [start,end)` loops for any values.
 *   gdsp_apply_intervals: read_intervals genodsp.c:1305-1331 (sum/min/max, and the
 *     first-touch rule when `clear`), op_add/op_subtract add.c:282-283, :575-576
 *     (pass negated values for subtract).
 *   gdsp_scale_intervals: op_multiply multiply.c:326-345, op_divide :706-740; bases
 *     under no interval become 0 (multiply) or +-infinityVal (divide). */
#define GDSP_CLEAR_FIRST_TOUCH 1   /* a base still equal to missingVal is assigned (genodsp.c:1311) */
#define GDSP_CLEAR_FILL        2   /* every base starts from missingVal (genodsp.c:1225-1240)       */
#define GDSP_CLEAR_BOTH        3   /* what read_intervals(clear=true) does                          */
#define GDSP_MASK_BINARIZE_FIRST 4 /* internal flag of gdsp_mask_intervals                            */
uint32_t gdsp_interval_tile (void);
int gdsp_bin_intervals   (uint32_t n, const uint32_t* h_start, const uint32_t* h_end, uint32_t count,
                          uint32_t* h_tileOffsets, uint32_t* h_tileList, uint64_t* listLen);
int gdsp_apply_intervals (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end,
                          const double* d_val, const uint32_t* d_tileOffsets, const uint32_t* d_tileList,
                          int overlapOp, int clear, double missingVal, void* stream);
int gdsp_scale_intervals (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end,
                          const double* d_val, const uint32_t* d_tileOffsets, const uint32_t* d_tileList,
                          int divide, double infinityVal, void* stream);

/* mask / masknot (mask.c:187-300, :483-640) and or / and (logical.c:439-560, :737-880):
 *   inside=1: bases under an interval become d_val[i]; inside=0: bases under NO interval become
 *   outsideVal (intervals sorted and non-overlapping, as for multiply);
 *   binarizeFirst: every nonzero base becomes 1.0 first (or, and).
 * minwith / maxwith (minmax.c:1893-2010, :2179-2294) are gdsp_apply_intervals with
 * GDSP_OVERLAP_MIN / GDSP_OVERLAP_MAX and clear=0. */
int gdsp_mask_intervals (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end,
                         const double* d_val, const uint32_t* d_tileOffsets, const uint32_t* d_tileList,
                         int inside, double outsideVal, int binarizeFirst, void* stream);

/* minover / maxover (minmax.c:193-390, :596-793): inside each sorted, non-overlapping interval only
 * the extreme survives, at the tied position nearest the interval's centre; everything else becomes
 * `fill`.  d_work >= gdsp_extreme_in_intervals_work(count) bytes. */
size_t gdsp_extreme_in_intervals_work (uint32_t count);
int gdsp_extreme_in_intervals (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end, uint32_t count,
                               const uint32_t* d_tileOffsets, const uint32_t* d_tileList,
                               int wantMax, double fill, void* d_work, void* stream);

/* ---- genodsp.c report_intervals:1561-1691 --------------------------------------- */

/* Run-length encode one chromosome on the device.  d_runs receives up to cap
 * (start,end) u32 pairs and d_vals the run values; *d_count the number of runs
 * found (may exceed cap: call again with more room).  uncovered: 0 hide, 1 show, -1 NA. */
size_t gdsp_report_runs_work (uint32_t n);
int gdsp_report_runs (const double* d_v, uint32_t n, int collapse, int uncovered,
                      uint32_t* d_runStart, uint32_t* d_runEnd, double* d_runVal, uint32_t cap,
                      uint32_t* d_count, void* d_work, void* stream);

/* ---- synthetic coverage signal for benchmarks and parity tests (not in the reference) */
int gdsp_synth_coverage (double* d_out, uint64_t seed, uint32_t chromIndex, uint32_t start,
                         uint32_t count, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GENODSP_HIP_H */
