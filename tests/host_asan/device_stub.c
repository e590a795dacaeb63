/* device_stub.c -- TEST INFRASTRUCTURE ONLY: stands in for libgenodsp_hip.so so that the HOST side of the driver
 * (genodsp_hip.c, ingest.c, utilities.c, the operators' parse functions: everything that reads untrusted text) can run
 * on a CPU under -fsanitize=address,undefined (tests/test_host_sanitizers.py, `make -C tests/host_asan`).
 *
 * "Device memory" is plain malloc, so the sanitizer also sees every staging copy the driver makes.  The few entry
 * points a text-in / text-out run needs are answered by the oracle (oracle/gdsp_oracle.c, the tests' checker) or by a
 * few lines here; every other entry point stops the program: this is not a compute path and is never installed.
 * The product binary (genodsp_amd/genodsp_hip) links the HIP library and nothing of this.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/gdsp_oracle.h"

#define OK 0
#define IV_TILE 1024u
typedef struct item { const double* in;  double* out;  uint32_t n; } item;

const char* gdsp_last_error (void) { return "host-only sanitizer build"; }
const char* gdsp_version    (void) { return "host-only sanitizer build (no GPU code)"; }
int gdsp_poison (double* value) { if (value != NULL) *value = 0;  return 0; }

int gdsp_device_count (int* count) { *count = 1;  return OK; }
int gdsp_set_device (int d) { (void) d;  return OK; }
int gdsp_get_device (int* d) { *d = 0;  return OK; }
int gdsp_malloc (void** p, size_t bytes) { *p = malloc (bytes? bytes : 1);  return (*p == NULL)? 3 : OK; }
int gdsp_free (void* p) { free (p);  return OK; }
int gdsp_host_alloc (void** p, size_t bytes) { return gdsp_malloc (p, bytes); }
int gdsp_host_free (void* p) { free (p);  return OK; }
int gdsp_memcpy_h2d (void* d, const void* s, size_t n, void* st) { (void) st;  memcpy (d, s, n);  return OK; }
int gdsp_memcpy_d2h (void* d, const void* s, size_t n, void* st) { (void) st;  memcpy (d, s, n);  return OK; }
int gdsp_memcpy_d2d (void* d, const void* s, size_t n, void* st) { (void) st;  memmove (d, s, n);  return OK; }
int gdsp_memcpy_peer (void* d, int dd, const void* s, int sd, size_t n, void* st) { (void) dd;  (void) sd;  (void) st;  memmove (d, s, n);  return OK; }
int gdsp_memset (void* d, int b, size_t n, void* st) { (void) st;  memset (d, b, n);  return OK; }
int gdsp_stream_create (void** s) { *s = malloc (1);  return OK; }
int gdsp_stream_destroy (void* s) { free (s);  return OK; }
int gdsp_stream_sync (void* s) { (void) s;  return OK; }
int gdsp_device_sync (void) { return OK; }
int gdsp_event_create (void** e) { *e = malloc (1);  return OK; }
int gdsp_event_destroy (void* e) { free (e);  return OK; }
int gdsp_event_record (void* e, void* s) { (void) e;  (void) s;  return OK; }
int gdsp_stream_wait_event (void* s, void* e) { (void) e;  (void) s;  return OK; }
int gdsp_event_elapsed_ms (void* a, void* b, float* ms) { (void) a;  (void) b;  *ms = 0;  return OK; }

int gdsp_fill (double* v, uint32_t n, double val, void* st) { (void) st;  orc_fill (v, n, val);  return OK; }

/* the interval protocol of include/genodsp_hip.h: indices binned per 1024-base tile in file order, applied per base in
 * that order */
uint32_t gdsp_interval_tile (void) { return IV_TILE; }
int gdsp_bin_intervals (uint32_t n, const uint32_t* start, const uint32_t* end, uint32_t count, uint32_t* off, uint32_t* list, uint64_t* listLen)
	{
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + IV_TILE - 1) / IV_TILE);
	uint64_t total = 0;
	for (uint32_t t=0 ; t<=ntiles ; t++) off[t] = 0;
	for (uint32_t i=0 ; i<count ; i++)
		{
		uint32_t s = start[i], e = (end[i] > n)? n : end[i];
		if (s >= e) continue;
		for (uint32_t t=s/IV_TILE ; t<=(e-1)/IV_TILE ; t++) { off[t+1]++;  total++; }
		}
	*listLen = total;
	for (uint32_t t=0 ; t<ntiles ; t++) off[t+1] += off[t];
	if (list == NULL) return OK;
	for (uint32_t i=0 ; i<count ; i++)
		{
		uint32_t s = start[i], e = (end[i] > n)? n : end[i];
		if (s >= e) continue;
		for (uint32_t t=s/IV_TILE ; t<=(e-1)/IV_TILE ; t++) list[off[t]++] = i;
		}
	for (uint32_t t=ntiles ; t>0 ; t--) off[t] = off[t-1];
	off[0] = 0;
	return OK;
	}

int gdsp_apply_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end, const double* val,
                          const uint32_t* off, const uint32_t* list, int overlapOp, int clear, double missingVal, void* st)
	{
	(void) st;
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + IV_TILE - 1) / IV_TILE);
	if (clear & 2) orc_fill (v, n, missingVal);
	for (uint32_t t=0 ; t<ntiles ; t++)
		{
		const uint32_t lo = t * IV_TILE, hi = (n - lo < IV_TILE)? n : lo + IV_TILE;
		for (uint32_t j=off[t] ; j<off[t+1] ; j++)
			{
			const uint32_t i = list[j];
			uint32_t s = (start[i] > lo)? start[i] : lo, e = (end[i] < hi)? end[i] : hi;
			/* one interval, one tile: the oracle's loop over that stretch (its own bounds are relative to the pointer) */
			if (s < e)
				{
				uint32_t zero = 0, len = e - s;
				orc_apply_intervals (v + s, len, &zero, &len, &val[i], 1, overlapOp, clear & 1, missingVal);
				}
			}
		}
	return OK;
	}

size_t gdsp_report_runs_work (uint32_t n) { (void) n;  return 16; }
int gdsp_report_runs (const double* v, uint32_t n, int collapse, int uncovered, uint32_t* rs, uint32_t* re, double* rv, uint32_t cap,
                      uint32_t* count, void* work, void* st)
	{
	(void) work;  (void) st;
	uint32_t* s = (uint32_t*) malloc (((size_t) n + 1) * sizeof(uint32_t));
	uint32_t* e = (uint32_t*) malloc (((size_t) n + 1) * sizeof(uint32_t));
	double*   x = (double*)   malloc (((size_t) n + 1) * sizeof(double));
	uint32_t  k = orc_report_runs (v, n, collapse, uncovered, s, e, x, n + 1);
	for (uint32_t i=0 ; (i<k) && (i<cap) ; i++) { rs[i] = s[i];  re[i] = e[i];  rv[i] = x[i]; }
	*count = k;
	free (s);  free (e);  free (x);
	return OK;
	}

/* ---- the operators, answered by the oracle (so that whole command lines, and all of the committed CLI fixtures, run
 * through the sanitized host code: parse functions, apply shims, named variables, scratch bookkeeping) ---- */
#define EINVAL_ 1
static void copy_out (double* out, const double* tmp, uint32_t n) { memcpy (out, tmp, (size_t) n * sizeof(double)); }
static double* dup_in (const double* in, uint32_t n)
	{ double* t = (double*) malloc (((size_t) n + 1) * sizeof(double));  memcpy (t, in, (size_t) n * sizeof(double));  return t; }

int gdsp_smooth (const double* in, double* out, uint32_t n, uint32_t W, int mode, void* st)
	{ (void) mode;  (void) st;  if ((W < 3) || !(W & 1) || (W > 50001)) return EINVAL_;  if (n) orc_smooth (in, n, W, out);  return OK; }
int gdsp_smooth_batch (const item* it, int k, uint32_t W, int mode, void* st)
	{ for (int i=0 ; i<k ; i++) { int rc = gdsp_smooth (it[i].in, it[i].out, it[i].n, W, mode, st);  if (rc) return rc; }  return OK; }
int gdsp_smooth_local_extrema_fusable (uint32_t W, uint32_t N) { (void) W;  (void) N;  return 0; }
int gdsp_dilate_erode_fusable (uint32_t a, uint32_t b, uint32_t c, uint32_t d) { (void) a;  (void) b;  (void) c;  (void) d;  return 0; }
int gdsp_sliding_sum (const double* in, double* out, uint32_t n, uint32_t W, double denom, void* st)
	{ (void) st;  if (n) orc_sliding_sum (in, n, W, denom, out);  return OK; }
int gdsp_sliding_sum_any (const double* in, double* out, uint32_t n, uint32_t W, double denom, void* w, size_t wb, void* st)
	{ (void) w;  (void) wb;  return gdsp_sliding_sum (in, out, n, W, denom, st); }
int gdsp_window_sum (double* v, uint32_t n, uint32_t W, double denom, int useActual, double zeroVal, void* st)
	{ (void) st;  if (n) orc_window_sum (v, n, W, denom, useActual, zeroVal);  return OK; }
int gdsp_cumulative_sum (double* v, uint32_t n, void* w, void* st) { (void) w;  (void) st;  if (n) orc_cumulative_sum (v, n);  return OK; }
int gdsp_local_extrema (const double* in, double* out, uint32_t n, uint32_t N, int wantMax, double fill, void* st)
	{ (void) st;  if (n) orc_local_extrema (in, n, N, wantMax, fill, out);  return OK; }
int gdsp_local_extrema_any (const double* in, double* out, uint32_t n, uint32_t N, int wantMax, double fill, void* w, size_t wb, void* st)
	{ (void) w;  (void) wb;  return gdsp_local_extrema (in, out, n, N, wantMax, fill, st); }
int gdsp_local_extrema_batch (const item* it, int k, uint32_t N, int wantMax, double fill, void* st)
	{ for (int i=0 ; i<k ; i++) gdsp_local_extrema (it[i].in, it[i].out, it[i].n, N, wantMax, fill, st);  return OK; }
int gdsp_best_extrema (const double* in, double* out, uint32_t n, uint32_t W, int wantMax, void* st)
	{ (void) st;  if (n) orc_best_extrema (in, n, W, wantMax, out);  return OK; }
int gdsp_best_extrema_any (const double* in, double* out, uint32_t n, uint32_t W, int wantMax, void* w, size_t wb, void* st)
	{ (void) w;  (void) wb;  return gdsp_best_extrema (in, out, n, W, wantMax, st); }
int gdsp_best_extrema_batch (const item* it, int k, uint32_t W, int wantMax, void* st)
	{ for (int i=0 ; i<k ; i++) gdsp_best_extrema (it[i].in, it[i].out, it[i].n, W, wantMax, st);  return OK; }

#define MORPH(name, call)                                                                                             \
int gdsp_##name (const double* in, double* out, uint32_t n, uint32_t left, uint32_t right, double T, double one, double zero, void* st) \
	{ (void) st;  if (n == 0) return OK;  double* t = dup_in (in, n);  call;  copy_out (out, t, n);  free (t);  return OK; }   \
int gdsp_##name##_any (const double* in, double* out, uint32_t n, uint32_t left, uint32_t right, double T, double one, double zero, void* w, size_t wb, void* st) \
	{ (void) w;  (void) wb;  return gdsp_##name (in, out, n, left, right, T, one, zero, st); }                         \
int gdsp_##name##_batch (const item* it, int k, uint32_t left, uint32_t right, double T, double one, double zero, void* st) \
	{ for (int i=0 ; i<k ; i++) gdsp_##name (it[i].in, it[i].out, it[i].n, left, right, T, one, zero, st);  return OK; }
MORPH (dilate, orc_dilate (t, n, left, right, T, one, zero))
MORPH (erode,  orc_erode  (t, n, left, right, T, one, zero))
#define RUNS(name, call)                                                                                              \
int gdsp_##name (const double* in, double* out, uint32_t n, double length, double T, double one, double zero, void* st)  \
	{ (void) st;  if (n == 0) return OK;  double* t = dup_in (in, n);  call;  copy_out (out, t, n);  free (t);  return OK; }   \
int gdsp_##name##_any (const double* in, double* out, uint32_t n, double length, double T, double one, double zero, void* w, size_t wb, void* st) \
	{ (void) w;  (void) wb;  return gdsp_##name (in, out, n, length, T, one, zero, st); }
RUNS (close, orc_close (t, n, length, T, one, zero))
RUNS (open,  orc_open  (t, n, length, T, one, zero))

int gdsp_binarize (double* v, uint32_t n, double T, int ties, double one, double zero, void* st) { (void) st;  orc_binarize (v, n, T, ties, one, zero);  return OK; }
int gdsp_clip (double* v, uint32_t n, int hMin, double lo, int hMax, double hi, void* st) { (void) st;  orc_clip (v, n, hMin, lo, hMax, hi);  return OK; }
int gdsp_erase (double* v, uint32_t n, int hMin, double lo, int hMax, double hi, int keepInside, double zero, void* st)
	{ (void) st;  orc_erase (v, n, hMin, lo, hMax, hi, keepInside, zero);  return OK; }
int gdsp_add_constant (double* v, uint32_t n, double c, void* st) { (void) st;  if (c != 0.0) orc_add_constant (v, n, c);  return OK; }
int gdsp_abs (double* v, uint32_t n, void* st) { (void) st;  orc_abs (v, n);  return OK; }
int gdsp_invert (double* v, uint32_t n, double mid, void* st) { (void) st;  orc_invert (v, n, mid);  return OK; }
int gdsp_map (double* v, uint32_t n, const double* kin, const double* kout, uint32_t nk, void* st) { (void) st;  orc_map (v, n, kin, kout, nk);  return OK; }
int gdsp_clump (double* v, uint32_t n, double avg, uint32_t minLen, int above, double one, double zero, void* w, void* st)
	{ (void) w;  (void) st;  if (n) orc_clump (v, n, avg, minLen, above, one, zero);  return OK; }
int gdsp_binarize_batch (const item* it, int k, double T, int ties, double one, double zero, void* st)
	{ for (int i=0 ; i<k ; i++) gdsp_binarize (it[i].out, it[i].n, T, ties, one, zero, st);  return OK; }
int gdsp_clip_batch (const item* it, int k, int hMin, double lo, int hMax, double hi, void* st)
	{ for (int i=0 ; i<k ; i++) gdsp_clip (it[i].out, it[i].n, hMin, lo, hMax, hi, st);  return OK; }
int gdsp_erase_batch (const item* it, int k, int hMin, double lo, int hMax, double hi, int keepInside, double zero, void* st)
	{ for (int i=0 ; i<k ; i++) gdsp_erase (it[i].out, it[i].n, hMin, lo, hMax, hi, keepInside, zero, st);  return OK; }
int gdsp_add_constant_batch (const item* it, int k, double c, void* st)
	{ for (int i=0 ; i<k ; i++) gdsp_add_constant (it[i].out, it[i].n, c, st);  return OK; }
int gdsp_abs_batch (const item* it, int k, void* st)
	{ for (int i=0 ; i<k ; i++) gdsp_abs (it[i].out, it[i].n, st);  return OK; }
size_t gdsp_long_window_work (uint32_t n) { (void) n;  return 16; }
size_t gdsp_clump_work (uint32_t n) { (void) n;  return 16; }
size_t gdsp_cumulative_sum_work (uint32_t n) { (void) n;  return 16; }

int gdsp_minmax_init (double* acc, void* st) { (void) st;  acc[0] = 1.7976931348623157e308;  acc[1] = -1.7976931348623157e308;  acc[2] = 0;  return OK; }
int gdsp_minmax_update (const double* v, uint32_t n, uint32_t window, double lo, double hi, double* acc, void* st)
	{
	(void) st;
	for (uint32_t i=0 ; i<n ; i+=window)
		{
		if ((v[i] < lo) || (v[i] > hi)) continue;
		if (v[i] < acc[0]) acc[0] = v[i];
		if (v[i] > acc[1]) acc[1] = v[i];
		acc[2] += 1;
		}
	return OK;
	}

typedef struct source { const double* v;  uint32_t n;  int device;  void* stream; } source;
int gdsp_percentiles (const source* src, int nsrc, uint32_t window, double lo, double hi, const uint32_t* pts, int npts,
                      int strategy, uint32_t sampleTarget, void* reduce, void* ctx, double* values, uint64_t* count)
	{
	(void) strategy;  (void) sampleTarget;  (void) reduce;  (void) ctx;
	const double** vecs = (const double**) calloc (nsrc? nsrc : 1, sizeof(double*));
	uint32_t*      lens = (uint32_t*) calloc (nsrc? nsrc : 1, sizeof(uint32_t));
	int k = 0;
	for (int i=0 ; i<nsrc ; i++) { if (src[i].n != 0) { vecs[k] = src[i].v;  lens[k] = src[i].n;  k++; } }
	*count = orc_percentile (vecs, lens, k, window, lo, hi, pts, npts, values);
	free (vecs);  free (lens);
	return OK;
	}
typedef struct fusebin { int which;  int tiesAbove;  double one, zero;  double* const* out; } fusebin;
int gdsp_percentiles_binarize (const source* src, int nsrc, uint32_t window, double lo, double hi, const uint32_t* pts, int npts,
                               int strategy, uint32_t sampleTarget, void* reduce, void* ctx, double* values, uint64_t* count,
                               const fusebin* fuse, int* onePass)
	{
	gdsp_percentiles (src, nsrc, window, lo, hi, pts, npts, strategy, sampleTarget, reduce, ctx, values, count);
	*onePass = 0;
	if (*count == 0) return OK;
	for (int i=0 ; i<nsrc ; i++)
		{
		if (src[i].n == 0) continue;
		memcpy (fuse->out[i], src[i].v, (size_t) src[i].n * sizeof(double));
		orc_binarize (fuse->out[i], src[i].n, values[fuse->which], fuse->tiesAbove, fuse->one, fuse->zero);
		}
	return OK;
	}
int gdsp_percentiles_use_comm (void* c) { (void) c;  return OK; }
void gdsp_percentiles_stats (uint64_t out[8]) { memset (out, 0, 8 * sizeof(uint64_t)); }

/* the interval-file operators that walk sorted intervals: the CSR lists them tile by tile in file order */
static uint32_t gather (uint32_t n, const uint32_t* start, const uint32_t* end, const double* val, const uint32_t* off, const uint32_t* list,
                        uint32_t** s, uint32_t** e, double** x)
	{
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + IV_TILE - 1) / IV_TILE);
	const uint32_t total  = off[ntiles];
	*s = (uint32_t*) malloc (((size_t) total + 1) * sizeof(uint32_t));
	*e = (uint32_t*) malloc (((size_t) total + 1) * sizeof(uint32_t));
	*x = (double*)   malloc (((size_t) total + 1) * sizeof(double));
	uint32_t k = 0, last = 0xFFFFFFFFu;
	for (uint32_t j=0 ; j<total ; j++)
		{
		if (list[j] == last) continue;                         /* (an interval spanning tiles is listed once per tile) */
		last = list[j];
		(*s)[k] = start[last];  (*e)[k] = end[last];  (*x)[k] = val? val[last] : 1.0;  k++;
		}
	return k;
	}
int gdsp_scale_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end, const double* val, const uint32_t* off,
                          const uint32_t* list, int divide, double infinityVal, void* st)
	{
	(void) st;
	uint32_t *s, *e;  double* x;
	uint32_t k = gather (n, start, end, val, off, list, &s, &e, &x);
	orc_scale_intervals (v, n, s, e, x, k, divide, infinityVal);
	free (s);  free (e);  free (x);
	return OK;
	}
int gdsp_mask_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end, const double* val, const uint32_t* off,
                         const uint32_t* list, int inside, double outsideVal, int binarizeFirst, void* st)
	{
	(void) st;
	uint32_t *s, *e;  double* x;
	uint32_t k = gather (n, start, end, val, off, list, &s, &e, &x);
	orc_mask_intervals (v, n, s, e, x, k, inside, outsideVal, binarizeFirst);
	free (s);  free (e);  free (x);
	return OK;
	}
size_t gdsp_extreme_in_intervals_work (uint32_t n) { (void) n;  return 16; }
int gdsp_extreme_in_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end, uint32_t count, const uint32_t* off,
                               const uint32_t* list, int wantMax, double fill, void* w, void* st)
	{ (void) off;  (void) list;  (void) w;  (void) st;  orc_extreme_in_intervals (v, n, start, end, count, wantMax, fill);  return OK; }

/* everything else: not in the host-only build */
#define NOT_HERE(name) int name (void) { fprintf (stderr, "[host-only sanitizer build] " #name " needs the GPU library\n");  exit (97); }
NOT_HERE (gdsp_smooth_local_extrema) NOT_HERE (gdsp_smooth_local_extrema_batch) NOT_HERE (gdsp_dilate_erode) NOT_HERE (gdsp_dilate_erode_batch)
NOT_HERE (gdsp_comm_create) NOT_HERE (gdsp_comm_destroy) NOT_HERE (gdsp_comm_rccl_version) NOT_HERE (gdsp_comm_allreduce_f64)
