/* gdsp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C99, of the reference algorithms on the hot path
 * (SURVEY.md section 8a).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (libgenodsp_hip.so, the host
 * driver) never links or calls it.
 *
 * Parity pin: every function here is checked bit-for-bit against the compiled,
 * unmodified reference (oracle/_ref/libgenodsp_ref.so, built by `make ref`)
 * by tests/test_oracle_vs_reference.py where that build exists, and against the
 * committed golden vectors in tests/golden/ (generated from that same build by
 * tests/golden/make_golden.py) everywhere else.
 */
#ifndef GDSP_ORACLE_H
#define GDSP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* interval overlap operators, values as genodsp_interface.h:157-159 */
#define ORC_OVERLAP_SUM 0
#define ORC_OVERLAP_MIN 1
#define ORC_OVERLAP_MAX 2

/* sum.c */
void orc_hann_window    (uint32_t W, double* w);
void orc_fir            (const double* v, uint32_t n, const double* w, uint32_t W, double* out);
void orc_smooth         (const double* v, uint32_t n, uint32_t W, double* out);
int  orc_smooth_threads (const double* const* vecs, const uint32_t* lens, double* const* outs, int nvec, uint32_t W,
                         int threads);   /* orc_smooth of several vectors over `threads` pthreads; -> threads started */
void orc_sliding_sum    (const double* v, uint32_t n, uint32_t W, double denom, double* out);
void orc_window_sum     (double* v, uint32_t n, uint32_t W, double denom, int useActual, double zeroVal);
void orc_cumulative_sum (double* v, uint32_t n);

/* minmax.c */
void orc_local_extrema  (const double* v, uint32_t n, uint32_t N, int wantMax, double fill, double* out);
void orc_best_extrema   (const double* v, uint32_t n, uint32_t W, int wantMax, double* out);

/* morphology.c */
void orc_dilate (double* v, uint32_t n, uint32_t left, uint32_t right, double T, double one, double zero);
void orc_erode  (double* v, uint32_t n, uint32_t left, uint32_t right, double T, double one, double zero);
void orc_close  (double* v, uint32_t n, double closingLength, double T, double one, double zero);
void orc_open   (double* v, uint32_t n, double openingLength, double T, double one, double zero);

/* logical.c, mask.c, add.c */
void orc_binarize     (double* v, uint32_t n, double T, int tiesAbove, double one, double zero);
void orc_clip         (double* v, uint32_t n, int haveMin, double minVal, int haveMax, double maxVal);
void orc_erase        (double* v, uint32_t n, int haveMin, double minVal, int haveMax, double maxVal,
                       int keepInside, double zero);
void orc_add_constant (double* v, uint32_t n, double c);
void orc_abs          (double* v, uint32_t n);
void orc_genome_minmax(const double* const* vecs, const uint32_t* lens, int nchrom, double* minOut, double* maxOut);
void orc_invert       (double* v, uint32_t n, double mid);
void orc_map          (double* v, uint32_t n, const double* kin, const double* kout, uint32_t nknots);   /* map.c */
void orc_clump        (double* v, uint32_t n, double average, uint32_t minLength, int above,
                       double one, double zero);                                                   /* clump.c */

/* percentile.c -- vecs/lens in the reference's processing order (longest first);
 * pThousandths[i] is the percentile in units of 0.001 %.  Returns the number of
 * sampled values (0 => nothing qualifies, outputs untouched).  Non-destructive. */
uint32_t orc_percentile (const double* const* vecs, const uint32_t* lens, int nchrom,
                         uint32_t window, double minAllowed, double maxAllowed,
                         const uint32_t* pThousandths, int np, double* out);

/* genodsp.c read_intervals / add.c / multiply.c, one chromosome's worth of
 * already-routed, already origin-shifted intervals [start,end) in file order */
void orc_fill            (double* v, uint32_t n, double val);
void orc_apply_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end,
                          const double* val, uint32_t count, int overlapOp, int clear, double missingVal);
void orc_scale_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end,
                          const double* val, uint32_t count, int divide, double infinityVal);

void orc_mask_intervals  (double* v, uint32_t n, const uint32_t* start, const uint32_t* end,
                          const double* val, uint32_t count, int inside, double outsideVal, int binarizeFirst);

void orc_extreme_in_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end,
                               uint32_t count, int wantMax, double fill);          /* minmax.c minover/maxover */

/* genodsp.c report_intervals: run-length encode one chromosome; returns the
 * number of runs written (at most cap).  uncovered: 0 hide, 1 show, -1 NA */
uint32_t orc_report_runs (const double* v, uint32_t n, int collapse, int uncovered,
                          uint32_t* runStart, uint32_t* runEnd, double* runVal, uint32_t cap);

/* synthetic coverage signal (ours, not the reference's): counter-based, so any
 * sub-range of any chromosome can be regenerated on either side.
 * mode 0: integer depth;  mode 1: depth * U(0.5,1.5) */
void orc_synth_coverage (uint64_t seed, uint32_t chromIndex, uint32_t start, uint32_t count,
                         int mode, double* out);

#ifdef __cplusplus
}
#endif
#endif
