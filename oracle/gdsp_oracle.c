/* gdsp_oracle.c -- TEST INFRASTRUCTURE ONLY (see gdsp_oracle.h).
 *
 * Scalar CPU restatement of the reference's hot-path algorithms.  Each function
 * names the reference lines it follows.  Built with -ffp-contract=off so that
 * every multiply and add rounds on its own, as in the reference's stock build
 * (x86-64 baseline has no FMA).
 *
 * Where the reference has undefined behaviour (smooth on vectors no longer than
 * the half window, erode on runs ending left of the erosion length) the
 * restatement uses the zero-padded definition the reference's own usage text
 * states ("values beyond the ends of the vector are considered to be zero",
 * sum.c:509-511); tests keep away from those inputs when comparing with the
 * reference itself.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include "gdsp_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846264   /* genodsp_interface.h:133-135 */
#endif

/* ---------------------------------------------------------------- sum.c -- */

/* Hann taps, sum.c:632-645: mirrored store of (1-cos(2*pi*x))/2 with
 * x=(k+1)/(W+1) for k=0..hOff, ascending sum, then each tap divided by it. */
void orc_hann_window (uint32_t W, double* w)
	{
	uint32_t hOff = (W - 1) / 2, k;
	double   x, total;

	for (k=0 ; k<=hOff ; k++)
		{
		x = (k+1) / (double) (W+1);
		w[k] = w[W-1-k] = (1 - cos (2*M_PI*x)) / 2;
		}
	total = 0.0;
	for (k=0 ; k<W ; k++) total += w[k];
	for (k=0 ; k<W ; k++) w[k] /= total;
	}

/* Direct W-tap FIR, sum.c:651-663: for each output the taps that fall inside
 * [0,n) are accumulated in ascending tap order, multiply then add, from +0.0.
 * Signed 64-bit bounds replace the reference's u32 arithmetic, so the result
 * is the zero-padded one for every n (the reference agrees when n > hOff). */
void orc_fir (const double* v, uint32_t n, const double* w, uint32_t W, double* out)
	{
	int64_t hOff = (W - 1) / 2, ix, k, kLo, kHi;
	double  acc;

	for (ix=0 ; ix<(int64_t)n ; ix++)
		{
		kLo = (ix < hOff)? hOff-ix : 0;
		kHi = (int64_t) W - 1;
		if (ix - hOff + kHi > (int64_t) n - 1) kHi = (int64_t) n - 1 + hOff - ix;
		acc = 0.0;
		for (k=kLo ; k<=kHi ; k++)
			acc += w[k] * v[ix-hOff+k];
		out[ix] = acc;
		}
	}

void orc_smooth (const double* v, uint32_t n, uint32_t W, double* out)
	{
	double* w = (double*) malloc ((size_t) W * sizeof(double));
	orc_hann_window (W, w);
	orc_fir (v, n, w, W, out);
	free (w);
	}

/* The same loop over every host core (bench.py's cpu_baseline_all_cores: the reference itself is single-threaded,
 * chromosomes and stretches of them are independent for this operator, genodsp.c:909-921): thread t of T evaluates
 * outputs [n t/T, n (t+1)/T) of every vector, reading its neighbours' inputs in place, writing into out[] that the
 * caller has allocated AND touched (first-touch page faults are not smoothing).  Same arithmetic, same bits. */
#include <pthread.h>
typedef struct { const double* const* vecs;  const uint32_t* lens;  double* const* outs;  int nvec;  const double* w;
                 uint32_t W;  int t, T; } smooth_job;

static void* smooth_worker (void* arg)
	{
	const smooth_job* j = (const smooth_job*) arg;
	int64_t hOff = (j->W - 1) / 2, ix, k, kLo, kHi;
	int     c;
	for (c=0 ; c<j->nvec ; c++)
		{
		const double* v = j->vecs[c];
		const int64_t n = j->lens[c], a = n * j->t / j->T, b = n * (j->t + 1) / j->T;
		for (ix=a ; ix<b ; ix++)
			{
			double acc = 0.0;
			kLo = (ix < hOff)? hOff-ix : 0;
			kHi = (int64_t) j->W - 1;
			if (ix - hOff + kHi > n - 1) kHi = n - 1 + hOff - ix;
			for (k=kLo ; k<=kHi ; k++)
				acc += j->w[k] * v[ix-hOff+k];
			j->outs[c][ix] = acc;
			}
		}
	return NULL;
	}

int orc_smooth_threads (const double* const* vecs, const uint32_t* lens, double* const* outs, int nvec, uint32_t W, int threads)
	{
	double*     w    = (double*) malloc ((size_t) W * sizeof(double));
	pthread_t*  tid  = (pthread_t*) malloc ((size_t) threads * sizeof(pthread_t));
	smooth_job* jobs = (smooth_job*) malloc ((size_t) threads * sizeof(smooth_job));
	int t, started = 0;
	orc_hann_window (W, w);
	for (t=0 ; t<threads ; t++)
		{
		smooth_job j = { vecs, lens, outs, nvec, w, W, t, threads };
		jobs[t] = j;
		if (pthread_create (&tid[t], NULL, smooth_worker, &jobs[t]) != 0) break;
		started++;
		}
	for (t=started ; t<threads ; t++) smooth_worker (&jobs[t]);        /* (threads that could not be started: done here) */
	for (t=0 ; t<started ; t++) pthread_join (tid[t], NULL);
	free (w);  free (tid);  free (jobs);
	return started;
	}

/* Running window sum, sum.c:436-460: one accumulator walks the vector, adding
 * the entering value and subtracting the leaving one; the sum after step ix is
 * the output for centre ix-hOff; a final pass divides by the denominator. */
void orc_sliding_sum (const double* v, uint32_t n, uint32_t W, double denom, double* out)
	{
	uint64_t hOff = (W - 1) / 2, ix;
	double   acc = 0.0;

	for (ix=0 ; ix<(uint64_t)n+hOff ; ix++)
		{
		if (ix < n)  acc += v[ix];
		if (ix >= W) acc -= v[ix-W];
		if (ix >= hOff) out[ix-hOff] = acc;
		}
	for (ix=0 ; ix<n ; ix++) out[ix] /= denom;
	}

/* Non-overlapping window sums, sum.c:225-250: the window's first slot gets
 * (ascending sum)/denominator, the other slots the zero value; the last window
 * may be short, and "actual" divides by the true window length. */
void orc_window_sum (double* v, uint32_t n, uint32_t W, double denom, int useActual, double zeroVal)
	{
	uint64_t s, e, ix;
	double   acc;

	for (s=0 ; s<n ; s+=W)
		{
		e = s + W;  if (e > n) e = n;
		acc = v[s];
		for (ix=s+1 ; ix<e ; ix++) acc += v[ix];
		v[s] = useActual? acc / (double) (e-s) : acc / denom;
		for (ix=s+1 ; ix<e ; ix++) v[ix] = zeroVal;
		}
	}

/* Inclusive prefix sum, sum.c:785-790 */
void orc_cumulative_sum (double* v, uint32_t n)
	{
	uint32_t ix;
	double   acc = 0.0;
	for (ix=0 ; ix<n ; ix++) { acc += v[ix];  v[ix] = acc; }
	}

/* ------------------------------------------------------------- minmax.c -- */

/* localmax / localmin, minmax.c:1201-1221 and :999-1016: v[ix] survives unless
 * some OTHER position within +-hOff (clamped to the vector) is strictly
 * greater (less); otherwise the fill value replaces it. */
void orc_local_extrema (const double* v, uint32_t n, uint32_t N, int wantMax, double fill, double* out)
	{
	int64_t hOff = (N - 1) / 2, ix, j, lo, hi;
	double  val;

	for (ix=0 ; ix<(int64_t)n ; ix++)
		{
		lo = ix - hOff;  if (lo < 0) lo = 0;
		hi = ix + hOff;  if (hi > (int64_t) n - 1) hi = (int64_t) n - 1;
		val = v[ix];
		for (j=lo ; j<=hi ; j++)
			{
			if (j == ix) continue;
			if (wantMax? (v[j] > val) : (v[j] < val)) { val = fill;  break; }
			}
		out[ix] = val;
		}
	}

/* bestmax / bestmin, minmax.c:1634-1712 and :1387-1465: extreme over the
 * window [ix-wLft, ix+wRgt] (clamped), wLft=(W-1)/2, wRgt=W-1-wLft, carried
 * from ix-1 by the reference's three cases: the entering value ties or beats
 * the old extreme; the leaving value was not the extreme; else rescan. */
void orc_best_extrema (const double* v, uint32_t n, uint32_t W, int wantMax, double* out)
	{
	int64_t wLft = (W - 1) / 2, wRgt = (int64_t) (W - 1) - wLft;
	int64_t ix, j, enter, leave, lo, hi;
	double  best;

	best = v[0];
	for (j=1 ; j<=wRgt && j<(int64_t)n ; j++)
		{ if (wantMax? (v[j] > best) : (v[j] < best)) best = v[j]; }
	out[0] = best;

	for (ix=1 ; ix<(int64_t)n ; ix++)
		{
		leave = ix - (wLft+1);                             /* <0 => nothing leaves   */
		enter = (ix + wRgt >= (int64_t) n)? -1 : ix + wRgt; /* -1 => nothing enters   */

		if ((enter >= 0) && (wantMax? (v[enter] >= best) : (v[enter] <= best)))
			best = v[enter];
		else if ((leave < 0) || (wantMax? (v[leave] < best) : (v[leave] > best)))
			;
		else
			{
			lo = (leave < 0)? 0 : leave+1;
			hi = (enter < 0)? (int64_t) n - 1 : enter;
			best = v[lo];
			for (j=lo+1 ; j<=hi ; j++)
				{ if (wantMax? (v[j] > best) : (v[j] < best)) best = v[j]; }
			}
		out[ix] = best;
		}
	}

/* --------------------------------------------------------- morphology.c -- */

static void orc_span (double* v, uint64_t s, uint64_t e, double val)
	{ uint64_t i;  for (i=s ; i<e ; i++) v[i] = val; }

/* dilate, morphology.c:928-1062.  Positions in the set become `one` at once;
 * each maximal gap [g,e) is then narrowed from both sides: its first `right`
 * positions (if it has a left neighbour) and last `left` positions (if it has
 * a right neighbour) become `one`, the rest `zero`.  Membership follows the
 * reference's tests exactly: `v[0] > T` for position 0 (:930) but
 * `!(v[ix] <= T)` for ix>=1 (:935), which differ only for NaN. */
void orc_dilate (double* v, uint32_t n, uint32_t left, uint32_t right, double T, double one, double zero)
	{
	uint64_t ix = 0, g, e, rightIx, leftIx;
	int      in;

	while (ix < n)
		{
		in = (ix == 0)? (v[ix] > T) : !(v[ix] <= T);
		if (in) { v[ix++] = one;  continue; }

		g = ix;
		while ((ix < n) && !((ix == 0)? (v[ix] > T) : !(v[ix] <= T))) ix++;
		e = ix;

		if ((g == 0) && (e == n)) orc_span (v, 0, n, zero);
		else if (g == 0)
			{
			leftIx = (e <= left)? 0 : e - left;
			orc_span (v, 0, leftIx, zero);  orc_span (v, leftIx, e, one);
			}
		else if (e == n)
			{
			rightIx = g + right;
			if (rightIx >= n) orc_span (v, g, n, one);
			else { orc_span (v, g, rightIx, one);  orc_span (v, rightIx, n, zero); }
			}
		else
			{
			rightIx = g + right;
			leftIx  = (e <= left)? 0 : e - left;
			if (rightIx >= leftIx) orc_span (v, g, e, one);
			else { orc_span (v, g, rightIx, one);  orc_span (v, rightIx, leftIx, zero);  orc_span (v, leftIx, e, one); }
			}
		}
	}

/* erode, morphology.c:1381-1443.  Gap positions become `zero`; each maximal
 * run [s,e) keeps only [s+right, e-left) as `one`.  Both vector ends count as
 * gaps.  The reference computes e-left in u32 and misbehaves when e < left
 * (SURVEY Appendix B #1); here that run is simply erased. */
void orc_erode (double* v, uint32_t n, uint32_t left, uint32_t right, double T, double one, double zero)
	{
	uint64_t ix = 0, s, e, keepLo, keepHi;

	while (ix < n)
		{
		if (!(v[ix] > T)) { v[ix++] = zero;  continue; }
		s = ix;
		while ((ix < n) && (v[ix] > T)) ix++;
		e = ix;
		keepLo = s + right;
		keepHi = (e >= left)? e - left : 0;
		if (keepLo >= keepHi) orc_span (v, s, e, zero);
		else { orc_span (v, s, keepLo, zero);  orc_span (v, keepLo, keepHi, one);  orc_span (v, keepHi, e, zero); }
		}
	}

/* close, morphology.c:265-309.  Set positions become `one`; a gap is filled
 * with `one` only if it touches neither end of the vector and is no longer
 * than closingLength (compared as a double, :283). */
void orc_close (double* v, uint32_t n, double closingLength, double T, double one, double zero)
	{
	uint64_t ix = 0, g, e;

	while (ix < n)
		{
		if (!(v[ix] <= T)) { v[ix++] = one;  continue; }
		g = ix;
		while ((ix < n) && (v[ix] <= T)) ix++;
		e = ix;
		if ((g == 0) || (e == n) || ((double) (e - g) > closingLength)) orc_span (v, g, e, zero);
		else                                                            orc_span (v, g, e, one);
		}
	}

/* open, morphology.c:563-595.  Gap positions become `zero`; a run survives as
 * `one` only if it is longer than openingLength. */
void orc_open (double* v, uint32_t n, double openingLength, double T, double one, double zero)
	{
	uint64_t ix = 0, s, e;

	while (ix < n)
		{
		if (!(v[ix] > T)) { v[ix++] = zero;  continue; }
		s = ix;
		while ((ix < n) && (v[ix] > T)) ix++;
		e = ix;
		orc_span (v, s, e, ((double) (e - s) > openingLength)? one : zero);
		}
	}

/* ------------------------------------------- logical.c, mask.c, add.c ---- */

/* logical.c:247-257 */
void orc_binarize (double* v, uint32_t n, double T, int tiesAbove, double one, double zero)
	{
	uint32_t ix;
	if (tiesAbove) for (ix=0 ; ix<n ; ix++) v[ix] = (v[ix] >= T)? one : zero;
	else           for (ix=0 ; ix<n ; ix++) v[ix] = (v[ix] >  T)? one : zero;
	}

/* mask.c:893-911 */
void orc_clip (double* v, uint32_t n, int haveMin, double minVal, int haveMax, double maxVal)
	{
	uint32_t ix;
	if (!haveMax)      { for (ix=0 ; ix<n ; ix++) if (v[ix] < minVal) v[ix] = minVal; }
	else if (!haveMin) { for (ix=0 ; ix<n ; ix++) if (v[ix] > maxVal) v[ix] = maxVal; }
	else for (ix=0 ; ix<n ; ix++)
		{
		if      (v[ix] < minVal) v[ix] = minVal;
		else if (v[ix] > maxVal) v[ix] = maxVal;
		}
	}

/* mask.c:1187-1227 */
void orc_erase (double* v, uint32_t n, int haveMin, double minVal, int haveMax, double maxVal,
                int keepInside, double zero)
	{
	uint32_t ix;
	if (keepInside)
		{
		if (!haveMax)      { for (ix=0 ; ix<n ; ix++) if (v[ix] < minVal) v[ix] = zero; }
		else if (!haveMin) { for (ix=0 ; ix<n ; ix++) if (v[ix] > maxVal) v[ix] = zero; }
		else               { for (ix=0 ; ix<n ; ix++) if ((v[ix] < minVal) || (v[ix] > maxVal)) v[ix] = zero; }
		}
	else
		{
		if (!haveMax)      { for (ix=0 ; ix<n ; ix++) if (v[ix] >= minVal) v[ix] = zero; }
		else if (!haveMin) { for (ix=0 ; ix<n ; ix++) if (v[ix] <= maxVal) v[ix] = zero; }
		else               { for (ix=0 ; ix<n ; ix++) if ((v[ix] >= minVal) && (v[ix] <= maxVal)) v[ix] = zero; }
		}
	}

/* add.c:736-739 (a zero constant leaves the vector untouched) */
void orc_add_constant (double* v, uint32_t n, double c)
	{
	uint32_t ix;
	if (c == 0.0) return;
	for (ix=0 ; ix<n ; ix++) v[ix] += c;
	}

/* add.c:1046-1047 */
void orc_abs (double* v, uint32_t n)
	{
	uint32_t ix;
	for (ix=0 ; ix<n ; ix++) { if (v[ix] < 0) v[ix] = -v[ix]; }
	}

/* map.c:194-381, with the piece found by the reference's binary search (:300-315) for every
 * base; the reference's "same piece as last time" shortcut picks the same piece when the
 * knots are strictly increasing.  knots sorted by `in` (the reference qsorts them, :452). */
void orc_map (double* v, uint32_t n, const double* kin, const double* kout, uint32_t nknots)
	{
	uint32_t maxIx = nknots - 1, ix, lo, hi, mid;
	double   x;
	for (ix=0 ; ix<n ; ix++)
		{
		x = v[ix];
		if (x <= kin[0])     { v[ix] = kout[0];      continue; }
		if (x >= kin[maxIx]) { v[ix] = kout[maxIx];  continue; }
		if (x != x) continue;
		lo = 0;  hi = maxIx;
		while (lo + 1 < hi)
			{
			mid = (lo + hi) / 2;
			if      (x < kin[mid]) hi = mid;
			else if (x > kin[mid]) lo = mid;
			else                 { lo = mid;  break; }
			}
		while ((lo < maxIx) && (kin[lo] == kin[lo+1])) lo++;
		if      (x == kin[lo])   v[ix] = kout[lo];
		else if (x == kin[lo+1]) v[ix] = kout[lo+1];
		else v[ix] = kout[lo] + (x - kin[lo]) * (kout[lo+1] - kout[lo]) / (kin[lo+1] - kin[lo]);
		}
	}

/* clump.c:494-736 (clump_search): mark every stretch of at least minLength bases whose average is
 * >= average (above) or <= average (!above), merge them, trim each merged run to its first and
 * last base on the right side of the threshold.  Same walk as the reference: one running sum of
 * (v - average), the list of its strict record minima (value, position) starting with (0, -1), a
 * cursor on the earliest record not above the running sum (moved only in the direction the sum
 * just moved, :591-607), and the longest qualifying stretch ending at each base (:613-646).
 * relative lengths (:512-518) are resolved by the caller. */
void orc_clump (double* v, uint32_t n, double average, uint32_t minLength, int above, double one, double zero)
	{
	double*   mark  = (double*)  malloc (((size_t) n + 1) * sizeof(double));
	double*   recV  = (double*)  malloc (((size_t) n + 1) * sizeof(double));
	uint32_t* recAt = (uint32_t*) malloc (((size_t) n + 1) * sizeof(uint32_t));
	uint32_t  nrec, cursor, ix, iy, from, to, runFrom = (uint32_t) -1, runTo = (uint32_t) -1, scan;
	double    sum = 0.0, lowest = 0.0, d;
	int       hopeless = 1;

	for (ix=0 ; ix<n ; ix++)                                       /* :548-569 */
		{
		d = above? v[ix] - average : average - v[ix];
		if (d >= 0.0) { hopeless = 0;  break; }
		}
	if (hopeless)
		{
		for (ix=0 ; ix<n ; ix++) v[ix] = zero;
		goto done;
		}

	recV[0] = 0.0;  recAt[0] = (uint32_t) -1;  nrec = 1;  cursor = 0;
	for (ix=0 ; ix<n ; ix++)
		{
		d = above? v[ix] - average : average - v[ix];
		mark[ix] = zero;
		sum += d;
		if (sum < lowest) { lowest = sum;  recV[nrec] = sum;  recAt[nrec] = ix;  nrec++; }
		if      (d < 0) { while (recV[cursor] > sum) cursor++; }
		else if (d > 0) { while ((cursor > 0) && (recV[cursor-1] <= sum)) cursor--; }

		if (ix - recAt[cursor] < minLength) continue;              /* u32 arithmetic, as :614 */
		from = recAt[cursor] + 1;  to = ix;
		if ((runFrom == (uint32_t) -1) || (from > runTo + 1))
			{ for (iy=from ; iy<=to ; iy++) mark[iy] = one;  runFrom = from;  runTo = to; }
		else if (from >= runFrom)
			{ for (iy=runTo+1 ; iy<=to ; iy++) mark[iy] = one;  runTo = to; }
		else
			{
			for (iy=from ; iy<runFrom ; iy++) mark[iy] = one;
			for (iy=runTo+1 ; iy<=to ; iy++) mark[iy] = one;
			runFrom = from;  runTo = to;
			}
		}

	scan = 0;                                                      /* :651-716 */
	for (;;)
		{
		for (from=scan ; from<n ; from++) { if (mark[from] != zero) break;  v[from] = zero; }
		if (from >= n) break;
		for (ix=from ; ix<n ; ix++)
			{
			if (mark[ix] == zero) break;
			if (above? (v[ix] >= average) : (v[ix] <= average)) break;
			v[ix] = zero;
			}
		if ((ix >= n) || (mark[ix] == zero)) { scan = ix;  continue; }
		from = ix;  to = ix++;
		for ( ; ix<n ; ix++)
			{
			if (mark[ix] == zero) break;
			if (above? (v[ix] >= average) : (v[ix] <= average)) to = ix;
			}
		for (iy=from ; iy<=to ; iy++) v[iy] = one;
		for (iy=to+1 ; iy<ix ; iy++) v[iy] = zero;
		scan = ix;
		}
done:
	free (mark);  free (recV);  free (recAt);
	}

/* add.c:909-923: genome-wide min and max, seeded with the first element of
 * the first (longest) chromosome */
void orc_genome_minmax (const double* const* vecs, const uint32_t* lens, int nchrom, double* minOut, double* maxOut)
	{
	double   lo = vecs[0][0], hi = vecs[0][0];
	int      c;
	uint32_t ix;
	for (c=0 ; c<nchrom ; c++)
		for (ix=0 ; ix<lens[c] ; ix++)
			{
			if (vecs[c][ix] < lo) lo = vecs[c][ix];
			if (vecs[c][ix] > hi) hi = vecs[c][ix];
			}
	*minOut = lo;  *maxOut = hi;
	}

/* add.c:935-936 */
void orc_invert (double* v, uint32_t n, double mid)
	{
	uint32_t ix;
	for (ix=0 ; ix<n ; ix++) v[ix] = 2*mid - v[ix];
	}

/* --------------------------------------------------------- percentile.c -- */

static int orc_ascending (const void* a, const void* b)   /* genodsp.c:2262-2270 */
	{
	double x = *(const double*) a, y = *(const double*) b;
	return (x > y) - (x < y);
	}

/* percentile.c:547-710 computes, for each requested percentile pt (thousandths
 * of a percent), the k-th smallest of the sample {v[ix] : ix % window == 0,
 * min <= v[ix] <= max} taken over all chromosomes, with
 *   k = (u32) ((u64) numValues * pt / (100.0*1000))          (:587-589, :681)
 * and k == numValues meaning the largest value (:688-710).  The reference gets
 * there by shuffling the sample to the front of the genome and sorting in
 * place; the order statistic itself is what is restated here (copy + qsort
 * with the reference's comparator), leaving the signal untouched.  The p=0 and
 * p=100 short cuts (:434-530) return the same min/max this does. */
uint32_t orc_percentile (const double* const* vecs, const uint32_t* lens, int nchrom,
                         uint32_t window, double minAllowed, double maxAllowed,
                         const uint32_t* pThousandths, int np, double* out)
	{
	uint64_t cap = 0, ix;
	uint32_t count = 0, k;
	double*  sample;
	int      c, i;

	if (window == 0) window = 1;
	for (c=0 ; c<nchrom ; c++) cap += ((uint64_t) lens[c] + window - 1) / window;
	sample = (double*) malloc ((cap? cap : 1) * sizeof(double));
	for (c=0 ; c<nchrom ; c++)
		for (ix=0 ; ix<lens[c] ; ix+=window)
			{
			if (vecs[c][ix] < minAllowed) continue;
			if (vecs[c][ix] > maxAllowed) continue;
			sample[count++] = vecs[c][ix];
			}
	if (count == 0) { free (sample);  return 0; }

	qsort (sample, count, sizeof(double), orc_ascending);
	for (i=0 ; i<np ; i++)
		{
		k = (uint32_t) (((uint64_t) count) * pThousandths[i] / (100.0*1000));
		if (k >= count) k = count-1;
		out[i] = sample[k];
		}
	free (sample);
	return count;
	}

/* ------------------------------------- genodsp.c / add.c / multiply.c ---- */

void orc_fill (double* v, uint32_t n, double val)
	{ uint32_t ix;  for (ix=0 ; ix<n ; ix++) v[ix] = val; }

/* genodsp.c:1305-1331 (and add.c:282-283 for overlapOp=sum, clear=0): each
 * interval, in file order, is accumulated over [start,end); with `clear` a
 * position still holding the missing value is assigned instead. */
void orc_apply_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end,
                          const double* val, uint32_t count, int overlapOp, int clear, double missingVal)
	{
	uint32_t i, ix, s, e;
	double   x;

	for (i=0 ; i<count ; i++)
		{
		s = start[i];  e = end[i];  x = val[i];
		if (e > n) e = n;
		for (ix=s ; ix<e ; ix++)
			{
			if (clear && (v[ix] == missingVal)) v[ix] = x;
			else if (overlapOp == ORC_OVERLAP_MIN) { if (x < v[ix]) v[ix] = x; }
			else if (overlapOp == ORC_OVERLAP_MAX) { if (x > v[ix]) v[ix] = x; }
			else v[ix] += x;
			}
		}
	}

/* multiply.c:311-345 / :700-740: sorted, non-overlapping intervals; inside an
 * interval v*=val (or v/=val); everything outside any interval becomes 0
 * (multiply) or, for divide, (v>=0)? +infinityVal : -infinityVal (:711). */
void orc_scale_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end,
                          const double* val, uint32_t count, int divide, double infinityVal)
	{
	uint32_t i, ix, prevEnd = 0, s, e;

	for (i=0 ; i<=count ; i++)
		{
		s = (i < count)? start[i] : n;
		e = (i < count)? end[i]   : n;
		for (ix=prevEnd ; ix<s ; ix++)
			{
			if (!divide) v[ix] = 0.0;
			else         v[ix] = (v[ix] >= 0)? infinityVal : -infinityVal;
			}
		if (i == count) break;
		for (ix=s ; ix<e ; ix++)
			{ if (divide) v[ix] /= val[i];  else v[ix] *= val[i]; }
		prevEnd = e;
		}
	}

/* mask.c:283-284 (mask) and logical.c:471-472,:535-536 (or): optional "nonzero -> 1.0" pass,
 * then every base under an interval is assigned that interval's value, in file order.
 * mask.c:593-611 (masknot) and logical.c:766-767,:862-880 (and): optional "nonzero -> 1.0"
 * pass, then every base under NO interval (sorted, non-overlapping) gets outsideVal. */
void orc_mask_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end,
                         const double* val, uint32_t count, int inside, double outsideVal, int binarizeFirst)
	{
	uint32_t i, ix, prevEnd = 0, e;
	if (binarizeFirst) for (ix=0 ; ix<n ; ix++) { if (v[ix] != 0.0) v[ix] = 1.0; }
	if (inside)
		{
		for (i=0 ; i<count ; i++)
			{
			e = (end[i] > n)? n : end[i];
			for (ix=start[i] ; ix<e ; ix++) v[ix] = val[i];
			}
		return;
		}
	for (i=0 ; i<count ; i++)
		{
		for (ix=prevEnd ; ix<start[i] ; ix++) v[ix] = outsideVal;
		prevEnd = (end[i] > n)? n : end[i];
		}
	for (ix=prevEnd ; ix<n ; ix++) v[ix] = outsideVal;
	}

/* minmax.c:270-352 (minover) and :673-755 (maxover): per sorted, non-overlapping interval the
 * extreme value and, among ties, the position with the largest inset min(ix-start, end-ix)
 * (the earliest on equal insets) survive; every other base becomes `fill`. */
void orc_extreme_in_intervals (double* v, uint32_t n, const uint32_t* start, const uint32_t* end,
                               uint32_t count, int wantMax, double fill)
	{
	uint32_t i, ix, prevEnd = 0, s, e, bestIx, inset, maxInset;
	double   best;
	for (i=0 ; i<count ; i++)
		{
		s = start[i];  e = (end[i] > n)? n : end[i];
		for (ix=prevEnd ; ix<s ; ix++) v[ix] = fill;
		best = v[s];  bestIx = s;  maxInset = 0;
		for (ix=s+1 ; ix<e ; ix++)
			{
			if (wantMax? (v[ix] < best) : (v[ix] > best)) continue;
			if (wantMax? (v[ix] > best) : (v[ix] < best))
				{
				best = v[ix];  bestIx = ix;
				maxInset = (ix-s < e-ix)? ix-s : e-ix;
				continue;
				}
			inset = (ix-s < e-ix)? ix-s : e-ix;
			if (inset > maxInset) { bestIx = ix;  maxInset = inset; }
			}
		for (ix=s ; ix<e ; ix++) { if (ix != bestIx) v[ix] = fill; }
		prevEnd = e;
		}
	for (ix=prevEnd ; ix<n ; ix++) v[ix] = fill;
	}

/* genodsp.c:1587-1678: run-length encoding of one
 * chromosome as report_intervals emits it.  Exact zeros end a run and are not
 * reported unless uncovered==show; equal neighbours collapse when asked to. */
uint32_t orc_report_runs (const double* v, uint32_t n, int collapse, int uncovered,
                          uint32_t* runStart, uint32_t* runEnd, double* runVal, uint32_t cap)
	{
	uint32_t ix, start = 0, runs = 0;
	int      active = (uncovered != 0);
	double   val = 0.0;

#define ORC_EMIT(s,e,x) do { if (runs < cap) { runStart[runs]=(s); runEnd[runs]=(e); runVal[runs]=(x); } runs++; } while (0)
	for (ix=0 ; ix<n ; ix++)
		{
		if ((v[ix] == 0) && (uncovered != 1))
			{
			if (active && (ix != start)) ORC_EMIT (start, ix, val);
			active = 0;  start = 0;  val = 0.0;
			continue;
			}
		if (!active) { active = 1;  start = ix;  val = v[ix];  continue; }
		if ((v[ix] == val) && collapse) continue;
		if (ix != start) ORC_EMIT (start, ix, val);
		active = 1;  start = ix;  val = v[ix];
		}
	if (active && (n != start)) ORC_EMIT (start, n, val);
#undef ORC_EMIT
	return runs;
	}

/* ---------------------------------------------------- synthetic signal ---- */

static uint64_t orc_mix64 (uint64_t x)            /* splitmix64 finaliser */
	{
	x += 0x9E3779B97F4A7C15ULL;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
	return x ^ (x >> 31);
	}

static uint32_t orc_cell_depth (uint64_t key, int64_t cell)
	{
	uint64_t h = orc_mix64 (key ^ ((uint64_t) cell * 0xD1342543DE82EF95ULL));
	if ((h >> 8) % 100 < 35) return 0;
	return 1 + (uint32_t) ((h >> 16) & 15) + (uint32_t) ((h >> 20) & 15)
	         + (uint32_t) ((h >> 24) & 15) + (uint32_t) ((h >> 28) & 15);
	}

/* Coverage-like signal: the chromosome is cut into 128-base cells; cell c has
 * one breakpoint b(c) in [0,128); positions left of it continue cell c-1's
 * depth, positions from it on take cell c's depth.  Depths are 0 (35 %) or a
 * bell-shaped 1..61.  Mode 1 multiplies by a per-position factor in [0.5,1.5). */
void orc_synth_coverage (uint64_t seed, uint32_t chromIndex, uint32_t start, uint32_t count,
                         int mode, double* out)
	{
	uint64_t key = orc_mix64 (seed ^ ((uint64_t) (chromIndex+1) << 40));
	uint64_t i, pos, hb, hp;
	int64_t  cell;
	uint32_t off, brk, d;
	double   x;

	for (i=0 ; i<count ; i++)
		{
		pos  = (uint64_t) start + i;
		cell = (int64_t) (pos >> 7);
		off  = (uint32_t) (pos & 127);
		hb   = orc_mix64 (key ^ 0xA5A5A5A5ULL ^ ((uint64_t) cell * 0x9E3779B97F4A7C15ULL));
		brk  = (uint32_t) (hb & 127);
		d    = (off >= brk)? orc_cell_depth (key, cell) : orc_cell_depth (key, cell-1);
		x    = (double) d;
		if (mode == 1)
			{
			hp = orc_mix64 (key ^ 0x5bd1e995ULL ^ (pos * 0xC2B2AE3D27D4EB4FULL));
			x  = x * (0.5 + (double) (hp >> 11) * (1.0 / 9007199254740992.0));
			}
		out[i] = x;
		}
	}
