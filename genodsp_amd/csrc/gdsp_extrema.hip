// gdsp_extrema.hip -- localmin/localmax and bestmin/bestmax (sliding min/max).
//
// Reference: op_local_maxima_apply minmax.c:1183-1227, op_local_minima_apply :981-1022,
// op_best_local_max_apply :1616-1721, op_best_local_min_apply :1369-1474.
//
// Both families reduce to one primitive, the extreme E[i] of v over a window
// [i-lft, i+rgt] clamped to the vector:
//   bestmax:  out[i] = E[i]                       (lft=(W-1)/2, rgt=W-1-lft)
//   localmax: out[i] = (E[i] > v[i]) ? fill : v[i]   (lft=rgt=(N-1)/2)
// The second line is the reference's "some other position in the neighbourhood
// is strictly greater" test: the centre itself never beats itself, NaN
// neighbours never knock a value out and a NaN centre is never knocked out,
// because E is formed with comparisons that ignore NaN exactly as `v[j] > val`
// does.  All work is comparisons, so results are bit-identical to the
// reference (ties between +0.0 and -0.0 in bestmin/bestmax excepted: the
// reference's pick depends on its scan history).
//
// HBM-bound (16 B/base).  A workgroup stages a tile plus its halo in LDS with
// 16-byte loads.  Windows up to 4 bases are scanned directly from LDS (two
// adjacent outputs per lane); windows of 5 .. 3584 bases use the block form
// (extrema_blocks_kernel: tail of a block + whole blocks from one level of a
// sparse table + head of a block, four comparisons per base for any window,
// 5.5-6.0 TB/s measured from 33 to 2049 bases); longer ones the van Herk /
// Gil-Werman decomposition over window-long segments in one LDS tile, and past
// that gdsp_longwin.hip through HBM workspace.

#include <float.h>
#include <stdlib.h>
#include "gdsp_common.h"

#define EX_THREADS 256
#define EX_TILE    4096              // outputs per workgroup
#define EX_DIRECT_MAX_SPAN 4         // windows up to this many bases are scanned directly
#define EX_LDS_DOUBLES 18432         // 144 KiB of the 160 KiB LDS

template <bool MAX> __device__ __forceinline__ bool ex_beats (double a, double b)
	{ return MAX? (a > b) : (a < b); }
// extreme of two values with NaN never winning: v_max_f64 / v_min_f64 (IEEE maxNum/minNum)
template <bool MAX> __device__ __forceinline__ double ex_pick (double a, double b)
	{ return MAX? fmax (a, b) : fmin (a, b); }

// Exclusive segmented scan of one (value, flag) pair per thread over the workgroup, with
// `pick` as the operator: returns the extreme of the values of the threads before this one
// (after it, when REVERSE) back to the nearest thread whose flag is set, that thread included;
// `pad` when there is none.  Wave shuffles inside a wave, one LDS hop across the four waves.
template <bool MAX, bool REVERSE>
__device__ __forceinline__ double ex_seg_carry (double val, bool flag, double pad, double* scanV, int* scanF)
	{
	const int lane = REVERSE? 63 - (int) (threadIdx.x & 63) : (int) (threadIdx.x & 63);
	const int wave = REVERSE? (EX_THREADS/64 - 1) - (int) (threadIdx.x >> 6) : (int) (threadIdx.x >> 6);
	double v = val;
	int    f = flag? 1 : 0;
	for (int d=1 ; d<64 ; d*=2)
		{
		const double v2 = REVERSE? __shfl_down (v, d, 64) : __shfl_up (v, d, 64);
		const int    f2 = REVERSE? __shfl_down (f, d, 64) : __shfl_up (f, d, 64);
		if (lane >= d) { if (!f) v = ex_pick<MAX> (v2, v);  f |= f2; }
		}
	// exclusive within the wave
	double cv = REVERSE? __shfl_down (v, 1, 64) : __shfl_up (v, 1, 64);
	int    cf = REVERSE? __shfl_down (f, 1, 64) : __shfl_up (f, 1, 64);
	if (lane == 0) { cv = pad;  cf = 0; }
	__syncthreads ();                            // scanV/scanF may still be read by the previous scan
	if (lane == 63) { scanV[wave] = v;  scanF[wave] = f; }
	__syncthreads ();
	double pv = pad;
	for (int w=0 ; w<wave ; w++) pv = scanF[w]? scanV[w] : ex_pick<MAX> (pv, scanV[w]);
	return cf? cv : ex_pick<MAX> (pv, cv);
	}

// LOCAL: localmin/localmax semantics; otherwise bestmin/bestmax
enum { EX_DIRECT = 0, EX_SEGMENTS = 2 };

template <bool MAX, bool LOCAL, int METHOD>
__global__ __launch_bounds__(EX_THREADS)
void extrema_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t ntiles,
                     uint32_t lft, uint32_t rgt, double fill, int tile)
	{
	extern __shared__ __attribute__((aligned(16))) double lds[];
	__shared__ double scanV[EX_THREADS/64];
	__shared__ int    scanF[EX_THREADS/64];
	const double   pad       = MAX? -INFINITY : INFINITY;   // never beats anything, like "outside the vector"
	const uint32_t t         = gdsp_xcd_tile (blockIdx.x, ntiles);
	const int64_t  tileStart = (int64_t) t * tile;
	const int      sh        = (int) (lft & 1);
	const int64_t  g0        = tileStart - lft - sh;
	const int      span      = (int) (lft + rgt + 1);
	const int      L         = (tile + span - 1 + sh + 1) & ~1;

	gdsp_stage_f64<EX_THREADS> (lds, in, n, g0, L, pad);
	__syncthreads ();

	const double* x = lds + sh;                  // x[o + k], k in [0,span): window of output o

	if (METHOD == EX_DIRECT)
		{
		for (int o = 2*threadIdx.x ; o < tile ; o += 2*EX_THREADS)
			{
			// two adjacent outputs share all but one input each
			double e0 = x[o], e1 = x[o+span];
			double mid = x[o+1];
			for (int k=2 ; k<span ; k++) mid = ex_pick<MAX> (mid, x[o+k]);
			if (span > 1) { e0 = ex_pick<MAX> (e0, mid);  e1 = ex_pick<MAX> (mid, e1); }
			else          { e1 = x[o+1]; }
			if (LOCAL)
				{
				double c0 = x[o+lft], c1 = x[o+1+lft];
				e0 = ex_beats<MAX> (e0, c0)? fill : c0;
				e1 = ex_beats<MAX> (e1, c1)? fill : c1;
				}
			int64_t g = tileStart + o;
			if (g + 1 < (int64_t) n) gdsp_st2 (reinterpret_cast<double2*> (out + g), make_double2 (e0, e1));
			else if (g < (int64_t) n) out[g] = e0;
			}
		return;
		}

	// Long windows, van Herk / Gil-Werman: cut the staged stretch into segments of `span`
	// bases; G[q] = extreme from the start of q's segment to q, Hs[q] = extreme from q to the
	// end of its segment.  A window of `span` bases starting at o covers the tail of one
	// segment and the head of the next, so its extreme is pick(Hs[o], G[o+span-1]): three
	// comparisons per base whatever the window length.  Each thread scans C consecutive
	// bases; what crosses thread boundaries is a 256-element segmented scan.
	double*   Hs = lds + L;                      // second array, x-space index
	double*   G  = lds + sh;                     // first array, overwritten in place after Hs is built
	const int Lx = L - sh;
	int       C  = (Lx + EX_THREADS - 1) / EX_THREADS;
	C |= 1;                                      // odd lane stride: conflict-free ds_read_b64
	const int q0 = threadIdx.x * C;
	const int q1 = (q0 + C < Lx)? q0 + C : Lx;

	// ---- suffix extremes (right to left); a segment ends at q % span == span-1 or at Lx-1
		{
		double run = pad;
		int    lastEnd = -1;                     // right-most segment end inside this chunk
		int    r = (q1 > q0)? (q1 - 1) % span : 0;
		for (int q=q1-1 ; q>=q0 ; q--)
			{
			if ((r == span-1) || (q == Lx-1)) { run = pad;  if (lastEnd < 0) lastEnd = q; }
			run = ex_pick<MAX> (run, x[q]);
			Hs[q] = run;
			r = (r == 0)? span-1 : r-1;
			}
		const double carry = ex_seg_carry<MAX, true> (run, lastEnd >= 0, pad, scanV, scanF);
		const int stop = (lastEnd >= 0)? lastEnd : q0 - 1;      // bases right of the last end see the carry
		for (int q=q1-1 ; q>stop ; q--) Hs[q] = ex_pick<MAX> (Hs[q], carry);
		}
	__syncthreads ();

	// ---- prefix extremes (left to right), in place over the staged values
		{
		double run = pad;
		int    firstStart = -1;
		int    r = q0 % span;
		for (int q=q0 ; q<q1 ; q++)
			{
			if (r == 0) { run = pad;  if (firstStart < 0) firstStart = q; }
			run = ex_pick<MAX> (run, G[q]);
			G[q] = run;
			r = (r == span-1)? 0 : r+1;
			}
		const double carry = ex_seg_carry<MAX, false> (run, firstStart >= 0, pad, scanV, scanF);
		const int stop = (firstStart >= 0)? firstStart : q1;
		for (int q=q0 ; q<stop ; q++) G[q] = ex_pick<MAX> (G[q], carry);
		}
	__syncthreads ();

	for (int o = 2*threadIdx.x ; o < tile ; o += 2*EX_THREADS)
		{
		int64_t g = tileStart + o;
		if (g >= (int64_t) n) break;
		double e0 = ex_pick<MAX> (Hs[o],   G[o   + span - 1]);
		double e1 = ex_pick<MAX> (Hs[o+1], G[o+1 + span - 1]);
		if (LOCAL)
			{
			// the staged copy was consumed by the prefix pass; centres come back from L2
			double c0 = in[g], c1 = (g + 1 < (int64_t) n)? in[g+1] : 0.0;
			e0 = ex_beats<MAX> (e0, c0)? fill : c0;
			e1 = ex_beats<MAX> (e1, c1)? fill : c1;
			}
		if (g + 1 < (int64_t) n) gdsp_st2 (reinterpret_cast<double2*> (out + g), make_double2 (e0, e1));
		else                     out[g] = e0;
		}
	}

// ------------------------------------------------------------- block form ----
// Windows of 5 .. EXB_MAX_SPAN bases.  The staged tile is cut into blocks of G = 16 (8, 4 for short windows); a window [a,b] is
// the tail of a's block + the whole blocks between + the head of b's block, and an extreme does
// not mind overlap, so the blocks between are two overlapping power-of-two ranges of block
// extremes (one level of a sparse table, built by doubling over the 256 block extremes of the tile).
// One thread owns one block as the right end b of 16 windows: prefix extremes of its own block in
// registers, then one backward walk over the 16+dr elements that hold the 16 left ends, finishing
// one output per step -- four v_max_f64 per base whatever the window, ~45 B/base of LDS traffic
// against 16 B/base of HBM.  The LDS image has a pitch of 17 per block of 16, so the lane-strided
// reads of both walks are conflict free (the same layout as gdsp_hann.hip).
#define EXB_THREADS  256
#define EXB_MIN_SPAN 5
#define EXB_MAX_SPAN 3584

// G = elements per block (and per thread): 16 for windows of 17 bases and more, 8 for 9..16, 4 for 5..8 --
// a window has to reach past its right end's block for the decomposition to apply
// SET: 0 = values as they are; 1 / 2 = dilate / erode (morphology.c:882-1072, :1331-1454): what is staged is
// the membership of a base in the set (1.0 / 0.0, nothing outside the vector), the window extreme is then
// "any" (MAX) or "all" (MIN) over [i-right, i+left], and `one` or `zero` is written
#define EXB_VALUES 0
#define EXB_DILATE 1
#define EXB_ERODE  2
struct ExbSet { double T, one, zero; };

template <int SET>
__device__ __forceinline__ double exb_staged (double x, int64_t g, double T)
	{
	if (SET == EXB_DILATE) return ((g == 0)? (x > T) : !(x <= T))? 1.0 : 0.0;      // morphology.c:930 vs :935
	if (SET == EXB_ERODE)  return (x > T)? 1.0 : 0.0;                              // :1384, :1391
	return x;
	}

template <bool MAX, bool LOCAL, int G, int SET>
__device__ __forceinline__
void extrema_blocks_tile (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t tile,
                          int rgt, int dq, int dr, int level, int sh, double fill, const ExbSet& set)
	{
	constexpr int PITCH = G + 1;
	constexpr int ELEMS = EXB_THREADS * G;
	constexpr int LOG_G = (G == 16)? 4 : ((G == 8)? 3 : 2);
	__shared__ __attribute__((aligned(16))) double lds[EXB_THREADS * PITCH];
	__shared__ double blockExt[2][EXB_THREADS];
	const double pad   = MAX? -INFINITY : INFINITY;           // never beats anything, like "outside the vector"
	const double away  = (SET == EXB_VALUES)? pad : 0.0;      // what is staged for positions outside the vector
	const int    haloL = dq + 1;                              // leading blocks that only feed
	const int    outs  = (EXB_THREADS - haloL) * G - 2*sh;    // outputs stored per tile (even)
	const int    nt    = dq - 1;                              // whole blocks always between
	const int    lead  = haloL * G - rgt + sh;                // staged elements before output 0 of the tile (even)
	const int64_t  out0 = (int64_t) tile * outs;
	const int64_t  e0   = out0 - lead;
	const int      p    = threadIdx.x;

	// ---- stage 256*G elements
	if ((e0 >= 0) && (e0 + ELEMS <= (int64_t) n))
		{
		const double2* src = reinterpret_cast<const double2*> (in + e0);
		double2 r[G/2];
#pragma unroll
		for (int u=0 ; u<G/2 ; u++) r[u] = src[u*EXB_THREADS + p];      // (plain loads: with a halo of up to half a tile the neighbours' re-reads should find the L2; non-temporal cost 4 % at 1001 bases)
#pragma unroll
		for (int u=0 ; u<G/2 ; u++)
			{
			const int e = 2 * (u*EXB_THREADS + p);
			double* dst = lds + e + (e >> LOG_G);
			dst[0] = exb_staged<SET> (r[u].x, e0 + e, set.T);  dst[1] = exb_staged<SET> (r[u].y, e0 + e + 1, set.T);
			}
		}
	else
		{
		for (int e=p ; e<ELEMS ; e+=EXB_THREADS)
			{
			const int64_t g = e0 + e;
			lds[e + (e >> LOG_G)] = ((g >= 0) && (g < (int64_t) n))? exb_staged<SET> (in[g], g, set.T) : away;
			}
		}
	__syncthreads ();

	// ---- prefix extremes of the own block; its extreme to the table
	double P[G];
		{
		const double* xb = lds + p * PITCH;
		double run = pad;
#pragma unroll
		for (int u=0 ; u<G ; u++) { run = ex_pick<MAX> (run, xb[u]);  P[u] = run; }
		blockExt[0][p] = run;
		}
	__syncthreads ();
	// one level of the sparse table: entry q covers blocks q .. q + 2^level - 1
	for (int l=1 ; l<=level ; l++)
		{
		const int    half = 1 << (l - 1);
		const double a    = blockExt[(l-1) & 1][p];
		const double b    = (p + half < EXB_THREADS)? blockExt[(l-1) & 1][p + half] : pad;
		blockExt[l & 1][p] = ex_pick<MAX> (a, b);
		__syncthreads ();
		}

	// ---- one output per left end
	const bool live = (p >= haloL);
	if (live)
		{
		const double* ext = blockExt[level & 1];
		double T = pad;                                           // whole blocks p-nt .. p-1
		if (nt > 0) T = ex_pick<MAX> (ext[p - nt], ext[p - (1 << level)]);
		const double* lb = lds + (p - dq) * PITCH;                // block of the left ends of s >= dr
		const double* la = lb - PITCH + G;                        // the block before it, indexed by u - dr < 0
		double run = pad;
#pragma unroll
		for (int u=2*G-2 ; u>=0 ; u--)
			{
			if (u >= G + dr) continue;                              // (uniform) the walk starts at u = G-1 + dr
			if (u == dr - 1) { T = ex_pick<MAX> (T, run);  run = pad; }   // that block is whole for the remaining windows
			const int rel = u - dr;
			run = ex_pick<MAX> (run, (rel >= 0)? lb[rel] : la[rel]);
			if (u < G)
				{
				double e = ex_pick<MAX> (ex_pick<MAX> (run, T), P[u]);
				if (LOCAL)
					{
					const int    c = G * p + u - rgt;                   // the centre of this window
					const double v = lds[c + (c >> LOG_G)];
					e = ex_beats<MAX> (e, v)? fill : v;
					}
				if (SET != EXB_VALUES) e = (e != 0.0)? set.one : set.zero;
				P[u] = e;
				}
			}
		}
	__syncthreads ();                                              // every read of the staged inputs is done

	// ---- results back through LDS: thread haloL + k holds outputs G*k - sh .. G*k + G-1 - sh of the tile
	if (live)
		{
		double* mine = lds + (p - haloL) * PITCH;
#pragma unroll
		for (int u=0 ; u<G ; u++) mine[u] = P[u];
		}
	__syncthreads ();
	if (out0 + outs <= (int64_t) n)
		{
		double2* dst = reinterpret_cast<double2*> (out + out0);
		for (int q=p ; q<outs/2 ; q+=EXB_THREADS)
			{
			const int o = 2*q + sh;
			const int o1 = o + 1;
			gdsp_st2 (&dst[q], make_double2 (lds[o + (o >> LOG_G)], lds[o1 + (o1 >> LOG_G)]));
			}
		}
	else
		{
		for (int q=p ; q<outs ; q+=EXB_THREADS)
			{
			const int o = q + sh;
			if (out0 + q < (int64_t) n) out[out0 + q] = lds[o + (o >> LOG_G)];
			}
		}
	}

template <bool MAX, bool LOCAL, int G, int SET>
__global__ __launch_bounds__(EXB_THREADS)
void extrema_blocks_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t ntiles,
                            int rgt, int dq, int dr, int level, int sh, double fill, ExbSet set)
	{ extrema_blocks_tile<MAX, LOCAL, G, SET> (in, out, n, gdsp_xcd_tile (blockIdx.x, ntiles), rgt, dq, dr, level, sh, fill, set); }

template <bool MAX, bool LOCAL, int G, int SET>                   // one grid over every vector of the table (gdsp_common.h)
__global__ __launch_bounds__(EXB_THREADS)
void extrema_blocks_batch_kernel (GdspBatch B, int rgt, int dq, int dr, int level, int sh, double fill, ExbSet set)
	{
	const double* in;  double* out;  uint32_t n;
	const uint32_t tile = gdsp_batch_tile (B, in, out, n);
	extrema_blocks_tile<MAX, LOCAL, G, SET> (in, out, n, tile, rgt, dq, dr, level, sh, fill, set);
	}

template <bool MAX, bool LOCAL, int G, int SET = EXB_VALUES>
static void extrema_blocks_batch_launch (const gdsp_batch_item* items, int nitems, uint32_t lft, uint32_t rgt, double fill,
                                         hipStream_t s, ExbSet set = ExbSet ())
	{
	const int d  = (int) (lft + rgt);
	const int dq = d / G, dr = d % G;
	const int nt = dq - 1;
	int level = 0;
	while ((2 << level) <= nt) level++;
	const int sh   = ((dq + 1) * G - (int) rgt) & 1;
	const int outs = (EXB_THREADS - (dq + 1)) * G - 2*sh;
	gdsp_batch_run (items, nitems, [=] (uint32_t n) { return ((uint64_t) n + outs - 1) / outs; },
		[&] (const GdspBatch& B, uint32_t tiles)
			{
			hipLaunchKernelGGL ((extrema_blocks_batch_kernel<MAX, LOCAL, G, SET>), dim3(tiles), dim3(EXB_THREADS), 0, s,
			                    B, (int) rgt, dq, dr, level, sh, fill, set);
			});
	}

template <bool MAX, bool LOCAL, int G, int SET = EXB_VALUES>
static void extrema_blocks_launch (const double* d_in, double* d_out, uint32_t n, uint32_t lft, uint32_t rgt, double fill,
                                   hipStream_t s, ExbSet set = ExbSet ())
	{
	const int d  = (int) (lft + rgt);                             // left end = right end - d
	const int dq = d / G, dr = d % G;
	const int nt = dq - 1;
	int level = 0;
	while ((2 << level) <= nt) level++;                           // largest power of two <= nt (0 when nt <= 1)
	const int sh   = ((dq + 1) * G - (int) rgt) & 1;              // keeps the first staged element even
	const int outs = (EXB_THREADS - (dq + 1)) * G - 2*sh;
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + outs - 1) / outs);
	hipLaunchKernelGGL ((extrema_blocks_kernel<MAX, LOCAL, G, SET>), dim3(ntiles), dim3(EXB_THREADS), 0, s,
	                    d_in, d_out, n, ntiles, (int) rgt, dq, dr, level, sh, fill, set);
	}

// dilate / erode with a reach the block form covers: the window of base i is [i-right, i+left]
bool gdsp_morph_blocks_available (uint32_t left, uint32_t right)
	{ const uint64_t span = (uint64_t) left + right + 1;  return (span >= 17) && (span <= EXB_MAX_SPAN); }

void gdsp_morph_blocks (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right, int erode,
                        double T, double one, double zero, void* stream)
	{
	ExbSet set = { T, one, zero };
	if (erode) extrema_blocks_launch<false, false, 16, EXB_ERODE>  (d_in, d_out, n, right, left, 0.0, gdsp_stream (stream), set);
	else       extrema_blocks_launch<true,  false, 16, EXB_DILATE> (d_in, d_out, n, right, left, 0.0, gdsp_stream (stream), set);
	}

template <bool MAX, bool LOCAL>
static int extrema_launch (const double* d_in, double* d_out, uint32_t n, uint32_t lft, uint32_t rgt,
                           double fill, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_in != NULL) && (d_out != NULL), "NULL vector");
	GDSP_REQUIRE (d_in != d_out, "out-of-place operator: d_out must not alias d_in");
	GDSP_REQUIRE (gdsp_aligned16 (d_in) && gdsp_aligned16 (d_out), "vectors must be 16-byte aligned");

	// windows longer than the vector see the whole vector anyway
	if (lft > n) lft = n;
	if (rgt > n) rgt = n;
	const uint64_t span   = (uint64_t) lft + rgt + 1;
	if ((span >= EXB_MIN_SPAN) && (span <= EXB_MAX_SPAN))
		{
		if      (span >= 17) extrema_blocks_launch<MAX, LOCAL, 16> (d_in, d_out, n, lft, rgt, fill, gdsp_stream (stream));
		else if (span >= 9)  extrema_blocks_launch<MAX, LOCAL, 8>  (d_in, d_out, n, lft, rgt, fill, gdsp_stream (stream));
		else                 extrema_blocks_launch<MAX, LOCAL, 4>  (d_in, d_out, n, lft, rgt, fill, gdsp_stream (stream));
		GDSP_LAUNCH_CHECK ();
		return GDSP_OK;
		}
	const bool     direct = (span <= EX_DIRECT_MAX_SPAN);
	int            tile   = EX_TILE;
	size_t         ldsDoubles;
	if (direct) ldsDoubles = (size_t) tile + span + 2;
	else
		{
		// two arrays of tile+span; the tile is the smallest that keeps the halo <= tile/2 that still fits
		tile = 1024;
		while (((uint64_t) tile < 2*(span-1)) && (2*(2*(uint64_t) tile + span + 2) <= EX_LDS_DOUBLES)) tile *= 2;
		ldsDoubles = 2 * ((size_t) tile + span + 2);
		if (ldsDoubles > EX_LDS_DOUBLES)
			{
			gdsp_set_error ("%s: window of %llu bases exceeds what one LDS tile holds (max %d)",
			                __func__, (unsigned long long) span, EX_LDS_DOUBLES/2 - 1024 - 2);
			return GDSP_EINVAL;
			}
		}
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + tile - 1) / tile);
	const size_t   bytes  = ldsDoubles * sizeof(double);
	hipStream_t    s      = gdsp_stream (stream);
	if (direct)
		hipLaunchKernelGGL ((extrema_kernel<MAX, LOCAL, EX_DIRECT>),  dim3(ntiles), dim3(EX_THREADS), bytes, s,
		                    d_in, d_out, n, ntiles, lft, rgt, fill, tile);
	else
		{
		if (bytes > 64*1024)          // more than 64 KiB of dynamic LDS has to be asked for
			GDSP_HIP_TRY (hipFuncSetAttribute ((const void*) extrema_kernel<MAX, LOCAL, EX_SEGMENTS>,
			                                   hipFuncAttributeMaxDynamicSharedMemorySize,
			                                   (int) (EX_LDS_DOUBLES*sizeof(double))));
		hipLaunchKernelGGL ((extrema_kernel<MAX, LOCAL, EX_SEGMENTS>), dim3(ntiles), dim3(EX_THREADS), bytes, s,
		                    d_in, d_out, n, ntiles, lft, rgt, fill, tile);
		}
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

// every vector of a device in one launch where the block form applies to all of them; vector by vector otherwise
template <bool MAX, bool LOCAL>
static int extrema_batch (const gdsp_batch_item* items, int nitems, uint32_t lft, uint32_t rgt, double fill, void* stream)
	{
	int rc = gdsp_batch_check (items, nitems, false);
	if (rc != GDSP_OK) return rc;
	const uint64_t span = (uint64_t) lft + rgt + 1;
	bool blocks = (span >= EXB_MIN_SPAN) && (span <= EXB_MAX_SPAN);
	for (int i=0 ; i<nitems ; i++)
		{ if ((items[i].n != 0) && ((items[i].n < lft) || (items[i].n < rgt))) blocks = false; }   // (the single form clamps the window to the vector)
	if (!blocks)
		{
		for (int i=0 ; i<nitems ; i++)
			{ rc = extrema_launch<MAX, LOCAL> (items[i].d_in, items[i].d_out, items[i].n, lft, rgt, fill, stream);  if (rc != GDSP_OK) return rc; }
		return GDSP_OK;
		}
	hipStream_t s = gdsp_stream (stream);
	if      (span >= 17) extrema_blocks_batch_launch<MAX, LOCAL, 16> (items, nitems, lft, rgt, fill, s);
	else if (span >= 9)  extrema_blocks_batch_launch<MAX, LOCAL, 8>  (items, nitems, lft, rgt, fill, s);
	else                 extrema_blocks_batch_launch<MAX, LOCAL, 4>  (items, nitems, lft, rgt, fill, s);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

// dilate / erode over a batch in the block form; false when some vector needs another kernel
bool gdsp_morph_blocks_batch (const gdsp_batch_item* items, int nitems, uint32_t left, uint32_t right, int erode,
                              double T, double one, double zero, void* stream)
	{
	if (!gdsp_morph_blocks_available (left, right)) return false;
	for (int i=0 ; i<nitems ; i++)
		{ if ((items[i].n != 0) && ((items[i].n < left) || (items[i].n < right))) return false; }
	ExbSet set = { T, one, zero };
	if (erode) extrema_blocks_batch_launch<false, false, 16, EXB_ERODE>  (items, nitems, right, left, 0.0, gdsp_stream (stream), set);
	else       extrema_blocks_batch_launch<true,  false, 16, EXB_DILATE> (items, nitems, right, left, 0.0, gdsp_stream (stream), set);
	return true;
	}

extern "C" {

int gdsp_local_extrema_batch (const gdsp_batch_item* items, int nitems, uint32_t N, int wantMax, double fill, void* stream)
	{
	GDSP_REQUIRE (N >= 1, "neighborhood must be >= 1");
	const uint32_t hOff = (N - 1) / 2;
	if (wantMax) return extrema_batch<true,  true> (items, nitems, hOff, hOff, fill, stream);
	return              extrema_batch<false, true> (items, nitems, hOff, hOff, fill, stream);
	}

int gdsp_best_extrema_batch (const gdsp_batch_item* items, int nitems, uint32_t W, int wantMax, void* stream)
	{
	GDSP_REQUIRE (W >= 1, "window must be >= 1");
	const uint32_t lft = (W - 1) / 2, rgt = (W - 1) - lft;
	if (wantMax) return extrema_batch<true,  false> (items, nitems, lft, rgt, 0.0, stream);
	return              extrema_batch<false, false> (items, nitems, lft, rgt, 0.0, stream);
	}

int gdsp_local_extrema (const double* d_in, double* d_out, uint32_t n, uint32_t N,
                        int wantMax, double fill, void* stream)
	{
	GDSP_REQUIRE (N >= 1, "neighborhood must be >= 1");
	const uint32_t hOff = (N - 1) / 2;                 // minmax.c:1201
	if (wantMax) return extrema_launch<true,  true> (d_in, d_out, n, hOff, hOff, fill, stream);
	return              extrema_launch<false, true> (d_in, d_out, n, hOff, hOff, fill, stream);
	}

int gdsp_best_extrema (const double* d_in, double* d_out, uint32_t n, uint32_t W,
                       int wantMax, void* stream)
	{
	GDSP_REQUIRE (W >= 1, "window must be >= 1");
	const uint32_t lft = (W - 1) / 2, rgt = (W - 1) - lft;   // minmax.c:1634-1635
	if (wantMax) return extrema_launch<true,  false> (d_in, d_out, n, lft, rgt, 0.0, stream);
	return              extrema_launch<false, false> (d_in, d_out, n, lft, rgt, 0.0, stream);
	}

} // extern "C"
