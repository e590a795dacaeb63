// gdsp_sums.hip -- slidingsum, sum (non-overlapping windows), cumulativesum.
//
// Reference: op_sliding_sum_apply sum.c:420-463, op_window_sum_apply :211-252,
// op_cumulative_sum_apply :776-792.
//
// Parity policy.  The reference forms these
// with ONE accumulator walking the whole chromosome, so its rounding error is a
// function of everything to the left of a position; no parallel evaluation can
// reproduce those bits for arbitrary reals.  What is reproduced exactly:
//   * `sum`: each window is summed by one thread in ascending order, exactly the
//     reference's order, for windows up to WS_SEQ_MAX bases -> bit-identical;
//   * `slidingsum`, `cumulativesum`, and `sum` over longer windows: bit-identical
//     whenever every partial sum is exactly representable (integer or dyadic
//     signals such as read depth, the operator's normal input); otherwise within
//     the reference's own accumulated rounding error (tests state the bound).
// All three are HBM-bound (16 B/base).

#include <mutex>
#include <vector>
#include <stdlib.h>
#include "gdsp_common.h"

#define SU_THREADS 256

// ------------------------------------------------- block-wide inclusive scan ----
// Inclusive prefix sum of lds[0..L) in place.  Each thread owns C consecutive
// values, C odd so the lane stride is conflict-free for ds_read_b64.
__device__ __forceinline__ void block_prefix_sum (double* lds, int L, int C, double* waveTotals)
	{
	const int base = threadIdx.x * C;
	double run = 0.0;
	for (int k=0 ; k<C ; k++)
		{ if (base + k < L) { run += lds[base+k];  lds[base+k] = run; } }

	// exclusive scan of the per-thread totals: wave shuffle scan, then across the 4 waves
	double incl = run;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int d=1 ; d<64 ; d*=2)
		{
		double up = __shfl_up (incl, d, 64);
		if (lane >= d) incl += up;
		}
	if (lane == 63) waveTotals[wave] = incl;
	__syncthreads ();
	double offset = incl - run;
	for (int w=0 ; w<wave ; w++) offset += waveTotals[w];
	for (int k=0 ; k<C ; k++)
		{ if (base + k < L) lds[base+k] += offset; }
	__syncthreads ();
	}

// ------------------------------------------------------------- sliding sum ----
// out[c] = (sum of v over [c-lft, c+rgt] inside the vector) / denom, with
// rgt = hOff = (W-1)/2 and lft = W-1-hOff (sum.c:436-455: the running sum after
// step ix holds v[ix-W+1 .. ix] and is stored at centre ix-hOff).
__global__ __launch_bounds__(SU_THREADS)
void sliding_sum_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t ntiles,
                         int tile, int W, int lft, double denom, int C)
	{
	extern __shared__ __attribute__((aligned(16))) double suLds[];
	__shared__ double waveTotals[SU_THREADS/64];

	const uint32_t t         = gdsp_xcd_tile (blockIdx.x, ntiles);
	const int64_t  tileStart = (int64_t) t * tile;
	// stage one extra leading zero so that P[-1] = 0 needs no special case
	const int      sh        = ((lft + 1) & 1);
	const int64_t  g0        = tileStart - lft - 1 - sh;          // even
	const int      L         = (tile + W + sh + 1) & ~1;

	gdsp_stage_f64<SU_THREADS> (suLds, in, n, g0, L, 0.0);
	__syncthreads ();
	// x[k] = staged value of global tileStart-lft-1+k, k=0 is the extra leading element;
	// it must not count towards the first window, so it is cleared before the scan
	if (threadIdx.x == 0) { for (int k=0 ; k<=sh ; k++) suLds[k] = 0.0; }
	__syncthreads ();

	block_prefix_sum (suLds, L, C, waveTotals);

	const double* P = suLds + sh;                 // P[k] = sum of staged x[1..k]
	for (int o = 2*threadIdx.x ; o < tile ; o += 2*SU_THREADS)
		{
		const int64_t g = tileStart + o;
		if (g >= (int64_t) n) break;
		double s0 = (P[o + W]     - P[o])     / denom;
		double s1 = (P[o + 1 + W] - P[o + 1]) / denom;
		if (g + 1 < (int64_t) n) gdsp_st2 (reinterpret_cast<double2*> (out + g), make_double2 (s0, s1));
		else                     out[g] = s0;
		}
	}

// ---------------------------------------------------- sliding sum, block form ----
// Windows of 17 .. SLB_MAX_W bases.  The staged tile is cut into blocks of 16: a window [a,b] is the tail
// of a's block + the whole blocks between + the head of b's block.  One thread owns a block as the right
// end b of 16 windows: prefix sums of its own block in registers, block totals through LDS, then one
// backward walk over the 16+dr elements that hold the 16 left ends, finishing one output per step.  The
// blocks between are added one by one (up to 8 of them) or taken as the difference of two entries of a
// running sum over the tile's 256 block totals.  Same LDS image (pitch 17) as the extrema block form;
// about a third of the LDS traffic of the in-place prefix form above.
#define SLB_THREADS 256
#define SLB_G       16
#define SLB_PITCH   17
#define SLB_ELEMS   (SLB_THREADS * SLB_G)
#define SLB_MIN_W   17
#define SLB_MAX_W   2048

__global__ __launch_bounds__(SLB_THREADS)
void sliding_blocks_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t ntiles,
                            int rgt, int dq, int dr, int sh, double denom)
	{
	__shared__ __attribute__((aligned(16))) double lds[SLB_THREADS * SLB_PITCH];
	__shared__ double blockTot[SLB_THREADS + 1];              // [k+1] = total of blocks 0..k once scanned
	__shared__ double waveTot[SLB_THREADS/64];
	const int    haloL = dq + 1;
	const int    outs  = (SLB_THREADS - haloL) * SLB_G - 2*sh;
	const int    nt    = dq - 1;                              // whole blocks always between
	const int    lead  = haloL * SLB_G - rgt + sh;
	const uint32_t tile = gdsp_xcd_tile (blockIdx.x, ntiles);
	const int64_t  out0 = (int64_t) tile * outs;
	const int64_t  e0   = out0 - lead;
	const int      p    = threadIdx.x, lane = p & 63, wave = p >> 6;

	if ((e0 >= 0) && (e0 + SLB_ELEMS <= (int64_t) n))
		{
		const double2* src = reinterpret_cast<const double2*> (in + e0);
		double2 r[SLB_G/2];
#pragma unroll
		for (int u=0 ; u<SLB_G/2 ; u++) r[u] = gdsp_ld2 (&src[u*SLB_THREADS + p]);
#pragma unroll
		for (int u=0 ; u<SLB_G/2 ; u++)
			{
			const int e = 2 * (u*SLB_THREADS + p);
			double* dst = lds + e + (e >> 4);
			dst[0] = r[u].x;  dst[1] = r[u].y;
			}
		}
	else
		{
		for (int e=p ; e<SLB_ELEMS ; e+=SLB_THREADS)
			{
			const int64_t g = e0 + e;
			lds[e + (e >> 4)] = ((g >= 0) && (g < (int64_t) n))? in[g] : 0.0;
			}
		}
	__syncthreads ();

	double P[SLB_G];
	double run = 0.0;
		{
		const double* xb = lds + p * SLB_PITCH;
#pragma unroll
		for (int u=0 ; u<SLB_G ; u++) { run += xb[u];  P[u] = run; }
		}
	// running sum over the block totals (only consulted when more than 8 blocks lie between)
	double incl = run;
	for (int d=1 ; d<64 ; d*=2) { const double up = __shfl_up (incl, d, 64);  if (lane >= d) incl += up; }
	if (lane == 63) waveTot[wave] = incl;
	if (p == 0) blockTot[0] = 0.0;
	__syncthreads ();
		{
		double before = 0.0;
		for (int w=0 ; w<wave ; w++) before += waveTot[w];
		blockTot[p + 1] = (nt > 8)? before + incl : run;        // short stretches: the plain totals, added one by one
		}
	__syncthreads ();

	if (p >= haloL)
		{
		double T = 0.0;                                           // whole blocks p-nt .. p-1
		if (nt > 8) T = blockTot[p] - blockTot[p - nt];
		else        { for (int d=nt ; d>=1 ; d--) T += blockTot[p - d + 1]; }
		const double* lb = lds + (p - dq) * SLB_PITCH;
		const double* la = lb - SLB_PITCH + SLB_G;
		double s = 0.0;
#pragma unroll
		for (int u=2*SLB_G-2 ; u>=0 ; u--)
			{
			if (u >= SLB_G + dr) continue;
			if (u == dr - 1) { T += s;  s = 0.0; }
			const int rel = u - dr;
			s += (rel >= 0)? lb[rel] : la[rel];
			if (u < SLB_G) P[u] = ((s + T) + P[u]) / denom;
			}
		}
	__syncthreads ();
	if (p >= haloL)
		{
		double* mine = lds + (p - haloL) * SLB_PITCH;
#pragma unroll
		for (int u=0 ; u<SLB_G ; u++) mine[u] = P[u];
		}
	__syncthreads ();
	if (out0 + outs <= (int64_t) n)
		{
		double2* dst = reinterpret_cast<double2*> (out + out0);
		for (int q=p ; q<outs/2 ; q+=SLB_THREADS)
			{
			const int o = 2*q + sh, o1 = o + 1;
			gdsp_st2 (&dst[q], make_double2 (lds[o + (o >> 4)], lds[o1 + (o1 >> 4)]));
			}
		}
	else
		{
		for (int q=p ; q<outs ; q+=SLB_THREADS)
			{
			const int o = q + sh;
			if (out0 + q < (int64_t) n) out[out0 + q] = lds[o + (o >> 4)];
			}
		}
	}

// -------------------------------------------------------------- window sum ----
#define WS_TILE_MAX_W 1024               // windows up to this long go through the tiled kernel, longer ones through the rows kernel
#ifndef WS_TILE_BASES
#define WS_TILE_BASES 8192               // bases staged per workgroup (whole windows only)
#endif

// Tiled form: a workgroup stages K whole windows with coalesced 16-byte loads, one thread
// then sums one window from LDS in ascending order (bit-identical to sum.c:230-249), and
// the tile is rewritten with coalesced 16-byte stores.  Rows are padded to an odd pitch so
// that lanes walking different windows hit different LDS banks.
__global__ __launch_bounds__(SU_THREADS)
void window_sum_tile_kernel (double* __restrict__ v, uint32_t n, uint32_t W, uint32_t K, uint32_t ntiles,
                             double denom, int useActual, double zeroVal)
	{
	extern __shared__ __attribute__((aligned(16))) double suLds[];
	const uint32_t pitch = W | 1;                         // odd
	double*        res   = suLds + (size_t) K * pitch;    // K results
	__shared__ uint32_t inOrder;                          // windows (bit w) that have to be added in ascending order
	if (threadIdx.x == 0) inOrder = 0;
	const uint32_t tile  = gdsp_xcd_tile (blockIdx.x, ntiles);
	const uint64_t base  = (uint64_t) tile * K * W;
	const uint32_t span  = (uint32_t) ((base + (uint64_t) K*W <= n)? K*W : n - base);   // bases in this tile

	// a full tile that starts on a 16-byte boundary moves as 16-byte words, eight loads per lane in flight:
	// thread handles the pairs of bases (2t, 2t+1), +512, ...; (row, col) advance without divisions
	const bool wide = (span == K*W) && ((span & 1) == 0) && ((base & 1) == 0) && (W >= 2);
	if (wide)
		{
		uint32_t row = (2*threadIdx.x) / W, col = (2*threadIdx.x) % W;
		const uint32_t dRow = (2*SU_THREADS) / W, dCol = (2*SU_THREADS) % W;
		const double2* src = reinterpret_cast<const double2*> (v + base);
		const uint32_t np  = span / 2;
		for (uint32_t q0=threadIdx.x ; q0<np ; q0+=8*SU_THREADS)
			{
			double2 x[8];
#pragma unroll
			for (int u=0 ; u<8 ; u++)
				{ const uint32_t q = q0 + u*SU_THREADS;  x[u] = gdsp_ld2 (&src[(q < np)? q : np-1]); }
#pragma unroll
			for (int u=0 ; u<8 ; u++)
				{
				const uint32_t q = q0 + u*SU_THREADS;
				if (q < np)
					{
					suLds[(size_t) row * pitch + col] = x[u].x;
					if (col + 1 < W) suLds[(size_t) row * pitch + col + 1] = x[u].y;
					else             suLds[(size_t) (row + 1) * pitch]     = x[u].y;
					}
				row += dRow;  col += dCol;
				if (col >= W) { col -= W;  row++; }
				}
			}
		}
	uint32_t row = threadIdx.x / W, col = threadIdx.x % W;
	const uint32_t dRow = SU_THREADS / W, dCol = SU_THREADS % W;
	for (uint32_t p0=threadIdx.x ; !wide && (p0<span) ; p0+=4*SU_THREADS)
		{
		double x[4];
#pragma unroll
		for (int u=0 ; u<4 ; u++)
			{ const uint32_t p = p0 + u*SU_THREADS;  x[u] = v[base + ((p < span)? p : span-1)]; }
#pragma unroll
		for (int u=0 ; u<4 ; u++)
			{
			const uint32_t p = p0 + u*SU_THREADS;
			if (p < span) suLds[(size_t) row * pitch + col] = x[u];
			row += dRow;  col += dCol;
			if (col >= W) { col -= W;  row++; }
			}
		}
	__syncthreads ();

	// One thread sums one window, in ascending order: the reference's order, hence its bits.  With few, long windows per
	// tile (W > 256: at most 15) that leaves all but a handful of lanes idle behind a chain of W dependent adds -- unless
	// the order cannot matter: when every base of the window is a multiple of 2^-20 below 2^19 in magnitude (read depth,
	// counts, anything an interval file of small integers or dyadic fractions adds up to), every partial sum of up to 8192
	// of them is exactly representable (< 2^32 at a resolution of 2^-20: 52 bits), so any order gives the sequential
	// result bit for bit.  A wave checks that while it adds the window lane-parallel; the windows that fail the check are
	// then added in order, one lane each.
	const bool coop = (W > 256);                          // (at most 15 windows in the tile)
	if (coop)
		{
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		for (uint32_t w=wave ; w<K ; w+=SU_THREADS/64)
			{
			const uint64_t s = (uint64_t) w * W;
			if (s >= span) break;
			const uint32_t len = (uint32_t) ((s + W <= span)? W : span - s);
			const double*  x   = suLds + (size_t) w * pitch;
			double part = -0.0;                                     // (so that a window of nothing but -0 sums to -0, as it does in order)
			bool   ok   = true;
			for (uint32_t k=lane ; k<len ; k+=64)
				{
				const double t = x[k], scaled = t * 1048576.0;
				ok = ok && (fabs (t) < 524288.0) && (scaled == rint (scaled));
				part += t;
				if (__builtin_amdgcn_ballot_w64 (!ok) != 0) break;   // (real-valued data leaves after the first 64 bases)
				}
			const bool exact = (__builtin_amdgcn_ballot_w64 (!ok) == 0);
			for (int off=32 ; off>0 ; off>>=1) part += __shfl_down (part, off, 64);
			if (lane == 0)
				{
				if (exact) res[w] = useActual? part / (double) len : part / denom;   // no rounding anywhere: the ascending sum
				else       atomicOr (&inOrder, 1u << w);
				}
			}
		__syncthreads ();
		}
	// (the windows left for the ascending order share one wave's lanes: a chain of dependent adds occupies the FP64 pipe
	// the same whether one lane or sixty-four are active)
	for (uint32_t w=threadIdx.x ; w<K ; w+=SU_THREADS)
		{
		const uint64_t s = (uint64_t) w * W;
		if (s >= span) break;
		if (coop && (((inOrder >> w) & 1) == 0)) continue;
		const uint32_t len = (uint32_t) ((s + W <= span)? W : span - s);
		const double*  x   = suLds + (size_t) w * pitch;
		// The adds stay one after the other in the reference's order; what can be hidden is the latency of the LDS
		// reads: sixteen values are fetched while the sixteen before them are being added (with one short batch in
		// flight a 1000-base window spent more time waiting for LDS than adding).
		double acc = x[0];
		uint32_t k = 1;
		if (len >= 1 + 16)
			{
			double A[16], B[16];
#pragma unroll
			for (int u=0 ; u<16 ; u++) A[u] = x[k+u];
			k += 16;
			while (k + 32 <= len)
				{
#pragma unroll
				for (int u=0 ; u<16 ; u++) B[u] = x[k+u];
#pragma unroll
				for (int u=0 ; u<16 ; u++) acc += A[u];
#pragma unroll
				for (int u=0 ; u<16 ; u++) A[u] = x[k+16+u];
#pragma unroll
				for (int u=0 ; u<16 ; u++) acc += B[u];
				k += 32;
				}
#pragma unroll
			for (int u=0 ; u<16 ; u++) acc += A[u];
			}
		for ( ; k<len ; k++) acc += x[k];
		res[w] = useActual? acc / (double) len : acc / denom;
		}
	__syncthreads ();

	if (wide)
		{
		uint32_t r2 = (2*threadIdx.x) / W, c2 = (2*threadIdx.x) % W;
		const uint32_t dR2 = (2*SU_THREADS) / W, dC2 = (2*SU_THREADS) % W;
		double2* dst = reinterpret_cast<double2*> (v + base);
		for (uint32_t q=threadIdx.x ; q<span/2 ; q+=SU_THREADS)
			{
			const double a = (c2 == 0)? res[r2] : zeroVal;
			const double b = (c2 + 1 == W)? res[r2 + 1] : zeroVal;   // (the next window starts on the odd base)
			gdsp_st2 (&dst[q], make_double2 (a, b));
			r2 += dR2;  c2 += dC2;
			if (c2 >= W) { c2 -= W;  r2++; }
			}
		return;
		}
	row = threadIdx.x / W;  col = threadIdx.x % W;
	for (uint32_t p=threadIdx.x ; p<span ; p+=SU_THREADS)
		{
		v[base + p] = (col == 0)? res[row] : zeroVal;
		row += dRow;  col += dCol;
		if (col >= W) { col -= W;  row++; }
		}
	}

// Windows of 1025..8192 bases, first pass: one wave per window reads it with coalesced 16-byte loads, adds it
// lane-parallel and checks, as window_sum_tile_kernel does, that no partial sum of it can round (every base a multiple
// of 2^-20 below 2^19 in magnitude).  Such a window is rewritten on the spot -- any order is the reference's sum bit for
// bit -- and flagged done; a window that fails is left untouched for window_sum_rows_kernel (real-valued data fails
// within the first 128 bases of each window: one kilobyte read in vain per window).
#define WX_THREADS 256
#define WX_BATCH   8

__global__ __launch_bounds__(WX_THREADS)
void window_sum_exact_kernel (double* __restrict__ v, uint32_t n, uint32_t W, uint32_t nwin,
                              double denom, int useActual, double zeroVal, unsigned char* __restrict__ done)
	{
	const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint64_t w    = (uint64_t) blockIdx.x * (WX_THREADS/64) + wave;
	if (w >= nwin) return;
	const uint64_t s  = w * W;
	const uint64_t e  = (s + W <= n)? s + W : n;
	const uint64_t p0 = (s + 1) / 2, p1 = e / 2;               // whole 16-byte words [p0, p1) of the window
	double2*       words = reinterpret_cast<double2*> (v);

	double part = -0.0;                                        // (a window of nothing but -0 sums to -0)
	bool   ok   = true;
	auto take = [&] (double t)
		{
		const double scaled = t * 1048576.0;
		ok = ok && (fabs (t) < 524288.0) && (scaled == rint (scaled));
		part += t;
		};
	if ((lane == 0) && (s & 1)) take (v[s]);                   // a base before the first whole word
	if ((lane == 1) && (e & 1)) take (v[e-1]);                 // ... and one after the last (e-1 = s only when s is even)
	uint64_t p = p0 + lane;
	if (p < p1) { const double2 x = words[p];  take (x.x);  take (x.y); }   // a first, short look: real-valued data stops here
	p += 64;
	while ((__builtin_amdgcn_ballot_w64 (!ok) == 0) && (__builtin_amdgcn_ballot_w64 (p < p1) != 0))
		{
		double2 x[WX_BATCH];
#pragma unroll
		for (int u=0 ; u<WX_BATCH ; u++)
			{
			const uint64_t q = p + 64*u;
			x[u] = gdsp_ld2 (&words[(q < p1)? q : p1 - 1]);                     // (p1 > p0 here; clamped, the extra copies are not added)
			}
#pragma unroll
		for (int u=0 ; u<WX_BATCH ; u++)
			{ if (p + 64*u < p1) { take (x[u].x);  take (x[u].y); } }
		p += 64*WX_BATCH;
		}
	if (__builtin_amdgcn_ballot_w64 (!ok) != 0)
		{ if (lane == 0) done[w] = 0;  return; }

	for (int off=32 ; off>0 ; off>>=1) part += __shfl_down (part, off, 64);
	part = __shfl (part, 0, 64);
	const double val = useActual? part / (double) (e - s) : part / denom;
	if ((lane == 0) && (s & 1)) v[s]   = val;
	if ((lane == 1) && (e & 1)) v[e-1] = (e - 1 == s)? val : zeroVal;
	for (uint64_t q=p0+lane ; q<p1 ; q+=64)
		gdsp_st2 (&words[q], make_double2 ((2*q == s)? val : zeroVal, zeroVal));
	if (lane == 0) done[w] = 1;
	}

// ---- long windows (W > WS_TILE_MAX_W): one LANE per window, one 128-byte line of every window at a time.
// In the tiled kernel a 1000-base window keeps one lane busy for a thousand dependent adds while the other 252
// threads of its workgroup wait, and LDS holds only a handful of whole windows per CU.  Here a wave takes 64
// consecutive windows and walks them in step, line by line: per stage it fetches the next 128-byte aligned line of
// each window (eight 16-byte loads per line, eight windows per load instruction), turns it through LDS, and every
// lane adds the bases of its own window in that line in the reference's order (sum.c:230-249: bit-identical; a
// window's first and last line are partial).  All 64 lanes add, the next stage's loads are in flight meanwhile, and
// a wave needs 17 KiB of LDS however long the windows are.  A line that has landed in LDS is overwritten with
// zeroVal in HBM straight away -- whole lines, so the memory side never has to merge a partial one, except at the
// two ends of a window -- and the sum goes to the window's first slot at the end.
#define WR_PIECE   16                       // bases per window per stage: one 128-byte line (two lines a stage and half the waves: slower)
#define WR_PITCH   (WR_PIECE + 1)
#define WR_THREADS 256                      // four waves: 4 x 2 x 64 x 17 x 8 B = 68 KiB of LDS per workgroup
#define WR_LPR     (WR_PIECE / 2)           // lanes (16-byte words) per window per load instruction
#define WR_RPI     (64 / WR_LPR)            // windows per load instruction
#define WR_LOADS   (64 / WR_RPI)            // load instructions per stage

__global__ __launch_bounds__(WR_THREADS)
void window_sum_rows_kernel (double* __restrict__ v, uint32_t n, uint32_t W, uint32_t nwin,
                             double denom, int useActual, double zeroVal, const unsigned char* __restrict__ done)
	{
	__shared__ __attribute__((aligned(16))) double stage[WR_THREADS/64][2][64 * WR_PITCH];
	const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint64_t w0   = ((uint64_t) blockIdx.x * (WR_THREADS/64) + wave) * 64;      // first window of this wave
	if (w0 >= nwin) return;                                        // (whole waves only: no workgroup barrier below)
	const uint64_t myWin   = w0 + lane;
	// windows that window_sum_exact_kernel has already rewritten count as empty here: nothing of theirs is added or stored
	const bool     mine    = (myWin < nwin) && (done[myWin] == 0);
	const uint64_t todo    = __builtin_amdgcn_ballot_w64 (mine);   // bit r: window w0+r is still to be summed
	if (todo == 0) return;
	const uint64_t myStart = myWin * W;
	const uint64_t myEnd   = (!mine)? myStart : ((myStart + W <= n)? myStart + W : n);
	const uint64_t myLine0 = myStart / WR_PIECE;
	const uint32_t nstages = W / WR_PIECE + 2;                     // pieces a window can touch
	double2*       base    = reinterpret_cast<double2*> (v);
	const uint64_t lastPair = ((uint64_t) n - 1) / 2;

	// the window and the 16-byte word of its piece that this lane fetches in load instruction j
	const int myRowInGroup = lane / WR_LPR, myWord = lane % WR_LPR;
	double2 fetched[WR_LOADS];
	auto issue = [&] (uint32_t st)
		{
#pragma unroll
		for (int j=0 ; j<WR_LOADS ; j++)
			{
			const uint64_t line0 = ((w0 + (uint64_t) (WR_RPI*j + myRowInGroup)) * W) / WR_PIECE;
			uint64_t pair = (line0 + st) * (WR_PIECE/2) + myWord;
			if (pair > lastPair) pair = lastPair;                  // clamped: what lies past the data is never added
			fetched[j] = base[pair];
			}
		};
	auto land = [&] (int buf, uint32_t st)
		{
#pragma unroll
		for (int j=0 ; j<WR_LOADS ; j++)
			{
			double* dst = &stage[wave][buf][(WR_RPI*j + myRowInGroup) * WR_PITCH + 2*myWord];
			dst[0] = fetched[j].x;  dst[1] = fetched[j].y;
			// the bases of this piece that belong to the window (all but its first slot) become zeroVal
			const uint64_t win   = w0 + (uint64_t) (WR_RPI*j + myRowInGroup);
			const uint64_t start = win * W;
			const uint64_t end   = (((todo >> (WR_RPI*j + myRowInGroup)) & 1) == 0)? start : ((start + W <= n)? start + W : n);
			const uint64_t e0    = (start / WR_PIECE + st) * WR_PIECE + 2*myWord;
			const bool in0 = (e0     > start) && (e0     < end);
			const bool in1 = (e0 + 1 > start) && (e0 + 1 < end);
			if (in0 && in1) base[e0 / 2] = make_double2 (zeroVal, zeroVal);
			else if (in0)   v[e0] = zeroVal;
			else if (in1)   v[e0 + 1] = zeroVal;
			}
		};

	double acc = 0.0;
	issue (0);
	for (uint32_t st=0 ; st<nstages ; st++)
		{
		const int buf = st & 1;
		land (buf, st);                                            // (waits for the loads of this stage)
		if (st + 1 < nstages) issue (st + 1);                      // next stage's loads fly while this one is added
		__builtin_amdgcn_wave_barrier ();
		const uint64_t lineLo = (myLine0 + st) * WR_PIECE;
		const double*  x      = &stage[wave][buf][lane * WR_PITCH];
		if ((lineLo > myStart) && (lineLo + WR_PIECE <= myEnd))    // a whole piece inside the window
			{
#pragma unroll
			for (int h=0 ; h<WR_PIECE ; h+=16)
				{
				double t[16];
#pragma unroll
				for (int u=0 ; u<16 ; u++) t[u] = x[h+u];
#pragma unroll
				for (int u=0 ; u<16 ; u++) acc += t[u];
				}
			}
		else
			{
			for (int u=0 ; u<WR_PIECE ; u++)
				{
				const uint64_t e = lineLo + u;
				if (e == myStart) { if (e < myEnd) acc = x[u]; }  // sum.c:231-236: the window's first base starts the sum
				else if ((e > myStart) && (e < myEnd)) acc += x[u];
				}
			}
		__builtin_amdgcn_wave_barrier ();                          // (the buffer is rewritten two stages on)
		}
	if (myEnd > myStart) v[myStart] = useActual? acc / (double) (myEnd - myStart) : acc / denom;
	}

#define WS_SEQ_MAX 8192
// one thread per window, ascending adds (bit-identical to sum.c:230-249)
__global__ __launch_bounds__(SU_THREADS)
void window_sum_seq_kernel (double* __restrict__ v, uint32_t n, uint32_t W, uint32_t nwin,
                            double denom, int useActual, double zeroVal)
	{
	const uint32_t w = blockIdx.x * SU_THREADS + threadIdx.x;
	if (w >= nwin) return;
	const uint64_t s = (uint64_t) w * W;
	uint64_t       e = s + W;  if (e > n) e = n;
	double acc = v[s];
	for (uint64_t ix=s+1 ; ix<e ; ix++) acc += v[ix];
	v[s] = useActual? acc / (double) (e - s) : acc / denom;
	for (uint64_t ix=s+1 ; ix<e ; ix++) v[ix] = zeroVal;
	}

// one workgroup per (long) window: strided partial sums, then a tree
__global__ __launch_bounds__(SU_THREADS)
void window_sum_wide_kernel (double* __restrict__ v, uint32_t n, uint32_t W,
                             double denom, int useActual, double zeroVal)
	{
	__shared__ double part[SU_THREADS];
	const uint64_t s = (uint64_t) blockIdx.x * W;
	uint64_t       e = s + W;  if (e > n) e = n;
	double acc = 0.0;
	for (uint64_t ix=s+threadIdx.x ; ix<e ; ix+=SU_THREADS) acc += v[ix];
	part[threadIdx.x] = acc;
	__syncthreads ();
	for (int d=SU_THREADS/2 ; d>0 ; d>>=1)
		{
		if ((int) threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
		__syncthreads ();
		}
	const double total = part[0];
	for (uint64_t ix=s+threadIdx.x ; ix<e ; ix+=SU_THREADS)
		v[ix] = (ix == s)? (useActual? total / (double) (e - s) : total / denom) : zeroVal;
	}

// ---------------------------------------------------------- cumulative sum ----
// Reduce-then-scan in three launches: chunk totals (8 B/base read), an exclusive scan of
// the totals (tiny), then a local scan of every chunk plus its offset (8 B read + 8 B
// write): 24 B/base of traffic for the 16 B/base the operator needs.  All global accesses
// are 16 bytes per lane.
#define CS_CHUNK 4096                     // bases per workgroup (32 KiB of LDS in the scan pass)
#define CS_PER   (CS_CHUNK / SU_THREADS)  // 16 bases per thread

__global__ __launch_bounds__(SU_THREADS)
void cumsum_totals_kernel (const double* __restrict__ v, uint32_t n, uint32_t nchunks, double* __restrict__ totals)
	{
	__shared__ double part[SU_THREADS/64];
	const uint32_t chunk = gdsp_xcd_tile (blockIdx.x, nchunks);
	const uint64_t s     = (uint64_t) chunk * CS_CHUNK;
	double acc = 0.0;
	if (s + CS_CHUNK <= n)
		{
		const double2* p = reinterpret_cast<const double2*> (v + s) + threadIdx.x;
		double2 d[CS_PER/2];
#pragma unroll
		for (int u=0 ; u<CS_PER/2 ; u++) d[u] = gdsp_ld2 (&p[u*SU_THREADS]);
#pragma unroll
		for (int u=0 ; u<CS_PER/2 ; u++) acc += d[u].x + d[u].y;
		}
	else
		{
		for (uint64_t ix=s+threadIdx.x ; ix<n ; ix+=SU_THREADS) acc += v[ix];
		}
	for (int off=32 ; off>0 ; off>>=1) acc += __shfl_down (acc, off, 64);
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
	__syncthreads ();
	if (threadIdx.x == 0) totals[chunk] = (part[0] + part[1]) + (part[2] + part[3]);
	}

// exclusive scan of the chunk totals, one workgroup of 1024: tiles of 8192 totals, eight consecutive ones per thread
// (64 contiguous bytes per lane, so the loads coalesce; a thread walking its own far-apart slice of the array, as this
// kernel first did, spends ~90 us on dependent loads for 35 k chunks), lanes, then waves, then a carry from tile to tile
#define CO_THREADS 1024
#define CO_PER     8
__global__ __launch_bounds__(CO_THREADS)
void cumsum_offsets_kernel (double* __restrict__ totals, uint32_t nchunks)
	{
	__shared__ double part[CO_THREADS/64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double carry = 0.0;
	for (uint32_t t0=0 ; t0<nchunks ; t0+=CO_THREADS*CO_PER)
		{
		const uint32_t c0 = t0 + threadIdx.x * CO_PER;
		double T[CO_PER];
#pragma unroll
		for (int i=0 ; i<CO_PER ; i++) T[i] = (c0 + i < nchunks)? totals[c0 + i] : 0.0;
		double sum = 0.0;
#pragma unroll
		for (int i=0 ; i<CO_PER ; i++) sum += T[i];
		double incl = sum;
		for (int d=1 ; d<64 ; d*=2)
			{
			const double up = __shfl_up (incl, d, 64);
			if (lane >= d) incl = up + incl;
			}
		double excl = __shfl_up (incl, 1, 64);
		if (lane == 0) excl = 0.0;
		if (lane == 63) part[wave] = incl;
		__syncthreads ();
		double run = carry, all = carry;
		for (int w=0 ; w<CO_THREADS/64 ; w++) { if (w < wave) run += part[w];  all += part[w]; }
		__syncthreads ();
		carry = all;
		run += excl;
#pragma unroll
		for (int i=0 ; i<CO_PER ; i++)
			{
			if (c0 + i < nchunks) totals[c0 + i] = run;
			run += T[i];
			}
		}
	}

__global__ __launch_bounds__(SU_THREADS)
void cumsum_apply_kernel (double* __restrict__ v, uint32_t n, uint32_t nchunks, const double* __restrict__ offsets)
	{
	__shared__ __attribute__((aligned(16))) double csLds[CS_CHUNK + 2];
	__shared__ double waveTotals[SU_THREADS/64];
	const uint32_t chunk = gdsp_xcd_tile (blockIdx.x, nchunks);
	const int64_t  s     = (int64_t) chunk * CS_CHUNK;
	gdsp_stage_f64<SU_THREADS> (csLds, v, n, s, CS_CHUNK, 0.0);
	__syncthreads ();
	block_prefix_sum (csLds, CS_CHUNK, CS_PER + 1, waveTotals);   // 17 values per thread (odd stride)
	const double off = offsets[chunk];
	if (s + CS_CHUNK <= (int64_t) n)
		{
		double2*       dst = reinterpret_cast<double2*> (v + s);
		const double2* src = reinterpret_cast<const double2*> (csLds);
#pragma unroll
		for (int u=0 ; u<CS_PER/2 ; u++)
			{
			double2 d = src[threadIdx.x + u*SU_THREADS];
			d.x = off + d.x;  d.y = off + d.y;
			gdsp_st2 (&dst[threadIdx.x + u*SU_THREADS], d);
			}
		}
	else
		{
		for (int p=threadIdx.x ; p<CS_CHUNK ; p+=SU_THREADS)
			{ if (s + p < (int64_t) n) v[s+p] = off + csLds[p]; }
		}
	}

// (A single pass -- 16 B/base instead of 24 -- was built twice in round 2 and dropped.  It needs every chunk's prefix
// while the chunk is still on chip, i.e. a decoupled look-back; to keep the rounding independent of timing the
// association was fixed by the chunk index (aggregates of aligned groups only, published as one 8-byte word that is
// value and flag, waited for in ticket order).  Bit-identical results, but each hop through an agent-scope atomic
// store and its polling loads costs microseconds on this part: a workgroup per 4096-base chunk with a binary tree took
// 1.41 ms per 249 Mbp, a wave per 1024-base chunk with a 32-ary tree 2.83 ms, against 1.05 ms for the three launches
// below; with the waits removed (wrong sums) the first form ran in 0.81 ms, so even free waits would buy little.
// Round 3 tried it once more with a flat scheme -- groups of 512 chunks, subgroups of 32, a chunk polling one word per
// lane (its subgroup's earlier totals, its group's earlier subgroup sums, the running sum in front of its group from one
// of 64 copies) so that a chunk waits for ONE hop of the chain over groups: bit-identical again, 2.49 ms.  What it
// showed on the way: handing chunks out through one ticket counter costs 4.3 ms by itself (60 k atomic adds with return
// on one word queue at its memory channel, 40-70 ns each); 31 M uncached 8-byte polls cost as much (the memory side
// serves ~5 requests per ns, whatever their size); and with both gone a hop still takes ~20 us, because every small
// request of the chain queues behind the streaming traffic of 1024 workgroups while LDS capacity caps what is in
// flight at 32 MiB -- two groups.  An order-free sum (the chunk totals kept exactly, in a wide fixed-point word, so
// that any look-back grouping rounds to the same double) is the form that could still work; not built.)

// ---------------------------------------------------- cumulative sum, one pass ----
// Round 5.  What sank the three earlier single-pass forms (above) was (a) a SERIAL chain -- a group's running sum handed
// from group to group, one hop of ~20 us under load per hop -- and (b) the chunk waiting for its prefix in LDS, which
// caps what the chip holds in flight at 32 MiB.  This form has neither:
//   * no chain.  Chunk c = (a, b, j) (4096 chunks of 8192 bases per super-group a, 64 per group b) needs
//         prefix(c) = sum of S[a' < a]  +  sum of G[a][b' < b]  +  sum of T[a][b][j' < j]
//     T = a chunk's own total, G = a group's, S = a super-group's.  Every chunk publishes T as soon as it has it; the
//     LAST chunk of a group, which reads the other 63 totals anyway, publishes G, the last chunk of a super-group S.  A
//     chunk's wait is therefore at most three hops deep whatever its place in the vector, and one in 63 cases of 64.
//   * the association is FIXED: each of the three sums is a butterfly over the 64 lanes that fetched its terms (absent
//     terms are +0), the three are added in one order, and a chunk's own scan is in registers in one order -- the bits do
//     not depend on timing, run to run, and on exactly summable vectors (read depth) they are the reference's.
//   * the chunk waits in REGISTERS (16 values a thread, loaded 16 bytes a lane, coalesced; its scans are DPP moves, no
//     LDS round trips): 76 registers -> six waves per SIMD = three workgroups of 512 per CU, 48 MiB in flight chip-wide,
//     and LDS holds 68 words.
// A published word is value and flag at once (one 8-byte store / load at agent scope, no fence): the work area is filled
// with the one bit pattern no total can take (an all-ones NaN; a total that comes out as exactly that is published as the
// default NaN).  Workgroups take chunks in launch order -- the dispatcher hands them out in that order, so whatever a
// chunk waits for was dispatched before it: no ticket counter (4.3 ms by itself in round 3).
// 16 B/base moved (24 in the three launches above).
// Measured, 249 Mbp (tools/bench_scan.py, tools/exp_scan.sh; profiles/r05_cumsum_forms.txt): **0.751 ms = 0.66 of HBM**
// on 16 B/base, read depth and real values alike, against 1.016 ms for the three launches (GDSP_CUMSUM=3 keeps them) and
// 0.739 ms with the waits compiled out (-DCL_NOWAIT: wrong sums).  The first form -- chunks of 4096 bases, 256 threads --
// took 0.946 ms (0.915 with one lane waiting for the newest term before the 64-lane fetch) against 0.725 without waits,
// whatever the number of workgroups per CU (4 or 6: 0.944 / 0.946 ms; 8, with spills, 1.12): a chunk's terms come from
// its contemporaries, which publish when it does, so it waits about one store-to-load round trip behind its own loads --
// per CHUNK, so half as many chunks of twice the size pay half of it, and nearly all of the rest hides.  Two forms built
// to take the wait out of the data's way were slower (measured on the 4096-base chunks): a tree of radix 16 in which a group's first chunk publishes the prefix in front of the group (a
// chunk fetches <= 15 totals and one word: fewer polls, one hop deeper) 0.982 ms; every chunk visited twice by
// workgroups a fixed lag apart, the first adding it up and publishing, the second -- its terms long there -- reading it
// again out of the Infinity Cache and writing it: 1.04-1.05 ms at lags of 256 / 1024 / 4096 chunks (the second read
// costs what the first does).
// (Chunks of 8192 bases, 512 threads: with 4096 / 256 -- the same 16 values a thread, twice the chunks -- the pass took
// 0.915 ms per 249 Mbp instead of 0.751 against 0.739 with the waits compiled out: what the waits cost is paid per chunk.)
#ifndef CL_THREADS
#define CL_THREADS 512
#endif
#ifndef CL_CHUNK
#define CL_CHUNK   8192
#endif
#define CL_ROWS    (CL_CHUNK / (2 * CL_THREADS))          // 8 rows of 512 elements: thread t holds elements 512 u + 2 t, + 1
#define CL_GROUP   64
#define CL_SUPER   (CL_GROUP * CL_GROUP)
static_assert (CL_ROWS * (CL_THREADS / 64) <= 64, "one wave scans the (row, wave) totals of a chunk");
#define CL_SENTINEL 0xFFFFFFFFFFFFFFFFull
#ifndef CL_TSTRIDE
#define CL_TSTRIDE 1                                         // 8-byte words from one chunk's total to the next (32: a 256-byte line each)
#endif
#ifndef CL_SLEEP
#define CL_SLEEP 2                                           // x 64 cycles between two tries of a poll
#endif

__device__ __forceinline__ void cl_publish (unsigned long long* slot, double v)
	{
	unsigned long long w = (unsigned long long) __double_as_longlong (v);
	if (w == CL_SENTINEL) w = 0x7FF8000000000000ull;             // (a NaN either way; its payload is nobody's promise)
	__hip_atomic_store (slot, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}

// x of the lane a DPP control names; +0 where that lane does not exist or the destination lane is masked off.  Controls
// (gfx9 encoding): 0x110 + n = row_shr:n (rows of 16 lanes), 0x142 / 0x143 = row_bcast:15 / :31, 0x138 = wave_shr:1
template <int CTRL, int ROWS, int BANKS>
__device__ __forceinline__ double cl_dpp_move0 (double x)
	{
	const int lo = __builtin_amdgcn_update_dpp (0, __double2loint (x), CTRL, ROWS, BANKS, true);
	const int hi = __builtin_amdgcn_update_dpp (0, __double2hiint (x), CTRL, ROWS, BANKS, true);
	return __hiloint2double (hi, lo);
	}

// inclusive sum over the 64 lanes in one fixed order: inside rows of 16 by doubling, then row 0 -> 1 and 2 -> 3, then 0..1 -> 2..3
__device__ __forceinline__ double cl_wave_scan (double x)
	{
	x += cl_dpp_move0<0x111, 0xF, 0xF> (x);
	x += cl_dpp_move0<0x112, 0xF, 0xF> (x);
	x += cl_dpp_move0<0x114, 0xF, 0xF> (x);
	x += cl_dpp_move0<0x118, 0xF, 0xF> (x);
	x += cl_dpp_move0<0x142, 0xA, 0xF> (x);
	x += cl_dpp_move0<0x143, 0xC, 0xF> (x);
	return x;
	}

// the sum of slots[0 .. count) once every one of them has been published, count <= 64: lane i fetches slot i
template <int STRIDE = 1>
__device__ __forceinline__ double cl_gather (const unsigned long long* slots, int count, int lane)
	{
	unsigned long long w = 0;                                    // +0.0
#ifdef CL_NOWAIT                                                 // (timing experiment: wrong sums, no waits)
	count = 0;
#endif
#ifndef CL_NOSPIN1
	// the newest term is the last to arrive: ONE lane waits for it (a 64-byte request per try where the 64-lane fetch is
	// four lines, and every waiting chunk on the chip tries again as soon as its last try has come back), then all fetch
	if ((count > 0) && (lane == 0))
		{
		while (__hip_atomic_load (slots + (size_t) (count - 1) * STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == CL_SENTINEL)
			__builtin_amdgcn_s_sleep (CL_SLEEP);
		}
#endif
	if (lane < count)
		{
		for (;;)
			{
			w = __hip_atomic_load (slots + (size_t) lane * STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (w != CL_SENTINEL) break;
			__builtin_amdgcn_s_sleep (CL_SLEEP);
			}
		}
	double x = __longlong_as_double ((long long) w);
	for (int off=32 ; off>0 ; off>>=1) x += __shfl_xor (x, off, 64);
	return x;
	}

// WHOLE: every chunk of the grid is whole (the launch over the vector's n / 4096 whole chunks); the ragged last chunk, if
// any, is a second launch of one workgroup (chunk0 = its number): its terms are all published by then
#ifndef CL_WAVES
#define CL_WAVES 6                                           // workgroups per CU: 76 registers without a spill
#endif
template <bool WHOLE>
__global__ __launch_bounds__(CL_THREADS) __attribute__((amdgpu_waves_per_eu(CL_WAVES)))
void cumsum_lookback_kernel (double* __restrict__ v, uint32_t n, uint32_t chunk0, unsigned long long* __restrict__ T,
                             unsigned long long* __restrict__ G, unsigned long long* __restrict__ S)
	{
	__shared__ double waveTot[CL_ROWS][CL_THREADS/64];
	__shared__ double part[4];                                   // sums of T, G, S terms; the chunk's own total
	const uint32_t c = chunk0 + blockIdx.x;
	const int      t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const uint64_t s0 = (uint64_t) c * CL_CHUNK;
	constexpr bool whole = WHOLE;

	// ---- the chunk: 8 x 16 bytes per lane, coalesced
	double2 x[CL_ROWS];
	if (whole)
		{
		const double2* src = reinterpret_cast<const double2*> (v + s0);
#pragma unroll
		for (int u=0 ; u<CL_ROWS ; u++) x[u] = gdsp_ld2 (&src[u*CL_THREADS + t]);
		}
	else
		{
#pragma unroll
		for (int u=0 ; u<CL_ROWS ; u++)
			{
			const uint64_t e = s0 + 2 * (uint64_t) (u*CL_THREADS + t);
			x[u].x = (e     < n)? v[e]     : 0.0;
			x[u].y = (e + 1 < n)? v[e + 1] : 0.0;
			}
		}

	// ---- scan inside each row: pairs, then lanes (exclusive of the own pair), by DPP moves -- no LDS round trips
	double before[CL_ROWS];                                      // sum of the row's elements before this thread's pair, own wave only
#pragma unroll
	for (int u=0 ; u<CL_ROWS ; u++)
		{
		const double incl = cl_wave_scan (x[u].x + x[u].y);
		before[u] = cl_dpp_move0<0x138, 0xF, 0xF> (incl);           // wave_shr:1: the lane before's inclusive sum, +0 for lane 0
		if (lane == 63) waveTot[u][wave] = incl;
		}
	__syncthreads ();
	// ---- the 32 (row, wave) totals in order: one scan over 32 lanes of wave 0; its last lane has the chunk's total and
	//      publishes it; then the three gathers, a wave each
	const uint32_t j = c % CL_GROUP, b = (c / CL_GROUP) % CL_GROUP, a = c / CL_SUPER;
	double* const flat = &waveTot[0][0];
	if (wave == 0)
		{
		const double mine  = (lane < CL_ROWS * (CL_THREADS/64))? flat[lane] : 0.0;
		const double incl  = cl_wave_scan (mine);
		const double excl  = cl_dpp_move0<0x138, 0xF, 0xF> (incl);
		const double total = __shfl (incl, CL_ROWS * (CL_THREADS/64) - 1, 64);
		if (lane < CL_ROWS * (CL_THREADS/64)) flat[lane] = excl;   // (every lane has read its own word)
		if (lane == 0) { cl_publish (&T[(size_t) c * CL_TSTRIDE], total);  part[3] = total; }
		const double sumT = cl_gather<CL_TSTRIDE> (T + (size_t) (c - j) * CL_TSTRIDE, (int) j, lane);
		if ((j == CL_GROUP - 1) && (lane == 0)) cl_publish (&G[c / CL_GROUP], sumT + total);
		if (lane == 0) part[0] = sumT;
		}
	else if (wave == 1)
		{
		const double sumG = cl_gather (G + (size_t) a * CL_GROUP, (int) b, lane);
		if (lane == 0) part[1] = sumG;
		}
	else if (wave == 2)
		{
		double sumS = 0.0;
		for (uint32_t a0=0 ; a0<a ; a0+=64)                      // (more than 64 super-groups: a vector beyond 2^30 bases)
			sumS += cl_gather (S + a0, (int) ((a - a0 < 64)? a - a0 : 64), lane);
		if (lane == 0) part[2] = sumS;
		}
	__syncthreads ();
	const double sumT = part[0], sumG = part[1], sumS = part[2];
	if ((t == 0) && (j == CL_GROUP - 1) && (b == CL_GROUP - 1)) cl_publish (&S[a], sumG + (sumT + part[3]));
#pragma unroll
	for (int u=0 ; u<CL_ROWS ; u++) before[u] = waveTot[u][wave] + before[u];
	const double prefix = (sumS + sumG) + sumT;

	// ---- the chunk's values
	if (whole)
		{
		double2* dst = reinterpret_cast<double2*> (v + s0);
#pragma unroll
		for (int u=0 ; u<CL_ROWS ; u++)
			{
			const double base = prefix + before[u];
			double2 o;
			o.x = base + x[u].x;
			o.y = o.x + x[u].y;
			gdsp_st2 (&dst[u*CL_THREADS + t], o);
			}
		}
	else
		{
#pragma unroll
		for (int u=0 ; u<CL_ROWS ; u++)
			{
			const uint64_t e = s0 + 2 * (uint64_t) (u*CL_THREADS + t);
			const double base = prefix + before[u];
			const double o0 = base + x[u].x;
			if (e     < n) v[e]     = o0;
			if (e + 1 < n) v[e + 1] = o0 + x[u].y;
			}
		}
	}

static size_t cl_work_words (uint32_t n)
	{
	const size_t nchunks = ((size_t) n + CL_CHUNK - 1) / CL_CHUNK;
	return nchunks * CL_TSTRIDE + (nchunks + CL_GROUP - 1) / CL_GROUP + (nchunks + CL_SUPER - 1) / CL_SUPER + 8;
	}

// one flag per window for the two passes above, kept per (device, stream): calls on one stream follow one another,
// calls on different streams must not share them
struct WsFlags { int device;  void* stream;  size_t nwin;  unsigned char* d_done; };
static std::vector<WsFlags> wsFlags;
static std::mutex           wsLock;

static int ws_done_flags (void* stream, size_t nwin, unsigned char** out)
	{
	int device = 0;
	GDSP_HIP_TRY (hipGetDevice (&device));
	WsFlags* f = NULL;
	for (WsFlags& x : wsFlags) { if ((x.device == device) && (x.stream == stream)) f = &x; }
	if (f == NULL) { wsFlags.push_back (WsFlags { device, stream, 0, NULL });  f = &wsFlags.back (); }
	if (f->nwin < nwin)
		{
		if (f->d_done != NULL) { GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (stream)));  GDSP_HIP_TRY (hipFree (f->d_done)); }
		f->d_done = NULL;  f->nwin = 0;
		GDSP_HIP_TRY (hipMalloc ((void**) &f->d_done, nwin + 64));
		f->nwin = nwin;
		}
	*out = f->d_done;
	return GDSP_OK;
	}

extern "C" {

int gdsp_sliding_sum (const double* d_in, double* d_out, uint32_t n, uint32_t W, double denom, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_in != NULL) && (d_out != NULL), "NULL vector");
	GDSP_REQUIRE (d_in != d_out, "out-of-place operator: d_out must not alias d_in");
	GDSP_REQUIRE (gdsp_aligned16 (d_in) && gdsp_aligned16 (d_out), "vectors must be 16-byte aligned");
	GDSP_REQUIRE (W >= 1, "window must be >= 1");
	GDSP_REQUIRE (denom != 0.0, "denominator can't be zero");

	// a window longer than the vector plus its own reach sees everything anyway
	uint64_t Weff = W, hOff = (W - 1) / 2, lft = W - 1 - hOff;
	if ((W >= SLB_MIN_W) && (W <= SLB_MAX_W))
		{
		const int d  = (int) W - 1, dq = d / SLB_G, dr = d % SLB_G;
		const int sh = ((dq + 1) * SLB_G - (int) hOff) & 1;
		const int outs = (SLB_THREADS - (dq + 1)) * SLB_G - 2*sh;
		const uint32_t ntiles = (uint32_t) (((uint64_t) n + outs - 1) / outs);
		hipLaunchKernelGGL (sliding_blocks_kernel, dim3(ntiles), dim3(SLB_THREADS), 0, gdsp_stream (stream),
		                    d_in, d_out, n, ntiles, (int) hOff, dq, dr, sh, denom);
		GDSP_LAUNCH_CHECK ();
		return GDSP_OK;
		}
	const int tile = 4096;
	const size_t maxDoubles = 18432;                       // 144 KiB of LDS
	if (Weff + tile + 4 > maxDoubles)
		{
		gdsp_set_error ("gdsp_sliding_sum: window of %u bases exceeds what one LDS tile holds (max %zu)",
		                W, maxDoubles - tile - 4);
		return GDSP_EINVAL;
		}
	const int L = (int) (tile + Weff + 2);
	int C = (L + SU_THREADS - 1) / SU_THREADS;  if ((C & 1) == 0) C++;
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + tile - 1) / tile);
	const size_t   bytes  = ((size_t) L + 2) * sizeof(double);
	if (bytes > 64*1024)
		GDSP_HIP_TRY (hipFuncSetAttribute ((const void*) sliding_sum_kernel,
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int) (maxDoubles*sizeof(double))));
	hipLaunchKernelGGL (sliding_sum_kernel, dim3(ntiles), dim3(SU_THREADS), bytes, gdsp_stream (stream),
	                    d_in, d_out, n, ntiles, tile, (int) Weff, (int) lft, denom, C);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

int gdsp_window_sum (double* d_v, uint32_t n, uint32_t W, double denom, int useActual, double zeroVal, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE (d_v != NULL, "NULL vector");
	GDSP_REQUIRE (W >= 1, "window must be >= 1");
	GDSP_REQUIRE (useActual || (denom != 0.0), "denominator can't be zero");
	if (W > n) W = n;
	const uint32_t nwin = (uint32_t) (((uint64_t) n + W - 1) / W);
	if (W <= WS_TILE_MAX_W)
		{
		// long windows keep few lanes busy in the summing phase: smaller tiles, more workgroups per CU to overlap it
		uint32_t       K      = (W > 256)? ((4096 / W > 2)? 4096 / W : 2) : WS_TILE_BASES / W;
		if ((K * W) & 1) K--;                                              // tiles start on 16-byte boundaries
		const uint32_t ntiles = (uint32_t) (((uint64_t) n + (uint64_t) K*W - 1) / ((uint64_t) K*W));
		const size_t   bytes  = ((size_t) K * (W | 1) + K + 2) * sizeof(double);
		if (bytes > 64*1024)
			GDSP_HIP_TRY (hipFuncSetAttribute ((const void*) window_sum_tile_kernel,
			                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96*1024));
		hipLaunchKernelGGL (window_sum_tile_kernel, dim3(ntiles), dim3(SU_THREADS), bytes, gdsp_stream (stream),
		                    d_v, n, W, K, ntiles, denom, useActual, zeroVal);
		}
	else if ((W <= WS_SEQ_MAX) && gdsp_aligned16 (d_v))
		{
		unsigned char* d_done = NULL;
		{
		std::lock_guard<std::mutex> hold (wsLock);
		const int rc = ws_done_flags (stream, nwin, &d_done);
		if (rc != GDSP_OK) return rc;
		}
		hipLaunchKernelGGL (window_sum_exact_kernel, dim3((nwin + WX_THREADS/64 - 1)/(WX_THREADS/64)), dim3(WX_THREADS), 0,
		                    gdsp_stream (stream), d_v, n, W, nwin, denom, useActual, zeroVal, d_done);
		hipLaunchKernelGGL (window_sum_rows_kernel, dim3((nwin + WR_THREADS - 1)/WR_THREADS), dim3(WR_THREADS), 0,
		                    gdsp_stream (stream), d_v, n, W, nwin, denom, useActual, zeroVal, d_done);
		}
	else if (W <= WS_SEQ_MAX)
		hipLaunchKernelGGL (window_sum_seq_kernel, dim3((nwin + SU_THREADS - 1)/SU_THREADS), dim3(SU_THREADS), 0,
		                    gdsp_stream (stream), d_v, n, W, nwin, denom, useActual, zeroVal);
	else
		hipLaunchKernelGGL (window_sum_wide_kernel, dim3(nwin), dim3(SU_THREADS), 0,
		                    gdsp_stream (stream), d_v, n, W, denom, useActual, zeroVal);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

size_t gdsp_cumulative_sum_work (uint32_t n)
	{
	const size_t three = ((((size_t) n + CS_CHUNK - 1) / CS_CHUNK) + 1) * sizeof(double), one = cl_work_words (n) * sizeof(double);
	return (one > three)? one : three;
	}

int gdsp_cumulative_sum (double* d_v, uint32_t n, void* d_work, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_v != NULL) && (d_work != NULL), "NULL pointer");
	const uint32_t nchunks = (uint32_t) (((uint64_t) n + CS_CHUNK - 1) / CS_CHUNK);
	double*        totals  = (double*) d_work;
	hipStream_t    s       = gdsp_stream (stream);
	GDSP_REQUIRE (gdsp_aligned16 (d_v), "vector must be 16-byte aligned");
	// GDSP_CUMSUM=3: the three launches (rounds 1-4; A/B); default: one pass with a three-level look-back
	static int passes = 0;
	if (passes == 0) { const char* e = getenv ("GDSP_CUMSUM");  passes = ((e != NULL) && (e[0] == '3'))? 3 : 1; }
	if (passes == 1)
		{
		const uint32_t nch = (uint32_t) (((uint64_t) n + CL_CHUNK - 1) / CL_CHUNK);
		unsigned long long* T = (unsigned long long*) d_work;
		unsigned long long* G = T + (size_t) nch * CL_TSTRIDE;
		unsigned long long* S = G + (nch + CL_GROUP - 1) / CL_GROUP;
		GDSP_HIP_TRY (hipMemsetAsync (d_work, 0xFF, cl_work_words (n) * sizeof(double), s));
		const uint32_t nwhole = n / CL_CHUNK;
		if (nwhole != 0) hipLaunchKernelGGL (cumsum_lookback_kernel<true>, dim3(nwhole), dim3(CL_THREADS), 0, s, d_v, n, 0u, T, G, S);
		if (nwhole != nch) hipLaunchKernelGGL (cumsum_lookback_kernel<false>, dim3(1), dim3(CL_THREADS), 0, s, d_v, n, nwhole, T, G, S);
		GDSP_LAUNCH_CHECK ();
		return GDSP_OK;
		}
	hipLaunchKernelGGL (cumsum_totals_kernel,  dim3(nchunks), dim3(SU_THREADS), 0, s, d_v, n, nchunks, totals);
	hipLaunchKernelGGL (cumsum_offsets_kernel, dim3(1),       dim3(CO_THREADS), 0, s, totals, nchunks);
	hipLaunchKernelGGL (cumsum_apply_kernel,   dim3(nchunks), dim3(SU_THREADS), 0, s, d_v, n, nchunks, totals);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

} // extern "C"
