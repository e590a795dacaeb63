// gdsp_clump.hip -- clump / anticlump: intervals whose average is on the right side of a threshold.
//
// Reference: clump_search, clump.c:494-736 (Allison's maximal-average scan).  For every base ix the
// reference keeps one running sum P of val = +-(v - avg) and a monotone list of its record minima and
// marks the longest stretch ending at ix whose sum is >= 0 when it is at least minLength (L) long;
// marked stretches are merged and each merged run is then trimmed to its first and last base on the
// right side of the threshold (clump.c:668-716).  That walk is sequential, but what it marks is a pure
// function of the prefix sums:
//     base t is marked  <=>  there are j < t <= i with i-j >= L and P[j] <= P[i]            (P[-1] = 0)
//                       <=>  Q[t-L-1] <= R[t]   or   good[j] for some j in [t-L, t-1]
// with Q = running minimum of P from the left (P[-1] included), R = running maximum of P from the right
// and good[j] = (P[j] <= R[j+L]).
//
// Round 1 ran this as six whole-vector scans over f64 arrays, ~290 B of HBM traffic per base.  Here the
// only f64 array that is materialised is R; everything else is either recomputed from v in flight or
// lives as one BIT per base:
//   pass 1  (read v, write R')  per 4096-base chunk: the sum of val and the smallest / largest prefix sum relative to
//                               the chunk's start, and R'[t] = the largest prefix sum from t to the end of the chunk, also
//                               relative to its start (prefix sums built in LDS); one small kernel then turns the
//                               chunk figures into each chunk's offset, the running minimum before it and the running
//                               maximum after it.  R[t] = max (offset[chunk] + R'[t], maximum after the chunk) is formed
//                               where it is used (rounding is monotone: the largest of offset + x is offset + the largest
//                               x), so the pass that wrote R from a second read of v (round 2: 48 B/base) is gone
//   pass 2  (read v, read R' shifted by L)  P and Q rebuilt the same way; three bits per base:
//                               g[j] = good[j],  a[k] = (Q[k] <= R[k+L+1])  (the first test, for t = k+L+1),
//                               s[k] = base k is on the right side of the threshold
//   bits    marked[t] = a[t-L-1] | any g in [t-L, t-1];  trimming = "the last event at or before t and the
//                               first at or after t are on-side bases of the same marked run": index-of-
//                               last/first-set-bit scans over 64-bit words (n/64 of them)
//   pass 3  (write v)           one / zero from the final bit
// 40 B of HBM traffic per base plus ~1 B for the bit arrays.
//
// One definition of P serves every pass: P[k] = offset[chunk] + rel[k], rel = the chunk-relative prefix sum
// built by chunk_prefix() (16 consecutive bases per thread, then lanes, then waves, always in that order),
// so the passes agree bit for bit with each other; min / max of P over a chunk follow from min / max of
// rel because rounding is monotone.  Against the reference, which sums left to right, the result is
// bit-identical whenever P is exactly representable (read depth against an integer or dyadic threshold),
// like slidingsum; comparisons are exact.

#include <math.h>
#include "gdsp_common.h"

#define CL_THREADS 256
#define CL_PER     16
#define CL_CHUNK   (CL_THREADS * CL_PER)
#define CL_PITCH   (CL_PER + 1)
#define CL_WORDS   (CL_CHUNK / 64)               // 64-bit words of flags per chunk

// val of the chunk starting at base k0, into LDS with a pitch of 17 per 16 (bases past n count as 0, which
// leaves every prefix after the last base equal to the last one); thread p then owns bases 16p .. 16p+15
__device__ __forceinline__ void cl_stage_vals (double* turn, const double* __restrict__ v, size_t n, size_t k0,
                                               double avg, int above)
	{
	if ((k0 + CL_CHUNK <= n) && gdsp_aligned16 (v))
		{
		const double2* src = reinterpret_cast<const double2*> (v + k0);
		double2 r[CL_PER/2];
#pragma unroll
		for (int u=0 ; u<CL_PER/2 ; u++) r[u] = gdsp_ld2 (&src[u*CL_THREADS + threadIdx.x]);
#pragma unroll
		for (int u=0 ; u<CL_PER/2 ; u++)
			{
			const int e = 2 * (u*CL_THREADS + (int) threadIdx.x);
			double* dst = turn + e + (e >> 4);
			dst[0] = above? r[u].x - avg : avg - r[u].x;             // clump.c:583-584
			dst[1] = above? r[u].y - avg : avg - r[u].y;
			}
		}
	else
		{
		for (int e=threadIdx.x ; e<CL_CHUNK ; e+=CL_THREADS)
			{
			const size_t k = k0 + e;
			const double x = (k < n)? v[k] : avg;
			turn[e + (e >> 4)] = above? x - avg : avg - x;
			}
		}
	__syncthreads ();
	}

// rel[i] = prefix sum of val from the chunk's first base through base 16p+i (the one association every pass uses).
// waveTot: 4 doubles of LDS.  Ends with the workgroup in step (one barrier inside).
__device__ __forceinline__ void cl_chunk_prefix (const double* turn, double* waveTot, double (&rel)[CL_PER])
	{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const double* mine = turn + threadIdx.x * CL_PITCH;
	double run = 0.0;
#pragma unroll
	for (int i=0 ; i<CL_PER ; i++) { run += mine[i];  rel[i] = run; }
	double incl = run;
	for (int d=1 ; d<64 ; d*=2)
		{
		const double up = __shfl_up (incl, d, 64);
		if (lane >= d) incl = up + incl;
		}
	double excl = __shfl_up (incl, 1, 64);
	if (lane == 0) excl = 0.0;
	if (lane == 63) waveTot[wave] = incl;
	__syncthreads ();
	double before = 0.0;
	for (int w=0 ; w<wave ; w++) before += waveTot[w];
	before += excl;
#pragma unroll
	for (int i=0 ; i<CL_PER ; i++) rel[i] = before + rel[i];
	}

template <bool MAX> __device__ __forceinline__ double cl_pick (double a, double b) { return MAX? fmax (a, b) : fmin (a, b); }

// ---- pass 1: per chunk, the total and the smallest / largest relative prefix sum, and R' (the largest relative prefix sum
//      from each base to the end of its chunk)
__global__ __launch_bounds__(CL_THREADS)
void clump_chunk_stats_kernel (const double* __restrict__ v, size_t n, double avg, int above,
                               double* __restrict__ total, double* __restrict__ lowest, double* __restrict__ highest,
                               double* __restrict__ Rrel)
	{
	__shared__ __attribute__((aligned(16))) double turn[CL_THREADS * CL_PITCH];
	__shared__ double waveTot[4], waveLo[4], waveHi[4];
	const size_t k0 = (size_t) blockIdx.x * CL_CHUNK;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	cl_stage_vals (turn, v, n, k0, avg, above);
	double rel[CL_PER];
	cl_chunk_prefix (turn, waveTot, rel);
	double lo = rel[0];
#pragma unroll
	for (int i=1 ; i<CL_PER ; i++) lo = fmin (lo, rel[i]);
	if (threadIdx.x == CL_THREADS-1) total[blockIdx.x] = rel[CL_PER-1];
	double m = -INFINITY;
#pragma unroll
	for (int i=CL_PER-1 ; i>=0 ; i--) { m = fmax (m, rel[i]);  rel[i] = m; }         // maximum from base i to the thread's last
	double incl = m;                                               // ... then over the lanes to the right, the waves to the right
	for (int d=1 ; d<64 ; d*=2)
		{
		const double dn = __shfl_down (incl, d, 64);
		if (lane + d < 64) incl = fmax (dn, incl);
		}
	double excl = __shfl_down (incl, 1, 64);
	if (lane == 63) excl = -INFINITY;
	for (int off=32 ; off>0 ; off>>=1) lo = fmin (lo, __shfl_down (lo, off, 64));
	if (lane == 0) { waveLo[wave] = lo;  waveHi[wave] = incl; }
	__syncthreads ();                                              // (also: every read of the staged values is done)
	if (threadIdx.x == 0)
		{
		lowest[blockIdx.x]  = fmin (fmin (waveLo[0], waveLo[1]), fmin (waveLo[2], waveLo[3]));
		highest[blockIdx.x] = fmax (fmax (waveHi[0], waveHi[1]), fmax (waveHi[2], waveHi[3]));
		}
	double beyond = excl;
	for (int w=3 ; w>wave ; w--) beyond = fmax (beyond, waveHi[w]);
	double* mine = turn + threadIdx.x * CL_PITCH;
#pragma unroll
	for (int i=0 ; i<CL_PER ; i++) mine[i] = fmax (rel[i], beyond);
	__syncthreads ();
	if ((k0 + CL_CHUNK <= n) && gdsp_aligned16 (Rrel))
		{
		double2* dst = reinterpret_cast<double2*> (Rrel + k0);
#pragma unroll
		for (int u=0 ; u<CL_PER/2 ; u++)
			{
			const int e = 2 * (u*CL_THREADS + (int) threadIdx.x);
			const double* src = turn + e + (e >> 4);
			gdsp_st2 (&dst[u*CL_THREADS + threadIdx.x], make_double2 (src[0], src[1]));
			}
		}
	else
		{
		for (int e=threadIdx.x ; e<CL_CHUNK ; e+=CL_THREADS)
			{ if (k0 + e < n) Rrel[k0 + e] = turn[e + (e >> 4)]; }
		}
	}

// ---- between passes 1 and 2, one workgroup: total -> offset of each chunk (exclusive, left to right);
//      lowest -> running minimum of P before the chunk (P[-1] = 0 included); highest -> running maximum after it.
//      Tiles of 4096 chunks, four consecutive entries per thread (32 contiguous bytes per lane: coalesced), a carry
//      from tile to tile; the forward sweep does offsets and minima, a backward sweep the maxima.
#define CS_THREADS 1024
#define CS_PER     8
#define CS_TILE    (CS_THREADS * CS_PER)

template <int OP>   // 0 add, 1 min, 2 max
__device__ __forceinline__ double cs_op (double a, double b) { return (OP == 0)? a + b : ((OP == 1)? fmin (a, b) : fmax (a, b)); }
template <int OP> __device__ __forceinline__ double cs_idle () { return (OP == 0)? 0.0 : ((OP == 1)? INFINITY : -INFINITY); }

// exclusive scan of one value per thread over the workgroup, left to right (or right to left), `carry` joined in front
template <int OP, bool FROM_RIGHT>
__device__ __forceinline__ double cs_block_exclusive (double x, double carry, double* part, double* blockTotal)
	{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double incl = x;
	for (int d=1 ; d<64 ; d*=2)
		{
		const double o = FROM_RIGHT? __shfl_down (incl, d, 64) : __shfl_up (incl, d, 64);
		const bool have = FROM_RIGHT? (lane + d < 64) : (lane >= d);
		if (have) incl = FROM_RIGHT? cs_op<OP> (incl, o) : cs_op<OP> (o, incl);
		}
	double excl = FROM_RIGHT? __shfl_down (incl, 1, 64) : __shfl_up (incl, 1, 64);
	if (lane == (FROM_RIGHT? 63 : 0)) excl = cs_idle<OP> ();
	if (lane == (FROM_RIGHT? 0 : 63)) part[wave] = incl;
	__syncthreads ();
	double before = carry, all = carry;
	for (int i=0 ; i<CS_THREADS/64 ; i++)
		{
		const int w = FROM_RIGHT? CS_THREADS/64 - 1 - i : i;
		if (FROM_RIGHT? (w > wave) : (w < wave)) before = cs_op<OP> (before, part[w]);
		all = cs_op<OP> (all, part[w]);
		}
	__syncthreads ();
	*blockTotal = all;                                             // carry for the next tile
	return cs_op<OP> (before, excl);
	}

// The scan over the chunks in three launches of one workgroup per tile of 8192 chunks (one workgroup walking the tiles
// one after the other, forwards and back, took 195 us for the 60 k chunks of a 249 Mbp chromosome: sixteen dependent
// steps of loads, two block scans and stores on a single CU):
//   A  the sum of each tile's totals
//   B  offsets (the tiles before summed in order, then the tile's own exclusive scan), the smallest / largest P of every
//      chunk, and each tile's smallest / largest of those
//   C  the running minimum before every chunk (P[-1] = 0 included) and the running maximum after it
__global__ __launch_bounds__(CS_THREADS)
void clump_scan_sums_kernel (const double* __restrict__ total, uint32_t nchunks, double* __restrict__ tileSum)
	{
	__shared__ double part[CS_THREADS/64];
	const uint32_t c0 = blockIdx.x * CS_TILE + threadIdx.x * CS_PER;
	double sum = 0.0;
#pragma unroll
	for (int i=0 ; i<CS_PER ; i++) sum += (c0 + i < nchunks)? total[c0 + i] : 0.0;
	double all;
	(void) cs_block_exclusive<0, false> (sum, 0.0, part, &all);
	if (threadIdx.x == 0) tileSum[blockIdx.x] = all;
	}

__global__ __launch_bounds__(CS_THREADS)
void clump_scan_offsets_kernel (double* __restrict__ total, double* __restrict__ lowest, double* __restrict__ highest, uint32_t nchunks,
                                const double* __restrict__ tileSum, double* __restrict__ tileMin, double* __restrict__ tileMax)
	{
	__shared__ double part[CS_THREADS/64];
	double carry = 0.0;
	for (uint32_t t=0 ; t<blockIdx.x ; t++) carry += tileSum[t];    // (uniform: at most a few hundred tiles)
	const uint32_t c0 = blockIdx.x * CS_TILE + threadIdx.x * CS_PER;
	double T[CS_PER], lo[CS_PER], hi[CS_PER];
#pragma unroll
	for (int i=0 ; i<CS_PER ; i++)
		{
		const bool in = (c0 + i < nchunks);
		T[i]  = in? total[c0 + i]   : 0.0;
		lo[i] = in? lowest[c0 + i]  : INFINITY;
		hi[i] = in? highest[c0 + i] : -INFINITY;
		}
	double sum = 0.0;
#pragma unroll
	for (int i=0 ; i<CS_PER ; i++) sum += T[i];
	double all;
	double run = cs_block_exclusive<0, false> (sum, carry, part, &all);
	double mn = INFINITY, mx = -INFINITY;
#pragma unroll
	for (int i=0 ; i<CS_PER ; i++)
		{
		if (c0 + i < nchunks)
			{
			total[c0 + i]   = run;                                 // offset of the chunk
			lo[i] = run + lo[i];  lowest[c0 + i]  = lo[i];          // smallest P in it
			hi[i] = run + hi[i];  highest[c0 + i] = hi[i];          // largest P in it
			mn = fmin (mn, lo[i]);  mx = fmax (mx, hi[i]);
			}
		run += T[i];
		}
	double allMin, allMax;
	(void) cs_block_exclusive<1, false> (mn, INFINITY, part, &allMin);
	(void) cs_block_exclusive<2, false> (mx, -INFINITY, part, &allMax);
	if (threadIdx.x == 0) { tileMin[blockIdx.x] = allMin;  tileMax[blockIdx.x] = allMax; }
	}

__global__ __launch_bounds__(CS_THREADS)
void clump_scan_extremes_kernel (double* __restrict__ lowest, double* __restrict__ highest, uint32_t nchunks, uint32_t ntiles,
                                 const double* __restrict__ tileMin, const double* __restrict__ tileMax)
	{
	__shared__ double part[CS_THREADS/64];
	double carryMin = 0.0, carryMax = -INFINITY;                   // P[-1] = 0 is part of the running minimum
	for (uint32_t t=0 ; t<blockIdx.x ; t++) carryMin = fmin (carryMin, tileMin[t]);
	for (uint32_t t=blockIdx.x+1 ; t<ntiles ; t++) carryMax = fmax (carryMax, tileMax[t]);
	const uint32_t c0 = blockIdx.x * CS_TILE + threadIdx.x * CS_PER;
	double lo[CS_PER], hi[CS_PER];
	double mn = INFINITY, mx = -INFINITY;
#pragma unroll
	for (int i=0 ; i<CS_PER ; i++)
		{
		const bool in = (c0 + i < nchunks);
		lo[i] = in? lowest[c0 + i]  : INFINITY;
		hi[i] = in? highest[c0 + i] : -INFINITY;
		mn = fmin (mn, lo[i]);  mx = fmax (mx, hi[i]);
		}
	double unused;
	double before = cs_block_exclusive<1, false> (mn, carryMin, part, &unused);
	double after  = cs_block_exclusive<2, true>  (mx, carryMax, part, &unused);
#pragma unroll
	for (int i=0 ; i<CS_PER ; i++)
		{
		if (c0 + i >= nchunks) break;
		lowest[c0 + i] = before;
		before = fmin (before, lo[i]);
		}
#pragma unroll
	for (int i=CS_PER-1 ; i>=0 ; i--)
		{
		if (c0 + i >= nchunks) continue;
		highest[c0 + i] = after;
		after = fmax (after, hi[i]);
		}
	}

// 16 flags of a thread (bit i = base 16p+i) -> 64-bit words of four neighbouring lanes; lanes with p%4 == 0 hold them
__device__ __forceinline__ unsigned long long cl_pack (uint32_t bits16)
	{
	unsigned long long w = ((unsigned long long) bits16) << (16 * (threadIdx.x & 3));
	w |= __shfl_xor (w, 1, 64);
	w |= __shfl_xor (w, 2, 64);
	return w;
	}

// ---- pass 2: the three flags of every base.  R'[k0+L ..] is fetched into registers before anything else and goes
//      through the LDS image once the staged values have been consumed (35 KiB of LDS: four workgroups per CU)
#define CL_AHEAD ((CL_CHUNK / 2 + 2 + CL_THREADS - 1) / CL_THREADS)      // 16-byte words of R per thread: 9
__global__ __launch_bounds__(CL_THREADS)
void clump_flags_kernel (const double* __restrict__ v, size_t n, double avg, int above, uint64_t L,
                         const double* __restrict__ offset, const double* __restrict__ before, const double* __restrict__ after,
                         const double* __restrict__ R, uint32_t nchunks,
                         unsigned long long* __restrict__ gBits, unsigned long long* __restrict__ aBits,
                         unsigned long long* __restrict__ sBits, size_t nwords)
	{
	__shared__ __attribute__((aligned(16))) double turn[CL_THREADS * CL_PITCH + 2 * CL_PITCH];
	__shared__ double waveTot[4], waveLo[4];
	const size_t k0 = (size_t) blockIdx.x * CL_CHUNK;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

	// R[k0+L .. k0+L+4096], from an even element so that interior chunks move 16-byte words
	const uint64_t r0    = k0 + L;                                 // R index that meets base k0
	const uint64_t rBase = r0 & ~(uint64_t) 1;
	const int      rSkew = (int) (r0 - rBase);
	const bool     rFast = (rBase + CL_CHUNK + 4 <= n) && gdsp_aligned16 (R);
	double2 ahead[CL_AHEAD];
	if (rFast)
		{
		const double2* src = reinterpret_cast<const double2*> (R + rBase);
#pragma unroll
		for (int u=0 ; u<CL_AHEAD ; u++)
			{ const int q = u*CL_THREADS + (int) threadIdx.x;  ahead[u] = gdsp_ld2 (&src[(q < CL_CHUNK/2 + 2)? q : CL_CHUNK/2 + 1]); }
		}
	cl_stage_vals (turn, v, n, k0, avg, above);

	double P[CL_PER];
	cl_chunk_prefix (turn, waveTot, P);
	const double  off  = offset[blockIdx.x];
	double*       mine = turn + threadIdx.x * CL_PITCH;
	double Q[CL_PER];
	double m = INFINITY;
	uint32_t sd = 0;
	const size_t kb = k0 + (size_t) threadIdx.x * CL_PER;
#pragma unroll
	for (int i=0 ; i<CL_PER ; i++)
		{
		if ((kb + i < n) && (mine[i] >= 0.0)) sd |= 1u << i;       // clump.c:687-688 (v >= avg, or v <= avg)
		P[i] = off + P[i];  m = fmin (m, P[i]);  Q[i] = m;
		}
	double incl = m;
	for (int d=1 ; d<64 ; d*=2)
		{
		const double up = __shfl_up (incl, d, 64);
		if (lane >= d) incl = fmin (up, incl);
		}
	double excl = __shfl_up (incl, 1, 64);
	if (lane == 0) excl = INFINITY;
	if (lane == 63) waveLo[wave] = incl;
	__syncthreads ();                                              // (every read of the staged values is done too)
	double sofar = before[blockIdx.x];                             // running minimum before the chunk, P[-1] = 0 included
	for (int w=0 ; w<wave ; w++) sofar = fmin (sofar, waveLo[w]);
	sofar = fmin (sofar, excl);

	// R = max (offset + R', maximum after the chunk), with the figures of the chunk each staged element lies in: two
	// chunks, three when k0+L falls on a chunk's last bases and the +1 / +2 look into the one after
	const uint32_t c1 = (uint32_t) (rBase / CL_CHUNK);
	const int      eb1 = (int) ((uint64_t) (c1 + 1) * CL_CHUNK - rBase), eb2 = eb1 + CL_CHUNK;     // first staged element of chunks c1+1, c1+2
	const double   off1 = (c1     < nchunks)? offset[c1]     : 0.0, aft1 = (c1     < nchunks)? after[c1]     : -INFINITY;
	const double   off2 = (c1 + 1 < nchunks)? offset[c1 + 1] : 0.0, aft2 = (c1 + 1 < nchunks)? after[c1 + 1] : -INFINITY;
	const double   off3 = (c1 + 2 < nchunks)? offset[c1 + 2] : 0.0, aft3 = (c1 + 2 < nchunks)? after[c1 + 2] : -INFINITY;
#define CL_WHOLE(e, x) fmax ((((e) < eb1)? off1 : (((e) < eb2)? off2 : off3)) + (x), ((e) < eb1)? aft1 : (((e) < eb2)? aft2 : aft3))
	if (rFast)
		{
#pragma unroll
		for (int u=0 ; u<CL_AHEAD ; u++)
			{
			const int q = u*CL_THREADS + (int) threadIdx.x;
			if (q < CL_CHUNK/2 + 2)
				{ const int e = 2*q;  turn[e + (e >> 4)] = CL_WHOLE (e, ahead[u].x);  turn[(e+1) + ((e+1) >> 4)] = CL_WHOLE (e + 1, ahead[u].y); }
			}
		}
	else
		{
		for (int e=threadIdx.x ; e<CL_CHUNK + 4 ; e+=CL_THREADS)
			turn[e + (e >> 4)] = (rBase + e < n)? CL_WHOLE (e, R[rBase + e]) : 0.0;   // (never compared: the index tests below come first)
		}
#undef CL_WHOLE
	__syncthreads ();

	uint32_t g = 0, a = 0;
#pragma unroll
	for (int i=0 ; i<CL_PER ; i++)
		{
		const size_t k = kb + i;
		const int    e = (int) threadIdx.x * CL_PER + i + rSkew;
		const double rHere = turn[e + (e >> 4)];                   // R[k+L]
		const double rNext = turn[(e+1) + ((e+1) >> 4)];           // R[k+L+1]
		const bool inside = (k < n);
		if (inside && (k + L     < n) && (P[i] <= rHere))                g |= 1u << i;     // good[k]
		if (inside && (k + L + 1 < n) && (fmin (sofar, Q[i]) <= rNext))  a |= 1u << i;     // first test, for t = k+L+1
		}
	const unsigned long long gw = cl_pack (g), aw = cl_pack (a), sw = cl_pack (sd);
	if ((threadIdx.x & 3) == 0)
		{
		const size_t w = (size_t) blockIdx.x * CL_WORDS + (threadIdx.x >> 2);
		if (w < nwords) { gBits[w] = gw;  aBits[w] = aw;  sBits[w] = sw; }        // (bits past n are zero)
		}
	}

// ------------------------------------------------------------------ the bit arrays ----
// One thread per 64-bit word, 256 words per workgroup.  "last" = index of the last set bit at or before a
// position (-1: none), "first" = index of the first set bit at or after it (NOBIT: none).
#define CB_THREADS 256
#define NOBIT      ((long long) 1 << 62)

__device__ __forceinline__ long long cb_last_in (unsigned long long w, long long base)
	{ return (w == 0)? -1 : base + 63 - __builtin_clzll (w); }
__device__ __forceinline__ long long cb_first_in (unsigned long long w, long long base)
	{ return (w == 0)? NOBIT : base + __builtin_ctzll (w); }

// inclusive max-scan (or min-scan) over the workgroup's 256 values, returning the exclusive value for this thread
template <bool MAX, bool FROM_RIGHT>
__device__ __forceinline__ long long cb_block_exclusive (long long x, long long* part)
	{
	const long long idle = MAX? -1 : NOBIT;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	long long incl = x;
	for (int d=1 ; d<64 ; d*=2)
		{
		const long long o = FROM_RIGHT? __shfl_down (incl, d, 64) : __shfl_up (incl, d, 64);
		const bool have = FROM_RIGHT? (lane + d < 64) : (lane >= d);
		if (have) incl = MAX? max (incl, o) : min (incl, o);
		}
	long long excl = FROM_RIGHT? __shfl_down (incl, 1, 64) : __shfl_up (incl, 1, 64);
	if (lane == (FROM_RIGHT? 63 : 0)) excl = idle;
	if (lane == (FROM_RIGHT? 0 : 63)) part[wave] = incl;
	__syncthreads ();
	for (int w=0 ; w<4 ; w++)
		{
		const bool counts = FROM_RIGHT? (w > wave) : (w < wave);
		if (counts) excl = MAX? max (excl, part[w]) : min (excl, part[w]);
		}
	__syncthreads ();
	return excl;
	}

// per workgroup of 256 words: last set bit of B (blockLast) -- and, when `other` is given, the first set bit of it too
__global__ __launch_bounds__(CB_THREADS)
void clump_bits_extent_kernel (const unsigned long long* __restrict__ B, size_t nwords, long long* __restrict__ blockLast,
                               long long* __restrict__ blockFirst)
	{
	__shared__ long long part[4];
	const size_t w = (size_t) blockIdx.x * CB_THREADS + threadIdx.x;
	const unsigned long long x = (w < nwords)? B[w] : 0;
	long long last = cb_last_in (x, (long long) w * 64), first = cb_first_in (x, (long long) w * 64);
	for (int off=32 ; off>0 ; off>>=1) { last = max (last, __shfl_down (last, off, 64));  first = min (first, __shfl_down (first, off, 64)); }
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = last;
	__syncthreads ();
	if (threadIdx.x == 0) blockLast[blockIdx.x] = max (max (part[0], part[1]), max (part[2], part[3]));
	if (blockFirst == NULL) return;
	__syncthreads ();
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = first;
	__syncthreads ();
	if (threadIdx.x == 0) blockFirst[blockIdx.x] = min (min (part[0], part[1]), min (part[2], part[3]));
	}

// one workgroup: blockLast -> last set bit before each workgroup's words; blockFirst (if given) -> first after them.
// Thread p takes `per` consecutive entries; across the 1024 threads the scans are wave shuffles plus one pass over the
// sixteen waves' results (the log-step scan through LDS this kernel began with, twenty barriers per direction, took 42 us
// for the 15 k entries of a 249 Mbp chromosome, three times per clump).
__global__ __launch_bounds__(1024)
void clump_bits_scan_kernel (long long* __restrict__ blockLast, long long* __restrict__ blockFirst, uint32_t nblocks)
	{
	__shared__ long long part[16];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t per = (nblocks + 1023) / 1024;
	const uint32_t lo = min (threadIdx.x * per, nblocks), hi = min (lo + per, nblocks);
	long long m = -1;
	for (uint32_t b=lo ; b<hi ; b++) m = max (m, blockLast[b]);
	long long incl = m;
	for (int d=1 ; d<64 ; d*=2) { const long long o = __shfl_up (incl, d, 64);  if (lane >= d) incl = max (incl, o); }
	long long excl = __shfl_up (incl, 1, 64);
	if (lane == 0) excl = -1;
	if (lane == 63) part[wave] = incl;
	__syncthreads ();
	for (int w=0 ; w<wave ; w++) excl = max (excl, part[w]);
	m = excl;
	for (uint32_t b=lo ; b<hi ; b++) { const long long t = blockLast[b];  blockLast[b] = m;  m = max (m, t); }
	if (blockFirst == NULL) return;
	__syncthreads ();
	long long f = NOBIT;
	for (uint32_t b=lo ; b<hi ; b++) f = min (f, blockFirst[b]);
	incl = f;
	for (int d=1 ; d<64 ; d*=2) { const long long o = __shfl_down (incl, d, 64);  if (lane + d < 64) incl = min (incl, o); }
	excl = __shfl_down (incl, 1, 64);
	if (lane == 63) excl = NOBIT;
	if (lane == 0) part[wave] = incl;
	__syncthreads ();
	for (int w=wave+1 ; w<16 ; w++) excl = min (excl, part[w]);
	f = excl;
	for (uint32_t b=hi ; b>lo ; b--) { const long long t = blockFirst[b-1];  blockFirst[b-1] = f;  f = min (f, t); }
	}

// bit t of `bits` shifted: returns the 64 bits [first, first+64) of the array (zeros outside [0, 64*nwords))
__device__ __forceinline__ unsigned long long cb_window (const unsigned long long* __restrict__ bits, size_t nwords, long long first)
	{
	if (first <= -64) return 0;
	const long long w  = (first >= 0)? first >> 6 : -1;
	const int       sh = (int) (first - w * 64);                   // 0..63
	const unsigned long long lo = ((w >= 0) && ((size_t) w < nwords))? bits[w] : 0;
	const unsigned long long hi = ((w + 1 >= 0) && ((size_t) (w + 1) < nwords))? bits[w + 1] : 0;
	return (sh == 0)? lo : ((lo >> sh) | (hi << (64 - sh)));
	}

// marked[t] = a[t-L-1] | any g[j], j in [t-L, t-1]   (j = -1 and k = -1 as the two scalars gm1 / am1 computed here);
// writes the marked word, and the two event arrays of the trimming: U = not marked, O = marked and on-side
__global__ __launch_bounds__(CB_THREADS)
void clump_mark_kernel (const unsigned long long* __restrict__ gBits, const unsigned long long* __restrict__ aBits,
                        const unsigned long long* __restrict__ sBits, const long long* __restrict__ lastGBefore,
                        const double* __restrict__ R, const double* __restrict__ offset, const double* __restrict__ after,
                        size_t n, size_t nwords, uint64_t L,
                        unsigned long long* __restrict__ mBits, unsigned long long* __restrict__ uBits,
                        unsigned long long* __restrict__ oBits)
	{
	__shared__ long long part[4];
	const size_t    w    = (size_t) blockIdx.x * CB_THREADS + threadIdx.x;
	const long long base = (long long) w * 64;
	const unsigned long long g = (w < nwords)? gBits[w] : 0;
	// last good[j] before this word
	long long last = cb_block_exclusive<true, false> (cb_last_in (g, base), part);
	last = max (last, lastGBefore[blockIdx.x]);
	if (w >= nwords) return;
	auto whole = [&] (uint64_t i) { const uint64_t c = i / CL_CHUNK;  return fmax (offset[c] + R[i], after[c]); };   // R from R'
	const bool gm1 = (L >= 1) && (L - 1 < n) && (0.0 <= whole (L - 1));      // good[-1]: P[-1] = 0 against R[L-1]
	const bool am1 = (L < n) && (0.0 <= whole (L));                           // first test at t = L: Q = P[-1] = 0
	unsigned long long marked = cb_window (aBits, nwords, base - (long long) L - 1);
	if ((base <= (long long) L) && ((long long) L < base + 64) && am1) marked |= 1ull << (L - base);
	if (L >= 64)
		{
		// closed form: a good base inside the word is less than L behind every later base of the word; before the word's
		// first good base what counts is the last good base in front of the word
		auto upto = [] (long long q) { return (q < 0)? 0ull : ((q >= 63)? ~0ull : ((2ull << q) - 1)); };      // bits 0..q
		const int firstGood = (g != 0)? __builtin_ctzll (g) : 64;
		if (firstGood < 63) marked |= ~0ull << (firstGood + 1);
		if (last >= 0) marked |= upto (last - base + (long long) L) & upto ((firstGood < 64)? firstGood : 63);
		if (gm1) marked |= upto ((long long) L - 1 - base);
		}
	else
		for (int b=0 ; b<64 ; b++)
			{
			const long long t = base + b;
			// `last` = last j <= t-1 with good[j]
			if (((last >= 0) && (last >= t - (long long) L)) || (gm1 && (t <= (long long) L - 1))) marked |= 1ull << b;
			if ((g >> b) & 1) last = t;
			}
	const long long left = (long long) n - base;                   // bases in this word
	const unsigned long long valid = (left >= 64)? ~0ull : ((1ull << left) - 1);
	marked &= valid;
	mBits[w] = marked;
	uBits[w] = ~marked & valid;
	oBits[w] = marked & sBits[w];
	}

// final flag: t is marked, the last event at or before t is an on-side base (not an unmarked one) and so is the first at
// or after t; then one / zero for the 64 * 256 bases of the workgroup, 16-byte stores
__global__ __launch_bounds__(CB_THREADS)
void clump_write_kernel (const unsigned long long* __restrict__ mBits, const unsigned long long* __restrict__ uBits,
                         const unsigned long long* __restrict__ oBits,
                         const long long* __restrict__ lastUBefore, const long long* __restrict__ firstUAfter,
                         const long long* __restrict__ lastOBefore, const long long* __restrict__ firstOAfter,
                         size_t n, size_t nwords, double one, double zero, double* __restrict__ v)
	{
	__shared__ long long part[4];
	__shared__ unsigned long long outWord[CB_THREADS];
	const size_t    w    = (size_t) blockIdx.x * CB_THREADS + threadIdx.x;
	const long long base = (long long) w * 64;
	const unsigned long long m = (w < nwords)? mBits[w] : 0, u = (w < nwords)? uBits[w] : 0, o = (w < nwords)? oBits[w] : 0;
	long long lastU  = max (cb_block_exclusive<true,  false> (cb_last_in  (u, base), part), lastUBefore[blockIdx.x]);
	long long lastO  = max (cb_block_exclusive<true,  false> (cb_last_in  (o, base), part), lastOBefore[blockIdx.x]);
	long long firstU = min (cb_block_exclusive<false, true>  (cb_first_in (u, base), part), firstUAfter[blockIdx.x]);
	long long firstO = min (cb_block_exclusive<false, true>  (cb_first_in (o, base), part), firstOAfter[blockIdx.x]);
	unsigned long long fromLeft = 0, fromRight = 0;
	for (int b=0 ; b<64 ; b++)
		{
		if ((u >> b) & 1) lastU = base + b;
		if ((o >> b) & 1) lastO = base + b;
		if (lastO > lastU) fromLeft |= 1ull << b;
		}
	for (int b=63 ; b>=0 ; b--)
		{
		if ((u >> b) & 1) firstU = base + b;
		if ((o >> b) & 1) firstO = base + b;
		if (firstO < firstU) fromRight |= 1ull << b;
		}
	outWord[threadIdx.x] = m & fromLeft & fromRight;
	__syncthreads ();
	const size_t t0 = (size_t) blockIdx.x * CB_THREADS * 64;
	if ((t0 + CB_THREADS * 64 <= n) && gdsp_aligned16 (v))
		{
		double2* dst = reinterpret_cast<double2*> (v + t0);
		for (int q=threadIdx.x ; q<CB_THREADS*32 ; q+=CB_THREADS)
			{
			const unsigned long long word = outWord[q >> 5];
			const int b = (2*q) & 63;
			gdsp_st2 (&dst[q], make_double2 (((word >> b) & 1)? one : zero, ((word >> (b+1)) & 1)? one : zero));
			}
		}
	else
		{
		for (int e=threadIdx.x ; e<CB_THREADS*64 ; e+=CB_THREADS)
			{ if (t0 + e < n) v[t0 + e] = ((outWord[e >> 6] >> (e & 63)) & 1)? one : zero; }
		}
	}

extern "C" {

static size_t clump_words (size_t n)   { return (n + 63) / 64; }
static size_t clump_wblocks (size_t n) { return (clump_words (n) + CB_THREADS - 1) / CB_THREADS; }
static size_t clump_chunks (size_t n)  { return (n + CL_CHUNK - 1) / CL_CHUNK; }

size_t gdsp_clump_work (uint32_t n)
	{
	const size_t N = n;
	return ((N + 8) + 3 * (clump_chunks (N) + 2) + 3 * (clump_chunks (N) / CS_TILE + 4)) * sizeof(double)
	     + (6 * (clump_words (N) + 2)) * sizeof(unsigned long long) + (5 * (clump_wblocks (N) + 2)) * sizeof(long long) + 64;
	}

/* clump (above != 0) / anticlump: in place; d_work >= gdsp_clump_work(n) bytes */
int gdsp_clump (double* d_v, uint32_t n, double average, uint32_t minLength, int above,
                double one, double zero, void* d_work, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_v != NULL) && (d_work != NULL), "NULL pointer");
	GDSP_REQUIRE (gdsp_aligned16 (d_work), "workspace must be 16-byte aligned");
	hipStream_t    s = gdsp_stream (stream);
	const size_t   N = n;
	const uint64_t L = (minLength < 1)? 1 : minLength;
	const uint32_t nchunks  = (uint32_t) clump_chunks (N);
	const size_t   nwords   = clump_words (N);
	const uint32_t nwblocks = (uint32_t) clump_wblocks (N);

	double* R       = (double*) d_work;
	double* offset  = R + ((N + 8) & ~(size_t) 1);
	double* lowest  = offset + nchunks + 2;
	double* highest = lowest + nchunks + 2;
	const uint32_t nscan = (nchunks + CS_TILE - 1) / CS_TILE;
	double* tileSum = highest + nchunks + 2;
	double* tileMin = tileSum + nscan + 2;
	double* tileMax = tileMin + nscan + 2;
	unsigned long long* gBits = (unsigned long long*) (highest + nchunks + 2 + 3 * ((size_t) nchunks / CS_TILE + 4));
	unsigned long long* aBits = gBits + nwords + 2;
	unsigned long long* sBits = aBits + nwords + 2;
	unsigned long long* mBits = sBits + nwords + 2;
	unsigned long long* uBits = mBits + nwords + 2;
	unsigned long long* oBits = uBits + nwords + 2;
	long long* lastG  = (long long*) (oBits + nwords + 2);
	long long* lastU  = lastG + nwblocks + 2;
	long long* firstU = lastU + nwblocks + 2;
	long long* lastO  = firstU + nwblocks + 2;
	long long* firstO = lastO + nwblocks + 2;

	hipLaunchKernelGGL (clump_chunk_stats_kernel, dim3(nchunks), dim3(CL_THREADS), 0, s, d_v, N, average, above, offset, lowest, highest, R);
	hipLaunchKernelGGL (clump_scan_sums_kernel,     dim3(nscan), dim3(CS_THREADS), 0, s, offset, nchunks, tileSum);
	hipLaunchKernelGGL (clump_scan_offsets_kernel,  dim3(nscan), dim3(CS_THREADS), 0, s, offset, lowest, highest, nchunks, tileSum, tileMin, tileMax);
	hipLaunchKernelGGL (clump_scan_extremes_kernel, dim3(nscan), dim3(CS_THREADS), 0, s, lowest, highest, nchunks, nscan, tileMin, tileMax);
	hipLaunchKernelGGL (clump_flags_kernel,       dim3(nchunks), dim3(CL_THREADS), 0, s, d_v, N, average, above, L, offset, lowest, highest, R,
	                    nchunks, gBits, aBits, sBits, nwords);
	hipLaunchKernelGGL (clump_bits_extent_kernel, dim3(nwblocks), dim3(CB_THREADS), 0, s, gBits, nwords, lastG, (long long*) NULL);
	hipLaunchKernelGGL (clump_bits_scan_kernel,   dim3(1), dim3(1024), 0, s, lastG, (long long*) NULL, nwblocks);
	hipLaunchKernelGGL (clump_mark_kernel,        dim3(nwblocks), dim3(CB_THREADS), 0, s, gBits, aBits, sBits, lastG, R, offset, highest, N, nwords, L,
	                    mBits, uBits, oBits);
	hipLaunchKernelGGL (clump_bits_extent_kernel, dim3(nwblocks), dim3(CB_THREADS), 0, s, uBits, nwords, lastU, firstU);
	hipLaunchKernelGGL (clump_bits_extent_kernel, dim3(nwblocks), dim3(CB_THREADS), 0, s, oBits, nwords, lastO, firstO);
	hipLaunchKernelGGL (clump_bits_scan_kernel,   dim3(1), dim3(1024), 0, s, lastU, firstU, nwblocks);
	hipLaunchKernelGGL (clump_bits_scan_kernel,   dim3(1), dim3(1024), 0, s, lastO, firstO, nwblocks);
	hipLaunchKernelGGL (clump_write_kernel,       dim3(nwblocks), dim3(CB_THREADS), 0, s, mBits, uBits, oBits, lastU, firstU, lastO, firstO,
	                    N, nwords, one, zero, d_v);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

} // extern "C"
