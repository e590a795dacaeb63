// gdsp_morph.hip -- dilate / erode / close / open on the binarised signal.
//
// Reference: op_dilate_apply morphology.c:882-1072, op_erode_apply :1331-1454,
// op_close_apply :231-319, op_open_apply :529-605.  The reference walks each
// chromosome once with a run-length state machine; the same results follow
// from per-position questions about the set S = {i : v[i] > T}:
//   dilate: one  iff  S meets [i-right, i+left]
//   erode:  one  iff  [i-right, i+left] lies inside S    (outside the vector = not in S)
//   close:  one  iff  i in S, or i sits in a gap of S that touches neither end of
//                     the vector and is no longer than closingLength
//   open:   one  iff  i sits in a run of S longer than openingLength
// Only predicates and integer distances are involved, so the output is
// bit-identical to the reference.
//
// HBM-bound (8 B read + 8 B write per base) on MI355X.  A workgroup turns a tile
// of the signal plus its halo into a *bit* mask in LDS: lanes load 16 bytes each,
// wave ballots gather the predicate bits, and 128 bases become two 64-bit words
// -- a 1001-base halo costs 126 bytes of LDS, so tiles are large (16 K bases)
// and the halo re-read stays ~6 %.  Per-word "next set bit at or after" /
// "previous set bit at or before" tables (one block-wide scan) then answer
// every query above with one or two LDS reads, whatever the window length.

#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include "gdsp_common.h"

#define MO_THREADS   256
#define MO_MIN_TILE  16384
#define MO_MAX_STAGE 262144          // staged bases per workgroup (mask 32 KiB + tables 32 KiB)
#define MO_STAGE_UNROLL 8
#define MO_NONE_HI   0x3fffffff      // "no set bit to the right"
#define MO_NONE_LO   (-0x3fffffff)   // "no set bit to the left"

enum { MO_DILATE = 0, MO_ERODE = 1, MO_CLOSE = 2, MO_OPEN = 3 };

__device__ __forceinline__ uint64_t mo_spread32 (uint64_t x)
	{
	x &= 0xFFFFFFFFULL;
	x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
	x = (x | (x <<  8)) & 0x00FF00FF00FF00FFULL;
	x = (x | (x <<  4)) & 0x0F0F0F0F0F0F0F0FULL;
	x = (x | (x <<  2)) & 0x3333333333333333ULL;
	x = (x | (x <<  1)) & 0x5555555555555555ULL;
	return x;
	}

// membership tests, written as the reference writes them (they differ for NaN only)
template <int OP>
__device__ __forceinline__ bool mo_member (double x, double T, int64_t g)
	{
	if (OP == MO_DILATE) return (g == 0)? (x > T) : !(x <= T);   // morphology.c:930 vs :935
	if (OP == MO_CLOSE)  return !(x <= T);                        // morphology.c:265, :270
	return (x > T);                                               // :1384/:1391, :563/:568
	}

struct MoTables { const uint64_t* mask;  const int* nextFrom;  const int* prevTo; };

// smallest set position >= p (p inside the staged range), or MO_NONE_HI
__device__ __forceinline__ int mo_next (const MoTables& t, int p)
	{
	int      w    = p >> 6;
	uint64_t bits = t.mask[w] >> (p & 63);
	if (bits) return p + __builtin_ctzll (bits);
	return t.nextFrom[w+1];
	}
// largest set position <= p, or MO_NONE_LO
__device__ __forceinline__ int mo_prev (const MoTables& t, int p)
	{
	int      w    = p >> 6;
	uint64_t bits = t.mask[w] << (63 - (p & 63));
	if (bits) return p - __builtin_clzll (bits);
	return (w > 0)? t.prevTo[w-1] : MO_NONE_LO;
	}

// ---- stage: 128 bases per wave-step -> two mask words, MO_STAGE_UNROLL steps' loads in flight.
// Tiles whose whole staged range lies inside the vector (all but the two ends) load
// unconditionally: a predicated load gets its own branch and an s_waitcnt vmcnt(0), which
// would leave one load in flight per wave.
template <int OP>
__device__ __forceinline__ void mo_stage (uint64_t* mask, const double* __restrict__ in, uint32_t n,
                                          int64_t g0, int nwords, double T, bool complement)
	{
	const int  lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int  nchunks  = nwords / 2;
	const bool interior = (g0 >= 0) && (g0 + 64*(int64_t) nwords <= (int64_t) n);
	for (int c0 = wave*MO_STAGE_UNROLL ; c0 < nchunks ; c0 += (MO_THREADS/64)*MO_STAGE_UNROLL)
		{
		double2 d[MO_STAGE_UNROLL];
		bool    hx[MO_STAGE_UNROLL], hy[MO_STAGE_UNROLL];
		if (interior)
			{
#pragma unroll
			for (int u=0 ; u<MO_STAGE_UNROLL ; u++)
				{
				const int c = (c0+u < nchunks)? c0+u : nchunks-1;      // clamped: duplicates are harmless
				d[u]  = gdsp_ld2 (reinterpret_cast<const double2*> (in + g0 + 128*(int64_t) c + 2*lane));
				hx[u] = hy[u] = true;
				}
			}
		else
			{
#pragma unroll
			for (int u=0 ; u<MO_STAGE_UNROLL ; u++)
				{
				const int64_t g = g0 + 128*(int64_t) (c0+u) + 2*lane;
				const bool live = (c0+u < nchunks);
				hx[u] = live && (g >= 0) && (g < (int64_t) n);
				hy[u] = live && (g + 1 >= 0) && (g + 1 < (int64_t) n);
				d[u].x = hx[u]? in[g]   : 0.0;
				d[u].y = hy[u]? in[g+1] : 0.0;
				}
			}
#pragma unroll
		for (int u=0 ; u<MO_STAGE_UNROLL ; u++)
			{
			const int     c = c0 + u;
			const int64_t g = g0 + 128*(int64_t) c + 2*lane;
			const uint64_t E = __ballot (hx[u] && mo_member<OP> (d[u].x, T, g));
			const uint64_t O = __ballot (hy[u] && mo_member<OP> (d[u].y, T, g+1));
			if ((lane == 0) && (c < nchunks))
				{
				uint64_t wA = mo_spread32 (E)       | (mo_spread32 (O)       << 1);
				uint64_t wB = mo_spread32 (E >> 32) | (mo_spread32 (O >> 32) << 1);
				if (complement) { wA = ~wA;  wB = ~wB; }
				mask[2*c]   = wA;
				mask[2*c+1] = wB;
				}
			}
		}
	__syncthreads ();
	}

// ---- per-word tables by one block-wide scan: each thread owns K consecutive words.
// nextFrom[w] = first set position at or after word w (nwords+1 entries), prevTo[w] = last
// set position at or before the end of word w.  Ends with a barrier.
__device__ __forceinline__ void mo_build_tables (const uint64_t* mask, int* nextFrom, int* prevTo, int nwords,
                                                 int* scanA, int* scanB)
	{
	const int K  = (nwords + MO_THREADS - 1) / MO_THREADS;
	const int w0 = threadIdx.x * K, w1 = (w0 + K < nwords)? w0 + K : nwords;
	int firstSet = MO_NONE_HI, lastSet = MO_NONE_LO;
	for (int w=w0 ; w<w1 ; w++)
		{
		uint64_t m = mask[w];
		if (m)
			{
			if (firstSet == MO_NONE_HI) firstSet = 64*w + __builtin_ctzll (m);
			lastSet = 64*w + 63 - __builtin_clzll (m);
			}
		}
	scanA[threadIdx.x] = firstSet;       // suffix-min over threads
	scanB[threadIdx.x] = lastSet;        // prefix-max over threads
	__syncthreads ();
	for (int d=1 ; d<MO_THREADS ; d*=2)
		{
		int a = ((int) threadIdx.x + d < MO_THREADS)? scanA[threadIdx.x + d] : MO_NONE_HI;
		int b = ((int) threadIdx.x - d >= 0)?         scanB[threadIdx.x - d] : MO_NONE_LO;
		__syncthreads ();
		if (a < scanA[threadIdx.x]) scanA[threadIdx.x] = a;
		if (b > scanB[threadIdx.x]) scanB[threadIdx.x] = b;
		__syncthreads ();
		}
	int carryHi = ((int) threadIdx.x + 1 < MO_THREADS)? scanA[threadIdx.x + 1] : MO_NONE_HI;
	int carryLo = ((int) threadIdx.x - 1 >= 0)?         scanB[threadIdx.x - 1] : MO_NONE_LO;
	for (int w=w1-1 ; w>=w0 ; w--)
		{
		uint64_t m = mask[w];
		if (m) carryHi = 64*w + __builtin_ctzll (m);
		nextFrom[w] = carryHi;
		}
	for (int w=w0 ; w<w1 ; w++)
		{
		uint64_t m = mask[w];
		if (m) carryLo = 64*w + 63 - __builtin_clzll (m);
		prevTo[w] = carryLo;
		}
	if (threadIdx.x == 0) nextFrom[nwords] = MO_NONE_HI;
	__syncthreads ();
	}

template <int OP>
__global__ __launch_bounds__(MO_THREADS)
void morph_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t ntiles,
                   int tile, int haloL, int nwords,
                   int left, int right, double length, double T, double one, double zero)
	{
	extern __shared__ __attribute__((aligned(16))) uint64_t moLds[];
	uint64_t* mask     = moLds;                                    // nwords
	int*      nextFrom = reinterpret_cast<int*> (mask + nwords);   // nwords+1
	int*      prevTo   = nextFrom + nwords + 1;                    // nwords
	__shared__ int scanA[MO_THREADS], scanB[MO_THREADS];

	const uint32_t t         = gdsp_xcd_tile (blockIdx.x, ntiles);
	const int64_t  tileStart = (int64_t) t * tile;
	const int64_t  g0        = tileStart - haloL;                  // multiple of 128

	// erode and open ask about the complement (positions NOT in S; outside the vector counts)
	mo_stage<OP> (mask, in, n, g0, nwords, T, (OP == MO_ERODE) || (OP == MO_OPEN));
	mo_build_tables (mask, nextFrom, prevTo, nwords, scanA, scanB);

	const MoTables tb = { mask, nextFrom, prevTo };

	if ((OP == MO_CLOSE) || (OP == MO_OPEN))
		{
		// ---- answer by WORDS: one lane settles 64 bases at a time by walking the runs that cross its word (a run
		// is decided once, from its two ends), then the workgroup turns the answer bits into one / zero with
		// 16-byte stores.  A per-base evaluation spends ~30 instructions a base on table look-ups and 64-bit
		// shifts, enough to hold this kernel below the HBM rate; per word it is ~5.
		uint64_t* answer = reinterpret_cast<uint64_t*> (prevTo + nwords + 1);       // tile/64 words behind the tables
		const int firstWord = haloL >> 6, tileWords = tile >> 6;
		for (int k = threadIdx.x ; k < tileWords ; k += MO_THREADS)
			{
			const int      w    = firstWord + k;
			const uint64_t m    = mask[w];
			uint64_t       res  = (OP == MO_CLOSE)? m : 0;     // close: S stays; open: only long runs of S survive
			uint64_t       todo = ~m;                          // positions to decide: gaps of S (close) / runs of S (open: the mask is the complement)
			while (todo != 0)
				{
				const int b = __builtin_ctzll (todo);
				const int p = 64*w + b;
				const int e = mo_next (tb, p);                 // first mask position after the run
				const int s = mo_prev (tb, p);                 // last one before it
				bool set;
				if (OP == MO_CLOSE) set = (e != MO_NONE_HI) && (s != MO_NONE_LO) && !((double) (e - (s+1)) > length);
				else                set = (e == MO_NONE_HI) || (s == MO_NONE_LO) || ((double) (e - (s+1)) > length);
				const int      last  = (e - 64*w >= 64)? 64 : e - 64*w;                 // the run covers bits [b, last)
				const uint64_t below = (last >= 64)? ~0ULL : ((1ULL << last) - 1);
				const uint64_t run   = below & ~((1ULL << b) - 1);
				if (set) res |= run;
				todo &= ~run;
				}
			answer[k] = res;
			}
		__syncthreads ();
		for (int q = threadIdx.x ; q < tile/2 ; q += MO_THREADS)
			{
			const int64_t g = tileStart + 2*q;
			if (g >= (int64_t) n) break;
			const uint64_t word = answer[q >> 5];
			const int      b    = (2*q) & 63;
			const double   r0   = ((word >> b) & 1)? one : zero, r1 = ((word >> (b+1)) & 1)? one : zero;
			if (g + 1 < (int64_t) n) gdsp_st2 (reinterpret_cast<double2*> (out + g), make_double2 (r0, r1));
			else                     out[g] = r0;
			}
		return;
		}

	// ---- answer: two adjacent bases per lane, one 16-byte store
	for (int o = 2*threadIdx.x ; o < tile ; o += 2*MO_THREADS)
		{
		const int64_t g = tileStart + o;
		if (g >= (int64_t) n) break;
		double r[2];
#pragma unroll
		for (int u=0 ; u<2 ; u++)
			{
			const int p = haloL + o + u;           // staged position of this base
			const bool isOne = (OP == MO_DILATE)? (mo_next (tb, p - right) <= p + left)
			                                    : (mo_next (tb, p - right) >  p + left);   // erode: first position outside S at or after p-right
			r[u] = isOne? one : zero;
			}
		if (g + 1 < (int64_t) n) gdsp_st2 (reinterpret_cast<double2*> (out + g), make_double2 (r[0], r[1]));
		else                     out[g] = r[0];
		}
	}

// ---- `= dilate = erode [= binarize]` in one pass (BASELINE configs[3]): 16 B/base for the
// chain instead of 16 B/base per operator.  The dilated set never leaves LDS: it is rebuilt
// as a second bit mask (one ballot per 64 bases) from the first mask's tables, the tables
// are rebuilt for its complement, and the erosion is answered from those.  Every decision is
// the same integer predicate the separate kernels evaluate, so the output is bit-identical
// to running them one after the other.
//   dMemberOne / dMemberZero: whether dilate's `one` / `zero` output value counts as "in" for
//   erode's own threshold; vOne / vZero: erode's outputs already passed through binarize.
__device__ __forceinline__
void morph_dilate_erode_tile (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t t,
                              int tile, int haloL, int nwords, int dLeft, int dRight, int eLeft, int eRight,
                              double T, int dMemberOne, int dMemberZero, double vOne, double vZero)
	{
	extern __shared__ __attribute__((aligned(16))) uint64_t moLds[];
	uint64_t* mask     = moLds;                                    // nwords: S, the input set
	uint64_t* dmask    = mask + nwords;                            // nwords: complement of erode's input set
	int*      nextFrom = reinterpret_cast<int*> (dmask + nwords);  // nwords+1
	int*      prevTo   = nextFrom + nwords + 1;                    // nwords
	__shared__ int scanA[MO_THREADS], scanB[MO_THREADS];

	const int64_t  tileStart = (int64_t) t * tile;
	const int64_t  g0        = tileStart - haloL;
	const int      lane      = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int      S         = 64 * nwords;

	mo_stage<MO_DILATE> (mask, in, n, g0, nwords, T, false);
	mo_build_tables (mask, nextFrom, prevTo, nwords, scanA, scanB);

	// dilated set, one word per wave-step; positions outside the vector are not in any set
		{
		const MoTables tb = { mask, nextFrom, prevTo };
		for (int w = wave ; w < nwords ; w += MO_THREADS/64)
			{
			const int     p = 64*w + lane;
			const int64_t g = g0 + p;
			bool dil = false;
			if ((g >= 0) && (g < (int64_t) n) && (p - dRight >= 0) && (p + dLeft < S))
				dil = (mo_next (tb, p - dRight) <= p + dLeft);
			const bool member = (g >= 0) && (g < (int64_t) n) && (dil? (dMemberOne != 0) : (dMemberZero != 0));
			const uint64_t word = __ballot (member);
			if (lane == 0) dmask[w] = ~word;               // erode asks about the complement
			}
		}
	__syncthreads ();
	mo_build_tables (dmask, nextFrom, prevTo, nwords, scanA, scanB);

	const MoTables tb = { dmask, nextFrom, prevTo };
	for (int o = 2*threadIdx.x ; o < tile ; o += 2*MO_THREADS)
		{
		const int64_t g = tileStart + o;
		if (g >= (int64_t) n) break;
		const int  p  = haloL + o;
		const bool k0 = (mo_next (tb, p     - eRight) > p     + eLeft);
		const bool k1 = (mo_next (tb, p + 1 - eRight) > p + 1 + eLeft);
		if (g + 1 < (int64_t) n) gdsp_st2 (reinterpret_cast<double2*> (out + g), make_double2 (k0? vOne : vZero, k1? vOne : vZero));
		else                     out[g] = k0? vOne : vZero;
		}
	}

__global__ __launch_bounds__(MO_THREADS)
void morph_dilate_erode_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t ntiles,
                                int tile, int haloL, int nwords, int dLeft, int dRight, int eLeft, int eRight,
                                double T, int dMemberOne, int dMemberZero, double vOne, double vZero)
	{
	morph_dilate_erode_tile (in, out, n, gdsp_xcd_tile (blockIdx.x, ntiles), tile, haloL, nwords, dLeft, dRight, eLeft, eRight,
	                         T, dMemberOne, dMemberZero, vOne, vZero);
	}

__global__ __launch_bounds__(MO_THREADS)                          // one grid over every vector of the table (gdsp_common.h)
void morph_dilate_erode_batch_kernel (GdspBatch B, int tile, int haloL, int nwords, int dLeft, int dRight, int eLeft, int eRight,
                                      double T, int dMemberOne, int dMemberZero, double vOne, double vZero)
	{
	const double* in;  double* out;  uint32_t n;
	const uint32_t t = gdsp_batch_tile (B, in, out, n);
	morph_dilate_erode_tile (in, out, n, t, tile, haloL, nwords, dLeft, dRight, eLeft, eRight, T, dMemberOne, dMemberZero, vOne, vZero);
	}

// ------------------------------------------------------------ any reach: the set as bits in HBM ----
// A window longer than one LDS tile can stage (262 k bases) -- the reference accepts any length, morphology.c:696-866,
// :1163-1315 -- is answered from the same three things the tile kernel keeps in LDS, now for the whole vector in HBM
// workspace: the membership bits (n/8 bytes), and per 64-bit word the first member at or after it and the last member
// at or before its end (two 8-byte tables, 0.25 B/base).  Four small passes build them (bits; extents of 256-word
// groups; one workgroup joining the groups; the per-word tables); the answer pass settles 64 bases per lane by
// walking the runs that cross its word -- a run is decided once, however long the reach -- and the workgroup writes
// one / zero with 16-byte stores.  8 B/base read + 8 B/base written + ~0.5 B/base of tables, for any length.
#define MG_THREADS 256
#define MG_NONE_HI ((long long) 1 << 62)
#define MG_NONE_LO ((long long) -1)

template <int OP>
__global__ __launch_bounds__(MG_THREADS)
void mg_bits_kernel (const double* __restrict__ in, uint32_t n, double T, bool complement, unsigned long long* __restrict__ bits,
                     size_t nwords)
	{
	const int    lane = threadIdx.x & 63;
	const size_t wave = (size_t) blockIdx.x * (MG_THREADS/64) + (threadIdx.x >> 6);
	const size_t nwaves = (size_t) gridDim.x * (MG_THREADS/64);
	const size_t nchunks = (nwords + 1) / 2;                       // 128 bases each
	for (size_t c=wave ; c<nchunks ; c+=nwaves)
		{
		const int64_t g = 128 * (int64_t) c + 2*lane;
		double x = 0.0, y = 0.0;
		const bool hx = (g < (int64_t) n), hy = (g + 1 < (int64_t) n);
		if (hy) { const double2 d = gdsp_ld2 (reinterpret_cast<const double2*> (in + g));  x = d.x;  y = d.y; }
		else if (hx) x = in[g];
		const uint64_t E = __ballot (hx && mo_member<OP> (x, T, g));
		const uint64_t O = __ballot (hy && mo_member<OP> (y, T, g+1));
		if (lane == 0)
			{
			uint64_t wA = mo_spread32 (E)       | (mo_spread32 (O)       << 1);
			uint64_t wB = mo_spread32 (E >> 32) | (mo_spread32 (O >> 32) << 1);
			if (complement) { wA = ~wA;  wB = ~wB; }               // (positions past n too: outside the vector is "not in S")
			bits[2*c] = wA;
			if (2*c + 1 < nwords) bits[2*c+1] = wB;
			}
		}
	}

__device__ __forceinline__ long long mg_first_in (unsigned long long w, long long base) { return (w == 0)? MG_NONE_HI : base + __builtin_ctzll (w); }
__device__ __forceinline__ long long mg_last_in  (unsigned long long w, long long base) { return (w == 0)? MG_NONE_LO : base + 63 - __builtin_clzll (w); }

// exclusive min-from-the-right / max-from-the-left over the workgroup's 256 values
template <bool MAX>
__device__ __forceinline__ long long mg_block_exclusive (long long x, long long* part)
	{
	const long long idle = MAX? MG_NONE_LO : MG_NONE_HI;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	long long incl = x;
	for (int d=1 ; d<64 ; d*=2)
		{
		const long long o = MAX? __shfl_up (incl, d, 64) : __shfl_down (incl, d, 64);
		const bool have = MAX? (lane >= d) : (lane + d < 64);
		if (have) incl = MAX? max (incl, o) : min (incl, o);
		}
	long long excl = MAX? __shfl_up (incl, 1, 64) : __shfl_down (incl, 1, 64);
	if (lane == (MAX? 0 : 63)) excl = idle;
	if (lane == (MAX? 63 : 0)) part[wave] = incl;
	__syncthreads ();
	for (int w=0 ; w<4 ; w++)
		{ if (MAX? (w < wave) : (w > wave)) excl = MAX? max (excl, part[w]) : min (excl, part[w]); }
	__syncthreads ();
	return excl;
	}

__global__ __launch_bounds__(MG_THREADS)
void mg_extent_kernel (const unsigned long long* __restrict__ bits, size_t nwords, long long* __restrict__ groupFirst,
                       long long* __restrict__ groupLast)
	{
	__shared__ long long partF[4], partL[4];
	const size_t w = (size_t) blockIdx.x * MG_THREADS + threadIdx.x;
	const unsigned long long x = (w < nwords)? bits[w] : 0;
	long long first = mg_first_in (x, (long long) w * 64), last = mg_last_in (x, (long long) w * 64);
	for (int off=32 ; off>0 ; off>>=1) { first = min (first, __shfl_down (first, off, 64));  last = max (last, __shfl_down (last, off, 64)); }
	if ((threadIdx.x & 63) == 0) { partF[threadIdx.x >> 6] = first;  partL[threadIdx.x >> 6] = last; }
	__syncthreads ();
	if (threadIdx.x == 0)
		{
		groupFirst[blockIdx.x] = min (min (partF[0], partF[1]), min (partF[2], partF[3]));
		groupLast[blockIdx.x]  = max (max (partL[0], partL[1]), max (partL[2], partL[3]));
		}
	}

// one workgroup: groupFirst -> first member after each group, groupLast -> last member before it
__global__ __launch_bounds__(1024)
void mg_join_kernel (long long* __restrict__ groupFirst, long long* __restrict__ groupLast, uint32_t ngroups)
	{
	__shared__ long long a[1024];
	const uint32_t per = (ngroups + 1023) / 1024;
	const uint32_t lo = min (threadIdx.x * per, ngroups), hi = min (lo + per, ngroups);
	long long m = MG_NONE_LO;
	for (uint32_t b=lo ; b<hi ; b++) m = max (m, groupLast[b]);
	a[threadIdx.x] = m;
	__syncthreads ();
	for (int d=1 ; d<1024 ; d*=2)
		{
		const long long up = ((int) threadIdx.x >= d)? a[threadIdx.x - d] : MG_NONE_LO;
		__syncthreads ();
		a[threadIdx.x] = max (up, a[threadIdx.x]);
		__syncthreads ();
		}
	m = (threadIdx.x > 0)? a[threadIdx.x - 1] : MG_NONE_LO;
	for (uint32_t b=lo ; b<hi ; b++) { const long long t = groupLast[b];  groupLast[b] = m;  m = max (m, t); }
	__syncthreads ();
	long long f = MG_NONE_HI;
	for (uint32_t b=lo ; b<hi ; b++) f = min (f, groupFirst[b]);
	a[threadIdx.x] = f;
	__syncthreads ();
	for (int d=1 ; d<1024 ; d*=2)
		{
		const long long dn = ((int) threadIdx.x + d < 1024)? a[threadIdx.x + d] : MG_NONE_HI;
		__syncthreads ();
		a[threadIdx.x] = min (dn, a[threadIdx.x]);
		__syncthreads ();
		}
	f = (threadIdx.x < 1023)? a[threadIdx.x + 1] : MG_NONE_HI;
	for (uint32_t b=hi ; b>lo ; b--) { const long long t = groupFirst[b-1];  groupFirst[b-1] = f;  f = min (f, t); }
	}

// per word: nextFrom[w] = first member at or after the start of word w (nwords+1 entries), prevTo[w] = last member at
// or before the end of word w
__global__ __launch_bounds__(MG_THREADS)
void mg_tables_kernel (const unsigned long long* __restrict__ bits, size_t nwords, const long long* __restrict__ firstAfterGroup,
                       const long long* __restrict__ lastBeforeGroup, long long* __restrict__ nextFrom, long long* __restrict__ prevTo)
	{
	__shared__ long long part[4];
	const size_t w = (size_t) blockIdx.x * MG_THREADS + threadIdx.x;
	const unsigned long long x = (w < nwords)? bits[w] : 0;
	const long long first = mg_first_in (x, (long long) w * 64), last = mg_last_in (x, (long long) w * 64);
	const long long after  = min (mg_block_exclusive<false> (first, part), firstAfterGroup[blockIdx.x]);
	const long long before = max (mg_block_exclusive<true>  (last,  part), lastBeforeGroup[blockIdx.x]);
	if (w < nwords) { nextFrom[w] = min (first, after);  prevTo[w] = max (last, before); }
	if (w == nwords) nextFrom[w] = MG_NONE_HI;
	}

struct MgTables { const unsigned long long* bits;  const long long* nextFrom;  const long long* prevTo;  long long nbits; };

// first member >= p / last member <= p, for any p (nothing outside [0, nbits))
__device__ __forceinline__ long long mg_next (const MgTables& t, long long p)
	{
	if (p < 0) p = 0;
	if (p >= t.nbits) return MG_NONE_HI;
	const long long w = p >> 6;
	const unsigned long long b = t.bits[w] >> (p & 63);
	if (b) return p + __builtin_ctzll (b);
	return t.nextFrom[w+1];
	}
__device__ __forceinline__ long long mg_prev (const MgTables& t, long long p)
	{
	if (p < 0) return MG_NONE_LO;
	if (p >= t.nbits) p = t.nbits - 1;
	const long long w = p >> 6;
	const unsigned long long b = t.bits[w] << (63 - (p & 63));
	if (b) return p - __builtin_clzll (b);
	return (w > 0)? t.prevTo[w-1] : MG_NONE_LO;
	}

template <int OP>
__global__ __launch_bounds__(MG_THREADS)
void mg_answer_kernel (MgTables tb, uint32_t n, long long left, long long right, double length,
                       double one, double zero, double* __restrict__ out)
	{
	__shared__ unsigned long long answer[MG_THREADS];
	const long long w = (long long) blockIdx.x * MG_THREADS + threadIdx.x;
	const long long P = w * 64;
	unsigned long long res = 0;
	if (P < (long long) n)
		{
		const long long stop = (P + 64 < (long long) n)? P + 64 : (long long) n;     // bases [P, stop) are settled here
		if ((OP == MO_DILATE) || (OP == MO_ERODE))
			{
			// dilate: the bits are S; a base is covered while some member lies in [p-right, p+left].
			// erode: the bits are the complement (outside the vector included); a base survives while none does.
			long long p = P;
			while (p < stop)
				{
				long long e = mg_next (tb, p - right);              // first member at or after the window's left end
				if ((OP == MO_ERODE) && (p - right < 0)) e = p - right;                    // the window pokes out on the left: a non-member at once
				if ((OP == MO_ERODE) && (e == MG_NONE_HI)) e = (long long) n;              // ... and the first one past the right end
				long long upto;                                     // the same answer holds for [p, upto)
				bool covered;
				if (e <= p + left) { covered = true;   upto = (OP == MO_ERODE && e < 0)? right : e + right + 1; }   // until the member leaves the window
				else               { covered = false;  upto = e - left; }                  // until the window reaches it
				if (upto > stop) upto = stop;
				if (upto <= p) upto = p + 1;
				const bool isOne = (OP == MO_DILATE)? covered : !covered;
				if (isOne)
					{
					const int a = (int) (p - P), b = (int) (upto - P);
					res |= ((b >= 64)? ~0ULL : ((1ULL << b) - 1)) & ~((1ULL << a) - 1);
					}
				p = upto;
				}
			}
		else
			{
			const unsigned long long m = tb.bits[w];
			res = (OP == MO_CLOSE)? m : 0;
			unsigned long long todo = ~m;
			if (stop - P < 64) todo &= (1ULL << (stop - P)) - 1;
			while (todo != 0)
				{
				const int       b = __builtin_ctzll (todo);
				const long long p = P + b;
				long long e = mg_next (tb, p), s = mg_prev (tb, p);
				bool set;
				if (OP == MO_CLOSE) set = (e != MG_NONE_HI) && (s != MG_NONE_LO) && !((double) (e - (s+1)) > length);   // a gap that touches an end of the vector stays
				else
					{
					// (here the bits span the whole vector: no member of the complement beyond means the run ends with the vector)
					if (e == MG_NONE_HI) e = (long long) n;
					if (s == MG_NONE_LO) s = -1;
					set = ((double) (e - (s+1)) > length);
					}
				const long long last = (e - P >= 64)? 64 : e - P;
				const unsigned long long run = ((last >= 64)? ~0ULL : ((1ULL << last) - 1)) & ~((1ULL << b) - 1);
				if (set) res |= run;
				todo &= ~run;
				}
			}
		}
	answer[threadIdx.x] = res;
	__syncthreads ();
	const size_t t0 = (size_t) blockIdx.x * MG_THREADS * 64;
	for (int q=threadIdx.x ; q<MG_THREADS*32 ; q+=MG_THREADS)
		{
		const size_t g = t0 + 2*(size_t) q;
		if (g >= n) break;
		const unsigned long long word = answer[q >> 5];
		const int    b  = (2*q) & 63;
		const double r0 = ((word >> b) & 1)? one : zero, r1 = ((word >> (b+1)) & 1)? one : zero;
		if (g + 1 < n) gdsp_st2 (reinterpret_cast<double2*> (out + g), make_double2 (r0, r1));
		else           out[g] = r0;
		}
	}

static size_t mg_words (size_t n)  { return (n + 63) / 64; }
static size_t mg_groups (size_t n) { return (mg_words (n) + 1 + MG_THREADS - 1) / MG_THREADS; }
static size_t mg_work_bytes (size_t n)
	{ return (mg_words (n) + 2) * 8 + 2 * (mg_words (n) + 2) * 8 + 2 * (mg_groups (n) + 2) * 8 + 64; }

template <int OP>
static int morph_any (const double* d_in, double* d_out, uint32_t n, uint64_t left, uint64_t right, double length,
                      double T, double one, double zero, void* d_work, size_t workBytes, void* stream)
	{
	GDSP_REQUIRE (d_work != NULL, "this reach needs workspace (gdsp_long_window_work)");
	GDSP_REQUIRE (workBytes >= mg_work_bytes (n), "workspace too small (gdsp_long_window_work)");
	GDSP_REQUIRE (gdsp_aligned16 (d_work), "workspace must be 16-byte aligned");
	const size_t   nwords  = mg_words (n);
	const uint32_t ngroups = (uint32_t) mg_groups (n);
	unsigned long long* bits = (unsigned long long*) d_work;
	long long* nextFrom   = (long long*) (bits + nwords + 2);
	long long* prevTo     = nextFrom + nwords + 2;
	long long* groupFirst = prevTo + nwords + 2;
	long long* groupLast  = groupFirst + ngroups + 2;
	hipStream_t s = gdsp_stream (stream);
	const bool complement = (OP == MO_ERODE) || (OP == MO_OPEN);
	const size_t   nchunks = (nwords + 1) / 2;
	const uint32_t bblocks = (uint32_t) std::min<size_t> ((nchunks + 3) / 4, 16384);
	hipLaunchKernelGGL ((mg_bits_kernel<OP>), dim3(bblocks? bblocks : 1), dim3(MG_THREADS), 0, s, d_in, n, T, complement, bits, nwords);
	hipLaunchKernelGGL (mg_extent_kernel, dim3(ngroups), dim3(MG_THREADS), 0, s, bits, nwords, groupFirst, groupLast);
	hipLaunchKernelGGL (mg_join_kernel,   dim3(1), dim3(1024), 0, s, groupFirst, groupLast, ngroups);
	hipLaunchKernelGGL (mg_tables_kernel, dim3(ngroups), dim3(MG_THREADS), 0, s, bits, nwords, groupFirst, groupLast, nextFrom, prevTo);
	MgTables tb = { bits, nextFrom, prevTo, (long long) nwords * 64 };
	if (left  > n) left  = n;
	if (right > n) right = n;
	const uint32_t ablocks = (uint32_t) ((nwords + MG_THREADS - 1) / MG_THREADS);
	hipLaunchKernelGGL ((mg_answer_kernel<OP>), dim3(ablocks), dim3(MG_THREADS), 0, s, tb, n, (long long) left, (long long) right, length,
	                    one, zero, d_out);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

// GDSP_MORPH_TILE=<bases>: tile size for tuning experiments; read once, values a tile cannot have are ignored
static uint64_t mo_tile_override (void)
	{
	static const uint64_t value = [] () -> uint64_t
		{
		const char* e = getenv ("GDSP_MORPH_TILE");
		if (e == NULL) return 0;
		const long long v = atoll (e);
		return ((v >= 512) && (v <= MO_MAX_STAGE))? (uint64_t) v : 0;
		} ();
	return value;
	}

template <int OP>
static int morph_launch (const double* d_in, double* d_out, uint32_t n, uint64_t left, uint64_t right,
                         double length, double T, double one, double zero, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_in != NULL) && (d_out != NULL), "NULL vector");
	GDSP_REQUIRE (d_in != d_out, "out-of-place operator: d_out must not alias d_in");
	GDSP_REQUIRE (gdsp_aligned16 (d_in) && gdsp_aligned16 (d_out), "vectors must be 16-byte aligned");

	// how far a query can look to either side of its base
	uint64_t reachL, reachR;
	if ((OP == MO_DILATE) || (OP == MO_ERODE))
		{
		// nothing beyond the vector matters: clamp so the arithmetic stays in int
		if (left  > n) left  = n;
		if (right > n) right = n;
		reachL = right;  reachR = left + 1;
		// windows of 17 .. 3584 bases: window-any / window-all in the block form of gdsp_extrema.hip (5.9 TB/s
		// against 5.3 for the bit-mask tile below, which serves every other reach up to 262 k bases)
		if (gdsp_morph_blocks_available ((uint32_t) left, (uint32_t) right))
			{
			gdsp_morph_blocks (d_in, d_out, n, (uint32_t) left, (uint32_t) right, OP == MO_ERODE, T, one, zero, stream);
			GDSP_LAUNCH_CHECK ();
			return GDSP_OK;
			}
		}
	else
		{
		double l = (length < 0)? 0 : length;
		uint64_t h = (l >= (double) n)? (uint64_t) n + 1 : (uint64_t) floor (l) + 1;
		reachL = reachR = h + 1;
		}
	const uint64_t haloL = ((reachL + 127) / 128) * 128;
	const uint64_t haloR = ((reachR + 127) / 128) * 128;
	if (haloL + haloR + 512 > MO_MAX_STAGE)
		{
		gdsp_set_error ("%s: window reaching %llu+%llu bases exceeds what one LDS tile holds (max %d)",
		                __func__, (unsigned long long) reachL, (unsigned long long) reachR, MO_MAX_STAGE - 512);
		return GDSP_EINVAL;
		}
	uint64_t tile = 4 * (haloL + haloR);
	if (tile < MO_MIN_TILE) tile = MO_MIN_TILE;
	if (mo_tile_override () != 0) tile = mo_tile_override ();          // (tuning experiments)
	if (tile + haloL + haloR > MO_MAX_STAGE) tile = MO_MAX_STAGE - haloL - haloR;
	tile = (tile / 512) * 512;
	const int      nwords = (int) ((haloL + tile + haloR) / 64);
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + tile - 1) / tile);
	const size_t   bytes  = (size_t) nwords * 8 + ((size_t) 2*nwords + 2) * 4 + (tile / 64) * 8;     // mask, tables, answer words

	hipLaunchKernelGGL ((morph_kernel<OP>), dim3(ntiles), dim3(MO_THREADS), bytes, gdsp_stream (stream),
	                    d_in, d_out, n, ntiles, (int) tile, (int) haloL, nwords,
	                    (int) left, (int) right, length, T, one, zero);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

extern "C" {

int gdsp_dilate (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right,
                 double T, double one, double zero, void* stream)
	{ return morph_launch<MO_DILATE> (d_in, d_out, n, left, right, 0.0, T, one, zero, stream); }

int gdsp_erode (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right,
                double T, double one, double zero, void* stream)
	{ return morph_launch<MO_ERODE> (d_in, d_out, n, left, right, 0.0, T, one, zero, stream); }

/* every vector of a device in one launch when the block form of gdsp_extrema.hip serves them all (windows of 17..3584
 * bases); vector by vector otherwise */
int gdsp_dilate_batch (const gdsp_batch_item* items, int nitems, uint32_t left, uint32_t right,
                       double T, double one, double zero, void* stream)
	{
	int rc = gdsp_batch_check (items, nitems, false);
	if (rc != GDSP_OK) return rc;
	if (gdsp_morph_blocks_batch (items, nitems, left, right, 0, T, one, zero, stream)) { GDSP_LAUNCH_CHECK ();  return GDSP_OK; }
	for (int i=0 ; i<nitems ; i++)
		{ rc = gdsp_dilate (items[i].d_in, items[i].d_out, items[i].n, left, right, T, one, zero, stream);  if (rc != GDSP_OK) return rc; }
	return GDSP_OK;
	}

int gdsp_erode_batch (const gdsp_batch_item* items, int nitems, uint32_t left, uint32_t right,
                      double T, double one, double zero, void* stream)
	{
	int rc = gdsp_batch_check (items, nitems, false);
	if (rc != GDSP_OK) return rc;
	if (gdsp_morph_blocks_batch (items, nitems, left, right, 1, T, one, zero, stream)) { GDSP_LAUNCH_CHECK ();  return GDSP_OK; }
	for (int i=0 ; i<nitems ; i++)
		{ rc = gdsp_erode (items[i].d_in, items[i].d_out, items[i].n, left, right, T, one, zero, stream);  if (rc != GDSP_OK) return rc; }
	return GDSP_OK;
	}

/* `= dilate = erode [= binarize]` fused (see morph_dilate_erode_kernel).  Each stage keeps its
 * own threshold and output values, exactly as the three operators would apply them. */
struct MoFusedGeom { int tile, haloL, nwords;  size_t bytes;  uint32_t dLeft, dRight, eLeft, eRight; };
static int mo_fused_geom (uint32_t nmax, uint32_t dLeft, uint32_t dRight, uint32_t eLeft, uint32_t eRight, MoFusedGeom* g)
	{
	if (dLeft > nmax) dLeft = nmax;                                    // (nothing beyond the vector matters: keeps the arithmetic in int)
	if (dRight > nmax) dRight = nmax;
	if (eLeft > nmax) eLeft = nmax;
	if (eRight > nmax) eRight = nmax;
	const uint64_t reachL = (uint64_t) dRight + eRight, reachR = (uint64_t) dLeft + eLeft + 1;
	const uint64_t haloL = ((reachL + 127) / 128) * 128, haloR = ((reachR + 127) / 128) * 128;
	if (haloL + haloR + 512 > MO_MAX_STAGE/2)
		{
		gdsp_set_error ("gdsp_dilate_erode: combined reach of %llu+%llu bases exceeds what one LDS tile holds",
		                (unsigned long long) reachL, (unsigned long long) reachR);
		return GDSP_EINVAL;
		}
	uint64_t tile = 4 * (haloL + haloR);
	if (tile < MO_MIN_TILE) tile = MO_MIN_TILE;
	if (mo_tile_override () != 0) tile = mo_tile_override ();          // (tuning experiments)
	if (tile + haloL + haloR > MO_MAX_STAGE/2) tile = MO_MAX_STAGE/2 - haloL - haloR;
	tile = (tile / 512) * 512;
	g->tile = (int) tile;  g->haloL = (int) haloL;  g->nwords = (int) ((haloL + tile + haloR) / 64);
	g->bytes = (size_t) g->nwords * 16 + ((size_t) 2*g->nwords + 2) * 4;
	g->dLeft = dLeft;  g->dRight = dRight;  g->eLeft = eLeft;  g->eRight = eRight;
	return GDSP_OK;
	}

/* 1 when the chain has a fused kernel whatever the vectors' lengths (the combined reach fits one LDS tile) */
int gdsp_dilate_erode_fusable (uint32_t dLeft, uint32_t dRight, uint32_t eLeft, uint32_t eRight)
	{
	const uint64_t reachL = (uint64_t) dRight + eRight, reachR = (uint64_t) dLeft + eLeft + 1;
	const uint64_t haloL = ((reachL + 127) / 128) * 128, haloR = ((reachR + 127) / 128) * 128;
	return haloL + haloR + 512 <= MO_MAX_STAGE/2;
	}

int gdsp_dilate_erode (const double* d_in, double* d_out, uint32_t n,
                       uint32_t dLeft, uint32_t dRight, double dT, double dOne, double dZero,
                       uint32_t eLeft, uint32_t eRight, double eT, double eOne, double eZero,
                       int binarize, double bT, int bTiesAbove, double bOne, double bZero, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_in != NULL) && (d_out != NULL) && (d_in != d_out), "vectors must be distinct and non-NULL");
	GDSP_REQUIRE (gdsp_aligned16 (d_in) && gdsp_aligned16 (d_out), "vectors must be 16-byte aligned");
	MoFusedGeom g;
	int rc = mo_fused_geom (n, dLeft, dRight, eLeft, eRight, &g);
	if (rc != GDSP_OK) return rc;
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + g.tile - 1) / g.tile);

	double vOne = eOne, vZero = eZero;
	if (binarize)
		{
		vOne  = (bTiesAbove? (eOne  >= bT) : (eOne  > bT))? bOne : bZero;     // logical.c:247-257
		vZero = (bTiesAbove? (eZero >= bT) : (eZero > bT))? bOne : bZero;
		}
	hipLaunchKernelGGL (morph_dilate_erode_kernel, dim3(ntiles), dim3(MO_THREADS), g.bytes, gdsp_stream (stream),
	                    d_in, d_out, n, ntiles, g.tile, g.haloL, g.nwords,
	                    (int) g.dLeft, (int) g.dRight, (int) g.eLeft, (int) g.eRight,
	                    dT, (int) (dOne > eT), (int) (dZero > eT), vOne, vZero);       // erode membership: v > T
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

/* the same for every vector of a device in one launch (gdsp_common.h: GdspBatch) */
int gdsp_dilate_erode_batch (const gdsp_batch_item* items, int nitems,
                             uint32_t dLeft, uint32_t dRight, double dT, double dOne, double dZero,
                             uint32_t eLeft, uint32_t eRight, double eT, double eOne, double eZero,
                             int binarize, double bT, int bTiesAbove, double bOne, double bZero, void* stream)
	{
	int rc = gdsp_batch_check (items, nitems, false);
	if (rc != GDSP_OK) return rc;
	uint32_t nmax = 0;
	for (int i=0 ; i<nitems ; i++) { if (items[i].n > nmax) nmax = items[i].n; }
	if (nmax == 0) return GDSP_OK;
	MoFusedGeom g;
	rc = mo_fused_geom (nmax, dLeft, dRight, eLeft, eRight, &g);
	if (rc != GDSP_OK) return rc;
	double vOne = eOne, vZero = eZero;
	if (binarize)
		{
		vOne  = (bTiesAbove? (eOne  >= bT) : (eOne  > bT))? bOne : bZero;
		vZero = (bTiesAbove? (eZero >= bT) : (eZero > bT))? bOne : bZero;
		}
	hipStream_t s = gdsp_stream (stream);
	const int tile = g.tile;
	gdsp_batch_run (items, nitems, [=] (uint32_t n) { return ((uint64_t) n + tile - 1) / tile; },
		[&] (const GdspBatch& B, uint32_t tiles)
			{
			hipLaunchKernelGGL (morph_dilate_erode_batch_kernel, dim3(tiles), dim3(MO_THREADS), g.bytes, s,
			                    B, g.tile, g.haloL, g.nwords, (int) g.dLeft, (int) g.dRight, (int) g.eLeft, (int) g.eRight,
			                    dT, (int) (dOne > eT), (int) (dZero > eT), vOne, vZero);
			});
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

int gdsp_close (const double* d_in, double* d_out, uint32_t n, double closingLength,
                double T, double one, double zero, void* stream)
	{ return morph_launch<MO_CLOSE> (d_in, d_out, n, 0, 0, closingLength, T, one, zero, stream); }

int gdsp_open (const double* d_in, double* d_out, uint32_t n, double openingLength,
               double T, double one, double zero, void* stream)
	{ return morph_launch<MO_OPEN> (d_in, d_out, n, 0, 0, openingLength, T, one, zero, stream); }

/* any length (the tile kernels above return GDSP_EINVAL beyond 262 k bases of reach): the tile kernel when it applies,
 * otherwise the set as bits in d_work (>= gdsp_long_window_work(n) bytes).  GDSP_MORPH_FORCE_BITS=1 sends every call
 * through the workspace form (tests). */
#define MORPH_ANY(OP, LEFT, RIGHT, LENGTH)                                                                              \
	{                                                                                                                   \
	const bool force = (getenv ("GDSP_MORPH_FORCE_BITS") != NULL);                                                      \
	int rc = force? GDSP_EINVAL : morph_launch<OP> (d_in, d_out, n, LEFT, RIGHT, LENGTH, T, one, zero, stream);         \
	if ((rc != GDSP_EINVAL) || (n == 0) || (d_in == NULL) || (d_out == NULL) || (d_in == d_out)                         \
	 || !gdsp_aligned16 (d_in) || !gdsp_aligned16 (d_out)) return rc;                                                   \
	return morph_any<OP> (d_in, d_out, n, LEFT, RIGHT, LENGTH, T, one, zero, d_work, workBytes, stream);                \
	}
int gdsp_dilate_any (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right, double T, double one, double zero,
                     void* d_work, size_t workBytes, void* stream)
	MORPH_ANY (MO_DILATE, left, right, 0.0)
int gdsp_erode_any (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right, double T, double one, double zero,
                    void* d_work, size_t workBytes, void* stream)
	MORPH_ANY (MO_ERODE, left, right, 0.0)
int gdsp_close_any (const double* d_in, double* d_out, uint32_t n, double closingLength, double T, double one, double zero,
                    void* d_work, size_t workBytes, void* stream)
	MORPH_ANY (MO_CLOSE, 0, 0, closingLength)
int gdsp_open_any (const double* d_in, double* d_out, uint32_t n, double openingLength, double T, double one, double zero,
                   void* d_work, size_t workBytes, void* stream)
	MORPH_ANY (MO_OPEN, 0, 0, openingLength)

} // extern "C"
