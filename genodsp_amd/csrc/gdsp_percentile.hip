// gdsp_percentile.hip -- the `percentile` operator end to end: exact order statistics of the
// sampled genome in about one read of it.
//
// Reference: op_percentile_apply, percentile.c:392-751 (population = every window-th value with
// !(v < lo) && !(v > hi), :559-561; rank of percentile p over N values = (u32)(N*p/100000),
// :587-589, :681; rank N means the largest, :688-710).  gdsp_select.hip finds one rank by radix
// select: five reads of the population per requested percentile.  This file keeps that as the
// fallback and puts a bracketing step in front of it:
//   1. a strided subsample (<= 2^24 values over all ranks) is gathered as order-preserving keys;
//   2. for every requested percentile the subsample's order statistics a few standard deviations
//      either side of the target rank are found (radix select over the small subsample);
//   3. ONE pass over the population counts the values below / equal to / between / above those
//      pivots and appends the few that fall strictly inside a bracket to a candidate list
//      (wave-aggregated: a ballot per element slot, one global atomic per 64 candidates);
//   4. the exact population count N and with it every rank are now known; the bin that holds a
//      rank is either a pivot itself (heavy ties: read depth) or a bracket, where the answer is
//      an order statistic of its candidates (radix select over that short list).
// The subsample only decides how tight the brackets are: the counts of step 3 are exact, and a
// rank that falls outside every bracket (or a candidate list that overflows) sends that
// percentile through the full radix select, so the result never depends on the sampling.
// Across GPUs the same decisions are taken everywhere because every count, histogram and flag
// is reduced over all of them -- a few KiB per step, the path's only collective.  Three ways,
// by who owns the GPUs:
//   * one process, several devices, a communicator given (gdsp_percentiles_use_comm): the
//     histograms and counters are all-reduced in place in HBM by RCCL (gdsp_comm.hip) and the
//     host reads one device's copy -- what `genodsp_hip --gpus=N` does;
//   * one process per GPU with a device hook (gdsp_percentiles_use_device_reduce): the same
//     buffers are handed to the caller's collective as device pointers on the stream
//     (bench.py: torch.distributed over RCCL, no host copy);
//   * neither: the host adds the copies of this process's devices, then the caller's host hook
//     (`reduce`) adds across processes (tests over gloo; `--reduce=host`).
// Traffic: 8 B per sampled base once, against 40 B with five select passes.

#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include <algorithm>
#include "gdsp_common.h"

#define PC_THREADS     256
#define PC_MAX_BLOCKS  2048
#define PC_MAX_PIVOTS  32
#define PC_WAVE_BUF    512
#define PC_SAMPLE_TARGET (1u << 20)

// sorted distinct pivots as doubles (padded with NaN, which compares false); collect bit j: keep what lies
// strictly between pivot j-1 and pivot j (bit 0: below the first, bit m: above the last)
struct PcPivots { double val[PC_MAX_PIVOTS];  uint64_t collect;  int m; };

// ---------------------------------------------------------------- kernels ----
// subsample: population element j*sstride of this vector, j = 0.., written to slot j as a key, or as
// PC_NO_KEY when it does not qualify.  (gdsp_key_of folds -0.0 onto +0.0, so the image of -0.0 is
// the one key no value has.)  No atomics: a counter bumped once per wave costs more than the gather.
#define PC_NO_KEY 0x7FFFFFFFFFFFFFFFULL
__global__ __launch_bounds__(PC_THREADS)
void pc_sample_kernel (const double* __restrict__ v, uint32_t n, uint32_t window, double lo, double hi,
                       uint32_t sstride, uint64_t* __restrict__ sample)
	{
	const size_t npop   = ((size_t) n + window - 1) / window;
	const size_t nsamp  = (npop + sstride - 1) / sstride;
	const size_t stride = (size_t) gridDim.x * PC_THREADS;
	for (size_t j = (size_t) blockIdx.x * PC_THREADS + threadIdx.x ; j < nsamp ; j += stride)
		{
		const double x = v[j * sstride * window];
		sample[j] = (!(x < lo) && !(x > hi))? gdsp_key_of (x) : PC_NO_KEY;
		}
	}

// the same for up to PC_TAB sources in one launch: blockIdx.y picks the source (a launch per chromosome was 24 launches
// of a few microseconds each, issued more slowly than they ran)
#define PC_TAB 32
struct PcSampleTab { const double* v[PC_TAB];  uint64_t* out[PC_TAB];  uint32_t n[PC_TAB]; };
__global__ __launch_bounds__(PC_THREADS)
void pc_sample_tab_kernel (PcSampleTab T, uint32_t window, double lo, double hi, uint32_t sstride)
	{
	const double* __restrict__ v = T.v[blockIdx.y];
	uint64_t* __restrict__ sample = T.out[blockIdx.y];
	const uint32_t n = T.n[blockIdx.y];
	const size_t npop   = ((size_t) n + window - 1) / window;
	const size_t nsamp  = (npop + sstride - 1) / sstride;
	const size_t stride = (size_t) gridDim.x * PC_THREADS;
	for (size_t j = (size_t) blockIdx.x * PC_THREADS + threadIdx.x ; j < nsamp ; j += stride)
		{
		const double x = v[j * sstride * window];
		sample[j] = (!(x < lo) && !(x > hi))? gdsp_key_of (x) : PC_NO_KEY;
		}
	}

// The counting pass.  Nothing is binned and nothing is masked per element: every lane counts, for
// each pivot, the values above it and the values equal to it (a v_cmp_f64 and an add-with-carry
// each), and -- only when the operator has --min/--max bounds -- the values >= lo and > hi.  Counts
// restricted to the population follow by subtraction on the host, because lo <= pivots <= hi:
//     population          = #(x >= lo) - #(x > hi) + NaNs        (NaNs pass the reference's filter)
//     above pivot j       = #(x > p_j) - #(x > hi)
//     below pivot 0       = population - above_0 - equal_0 - positive NaNs
//     between j-1 and j   = #(x > p_{j-1}) - #(x > p_j) - equal_j
//     above the last      = above_{m-1} + positive NaNs          (as keys they lie above every number)
// The compare masks also say which lanes sit strictly inside a collecting bin; their key images
// are appended to the candidate list through a per-wave buffer.  Non-finite values are the rare
// branch (one v_cmp_class per element): NaNs are counted by sign; without bounds the infinities are
// what the default bounds -DBL_MAX..DBL_MAX exclude, so they are counted to be subtracted.
// Elements past the end of a vector are read as -inf, which every count ignores or excludes.
#define PC_CTR_GELO   0
#define PC_CTR_GT     1
#define PC_CTR_EQ     (1 + PC_MAX_PIVOTS)
#define PC_CTR_NANPOS (1 + 2*PC_MAX_PIVOTS)
#define PC_CTR_NANNEG (2 + 2*PC_MAX_PIVOTS)
#define PC_CTR_GTHI   (3 + 2*PC_MAX_PIVOTS)            // bounded: #(x > hi); unbounded: #(+inf)
#define PC_CTR_NEGINF (4 + 2*PC_MAX_PIVOTS)            // unbounded: #(-inf), padding included
#define PC_CTR_WORDS  (5 + 2*PC_MAX_PIVOTS)

// One workgroup walks up to 16 consecutive 32 KiB tiles (a contiguous stretch, dealt so that each XCD
// covers a contiguous eighth of the vector; fewer on short vectors, to keep >= ~1000 workgroups); the
// next tile's loads are in flight while the current one is counted.  (Measured: 2 tiles per workgroup
// is 15 % slower, a 2048-workgroup grid-stride loop the same as this.)  Counters go to one of PC_REPL replicas (workgroup id mod PC_REPL) so that the
// end-of-workgroup atomics do not queue on a handful of addresses; the host adds the replicas.
#define PC_TILE         (PC_THREADS * 16)
#ifndef PC_TILES_PER_WG
#define PC_TILES_PER_WG 16                              // at most
#endif
#ifndef PC_MIN_WGS
#define PC_MIN_WGS      1024
#endif
#ifndef PC_WGS_PER_CU
#define PC_WGS_PER_CU   5                               // of the two-pivot forms (tools/exp_kernel.sh: -DPC_WGS_PER_CU=3 ...)
#endif
#define PC_REPL         64
#define PC_CTR_ALL      (PC_REPL * PC_CTR_WORDS + 1)      // replicas, then the candidate count
#define PC_CTR_SPLIT    (PC_CTR_ALL + 2)                  // split resident route, summed over ranks with the rest: padded elements, lists that overflowed

// state of the resident route (see pc_res_digit_kernel below)
#define PC_RES_MAXP   8                                        // percentiles per call
#define PC_LS_GRID    64                                       // grid keys per percentile (one wave looks them up)
#define PC_LS_KEYS    8192                                     // keys one workgroup can sort in LDS (the cell that holds a candidates' rank)
#define PC_LS_SUB     2048                                     // the strided share of a list a grid is laid on
#define PC_RES_BINS   (1 << 13)
#define PC_RES_SAMPLE 0
#define PC_RES_CAND   1
enum { PC_RES_OK = 0, PC_RES_FEW = 1, PC_RES_PIVOTS = 2, PC_RES_DISAGREE = 3, PC_RES_NOFUSE = 4 };
struct PcResident
	{
	uint32_t status, pad0;
	uint64_t sTotal;                                           // values of the subsample that qualify
	uint64_t selPrefix[2], selK[2], selKey[2];                 // the (at most two) ranks the select in flight is chasing
	uint32_t selDone[2];
	uint64_t bLo[PC_RES_MAXP], bHi[PC_RES_MAXP];               // per percentile: its bracket's ends as keys
	uint32_t openLo[PC_RES_MAXP], openHi[PC_RES_MAXP];         // ... or no end on that side
	uint64_t piv[PC_MAX_PIVOTS];                               // sorted distinct pivots as keys
	PcPivots P;                                                // ... as the counting pass wants them
	double   vLo, vHi;                                         // the fused binarize's bracket, and its ends among the pivots
	int      jLo, jHi;
	uint64_t N, candCount, overflow;                           // after the counting pass: population, candidates kept, list overflowed
	uint64_t bins[2*PC_MAX_PIVOTS + 1];
	uint32_t how[PC_RES_MAXP];                                 // 0: answered (a pivot's ties, or nothing to answer)  1: an order statistic of the candidates  2: needs the plain route  3: the selects in LDS gave up: the candidates' digits, from the host
	uint64_t scopeLo[PC_RES_MAXP], scopeHi[PC_RES_MAXP], rankIn[PC_RES_MAXP], binCount[PC_RES_MAXP];
	uint32_t candTop[PC_RES_MAXP];                             // the bits of the scope's keys below their common prefix
	double   values[PC_RES_MAXP];
	// the selects in one workgroup's LDS (pc_ls_* below): per percentile a grid of ascending distinct keys across the stretch
	// its rank lies in, and whether the candidates' answer has been written already
	uint64_t grid[PC_RES_MAXP][PC_LS_GRID];
	uint32_t gridN[PC_RES_MAXP], candDone[PC_RES_MAXP];
#ifdef PC_RES_TIMING
	unsigned long long dbg[2][5][8];
#endif
	};
#define PC_RES_THREADS 1024
#define PC_RES_MAXB    32                                      // workgroups of a digit pass, at most
struct PcResHist                                               // behind the PcResident in one allocation, zeroed with it before a call
	{
	unsigned long long notMin[2], max[2];                      // ~(smallest key counted), largest key counted: zero = nothing counted
	uint32_t ticket, pad0;
	unsigned long long localCand;                              // split route: candidates THIS device kept (the counter itself is summed over ranks)
	unsigned long long compactCount;                           // pc_ls_cand_pick_kernel: keys kept inside the grid's span
	uint32_t pad[2];
	uint32_t slab[1][2][PC_RES_BINS];                          // the shared histogram(s) of the pass in flight: zero between passes
	};
struct PcPts { uint32_t v[PC_RES_MAXP];  int n; };
#define PC_RES_STATE_BYTES ((sizeof(PcResident) + 255) / 256 * 256)

// FUSE: `= percentile P = binarize --threshold=percentileP` in the same read (logical.c:216-268 behind percentile.c:392-751).
// The threshold T -- the percentile -- is not known yet, but its bracket [vLo, vHi] is: a value above vHi is above T, a
// value below vLo (or a NaN) is not, and both are written as `one` / `zero` right here (16-byte stores, out of place).
// What lies inside the bracket, pivot ties included (0.1 % of real-valued coverage, a few per cent of read depth), is
// left for pc_fixup_kernel: its position goes to a list (wave-aggregated like the candidates), `zero` is written
// meanwhile.  8 B read + 8 B written per base for the pair of operators instead of 24.
struct PcFuse { double vLo, vHi, one, zero;  double* out;  uint32_t* pos;  unsigned long long* posCount;  uint32_t posCap;
                int jLo, jHi; };                              // the bracket's ends as pivots of the counting pass (-1: that side is open)

// bid / nblk: this workgroup's number among the workgroups of its source, and how many those are (the whole grid when the
// launch covers one source; a stretch of it when a table of sources shares one launch, pc_partition_tab_kernel below)
template <int M, bool BOUNDED, bool DENSE, bool FUSE>
__device__ __forceinline__
void pc_partition_body (const double* __restrict__ v, uint32_t n, uint32_t window, double lo, double hi, PcPivots P,
                        unsigned long long* __restrict__ ctr, uint64_t* __restrict__ cand, unsigned long long cap,
                        uint32_t ntiles, PcFuse F, const PcResident* __restrict__ res, const uint32_t bid, const uint32_t nblk)
	{
	constexpr int NC = 2*M + 5;                                // counters of this instantiation
	if (res != NULL)                                           // the resident route: pivots and bracket were decided on the device
		{
		if (res->status != PC_RES_OK) return;
		P = res->P;
		if (FUSE) { F.vLo = res->vLo;  F.vHi = res->vHi;  F.jLo = res->jLo;  F.jHi = res->jHi; }
		}
	__shared__ uint64_t wbuf[PC_THREADS/64][PC_WAVE_BUF];
	__shared__ uint32_t wcount[PC_THREADS/64][NC];
	__shared__ uint32_t ubuf[FUSE? PC_THREADS/64 : 1][FUSE? PC_WAVE_BUF : 1];
	uint32_t uheld = 0;                                        // undecided positions waiting in this wave's buffer (wave uniform)

	const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t   npop = DENSE? (size_t) n : ((size_t) n + window - 1) / window;
	// a workgroup's tiles are bid, + nblk, ...: the workgroups in flight read one compact stretch of the vector
	// between them (a workgroup streaming its own contiguous 512 KiB, next tile prefetched, ran at 5.1 TB/s; this way, with the
	// registers of the prefetch given back for occupancy, 5.5)
	const uint32_t step = nblk;
	unsigned long long* candCount = ctr + (size_t) PC_REPL * PC_CTR_WORDS;
	uint32_t held = 0;                                         // candidates waiting in this wave's buffer (wave uniform)
	uint32_t cGt[M], cEq[M], cGeLo = 0, cGtHi = 0, cNanPos = 0, cNanNeg = 0, cNegInf = 0;   // per lane
	uint64_t take[M+1];                                        // all ones where bin j collects (scalar)
#pragma unroll
	for (int j=0 ; j<M ; j++) { cGt[j] = 0;  cEq[j] = 0; }
#pragma unroll
	for (int j=0 ; j<=M ; j++) take[j] = ((P.collect >> j) & 1)? ~0ULL : 0ULL;
	const uint64_t takeTop = ((P.collect >> P.m) & 1)? ~0ULL : 0ULL;    // bin m (pivots beyond m are NaN pads)

	auto flush = [&] ()
		{
		unsigned long long base = 0;
		if (lane == 0) base = atomicAdd (candCount, (unsigned long long) held);
		base = __shfl (base, 0, 64);
		for (uint32_t i=lane ; i<held ; i+=64) { if (base + i < cap) cand[base + i] = wbuf[wave][i]; }
		held = 0;
		};

	// element e of a tile is read by thread (e/2) % 256 in its load (e/2) / 256: 16 bytes per lane, coalesced
	auto load = [&] (uint32_t tile, double2 (&d)[8])
		{
		const size_t base = (size_t) tile * PC_TILE;
		if (DENSE && (base + PC_TILE <= npop))
			{
			const double2* p = reinterpret_cast<const double2*> (v + base) + threadIdx.x;
#pragma unroll
			for (int u=0 ; u<8 ; u++) d[u] = gdsp_ld2 (&p[u*PC_THREADS]);
			}
		else
			{
#pragma unroll
			for (int u=0 ; u<8 ; u++)
				{
				const size_t e = base + 2 * ((size_t) u*PC_THREADS + threadIdx.x);
				d[u].x = (e   < npop)? v[e       * (DENSE? 1 : window)] : -INFINITY;
				d[u].y = (e+1 < npop)? v[(e + 1) * (DENSE? 1 : window)] : -INFINITY;
				}
			}
		};

	// counter += 1 in the lanes of a compare mask: one add-with-carry fed by the mask.  Written as an
	// instruction because the compiler otherwise sinks these adds below the rare branches and keeps
	// every mask alive until then (eight v_writelane / v_readlane per value)
	auto bump = [] (uint32_t& counter, uint64_t mask)
		{ asm volatile ("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(counter) : "s"(mask) : "vcc"); };

	// The compares.  A double that is finite and not negative orders like its bit pattern, and so does a pivot above zero:
	// "above pivot j" is then a 32-bit integer compare of the high words, settled by the low words only in the waves where
	// some lane's high word equals the pivot's (ties of read depth; one wave in thousands on real-valued coverage).  A
	// wave with a lane that is negative, an infinity or a NaN (one integer compare says so: high word >= 0x7FF00000) takes
	// the FP64 compares, and so does every wave when a pivot is not above zero.  v_cmp_*_f64 issues at half the rate of
	// the 32-bit compares and these were 5 of the ~25 vector instructions per value.
	// (Up to four pivots -- two percentiles: beyond that the pivots' words no longer fit the scalar registers.)
	constexpr bool INTS = (M <= 4);
	uint32_t pH[INTS? M : 1], pL[INTS? M : 1];
	bool intOK = INTS;
	if (INTS)
		{
#pragma unroll
		for (int j=0 ; j<M ; j++)
			{
			pH[j] = (uint32_t) __double2hiint (P.val[j]);  pL[j] = (uint32_t) __double2loint (P.val[j]);
			if ((j < P.m) && !((P.val[j] > 0.0) && (P.val[j] <= DBL_MAX))) intOK = false;
			}
		}
	uint64_t fuseGeLo = 0, fuseGtHi = 0, fuseOdd = 0;              // of the element counted last: at or above the bracket's low pivot, above its high one, not finite
	auto count = [&] (double x)
		{
		uint64_t keep = 0, above = ~0ULL, gthi = 0;
		if (FUSE) { fuseGeLo = ~0ULL;  fuseGtHi = 0; }
		if (BOUNDED)
			{
			above = __ballot (x >= lo);  gthi = __ballot (x > hi);
			bump (cGeLo, above);  bump (cGtHi, gthi);
			}
		const uint32_t xh = (uint32_t) __double2hiint (x), xl = (uint32_t) __double2loint (x);
		const uint64_t special = __ballot (xh >= 0x7FF00000u);     // negative, infinite or NaN somewhere in the wave
		const bool     fast = intOK && (special == 0);
#pragma unroll
		for (int j=0 ; j<M ; j++)
			{
			uint64_t gt, eq;
			if (INTS && fast)
				{
				gt = __ballot (xh > pH[INTS? j : 0]);  eq = 0;
				const uint64_t eqH = __ballot (xh == pH[INTS? j : 0]);
				if (eqH != 0) { gt |= eqH & __ballot (xl > pL[INTS? j : 0]);  eq = eqH & __ballot (xl == pL[INTS? j : 0]); }
				}
			else { gt = __ballot (x >  P.val[j]);  eq = __ballot (x == P.val[j]); }
			bump (cGt[j], gt);  bump (cEq[j], eq);
			if (FUSE) { if (j == F.jLo) fuseGeLo = gt | eq;  if (j == F.jHi) fuseGtHi = gt; }       // (scalar selects)
			keep |= above & ~gt & ~eq & take[j];
			above = gt;
			}
		keep = (keep | (above & take[M])) & ~gthi;             // (with fewer than M pivots the top bin is met inside the loop)
		const uint64_t odd = (special == 0)? 0 : __ballot (!__builtin_isfinite (x));
		if (FUSE) fuseOdd = odd;
		if (odd != 0)
			{
			const bool     nanp = (x != x) && !signbit (x), nann = (x != x) && signbit (x);
			const bool     infp = !BOUNDED && (x ==  INFINITY), infn = !BOUNDED && (x == -INFINITY);
			cNanPos += nanp;  cNanNeg += nann;
			if (!BOUNDED) { cGtHi += infp;  cNegInf += infn; }
			keep &= ~odd;                                      // no infinity is in the population, and a NaN
			keep |= __ballot (nann) & take[0];                 // sits in the bottom or in the top bin
			keep |= __ballot (nanp) & takeTop;
			}
		if (keep != 0)
			{
			if ((keep >> lane) & 1) wbuf[wave][held + __popcll (keep & ((1ULL << lane) - 1))] = gdsp_key_of (x);
			held += (uint32_t) __popcll (keep);
			__builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier ();
			if (held > PC_WAVE_BUF - 64) flush ();
			}
		};

	auto uflush = [&] ()
		{
		unsigned long long base = 0;
		if (lane == 0) base = atomicAdd (F.posCount, (unsigned long long) uheld);
		base = __shfl (base, 0, 64);
		for (uint32_t i=lane ; i<uheld ; i+=64) { if (base + i < F.posCap) F.pos[base + i] = ubuf[wave][i]; }
		uheld = 0;
		};
	// the binarized value of x where the bracket decides it, `zero` and a queued position where it does not.  The masks of
	// the element just counted say where every lane stands (the bracket's ends are pivots of the pass): the value written is
	// a select on the "above the high end" mask itself, the undecided lanes are mask arithmetic; only a wave with an element
	// that is not finite compares again -- an infinity lies beyond an end, a NaN is `zero`.
	const uint32_t oneLo = (uint32_t) __double2loint (F.one),  oneHi = (uint32_t) __double2hiint (F.one);
	const uint32_t zeroLo = (uint32_t) __double2loint (F.zero), zeroHi = (uint32_t) __double2hiint (F.zero);
	auto by_mask = [] (uint64_t mask, uint32_t ifSet, uint32_t ifClear)     // per lane: its bit of the mask picks
		{
		uint32_t r;
		const uint64_t m = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane ((int) (mask >> 32)) << 32)      // (wave uniform already:
		                 | (uint32_t) __builtin_amdgcn_readfirstlane ((int) (uint32_t) mask);                      //  this only tells the compiler)
		asm ("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(ifClear), "v"(ifSet), "s"(m));
		return r;
		};
	auto settle = [&] (double x, size_t e, uint64_t exists)
		{
		uint64_t isOne = fuseGtHi, und = exists & ~fuseGtHi & fuseGeLo;
		if (fuseOdd != 0)
			{
			const bool one = (x > F.vHi);
			isOne = __ballot (one);
			und = exists & __ballot (!one && !(x < F.vLo) && (x == x));
			}
		if (und != 0)
			{
			if ((und >> lane) & 1) ubuf[wave][uheld + __popcll (und & ((1ULL << lane) - 1))] = (uint32_t) e;
			uheld += (uint32_t) __popcll (und);
			__builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier ();
			if (uheld > PC_WAVE_BUF - 64) uflush ();
			}
		return __hiloint2double ((int) by_mask (isOne, oneHi, zeroHi), (int) by_mask (isOne, oneLo, zeroLo));
		};

	// one tile through the generic loads (a strided or unaligned source, the ragged last tile)
	auto tile_plain = [&] (uint32_t tile)
		{
		double2 cur[8];
		load (tile, cur);
		if (!FUSE)
			{
#pragma unroll
			for (int u=0 ; u<8 ; u++) { count (cur[u].x);  count (cur[u].y); }
			}
		else                                                       // (DENSE: element e of the vector is population element e)
			{
			const size_t base = (size_t) tile * PC_TILE;
#pragma unroll
			for (int u=0 ; u<8 ; u++)
				{
				const size_t e = base + 2 * ((size_t) u*PC_THREADS + threadIdx.x);
				count (cur[u].x);
				const double r0 = settle (cur[u].x, e, __ballot (e < npop));
				count (cur[u].y);
				const double r1 = settle (cur[u].y, e + 1, __ballot (e + 1 < npop));
				if (e < npop) F.out[e] = r0;
				if (e + 1 < npop) F.out[e + 1] = r1;
				}
			}
		};
	if (DENSE)
		{
		// whole tiles in two halves: while one half is counted the other half's loads (of this tile, then of the next) are in
		// flight, so a workgroup always has 16 KiB on its way without holding more than one tile in registers (all eight
		// loads up front and none while counting: 3-6 % slower).  Measured on the fused form, 24 chromosomes, same box
		// (tools/exp_kernel.sh): as it stands 388 us per launch = 5.3 TB/s on the 16 B/base it moves; without the counting
		// (load, one compare, store) 365 us, without the stores 253 us -- so it is the read + write streams, not the
		// arithmetic, that set the pace; three halves in flight at 4 workgroups per CU 395 us; each XCD walking its own
		// eighth of the vector 406 us; plain stores 404 us, plain loads 399 us (both are non-temporal here); the four
		// stores of a half issued together behind its counting 388-390 us against 385
		auto load_half = [&] (uint32_t tile, int half, double2 (&d)[4])
			{
			const double2* p = reinterpret_cast<const double2*> (v + (size_t) tile * PC_TILE) + threadIdx.x + half * 4 * PC_THREADS;
#pragma unroll
			for (int u=0 ; u<4 ; u++) d[u] = gdsp_ld2 (&p[u*PC_THREADS]);
			};
		auto work_half = [&] (uint32_t tile, int half, double2 (&d)[4])
			{
#pragma unroll
			for (int u=0 ; u<4 ; u++)
				{
				if (!FUSE) { count (d[u].x);  count (d[u].y); }
				else
					{
					const size_t e = (size_t) tile * PC_TILE + 2 * ((size_t) (half*4 + u) * PC_THREADS + threadIdx.x);
					count (d[u].x);
					const double r0 = settle (d[u].x, e, ~0ULL);
					count (d[u].y);
					const double r1 = settle (d[u].y, e + 1, ~0ULL);
					gdsp_st2 (reinterpret_cast<double2*> (F.out + e), make_double2 (r0, r1));
					}
				}
			};
		const uint32_t whole = (uint32_t) (npop / PC_TILE);
		uint32_t tile = bid;
		double2 A[4], B[4];
		if (tile < whole) load_half (tile, 0, A);
		while (tile < whole)
			{
			load_half (tile, 1, B);
			work_half (tile, 0, A);
			const uint32_t next = tile + step;
			if (next < whole) load_half (next, 0, A);
			work_half (tile, 1, B);
			tile = next;
			}
		if ((whole < ntiles) && (whole % step == bid)) tile_plain (whole);
		}
	else
		for (uint32_t tile=bid ; tile<ntiles ; tile+=step) tile_plain (tile);
	if (held) flush ();
	if (FUSE && uheld) uflush ();

	auto wave_sum = [&] (uint32_t c)
		{
		for (int off=32 ; off>0 ; off>>=1) c += __shfl_down (c, off, 64);
		return c;
		};
	uint32_t mine[NC];
	mine[0] = cGeLo;  mine[1] = cGtHi;  mine[2] = cNanPos;  mine[3] = cNanNeg;  mine[4] = cNegInf;
#pragma unroll
	for (int j=0 ; j<M ; j++) { mine[5+j] = cGt[j];  mine[5+M+j] = cEq[j]; }
#pragma unroll
	for (int k=0 ; k<NC ; k++) { const uint32_t c = wave_sum (mine[k]);  if (lane == 0) wcount[wave][k] = c; }
	__syncthreads ();
	if (threadIdx.x < NC)
		{
		unsigned long long c = 0;
		for (int w=0 ; w<PC_THREADS/64 ; w++) c += wcount[w][threadIdx.x];
		const int t = threadIdx.x;
		const int slot = (t == 0)? PC_CTR_GELO : (t == 1)? PC_CTR_GTHI : (t == 2)? PC_CTR_NANPOS : (t == 3)? PC_CTR_NANNEG
		               : (t == 4)? PC_CTR_NEGINF : (t < 5+M)? PC_CTR_GT + (t-5) : PC_CTR_EQ + (t-5-M);
		if (c) atomicAdd (ctr + (size_t) (blockIdx.x % PC_REPL) * PC_CTR_WORDS + slot, c);
		}
	}

// The launch covers up to PC_TAB sources: source i owns workgroups block0[i] .. block0[i+1]-1, which walk its tiles
// exactly as the launch of its own would.  24 launches of 130 Mbp each ramp up and drain 24 times (4.8 TB/s over the
// genome where one chromosome-sized launch reaches 5.2 under the profiler); here the next source's first workgroups start
// while the last ones of the source before are still counting.
struct PcCountTab
	{
	const double*       v[PC_TAB];
	double*             out[PC_TAB];                               // FUSE: the binarized signal
	uint32_t*           pos[PC_TAB];                               // ... the strip of undecided positions, its count and its capacity
	unsigned long long* posCount[PC_TAB];
	uint32_t            posCap[PC_TAB];
	uint32_t            n[PC_TAB], ntiles[PC_TAB];
	uint32_t            block0[PC_TAB + 1];
	int                 nsrc;
	};
template <int M, bool BOUNDED, bool DENSE, bool FUSE>
__global__ __launch_bounds__(PC_THREADS, (M <= 2)? PC_WGS_PER_CU : 1)     // two pivots: five workgroups per CU (the fused form takes 98 registers otherwise: 4 waves per SIMD)
void pc_partition_tab_kernel (PcCountTab T, uint32_t window, double lo, double hi, PcPivots P,
                              unsigned long long* __restrict__ ctr, uint64_t* __restrict__ cand, unsigned long long cap,
                              PcFuse F, const PcResident* __restrict__ res)
	{
	int i = 0;
	while ((i + 1 < T.nsrc) && (T.block0[i + 1] <= blockIdx.x)) i++;        // (scalar: a few compares per workgroup of up to 16 tiles)
	if (FUSE) { F.out = T.out[i];  F.pos = T.pos[i];  F.posCount = T.posCount[i];  F.posCap = T.posCap[i]; }
	pc_partition_body<M, BOUNDED, DENSE, FUSE> (T.v[i], T.n[i], window, lo, hi, P, ctr, cand, cap, T.ntiles[i], F, res,
	                                            blockIdx.x - T.block0[i], T.block0[i + 1] - T.block0[i]);
	}

// one digit histogram over a short list of keys, restricted to keyLo <= key <= keyHi
__global__ __launch_bounds__(PC_THREADS)
void pc_hist_keys_kernel (const uint64_t* __restrict__ keys, unsigned long long count, uint64_t keyLo, uint64_t keyHi,
                          int bounded, int shift, int bits, uint64_t prefix, unsigned long long* __restrict__ hist)
	{
	__shared__ uint32_t lb[1 << 13];
	const int      nbins = 1 << bits;
	const uint64_t mask  = (uint64_t) nbins - 1;
	const int      above = shift + bits;
	for (int b=threadIdx.x ; b<nbins ; b+=PC_THREADS) lb[b] = 0;
	__syncthreads ();
	uint64_t kmin = ~0ULL, kmax = 0;
	const size_t stride = (size_t) gridDim.x * PC_THREADS;
	for (size_t i = (size_t) blockIdx.x * PC_THREADS + threadIdx.x ; i < count ; i += stride)
		{
		const uint64_t key = keys[i];
		if (key == PC_NO_KEY) continue;
		if (bounded && !((key >= keyLo) && (key <= keyHi))) continue;
		if ((above < 64) && ((key >> above) != (prefix >> above))) continue;
		atomicAdd (&lb[(uint32_t) ((key >> shift) & mask)], 1u);
		if (key < kmin) kmin = key;
		if (key > kmax) kmax = key;
		}
	__syncthreads ();
	for (int b=threadIdx.x ; b<nbins ; b+=PC_THREADS)
		{ uint32_t c = lb[b];  if (c) atomicAdd (&hist[b], (unsigned long long) c); }
	for (int off=32 ; off>0 ; off>>=1)
		{
		uint64_t a = __shfl_down ((unsigned long long) kmin, off, 64);
		uint64_t b = __shfl_down ((unsigned long long) kmax, off, 64);
		if (a < kmin) kmin = a;
		if (b > kmax) kmax = b;
		}
	if (((threadIdx.x & 63) == 0) && (kmin <= kmax))
		{
		atomicMin (&hist[nbins],   (unsigned long long) kmin);
		atomicMax (&hist[nbins+1], (unsigned long long) kmax);
		}
	}

// the bases the bracket left open, now that the threshold is known (logical.c:247-257)
__global__ __launch_bounds__(PC_THREADS)
void pc_fixup_kernel (const double* __restrict__ v, double* __restrict__ out, const uint32_t* __restrict__ pos,
                      const unsigned long long* __restrict__ posCount, uint32_t cap, double T, int tiesAbove, double one, double zero)
	{
	const unsigned long long count = (*posCount < cap)? *posCount : cap;
	const size_t stride = (size_t) gridDim.x * PC_THREADS;
	for (size_t i = (size_t) blockIdx.x * PC_THREADS + threadIdx.x ; i < count ; i += stride)
		{
		const uint32_t e = pos[i];
		const double   x = v[e];
		out[e] = (tiesAbove? (x >= T) : (x > T))? one : zero;
		}
	}

// ... and the whole vector when the one-pass route could not be taken for it (a list that overflowed, an unaligned or
// strided source, a percentile that fell outside its bracket): binarize, out of place
__global__ __launch_bounds__(PC_THREADS)
void pc_binarize_kernel (const double* __restrict__ v, double* __restrict__ out, uint32_t n, double T, int tiesAbove, double one, double zero)
	{
	const size_t stride = (size_t) gridDim.x * PC_THREADS;
	for (size_t i = (size_t) blockIdx.x * PC_THREADS + threadIdx.x ; i < n ; i += stride)
		{
		const double x = v[i];
		out[i] = (tiesAbove? (x >= T) : (x > T))? one : zero;
		}
	}

// ---- the digits after the first, chained on the device.  A radix select looks at one digit per pass and needs the
// bucket of the previous pass before the next can start; with the histogram read back and the bucket picked on the
// host that is a round trip per digit and rank (fourteen of them were two thirds of a 249 Mbp call, more on a box whose
// host is slow).  Here the state of every rank being looked up -- prefix so far, rank within it, the answer -- lives
// in HBM: the histogram kernel takes its prefix from there, a one-workgroup kernel picks the bucket from the (all-
// reduced) histogram, advances the state and clears the histogram for the next pass.  The passes are queued back to
// back; the host reads the answers once.
#define PC_CHAIN_MAX (2 * 16)
struct PcChain { uint64_t prefix[PC_CHAIN_MAX], k[PC_CHAIN_MAX], key[PC_CHAIN_MAX];  uint32_t done[PC_CHAIN_MAX]; };

__global__ __launch_bounds__(PC_THREADS)
void pc_hist_chain_kernel (const uint64_t* __restrict__ keys, unsigned long long count, uint64_t keyLo, uint64_t keyHi,
                           int bounded, int shift, int bits, const PcChain* __restrict__ chain, int r,
                           unsigned long long* __restrict__ hist)
	{
	__shared__ uint32_t lb[1 << 13];
	if (chain->done[r]) return;                                    // (uniform: one word of HBM)
	const uint64_t prefix = chain->prefix[r];
	const int      nbins = 1 << bits;
	const uint64_t mask  = (uint64_t) nbins - 1;
	const int      above = shift + bits;
	for (int b=threadIdx.x ; b<nbins ; b+=PC_THREADS) lb[b] = 0;
	__syncthreads ();
	uint64_t kmin = ~0ULL, kmax = 0;
	const size_t stride = (size_t) gridDim.x * PC_THREADS;
	for (size_t i = (size_t) blockIdx.x * PC_THREADS + threadIdx.x ; i < count ; i += stride)
		{
		const uint64_t key = keys[i];
		if (key == PC_NO_KEY) continue;
		if (bounded && !((key >= keyLo) && (key <= keyHi))) continue;
		if ((above < 64) && ((key >> above) != (prefix >> above))) continue;
		atomicAdd (&lb[(uint32_t) ((key >> shift) & mask)], 1u);
		if (key < kmin) kmin = key;
		if (key > kmax) kmax = key;
		}
	__syncthreads ();
	for (int b=threadIdx.x ; b<nbins ; b+=PC_THREADS)
		{ uint32_t c = lb[b];  if (c) atomicAdd (&hist[b], (unsigned long long) c); }
	for (int off=32 ; off>0 ; off>>=1)
		{
		uint64_t a = __shfl_down ((unsigned long long) kmin, off, 64);
		uint64_t b = __shfl_down ((unsigned long long) kmax, off, 64);
		if (a < kmin) kmin = a;
		if (b > kmax) kmax = b;
		}
	if (((threadIdx.x & 63) == 0) && (kmin <= kmax))
		{
		atomicMin (&hist[nbins],   (unsigned long long) kmin);
		atomicMax (&hist[nbins+1], (unsigned long long) kmax);
		}
	}

// one workgroup: the bucket of rank chain->k[r] in the histogram (gdsp_select_pick), the state advanced, the histogram
// cleared (counts 0, smallest key all ones, largest 0: what gdsp_select_hist_init leaves)
#define PC_PICK_THREADS 1024
__global__ __launch_bounds__(PC_PICK_THREADS)
void pc_pick_kernel (unsigned long long* __restrict__ hist, int shift, int bits, int last, PcChain* __restrict__ chain, int r)
	{
	__shared__ unsigned long long wsum[PC_PICK_THREADS/64];
	const int p = threadIdx.x, lane = p & 63, wave = p >> 6;
	const int nbins = 1 << bits, per = (nbins + PC_PICK_THREADS - 1) / PC_PICK_THREADS;
	if (!chain->done[r])
		{
		const unsigned long long kmin = hist[nbins], kmax = hist[nbins+1], k = chain->k[r];
		unsigned long long mine = 0;
		for (int j=0 ; j<per ; j++) { const int b = p * per + j;  if (b < nbins) mine += hist[b]; }
		unsigned long long incl = mine;
		for (int d=1 ; d<64 ; d*=2) { const unsigned long long up = __shfl_up (incl, d, 64);  if (lane >= d) incl += up; }
		if (lane == 63) wsum[wave] = incl;
		__syncthreads ();
		unsigned long long before = 0;
		for (int w=0 ; w<wave ; w++) before += wsum[w];
		const unsigned long long lo = before + incl - mine, hi = before + incl;      // this thread's bins hold ranks [lo, hi)
		if (kmin == kmax)                                          // one distinct value left (or nothing: then the key is all ones)
			{ if (p == 0) { chain->key[r] = kmin;  chain->done[r] = 1; } }
		else if ((k >= lo) && (k < hi))
			{
			unsigned long long seen = lo;
			for (int j=0 ; j<per ; j++)
				{
				const int b = p * per + j;
				if (b >= nbins) break;
				if (k < seen + hist[b])
					{
					const uint64_t prefix = chain->prefix[r] | (((uint64_t) b) << shift);
					chain->prefix[r] = prefix;  chain->k[r] = k - seen;
					if (last) { chain->key[r] = prefix;  chain->done[r] = 1; }
					break;
					}
				seen += hist[b];
				}
			}
		}
	__syncthreads ();
	for (int b=p ; b<nbins+2 ; b+=PC_PICK_THREADS) hist[b] = (b == nbins)? ~0ULL : 0ULL;
	}

// ---- the resident route: a call on ONE device whose counts nobody else has to see is decided entirely there.  Every
// step of the bracket route that the host used to take between kernels -- the subsample's size, the ranks either side
// of a target, the bucket of a digit, the pivots, the bins and ranks after the counting pass, the scope of the
// candidate select, whether the fused binarize's bracket held -- is taken by the LAST workgroup of the launch whose
// output it needs (an arrival ticket; nobody waits, so nothing can hang) or by a one-workgroup kernel, and its outcome
// stays in HBM (PcResident) where the next launch reads it.  The launches are queued back to back (a boundary is
// 1.5-2 us, cheaper than a grid barrier); the host reads the state ONCE, at the end.  A digit pass chases the two
// ranks of a bracket together (two histograms in LDS when their prefixes have parted).  Anything the route cannot
// settle (a subsample too small to place pivots, a NaN pivot, a fused bracket with a NaN end) sets `status` before
// anything observable has been written: every later kernel returns at once and the host runs the call again the
// old way.
__device__ __forceinline__ uint32_t pc_res_rank (uint64_t N, uint32_t pThousandths)     // gdsp_percentile_rank (percentile.c:587-589, :688-710)
	{
	const uint32_t numValues = (uint32_t) N;
	uint32_t k = (uint32_t) (((uint64_t) numValues) * pThousandths / (100.0*1000));
	if ((numValues != 0) && (k >= numValues)) k = numValues - 1;
	return k;
	}

// one digit of a select over a list of keys.  stage SAMPLE: the subsample, ranks either side of percentile `which`'s
// target (decided after digit 0, when the subsample's size is known), digits at the fixed positions of the plain select;
// stage CAND: the candidates within the scope the bins kernel set for percentile `which`, one rank, 13-bit digits from
// the first bit the scope's ends differ in downwards.  count = min(*countPtr, countCap) when countPtr is given.
// A few big workgroups (<= PC_RES_MAXB of 1024 threads): each counts its share in LDS and adds its non-empty bins to
// the shared histogram (agent-scope atomics: the XCDs' L2s are not coherent; about 5 per ns, 13 us of a 25-30 us pass
// over 243 k keys); the workgroup whose ticket is last reads it (32 bytes per thread), picks and clears it.  Measured
// against it: every workgroup writing its histogram whole to a slab of its own with write-through stores and the last
// one adding the slabs up -- 3 us to write, but one CU reads the 0.5-1 MB of slabs at 40-65 GB/s: 13-22 us, slower.
// Where the time of a pass goes (stamps: -DPC_RES_TIMING prints them per digit): 5-10 us counting, 3 us adding, 2 us
// ticket, 9 us for the last workgroup -- each phase several times its instruction count at 2.4 GHz; seven such passes are
// most of what a 249 Mbp call spends outside its counting pass.  A form with fewer dependent steps, not built: select
// in ONE workgroup's LDS among a strided 16 k of the keys (two ranks, 11-bit digits) for a bracket of the wanted ranks,
// then ONE multi-workgroup pass that counts the keys below / on / between a grid of 64 values across that bracket (a
// few counters per workgroup, no histogram to merge) and keeps each cell's smallest and largest key: the subsample's
// pivots are then the extreme keys of the cells the ranks fall in (any value brackets), the candidates' answer an
// LDS select over the one cell that holds the rank.
// mode WHOLE: as described.  Across ranks (several devices of a process with a communicator, or a process per GPU with a
// device-side reduction hook) a pass is cut at its reduction: mode COUNT adds this device's keys to its histogram(s) and
// stops; the caller all-reduces the histogram words where they lie (a sum of 64 KiB queued on the stream); mode PICK,
// one workgroup, does what the last workgroup does.  Every device then holds the same histogram and the same state, so
// every device takes the same decision; the smallest / largest key seen are left out (they only end a select early).
enum { PC_RES_WHOLE = 0, PC_RES_COUNT = 1, PC_RES_PICK = 2 };
__global__ __launch_bounds__(PC_RES_THREADS)
void pc_res_digit_kernel (const uint64_t* __restrict__ keys, const unsigned long long* __restrict__ countPtr, unsigned long long countCap,
                          int stage, int digit, int which, uint32_t pThousandths, PcResident* __restrict__ R, PcResHist* __restrict__ H,
                          int mode)
	{
	__shared__ uint32_t lb[2][PC_RES_BINS];
	__shared__ unsigned long long sMin[2][PC_RES_THREADS/64], sMax[2][PC_RES_THREADS/64];
	__shared__ unsigned long long sK[2], sBucket[2], sWithin[2];
	__shared__ uint32_t sFound[2], sLast;
	if (R->status != PC_RES_OK) return;
	if ((stage == PC_RES_CAND) && (R->how[which] != 1)) return;
	const bool     a0 = (digit > 0) && !R->selDone[0], a1 = (digit > 0) && !R->selDone[1];
	if ((digit > 0) && !a0 && !a1) return;
	int shift, bits;
	if (stage == PC_RES_SAMPLE) { shift = (digit == 0)? 52 : 52 - 13*digit;  bits = (digit == 0)? 12 : 13; }
	else
		{
		const int top = (int) R->candTop[which] - 13*digit;        // bits not yet looked at (>= 1 at digit 0)
		if (top <= 0) return;                                      // (every rank is settled by then: selDone says so already)
		bits = (top < 13)? top : 13;  shift = top - bits;
		}
	const int      nbins = 1 << bits, above = shift + bits;
	const uint64_t mask  = (uint64_t) nbins - 1;
	const bool     bounded = (stage == PC_RES_CAND);
	const uint64_t keyLo = bounded? R->scopeLo[which] : 0, keyHi = bounded? R->scopeHi[which] : ~0ULL;
	// digit 0 counts everything in scope (the candidates' scope shares every bit above its first digit)
	const uint64_t p0 = (digit == 0)? 0 : R->selPrefix[0] >> (above & 63), p1 = (digit == 0)? 0 : R->selPrefix[1] >> (above & 63);
	const bool     two = a0 && a1 && (p0 != p1);               // the two ranks have parted: a histogram each
	const uint64_t pA  = ((digit == 0) || a0)? p0 : p1, pB = p1;   // histogram 0 counts under pA, histogram 1 (when two) under pB
	unsigned long long count = countCap;
	if (countPtr != NULL) { const unsigned long long c = *countPtr;  if (c < count) count = c; }
#ifdef PC_RES_TIMING
	unsigned long long stamp[8];
#define PC_STAMP(k) stamp[k] = wall_clock64 ()
#else
#define PC_STAMP(k)
#endif
	PC_STAMP (0);

	const int p = threadIdx.x, lane = p & 63, wave = p >> 6;
	if (mode != PC_RES_PICK)
	{
	for (int b=p ; b<nbins ; b+=PC_RES_THREADS) { lb[0][b] = 0;  if (two) lb[1][b] = 0; }
	__syncthreads ();
	uint64_t kminA = ~0ULL, kmaxA = 0, kminB = ~0ULL, kmaxB = 0;
	auto tally = [&] (uint64_t key)
		{
		if (key == PC_NO_KEY) return;
		if (bounded && !((key >= keyLo) && (key <= keyHi))) return;
		const uint64_t top = (above < 64)? (key >> above) : 0;
		if ((digit == 0) || (top == pA))
			{
			atomicAdd (&lb[0][(uint32_t) ((key >> shift) & mask)], 1u);
			if (key < kminA) kminA = key;
			if (key > kmaxA) kmaxA = key;
			}
		else if (two && (top == pB))
			{
			atomicAdd (&lb[1][(uint32_t) ((key >> shift) & mask)], 1u);
			if (key < kminB) kminB = key;
			if (key > kmaxB) kmaxB = key;
			}
		};
	const size_t stride = (size_t) gridDim.x * PC_RES_THREADS;
	size_t i = (size_t) blockIdx.x * PC_RES_THREADS + p;
	for ( ; i < count ; i += 16*stride)                          // sixteen loads in flight (a list of 16 keys per thread: one round)
		{
		uint64_t k[16];
#pragma unroll
		for (int u=0 ; u<16 ; u++) k[u] = (i + u*stride < count)? keys[i + u*stride] : PC_NO_KEY;
#pragma unroll
		for (int u=0 ; u<16 ; u++) tally (k[u]);
		}
	__syncthreads ();
	PC_STAMP (1);
	// into the shared histogram(s): agent-scope adds of this workgroup's non-empty bins
	for (int h=0 ; h<(two? 2 : 1) ; h++)
		for (int b=p ; b<nbins ; b+=PC_RES_THREADS) { const uint32_t c = lb[h][b];  if (c) atomicAdd (&H->slab[0][h][b], c); }
	if (mode == PC_RES_COUNT) return;                              // (the kernel boundary publishes the adds)
	for (int off=32 ; off>0 ; off>>=1)
		{
		uint64_t a = __shfl_down ((unsigned long long) kminA, off, 64), b = __shfl_down ((unsigned long long) kmaxA, off, 64);
		if (a < kminA) kminA = a;
		if (b > kmaxA) kmaxA = b;
		a = __shfl_down ((unsigned long long) kminB, off, 64);  b = __shfl_down ((unsigned long long) kmaxB, off, 64);
		if (a < kminB) kminB = a;
		if (b > kmaxB) kmaxB = b;
		}
	if (lane == 0) { sMin[0][wave] = kminA;  sMax[0][wave] = kmaxA;  sMin[1][wave] = kminB;  sMax[1][wave] = kmaxB; }
	asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave, before the barrier the ticket waits behind
	__syncthreads ();
	PC_STAMP (2);
	if (p == 0)
		{
		for (int h=0 ; h<2 ; h++)
			{
			uint64_t lo = ~0ULL, hi = 0;
			for (int w=0 ; w<PC_RES_THREADS/64 ; w++) { if (sMin[h][w] < lo) lo = sMin[h][w];  if (sMax[h][w] > hi) hi = sMax[h][w]; }
			if (lo <= hi) { atomicMax (&H->notMin[h], (unsigned long long) ~lo);  atomicMax (&H->max[h], (unsigned long long) hi); }
			}
		// release: the adds of every wave of this workgroup (ordered before this point by the barrier) are visible at
		// agent scope before the ticket is; acquire: the last arrival sees every earlier workgroup's
		__threadfence ();
		const uint32_t t = __hip_atomic_fetch_add (&H->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		sLast = (t == gridDim.x - 1)? 1 : 0;
		}
	__syncthreads ();
	if (!sLast) return;
	__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "agent");            // every reading wave, not only the one that drew the ticket
	}
	PC_STAMP (3);

	// ---- the last workgroup: bucket(s) of the rank(s) in the summed histogram(s) (thread p: bins per*p ..), the state
	// advanced.  What thread 0 needs of the state is fetched now, in one round of latency, not word by word as it goes.
	const int per = nbins / PC_RES_THREADS;                        // 8, 4 (12 bits) or fewer (a last digit of under 10 bits: threads past nbins idle)
	const int mineN = (per >= 1)? per : ((p < nbins)? 1 : 0), first = (per >= 1)? per * p : p;
	uint64_t stK[2] = { 0, 0 }, stPrefix[2] = { 0, 0 }, stMin[2] = { 0, 0 }, stMax[2] = { 0, 0 }, stRank = 0, stBin = 0;
	if (p == 0)
		{
		stK[0] = R->selK[0];  stK[1] = R->selK[1];  stPrefix[0] = R->selPrefix[0];  stPrefix[1] = R->selPrefix[1];
		stMin[0] = ~H->notMin[0];  stMin[1] = ~H->notMin[1];  stMax[0] = H->max[0];  stMax[1] = H->max[1];     // (split route: never written, ~0 and 0)
		if (stage == PC_RES_CAND) { stRank = R->rankIn[which];  stBin = R->binCount[which]; }
		}
	uint32_t sum[2][8];
	for (int h=0 ; h<2 ; h++) { for (int j=0 ; j<8 ; j++) sum[h][j] = 0; }
	for (int h=0 ; h<(two? 2 : 1) ; h++)
		{
		if (per == 8)
			{
			const uint4* src = reinterpret_cast<const uint4*> (&H->slab[0][h][8*p]);
			const uint4 x = src[0], y = src[1];
			sum[h][0] = x.x;  sum[h][1] = x.y;  sum[h][2] = x.z;  sum[h][3] = x.w;  sum[h][4] = y.x;  sum[h][5] = y.y;  sum[h][6] = y.z;  sum[h][7] = y.w;
			}
		else if (per == 4)
			{
			const uint4 x = *reinterpret_cast<const uint4*> (&H->slab[0][h][4*p]);
			sum[h][0] = x.x;  sum[h][1] = x.y;  sum[h][2] = x.z;  sum[h][3] = x.w;
			}
		else                                                       // a short last digit: two bins, one or none per thread
			for (int j=0 ; j<mineN ; j++) sum[h][j] = H->slab[0][h][first + j];
		}
	asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
	PC_STAMP (6);
	// inclusive scans of both histograms in the same rounds
	__shared__ unsigned long long sW2[2][PC_RES_THREADS/64];
	unsigned long long mine[2] = { 0, 0 }, incl[2];
#pragma unroll
	for (int j=0 ; j<8 ; j++) { mine[0] += sum[0][j];  mine[1] += sum[1][j]; }
	incl[0] = mine[0];  incl[1] = mine[1];
	for (int d=1 ; d<64 ; d*=2)
		{
		const unsigned long long u0 = __shfl_up (incl[0], d, 64), u1 = __shfl_up (incl[1], d, 64);
		if (lane >= d) { incl[0] += u0;  incl[1] += u1; }
		}
	if (lane == 63) { sW2[0][wave] = incl[0];  sW2[1][wave] = incl[1]; }
	if (p < 2) sFound[p] = 0;
	if (p == 0) { sK[0] = (digit > 0)? stK[0] : ~0ULL;  sK[1] = (digit > 0)? stK[1] : ~0ULL; }
	__syncthreads ();
	unsigned long long lo[2], hi[2], all[2];
	for (int h=0 ; h<2 ; h++)
		{
		unsigned long long before = 0, t = 0;
		for (int w=0 ; w<PC_RES_THREADS/64 ; w++) { const unsigned long long x = sW2[h][w];  if (w < wave) before += x;  t += x; }
		all[h] = t;  lo[h] = before + incl[h] - mine[h];  hi[h] = before + incl[h];
		}
	PC_STAMP (7);
	const int hs[2] = { 0, two? 1 : 0 };                         // the histogram each rank is looked up in
	bool v0 = false, v1 = false;                                   // digit 0: the ranks exist
	if (digit == 0)
		{
		// how many there are decides the ranks (host: pc_run step 2); both are looked up in the one histogram
		if (p == 0)
			{
			const uint64_t total = all[0];
			if (stage == PC_RES_SAMPLE)
				{
				R->sTotal = total;
				if (total < 256) R->status = PC_RES_FEW;
				else
					{
					const double pp = pThousandths / 100000.0;
					const double ks = floor ((double) total * pp);
					const double dl = ceil (4.0 * sqrt ((double) total * pp * (1.0 - pp))) + 16.0;
					v0 = (ks - dl >= 0.0);  v1 = (ks + dl <= (double) total - 1);
					if (v0) sK[0] = (uint64_t) (ks - dl);
					if (v1) sK[1] = (uint64_t) (ks + dl);
					}
				R->openLo[which] = v0? 0 : 1;  R->openHi[which] = v1? 0 : 1;
				}
			else
				{
				if (total != stBin) R->status = PC_RES_DISAGREE;
				else { v0 = true;  sK[0] = stRank; }
				}
			}
		__syncthreads ();
		}
	for (int r=0 ; r<2 ; r++)
		{
		const int h = hs[r];
		const unsigned long long k = sK[r];
		if (!((k >= lo[h]) && (k < hi[h]))) continue;
		unsigned long long seen = lo[h];
#pragma unroll
		for (int j=0 ; j<8 ; j++)
			{
			if ((j < mineN) && (k >= seen) && (k < seen + sum[h][j])) { sBucket[r] = (unsigned long long) (first + j);  sWithin[r] = k - seen;  sFound[r] = 1; }
			seen += sum[h][j];
			}
		}
	__syncthreads ();
	if (p == 0)
		{
		const bool lastDigit = (shift == 0);
		// what every key counted at digit 0 has above it (nothing for the subsample, the scope's common bits for the candidates)
		const uint64_t common = (above < 64)? ((keyLo >> above) << above) : 0;
		uint32_t status = (digit == 0)? R->status : PC_RES_OK;       // (digit 0: what the rank decision just wrote)
		for (int s=0 ; s<2 ; s++)
			{
			const bool active = (digit == 0)? (s == 0? v0 : v1) : (s == 0? a0 : a1);
			if (digit == 0) R->selDone[s] = active? 0 : 1;
			if (!active || (status != PC_RES_OK)) continue;
			const int      h = hs[s];
			const uint64_t kmin = stMin[h], kmax = stMax[h];
			uint64_t key = 0;
			bool     done = false;
			if ((all[h] != 0) && (kmin == kmax)) { key = kmin;  done = true; }         // one distinct value left under this prefix
			else if (!sFound[s]) { status = PC_RES_DISAGREE;  R->status = status;  continue; }
			else
				{
				const uint64_t prefix = ((digit == 0)? common : stPrefix[s]) | (((uint64_t) sBucket[s]) << shift);
				R->selPrefix[s] = prefix;  R->selK[s] = sWithin[s];
				if (lastDigit) { key = prefix;  done = true; }
				}
			if (done)                                                  // a rank that has been settled is written where its stage keeps it
				{
				R->selKey[s] = key;  R->selDone[s] = 1;
				if (stage == PC_RES_SAMPLE) { if (s == 0) R->bLo[which] = key;  else R->bHi[which] = key; }
				else R->values[which] = gdsp_value_of (key);
				}
			}
		}
	__syncthreads ();
	PC_STAMP (4);
	for (int b=p ; b<PC_RES_BINS ; b+=PC_RES_THREADS) { H->slab[0][0][b] = 0;  H->slab[0][1][b] = 0; }
	if (p == 0) { H->notMin[0] = H->notMin[1] = 0;  H->max[0] = H->max[1] = 0;  H->ticket = 0; }
	PC_STAMP (5);
#ifdef PC_RES_TIMING
	if (p == 0) { for (int k=0 ; k<8 ; k++) R->dbg[stage][digit][k] = stamp[k]; }
#endif
	}

// ---- the selects of the resident route in a workgroup's LDS (one device, nobody to reduce with).  A radix select over a
// list costs a launch per digit, each with a histogram to merge and a decision by the last workgroup: ten launches of
// 21 us in a 249 Mbp call whose counting pass takes 380 (profiles/r04_prof_percentile.txt).  But neither stage needs the
// digits.  The SUBSAMPLE only has to yield two keys either side of the target rank: one workgroup sorts a strided 8192 of
// its keys in LDS (pc_ls_sub_kernel) and lays a grid of <= 64 of them across four standard deviations either side of
// the rank; one pass over all the keys then counts them per grid cell (pc_ls_grid_kernel: a binary search per key, a few
// counters per workgroup, no histogram), and the last workgroup takes as the bracket the nearest grid keys outside the
// ranks the digit passes used to chase -- a cell wider at most on either side.  The CANDIDATES' answer is an exact order
// statistic: the same sort and grid over a strided 8192 of the candidates in scope (a list that short is answered on the
// spot), then one pass that counts the cells AND keeps what lies within the grid's span; the last workgroup knows the
// cell that holds the rank (a grid key itself when the rank falls on its ties), gathers that cell's keys from what was
// kept, sorts them in LDS and reads the answer off.  Four launches instead of ten per percentile -- two plus two per
// further percentile.  What the grids miss (a rank outside four standard deviations, a cell too big for LDS, a list kept
// past its capacity) marks the percentile for the plain route (how = 2), like a rank outside its bracket always has.
// (Round 5 built the form sketched in round 4's notes for ONE percentile -- the counting pass also counts its candidates
// into 256 even cells across the bracket, so that ONE launch behind it knows the rank's cell before it looks at a key,
// keeps that cell's keys and sorts them: no strided sort to lay a grid, no pass that counts every candidate per grid
// cell.  Per kernel it does what it should (candidates' launches 35 + 73 us -> 53 per 145 Mbp), per call it is worth
// 15-40 us of 565 (profiles/r05_percentile_hist.txt), and it was taken out again: the cells' flush -- some sixty
// global atomics per workgroup on 256 addresses -- costs the genome-wide call 0.6-0.9 ms of 4.45 (47 k workgroups), and
// even cells in key space cannot answer what the grid does: a median of read depth puts millions of equal keys into one
// cell (the grid counts a key's ties apart), so the two-launch form would have had to stay behind it for every call.)
// Bitonic, ascending; n a power of two <= PC_LS_KEYS; ends with the workgroup in step.  A workgroup barrier per stage made
// a sort of 8192 keys 91 barriers long -- ~55 us on a chip that a single workgroup does not bring up to speed
// (profiles/r04_percentile_lds_steps.txt).  So wave w owns the pairs of keys n/16 w .. n/16 (w+1) - 1: every stage whose
// partners lie n/32 or less apart stays inside one wave's stretch and needs no more than the wave's own LDS accesses in
// order; only the ten stages that reach across stretches (of 91, at 8192 keys) meet at a barrier.  Up to 1024 keys one
// wave sorts alone.
__device__ __forceinline__ void pc_ls_sort (uint64_t* a, int n)
	{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int waves = (n <= 1024)? 1 : PC_RES_THREADS / 64;
	const int per   = (n / 2) / waves;                              // pairs per wave and stage (>= 1: n >= 2)
	if (wave < waves)
		for (int k=2 ; k<=n ; k<<=1)
			for (int j=k>>1 ; j>0 ; j>>=1)
				{
				const bool across = (waves > 1) && (j > per);               // partners in different waves' stretches
				if (across) __syncthreads ();
				else { __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");  __builtin_amdgcn_wave_barrier (); }
				// four pairs per lane at a time, every load before the first store: written pair by pair the compiler keeps
				// each pair's stores ahead of the next pair's loads (it cannot know they do not meet) and a stage costs
				// four LDS round trips instead of one
				for (int r0=0 ; r0<per ; r0+=256)
					{
					uint64_t x[4], y[4];
					int      at[4];
#pragma unroll
					for (int q=0 ; q<4 ; q++)
						{
						const int r = r0 + q*64 + lane;
						const int t = wave * per + ((r < per)? r : 0);
						at[q] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
						x[q] = a[at[q]];  y[q] = a[at[q] + j];
						}
#pragma unroll
					for (int q=0 ; q<4 ; q++)
						{
						const bool up = ((at[q] & k) == 0);
						if ((r0 + q*64 + lane < per) && ((x[q] > y[q]) == up)) { a[at[q]] = y[q];  a[at[q] + j] = x[q]; }
						}
					}
				if (across) __syncthreads ();
				}
	__syncthreads ();
	}

// a strided share of `keys` -- about `sub` of them, sub <= PC_LS_KEYS -- (those within [keyLo, keyHi], PC_NO_KEY skipped) into
// LDS, sorted; returns how many, and the stride in *strideOut.  a[] holds PC_LS_KEYS words; sCount is a word of LDS.
__device__ __forceinline__ int pc_ls_gather_sorted (uint64_t* a, uint32_t* sCount, const uint64_t* __restrict__ keys, unsigned long long count,
                                                    uint64_t keyLo, uint64_t keyHi, unsigned long long sub, unsigned long long* strideOut)
	{
	const unsigned long long stride = (count + sub - 1) / sub;
	if (threadIdx.x == 0) *sCount = 0;
	__syncthreads ();
	const unsigned long long picks = (stride == 0)? 0 : (count + stride - 1) / stride;     // <= sub <= PC_LS_KEYS: eight per thread at most
	const int lane = threadIdx.x & 63;
	uint64_t k[PC_LS_KEYS / PC_RES_THREADS];
#pragma unroll
	for (int u=0 ; u<PC_LS_KEYS/PC_RES_THREADS ; u++)
		{
		const unsigned long long i = (unsigned long long) u * PC_RES_THREADS + threadIdx.x;
		k[u] = (i < picks)? keys[i * stride] : PC_NO_KEY;
		}
#pragma unroll
	for (int u=0 ; u<PC_LS_KEYS/PC_RES_THREADS ; u++)
		{
		if ((unsigned long long) u * PC_RES_THREADS >= picks) break;   // (uniform)
		const bool     in   = (k[u] != PC_NO_KEY) && (k[u] >= keyLo) && (k[u] <= keyHi);
		const uint64_t mask = __ballot (in);                           // (one LDS atomic per wave and round, not one per key on one address)
		if (mask == 0) continue;
		uint32_t base = 0;
		if (lane == 0) base = atomicAdd (sCount, (uint32_t) __popcll (mask));
		base = (uint32_t) __shfl ((int) base, 0, 64);
		if (in) a[base + __popcll (mask & ((1ULL << lane) - 1))] = k[u];
		}
	__syncthreads ();
	const int m = (int) *sCount;
	int n = 2;
	while (n < m) n <<= 1;
	for (int i=m+threadIdx.x ; i<n ; i+=PC_RES_THREADS) a[i] = ~0ULL;
	__syncthreads ();
	pc_ls_sort (a, n);
	*strideOut = stride;
	return m;
	}

// wave `w` lays percentile w's grid: <= PC_LS_GRID ascending distinct keys of the sorted a[0..m) at even steps of rank
// from rLo to rHi (lane g: rank rLo + g (rHi - rLo) / 63)
__device__ __forceinline__ void pc_ls_lay_grid (const uint64_t* a, long long rLo, long long rHi, uint64_t* __restrict__ grid, uint32_t* __restrict__ gridN)
	{
	const int lane = threadIdx.x & 63;
	const long long r = rLo + ((rHi - rLo) * lane) / (PC_LS_GRID - 1);
	const uint64_t key = a[r];
	const uint64_t prev = (uint64_t) __shfl_up ((unsigned long long) key, 1, 64);
	const uint64_t keep = __ballot ((lane == 0) || (key != prev));
	if ((keep >> lane) & 1) grid[__popcll (keep & ((1ULL << lane) - 1))] = key;
	if (lane == 0) *gridN = (uint32_t) __popcll (keep);
	}

// SAMPLE stage, first launch: one workgroup.  Grids for every percentile from a strided 8192 of the subsample.
__global__ __launch_bounds__(PC_RES_THREADS)
void pc_ls_sub_kernel (const uint64_t* __restrict__ keys, unsigned long long slots, PcPts pts, PcResident* __restrict__ R)
	{
	__shared__ uint64_t a[PC_LS_KEYS];
	__shared__ uint32_t sCount;
	if (R->status != PC_RES_OK) return;
	unsigned long long stride;
	const int m = pc_ls_gather_sorted (a, &sCount, keys, slots, 0, ~0ULL, PC_LS_SUB, &stride);
	const int wave = threadIdx.x >> 6;
	if (wave >= pts.n) return;
	if (m == 0) { if ((threadIdx.x & 63) == 0) R->gridN[wave] = 0;  return; }
	const double pp = pts.v[wave] / 100000.0;
	const double ks = floor ((double) m * pp);
	const double dl = ceil (4.0 * sqrt ((double) m * pp * (1.0 - pp))) + 8.0;
	long long rLo = (long long) (ks - dl), rHi = (long long) (ks + dl);
	if (rLo < 0) rLo = 0;
	if (rHi > m - 1) rHi = m - 1;
	pc_ls_lay_grid (a, rLo, rHi, R->grid[wave], &R->gridN[wave]);
	}

// the cell of a key in an ascending grid of distinct keys padded to PC_LS_GRID entries with ~0: j = how many grid keys lie
// below it; *onKey: it IS grid key j.  Six probes whatever the key: the searches of a thread's keys run side by side.
__device__ __forceinline__ int pc_ls_cell (const uint64_t* grid, uint64_t key, bool* onKey)
	{
	int lo = 0;
#pragma unroll
	for (int step=PC_LS_GRID/2 ; step>=1 ; step>>=1) { if (grid[lo + step - 1] < key) lo += step; }
	const uint64_t at = grid[lo];
	if (at < key) { lo++;  *onKey = false; }                       // (above every grid key: only where all 64 are real)
	else *onKey = (at == key);
	return lo;
	}

// cell counters of percentile i in the shared slab (zero between launches, like the digit passes' histograms):
// open[j] = keys strictly between grid keys j-1 and j (j = 0..gn), on[j] = keys equal to grid key j
#define PC_LS_CELLS (PC_LS_GRID + 1)
__device__ __forceinline__ uint32_t* pc_ls_open (PcResHist* H, int i) { return &H->slab[0][0][(2*i)     * PC_LS_CELLS]; }
__device__ __forceinline__ uint32_t* pc_ls_on   (PcResHist* H, int i) { return &H->slab[0][0][(2*i + 1) * PC_LS_CELLS]; }

// the last workgroup of a launch (the digit passes' arrival ticket): true in every thread of it, with the other
// workgroups' adds visible
__device__ __forceinline__ bool pc_ls_last (PcResHist* H, uint32_t* sLast)
	{
	asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads ();
	if (threadIdx.x == 0)
		{
		__threadfence ();
		const uint32_t t = __hip_atomic_fetch_add (&H->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
		*sLast = (t == gridDim.x - 1)? 1 : 0;
		}
	__syncthreads ();
	if (!*sLast) return false;
	__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "agent");
	return true;
	}

// the pivots of the counting pass out of every percentile's bracket (host: pc_run, the end of step 2)
__device__ __noinline__ void pc_res_pivots_body (PcResident* __restrict__ R, int np, int fuseWhich)      // (one thread)
	{
	if (R->status != PC_RES_OK) return;
	uint64_t piv[2*PC_RES_MAXP];
	int m = 0;
	for (int i=0 ; i<np ; i++)
		{
		if (!R->openLo[i]) piv[m++] = R->bLo[i];
		if (!R->openHi[i]) piv[m++] = R->bHi[i];
		}
	for (int a=1 ; a<m ; a++)                                      // (insertion sort of <= 16 keys)
		{ const uint64_t x = piv[a];  int b = a;  while ((b > 0) && (piv[b-1] > x)) { piv[b] = piv[b-1];  b--; }  piv[b] = x; }
	int u = 0;
	for (int a=0 ; a<m ; a++) { if ((a == 0) || (piv[a] != piv[u-1])) piv[u++] = piv[a]; }
	m = u;
	bool usable = (m >= 1);
	for (int j=0 ; j<PC_MAX_PIVOTS ; j++)
		{
		const double val = (j < m)? gdsp_value_of (piv[j]) : NAN;
		R->P.val[j] = val;  R->piv[j] = (j < m)? piv[j] : 0;
		if ((j < m) && (val != val)) usable = false;               // a NaN pivot cannot be compared as a double
		}
	R->P.m = m;
	uint64_t collect = 0;
	for (int j=0 ; j<=m ; j++)                                     // open bin j: above pivot j-1, below pivot j
		for (int i=0 ; i<np ; i++)
			{
			const bool fromBelow = (j == 0)? (R->openLo[i] != 0) : (!R->openLo[i]? (R->bLo[i] <= piv[j-1]) : true);
			const bool toAbove   = (j == m)? (R->openHi[i] != 0) : (!R->openHi[i]? (piv[j] <= R->bHi[i])   : true);
			if (fromBelow && toAbove) collect |= (1ULL << j);
			}
	R->P.collect = collect;
	if (!usable) { R->status = PC_RES_PIVOTS;  return; }
	if (fuseWhich >= 0)
		{
		const int w = fuseWhich;
		const double vLo = R->openLo[w]? -INFINITY : gdsp_value_of (R->bLo[w]);
		const double vHi = R->openHi[w]?  INFINITY : gdsp_value_of (R->bHi[w]);
		int jLo = -1, jHi = -1;
		for (int j=0 ; j<m ; j++)
			{
			if (!R->openLo[w] && (piv[j] == R->bLo[w])) jLo = j;
			if (!R->openHi[w] && (piv[j] == R->bHi[w])) jHi = j;
			}
		R->vLo = vLo;  R->vHi = vHi;  R->jLo = jLo;  R->jHi = jHi;
		if (!((vLo == vLo) && (vHi == vHi)) || (!R->openLo[w] && (jLo < 0)) || (!R->openHi[w] && (jHi < 0))) R->status = PC_RES_NOFUSE;
		}
	}

__global__ void pc_res_pivots_kernel (PcResident* __restrict__ R, int np, int fuseWhich)
	{
	if ((threadIdx.x != 0) || (blockIdx.x != 0)) return;
	pc_res_pivots_body (R, np, fuseWhich);
	}

// SAMPLE stage, second launch: every key of the subsample into its cell of every percentile's grid; the last workgroup
// turns the counts into brackets (what digit 0's rank decision and the last digit's pick did: pc_res_digit_kernel).
__global__ __launch_bounds__(PC_RES_THREADS)
void pc_ls_grid_kernel (const uint64_t* __restrict__ keys, unsigned long long slots, PcPts pts, PcResident* __restrict__ R, PcResHist* __restrict__ H,
                        int fuseWhich)                              // (round 5: the last workgroup goes on to lay the pivots, pc_res_pivots_body: a launch less)
	{
	__shared__ uint64_t grid[PC_RES_MAXP][PC_LS_GRID];
	__shared__ uint32_t gn[PC_RES_MAXP];
	__shared__ uint32_t cOpen[PC_RES_MAXP][PC_LS_CELLS], cOn[PC_RES_MAXP][PC_LS_CELLS];
	__shared__ uint32_t sLast;
	if (R->status != PC_RES_OK) return;
	const int np = pts.n, p = threadIdx.x, lane = p & 63, wave = p >> 6;
	if (p < np) gn[p] = R->gridN[p];
	__syncthreads ();
	for (int q=p ; q<np*PC_LS_GRID ; q+=PC_RES_THREADS)
		grid[q / PC_LS_GRID][q % PC_LS_GRID] = ((uint32_t) (q % PC_LS_GRID) < gn[q / PC_LS_GRID])? R->grid[q / PC_LS_GRID][q % PC_LS_GRID] : ~0ULL;
	for (int q=p ; q<np*PC_LS_CELLS ; q+=PC_RES_THREADS) { cOpen[q / PC_LS_CELLS][q % PC_LS_CELLS] = 0;  cOn[q / PC_LS_CELLS][q % PC_LS_CELLS] = 0; }
	__syncthreads ();
	// nearly every key lies below or above a grid (which spans a few hundred of the subsample's ranks): those are counted
	// in registers, a ballot each; only the keys inside a grid's span are looked up and go through LDS atomics
	uint32_t under[PC_RES_MAXP], over[PC_RES_MAXP];
#pragma unroll
	for (int w=0 ; w<PC_RES_MAXP ; w++) { under[w] = 0;  over[w] = 0; }
	const size_t stride = (size_t) gridDim.x * PC_RES_THREADS;
	for (size_t i = (size_t) blockIdx.x * PC_RES_THREADS + p ; i < slots ; i += 16*stride)
		{
		uint64_t k[16];
#pragma unroll
		for (int u=0 ; u<16 ; u++) k[u] = (i + u*stride < slots)? keys[i + u*stride] : PC_NO_KEY;
#pragma unroll
		for (int w=0 ; w<PC_RES_MAXP ; w++)
			{
			if (w >= np) break;
			const int      g  = (int) gn[w];
			const uint64_t g0 = grid[w][0], g1 = (g > 0)? grid[w][g - 1] : 0;
#pragma unroll
			for (int u=0 ; u<16 ; u++)
				{
				const bool valid = (k[u] != PC_NO_KEY);
				const bool lowK  = valid && ((g == 0) || (k[u] < g0)), highK = valid && (g > 0) && (k[u] > g1);
				under[w] += lowK? 1u : 0u;  over[w] += highK? 1u : 0u;
				const bool inside = valid && !lowK && !highK;
				if (__ballot (inside) == 0) continue;
				if (inside)
					{
					bool onKey;
					const int j = pc_ls_cell (grid[w], k[u], &onKey);
					atomicAdd (onKey? &cOn[w][j] : &cOpen[w][j], 1u);
					}
				}
			}
		}
#pragma unroll
	for (int w=0 ; w<PC_RES_MAXP ; w++)
		{
		if (w >= np) break;
		uint32_t a = under[w], b = over[w];
		for (int off=32 ; off>0 ; off>>=1) { a += __shfl_down (a, off, 64);  b += __shfl_down (b, off, 64); }
		if (lane == 0) { if (a) atomicAdd (&cOpen[w][0], a);  if (b) atomicAdd (&cOpen[w][gn[w]], b); }
		}
	__syncthreads ();
	for (int q=p ; q<np*PC_LS_CELLS ; q+=PC_RES_THREADS)
		{
		const int w = q / PC_LS_CELLS, j = q % PC_LS_CELLS;
		if (cOpen[w][j]) atomicAdd (&pc_ls_open (H, w)[j], cOpen[w][j]);
		if (cOn[w][j])   atomicAdd (&pc_ls_on (H, w)[j],   cOn[w][j]);
		}
	if (!pc_ls_last (H, &sLast)) return;

	// ---- the last workgroup: wave w settles percentile w.  Lane j holds cell j (lane 63 cells 63 and 64).
	if (wave < np)
		{
		const int g = (int) gn[wave];
		const uint32_t* o = pc_ls_open (H, wave);
		const uint32_t* e = pc_ls_on (H, wave);
		unsigned long long mine = (unsigned long long) o[lane] + e[lane];              // cell `lane`: its open stretch, then its key's ties
		const unsigned long long tail = (lane == 63)? o[64] : 0;                       // (above the last grid key)
		unsigned long long incl = mine;
		for (int d=1 ; d<64 ; d*=2) { const unsigned long long up = __shfl_up (incl, d, 64);  if (lane >= d) incl += up; }
		const unsigned long long total = __shfl (incl, 63, 64) + __shfl (tail, 63, 64);
		const unsigned long long below = incl - mine + o[lane];                         // keys strictly below grid key `lane`
		const unsigned long long upto  = incl;                                          // keys at or below it
		const double pp = pts.v[wave] / 100000.0;
		const double ks = floor ((double) total * pp);
		const double dl = ceil (4.0 * sqrt ((double) total * pp * (1.0 - pp))) + 16.0;
		const bool v0 = (total >= 256) && (ks - dl >= 0.0), v1 = (total >= 256) && (ks + dl <= (double) total - 1);
		const unsigned long long kLo = v0? (unsigned long long) (ks - dl) : 0, kHi = v1? (unsigned long long) (ks + dl) : 0;
		// low end: the largest grid key with no more than kLo keys below it (the key of rank kLo is then at or above it)
		const uint64_t okLo = __ballot (v0 && (lane < g) && (below <= kLo));
		// high end: the smallest grid key with more than kHi keys at or below it
		const uint64_t okHi = __ballot (v1 && (lane < g) && (upto >= kHi + 1));
		if (lane == 0)
			{
			if (wave == 0) { R->sTotal = total;  if (total < 256) R->status = PC_RES_FEW; }
			R->openLo[wave] = (okLo != 0)? 0 : 1;  R->openHi[wave] = (okHi != 0)? 0 : 1;
			if (okLo != 0) R->bLo[wave] = grid[wave][63 - __builtin_clzll (okLo)];
			if (okHi != 0) R->bHi[wave] = grid[wave][__builtin_ctzll (okHi)];
			}
		}
	__syncthreads ();
	for (int q=p ; q<np*2*PC_LS_CELLS ; q+=PC_RES_THREADS) H->slab[0][0][q] = 0;
	if (p == 0) { H->ticket = 0;  pc_res_pivots_body (R, np, fuseWhich); }      // (the brackets above were written by this workgroup: visible behind its barrier)
	}

// CAND stage, first launch: one workgroup.  A strided 8192 of percentile `which`'s candidates in scope, sorted; a list
// that short is the whole list and the answer is read off; otherwise a grid around where the rank should fall.
__global__ __launch_bounds__(PC_RES_THREADS)
void pc_ls_cand_sub_kernel (const uint64_t* __restrict__ keys, const unsigned long long* __restrict__ countPtr, unsigned long long countCap,
                            int which, PcResident* __restrict__ R)
	{
	__shared__ uint64_t a[PC_LS_KEYS];
	__shared__ uint32_t sCount;
	if ((R->status != PC_RES_OK) || (R->how[which] != 1)) return;
	unsigned long long count = *countPtr;
	if (count > countCap) count = countCap;
	const uint64_t keyLo = R->scopeLo[which], keyHi = R->scopeHi[which];
	const unsigned long long rank = R->rankIn[which], inBin = R->binCount[which];
	unsigned long long stride;
	// the cell that will hold the rank has about N / (16 sqrt m) keys (N in scope, m of them gathered) and must fit the last
	// workgroup's LDS: long lists are sampled four times as densely
	const int m = pc_ls_gather_sorted (a, &sCount, keys, count, keyLo, keyHi, (count > (1ULL << 20))? PC_LS_KEYS : PC_LS_SUB, &stride);
	if (stride <= 1)                                               // every candidate was looked at
		{
		if (threadIdx.x == 0)
			{
			if (((unsigned long long) m != inBin) || (rank >= inBin)) R->status = PC_RES_DISAGREE;
			else { R->values[which] = gdsp_value_of (a[rank]);  R->candDone[which] = 1; }
			}
		return;
		}
	if (threadIdx.x >= 64) return;
	if (m < 2) { if (threadIdx.x == 0) R->how[which] = 3;  return; }   // (nothing to lay a grid on: the plain route)
	const double q  = (double) rank / (double) inBin;
	const double ks = floor ((double) m * q);
	const double dl = ceil (4.0 * sqrt ((double) m * q * (1.0 - q))) + 8.0;
	long long rLo = (long long) (ks - dl), rHi = (long long) (ks + dl);
	if (rLo < 0) rLo = 0;
	if (rHi > m - 1) rHi = m - 1;
	pc_ls_lay_grid (a, rLo, rHi, R->grid[which], &R->gridN[which]);
	}

// CAND stage, second launch: the candidates in scope counted per cell, those within the grid's span kept (`kept`, up to
// keptCap keys); the last workgroup finds the cell of the rank and sorts its keys.
__global__ __launch_bounds__(PC_RES_THREADS)
void pc_ls_cand_pick_kernel (const uint64_t* __restrict__ keys, const unsigned long long* __restrict__ countPtr, unsigned long long countCap,
                             int which, PcResident* __restrict__ R, PcResHist* __restrict__ H, uint64_t* __restrict__ kept, unsigned long long keptCap,
                             int giveUp)                       // (tests: behave as if the cell had outgrown LDS)
	{
	__shared__ uint64_t a[PC_LS_KEYS];
	__shared__ uint64_t grid[PC_LS_GRID];
	__shared__ uint32_t cOpen[PC_LS_CELLS], cOn[PC_LS_CELLS];
	__shared__ uint32_t sLast, sCount;
	__shared__ long long sCell;
	__shared__ unsigned long long sBefore;
	if ((R->status != PC_RES_OK) || (R->how[which] != 1) || R->candDone[which]) return;
	const int p = threadIdx.x, lane = p & 63;
	const int g = (int) R->gridN[which];
	if (p < PC_LS_GRID) grid[p] = (p < g)? R->grid[which][p] : ~0ULL;
	if (p < PC_LS_CELLS) { cOpen[p] = 0;  cOn[p] = 0; }
	if (p == 0) sCount = 0;
	__syncthreads ();
	unsigned long long count = *countPtr;
	if (count > countCap) count = countCap;
	const uint64_t keyLo = R->scopeLo[which], keyHi = R->scopeHi[which];
	const uint64_t spanLo = grid[0], spanHi = grid[g - 1];
	uint32_t under = 0, over = 0;                                  // in scope, below / above the grid's span: counted in registers
	const size_t stride = (size_t) gridDim.x * PC_RES_THREADS;
	for (size_t i = (size_t) blockIdx.x * PC_RES_THREADS + p ; i < count ; i += 8*stride)
		{
		uint64_t k[8];
#pragma unroll
		for (int u=0 ; u<8 ; u++) k[u] = (i + u*stride < count)? keys[i + u*stride] : PC_NO_KEY;
#pragma unroll
		for (int u=0 ; u<8 ; u++)
			{
			const bool in = (k[u] != PC_NO_KEY) && (k[u] >= keyLo) && (k[u] <= keyHi);
			const bool lowK = in && (k[u] < spanLo), highK = in && (k[u] > spanHi);
			under += lowK? 1u : 0u;  over += highK? 1u : 0u;
			const bool inside = in && !lowK && !highK;
			if (__ballot (inside) == 0) continue;
			bool onKey = false;
			if (inside)
				{
				const int j = pc_ls_cell (grid, k[u], &onKey);
				atomicAdd (onKey? &cOn[j] : &cOpen[j], 1u);
				}
			const bool keep = inside && !onKey;                        // strictly between two grid keys
			const uint64_t mask = __ballot (keep);
			if (mask != 0)                                             // into the workgroup's own buffer: an LDS atomic per wave and round
				{
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd (&sCount, (uint32_t) __popcll (mask));
				base = (uint32_t) __shfl ((int) base, 0, 64);
				const uint32_t at = base + (uint32_t) __popcll (mask & ((1ULL << lane) - 1));
				if (keep && (at < PC_LS_KEYS)) a[at] = k[u];
				}
			}
		}
	__syncthreads ();
	// what the workgroup kept goes out in one piece (one global atomic per workgroup: one per wave and round put nine
	// thousand of them on one address, 12 ns each); more than its buffer holds counts as more than the list holds
		{
		const uint32_t mine = sCount;
		if (p == 0) sBefore = atomicAdd (&H->compactCount, (mine <= PC_LS_KEYS)? (unsigned long long) mine : keptCap + 1);
		__syncthreads ();
		const unsigned long long base = sBefore;
		if (mine <= PC_LS_KEYS)
			for (uint32_t i=p ; i<mine ; i+=PC_RES_THREADS) { if (base + i < keptCap) kept[base + i] = a[i]; }
		}
	for (int off=32 ; off>0 ; off>>=1) { under += __shfl_down (under, off, 64);  over += __shfl_down (over, off, 64); }
	if (lane == 0) { if (under) atomicAdd (&cOpen[0], under);  if (over) atomicAdd (&cOpen[g], over); }
	__syncthreads ();
	if (p < PC_LS_CELLS)
		{
		if (cOpen[p]) atomicAdd (&pc_ls_open (H, 0)[p], cOpen[p]);
		if (cOn[p])   atomicAdd (&pc_ls_on (H, 0)[p],   cOn[p]);
		}
	if (!pc_ls_last (H, &sLast)) return;

	// ---- the last workgroup: the cell of the rank (wave 0), then its keys
	const unsigned long long rank = R->rankIn[which], inBin = R->binCount[which];
	if (p < 64)
		{
		const uint32_t* o = pc_ls_open (H, 0);
		const uint32_t* e = pc_ls_on (H, 0);
		const unsigned long long mine = (unsigned long long) o[lane] + e[lane], tail = (lane == 63)? o[64] : 0;
		unsigned long long incl = mine;
		for (int d=1 ; d<64 ; d*=2) { const unsigned long long up = __shfl_up (incl, d, 64);  if (lane >= d) incl += up; }
		const unsigned long long total = __shfl (incl, 63, 64) + __shfl (tail, 63, 64);
		const unsigned long long before = incl - mine;                                  // keys below cell `lane`'s open stretch
		const bool inOpen = (rank >= before) && (rank < before + o[lane]);
		const bool onKey  = (lane < g) && (rank >= before + o[lane]) && (rank < incl);
		const uint64_t mOpen = __ballot (inOpen), mOn = __ballot (onKey);
		if (lane == 0)
			{
			sCell = -1;  sBefore = 0;
			if (total != inBin) R->status = PC_RES_DISAGREE;
			else if (giveUp) R->how[which] = 3;
			else if (mOn != 0) { R->values[which] = gdsp_value_of (grid[__builtin_ctzll (mOn)]);  R->candDone[which] = 1; }
			else if ((mOpen != 0) && (__builtin_ctzll (mOpen) >= 1) && (__builtin_ctzll (mOpen) < g)
			         && (H->compactCount <= keptCap)) sCell = (long long) __builtin_ctzll (mOpen);
			else R->how[which] = 3;                                // beyond the grid's span, or more kept than there was room for
			}
		if (inOpen && (mOn == 0)) sBefore = before;
		}
	__syncthreads ();
	const long long cell = sCell;
	if (cell >= 1)                                                 // (uniform)
		{
		const uint64_t lo = grid[cell - 1], hi = grid[cell];       // the cell's keys lie strictly between
		const unsigned long long nkept = H->compactCount, want = pc_ls_open (H, 0)[cell];
		if (want > PC_LS_KEYS) { if (p == 0) R->how[which] = 3; }
		else
			{
			if (p == 0) sCount = 0;
			__syncthreads ();
			// (eight loads in flight per thread: one workgroup reading some fifty thousand keys one round trip at a time
			// was most of this launch)
			for (unsigned long long i0=0 ; i0<nkept ; i0+=8*PC_RES_THREADS)
				{
				uint64_t key[8];
#pragma unroll
				for (int u=0 ; u<8 ; u++) { const unsigned long long i = i0 + (unsigned long long) u * PC_RES_THREADS + p;  key[u] = (i < nkept)? kept[i] : 0; }
#pragma unroll
				for (int u=0 ; u<8 ; u++)
					{
					const bool     mine = (key[u] > lo) && (key[u] < hi);   // (a slot past the list reads as 0: below every cell, lo >= the first grid key)
					const uint64_t mask = __ballot (mine);
					if (mask == 0) continue;
					uint32_t base = 0;
					if (lane == 0) base = atomicAdd (&sCount, (uint32_t) __popcll (mask));
					base = (uint32_t) __shfl ((int) base, 0, 64);
					const uint32_t at = base + (uint32_t) __popcll (mask & ((1ULL << lane) - 1));
					if (mine && (at < PC_LS_KEYS)) a[at] = key[u];
					}
				}
			__syncthreads ();
			const int m = (int) sCount;
			if ((unsigned long long) m != want) { if (p == 0) R->status = PC_RES_DISAGREE; }
			else
				{
				int n = 2;
				while (n < m) n <<= 1;
				for (int i=m+p ; i<n ; i+=PC_RES_THREADS) a[i] = ~0ULL;
				__syncthreads ();
				pc_ls_sort (a, n);
				if (p == 0) { R->values[which] = gdsp_value_of (a[rank - sBefore]);  R->candDone[which] = 1; }
				}
			}
		}
	__syncthreads ();
	if (p < 2*PC_LS_CELLS) H->slab[0][0][p] = 0;
	if (p == 0) { H->ticket = 0;  H->compactCount = 0; }
	}

// after the counting pass: the replicas summed, bins, population and ranks (host: pc_run steps 3-4); which percentile is
// a pivot's tie, which an order statistic of the candidates (and of which), which has to take the plain route
// split route, before the counters are summed over ranks: what has to stay local is put aside (the candidates THIS device
// kept: the digit passes over its list read the count from H), what the ranks have to agree on joins the sum (candidates
// kept, elements counted with their padding, lists that overflowed)
__global__ void pc_res_local_kernel (unsigned long long* __restrict__ ctr, unsigned long long candCap, unsigned long long padded,
                                     PcResHist* __restrict__ H)
	{
	if ((threadIdx.x != 0) || (blockIdx.x != 0)) return;
	unsigned long long* tail = ctr + (size_t) PC_REPL * PC_CTR_WORDS;
	unsigned long long c = tail[0];
	const bool over = (c > candCap);
	if (over) c = candCap;
	H->localCand = c;
	tail[0] = c;  tail[1] = padded;  tail[2] = over? 1 : 0;
	}

__global__ __launch_bounds__(PC_THREADS)
void pc_res_bins_kernel (PcResident* __restrict__ R, const unsigned long long* __restrict__ ctr, unsigned long long candCap,
                         int bounded, unsigned long long padded, PcPts pts, int split)
	{
	__shared__ unsigned long long raw[PC_CTR_WORDS];
	if (R->status != PC_RES_OK) return;
	for (int c=threadIdx.x ; c<PC_CTR_WORDS ; c+=PC_THREADS)
		{
		unsigned long long t = 0;
		for (int r=0 ; r<PC_REPL ; r++) t += ctr[(size_t) r * PC_CTR_WORDS + c];
		raw[c] = t;
		}
	__syncthreads ();
	if (threadIdx.x != 0) return;
	const int m = R->P.m, nb = 2*m + 1;
	const unsigned long long* tail = ctr + (size_t) PC_REPL * PC_CTR_WORDS;
	unsigned long long cands = tail[0];
	bool overflow = (cands > candCap);
	if (overflow) cands = candCap;
	if (split) { cands = tail[0];  padded = tail[1];  overflow = (tail[2] != 0); }     // (pc_res_local_kernel, summed over ranks)
	R->candCount = cands;  R->overflow = overflow? 1 : 0;
	const unsigned long long nans  = raw[PC_CTR_NANPOS] + raw[PC_CTR_NANNEG];
	const unsigned long long total = bounded? raw[PC_CTR_GELO] - raw[PC_CTR_GTHI] + nans
	                                        : padded - raw[PC_CTR_GTHI] - raw[PC_CTR_NEGINF];
	uint64_t* bins = R->bins;
	for (int j=0 ; j<m ; j++) bins[2*j+1] = raw[PC_CTR_EQ + j];
	bins[0] = total - (raw[PC_CTR_GT] - raw[PC_CTR_GTHI]) - raw[PC_CTR_EQ] - raw[PC_CTR_NANPOS];
	for (int j=1 ; j<m ; j++) bins[2*j] = raw[PC_CTR_GT + j-1] - raw[PC_CTR_GT + j] - raw[PC_CTR_EQ + j];
	bins[2*m] = (raw[PC_CTR_GT + m-1] - raw[PC_CTR_GTHI]) + raw[PC_CTR_NANPOS];
	unsigned long long N = 0;
	for (int b=0 ; b<nb ; b++) N += bins[b];
	R->N = N;
	for (int i=0 ; i<pts.n ; i++)
		{
		R->how[i] = 0;
		if (N == 0) continue;
		const unsigned long long k = pc_res_rank (N, pts.v[i]);
		unsigned long long before = 0;
		for (int b=0 ; b<nb ; b++)
			{
			if ((k >= before) && (k < before + bins[b]))
				{
				const int j = b >> 1;
				if (b & 1) R->values[i] = gdsp_value_of (R->piv[j]);          // the rank lands on a pivot's ties
				else if (overflow || !((R->P.collect >> j) & 1)) R->how[i] = 2;
				else
					{
					R->how[i] = 1;
					const uint64_t sLo = (j == 0)? 0 : R->piv[j-1] + 1, sHi = (j == m)? ~0ULL : R->piv[j] - 1;
					R->scopeLo[i] = sLo;  R->scopeHi[i] = sHi;
					R->rankIn[i] = k - before;  R->binCount[i] = bins[b];
					R->candTop[i] = (sLo == sHi)? 1 : (uint32_t) (64 - __clzll ((long long) (sLo ^ sHi)));
					}
				break;
				}
			before += bins[b];
			}
		}
	}

// the fused binarize's open positions once the threshold is known on the device: nothing when it is not, when it fell
// outside its bracket or when this source's strip overflowed (the host looks at the same words afterwards and
// binarizes such a source whole)
__global__ __launch_bounds__(PC_THREADS)
void pc_res_fixup_kernel (const double* __restrict__ v, double* __restrict__ out, const uint32_t* __restrict__ pos,
                          const unsigned long long* __restrict__ posCount, uint32_t cap, const PcResident* __restrict__ R, int which,
                          int tiesAbove, double one, double zero)
	{
	if ((R->status != PC_RES_OK) || (R->N == 0) || (R->how[which] >= 2)) return;
	const double T = R->values[which];
	if (!((T >= R->vLo) && (T <= R->vHi))) return;
	const unsigned long long count = *posCount;
	if (count > cap) return;
	const size_t stride = (size_t) gridDim.x * PC_THREADS;
	for (size_t i = (size_t) blockIdx.x * PC_THREADS + threadIdx.x ; i < count ; i += stride)
		{
		const uint32_t e = pos[i];
		const double   x = v[e];
		out[e] = (tiesAbove? (x >= T) : (x > T))? one : zero;
		}
	}

// ... for up to PC_TAB sources in one launch (blockIdx.y picks the source)
struct PcFixTab { const double* v[PC_TAB];  double* out[PC_TAB];  const uint32_t* pos[PC_TAB];  const unsigned long long* posCount[PC_TAB];  uint32_t cap[PC_TAB]; };
__global__ __launch_bounds__(PC_THREADS)
void pc_res_fixup_tab_kernel (PcFixTab T, const PcResident* __restrict__ R, int which, int tiesAbove, double one, double zero)
	{
	if ((R->status != PC_RES_OK) || (R->N == 0) || (R->how[which] >= 2)) return;
	const double thr = R->values[which];
	if (!((thr >= R->vLo) && (thr <= R->vHi))) return;
	const double* __restrict__ v = T.v[blockIdx.y];
	double* __restrict__ out = T.out[blockIdx.y];
	const uint32_t* __restrict__ pos = T.pos[blockIdx.y];
	const unsigned long long count = *T.posCount[blockIdx.y];
	if (count > T.cap[blockIdx.y]) return;
	const size_t stride = (size_t) gridDim.x * PC_THREADS;
	for (size_t i = (size_t) blockIdx.x * PC_THREADS + threadIdx.x ; i < count ; i += stride)
		{
		const uint32_t e = pos[i];
		const double   x = v[e];
		out[e] = (tiesAbove? (x >= thr) : (x > thr))? one : zero;
		}
	}

// ------------------------------------------------------------- host side ----
static const int pcShift[] = { 52, 39, 26, 13, 0 };
static const int pcBits[]  = { 12, 13, 13, 13, 13 };
#define PC_DIGITS 5
#define PC_HIST_WORDS ((1 << 13) + 2)

struct PcDevice                                               // scratch of one device, kept between calls
	{
	int       device;
	uint64_t* hist;       // PC_HIST_WORDS
	uint64_t* ctr;        // PC_CTR_ALL counters of the counting pass
	uint64_t* tmp;        // PC_TMP_WORDS: host scalars on their way through a device-side reduction
	uint64_t* sample;  size_t sampleCap;
	uint64_t* cand;    size_t candCap;
	uint32_t* pos;     size_t posCap;     // fused binarize: positions the bracket left open, every source's strip one after the other
	unsigned long long* posCount;         // one counter per source (PC_MAX_FUSED_SOURCES)
	PcChain*  chain;                      // ranks being looked up by the chained passes
	PcResident* res;                      // the resident route's state, a PcResHist behind it (one allocation)
	};
#define PC_MAX_FUSED_SOURCES 256
#define PC_TMP_WORDS 64
static uint64_t   pcStats[8];
static gdsp_comm*            pcComm    = NULL;                // see gdsp_percentiles_use_comm
static gdsp_device_reduce_fn pcDReduce = NULL;                // see gdsp_percentiles_use_device_reduce
static void*                 pcDReduceCtx = NULL;
static PcDevice   pcDev[64];
static int        pcDevLen = 0;
static std::mutex pcLock;

#define PC_RES_CTR_AT       ((PC_RES_STATE_BYTES + sizeof(PcResHist) + 255) / 256 * 256)
#define PC_RES_POSCOUNT_AT  ((PC_RES_CTR_AT + PC_CTR_SPLIT * sizeof(uint64_t) + 255) / 256 * 256)
#define PC_RES_BLOCK_BYTES  (PC_RES_POSCOUNT_AT + PC_MAX_FUSED_SOURCES * sizeof(unsigned long long))
static int pc_device (int device, size_t sampleCap, size_t candCap, PcDevice** out)
	{
	PcDevice* d = NULL;
	for (int i=0 ; i<pcDevLen ; i++) { if (pcDev[i].device == device) d = &pcDev[i]; }
	if (d == NULL)
		{
		GDSP_REQUIRE (pcDevLen < 64, "too many devices");
		d = &pcDev[pcDevLen++];
		memset (d, 0, sizeof(*d));
		d->device = device;
		GDSP_HIP_TRY (hipMalloc ((void**) &d->hist, PC_HIST_WORDS * sizeof(uint64_t)));
		GDSP_HIP_TRY (hipMalloc ((void**) &d->tmp,  PC_TMP_WORDS * sizeof(uint64_t)));
		GDSP_HIP_TRY (hipMalloc ((void**) &d->chain, sizeof(PcChain)));
		// the resident route's state, the digit passes' histograms, the counting pass's counters and the strips' counts in
		// ONE block: the route clears them with one fill (three fills of 4.5 us each were a twentieth of a 249 Mbp call)
		char* block = NULL;
		GDSP_HIP_TRY (hipMalloc ((void**) &block, PC_RES_BLOCK_BYTES));
		d->res      = reinterpret_cast<PcResident*> (block);
		d->ctr      = reinterpret_cast<uint64_t*> (block + PC_RES_CTR_AT);
		d->posCount = reinterpret_cast<unsigned long long*> (block + PC_RES_POSCOUNT_AT);
		}
	if (d->sampleCap < sampleCap)
		{
		if (d->sample != NULL) GDSP_HIP_TRY (hipFree (d->sample));
		d->sample = NULL;  d->sampleCap = 0;
		GDSP_HIP_TRY (hipMalloc ((void**) &d->sample, sampleCap * sizeof(uint64_t)));
		d->sampleCap = sampleCap;
		}
	if (d->candCap < candCap)
		{
		if (d->cand != NULL) GDSP_HIP_TRY (hipFree (d->cand));
		d->cand = NULL;  d->candCap = 0;
		GDSP_HIP_TRY (hipMalloc ((void**) &d->cand, candCap * sizeof(uint64_t)));
		d->candCap = candCap;
		}
	*out = d;
	return GDSP_OK;
	}

struct PcJob                                                  // one call of gdsp_percentiles
	{
	const gdsp_select_source* src;  int nsrc;
	uint32_t window;  double lo, hi;
	gdsp_reduce_fn reduce;  void* ctx;
	gdsp_comm* comm;                                          // devices of this process all-reduce in HBM
	gdsp_device_reduce_fn dreduce;  void* dctx;               // or: the caller's collective on device words
	std::vector<int>       devices;                           // distinct devices, in order of first use (the communicator's ranks when there is one)
	std::vector<PcDevice*> scratch;                           // same order
	std::vector<void*>     stream;                            // a stream of that device (its first source's)
	std::vector<uint64_t>  sampleCount, candCount;            // per device
	// `percentile = binarize` in one read (gdsp_percentiles_binarize): what is asked, and what the counting pass did about it
	const gdsp_percentile_binarize* fuse;
	std::vector<int>       fusedSource;                       // per source: its slot in the device's position strips, -1 = not fused
	std::vector<size_t>    posOffset, posCap;                 // per source
	double                 vLo, vHi;                          // the bracket the fused stores relied on
	bool                   fusedAny;
	bool                   fixupsDone;                        // the resident route has patched the open positions already ...
	std::vector<std::vector<unsigned long long> > queuedHost; // ... and these are the strips' counts it read back, per device
	};

#define PC_TRY(call) do { int rc_ = (call);  if (rc_ != GDSP_OK) return rc_; } while (0)

static int pc_reduce (PcJob& J, uint64_t* words, size_t count, int op)
	{
	if (J.reduce == NULL) return GDSP_OK;
	if (J.reduce (J.ctx, words, count, op) != 0) { gdsp_set_error ("gdsp_percentiles: the caller's reduction failed");  return GDSP_EHIP; }
	return GDSP_OK;
	}

static bool pc_on_device (const PcJob& J) { return (J.comm != NULL) || (J.dreduce != NULL); }

// all-reduce `count` words in place, in HBM: bufs[d] is device d's copy (J.devices order = communicator ranks)
static int pc_device_allreduce (PcJob& J, const std::vector<uint64_t*>& bufs, size_t count, int op)
	{
	if (J.comm != NULL)
		return gdsp_comm_allreduce_u64 (J.comm, bufs.data (), count, op, J.stream.data ());
	GDSP_HIP_TRY (hipSetDevice (J.devices[0]));
	if (J.dreduce (J.dctx, bufs[0], count, op, J.stream[0]) != 0)
		{ gdsp_set_error ("gdsp_percentiles: the caller's device reduction failed");  return GDSP_EHIP; }
	return GDSP_OK;
	}

// a few host words (already summed over the devices of this process) reduced over ranks
static int pc_reduce_scalars (PcJob& J, uint64_t* words, size_t count, int op)
	{
	if (J.dreduce == NULL) return pc_reduce (J, words, count, op);     // (with a communicator the host sum is already global)
	GDSP_REQUIRE (count <= PC_TMP_WORDS, "too many scalars");
	GDSP_HIP_TRY (hipSetDevice (J.devices[0]));
	hipStream_t s = gdsp_stream (J.stream[0]);
	GDSP_HIP_TRY (hipMemcpyAsync (J.scratch[0]->tmp, words, count * sizeof(uint64_t), hipMemcpyHostToDevice, s));
	std::vector<uint64_t*> one (1, J.scratch[0]->tmp);
	PC_TRY (pc_device_allreduce (J, one, count, op));
	GDSP_HIP_TRY (hipMemcpyAsync (words, J.scratch[0]->tmp, count * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	GDSP_HIP_TRY (hipStreamSynchronize (s));
	return GDSP_OK;
	}

// which keys a histogram pass runs over
enum { PC_OVER_VECTORS, PC_OVER_SAMPLE, PC_OVER_CANDIDATES };
struct PcScope { int over;  int bounded;  uint64_t keyLo, keyHi; };

// one digit histogram over the scope on every device of this process, summed, then reduced over ranks
static int pc_pass (PcJob& J, const PcScope& S, int digit, uint64_t prefix, uint64_t* h)
	{
	const int bits = pcBits[digit], nbins = 1 << bits;
	for (size_t d=0 ; d<J.devices.size () ; d++)
		{
		GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
		PC_TRY (gdsp_select_hist_init (J.scratch[d]->hist, bits, J.stream[d]));
		if (S.over == PC_OVER_VECTORS)
			{
			if (d == 0) pcStats[5]++;
			for (int i=0 ; i<J.nsrc ; i++)
				{
				if (J.src[i].device != J.devices[d]) continue;
				PC_TRY (gdsp_select_histogram (J.src[i].d_v, J.src[i].n, J.window, J.lo, J.hi, pcShift[digit], bits, prefix,
				                               J.scratch[d]->hist, J.stream[d]));
				}
			}
		else
			{
			const uint64_t* keys  = (S.over == PC_OVER_SAMPLE)? J.scratch[d]->sample : J.scratch[d]->cand;
			const uint64_t  count = (S.over == PC_OVER_SAMPLE)? J.sampleCount[d] : J.candCount[d];
			if (count > 0)
				{
				size_t   want   = (count + PC_THREADS*4 - 1) / (PC_THREADS*4);
				uint32_t blocks = (uint32_t) (want > 1024? 1024 : want);
				hipLaunchKernelGGL (pc_hist_keys_kernel, dim3(blocks), dim3(PC_THREADS), 0, gdsp_stream (J.stream[d]),
				                    keys, (unsigned long long) count, S.keyLo, S.keyHi, S.bounded, pcShift[digit], bits, prefix,
				                    (unsigned long long*) J.scratch[d]->hist);
				GDSP_LAUNCH_CHECK ();
				}
			}
		}
	if (pc_on_device (J))
		{
		// sum of the bins, then the smallest / largest matching key, all-reduced where they lie; one copy comes back
		std::vector<uint64_t*> bins (J.devices.size ()), lo (J.devices.size ()), hi (J.devices.size ());
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{ bins[d] = J.scratch[d]->hist;  lo[d] = bins[d] + nbins;  hi[d] = bins[d] + nbins + 1; }
		PC_TRY (pc_device_allreduce (J, bins, nbins, 0));
		PC_TRY (pc_device_allreduce (J, lo, 1, 1));
		PC_TRY (pc_device_allreduce (J, hi, 1, 2));
		GDSP_HIP_TRY (hipSetDevice (J.devices[0]));
		GDSP_HIP_TRY (hipMemcpyAsync (h, J.scratch[0]->hist, (size_t) (nbins + 2) * sizeof(uint64_t),
		                              hipMemcpyDeviceToHost, gdsp_stream (J.stream[0])));
		GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (J.stream[0])));
		if (J.comm != NULL)                                     // the other devices' streams have work queued too
			for (size_t d=1 ; d<J.devices.size () ; d++)
				{ GDSP_HIP_TRY (hipSetDevice (J.devices[d]));  GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (J.stream[d]))); }
		return GDSP_OK;
		}
	std::vector<uint64_t> part (nbins + 2);
	for (int b=0 ; b<nbins+2 ; b++) h[b] = 0;
	h[nbins] = ~(uint64_t) 0;
	for (size_t d=0 ; d<J.devices.size () ; d++)
		{
		GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
		GDSP_HIP_TRY (hipMemcpyAsync (part.data (), J.scratch[d]->hist, (size_t) (nbins + 2) * sizeof(uint64_t),
		                              hipMemcpyDeviceToHost, gdsp_stream (J.stream[d])));
		GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (J.stream[d])));
		for (int b=0 ; b<nbins ; b++) h[b] += part[b];
		if (part[nbins]   < h[nbins])   h[nbins]   = part[nbins];
		if (part[nbins+1] > h[nbins+1]) h[nbins+1] = part[nbins+1];
		}
	PC_TRY (pc_reduce (J, h, nbins, 0));
	PC_TRY (pc_reduce (J, h + nbins, 1, 1));
	PC_TRY (pc_reduce (J, h + nbins + 1, 1, 2));
	return GDSP_OK;
	}

// keys of the given 0-based ranks within the scope; `first` is the scope's first-digit histogram
// the same with the digits after the first chained on the device (see pc_hist_chain_kernel): for key lists, when no host
// hook has to see the histograms -- one process, its devices' histograms all-reduced in HBM or a single device
static int pc_select_chained (PcJob& J, const PcScope& S, const std::vector<uint64_t>& first, const std::vector<uint64_t>& ranks,
                              std::vector<uint64_t>& keys)
	{
	const size_t R = ranks.size ();
	keys.assign (R, 0);
	PcChain init;
	memset (&init, 0, sizeof(init));
	bool anyOpen = false;
	for (size_t r=0 ; r<R ; r++)
		{
		const int nbins0 = 1 << pcBits[0];
		if (first[nbins0] == first[nbins0+1]) { init.key[r] = first[nbins0];  init.done[r] = 1;  continue; }
		uint32_t bucket;  uint64_t within;
		PC_TRY (gdsp_select_pick (first.data (), pcBits[0], ranks[r], &bucket, &within));
		init.prefix[r] = ((uint64_t) bucket) << pcShift[0];  init.k[r] = within;
		anyOpen = true;
		}
	if (anyOpen)
		{
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{
			GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
			GDSP_HIP_TRY (hipMemcpyAsync (J.scratch[d]->chain, &init, sizeof(init), hipMemcpyHostToDevice, gdsp_stream (J.stream[d])));
			PC_TRY (gdsp_select_hist_init (J.scratch[d]->hist, 13, J.stream[d]));
			}
		for (size_t r=0 ; r<R ; r++)
			{
			if (init.done[r]) continue;
			for (int digit=1 ; digit<PC_DIGITS ; digit++)
				{
				const int bits = pcBits[digit], nbins = 1 << bits;
				for (size_t d=0 ; d<J.devices.size () ; d++)
					{
					GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
					const uint64_t* list  = (S.over == PC_OVER_SAMPLE)? J.scratch[d]->sample : J.scratch[d]->cand;
					const uint64_t  count = (S.over == PC_OVER_SAMPLE)? J.sampleCount[d] : J.candCount[d];
					if (count == 0) continue;
					size_t   want   = (count + PC_THREADS*4 - 1) / (PC_THREADS*4);
					uint32_t blocks = (uint32_t) (want > 1024? 1024 : want);
					hipLaunchKernelGGL (pc_hist_chain_kernel, dim3(blocks), dim3(PC_THREADS), 0, gdsp_stream (J.stream[d]),
					                    list, (unsigned long long) count, S.keyLo, S.keyHi, S.bounded, pcShift[digit], bits,
					                    J.scratch[d]->chain, (int) r, (unsigned long long*) J.scratch[d]->hist);
					GDSP_LAUNCH_CHECK ();
					}
				if (J.comm != NULL)
					{
					std::vector<uint64_t*> bins (J.devices.size ()), lo (J.devices.size ()), hi (J.devices.size ());
					for (size_t d=0 ; d<J.devices.size () ; d++)
						{ bins[d] = J.scratch[d]->hist;  lo[d] = bins[d] + nbins;  hi[d] = bins[d] + nbins + 1; }
					PC_TRY (pc_device_allreduce (J, bins, nbins, 0));
					PC_TRY (pc_device_allreduce (J, lo, 1, 1));
					PC_TRY (pc_device_allreduce (J, hi, 1, 2));
					}
				for (size_t d=0 ; d<J.devices.size () ; d++)
					{
					GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
					hipLaunchKernelGGL (pc_pick_kernel, dim3(1), dim3(PC_PICK_THREADS), 0, gdsp_stream (J.stream[d]),
					                    (unsigned long long*) J.scratch[d]->hist, pcShift[digit], bits, (int) (digit == PC_DIGITS-1),
					                    J.scratch[d]->chain, (int) r);
					GDSP_LAUNCH_CHECK ();
					}
				}
			}
		PcChain got;
		GDSP_HIP_TRY (hipSetDevice (J.devices[0]));
		GDSP_HIP_TRY (hipMemcpyAsync (&got, J.scratch[0]->chain, sizeof(got), hipMemcpyDeviceToHost, gdsp_stream (J.stream[0])));
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{ GDSP_HIP_TRY (hipSetDevice (J.devices[d]));  GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (J.stream[d]))); }
		for (size_t r=0 ; r<R ; r++) { if (!init.done[r]) { init.key[r] = got.key[r];  init.done[r] = got.done[r]; } }
		}
	for (size_t r=0 ; r<R ; r++)
		{
		if (!init.done[r]) { gdsp_set_error ("gdsp_percentiles: a chained select did not finish");  return GDSP_EHIP; }
		keys[r] = init.key[r];
		}
	return GDSP_OK;
	}

static int pc_select (PcJob& J, const PcScope& S, const std::vector<uint64_t>& first, const std::vector<uint64_t>& ranks,
                      std::vector<uint64_t>& keys)
	{
	const bool chainable = (S.over != PC_OVER_VECTORS) && (ranks.size () <= PC_CHAIN_MAX) && (J.reduce == NULL) && (J.dreduce == NULL)
	                    && ((J.devices.size () == 1) || (J.comm != NULL)) && (getenv ("GDSP_PERCENTILE_CHAIN_OFF") == NULL);
	if (chainable) return pc_select_chained (J, S, first, ranks, keys);
	std::vector<uint64_t> h (PC_HIST_WORDS);
	keys.assign (ranks.size (), 0);
	for (size_t r=0 ; r<ranks.size () ; r++)
		{
		uint64_t k = ranks[r], prefix = 0;
		bool     found = false;
		for (int digit=0 ; digit<PC_DIGITS ; digit++)
			{
			const int bits = pcBits[digit], nbins = 1 << bits;
			const uint64_t* use = first.data ();
			if (digit > 0) { PC_TRY (pc_pass (J, S, digit, prefix, h.data ()));  use = h.data (); }
			if (use[nbins] == use[nbins+1]) { keys[r] = use[nbins];  found = true;  break; }   // one distinct value left
			uint32_t bucket;  uint64_t within;
			PC_TRY (gdsp_select_pick (use, bits, k, &bucket, &within));
			prefix |= ((uint64_t) bucket) << pcShift[digit];
			k = within;
			}
		if (!found) keys[r] = prefix;
		}
	return GDSP_OK;
	}

static uint64_t pc_total (const std::vector<uint64_t>& first)
	{ uint64_t t = 0;  for (int b=0 ; b<(1 << pcBits[0]) ; b++) t += first[b];  return t; }

// the plain route: radix select over the population for the listed percentiles
static int pc_radix (PcJob& J, const uint32_t* pts, const std::vector<int>& which, double* values, uint64_t* count)
	{
	PcScope S = { PC_OVER_VECTORS, 0, 0, 0 };
	std::vector<uint64_t> first (PC_HIST_WORDS);
	PC_TRY (pc_pass (J, S, 0, 0, first.data ()));
	const uint64_t N = pc_total (first);
	*count = N;
	if (N == 0) return GDSP_OK;
	std::vector<uint64_t> ranks, keys;
	for (int i : which) ranks.push_back (gdsp_percentile_rank ((uint32_t) N, pts[i]));
	PC_TRY (pc_select (J, S, first, ranks, keys));
	for (size_t r=0 ; r<which.size () ; r++) values[which[r]] = gdsp_value_of (keys[r]);
	return GDSP_OK;
	}

// the subsample of every source (pc_run step 1): slots of the sources of a device follow one another in its buffer
static int pc_sample_launch (PcJob& J, uint32_t sstride)
	{
	for (size_t d=0 ; d<J.devices.size () ; d++)
		{
		GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
		size_t at = 0;
		PcSampleTab T;
		int    k = 0;
		size_t most = 0;
		auto go = [&] ()
			{
			if (k == 0) return;
			const size_t   want   = (most + PC_THREADS - 1) / PC_THREADS;
			const uint32_t blocks = (uint32_t) std::min<size_t> (std::max<size_t> (1, PC_MAX_BLOCKS / k), want);
			if (k == 1) hipLaunchKernelGGL (pc_sample_kernel, dim3(blocks), dim3(PC_THREADS), 0, gdsp_stream (J.stream[d]),
			                                T.v[0], T.n[0], J.window, J.lo, J.hi, sstride, T.out[0]);
			else        hipLaunchKernelGGL (pc_sample_tab_kernel, dim3(blocks, k), dim3(PC_THREADS), 0, gdsp_stream (J.stream[d]),
			                                T, J.window, J.lo, J.hi, sstride);
			k = 0;  most = 0;
			};
		for (int i=0 ; i<J.nsrc ; i++)
			{
			if ((J.src[i].device != J.devices[d]) || (J.src[i].n == 0)) continue;
			const size_t p  = ((size_t) J.src[i].n + J.window - 1) / J.window;
			const size_t ns = (p + sstride - 1) / sstride;
			T.v[k] = J.src[i].d_v;  T.n[k] = J.src[i].n;  T.out[k] = J.scratch[d]->sample + at;
			most = std::max (most, ns);
			at += ns;
			if (++k == PC_TAB) { go ();  GDSP_LAUNCH_CHECK (); }
			}
		go ();
		GDSP_LAUNCH_CHECK ();
		J.sampleCount[d] = at;                                     // slots; the ones that qualified are counted below
		}
	return GDSP_OK;
	}

// the counting pass over every source of this process (pc_run step 3): counters and position strips cleared, one launch
// per source.  mUse: the pivots the instantiation has room for (>= P.m).  resident: pivots and the fused bracket are
// read from the device's PcResident instead of P and the arguments.
static int pc_count_launch (PcJob& J, const PcPivots& P, int mUse, bool bounded, bool fuseUsable, int fuseJLo, int fuseJHi,
                            bool resident, uint64_t* padded)
	{
	for (size_t d=0 ; d<J.devices.size () ; d++)
		{
		GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
		if (!resident)                                              // (the resident route cleared them with its state)
			{
			GDSP_HIP_TRY (hipMemsetAsync (J.scratch[d]->ctr, 0, PC_CTR_SPLIT * sizeof(uint64_t), gdsp_stream (J.stream[d])));
			GDSP_HIP_TRY (hipMemsetAsync (J.scratch[d]->posCount, 0, PC_MAX_FUSED_SOURCES * sizeof(unsigned long long), gdsp_stream (J.stream[d])));
			}
		unsigned long long* ctr = (unsigned long long*) J.scratch[d]->ctr;
		const PcResident* RES = resident? J.scratch[d]->res : NULL;
		hipStream_t st = gdsp_stream (J.stream[d]);
		// GDSP_PERCENTILE_COUNT_PER_SOURCE=1: a launch per source (what rounds 2 and 3 measured), for the A/B on one box
		const char* perSourceEnv = getenv ("GDSP_PERCENTILE_COUNT_PER_SOURCE");
		const bool  perSource = (perSourceEnv != NULL) && (strcmp (perSourceEnv, "0") != 0);
		// sources of one flavour (fused / dense / strided) share a launch, up to PC_TAB at a time
		for (int flavour=0 ; flavour<3 ; flavour++)
			{
			PcCountTab T;
			memset (&T, 0, sizeof(T));
			PcFuse F;
			memset (&F, 0, sizeof(F));
			auto go = [&] ()
				{
				if (T.nsrc == 0) return;
				const uint32_t blocks = T.block0[T.nsrc];
				const bool fused = (flavour == 0), dense = (flavour == 1);
#define PC_LAUNCH_B(MM, BB)                                                                                                    \
				do { if (fused) hipLaunchKernelGGL ((pc_partition_tab_kernel<MM, BB, true, true>),  dim3(blocks), dim3(PC_THREADS), 0, \
				                                     st, T, J.window, J.lo, J.hi, P, ctr,                                            \
				                                     J.scratch[d]->cand, (unsigned long long) J.scratch[d]->candCap, F, RES);        \
				     else if (dense) hipLaunchKernelGGL ((pc_partition_tab_kernel<MM, BB, true, false>),  dim3(blocks), dim3(PC_THREADS), 0, \
				                                     st, T, J.window, J.lo, J.hi, P, ctr,                                            \
				                                     J.scratch[d]->cand, (unsigned long long) J.scratch[d]->candCap, F, RES);        \
				     else       hipLaunchKernelGGL ((pc_partition_tab_kernel<MM, BB, false, false>), dim3(blocks), dim3(PC_THREADS), 0, \
				                                     st, T, J.window, J.lo, J.hi, P, ctr,                                            \
				                                     J.scratch[d]->cand, (unsigned long long) J.scratch[d]->candCap, F, RES); } while (0)
#define PC_LAUNCH(MM) do { if (bounded) PC_LAUNCH_B (MM, true);  else PC_LAUNCH_B (MM, false); } while (0)
				if      (mUse <= 2)  PC_LAUNCH (2);
				else if (mUse <= 4)  PC_LAUNCH (4);
				else if (mUse <= 8)  PC_LAUNCH (8);
				else if (mUse <= 16) PC_LAUNCH (16);
				else                PC_LAUNCH (32);
#undef PC_LAUNCH_B
#undef PC_LAUNCH
				T.nsrc = 0;
				};
			for (int i=0 ; i<J.nsrc ; i++)
				{
				if ((J.src[i].device != J.devices[d]) || (J.src[i].n == 0)) continue;
				const size_t   p      = ((size_t) J.src[i].n + J.window - 1) / J.window;
				const uint32_t ntiles = (uint32_t) ((p + PC_TILE - 1) / PC_TILE);
				const uint32_t perWG  = std::max<uint32_t> (1, std::min<uint32_t> (PC_TILES_PER_WG, ntiles / PC_MIN_WGS));
				const uint32_t blocks = (ntiles + perWG - 1) / perWG;
				const bool     dense  = (J.window == 1) && gdsp_aligned16 (J.src[i].d_v);
				const bool     fused  = (J.fusedSource[i] >= 0) && fuseUsable;
				if ((fused? 0 : dense? 1 : 2) != flavour) continue;
				const int k = T.nsrc;
				T.v[k] = J.src[i].d_v;  T.n[k] = J.src[i].n;  T.ntiles[k] = ntiles;  T.block0[k + 1] = T.block0[k] + blocks;
				if (fused)
					{
					F.vLo = J.vLo;  F.vHi = J.vHi;  F.one = J.fuse->one;  F.zero = J.fuse->zero;  F.jLo = fuseJLo;  F.jHi = fuseJHi;
					T.out[k] = J.fuse->d_out[i];
					T.pos[k] = J.scratch[d]->pos + J.posOffset[i];  T.posCount[k] = J.scratch[d]->posCount + J.fusedSource[i];
					T.posCap[k] = (uint32_t) J.posCap[i];
					J.fusedAny = true;
					}
				else J.fusedSource[i] = -1;
				*padded += (uint64_t) ntiles * PC_TILE;
				T.nsrc++;
				if ((T.nsrc == PC_TAB) || perSource) { go ();  GDSP_LAUNCH_CHECK (); }
				}
			go ();
			GDSP_LAUNCH_CHECK ();
			}
		}
	return GDSP_OK;
	}

// the bracket route with every decision taken on the device (see pc_res_digit_kernel).  One device and nobody to reduce
// with: whole digit passes, the last workgroup of each picking.  Across ranks -- the devices of a communicator, or one
// device here and a device-side reduction hook that reaches the other processes -- every pass is cut at its reduction
// (count, all-reduce of the histogram words queued on the stream, pick) and so is the counting pass (counters summed
// before the bins are read): every rank queues the same launches and the same collectives whatever its data, every
// device ends with the same state, and the host reads ONE device's copy ONCE at any world size (percentile.c:547-683
// is the sort this replaces).  *took = false: the route declined (its status word says why, the same on every rank)
// before writing anything a caller could see; pc_run then carries on the old way.
static int pc_resident (PcJob& J, const uint32_t* pThousandths, int np, uint32_t sstride, double* values, uint64_t* count, bool* took)
	{
	*took = false;
	const size_t ND    = J.devices.size ();
	const bool   split = pc_on_device (J);
	std::vector<PcResident*> R (ND);
	std::vector<PcResHist*>  H (ND);
	std::vector<hipStream_t> st (ND);
	for (size_t d=0 ; d<ND ; d++)
		{
		GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
		st[d] = gdsp_stream (J.stream[d]);
		R[d]  = J.scratch[d]->res;
		H[d]  = reinterpret_cast<PcResHist*> (reinterpret_cast<char*> (J.scratch[d]->res) + PC_RES_STATE_BYTES);
		GDSP_HIP_TRY (hipMemsetAsync (R[d], 0, PC_RES_BLOCK_BYTES, st[d]));         // state, histograms, counters, strip counts
		}
	PcPts pts;
	memset (&pts, 0, sizeof(pts));
	pts.n = np;
	for (int i=0 ; i<np ; i++) pts.v[i] = pThousandths[i];

	// one digit pass over every device's list: whole, or count / all-reduce / pick
	auto digit_pass = [&] (int stage, int digit, int which) -> int
		{
		for (size_t d=0 ; d<ND ; d++)
			{
			GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
			PcDevice* sc = J.scratch[d];
			uint32_t blocks;
			const uint64_t* keys;  const unsigned long long* countPtr;  unsigned long long cap;
			if (stage == PC_RES_SAMPLE)
				{
				const unsigned long long slots = J.sampleCount[d];
				blocks = (uint32_t) std::min<unsigned long long> (PC_RES_MAXB, std::max<unsigned long long> (1, (slots + 16383) / 16384));
				keys = sc->sample;  countPtr = NULL;  cap = slots;
				}
			else
				{
				blocks = (uint32_t) std::min<size_t> (PC_RES_MAXB, std::max<size_t> (4, sc->candCap / 65536));
				keys = sc->cand;  cap = (unsigned long long) sc->candCap;
				countPtr = split? &H[d]->localCand : (const unsigned long long*) sc->ctr + (size_t) PC_REPL * PC_CTR_WORDS;
				}
			hipLaunchKernelGGL (pc_res_digit_kernel, dim3(blocks), dim3(PC_RES_THREADS), 0, st[d], keys, countPtr, cap,
			                    stage, digit, which, pts.v[which], R[d], H[d], split? PC_RES_COUNT : PC_RES_WHOLE);
			GDSP_LAUNCH_CHECK ();
			}
		if (!split) return GDSP_OK;
		// both histograms as 64-bit words: pairs of 32-bit counts whose sums stay below 2^32 (pc_run checks the totals)
		std::vector<uint64_t*> slabs (ND);
		for (size_t d=0 ; d<ND ; d++) slabs[d] = reinterpret_cast<uint64_t*> (&H[d]->slab[0][0][0]);
		PC_TRY (pc_device_allreduce (J, slabs, (size_t) 2 * PC_RES_BINS / 2, 0));
		for (size_t d=0 ; d<ND ; d++)
			{
			GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
			hipLaunchKernelGGL (pc_res_digit_kernel, dim3(1), dim3(PC_RES_THREADS), 0, st[d], (const uint64_t*) NULL,
			                    (const unsigned long long*) NULL, 0ULL, stage, digit, which, pts.v[which], R[d], H[d], PC_RES_PICK);
			GDSP_LAUNCH_CHECK ();
			}
		return GDSP_OK;
		};

	// one device, nobody to reduce with: the selects run in a workgroup's LDS (pc_ls_*), two launches per stage instead of a
	// launch per digit.  GDSP_PERCENTILE_LDS_SELECT=0 keeps the digit passes (A/B; the split route always takes them)
	const char* lsEnv = getenv ("GDSP_PERCENTILE_LDS_SELECT");
	const bool  lds = !split && (ND == 1) && !((lsEnv != NULL) && (strcmp (lsEnv, "0") == 0));
	const int fuseWhich = (J.fuse != NULL)? J.fuse->which : -1;
	// subsample, then the ranks either side of every percentile's target: five digits each, no answer awaited
	PC_TRY (pc_sample_launch (J, sstride));
	if (lds)
		{
		const unsigned long long slots = J.sampleCount[0];
		const uint32_t blocks = (uint32_t) std::min<unsigned long long> (PC_RES_MAXB, std::max<unsigned long long> (1, (slots + 16383) / 16384));
		hipLaunchKernelGGL (pc_ls_sub_kernel, dim3(1), dim3(PC_RES_THREADS), 0, st[0], (const uint64_t*) J.scratch[0]->sample, slots, pts, R[0]);
		hipLaunchKernelGGL (pc_ls_grid_kernel, dim3(blocks), dim3(PC_RES_THREADS), 0, st[0], (const uint64_t*) J.scratch[0]->sample, slots, pts, R[0], H[0],
		                    fuseWhich);
		GDSP_LAUNCH_CHECK ();
		}
	else
		for (int i=0 ; i<np ; i++)
			for (int digit=0 ; digit<PC_DIGITS ; digit++) PC_TRY (digit_pass (PC_RES_SAMPLE, digit, i));
	for (size_t d=0 ; (d<ND) && !lds ; d++)                      // (with the selects in LDS the grid launch's last workgroup has laid the pivots)
		{
		GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
		hipLaunchKernelGGL (pc_res_pivots_kernel, dim3(1), dim3(64), 0, st[d], R[d], np, fuseWhich);
		GDSP_LAUNCH_CHECK ();
		}

	// the counting pass reads its pivots (and the fused binarize its bracket) from R
	const bool bounded = !((J.lo == -DBL_MAX) && (J.hi == DBL_MAX));     // (hi = +inf keeps the +inf values: not the defaults)
	PcPivots none;
	memset (&none, 0, sizeof(none));
	uint64_t padded = 0;
	PC_TRY (pc_count_launch (J, none, 2*np, bounded, J.fuse != NULL, -1, -1, true, &padded));
	if (split)
		{
		std::vector<uint64_t*> ctrs (ND);
		for (size_t d=0 ; d<ND ; d++)
			{
			unsigned long long mine = 0;                               // what this device's launches counted, padding included
			for (int i=0 ; i<J.nsrc ; i++)
				{
				if ((J.src[i].device != J.devices[d]) || (J.src[i].n == 0)) continue;
				const size_t pp = ((size_t) J.src[i].n + J.window - 1) / J.window;
				mine += (unsigned long long) ((pp + PC_TILE - 1) / PC_TILE) * PC_TILE;
				}
			GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
			hipLaunchKernelGGL (pc_res_local_kernel, dim3(1), dim3(64), 0, st[d], (unsigned long long*) J.scratch[d]->ctr,
			                    (unsigned long long) J.scratch[d]->candCap, mine, H[d]);
			GDSP_LAUNCH_CHECK ();
			ctrs[d] = J.scratch[d]->ctr;
			}
		PC_TRY (pc_device_allreduce (J, ctrs, PC_CTR_SPLIT, 0));
		}
	for (size_t d=0 ; d<ND ; d++)
		{
		GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
		hipLaunchKernelGGL (pc_res_bins_kernel, dim3(1), dim3(PC_THREADS), 0, st[d], R[d], (const unsigned long long*) J.scratch[d]->ctr,
		                    (unsigned long long) J.scratch[d]->candCap, bounded? 1 : 0, (unsigned long long) padded, pts, split? 1 : 0);
		GDSP_LAUNCH_CHECK ();
		}

	// an order statistic of the candidates for every percentile that landed inside a bracket
	if (lds)
		{
		PcDevice* sc = J.scratch[0];
		const unsigned long long* countPtr = (const unsigned long long*) sc->ctr + (size_t) PC_REPL * PC_CTR_WORDS;
		const uint32_t blocks = (uint32_t) std::min<size_t> (4 * PC_RES_MAXB, std::max<size_t> (4, sc->candCap / 65536));
		for (int i=0 ; i<np ; i++)
			{
			hipLaunchKernelGGL (pc_ls_cand_sub_kernel, dim3(1), dim3(PC_RES_THREADS), 0, st[0], (const uint64_t*) sc->cand, countPtr,
			                    (unsigned long long) sc->candCap, i, R[0]);
			// (the subsample has served: its buffer keeps what the pass finds within the grid's span)
			hipLaunchKernelGGL (pc_ls_cand_pick_kernel, dim3(blocks), dim3(PC_RES_THREADS), 0, st[0], (const uint64_t*) sc->cand, countPtr,
			                    (unsigned long long) sc->candCap, i, R[0], H[0], sc->sample, (unsigned long long) sc->sampleCap,
			                    (getenv ("GDSP_PERCENTILE_LDS_GIVEUP") != NULL)? 1 : 0);
			}
		GDSP_LAUNCH_CHECK ();
		}
	else
		for (int i=0 ; i<np ; i++)
			for (int digit=0 ; digit<PC_DIGITS ; digit++) PC_TRY (digit_pass (PC_RES_CAND, digit, i));
	// the fused binarize's open positions
	if (J.fusedAny)
		{
		for (size_t d=0 ; d<ND ; d++)
			{
			GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
			PcDevice* sc = J.scratch[d];
			PcFixTab T;
			int    k = 0;
			size_t most = 0;
			auto go = [&] ()
				{
				if (k == 0) return;
				const uint32_t blocks = (uint32_t) std::min<size_t> (std::max<size_t> (1, 2048 / k), std::max<size_t> (1, most / 65536));
				if (k == 1) hipLaunchKernelGGL (pc_res_fixup_kernel, dim3(blocks), dim3(PC_THREADS), 0, st[d], T.v[0], T.out[0], T.pos[0], T.posCount[0],
				                                T.cap[0], R[d], fuseWhich, J.fuse->tiesAbove, J.fuse->one, J.fuse->zero);
				else        hipLaunchKernelGGL (pc_res_fixup_tab_kernel, dim3(blocks, k), dim3(PC_THREADS), 0, st[d], T, R[d], fuseWhich,
				                                J.fuse->tiesAbove, J.fuse->one, J.fuse->zero);
				k = 0;  most = 0;
				};
			for (int i=0 ; i<J.nsrc ; i++)
				{
				if ((J.fusedSource[i] < 0) || (J.src[i].device != J.devices[d])) continue;
				T.v[k] = J.src[i].d_v;  T.out[k] = J.fuse->d_out[i];  T.pos[k] = sc->pos + J.posOffset[i];
				T.posCount[k] = sc->posCount + J.fusedSource[i];  T.cap[k] = (uint32_t) J.posCap[i];
				most = std::max<size_t> (most, J.src[i].n);
				if (++k == PC_TAB) { go ();  GDSP_LAUNCH_CHECK (); }
				}
			go ();
			GDSP_LAUNCH_CHECK ();
			}
		}

	// ---- the one read-back of the call: the state from one device (every device's is the same), each device's strips' counts
	PcResident got;
	J.queuedHost.assign (ND, std::vector<unsigned long long> (PC_MAX_FUSED_SOURCES, 0));
	GDSP_HIP_TRY (hipSetDevice (J.devices[0]));
	GDSP_HIP_TRY (hipMemcpyAsync (&got, R[0], sizeof(got), hipMemcpyDeviceToHost, st[0]));
	if (J.fusedAny)
		for (size_t d=0 ; d<ND ; d++)
			{
			GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
			GDSP_HIP_TRY (hipMemcpyAsync (J.queuedHost[d].data (), J.scratch[d]->posCount, PC_MAX_FUSED_SOURCES * sizeof(unsigned long long),
			                              hipMemcpyDeviceToHost, st[d]));
			}
	for (size_t d=0 ; d<ND ; d++) { GDSP_HIP_TRY (hipSetDevice (J.devices[d]));  GDSP_HIP_TRY (hipStreamSynchronize (st[d])); }
	pcStats[2] = got.sTotal;
#ifdef PC_RES_TIMING
	for (int sg=0 ; sg<2 ; sg++)
		for (int dg=0 ; dg<5 ; dg++)
			{
			fprintf (stderr, "stage %d digit %d:", sg, dg);
			for (int k=1 ; k<8 ; k++) fprintf (stderr, " %6.2f", (double) (got.dbg[sg][dg][k] - got.dbg[sg][dg][0]) / 100.0);
			fprintf (stderr, "  us since entry (counted, flushed, ticket, found, cleared, loaded, scanned)\n");
			}
#endif
	if (got.status != PC_RES_OK)
		{
		if (got.status == PC_RES_DISAGREE) { gdsp_set_error ("gdsp_percentiles: candidate list and counts disagree");  return GDSP_EHIP; }
		J.fusedAny = false;                                        // nothing was written: every kernel behind the status word returned at once
		return GDSP_OK;
		}
	*took = true;
	*count = got.N;
	J.candCount[0] = got.candCount;
	J.vLo = got.vLo;  J.vHi = got.vHi;
	pcStats[1] = got.N;  pcStats[3] = got.candCount;  pcStats[5]++;  pcStats[7] = 1;
	if (got.N == 0) return GDSP_OK;
	std::vector<int> fallback;
	for (int i=0 ; i<np ; i++)
		{
		if (got.how[i] == 2) fallback.push_back (i);
		else if (got.how[i] == 3)
			{
			// the grid missed the rank, or its cell outgrew a workgroup's LDS (pc_ls_cand_*): the candidates are all there and
			// so are the rank among them and their scope -- the digit passes over that short list, not over the population
			PcScope inBin = { PC_OVER_CANDIDATES, 1, got.scopeLo[i], got.scopeHi[i] };
			std::vector<uint64_t> cFirst (PC_HIST_WORDS), keys;
			PC_TRY (pc_pass (J, inBin, 0, 0, cFirst.data ()));
			if (pc_total (cFirst) != got.binCount[i]) { gdsp_set_error ("gdsp_percentiles: candidate list and counts disagree");  return GDSP_EHIP; }
			PC_TRY (pc_select (J, inBin, cFirst, std::vector<uint64_t> (1, got.rankIn[i]), keys));
			values[i] = gdsp_value_of (keys[0]);
			pcStats[4]++;                                              // (counted with the fallbacks: more than the one read-back)
			}
		else values[i] = got.values[i];
		}
	J.fixupsDone = J.fusedAny && (got.how[fuseWhich < 0? 0 : fuseWhich] < 2);
	if (!fallback.empty ())
		{
		uint64_t again = 0;
		pcStats[4] += fallback.size ();
		PC_TRY (pc_radix (J, pThousandths, fallback, values, &again));
		}
	return GDSP_OK;
	}

static int pc_run (const gdsp_select_source* sources, int nsources, uint32_t window, double lo, double hi,
                   const uint32_t* pThousandths, int npercentiles, int strategy, uint32_t sampleTarget,
                   gdsp_reduce_fn reduce, void* reduceCtx, double* values, uint64_t* count, PcJob& J)
	{
	GDSP_REQUIRE ((values != NULL) && (count != NULL) && (pThousandths != NULL), "NULL pointer");
	GDSP_REQUIRE ((nsources >= 0) && (npercentiles >= 1), "nothing to do");
	GDSP_REQUIRE ((nsources == 0) || (sources != NULL), "NULL sources");
	GDSP_REQUIRE ((strategy >= GDSP_SELECT_AUTO) && (strategy <= GDSP_SELECT_BRACKET), "unknown strategy");
	for (int i=0 ; i<npercentiles ; i++) GDSP_REQUIRE (pThousandths[i] <= 100000, "percentile above 100");
	if (window == 0) window = 1;
	const bool defaultTarget = (sampleTarget == 0);
	if (defaultTarget) sampleTarget = PC_SAMPLE_TARGET;
	if (!(lo <= hi)) strategy = GDSP_SELECT_RADIX;                // only NaNs can pass such a filter: no brackets
	// a lower bound of -infinity lets the -inf values in (percentile.c:559-561: !(v < lo)), and -inf is what the counting
	// pass reads past the end of a vector: such a call -- the C ABI only; the command line spells infinity DBL_MAX --
	// takes the plain route, whose passes filter value by value (round 5: the bracket route counted the padding in)
	if (lo < -DBL_MAX) strategy = GDSP_SELECT_RADIX;

	memset (pcStats, 0, sizeof(pcStats));
	int homeDevice = 0;
	GDSP_HIP_TRY (hipGetDevice (&homeDevice));

	J.src = sources;  J.nsrc = nsources;  J.window = window;  J.lo = lo;  J.hi = hi;  J.reduce = reduce;  J.ctx = reduceCtx;
	J.comm = pcComm;  J.dreduce = pcDReduce;  J.dctx = pcDReduceCtx;
	GDSP_REQUIRE (!((J.comm != NULL) && (J.dreduce != NULL)), "a communicator and a device reduction hook are both set");
	GDSP_REQUIRE (!(pc_on_device (J) && (reduce != NULL)), "a host reduction hook next to a device-side reduction");
	if (J.comm != NULL)                                           // every rank of the communicator takes part, with or without data
		for (int r=0 ; r<gdsp_comm_size (J.comm) ; r++) { J.devices.push_back (gdsp_comm_device (J.comm, r));  J.stream.push_back (NULL); }
	uint64_t localPop = 0;
	for (int i=0 ; i<nsources ; i++)
		{
		GDSP_REQUIRE ((sources[i].n == 0) || (sources[i].d_v != NULL), "NULL vector");
		localPop += ((uint64_t) sources[i].n + window - 1) / window;
		auto at = std::find (J.devices.begin (), J.devices.end (), sources[i].device);
		if (at == J.devices.end ())
			{
			GDSP_REQUIRE (J.comm == NULL, "a source lives on a device outside the communicator");
			J.devices.push_back (sources[i].device);  J.stream.push_back (sources[i].stream);
			}
		else if (J.stream[at - J.devices.begin ()] == NULL) J.stream[at - J.devices.begin ()] = sources[i].stream;
		}
	GDSP_REQUIRE ((J.dreduce == NULL) || (J.devices.size () <= 1), "the device reduction hook serves one device per process");
	if (J.devices.empty ()) { J.devices.push_back (homeDevice);  J.stream.push_back (NULL); }   // a rank without data still reduces
	uint64_t pop = localPop;
	if (J.dreduce != NULL)                                         // its scratch (the staging words) is needed before the first reduction
		{
		GDSP_HIP_TRY (hipSetDevice (J.devices[0]));
		PcDevice* sc = NULL;
		PC_TRY (pc_device (J.devices[0], 0, 0, &sc));
		J.scratch.push_back (sc);
		PC_TRY (pc_reduce_scalars (J, &pop, 1, 0));
		J.scratch.clear ();
		}
	else PC_TRY (pc_reduce (J, &pop, 1, 0));

	// ---- how: brackets need a population worth sampling and pivots that fit the counting kernel
	const bool bracket = (strategy == GDSP_SELECT_BRACKET)
	                  || ((strategy == GDSP_SELECT_AUTO) && (pop > (uint64_t) sampleTarget) && (2*npercentiles <= PC_MAX_PIVOTS));
	// The subsample only has to place the pivots: a rank is bracketed within 8*sqrt(p(1-p)/s) of the population whatever s
	// is, so s trades the candidates the counting pass keeps (one 8-byte store each, then a few passes over them) against
	// the gather itself -- every sampled value costs a 64-byte sector -- and the ten or so histogram passes over the
	// subsample, each a launch and a host round trip.  One value in 1024, at least 2^16 and at most 2^20 of them
	// (chr1: 243 k sampled, 0.3 % of the chromosome kept as candidates for the 99th percentile; the 3.1 Gbp genome: 2^20,
	// 0.08 %).  Measured against one value in 64 up to 2^24 (round 1): a 249 Mbp call 1.25 -> 0.97 ms, the genome-wide
	// percentile of bench.py's workload 7.1 -> 6.6 ms.
	if (defaultTarget) sampleTarget = (uint32_t) std::min<uint64_t> (PC_SAMPLE_TARGET, std::max<uint64_t> (1u << 16, pop / 1024));
	const uint32_t sstride = (uint32_t) std::max<uint64_t> (1, (pop + sampleTarget - 1) / sampleTarget);

	// scratch per device: the subsample and a candidate list sized for the expected bracket widths
	for (size_t d=0 ; d<J.devices.size () ; d++)
		{
		size_t slots = 0, devPop = 0;
		for (int i=0 ; i<nsources ; i++)
			{
			if (sources[i].device != J.devices[d]) continue;
			const size_t p = ((size_t) sources[i].n + window - 1) / window;
			devPop += p;  slots += (p + sstride - 1) / sstride;
			}
		size_t cands = 0;
		if (bracket)
			{
			// a bracket spans ~8 standard deviations of a subsample rank: 8*sqrt(p(1-p)/s) of the population
			const double s = (double) std::max<uint64_t> (1, pop / sstride);
			cands = (size_t) (devPop * std::min (1.0, npercentiles * 8.0 * 0.5 / sqrt (s)) * 2) + 65536;
			}
		GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
		PcDevice* sc = NULL;
		PC_TRY (pc_device (J.devices[d], bracket? slots + 64 : 0, cands, &sc));
		J.scratch.push_back (sc);
		}
	// fused binarize: a strip of positions per dense source (a sixteenth of its bases; more undecided than that and the
	// source is binarized by a pass of its own)
	J.fusedSource.assign (nsources, -1);  J.posOffset.assign (nsources, 0);  J.posCap.assign (nsources, 0);  J.fusedAny = false;
	if ((J.fuse != NULL) && bracket && (window == 1))
		{
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{
			size_t words = 0;
			int    slot  = 0;
			for (int i=0 ; i<nsources ; i++)
				{
				if ((sources[i].device != J.devices[d]) || (sources[i].n == 0) || !gdsp_aligned16 (sources[i].d_v)
				 || !gdsp_aligned16 (J.fuse->d_out[i]) || (slot == PC_MAX_FUSED_SOURCES)) continue;
				J.fusedSource[i] = slot++;  J.posOffset[i] = words;  J.posCap[i] = (size_t) sources[i].n / 16 + 4096;
				words += J.posCap[i];
				}
			PcDevice* sc = J.scratch[d];
			if (sc->posCap < words)
				{
				GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
				if (sc->pos != NULL) GDSP_HIP_TRY (hipFree (sc->pos));
				sc->pos = NULL;  sc->posCap = 0;
				GDSP_HIP_TRY (hipMalloc ((void**) &sc->pos, (words + words / 8) * sizeof(uint32_t)));
				sc->posCap = words + words / 8;
				}
			}
		}
	J.sampleCount.assign (J.devices.size (), 0);
	J.candCount.assign (J.devices.size (), 0);
	J.fixupsDone = false;

	// one device and nobody to reduce with: the whole route on the device, one read-back
	// (or across ranks with the reductions queued on the streams: a communicator, or a device-side hook.  The conditions are
	// the same on every rank: global quantities only -- a bound on any rank's candidate list in place of its own capacity.)
	const double   sGlobal  = (double) std::max<uint64_t> (1, pop / sstride);
	const double   candsTop = (double) pop * std::min (1.0, npercentiles * 8.0 * 0.5 / sqrt (sGlobal)) * 2 + 65536.0 * 64;
	const bool resident = bracket && ((J.devices.size () == 1) || (J.comm != NULL)) && (J.reduce == NULL) && (npercentiles <= PC_RES_MAXP)
	                   && (candsTop < 4294967296.0) && (pop / sstride + 64 * 64 < (1ULL << 32))
	                   && (getenv ("GDSP_PERCENTILE_RESIDENT_OFF") == NULL);
	if (resident)
		{
		bool took = false;
		pcStats[0] = GDSP_SELECT_BRACKET;
		PC_TRY (pc_resident (J, pThousandths, npercentiles, sstride, values, count, &took));
		if (took) return GDSP_OK;
		}

	std::vector<int> all;
	for (int i=0 ; i<npercentiles ; i++) all.push_back (i);
	int rc = GDSP_OK;
	pcStats[0] = bracket? GDSP_SELECT_BRACKET : GDSP_SELECT_RADIX;
	if (!bracket) { rc = pc_radix (J, pThousandths, all, values, count);  pcStats[1] = *count;  return rc; }

	auto finish = [&] (int code) { return code; };

	// ---- 1. subsample
	PC_TRY (pc_sample_launch (J, sstride));
	PcScope overSample = { PC_OVER_SAMPLE, 0, 0, 0 };
	std::vector<uint64_t> sFirst (PC_HIST_WORDS);
	PC_TRY (pc_pass (J, overSample, 0, 0, sFirst.data ()));
	const uint64_t sTotal = pc_total (sFirst);
	pcStats[2] = sTotal;
	if (sTotal < 256)                                              // hardly anything qualifies
		{ pcStats[0] = GDSP_SELECT_RADIX;  rc = pc_radix (J, pThousandths, all, values, count);  pcStats[1] = *count;  return finish (rc); }

	// ---- 2. pivots: subsample order statistics either side of each target
	std::vector<uint64_t> wantRanks;
	std::vector<int>      slotOf (2*npercentiles, -1);           // -1: open end (0 below, ~0 above)
	for (int i=0 ; i<npercentiles ; i++)
		{
		const double p  = pThousandths[i] / 100000.0;
		const double ks = floor ((double) sTotal * p);
		const double dl = ceil (4.0 * sqrt ((double) sTotal * p * (1.0 - p))) + 16.0;
		if (ks - dl >= 0.0)               { slotOf[2*i]   = (int) wantRanks.size ();  wantRanks.push_back ((uint64_t) (ks - dl)); }
		if (ks + dl <= (double) sTotal-1) { slotOf[2*i+1] = (int) wantRanks.size ();  wantRanks.push_back ((uint64_t) (ks + dl)); }
		}
	std::vector<uint64_t> rankKeys;
	PC_TRY (pc_select (J, overSample, sFirst, wantRanks, rankKeys));
	// a bracket without a subsample rank on one side is open on that side
	std::vector<uint64_t> bLo (npercentiles), bHi (npercentiles), piv;
	std::vector<bool>     openLo (npercentiles), openHi (npercentiles);
	for (int i=0 ; i<npercentiles ; i++)
		{
		openLo[i] = (slotOf[2*i]   < 0);  bLo[i] = openLo[i]? 0 : rankKeys[slotOf[2*i]];
		openHi[i] = (slotOf[2*i+1] < 0);  bHi[i] = openHi[i]? ~(uint64_t) 0 : rankKeys[slotOf[2*i+1]];
		if (!openLo[i]) piv.push_back (bLo[i]);
		if (!openHi[i]) piv.push_back (bHi[i]);
		}
	std::sort (piv.begin (), piv.end ());
	piv.erase (std::unique (piv.begin (), piv.end ()), piv.end ());
	PcPivots P;
	memset (&P, 0, sizeof(P));
	P.m = (int) piv.size ();
	bool usable = (P.m >= 1) && (P.m <= PC_MAX_PIVOTS);
	for (int j=0 ; j<PC_MAX_PIVOTS ; j++)
		{
		P.val[j] = (j < P.m)? gdsp_value_of (piv[j]) : NAN;
		if ((j < P.m) && (P.val[j] != P.val[j])) usable = false;    // a NaN pivot cannot be compared as a double
		}
	if (!usable)
		{ pcStats[0] = GDSP_SELECT_RADIX;  rc = pc_radix (J, pThousandths, all, values, count);  pcStats[1] = *count;  return finish (rc); }
	for (int j=0 ; j<=P.m ; j++)                                 // open bin j: above pivot j-1, below pivot j
		{
		for (int i=0 ; i<npercentiles ; i++)
			{
			const bool fromBelow = (j == 0)?   openLo[i] : (!openLo[i]? (bLo[i] <= piv[j-1]) : true);
			const bool toAbove   = (j == P.m)? openHi[i] : (!openHi[i]? (piv[j] <= bHi[i])   : true);
			if (fromBelow && toAbove) P.collect |= (1ULL << j);
			}
		}

	// ---- 3. the counting pass (bounds as far out as the defaults only keep the infinities away)
	const bool bounded = !((lo == -DBL_MAX) && (hi == DBL_MAX));         // (hi = +inf keeps the +inf values: not the defaults)
	// a fused binarize rests on the bracket of ITS percentile: a NaN end cannot be compared, an open end decides nothing
	bool fuseUsable = false;
	int  fuseJLo = -1, fuseJHi = -1;
	if (J.fuse != NULL)
		{
		const int w = J.fuse->which;
		J.vLo = openLo[w]? -INFINITY : gdsp_value_of (bLo[w]);
		J.vHi = openHi[w]?  INFINITY : gdsp_value_of (bHi[w]);
		fuseUsable = (J.vLo == J.vLo) && (J.vHi == J.vHi);
		for (int j=0 ; j<P.m ; j++)                              // the bracket's ends among the (sorted, distinct) pivots
			{
			if (!openLo[w] && (piv[j] == bLo[w])) fuseJLo = j;
			if (!openHi[w] && (piv[j] == bHi[w])) fuseJHi = j;
			}
		if ((!openLo[w] && (fuseJLo < 0)) || (!openHi[w] && (fuseJHi < 0))) fuseUsable = false;
		}
	uint64_t   padded  = 0;                                      // elements the kernels count, padding included
	PC_TRY (pc_count_launch (J, P, P.m, bounded, fuseUsable, fuseJLo, fuseJHi, false, &padded));
	const int nb = 2*P.m + 1;
	std::vector<uint64_t> raw (PC_CTR_WORDS + 2, 0), part (PC_CTR_ALL);   // + padded count, + an overflowed candidate list anywhere
	const size_t replicated = (size_t) PC_REPL * PC_CTR_WORDS;    // the candidate count behind them stays per device
	if (pc_on_device (J))
		{
		std::vector<uint64_t*> ctrs (J.devices.size ());
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{
			GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
			GDSP_HIP_TRY (hipMemcpyAsync (&J.candCount[d], J.scratch[d]->ctr + replicated, sizeof(uint64_t), hipMemcpyDeviceToHost,
			                              gdsp_stream (J.stream[d])));
			ctrs[d] = J.scratch[d]->ctr;
			}
		PC_TRY (pc_device_allreduce (J, ctrs, replicated, 0));
		GDSP_HIP_TRY (hipSetDevice (J.devices[0]));
		GDSP_HIP_TRY (hipMemcpyAsync (part.data (), J.scratch[0]->ctr, replicated * sizeof(uint64_t), hipMemcpyDeviceToHost,
		                              gdsp_stream (J.stream[0])));
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{ GDSP_HIP_TRY (hipSetDevice (J.devices[d]));  GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (J.stream[d]))); }
		for (int r=0 ; r<PC_REPL ; r++)
			for (int b=0 ; b<PC_CTR_WORDS ; b++) raw[b] += part[(size_t) r * PC_CTR_WORDS + b];
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{ if (J.candCount[d] > J.scratch[d]->candCap) { raw[PC_CTR_WORDS + 1] = 1;  J.candCount[d] = J.scratch[d]->candCap; } }
		raw[PC_CTR_WORDS] = padded;
		PC_TRY (pc_reduce_scalars (J, raw.data () + PC_CTR_WORDS, 1, 0));
		PC_TRY (pc_reduce_scalars (J, raw.data () + PC_CTR_WORDS + 1, 1, 2));
		}
	else
		{
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{
			GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
			GDSP_HIP_TRY (hipMemcpyAsync (part.data (), J.scratch[d]->ctr, PC_CTR_ALL * sizeof(uint64_t), hipMemcpyDeviceToHost,
			                              gdsp_stream (J.stream[d])));
			GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (J.stream[d])));
			for (int r=0 ; r<PC_REPL ; r++)
				for (int b=0 ; b<PC_CTR_WORDS ; b++) raw[b] += part[(size_t) r * PC_CTR_WORDS + b];
			J.candCount[d] = part[replicated];
			if (J.candCount[d] > J.scratch[d]->candCap) { raw[PC_CTR_WORDS + 1] = 1;  J.candCount[d] = J.scratch[d]->candCap; }
			}
		raw[PC_CTR_WORDS] = padded;
		PC_TRY (pc_reduce (J, raw.data (), PC_CTR_WORDS + 1, 0));
		PC_TRY (pc_reduce (J, raw.data () + PC_CTR_WORDS + 1, 1, 2));
		}
	const bool overflow = (raw[PC_CTR_WORDS + 1] != 0);
	// counts within the population (see the kernel's header): every "above" loses what lies above hi
	const uint64_t nans  = raw[PC_CTR_NANPOS] + raw[PC_CTR_NANNEG];
	const uint64_t total = bounded? raw[PC_CTR_GELO] - raw[PC_CTR_GTHI] + nans
	                              : raw[PC_CTR_WORDS] - raw[PC_CTR_GTHI] - raw[PC_CTR_NEGINF];
	std::vector<uint64_t> bins (nb, 0);                          // even: open bins, odd: ties on a pivot
	for (int j=0 ; j<P.m ; j++) bins[2*j+1] = raw[PC_CTR_EQ + j];
	bins[0] = total - (raw[PC_CTR_GT] - raw[PC_CTR_GTHI]) - raw[PC_CTR_EQ] - raw[PC_CTR_NANPOS];
	for (int j=1 ; j<P.m ; j++) bins[2*j] = raw[PC_CTR_GT + j-1] - raw[PC_CTR_GT + j] - raw[PC_CTR_EQ + j];
	bins[2*P.m] = (raw[PC_CTR_GT + P.m-1] - raw[PC_CTR_GTHI]) + raw[PC_CTR_NANPOS];

	// ---- 4. ranks are exact now; read each answer off a pivot or off its bracket's candidates
	uint64_t N = 0;
	for (int b=0 ; b<nb ; b++) N += bins[b];
	*count = N;
	pcStats[1] = N;  pcStats[5]++;
	for (size_t d=0 ; d<J.devices.size () ; d++) pcStats[3] += J.candCount[d];
	if (N == 0) return finish (GDSP_OK);
	std::vector<int> fallback;
	for (int j=0 ; j<=P.m ; j++)
		{
		std::vector<int>      who;
		std::vector<uint64_t> ranks, keys;
		uint64_t before = 0;
		for (int b=0 ; b<2*j ; b++) before += bins[b];
		for (int i=0 ; i<npercentiles ; i++)
			{
			const uint64_t k = gdsp_percentile_rank ((uint32_t) N, pThousandths[i]);
			if ((k >= before) && (k < before + bins[2*j])) { who.push_back (i);  ranks.push_back (k - before); }
			}
		if (who.empty ()) continue;
		if (overflow || !((P.collect >> j) & 1)) { fallback.insert (fallback.end (), who.begin (), who.end ());  continue; }
		PcScope inBin = { PC_OVER_CANDIDATES, 1, (j == 0)? 0 : piv[j-1] + 1, (j == P.m)? ~(uint64_t) 0 : piv[j] - 1 };
		std::vector<uint64_t> cFirst (PC_HIST_WORDS);
		PC_TRY (pc_pass (J, inBin, 0, 0, cFirst.data ()));
		if (pc_total (cFirst) != bins[2*j])
			{ gdsp_set_error ("gdsp_percentiles: candidate list and counts disagree");  return finish (GDSP_EHIP); }
		PC_TRY (pc_select (J, inBin, cFirst, ranks, keys));
		for (size_t r=0 ; r<who.size () ; r++) values[who[r]] = gdsp_value_of (keys[r]);
		}
	for (int i=0 ; i<npercentiles ; i++)
		{
		const uint64_t k = gdsp_percentile_rank ((uint32_t) N, pThousandths[i]);
		uint64_t before = 0;
		for (int b=0 ; b<nb ; b++)
			{
			if ((k >= before) && (k < before + bins[b]))
				{
				if (b & 1) values[i] = gdsp_value_of (piv[b >> 1]);                 // the rank lands on a pivot's ties
				break;
				}
			before += bins[b];
			}
		}
	if (!fallback.empty ())
		{
		std::sort (fallback.begin (), fallback.end ());
		fallback.erase (std::unique (fallback.begin (), fallback.end ()), fallback.end ());
		uint64_t again = 0;
		pcStats[4] = fallback.size ();
		rc = pc_radix (J, pThousandths, fallback, values, &again);
		}
	return finish (rc);
	}

// binarize every source into its output now that the threshold is known: the bases the bracket left open where the
// counting pass wrote the rest, the whole vector elsewhere
static int pc_finish_binarize (PcJob& J, double T, bool* onePass)
	{
	const gdsp_percentile_binarize* f = J.fuse;
	const bool bracketHolds = J.fusedAny && (T >= J.vLo) && (T <= J.vHi);
	// positions queued per fused source: one copy per device
	std::vector<std::vector<unsigned long long> > queued (J.devices.size ());
	if (bracketHolds && J.fixupsDone) queued = J.queuedHost;       // (the resident route: counts read back already)
	else if (bracketHolds)
		{
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{
			queued[d].assign (PC_MAX_FUSED_SOURCES, 0);
			GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
			GDSP_HIP_TRY (hipMemcpyAsync (queued[d].data (), J.scratch[d]->posCount, PC_MAX_FUSED_SOURCES * sizeof(unsigned long long),
			                              hipMemcpyDeviceToHost, gdsp_stream (J.stream[d])));
			}
		for (size_t d=0 ; d<J.devices.size () ; d++)
			{ GDSP_HIP_TRY (hipSetDevice (J.devices[d]));  GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (J.stream[d]))); }
		}
	bool all = bracketHolds;
	for (int i=0 ; i<J.nsrc ; i++)
		{
		if (J.src[i].n == 0) continue;
		size_t d = 0;
		while (J.devices[d] != J.src[i].device) d++;
		GDSP_HIP_TRY (hipSetDevice (J.devices[d]));
		hipStream_t st = gdsp_stream (J.stream[d]);
		const int slot = bracketHolds? J.fusedSource[i] : -1;
		if ((slot >= 0) && (queued[d][slot] <= J.posCap[i]))
			{
			const unsigned long long count = queued[d][slot];
			if ((count == 0) || J.fixupsDone) continue;
			const uint32_t blocks = (uint32_t) std::min<unsigned long long> (2048, (count + PC_THREADS - 1) / PC_THREADS);
			hipLaunchKernelGGL (pc_fixup_kernel, dim3(blocks), dim3(PC_THREADS), 0, st, J.src[i].d_v, f->d_out[i],
			                    J.scratch[d]->pos + J.posOffset[i], J.scratch[d]->posCount + slot, (uint32_t) J.posCap[i],
			                    T, f->tiesAbove, f->one, f->zero);
			}
		else
			{
			all = false;
			const uint32_t blocks = (uint32_t) std::min<size_t> (4096, ((size_t) J.src[i].n + PC_THREADS - 1) / PC_THREADS);
			hipLaunchKernelGGL (pc_binarize_kernel, dim3(blocks), dim3(PC_THREADS), 0, st, J.src[i].d_v, f->d_out[i], J.src[i].n,
			                    T, f->tiesAbove, f->one, f->zero);
			}
		GDSP_LAUNCH_CHECK ();
		}
	*onePass = all;
	return GDSP_OK;
	}

static int pc_entry (const gdsp_select_source* sources, int nsources, uint32_t window, double lo, double hi,
                     const uint32_t* pThousandths, int npercentiles, int strategy, uint32_t sampleTarget,
                     gdsp_reduce_fn reduce, void* reduceCtx, double* values, uint64_t* count,
                     const gdsp_percentile_binarize* fuse, int* onePass)
	{
	std::lock_guard<std::mutex> hold (pcLock);
	int homeDevice = 0;
	GDSP_HIP_TRY (hipGetDevice (&homeDevice));
	struct Home { int device;  ~Home () { (void) hipSetDevice (device); } } home = { homeDevice };   // whatever path returns
	PcJob J;
	J.fuse = fuse;  J.fusedAny = false;  J.vLo = J.vHi = 0.0;
	if (onePass != NULL) *onePass = 0;
	if (fuse != NULL)
		{
		GDSP_REQUIRE ((fuse->which >= 0) && (fuse->which < npercentiles), "the fused binarize names no requested percentile");
		GDSP_REQUIRE (fuse->d_out != NULL, "NULL outputs");
		for (int i=0 ; i<nsources ; i++)
			GDSP_REQUIRE ((sources[i].n == 0) || ((fuse->d_out[i] != NULL) && (fuse->d_out[i] != sources[i].d_v)), "outputs must be distinct from the sources and non-NULL");
		}
	int rc = pc_run (sources, nsources, window, lo, hi, pThousandths, npercentiles, strategy, sampleTarget, reduce, reduceCtx,
	                 values, count, J);
	if ((rc != GDSP_OK) || (fuse == NULL) || (*count == 0)) return rc;
	bool all = false;
	rc = pc_finish_binarize (J, values[fuse->which], &all);
	pcStats[6] = all? 1 : 0;
	if (onePass != NULL) *onePass = all? 1 : 0;
	return rc;
	}

extern "C" {

int gdsp_percentiles (const gdsp_select_source* sources, int nsources, uint32_t window, double lo, double hi,
                      const uint32_t* pThousandths, int npercentiles, int strategy, uint32_t sampleTarget,
                      gdsp_reduce_fn reduce, void* reduceCtx, double* values, uint64_t* count)
	{
	return pc_entry (sources, nsources, window, lo, hi, pThousandths, npercentiles, strategy, sampleTarget, reduce, reduceCtx,
	                 values, count, NULL, NULL);
	}

int gdsp_percentiles_binarize (const gdsp_select_source* sources, int nsources, uint32_t window, double lo, double hi,
                               const uint32_t* pThousandths, int npercentiles, int strategy, uint32_t sampleTarget,
                               gdsp_reduce_fn reduce, void* reduceCtx, double* values, uint64_t* count,
                               const gdsp_percentile_binarize* fuse, int* onePass)
	{
	GDSP_REQUIRE (fuse != NULL, "nothing to fuse");
	return pc_entry (sources, nsources, window, lo, hi, pThousandths, npercentiles, strategy, sampleTarget, reduce, reduceCtx,
	                 values, count, fuse, onePass);
	}

// how the devices' counts are combined from now on (see the head of this file); NULL = on the host
int gdsp_percentiles_use_comm (gdsp_comm* comm)
	{
	std::lock_guard<std::mutex> hold (pcLock);
	pcComm = comm;
	return GDSP_OK;
	}

int gdsp_percentiles_use_device_reduce (gdsp_device_reduce_fn fn, void* ctx)
	{
	std::lock_guard<std::mutex> hold (pcLock);
	pcDReduce = fn;  pcDReduceCtx = ctx;
	return GDSP_OK;
	}

void gdsp_percentiles_stats (uint64_t out[8])
	{
	std::lock_guard<std::mutex> hold (pcLock);
	memcpy (out, pcStats, sizeof(pcStats));
	}

} // extern "C"
