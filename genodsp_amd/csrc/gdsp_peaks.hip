// gdsp_peaks.hip -- `= smooth W=101 = localmax|localmin N` (BASELINE configs[2]) in the reference's arithmetic, off the
// FP64 pipe: a filter at the speed of the block sums, then exact taps for the few bases the filter cannot decide.
//
// What has to come out is the reference's, bit for bit (sum.c:651-663 feeding minmax.c:1183-1227 / :981-1022): a
// surviving base carries its smoothed value -- 101 multiplies and 100 adds, each rounded, in order -- and the test is a
// strict comparison of such values.  Evaluating every base that way keeps the FP64 pipe busy at 0.28 of the HBM rate
// (fir_fixed_extrema_kernel, gdsp_fir.hip).  But the block sums of gdsp_hann_tile.h give every smoothed value to within
// eps = KAPPA * sum|w_k x_k| (KAPPA = 16 W 2^-52: sixteen times the bound the tolerance tests hold the block sums to),
// and an interval [s - eps, s + eps] around each settles almost everything:
//   * a base whose interval lies wholly beyond a neighbour's is beaten whatever the exact values are -> `fill`;
//   * a base of an all-zero window is exactly zero, and so are neighbours that do not beat it -> 0, no arithmetic;
//   * the rest -- the local extrema of a smoothed signal and what ties with them, ~0.5 % of the bases of real-valued
//     coverage -- are QUEUED (their positions, in HBM).
// Per table of vectors, one read-back of a few words:
//   P  probe: the filter on 64 tiles spread over each vector, counting only: the bases that would stay undecided, and
//      the bases of flat stretches (piecewise-constant read depth smooths into runs of exactly equal values: every base
//      a tie, kept, and written its run's value by the CWM form of the filter, below).  The host reads the counts and
//      launches the filter once per form over the vectors that take it; a vector on which more than 3/256 of the bases
//      would stay undecided even so takes the direct kernel instead (every block of the later launches reads the same
//      counters and decides alike).  GDSP_PEAKS_FLAT=0: no CWM form, no read-back (rounds 3-4a).
//   A  filter: block sums -> interval test -> `fill` / 0 written with 16-byte stores, undecided bases queued in the tile's
//      own strip in HBM (16-bit tile-local indices; an LDS atomic per wave); the certain peaks and the neighbourhoods of
//      a tile's few undecided bases are evaluated in place, by a chain of taps in the first waves while the others store.
//      40 KiB of LDS (image + block totals, everything else inside those two) and 128 registers: 4 workgroups per CU.
//      CWM form: a base whose whole window lies in a run of equal inputs is written the run's value -- from a table
//      (peaks_run_table_kernel) when the run's value is a count below PK_RUN_TABLE, by one chain of taps per run otherwise.
//   B  exact: one workgroup per tile, 16 lanes per queued base: its 2h+1 neighbours' windows staged in LDS, one lane per
//      neighbour evaluates tap by tap in the reference's order, the test is repeated on exact values, the base is rewritten.
//   C  the direct fused kernel, gated: its blocks leave at once unless the vector's probe chose it or its queue overflowed.
// Bit-identical to the direct kernel on every input (tests: ties, mirror images, zero stretches, mixed signs, NaN / inf,
// tile seams, whole chromosomes).  A tile holding NaN, an infinity, a magnitude >= 2^1017 or a nonzero magnitude below
// 2^-500 (products that underflow) queues every base.

#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include "gdsp_hann_tile.h"

// The filter's block sums take ALL the window's taps (no direct taps at the ends, HannGeom's E = 0): what it decides on is
// either two values' high words two units apart -- more than a part in 2^21 -- or intervals of KAPPA times the tile's
// largest magnitude, and the cancellation the direct taps are there to avoid in `smooth --smooth=hann` costs a part in
// 2^36 of a value at the worst (all of a window's weight under its smallest tap: (1 - cos w) = 1.9e-3 of the unweighted
// sum, against ~140 roundings of 2^-53 of it).  256 multiply-adds and 46 LDS reads per thread and tile less.
#ifndef PK_E
#define PK_E 0
#endif
#ifndef PK_WAVES
#define PK_WAVES       4                                  // workgroups per CU (= waves per SIMD) the filter is compiled for: 40 KiB of LDS and 128 registers let a fourth in
#endif
#define PK_HMAX        7                                  // neighbourhoods up to 15 bases: one lane of a 16-lane group per neighbour
#define PK_PROBE_TILES 64
#define PK_TILE_CAP    192                                // undecided bases a tile can queue (4.8 % of its 3974: nine times the average on real-valued coverage)
#define PK_WHOLE_TILE  0xFFFFFFFFu                        // a tile's count when every one of its bases is to be evaluated
#define PK_GROUP       16                                 // lanes per queued base in the exact kernel
#define PK_XS          144                                // doubles per group strip: 2h+1 + 100 <= 117 staged inputs; 144 puts the four groups of a wave 32 banks apart

template <int W> struct PeaksTaps { double w[W]; };
// Queues: one strip of PK_TILE_CAP 16-bit tile-local indices per tile of the grid, and the tile's count beside it -- a
// workgroup owns its strip, so queueing costs an LDS atomic per wave and no global one (a per-vector queue with one global
// atomic per wave held the filter at half its speed: 3 M atomics on 24 addresses, from all eight XCDs).

// route: 0 = the probe decides; 1 = the filter whatever the probe counts, 2 = the direct kernel (GDSP_PEAKS_ROUTE: tests)
__global__ void peaks_init_kernel (GdspPeaksCtl* ctl, GdspBatch B, uint32_t stride, int route)
	{
	const uint32_t v = threadIdx.x;
	if (v >= GDSP_BATCH_MAX) return;
	const uint32_t tiles = B.tile0[v+1] - B.tile0[v];
	GdspPeaksCtl c;
	c.count = 0;  c.overflow = 0;  c.probe = 0;  c.flat = 0;
	c.sampled = 4 * ((tiles < PK_PROBE_TILES)? tiles : PK_PROBE_TILES) * stride;     // quarter bases, like the probe's count
	if (route == 1) c.sampled = 0x40000000u;                       // (a probe counts at most 4 x 64 x 3984 quarter bases: gdsp_peaks_takes_direct never says yes)
	if (route == 2) { c.sampled = 0;  c.probe = 0x00800000u; }
	ctl[v] = c;
	}

// One tile of the filter.  PROBE: count the bases that need exact taps into ctl->probe and write nothing.
//
// The smoothed values never leave the registers: a thread holds the 16 of its block, gets the HH either side from its
// neighbours by wave shifts (the two lanes at a wave's ends through a few words of LDS), and the staged INPUTS stay in
// the LDS image -- so a base that is certainly a peak (its interval wholly beyond every neighbour's: the usual case on
// real-valued coverage) gets its exact value right here, one lane per base, tap by tap from that image, in the
// reference's order.  Only bases that tie or nearly tie with a neighbour -- the comparison itself needs exact values --
// go to the tile's strip in HBM for the exact kernel.
#define PK_SURE_CAP 256                                       // certain peaks a tile evaluates in place (one lane each); more go to the strip (CWM: less PK_LEAD_CAP)
#define PK_LEAD_CAP 64                                        // runs of equal inputs whose shared value a tile evaluates (below); with PK_SURE_CAP a workgroup's worth of lanes
// A tile with no more than PK_NEED_INPLACE undecided bases (the usual case on real-valued coverage: a flat top here and
// there) settles them itself as well: one lane per neighbour evaluates its exact value in the same pass as the certain
// peaks (the chain of 101 multiply-adds costs a wave the same whether 40 or 250 of its lanes run it), the comparison of
// minmax.c:1195-1216 follows on those values.  Such a tile is not listed for the exact kernel, which round 3 measured at
// 54 us per 145 Mbp for visiting nearly every tile of a chromosome for one or two bases each.
#define PK_NEED_INPLACE 12
#ifndef PK_AHEAD
#define PK_AHEAD 4                                           // passes of the store loop whose LDS reads are issued together
#endif
#ifndef PK_SHARE_INPLACE
#define PK_SHARE_INPLACE 1
#endif
#define PK_RUN_TABLE 256                                      // CWM: the smoothed value of a run of the count x, x below this, from a table (peaks_run_table_kernel)
// CWM (a vector of piecewise-constant input: read depth): every base of a flat stretch of such a signal ties with its
// neighbours and is KEPT, so it needs its exact value, and the filter as it stands can only queue it -- the probe then
// sends the vector to the direct kernel, 202 operations a base.  But a base whose whole WINDOW is one run of equal inputs
// c has the value T(c) = ((0 + w_0 c) + w_1 c) + ..., a function of c alone: the same bits for every such base, one
// chain of taps per run.  So: a change bit per staged element (it differs from the one before); a base is `cw` when no
// change falls inside its window; neighbours within HH of a cw base that are cw themselves lie in the same run (their
// windows overlap in >= 94 elements) and tie exactly -- only the others are compared, on the high words as always; a cw
// base they do not beat, with margin, is written T of its run, which the run's first cw base of the tile (its leader)
// gets from the same chain of taps as the certain peaks; without margin it is queued like any undecided base.
// inclusive sum / maximum over the 64 lanes by DPP moves (a lane that has no source, or is masked off, receives 0 -- the
// identity of both for values >= 0): no LDS round trips, where six __shfl_up steps cost six (ds_bpermute) -- the three
// scans behind the interval test were ~2000 cycles of a workgroup's ~27 000 by round 5's stamps
template <int CTRL, int ROWS> __device__ __forceinline__ int pk_dpp0 (int x) { return __builtin_amdgcn_update_dpp (0, x, CTRL, ROWS, 0xF, true); }
__device__ __forceinline__ int pk_wave_sum_scan (int x)
	{
	x += pk_dpp0<0x111, 0xF> (x);  x += pk_dpp0<0x112, 0xF> (x);  x += pk_dpp0<0x114, 0xF> (x);  x += pk_dpp0<0x118, 0xF> (x);
	x += pk_dpp0<0x142, 0xA> (x);  x += pk_dpp0<0x143, 0xC> (x);               // row 0 -> 1 and 2 -> 3, then rows 0..1 -> 2..3
	return x;
	}
__device__ __forceinline__ int pk_wave_max_scan (int x)                        // x >= 0
	{
	x = max (x, pk_dpp0<0x111, 0xF> (x));  x = max (x, pk_dpp0<0x112, 0xF> (x));  x = max (x, pk_dpp0<0x114, 0xF> (x));
	x = max (x, pk_dpp0<0x118, 0xF> (x));  x = max (x, pk_dpp0<0x142, 0xA> (x));  x = max (x, pk_dpp0<0x143, 0xC> (x));
	return x;
	}

template <int W, bool FMA, bool MAX, int HH, bool PROBE, bool CWM>
__device__ __forceinline__ void peaks_filter_tile (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t tile,
                                                   const HannConsts<W, PK_E>& K, const double* __restrict__ d_taps, double fill, GdspPeaksCtl* ctl,
                                                   uint16_t* __restrict__ strip, uint32_t* __restrict__ tileCount, uint32_t cap,
                                                   uint32_t* __restrict__ tileList, uint32_t gt, const double* __restrict__ runTable)
	{
	typedef HannGeom<W, PK_E> G;
	constexpr double KAPPA = 16.0 * W * 2.220446049250313e-16;
	constexpr int    h = HH, sh = HH & 1;                          // sh keeps the first staged index even
	constexpr int    stride = G::OUT - 2*h - 2*sh;                 // outputs kept per tile (even)
	constexpr int    NW = HN_THREADS / 64;
	// LDS: the staged image and the block totals, 40 KiB to the byte -- a quarter of a CU's, so that four workgroups fit
	// (round 5; 44.7 KiB and three until then).  Everything else lives inside those two:
	//   * what is alive while the block totals are (the waves' verdicts and statistics, the counters, the edge words
	//     exchanged between waves, CWM's change bits) in the image's PAD slots -- the 17th double of each block of 16, which
	//     nothing stages, reads or writes (gdsp_hann_tile.h: hann_pad_word).  Pad p: low word = slot p of the map below,
	//     high word = chgBits[p] (CWM); the edge values of the double-precision path take whole pads (they come after the
	//     change bits' last use)
	//   * what is written only after the barrier that follows the block sums' last read of the totals (B1 below) ON the
	//     totals: the blocks' code words, the lists, the exact values, CWM's run values
	__shared__ __attribute__((aligned(16))) double lds[HN_THREADS * HN_PITCH];
	__shared__ __attribute__((aligned(16))) double tot[3][HN_THREADS];
	constexpr int PAD_HUGE = 0, PAD_STATS = HN_PAD_STATS, PAD_NSURE = 8, PAD_QUEUED = 9, PAD_NLEAD = 10, PAD_NFLAT = 11, PAD_WLEAD = 12,
	              PAD_ELO = 16, PAD_EHI = 16 + NW * PK_HMAX, PAD_NCHAIN = PAD_EHI + NW * PK_HMAX;
	static_assert (PAD_NCHAIN < 210, "pad map");     // (PK_STAMPS keeps its clock in pads 210 and 211)
	(void) PAD_HUGE;
	auto padW = [&] (int i) -> uint32_t& { return *hann_pad_word (lds, i); };
	auto padD = [&] (int i) -> double&   { return lds[i * HN_PITCH + HN_G]; };
	auto chgBitsAt = [&] (int i) -> uint32_t& { return hann_pad_word (lds, i)[1]; };
	constexpr int SURE_CAP = CWM? PK_SURE_CAP - PK_LEAD_CAP : PK_SURE_CAP;
	char* const tb = reinterpret_cast<char*> (&tot[0][0]);
	constexpr int OFF_CODE = 0, OFF_SUREL = OFF_CODE + 4 * HN_THREADS,
	              OFF_NEEDL = OFF_SUREL + 2 * PK_SURE_CAP, OFF_EXACT = (OFF_NEEDL + 2 * PK_NEED_INPLACE + 7) & ~7,
	              OFF_LEADL = OFF_EXACT + 8 * PK_NEED_INPLACE * (2*HH + 1),
	              OFF_RUNT = (OFF_LEADL + 2 * PK_LEAD_CAP + 7) & ~7,
	              OFF_LEADOF = OFF_RUNT + (CWM? 8 * PK_LEAD_CAP : 0), OFF_BLKRUN = OFF_LEADOF + (CWM? HN_THREADS : 0),
	              OFF_END = OFF_BLKRUN + (CWM? HN_THREADS : 0);
	static_assert (OFF_END <= (int) sizeof(tot), "the lists do not fit on the block totals");
	// what the store loop writes on a base, two bits each, a word per block of 16 outputs: 0 `fill`, 1 an exact zero, 2 (CWM)
	// its run's value, 3 nothing -- the base is written by the lane that evaluates it (until round 5 a bit map each: three
	// LDS reads and their shifts per pair of outputs in a loop that is all instructions)
	uint32_t* const codeBits = reinterpret_cast<uint32_t*> (tb + OFF_CODE);
	uint16_t* const sureList = reinterpret_cast<uint16_t*> (tb + OFF_SUREL);
	uint16_t* const needList = reinterpret_cast<uint16_t*> (tb + OFF_NEEDL);   // the first undecided bases: settled in place when there are no more than these
	double*   const exactVal = reinterpret_cast<double*>   (tb + OFF_EXACT);   // ... from the exact values of their neighbourhoods
	uint16_t* const leadList = reinterpret_cast<uint16_t*> (tb + OFF_LEADL);
	// CWM: a run's value per LEADER (the first base of the run in the tile whose window lies inside it; at most PK_LEAD_CAP a
	// tile), the flat bases as a bit map like the zeros', and per block of 16 outputs the number of its leader / of the
	// leader of the run its flat bases belong to (a block meets one run: two runs' such bases lie a window apart)
	double*   const runT     = reinterpret_cast<double*>   (tb + OFF_RUNT);
	uint8_t*  const leaderOf = reinterpret_cast<uint8_t*>  (tb + OFF_LEADOF);
	uint8_t*  const blockRun = reinterpret_cast<uint8_t*>  (tb + OFF_BLKRUN);
	uint32_t& nsure = padW (PAD_NSURE), &queued = padW (PAD_QUEUED), &nlead = padW (PAD_NLEAD), &nflat = padW (PAD_NFLAT), &nchain = padW (PAD_NCHAIN);

	const int      p         = threadIdx.x, lane = p & 63, wave = p >> 6;
	const int64_t  keepStart = (int64_t) tile * stride;            // first output this tile stores
	const int64_t  compStart = keepStart - h - sh;                 // first smoothed value it computes (even)
	const int64_t  e0        = compStart - G::LEAD;
	const bool     live      = (p >= G::HALO_L) && (p < HN_THREADS - G::HALO_R);
	const int      blk       = p - G::HALO_L;                      // the block of smoothed values a live thread holds
	const double   never     = MAX? -DBL_MAX : DBL_MAX;            // what a position outside the vector holds: it beats nothing
	const int      validLo   = (compStart < 0)? (int) -compStart : 0;
	const int      validHi   = (compStart + G::OUT <= (int64_t) n)? G::OUT : (int) ((int64_t) n - compStart);
	const int      keepLo    = h + sh;                             // smoothed values [keepLo, keepHi) are this tile's outputs
	const int      keepHi    = (keepLo + stride < validHi)? keepLo + stride : validHi;

	if (p == 0) { nsure = 0;  queued = 0;  nlead = 0;  nflat = 0;  nchain = 0; }   // (the barriers of the block sums come before their first use; staging leaves the pads alone)
	double acc[HN_G];
	if (!PROBE) PK_STAMP0 (lds);
	bool direct = hann_tile_sums<W, false, PK_E, true, true, true, CWM> (lds, tot, NULL, in, n, e0, K, acc);     // (RAW: acc = S - C, the scale applied where magnitudes matter)
	if (!PROBE) PK_STAMP (lds, 3);                                 // phase 2 (the middle stretch) done in thread 0

	// ---- what the tile's inputs are like (is a sign bit set, is there a nonzero magnitude below 2^-500) was found while
	// they were staged, on the loading registers (hann_tile_sums, SSTATS: pad word PAD_STATS + wave; until round 5 a look at
	// the own block in the LDS image here -- sixteen more LDS reads and a hundred integer instructions a thread).  CWM: the
	// own block's change bits come from phase 1 of the block sums (CHG), in the pads' high words.
	// the high words of the own block's smoothed values, as 32-bit integers (see the test below), and the words the
	// neighbouring threads need: exchanged now, so that one barrier serves the statistics and the edges
	const uint32_t away = MAX? 0u : 0x7FEFFFFFu;                   // outside the vector: beats nothing
	uint32_t k[HN_G + 2*HH];
#pragma unroll
	for (int u=0 ; u<HN_G ; u++)
		{
		const int  c      = blk * HN_G + u;
		const bool inside = live && (c >= validLo) && (c < validHi);
		k[HH + u] = inside? (uint32_t) (__double_as_longlong (acc[u]) >> 32) & 0x7FFFFFFFu : away;          // (-0.0 is a zero)
		}
	if (lane == 0)  { for (int t=0 ; t<HH ; t++) padW (PAD_ELO + wave*HH + t) = k[HH + t]; }
	if (lane == 63) { for (int t=0 ; t<HH ; t++) padW (PAD_EHI + wave*HH + t) = k[HN_G + t]; }
#pragma unroll
	for (int t=0 ; t<HH ; t++)
		{
		k[t]             = (uint32_t) pk_dpp0<0x138, 0xF> ((int) k[HN_G + t]);         // wave_shr:1: the previous block's last HH words
		k[HN_G + HH + t] = (uint32_t) pk_dpp0<0x130, 0xF> ((int) k[HH + t]);           // wave_shl:1: the next block's first HH (the wave's end lanes: below)
		}
	__syncthreads ();                                              // B1: nobody reads the block totals any more
	if (!PROBE) PK_STAMP (lds, 4);                                 // statistics, high words, edges, barrier B1
#pragma unroll
	for (int t=0 ; t<HH ; t++)
		{
		if (lane == 0)  k[t]             = (wave == 0)?    away : padW (PAD_EHI + (wave-1)*HH + t);
		if (lane == 63) k[HN_G + HH + t] = (wave == NW-1)? away : padW (PAD_ELO + (wave+1)*HH + t);
		}
	uint32_t flagsAll = 0;
#pragma unroll
	for (int w=0 ; w<NW ; w++) flagsAll |= padW (PAD_STATS + w);
	if (flagsAll & 2u) direct = true;
	const bool nonneg = ((flagsAll & 1u) == 0);

	uint32_t isNeed = 0, isZero = 0, isSure = 0, cwLead = 0, cwFlat = 0;      // (cw*: CWM, a block's leaders and the bases written their run's value)
	if (nonneg && !direct)
		{
		// ---- no negative input: every smoothed value is >= 0, and doubles >= 0 order like their bit patterns.  The test
		// runs on the HIGH WORDS alone, as 32-bit integers (full-rate instructions; v_max_f64 / v_cmp_f64 are not): a value
		// whose high word is at least 2 above another's exceeds it by more than one part in 2^21, far beyond the two
		// intervals' reach (KAPPA ~ 4e-13), so "certainly beaten" and "certainly the extreme" are decided exactly as by
		// the intervals, only more cautiously -- what falls within two units of the neighbours' extreme (flat tops, ties)
		// is left to the exact kernel.  A high word of 0 is an exact zero (no magnitude below 2^-500 in such a tile).
		// A thread holds the 16 words of its block and has got HH either side from its neighbours by wave shifts (the two
		// lanes at a wave's ends through a few words of LDS).  Outside the vector: 0 (MAX) / DBL_MAX's (MIN): beats nothing.
		if (live)
			{
			// the extreme of the HH words before a base and of the HH after it, for all 16 bases at once: runs of 2 and 4 by doubling
			constexpr int NV = HN_G + 2*HH;
			auto runs_of = [&] (const uint32_t (&kk)[NV], uint32_t (&run)[HN_G + HH + 1])
				{
				uint32_t m2[NV], m4[NV];
#pragma unroll
				for (int i=0 ; i+1<NV ; i++) m2[i] = MAX? max (kk[i], kk[i+1]) : min (kk[i], kk[i+1]);
#pragma unroll
				for (int i=0 ; i+3<NV ; i++) m4[i] = MAX? max (m2[i], m2[i+2]) : min (m2[i], m2[i+2]);
#pragma unroll
				for (int i=0 ; i<HN_G+HH+1 ; i++)
					{
					uint32_t r;
					if      (HH == 1) r = kk[i];
					else if (HH == 2) r = m2[i];
					else if (HH == 3) r = MAX? max (m2[i], kk[i+2]) : min (m2[i], kk[i+2]);
					else if (HH == 4) r = m4[i];
					else if (HH == 5) r = MAX? max (m4[i], kk[i+4]) : min (m4[i], kk[i+4]);
					else if (HH == 6) r = MAX? max (m4[i], m2[i+4]) : min (m4[i], m2[i+4]);
					else              r = MAX? max (m4[i], m4[i+3]) : min (m4[i], m4[i+3]);
					run[i] = r;
					}
				};
			uint32_t run[HN_G + HH + 1];
			runs_of (k, run);

			// CWM: which of the block's bases and of the HH either side have a window without a change (bit i <-> base u = i - HH).
			// The window of base u of block p is the staged elements 16p+u-100 .. 16p+u: a change at element q rules out
			// u = q-16p .. q-16p+99.  The change bits of blocks p-7 .. p+1 as one string of 144 bits from element 16(p-7).
			uint32_t cwAll = 0, run2[HN_G + HH + 1];
			if (CWM)
				{
				// In 32-bit words: a (0..31), b (32..63), c (64..95), d (96..127), e (128..143) of the string.  Base u's window is the
				// string's positions u+13 .. u+112, so it is flat iff the last change at or before u+112 lies 100 or more back: the
				// last change below position Q0 = 112-HH, and the NV positions from Q0 as the word m (bit i of m <-> position Q0+i).
				const uint32_t a = chgBitsAt (p - 7) | (chgBitsAt (p - 6) << 16), b = chgBitsAt (p - 5) | (chgBitsAt (p - 4) << 16);
				const uint32_t c = chgBitsAt (p - 3) | (chgBitsAt (p - 2) << 16), d = chgBitsAt (p - 1) | (chgBitsAt (p)     << 16);
				const uint32_t e = (p + 1 < HN_THREADS)? chgBitsAt (p + 1) : 0xFFFFu;
				// (the probe only counts: where changes are dense it does not look further.  The filter must: a base's leader is
				//  found through what its neighbours' threads conclude from the same bits, so every thread concludes exactly)
				if (!PROBE || (__popc (a) + __popc (b) + __popc (c) + __popc (d) + __popc (e) <= 8))
					{
					constexpr int Q0 = 112 - HH;                                // the string's position that ends the window of bit 0
					static_assert ((Q0 > 96) && (Q0 + NV <= 144) && (NV <= 32), "the walk starts inside word d and ends inside e");
					const uint32_t d0 = d & ((1u << (Q0 - 96)) - 1u);
					int last = (d0 != 0)? (127 - __clz (d0)) : (c != 0)? (95 - __clz (c)) : (b != 0)? (63 - __clz (b)) : (a != 0)? (31 - __clz (a)) : -1000;
					const uint32_t m = (d >> (Q0 - 96)) | ((e & ((1u << HH) - 1u)) << (128 - Q0));
					// bit i's window is the positions Q0+i-99 .. Q0+i: nothing from Q0 up to it (i below m's lowest set bit), and the
					// last change below Q0 a hundred or more back (i >= last - Q0 + 100)
					const int from = last - Q0 + 100;
					cwAll = (~m & (m - 1u)) & ((from <= 0)? 0xFFFFFFFFu : (from >= 32)? 0u : (0xFFFFFFFFu << from));
					// (bases outside the vector are nobody's business)
					const int first = blk * HN_G - HH;                          // base of bit 0
					const int lo = max (validLo - first, 0), hi = min (validHi - first, NV);
					cwAll &= (lo < hi)? ((((hi - lo) >= 32)? 0xFFFFFFFFu : ((1u << (hi - lo)) - 1u)) << lo) : 0u;
					}
				uint32_t k2[NV];
#pragma unroll
				for (int i=0 ; i<NV ; i++)
					{
					const uint32_t all = (uint32_t) (((int32_t) (cwAll << (31 - i))) >> 31);      // bit i over the whole word
					k2[i] = MAX? (k[i] & ~all) : (k[i] | (all & away));                           // (away: 0 / the largest finite word, above which no word here lies)
					}
				runs_of (k2, run2);
				}
			uint32_t isFlat = 0;
#pragma unroll
			for (int u=0 ; u<HN_G ; u++)
				{
				const int      c   = blk * HN_G + u;
				const bool     cw  = CWM && (((cwAll >> (u + HH)) & 1u) != 0);
				const uint32_t ext = !cw? (MAX? max (run[u], run[u + HH + 1])  : min (run[u], run[u + HH + 1]))
				                        : (MAX? max (run2[u], run2[u + HH + 1]) : min (run2[u], run2[u + HH + 1]));     // cw: of the neighbours outside its run only
				const uint32_t x   = k[u + HH];
				if ((c < keepLo) || (c >= keepHi)) continue;
				if (MAX)
					{
					if (ext > x + 1) continue;                     // certainly beaten: `fill`
					if (x == 0)          isZero |= 1u << u;        // (then ext <= 1: among zeros a tie; a neighbour of word 1 is < 2^-1000: none here)
					else if (x > ext + 1) { if (cw) isFlat |= 1u << u;  else isSure |= 1u << u; }
					else                  isNeed |= 1u << u;
					}
				else
					{
					if (x == 0) { isZero |= 1u << u;  continue; }  // nothing is below zero here: kept
					if (ext + 1 < x) continue;                     // certainly beaten
					if (x + 1 < ext) { if (cw) isFlat |= 1u << u;  else isSure |= 1u << u; }
					else             isNeed |= 1u << u;
					}
				}
			if (CWM)
				{
				// leaders: the first cw base of a run inside this tile's computed stretch (cw, and the base before it is not --
				// or is another tile's); only runs with a base to write need one, but a run has some forty blocks and one leader
				const uint32_t cw16 = (cwAll >> HH) & 0xFFFFu, prev16 = (cwAll >> (HH - 1)) & 0xFFFFu;
				uint32_t lead16 = cw16 & ~prev16;
				if (blk == 0) lead16 |= cw16 & 1u;
				cwLead = lead16;  cwFlat = isFlat;
				}
			}
		}
	else
	{
	// ---- negative inputs in the tile (or a tile to be evaluated whole): the intervals themselves, in double precision.
	// sum|w x| is bounded by the tile's largest magnitude (the taps add up to 1): found now, from the staged inputs
	double amax = 0.0;
		{
		const double* xb = lds + p * HN_PITCH;
#pragma unroll
		for (int u=0 ; u<HN_G ; u++) amax = fmax (amax, fabs (xb[u]));
		for (int off=32 ; off>0 ; off>>=1) amax = fmax (amax, __shfl_xor (amax, off, 64));
		__syncthreads ();                                          // (the edge words have been read)
		if (lane == 0) padD (PAD_ELO + wave*HH) = amax;
		__syncthreads ();
		amax = 0.0;
		for (int w=0 ; w<NW ; w++) amax = fmax (amax, padD (PAD_ELO + w*HH));
		__syncthreads ();
		}
	const double epsAbs = KAPPA * amax;
	// `a certainly beats b`: a's interval wholly beyond b's
	const double eps2 = 2.0 * epsAbs * (1.0 + KAPPA);
	auto beats = [&] (double a, double b) { return MAX? (a - eps2 > b) : (a + eps2 < b); };
	// ---- the block and HH values either side of it, in registers
	double v[HN_G + 2*HH];
#pragma unroll
	for (int u=0 ; u<HN_G ; u++)
		{
		const int  c      = blk * HN_G + u;
		const bool inside = live && (c >= validLo) && (c < validHi);
		v[HH + u] = !inside? never : (acc[u] == 0.0)? 0.0 : K.scale * acc[u];     // (the intervals are absolute: the window's normalisation belongs in)
		}
	if (lane == 0)  { for (int t=0 ; t<HH ; t++) padD (PAD_ELO + wave*HH + t) = v[HH + t]; }
	if (lane == 63) { for (int t=0 ; t<HH ; t++) padD (PAD_EHI + wave*HH + t) = v[HN_G + t]; }
#pragma unroll
	for (int t=0 ; t<HH ; t++)
		{
		v[t]             = __shfl_up   (v[HN_G + t], 1, 64);       // the previous block's last HH values
		v[HN_G + HH + t] = __shfl_down (v[HH + t],   1, 64);       // the next block's first HH
		}
	__syncthreads ();
#pragma unroll
	for (int t=0 ; t<HH ; t++)
		{
		if (lane == 0)  v[t]             = (wave == 0)?    never : padD (PAD_EHI + (wave-1)*HH + t);
		if (lane == 63) v[HN_G + HH + t] = (wave == NW-1)? never : padD (PAD_ELO + (wave+1)*HH + t);
		}

	// ---- the interval test
	if (live)
		{
		if (direct)
			{
#pragma unroll
			for (int u=0 ; u<HN_G ; u++) { const int c = blk * HN_G + u;  if ((c >= keepLo) && (c < keepHi)) isNeed |= 1u << u; }
			}
		else
			{
			// the extreme of the HH values before a base and of the HH after it (the base itself left out), for all 16 bases
			// at once: extremes of runs of 2, 4, ... values by doubling, a run of HH put together from those
			double run[HN_G + HH + 1];                             // run[i] = extreme of v[i .. i+HH-1]
				{
				constexpr int NV = HN_G + 2*HH;
				double m2[NV], m4[NV];
#pragma unroll
				for (int i=0 ; i+1<NV ; i++) m2[i] = MAX? fmax (v[i], v[i+1]) : fmin (v[i], v[i+1]);
#pragma unroll
				for (int i=0 ; i+3<NV ; i++) m4[i] = MAX? fmax (m2[i], m2[i+2]) : fmin (m2[i], m2[i+2]);
#pragma unroll
				for (int i=0 ; i<HN_G+HH+1 ; i++)
					{
					double r;
					if      (HH == 1) r = v[i];
					else if (HH == 2) r = m2[i];
					else if (HH == 3) r = MAX? fmax (m2[i], v[i+2])    : fmin (m2[i], v[i+2]);
					else if (HH == 4) r = m4[i];
					else if (HH == 5) r = MAX? fmax (m4[i], v[i+4])    : fmin (m4[i], v[i+4]);
					else if (HH == 6) r = MAX? fmax (m4[i], m2[i+4])   : fmin (m4[i], m2[i+4]);
					else              r = MAX? fmax (m4[i], m4[i+3])   : fmin (m4[i], m4[i+3]);      // HH == 7: two runs of 4 overlapping by one
					run[i] = r;
					}
				}
#pragma unroll
			for (int u=0 ; u<HN_G ; u++)
				{
				const int    c   = blk * HN_G + u;
				const double ext = MAX? fmax (run[u], run[u + HH + 1]) : fmin (run[u], run[u + HH + 1]);
				const double x   = v[u + HH];
				if ((c < keepLo) || (c >= keepHi) || beats (ext, x)) continue;      // not this tile's, or certainly beaten: `fill`
				if (beats (x, ext)) isSure |= 1u << u;                 // certainly the extreme of its neighbourhood: only its exact value is missing
				else                isNeed |= 1u << u;                 // ties, near-ties: the comparison needs exact values
				}
			}
		}

	}

	if (PROBE)
		{
		int cnt = 4 * __popc (isNeed) + __popc (isSure) + __popc (cwLead);    // in quarter bases: a certain peak or a run's leader costs a lane, an undecided base sixteen
		int flat = __popc (cwFlat);
		for (int off=32 ; off>0 ; off>>=1) { cnt += __shfl_xor (cnt, off, 64);  flat += __shfl_xor (flat, off, 64); }
		if ((lane == 0) && (cnt != 0)) atomicAdd (&ctl->probe, (uint32_t) cnt);   // (ctl->sampled is in quarter bases too)
		if (CWM && (lane == 0) && (flat != 0)) atomicAdd (&ctl->flat, (uint32_t) flat);
		return;
		}

	if (!PROBE) PK_STAMP (lds, 8);
	// ---- CWM: the runs' leaders (a list like the certain peaks'; a block has at most one: its cw bases are one stretch), the
	//      last leader's block at or before every block.  Before the other lists: a run's value from the table is a load from
	//      memory, on its way while they are made
	int    leadIncl = -1, runSlot = -1;
	double runVal = 0.0;
	if (CWM && !direct)
		{
		const int cnt = (cwLead != 0)? 1 : 0;
		const int incl = pk_wave_sum_scan (cnt);
		const int waveTotal = __builtin_amdgcn_readlane (incl, 63);
		if (waveTotal != 0)
			{
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd (&nlead, (uint32_t) waveTotal);
			base = (uint32_t) __shfl ((int) base, 0, 64);
			const uint32_t at = base + (uint32_t) (incl - cnt);
			if ((cnt != 0) && (at < PK_LEAD_CAP))
				{
				// the run's value: for a count below PK_RUN_TABLE (read depth is a count) the chain of taps has been walked
				// once (peaks_run_table_kernel: the same operations in the same order on the same operands); any other
				// value's run goes on the list for this tile's own chains
				const int    u = __ffs ((int) cwLead) - 1;
				const int    e = G::LO + blk * HN_G + u;                  // the window's first staged element: like all its others
				const double X = lds[e + (e >> 4)];
				leaderOf[blk] = (uint8_t) at;
				if ((X >= 0.0) && (X < (double) PK_RUN_TABLE) && ((double) (int) X == X)) { runVal = runTable[(int) X];  runSlot = (int) at; }
				else leadList[atomicAdd (&nchain, 1u)] = (uint16_t) (blk * HN_G + u);
				}
			}
		leadIncl = (cwLead != 0)? blk : -1;
		leadIncl = pk_wave_max_scan (leadIncl + 1) - 1;                  // (-1: no leader at or before this block in the wave)
		if (lane == 63) padW (PAD_WLEAD + wave) = (uint32_t) leadIncl;
		if ((__builtin_amdgcn_ballot_w64 (cwFlat != 0) != 0) && (lane == 0)) nflat = 1;     // (a flag: is there a flat base in the tile)
		}
	if (!PROBE) PK_STAMP (lds, 9);
	// ---- certain peaks: a list in LDS (what does not fit joins the undecided)
	if (!direct)
		{
		const int cnt = __popc (isSure);
		const int incl = pk_wave_sum_scan (cnt);
		const int waveTotal = __builtin_amdgcn_readlane (incl, 63);
		if (waveTotal != 0)                                        // (uniform over the wave)
			{
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd (&nsure, (uint32_t) waveTotal);
			base = (uint32_t) __shfl ((int) base, 0, 64);
			uint32_t at = base + (uint32_t) (incl - cnt);
			uint32_t word = isSure;
			while (word != 0)
				{
				const int u = __ffs ((int) word) - 1;
				word &= word - 1;
				if (at < SURE_CAP) sureList[at] = (uint16_t) (blk * HN_G + u);
				else                  { isSure &= ~(1u << u);  isNeed |= 1u << u; }
				at++;
				}
			}
		}
	// ---- undecided bases: tile-local indices into the tile's strip (a tile that must be evaluated whole queues nothing)
	if (!direct)
		{
		const int cnt = __popc (isNeed);
		const int incl = pk_wave_sum_scan (cnt);
		const int waveTotal = __builtin_amdgcn_readlane (incl, 63);
		if (waveTotal != 0)
			{
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd (&queued, (uint32_t) waveTotal);
			base = (uint32_t) __shfl ((int) base, 0, 64);
			uint32_t at = base + (uint32_t) (incl - cnt);
			uint32_t word = isNeed;
			while (word != 0)
				{
				const int u = __ffs ((int) word) - 1;
				word &= word - 1;
				if (at < cap) strip[at] = (uint16_t) (blk * HN_G + u);
				if (at < PK_NEED_INPLACE) needList[at] = (uint16_t) (blk * HN_G + u);
				at++;
				}
			}
		}
	if (!PROBE) PK_STAMP (lds, 10);
	if (CWM && (runSlot >= 0)) runT[runSlot] = runVal;
	if (live)                                                      // every block of 16 has its own half word: plain stores
		{
		auto spread = [] (uint32_t x)                                // bit i of 16 -> bit 2i
			{
			x = (x | (x << 8)) & 0x00FF00FFu;  x = (x | (x << 4)) & 0x0F0F0F0Fu;  x = (x | (x << 2)) & 0x33333333u;  x = (x | (x << 1)) & 0x55555555u;
			return x;
			};
		// (a tile the exact kernel takes whole is written zeros and `fill` here; zero, certain and flat exclude each other)
		codeBits[blk] = direct? spread (isZero) : (spread (isZero | isSure) | (spread (cwFlat | isSure) << 1));
		}
	__syncthreads ();
	if (!PROBE) PK_STAMP (lds, 5);                                 // classification, lists, their barrier
	const int  sure    = (int) ((nsure < SURE_CAP)? nsure : SURE_CAP);
	const int  nq      = (int) queued;
	const int  nl      = CWM? (int) nchain : 0;                  // (the runs whose value this tile's chains find)
	if (CWM && (nlead > PK_LEAD_CAP)) direct = true;                  // (more runs than a tile of 4096 can hold a hundred bases apart: never; the exact kernel takes the tile whole)
	// (cap < PK_NEED_INPLACE only under the tests' GDSP_PEAKS_QUEUE_CAP: they want the strips and their overflow exercised)
	const bool inPlace = !direct && (nq != 0) && (nq <= PK_NEED_INPLACE) && (sure + nl + nq * (2*HH + 1) <= HN_THREADS) && (cap >= PK_NEED_INPLACE);
	if (p == 0)
		{
		const uint32_t q = direct? PK_WHOLE_TILE : inPlace? 0u : queued;
		*tileCount = q;
		if (q != 0) tileList[atomicAdd (&ctl->count, 1u)] = gt;    // (few tiles have anything for the exact kernel)
		if (!direct && (queued > cap)) ctl->overflow = 1;          // more undecided bases than a strip holds: the vector goes through the direct kernel
		}

	// ---- CWM: which run a block's flat bases belong to: the last leader at or before the block (this wave's by the scan
	//      above, the waves' before it by their last words).  Every pair that holds a flat base is written whole -- 16 bytes a
	//      lane like the rest of the tile (until round 5 a lane wrote each flat base of its block by itself: sixteen 8-byte
	//      stores a lane to sixty-four lines a wave, 1.5 x the tile's bytes at the memory).  With every run's value out of the
	//      table (read depth) the flat bases go out with the rest of the tile, beside the certain peaks' chains; a tile that
	//      has a run's value to find writes them behind those chains
	auto resolve_runs = [&] ()
		{
		if (CWM && (cwFlat != 0))
			{
			int lb = leadIncl;
			for (int w=0 ; w<wave ; w++) lb = max (lb, (int) padW (PAD_WLEAD + w));
			blockRun[blk] = leaderOf[(lb >= 0)? lb : 0];             // (lb >= 0: a flat base's run has a leader at or before it)
			}
		};
	const bool flatNow = CWM && !direct && (nflat != 0) && (nl == 0), flatLater = CWM && !direct && (nflat != 0) && (nl != 0);    // (uniform)
	// (the undecided bases settled in place are left to their lanes by the store loop: marked now, so that the waves without
	//  a chain of taps to walk can store while the others walk theirs)
	if (inPlace && (p < nq)) atomicOr (&codeBits[needList[p] >> 4], 3u << (2 * (needList[p] & 15)));
	if (flatNow) resolve_runs ();
	if (flatNow || inPlace) __syncthreads ();                      // (uniform)

	// ---- exact values of the certain peaks -- and of the neighbourhoods of the undecided bases settled in place -- one
	//      lane each, from the staged inputs: tap by tap in the reference's order (sum.c:655-663); a certain peak is written
	//      by the lane that evaluated it
	const int nbrs  = inPlace? nq * (2*HH + 1) : 0;
	const int evals = direct? 0 : sure + nbrs + nl;                // (a tile taken whole by the exact kernel evaluates nothing here)
	if (p < evals)
		{
		const bool lead = CWM && (p >= sure + nbrs);
		const int  mine = (p < sure)? 0 : (p - sure) / (2*HH + 1);
		const int  c = (p < sure)? (int) sureList[p] : lead? (int) leadList[p - sure - nbrs]
		                         : (int) needList[mine] - HH + ((p - sure) - mine * (2*HH + 1));
		const bool inside = (c >= validLo) && (c < validHi);       // (a neighbour outside the vector beats nothing; c is within the computed stretch either way)
		const int e = G::LO + (inside? c : validLo);               // the window's first staged element
		// element e+k sits at e+k + ((e+k) >> 4) in the image: with e = 16 q + r that is 17 q + r + k + ((r + k) >> 4), and
		// (r + k) >> 4 = (k >> 4) + ((k & 15) >= 16 - r).  Sixteen addresses per lane, one for each k & 15, leave every
		// tap's read with a compile-time offset.
		const double* at16[16];
#pragma unroll
		for (int j=0 ; j<16 ; j++) at16[j] = lds + (e + (e >> 4)) + ((j >= 16 - (e & 15))? 1 : 0);
		// (in batches of reads, then products, then sums: left to itself the compiler fetches two to four taps ahead and the
		//  chain -- wave 0's, while the workgroup's LDS waits for it -- sits out an LDS round trip every few taps: 55 cycles
		//  a tap by round 5's stamps, 41 in batches.  The window's taps as SCALAR operands, fetched a batch at a time from
		//  memory where the chain wants them: as kernel arguments their 202 words crowded out the block sums' constants, and
		//  as a second LDS read per tap -- a copy of the window in every tile's LDS -- they cost the chain 2 % of the kernel)
		double a = 0.0;
		constexpr int KB = 16, NB = (W + KB - 1) / KB;               // (a 64-byte scalar load is 8 taps)
#pragma unroll
		for (int b=0 ; b<NB ; b++)
			{
			double xv[KB], wv[KB];                                   // (two batches in flight -- the next one's reads behind this one's sums -- spill 130 registers)
#pragma unroll
			for (int j=0 ; j<KB ; j++)
				{
				const int k = b * KB + j;
				if (k < W) { xv[j] = at16[k & 15][k + (k >> 4)];  wv[j] = d_taps[k]; }
				}
			__builtin_amdgcn_sched_barrier (0);                      // every read of the batch before its first product
			if (FMA)
				{
#pragma unroll
				for (int j=0 ; j<KB ; j++) { if (b * KB + j < W) a = __builtin_fma (wv[j], xv[j], a); }
				}
			else
				{
#pragma unroll
				for (int j=0 ; j<KB ; j++) { if (b * KB + j < W) xv[j] = wv[j] * xv[j]; }      // the products, then the sums in order
				__builtin_amdgcn_sched_barrier (0);
#pragma unroll
				for (int j=0 ; j<KB ; j++) { if (b * KB + j < W) a = a + xv[j]; }
				}
			__builtin_amdgcn_sched_barrier (0);
			}
		if (p < sure)  out[compStart + c] = a;
		else if (lead) runT[CWM? leaderOf[c >> 4] : 0] = a;        // the value of every base of the run whose window lies inside it
		else           exactVal[p - sure] = inside? a : (MAX? -INFINITY : INFINITY);
		}
	if (!PROBE) PK_STAMP (lds, 6);                                 // thread 0's exact values (wave 0's chain of taps)
	// ---- the tile's other outputs: zero where an exact zero stands, `fill` elsewhere (a queued base is rewritten by the
	//      exact kernel), CWM: its run's value on a flat base; pairs, 16 bytes per lane; a pair that holds a certain peak
	//      leaves that element to its lane
	// (a wave that has evaluated exact values has done its share: the others divide the tile between them, and the
	//  workgroup is done when the slower of the two jobs is, not after one behind the other -- round 5's stamps: the chain
	//  of taps and the store loop were 17 % and 15 % of a workgroup's life, one behind the other in wave 0.  Giving wave 0
	//  a seventh of the stores for balance made the loop's own cost in wave 0 four times what it saved)
	// (four passes' LDS reads -- up to three levels of them, bits -> run -> value -- then their stores: a pass at a time
	//  waits the levels out once per pass, a sixth of a tile's life on read depth by round 5's stamps)
	double* dst = out + keepStart;
	auto store_pairs = [&] (int q, int qn, bool flatToo, bool flatOnly)
		{
		constexpr int AHEAD = PK_AHEAD;
		for (int c0 = keepLo + 2*q ; (q >= 0) && (c0 < keepHi) ; c0 += AHEAD * 2*qn)      // (keepLo is even)
			{
			uint32_t code[AHEAD];
			double   T[AHEAD];
#pragma unroll
			for (int j=0 ; j<AHEAD ; j++)
				{
				const int c  = c0 + j * 2*qn;
				const int cr = min (c, G::OUT - 2);                      // (a pair past the tile's end reads the last words and stores nothing)
				code[j] = (codeBits[cr >> 4] >> (2 * (cr & 15))) & 15u;   // (c is even: a pair lies in one block)
				T[j]    = (CWM && flatToo)? runT[blockRun[cr >> 4] & (PK_LEAD_CAP - 1)] : 0.0;      // (a block without a flat base: any run's)
				}
#pragma unroll
			for (int j=0 ; j<AHEAD ; j++)
				{
				const int c = c0 + j * 2*qn;
				if (c >= keepHi) continue;
				const uint32_t a0 = code[j] & 3u, a1 = code[j] >> 2;
				const bool     fl = CWM && ((a0 == 2u) || (a1 == 2u));
				if (CWM && (fl? !flatToo : flatOnly)) continue;          // (a pair with a flat base: with the others, or in the pass behind the leaders' chains)
				const double r0 = (a0 == 2u)? T[j] : (a0 == 1u)? 0.0 : fill, r1 = (a1 == 2u)? T[j] : (a1 == 1u)? 0.0 : fill;
				if (((a0 != 3u) && (a1 != 3u)) && (c + 1 < keepHi)) gdsp_st2 (reinterpret_cast<double2*> (dst + (c - keepLo)), make_double2 (r0, r1));
				else
					{
					if (a0 != 3u)                        dst[c - keepLo]     = r0;
					if ((a1 != 3u) && (c + 1 < keepHi)) dst[c - keepLo + 1] = r1;
					}
				}
			}
		};
	const int  chainWaves = (evals + 63) / 64;                     // waves 0 .. chainWaves-1 ran the chain of taps (uniform)
	const bool share      = (chainWaves >= 1) && (chainWaves < NW) && (!inPlace || (PK_SHARE_INPLACE && (chainWaves == 1)));
	store_pairs (share? p - 64 * chainWaves : p, share? HN_THREADS - 64 * chainWaves : HN_THREADS, flatNow, false);
	if (inPlace)                                                   // (uniform over the workgroup)
		{
		__syncthreads ();                                          // (the neighbourhoods' exact values are in)
		if (p < nq)
			{
			const double* v = &exactVal[p * (2*HH + 1)];
			const double centre = v[HH];
			double ext = MAX? -INFINITY : INFINITY;
#pragma unroll
			for (int t=0 ; t<2*HH+1 ; t++) { if (t != HH) ext = MAX? fmax (ext, v[t]) : fmin (ext, v[t]); }
			out[compStart + needList[p]] = (MAX? (ext > centre) : (ext < centre))? fill : centre;
			}
		}
	// (the flat bases of a tile with a run outside the table after the store loop, not before it: their values wait for the
	//  leaders' chains in the first waves, and the other waves have the tile's stores to issue meanwhile)
	if (flatLater)                                                 // (uniform)
		{
		resolve_runs ();
		__syncthreads ();
		store_pairs (p, HN_THREADS, true, true);
		}

	if (!PROBE) PK_STAMP (lds, 7);                                 // the store loop issued
	}

template <int W, bool MAX, int HH, bool CWM>
__global__ __launch_bounds__(HN_THREADS)
void peaks_probe_kernel (GdspBatch B, HannConsts<W, PK_E> K, GdspPeaksCtl* ctl)
	{
	const uint32_t v = blockIdx.x / PK_PROBE_TILES, j = blockIdx.x % PK_PROBE_TILES;
	const uint32_t tiles = B.tile0[v+1] - B.tile0[v];
	const uint32_t take  = (tiles < PK_PROBE_TILES)? tiles : PK_PROBE_TILES;
	if (j >= take) return;
	const uint32_t tile = (uint32_t) (((uint64_t) j * tiles) / take);
	peaks_filter_tile<W, false, MAX, HH, true, CWM> (B.in[v], NULL, B.n[v], tile, K, NULL, 0.0, &ctl[v], NULL, NULL, 0, NULL, 0, NULL);
	}

// The filter over the vectors of ONE form: plain, or (CWM) the form for vectors of flat stretches that writes a run's value
// to all its bases -- which form a vector takes is the probe's count (read depth: flat), read back by the host between the
// probe and this launch: the two forms in one kernel bring an LDS image each (one workgroup per CU), two launches over all
// the vectors leave a workgroup per tile idle on the vectors that are the other's (5-8 % of a real-valued genome), and a
// workgroup walking a group of tiles runs a third slower than a workgroup per tile.  S: the vectors of this form and the
// grid over them; M: each one's number in the whole table (control words) and its first tile there (strips, counts, lists).
struct PeaksSub { uint32_t v[GDSP_BATCH_MAX], gt0[GDSP_BATCH_MAX]; };
template <int W, bool FMA, bool MAX, int HH, bool CWM>
__global__ __launch_bounds__(HN_THREADS) __attribute__((amdgpu_waves_per_eu(PK_WAVES, PK_WAVES)))
void peaks_filter_kernel (GdspBatch S, PeaksSub M, HannConsts<W, PK_E> K, const double* __restrict__ d_taps, double fill, GdspPeaksCtl* ctl,
                          uint16_t* strips, uint32_t* counts, uint32_t cap, uint32_t* tileList, int gate, const double* __restrict__ runTable)
	{
	const double* in;  double* out;  uint32_t n, i;
	const uint32_t tile = gdsp_batch_tile (S, in, out, n, &i);
	const uint32_t v = M.v[i], gt = M.gt0[i] + tile;               // the tile's number in the whole table: its strip and its count
	// (only where the host did not look -- GDSP_PEAKS_FLAT=0: otherwise a load every workgroup would wait for before it
	//  may issue its tile's)
	if (gate && gdsp_peaks_takes_direct (ctl[v])) return;
	peaks_filter_tile<W, FMA, MAX, HH, false, CWM> (in, out, n, tile, K, d_taps, fill, &ctl[v], strips + (size_t) gt * cap, counts + gt, cap,
	                                                tileList + M.gt0[i], gt, runTable);
	}

// The exact kernel, for the bases whose comparison needs exact values (ties and near-ties): a workgroup takes a tile from
// its vector's list of tiles that queued something, 16 lanes per queued base i.  Inputs x[i-h-H .. i+h+H] (zero outside
// the vector) are staged in the group's LDS strip; lane L < 2h+1 evaluates the smoothed value of base i-h+L tap by tap,
// ascending, multiply then add (FMA: fused, for --smooth=fma) -- the reference's operations in the reference's order,
// sum.c:655-663; the test of minmax.c:1195-1216 follows on those values.
template <int W, bool FMA, bool MAX>
__global__ __launch_bounds__(HN_THREADS)
void peaks_exact_kernel (GdspBatch B, PeaksTaps<W> taps, int h, double fill, const GdspPeaksCtl* __restrict__ ctl,
                         const uint16_t* __restrict__ strips, const uint32_t* __restrict__ counts, uint32_t cap,
                         const uint32_t* __restrict__ tileList)
	{
	typedef HannGeom<W, PK_E> G;
	constexpr int H = (W - 1) / 2;
	constexpr int GROUPS = HN_THREADS / PK_GROUP;
	__shared__ double xs[GROUPS][PK_XS];
	const int    sh = h & 1, stride = G::OUT - 2*h - 2*sh, keepLo = h + sh;
	const int    grp = threadIdx.x / PK_GROUP, L = threadIdx.x % PK_GROUP;
	const double never = MAX? -INFINITY : INFINITY;
	const int    need = 2*h + 1 + 2*H;                             // staged inputs per base

	for (uint32_t v=0 ; v<B.nvec ; v++)
		{
		const GdspPeaksCtl c = ctl[v];
		if (gdsp_peaks_takes_direct (c) || (c.overflow != 0)) continue;      // the direct kernel rewrites the whole vector
		const double*  in  = B.in[v];
		double*        out = B.out[v];
		const int64_t  n   = (int64_t) B.n[v];
		for (uint32_t k=blockIdx.x ; k<c.count ; k+=gridDim.x)
			{
			const uint32_t gt   = tileList[B.tile0[v] + k];
			const uint32_t tile = gt - B.tile0[v];
			uint32_t count = counts[gt];
			const int64_t keepStart = (int64_t) tile * stride, compStart = keepStart - keepLo;
			const bool    whole = (count == PK_WHOLE_TILE);
			if (whole) count = (uint32_t) ((keepStart + stride <= n)? stride : (n - keepStart));
			const uint16_t* strip = strips + (size_t) gt * cap;
			// every group of a wave makes the same number of trips (the last ones idle), so the wave-level barriers below meet
			const uint32_t trips = (count + GROUPS - 1) / GROUPS;
			for (uint32_t t=0 ; t<trips ; t++)
				{
				const uint32_t idx  = t * GROUPS + grp;
				const bool     busy = (idx < count);
				const int64_t  i    = !busy? 0 : whole? keepStart + idx : compStart + strip[idx];
				const int64_t  g0   = i - h - H;
				if (busy)
					{
					for (int e=L ; e<need ; e+=PK_GROUP)
						{
						const int64_t g = g0 + e;
						xs[grp][e] = ((g >= 0) && (g < n))? in[g] : 0.0;
						}
					}
				__builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier ();
				double a = 0.0;
				if (busy && (L <= 2*h))
					{
					const double* x = &xs[grp][L];
#pragma unroll
					for (int kk=0 ; kk<W ; kk++)
						a = FMA? __builtin_fma (taps.w[kk], x[kk], a) : a + taps.w[kk] * x[kk];
					}
				const int64_t j = i - h + L;
				const double  s = (busy && (L <= 2*h) && (j >= 0) && (j < n))? a : never;
				const double  centre = __shfl (s, h, PK_GROUP);
				double ext = (L == h)? never : s;                      // the extreme of the neighbours
#pragma unroll
				for (int m=PK_GROUP/2 ; m>0 ; m>>=1)
					{
					const double o = __shfl_xor (ext, m, PK_GROUP);
					ext = MAX? fmax (ext, o) : fmin (ext, o);
					}
				if (busy && (L == h)) out[i] = (MAX? (ext > centre) : (ext < centre))? fill : centre;
				__builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier ();                          // the strip is free for the next trip
				}
			}
		}
	}

#ifdef PK_STAMPS
extern "C" int gdsp_peaks_stamps (unsigned long long* out, int clear)
	{
	GDSP_HIP_TRY (hipDeviceSynchronize ());
	GDSP_HIP_TRY (hipMemcpyFromSymbol (out, HIP_SYMBOL (pkStamps), 16 * sizeof(unsigned long long)));
	if (clear) { unsigned long long z[16] = { 0 };  GDSP_HIP_TRY (hipMemcpyToSymbol (HIP_SYMBOL (pkStamps), z, sizeof(z))); }
	return GDSP_OK;
	}
#endif

// ------------------------------------------------------------------- host ----
struct PeaksWork { int device;  void* stream;  GdspPeaksCtl* ctl;  uint16_t* strips;  uint32_t* counts;  uint32_t* tileList;  size_t tiles;  double* runTable; };

// The smoothed value of a base whose whole window holds the count x: the chain of taps of peaks_filter_tile (and of the
// reference, sum.c:655-663) on W equal operands -- tap by tap, ascending, multiply then add from 0.0; entry PK_RUN_TABLE + x
// the same with fused multiply-adds (--smooth=fma).  It depends on the window alone, so it is walked once per (device,
// stream), on that stream, before the first filter launch there.
template <int W>
__global__ void peaks_run_table_kernel (const double* __restrict__ d_taps, double* __restrict__ table)
	{
	const double x = (double) threadIdx.x;
	double a = 0.0, f = 0.0;
	for (int k=0 ; k<W ; k++) { a = a + d_taps[k] * x;  f = __builtin_fma (d_taps[k], x, f); }
	table[threadIdx.x] = a;  table[PK_RUN_TABLE + threadIdx.x] = f;
	}
static std::vector<PeaksWork> peaksWork;
static std::mutex peaksWorkLock;

// the control words, strips and counts of (device, stream): allocated on first use, grown when a table has more tiles
static int peaks_work (void* stream, size_t tiles, PeaksWork* out)
	{
	int device = 0;
	GDSP_HIP_TRY (hipGetDevice (&device));
	std::lock_guard<std::mutex> hold (peaksWorkLock);
	PeaksWork* w = NULL;
	for (PeaksWork& e : peaksWork) { if ((e.device == device) && (e.stream == stream)) w = &e; }
	if (w == NULL)
		{
		PeaksWork e = { device, stream, NULL, NULL, NULL, NULL, 0, NULL };
		const double* d_taps = NULL;
		int rc = gdsp_smooth_taps_device (101, &d_taps);
		if (rc != GDSP_OK) return rc;
		GDSP_HIP_TRY (hipMalloc ((void**) &e.ctl, GDSP_BATCH_MAX * sizeof(GdspPeaksCtl)));
		GDSP_HIP_TRY (hipMalloc ((void**) &e.runTable, 2 * PK_RUN_TABLE * sizeof(double)));
		hipLaunchKernelGGL ((peaks_run_table_kernel<101>), dim3(1), dim3(PK_RUN_TABLE), 0, gdsp_stream (stream), d_taps, e.runTable);
		GDSP_LAUNCH_CHECK ();
		peaksWork.push_back (e);
		w = &peaksWork.back ();
		}
	if (w->tiles < tiles)
		{
		if (w->strips != NULL)
			{
			GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (stream)));        // (a launch in flight may still read them)
			GDSP_HIP_TRY (hipFree (w->strips));
			GDSP_HIP_TRY (hipFree (w->counts));
			GDSP_HIP_TRY (hipFree (w->tileList));
			w->strips = NULL;  w->counts = NULL;  w->tileList = NULL;  w->tiles = 0;
			}
		const size_t want = tiles + tiles / 8 + 64;
		GDSP_HIP_TRY (hipMalloc ((void**) &w->strips, want * PK_TILE_CAP * sizeof(uint16_t)));
		GDSP_HIP_TRY (hipMalloc ((void**) &w->counts, want * sizeof(uint32_t)));
		GDSP_HIP_TRY (hipMalloc ((void**) &w->tileList, want * sizeof(uint32_t)));
		w->tiles = want;
		}
	*out = *w;
	return GDSP_OK;
	}

// --smooth=fma (and --smooth=hann in front of localmin / localmax, which is evaluated as fma) takes the filter too since round
// 4: with the filter on block sums of all taps it runs at 1.3 x the direct kernel's one instruction per tap.
// GDSP_PEAKS_FILTER=exact keeps the direct kernel for fma (rounds 2-3, tests), GDSP_PEAKS_FILTER=0 for both arithmetics
bool gdsp_peaks_filter_wanted_for_fma (void)
	{ const char* e = getenv ("GDSP_PEAKS_FILTER");  return (e == NULL) || (strcmp (e, "exact") != 0); }

bool gdsp_peaks_filter_available (uint32_t W, uint32_t N)
	{
	const char* e = getenv ("GDSP_PEAKS_FILTER");                  // GDSP_PEAKS_FILTER=0: every base through the direct kernel (A/B, tests)
	if ((e != NULL) && (strcmp (e, "0") == 0)) return false;
	return (W == 101) && (N >= 3) && ((N - 1) / 2 <= PK_HMAX);
	}

template <bool MAX, int HH>
static int peaks_launch (const gdsp_batch_item* items, int count, const HannConsts<101, PK_E>& K, const PeaksTaps<101>& taps, int fma,
                         double fill, const double* h_taps, void* stream)
	{
	typedef HannGeom<101, PK_E> G;
	constexpr int stride = G::OUT - 2*HH - 2*(HH & 1);
	hipStream_t s = gdsp_stream (stream);
	GdspBatch B;
	gdsp_batch_make (B, items, count, [] (uint32_t n) { return ((uint64_t) n + stride - 1) / stride; });
	// test hooks: GDSP_PEAKS_ROUTE=filter|direct overrides the probe, GDSP_PEAKS_QUEUE_CAP=<positions> shrinks every strip
	// (a strip that overflows sends its vector through the direct kernel as well)
	const char* routeEnv = getenv ("GDSP_PEAKS_ROUTE");
	const int   route    = (routeEnv == NULL)? 0 : (strcmp (routeEnv, "filter") == 0)? 1 : (strcmp (routeEnv, "direct") == 0)? 2 : 0;
	const char* capEnv   = getenv ("GDSP_PEAKS_QUEUE_CAP");
	uint32_t    cap      = PK_TILE_CAP;
	if ((capEnv != NULL) && (atoll (capEnv) >= 0) && (atoll (capEnv) < PK_TILE_CAP)) cap = (uint32_t) atoll (capEnv);
	const uint32_t tiles = B.tile0[GDSP_BATCH_MAX];
	PeaksWork w;
	int rc = peaks_work (stream, tiles, &w);
	if (rc != GDSP_OK) return rc;

	hipLaunchKernelGGL (peaks_init_kernel, dim3(1), dim3(64), 0, s, w.ctl, B, (uint32_t) stride, route);
	const double* d_taps = NULL;                                   // the reference's window on the device (gdsp_fir.hip: cached per device)
	rc = gdsp_smooth_taps_device (101, &d_taps);
	if (rc != GDSP_OK) return rc;
	// GDSP_PEAKS_FLAT=0: no form for flat stretches (rounds 3-4a: the probe sends read depth to the direct kernel) and no read-back
	const char* flatEnv = getenv ("GDSP_PEAKS_FLAT");
	const bool  flatForm = !((flatEnv != NULL) && (strcmp (flatEnv, "0") == 0));
	if (flatForm) hipLaunchKernelGGL ((peaks_probe_kernel<101, MAX, HH, true>),  dim3(count * PK_PROBE_TILES), dim3(HN_THREADS), 0, s, B, K, w.ctl);
	else          hipLaunchKernelGGL ((peaks_probe_kernel<101, MAX, HH, false>), dim3(count * PK_PROBE_TILES), dim3(HN_THREADS), 0, s, B, K, w.ctl);
	// which vectors take which form: the probe's counts, read back (the one wait of the route: ~15 us per table of up to 32
	// vectors, 0.2 % of a genome-wide launch); without the flat form every vector that does not go direct is plain and the
	// kernels decide by themselves
	GdspPeaksCtl h_ctl[GDSP_BATCH_MAX];
	memset (h_ctl, 0, sizeof(h_ctl));
	if (flatForm)
		{
		GDSP_HIP_TRY (hipMemcpyAsync (h_ctl, w.ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, s));
		GDSP_HIP_TRY (hipStreamSynchronize (s));
		}
	const int exactBlocks = 4096;
	for (int form=0 ; form<(flatForm? 2 : 1) ; form++)
		{
		std::vector<gdsp_batch_item> sub;
		PeaksSub M;
		memset (&M, 0, sizeof(M));
		for (int v=0 ; v<count ; v++)
			{
			const GdspPeaksCtl& c = h_ctl[v];
			// the rule of gdsp_common.h, which the kernels behind apply to the same words: a vector of flat stretches takes the
			// CWM form; in any other the bases that would have been written their run's value tie and are queued, and count
			// towards the direct kernel's threshold (sparse read depth with a few per cent of them would otherwise overflow its
			// strips in the filter and run the direct kernel as well)
			const bool flat   = flatForm && gdsp_peaks_flat_form (c);
			const bool direct = flatForm && gdsp_peaks_takes_direct (c);
			if (direct || (flat != (form == 1))) continue;
			M.v[sub.size ()] = (uint32_t) v;  M.gt0[sub.size ()] = B.tile0[v];
			sub.push_back (items[v]);
			}
		if (sub.empty ()) continue;
		GdspBatch S;
		gdsp_batch_make (S, sub.data (), (int) sub.size (), [] (uint32_t n) { return ((uint64_t) n + stride - 1) / stride; });
		const dim3 grid (S.tile0[GDSP_BATCH_MAX]), block (HN_THREADS);
#define PK_FILTER(FMAV, CWMV) hipLaunchKernelGGL ((peaks_filter_kernel<101, FMAV, MAX, HH, CWMV>), grid, block, 0, s, S, M, K, d_taps, fill, w.ctl, w.strips, w.counts, cap, w.tileList, flatForm? 0 : 1, w.runTable + (FMAV? PK_RUN_TABLE : 0))
		if (form == 1) { if (fma) PK_FILTER (true, true);   else PK_FILTER (false, true); }
		else           { if (fma) PK_FILTER (true, false);  else PK_FILTER (false, false); }
#undef PK_FILTER
		}
	if (fma) hipLaunchKernelGGL ((peaks_exact_kernel<101, true, MAX>),  dim3(exactBlocks), dim3(HN_THREADS), 0, s, B, taps, HH, fill, w.ctl, w.strips, w.counts, cap, w.tileList);
	else     hipLaunchKernelGGL ((peaks_exact_kernel<101, false, MAX>), dim3(exactBlocks), dim3(HN_THREADS), 0, s, B, taps, HH, fill, w.ctl, w.strips, w.counts, cap, w.tileList);
	GDSP_LAUNCH_CHECK ();
	return gdsp_fir_extrema_gated_launch (items, count, w.ctl, h_taps, fma, HH, MAX, fill, stream);
	}

int gdsp_peaks_filter_batch (const gdsp_batch_item* items, int nitems, const double* h_taps, int fma, uint32_t N, int wantMax,
                             double fill, void* stream)
	{
	const int h = (int) ((N - 1) / 2);
	GDSP_REQUIRE ((h >= 1) && (h <= PK_HMAX), "neighbourhood outside the filter's range");
	HannConsts<101, PK_E> K;
	hann_consts<101, PK_E> (K);
	PeaksTaps<101> taps;
	memcpy (taps.w, h_taps, sizeof(taps.w));
	std::vector<gdsp_batch_item> live;
	for (int i=0 ; i<nitems ; i++) { if (items[i].n != 0) live.push_back (items[i]); }
	for (size_t at=0 ; at<live.size () ; at+=GDSP_BATCH_MAX)
		{
		const int count = (int) std::min<size_t> (GDSP_BATCH_MAX, live.size () - at);
		int rc;
#define PK_CASE(HHH) case HHH: rc = wantMax? peaks_launch<true, HHH>  (&live[at], count, K, taps, fma, fill, h_taps, stream) \
                                           : peaks_launch<false, HHH> (&live[at], count, K, taps, fma, fill, h_taps, stream);  break;
		switch (h)
			{
			PK_CASE (1) PK_CASE (2) PK_CASE (3) PK_CASE (4) PK_CASE (5) PK_CASE (6)
			default: rc = wantMax? peaks_launch<true, 7>  (&live[at], count, K, taps, fma, fill, h_taps, stream)
			                     : peaks_launch<false, 7> (&live[at], count, K, taps, fma, fill, h_taps, stream);  break;
			}
#undef PK_CASE
		if (rc != GDSP_OK) return rc;
		}
	return GDSP_OK;
	}
