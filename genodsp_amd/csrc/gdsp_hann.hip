// gdsp_hann.hip -- `smooth` evaluated through the structure of its window (GDSP_FIR_HANN).
//
// The reference's window (sum.c:632-645) is a Hann window: tap k = c * (1 - cos(w*k)), k = 1..W,
// with w = 2*pi/(W+1) and c = 1/(2 * sum of the unnormalised taps).  Over any stretch of taps
//     sum_k (1 - cos(w*k)) * x[a-1+k]  =  S - C,   S = sum of the inputs under the stretch,
//                                                  C = Re( exp(j*phase) * sum of x[e]*exp(j*w*e) )
// and sums over a stretch need no multiply per tap.  Cut the line into blocks of 16 elements: a
// stretch [a',b'] is a suffix of a's block + the whole blocks between + a prefix of b's block --
// additions only, no differences of running sums, so nothing cancels in the sums themselves.
// S - C does cancel where the taps are small, so the E = 8 taps at either end of the window are
// evaluated directly (16 multiply-adds per base) and only the W-2E middle taps, all at least
// 0.15 of the largest, go through the block sums.  ~45 FP64 operations per base instead of 101
// multiply-adds: `smooth W=101` stops being bound by the FP64 pipe (gdsp_fir.hip) and becomes an
// HBM stream.
//
// One thread owns one block as the right end b' of the middle stretch of 16 windows:
//   phase 0  the 2E direct taps of its 16 outputs, sliding over 15+E inputs on either side;
//   phase 1  prefix sums of its own block, three sequences (x, x*cos, x*sin), kept in registers;
//            block totals to LDS;
//   phase 2  totals of the whole blocks between (rotated to this block's phase -- every phase
//            factor in the kernel is relative to the owner's block, so all of them are constants
//            of W that reach the VALU as scalar operands: no table, no sincos); then one
//            backward walk over the 16+dr elements that hold the 16 left ends a', accumulating
//            the suffix sums and finishing one output per step.
// Elements sit in LDS with a pitch of 17 per block of 16, so the lane-strided ds_read_b64 of
// all walks are conflict free; results return through the same LDS image for 16-byte coalesced
// stores.  40 KiB of LDS per workgroup to the byte and 128 registers: 4 workgroups per CU (round 5; 41 KiB, 146 registers
// and 3 until then: hann_blocks_tile below).
//
// Not bit-identical to the reference (different association, exact cosines instead of the
// normalised taps' roundings): within the north star's one rounding per floating-point
// operation, W * 2^-52 * sum|w_k v_k| as for FMA, and as close to an extended-precision
// evaluation as the reference itself -- both asserted in tests/test_hip_parity.py.
// It is also not shift invariant: which additions meet first depends on where an output falls in
// its block of 16, so a flat stretch of input gives outputs that differ in their last bits, where
// the direct kernels (every output the same operations in the same order) give one repeated value.
// Operators that compare neighbours strictly (localmax after smooth) and the run collapse of the
// report see such ties; that is why EXACT stays the default of the command line, why the fused
// smooth+extrema kernel evaluates directly whatever the mode, and why this mode is for the
// smoothed track as a floating-point result.
// Range: the block sums add up to 85 unweighted inputs, which would overflow on inputs beyond DBL_MAX/128
// where the weighted sum does not, and a window holding an infinity would give inf - inf.  So every thread
// looks at the exponents of its own block while it builds the prefix sums (one integer max per element), the
// four waves' verdicts ride the barrier that is there anyway, and a tile that holds a NaN, an infinity or a
// magnitude of 2^1017 or more is evaluated tap by tap instead (hann_direct_tile: ascending fused multiply-adds,
// the taps as data -- the very operations of GDSP_FIR_FMA, so such a tile is bit-identical to that mode and
// inf / NaN / DBL_MAX (what `localmin` leaves behind, minmax.c:901) come out as direct evaluation gives them).
// With that, the mode differs from the reference by rounding only, on any input.

#include <math.h>
#include <stdlib.h>
#include <float.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <vector>
#include "gdsp_common.h"

#include "gdsp_hann_tile.h"

// HN_FOUR: four workgroups per CU -- 40 960 bytes of LDS (the waves' verdicts in the image's pad slots) and 128 registers
// (the direct taps at the windows' ends behind phase 2, the prefix sums' ends re-made there: gdsp_hann_tile.h)
#ifndef HN_FOUR
#define HN_FOUR 1
#endif
#if HN_FOUR
#define HN_FOUR_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define HN_FOUR_WAVES
#endif
template <int W>
__device__ __forceinline__ void hann_blocks_tile (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t tile,
                                                  const HannConsts<W>& K, const double* __restrict__ taps)
	{
	typedef HannGeom<W> G;
	__shared__ __attribute__((aligned(16))) double lds[HN_THREADS * HN_PITCH];
	__shared__ double tot[3][HN_THREADS];
#if HN_FOUR
	uint32_t* const huge = NULL;                                   // (the waves' verdicts in the image's pad slots: 40 KiB to the byte)
#else
	__shared__ __attribute__((aligned(16))) uint32_t huge[HN_THREADS/64];
#endif

	const int64_t  out0 = (int64_t) tile * G::OUT;
	const int64_t  e0   = out0 - G::LEAD;                         // first staged element (even)
	const int      p    = threadIdx.x;
	const bool     live = (p >= G::HALO_L) && (p < HN_THREADS - G::HALO_R);

	double acc[HN_G];
	const bool direct = hann_tile_sums<W, false, HN_E, HN_FOUR != 0, false, false, false, HN_FOUR != 0> (lds, tot, huge, in, n, e0, K, acc);
	if (direct) hann_direct_tile (lds, taps, W, G::LO, G::OUT);    // (output o sits under taps LO+o .. LO+o+W-1 of the staged elements)
	else
		{
		__syncthreads ();                                          // every read of the staged inputs is done

		// ---- results back through LDS: output o of the tile belongs to thread HALO_L + o/16
		if (live)
			{
			double* mine = lds + (p - G::HALO_L) * HN_PITCH;
#pragma unroll
			for (int u=0 ; u<HN_G ; u++) mine[u] = acc[u];
			}
		__syncthreads ();
		}

	if (out0 + G::OUT <= (int64_t) n)
		{
		double2* dst = reinterpret_cast<double2*> (out + out0);
#pragma unroll
		for (int u=0 ; u<(G::OUT/2 + HN_THREADS - 1)/HN_THREADS ; u++)
			{
			const int q = u*HN_THREADS + p;
			if (q < G::OUT/2)
				{
				const int o = 2*q;
				const double* src = lds + o + (o >> 4);
				gdsp_st2 (&dst[q], make_double2 (src[0], src[1]));
				}
			}
		}
	else
		{
		for (int o=p ; o<G::OUT ; o+=HN_THREADS)
			{ if (out0 + o < (int64_t) n) out[out0 + o] = lds[o + (o >> 4)]; }
		}
	}

template <int W>
__global__ __launch_bounds__(HN_THREADS) HN_FOUR_WAVES
void hann_blocks_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t ntiles,
                         HannConsts<W> K, const double* __restrict__ taps)
	{ hann_blocks_tile<W> (in, out, n, gdsp_xcd_tile (blockIdx.x, ntiles), K, taps); }

template <int W>                                                  // one grid over every vector of the table (gdsp_common.h)
__global__ __launch_bounds__(HN_THREADS) HN_FOUR_WAVES
void hann_blocks_batch_kernel (GdspBatch B, HannConsts<W> K, const double* __restrict__ taps)
	{
	const double* in;  double* out;  uint32_t n;
	const uint32_t tile = gdsp_batch_tile (B, in, out, n);
	hann_blocks_tile<W> (in, out, n, tile, K, taps);
	}

template <int W>
static void hann_launch (const double* d_in, double* d_out, uint32_t n, const double* d_taps, hipStream_t s)
	{
	typedef HannGeom<W> G;
	HannConsts<W> K;
	hann_consts<W> (K);
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + G::OUT - 1) / G::OUT);
	hipLaunchKernelGGL ((hann_blocks_kernel<W>), dim3(ntiles), dim3(HN_THREADS), 0, s, d_in, d_out, n, ntiles, K, d_taps);
	}

// ------------------------------------------------------ any window, 81 .. 2001 ----
// The same evaluation with the geometry as run-time numbers: dq, dr, the number of whole blocks
// between (their rotations come from a small table in HBM, read as scalars), the halo.  Only the
// number of direct taps E stays a template parameter: the cancellation in S - C is bounded by the
// smallest tap that goes through the block sums, and the bound of one rounding per operation
// grows with W, which makes E ~ sqrt(0.15 W) enough (8 for W = 101, 16 for 1001, 20 for 2001);
// E also carries the parity that keeps a tile's first element 16-byte aligned.
#define HN_RT_SMALL_MAX_W 1701                                  // up to here the 256-thread form is the faster one (measured)
#define HN_RT_SCAN_MIN_W  1001                                  // from here on it scans its block totals instead of walking them (measured: 801 -4 %, 1001 +4 %, 1501 +20 %)
#define HN_RT_MAX_NT 512
#define HN_RT_BIG    768                                        // threads of the long-window form: 12288 staged elements, 122 KiB of LDS
struct HannRT
	{
	int    DQ, DR, NT, HALO_L, HALO_R, LO, LEAD, OUT, SEG;
	double scale;
	double edge[32];                                               // 1 - cos(w k), k = 1..E
	double ownC[HN_G], ownS[HN_G];
	double leftC[2*HN_G], leftS[2*HN_G];                           // exp(+j w (u - DM)), u = 0..15+DR
	double demC[HN_G], demS[HN_G];                                 // exp(+j w (W-E - s))
	};

// THREADS = 256 stages 4096 elements (41 KiB of LDS, three workgroups per CU); windows of a thousand taps and more lose a
// quarter to a half of such a tile to their halo and spend most of their time adding up the NT whole blocks between
// the two ends one by one.  THREADS = 768 stages 12288 elements (122 KiB, one workgroup of twelve waves per CU; two
// 448-thread workgroups per CU were slower at every window tried) and adds
// the blocks between in two levels: the totals of aligned groups of 16 blocks, plus at most 15 single blocks at either
// end of the range (additions only, as before; fewer roundings than one by one).
#ifndef HN_RT_REDO
#define HN_RT_REDO 4
#endif
template <int E, int THREADS, bool SCAN>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3)))
void hann_blocks_rt_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t ntiles,
                            HannRT K, const double2* __restrict__ rot, const double* __restrict__ taps, int W)
	{
	constexpr int NEDGE = HN_G + E - 1;
	constexpr int ELEMS = THREADS * HN_G;
	__shared__ __attribute__((aligned(16))) double lds[THREADS * HN_PITCH];
	__shared__ double tot[3][THREADS];                                // the blocks of p's segment up to and including p, in p's phase
	__shared__ double vsf[3][SCAN? THREADS : 1];                      // the blocks of p's segment from p on, in p's phase
	__shared__ __attribute__((aligned(16))) uint32_t huge[16];

	const uint32_t tile = gdsp_xcd_tile (blockIdx.x, ntiles);
	const int64_t  out0 = (int64_t) tile * K.OUT;
	const int64_t  e0   = out0 - K.LEAD;                          // first staged element (even)
	const int      p    = threadIdx.x;
	const bool     live = (p >= K.HALO_L) && (p < THREADS - K.HALO_R);

	if ((e0 >= 0) && (e0 + ELEMS <= (int64_t) n))
		{
		const double2* src = reinterpret_cast<const double2*> (in + e0);
		double2 r[HN_G/2];
#pragma unroll
		for (int u=0 ; u<HN_G/2 ; u++) r[u] = gdsp_ld2 (&src[u*THREADS + p]);
#pragma unroll
		for (int u=0 ; u<HN_G/2 ; u++)
			{
			const int e = 2 * (u*THREADS + p);
			double* dst = lds + e + (e >> 4);
			dst[0] = r[u].x;  dst[1] = r[u].y;
			}
		}
	else
		{
		for (int e=p ; e<ELEMS ; e+=THREADS)
			{
			const int64_t g = e0 + e;
			lds[e + (e >> 4)] = ((g >= 0) && (g < (int64_t) n))? in[g] : 0.0;
			}
		}
	__syncthreads ();

	// ---- phase 0: the E taps at either end of each of the 16 windows, directly
	double acc[HN_G];
#pragma unroll
	for (int s=0 ; s<HN_G ; s++) acc[s] = 0.0;
	if (live)
		{
		const double* xl = lds + (p - K.HALO_L) * HN_PITCH;         // element b'-BACK of s=0 is xl[LO]
#pragma unroll
		for (int j=0 ; j<NEDGE ; j++)
			{
			const int    o = K.LO + j;
			const double x = xl[o + (o >> 4)];
#pragma unroll
			for (int s=0 ; s<HN_G ; s++)
				{ if ((j - s >= 0) && (j - s < E)) acc[s] = __builtin_fma (K.edge[j-s], x, acc[s]); }
			}
		const double* xr = lds + p * HN_PITCH;
#pragma unroll
		for (int j=1 ; j<=NEDGE ; j++)
			{
			const double x = xr[j + (j >> 4)];
#pragma unroll
			for (int s=0 ; s<HN_G ; s++)
				{ if ((j - s >= 1) && (j - s <= E)) acc[s] = __builtin_fma (K.edge[E - (j-s)], x, acc[s]); }
			}
		}

	// ---- phase 1: prefix sums of the own block in the own phase
	double P0[HN_G], Pr[HN_G], Pi[HN_G];
		{
		const double* xb = lds + p * HN_PITCH;
		double a0 = 0.0, ar = 0.0, ai = 0.0;
		uint32_t big = 0;                                          // largest exponent seen in the own block
#pragma unroll
		for (int u=0 ; u<HN_G ; u++)
			{
			const double x = xb[u];
			big = max (big, hann_magnitude_hi (x));
			a0 += x;
			ar  = __builtin_fma (x, K.ownC[u], ar);
			ai  = __builtin_fma (x, K.ownS[u], ai);
			P0[u] = a0;  Pr[u] = ar;  Pi[u] = ai;
			}
		// The whole blocks between a window's two ends, p-NT .. p-1, are a run of NT consecutive block totals.  Cut the
		// tile's blocks into segments of SEG = 32: a run of NT >= SEG blocks is the tail of its
		// first segment + whole segments + the head of its last one.  Heads and tails come from two scans inside the
		// segments, done with wave shuffles right here (five steps each way, each step's rotation a constant of W),
		// so the walk over NT blocks of round 1 -- three LDS reads and a rotation per block -- becomes at most five
		// rotated terms per thread.  Additions only, as before.
		double i0 = a0, ir = ar, ii = ai, v0 = a0, vr = ar, vi = ai;
		if (SCAN && (K.SEG != 0))
			{
			const int inSeg = p & (K.SEG - 1);
			for (int d=1 ; d<K.SEG ; d*=2)
				{
				const double2 w  = rot[d-1];                        // exp(-j w 16 d)
				const double  u0 = __shfl_up (i0, d, 64), ur = __shfl_up (ir, d, 64), ui = __shfl_up (ii, d, 64);
				if (inSeg >= d) { i0 += u0;  ir += __builtin_fma (ur, w.x, -(ui * w.y));  ii += __builtin_fma (ur, w.y, ui * w.x); }
				const double  d0 = __shfl_down (v0, d, 64), dr = __shfl_down (vr, d, 64), di = __shfl_down (vi, d, 64);
				if (inSeg + d < K.SEG) { v0 += d0;  vr += __builtin_fma (dr, w.x, di * w.y);  vi += __builtin_fma (di, w.x, -(dr * w.y)); }
				}
			vsf[0][p] = v0;  vsf[1][p] = vr;  vsf[2][p] = vi;
			}
		tot[0][p] = i0;  tot[1][p] = ir;  tot[2][p] = ii;           // (SEG == 0: the block's own totals, for the walk)
		const bool any = (__builtin_amdgcn_ballot_w64 (big >= HN_HUGE_HI) != 0);   // the blocks are the whole tile
		if ((p & 63) == 0) huge[p >> 6] = any? 1u : 0u;
		}
	__syncthreads ();
	uint32_t anyHuge = 0;
#pragma unroll
	for (int w=0 ; w<THREADS/64 ; w++) anyHuge |= huge[w];
	const bool direct = (anyHuge != 0);                            // uniform over the workgroup

	// ---- phase 2: the middle stretch of one window per left end
	if (live && !direct)
		{
		double T0 = 0.0, Tr = 0.0, Ti = 0.0;                       // whole blocks p-NT .. p-1
		auto add_turned = [&] (const double* a0p, const double* arp, const double* aip, int q, int behind)   // entry q, `behind` blocks behind this thread's block
			{
			const double  b0 = a0p[q], br = arp[q], bi = aip[q];
			const double2 w  = rot[behind-1];                       // exp(-j w 16 behind)
			T0 += b0;
			Tr += __builtin_fma (br, w.x, -(bi * w.y));
			Ti += __builtin_fma (br, w.y,   bi * w.x);
			};
		if (!SCAN || (K.SEG == 0))
			{ for (int d=K.NT ; d>=1 ; d--) add_turned (tot[0], tot[1], tot[2], p - d, d); }
		else
			{
			const int a = p - K.NT, b = p - 1;                      // the run of blocks
			const int sh = (K.SEG == 32)? 5 : 4;
			const int sa = a >> sh, sb = b >> sh;
			add_turned (vsf[0], vsf[1], vsf[2], a, K.NT);           // the tail of the first segment, from block a on
			for (int sg=sa+1 ; sg<sb ; sg++) add_turned (vsf[0], vsf[1], vsf[2], sg * K.SEG, p - sg * K.SEG);   // whole segments, in their first block's phase
			if (sb != sa) add_turned (tot[0], tot[1], tot[2], b, 1);     // the head of the last one, up to block b
			}
		const double* lb = lds + (p - K.DQ) * HN_PITCH;             // block of the left ends of s >= DR
		const double* la = lb - HN_PITCH + HN_G;                    // the block before it, indexed by u - DR < 0
		double s0 = 0.0, sr = 0.0, si = 0.0;
#pragma unroll
		for (int u=2*HN_G-2 ; u>=0 ; u--)
			{
			if (u >= HN_G + K.DR) continue;                         // (uniform) the walk starts at u = 15 + DR
			if (u == K.DR - 1) { T0 += s0;  Tr += sr;  Ti += si;  s0 = 0.0;  sr = 0.0;  si = 0.0; }
			const int    rel = u - K.DR;
			const double x   = (rel >= 0)? lb[rel] : la[rel];
			s0 += x;
			sr  = __builtin_fma (x, K.leftC[u], sr);
			si  = __builtin_fma (x, K.leftS[u], si);
			if (u < HN_G)
				{
				// (the first prefix sums are made again from the block's first elements -- the same operations on the same
				//  operands -- instead of kept from phase 1 to the walk's end: gdsp_hann_tile.h, phase 2)
				double q0 = P0[u], qr = Pr[u], qi = Pi[u];
				if (u < HN_RT_REDO)
					{
					const double* xb = lds + p * HN_PITCH;
					q0 = 0.0;  qr = 0.0;  qi = 0.0;
#pragma unroll
					for (int v=0 ; v<HN_RT_REDO ; v++)
						{
						if (v > u) continue;
						const double xv = xb[v];
						q0 += xv;
						qr  = __builtin_fma (xv, K.ownC[v], qr);
						qi  = __builtin_fma (xv, K.ownS[v], qi);
						}
					}
				const double z0 = (s0 + T0) + q0;
				const double zr = (sr + Tr) + qr;
				const double zi = (si + Ti) + qi;
				const double c  = __builtin_fma (K.demC[u], zr, -(K.demS[u] * zi));
				acc[u] = K.scale * ((z0 - c) + acc[u]);
				}
			}
		}
	if (direct) hann_direct_tile (lds, taps, W, K.LO, K.OUT, THREADS);
	else
		{
		__syncthreads ();

		if (live)
			{
			double* mine = lds + (p - K.HALO_L) * HN_PITCH;
#pragma unroll
			for (int u=0 ; u<HN_G ; u++) mine[u] = acc[u];
			}
		__syncthreads ();
		}

	if (out0 + K.OUT <= (int64_t) n)
		{
		double2* dst = reinterpret_cast<double2*> (out + out0);
		for (int q=p ; q<K.OUT/2 ; q+=THREADS)
			{
			const int o = 2*q;
			const double* src = lds + o + (o >> 4);
			gdsp_st2 (&dst[q], make_double2 (src[0], src[1]));
			}
		}
	else
		{
		for (int o=p ; o<K.OUT ; o+=THREADS)
			{ if (out0 + o < (int64_t) n) out[out0 + o] = lds[o + (o >> 4)]; }
		}
	}

// plans of the run-time kernel, cached per (device, W): the geometry and the rotation table in HBM
struct HannPlanRT { int device;  uint32_t W;  int E;  int threads;  bool scan;  HannRT K;  double2* d_rot; };
#define HN_PLAN_CACHE 32
static HannPlanRT hannPlans[HN_PLAN_CACHE];
static int        hannPlanLen = 0;
static std::mutex hannPlanLock;

static int hann_direct_taps (uint32_t W)                            // E: enough taps, right parity, an instantiated value
	{
	// (instantiated values only: the 256-thread form needs up to 16 at W = 1701; the long-window form takes the next of 20/21, 28/29)
	static const int small[] = { 8, 9, 12, 13, 16, 17 }, big[] = { 20, 21, 28, 29 };
	const int H    = (int) (W - 1) / 2;
	const int need = std::max (8, (int) ceil (sqrt (0.15 * W)));
	if (W <= HN_RT_SMALL_MAX_W) { for (int e : small) { if ((e >= need) && (((H - e) & 1) == 0)) return e; } }
	else          { for (int e : big)   { if ((e >= need) && (((H - e) & 1) == 0)) return e; } }
	return -1;
	}

static int hann_plan_rt (uint32_t W, HannPlanRT** out)
	{
	int device = 0;
	GDSP_HIP_TRY (hipGetDevice (&device));
	std::lock_guard<std::mutex> hold (hannPlanLock);
	for (int i=0 ; i<hannPlanLen ; i++)
		{ if ((hannPlans[i].device == device) && (hannPlans[i].W == W)) { *out = &hannPlans[i];  return GDSP_OK; } }
	GDSP_REQUIRE (hannPlanLen < HN_PLAN_CACHE, "too many different smoothing windows in one run");
	HannPlanRT* pl = &hannPlans[hannPlanLen];
	memset (pl, 0, sizeof(*pl));
	pl->device = device;  pl->W = W;
	const int E = pl->E = hann_direct_taps (W);
	// long windows: the big tile, two-level block totals.  Round 4 measured the forms side by side on one box (249 Mbp,
	// profiles/r04_hann_forms.txt): 256 threads 214 / 186 / 180 Gbases/s at 1001 / 1501 / 1701 taps against 158 / 163 / 167
	// with 768 (a tile three times as long loses less to its halo, 8 % against 24 % at 1001 taps, but one workgroup of
	// twelve waves per CU waits at its barriers with nobody to run meanwhile); a 384-thread form (6144 staged elements,
	// 70 KiB: two workgroups per CU) ran at 123 / 111 / 109.  The threshold stays where it is.
	const int THREADS = pl->threads = (W > HN_RT_SMALL_MAX_W)? HN_RT_BIG : HN_THREADS;
	const int H = (int) (W - 1) / 2, DM = (int) W - 2*E - 1, BACK = DM + E;
	HannRT& K = pl->K;
	K.DQ = DM / HN_G;  K.DR = DM % HN_G;  K.NT = K.DQ - 1;
	K.HALO_L = (BACK + HN_G - 1) / HN_G;  K.HALO_R = (HN_G - 1 + E) / HN_G;
	K.LO   = K.HALO_L * HN_G - BACK;
	K.LEAD = K.HALO_L * HN_G - (H - E);
	K.OUT  = (THREADS - K.HALO_L - K.HALO_R) * HN_G;
	// the blocks between the ends: walked one by one below HN_RT_SCAN_MIN_W (a short walk, and the kernel without the scans
	// needs no spill to keep three waves per SIMD), segmented scans above; the segments need runs of at least 32 blocks,
	// the long-window form's 64 is what it was tuned with
	pl->scan = (THREADS > HN_THREADS) || ((W >= HN_RT_SCAN_MIN_W) && (getenv ("GDSP_HANN_WALK") == NULL));   // (GDSP_HANN_WALK: the walk at any window of the 256-thread form, for A/B timing)
	K.SEG    = (pl->scan && (K.NT >= ((THREADS > HN_THREADS)? 64 : 32)))? 32 : 0;
	const double pi = 3.14159265358979323846264;
	const long   M  = (long) W + 1;
	auto cs = [&] (long m, double* c, double* sn)
		{
		long r = ((m % M) + M) % M;
		double x = r / (double) M;
		*c = cos (2*pi*x);  *sn = sin (2*pi*x);
		};
	for (int k=1 ; k<=E ; k++)            { double c, sn;  cs (k, &c, &sn);  K.edge[k-1] = 1 - c; }
	for (int u=0 ; u<HN_G ; u++)          cs (u, &K.ownC[u], &K.ownS[u]);
	for (int u=0 ; u<HN_G + K.DR ; u++)   cs ((long) u - DM, &K.leftC[u], &K.leftS[u]);
	for (int u=0 ; u<HN_G ; u++)          cs ((long) W - E - u, &K.demC[u], &K.demS[u]);
	double total = 0.0;                                            // as gdsp_hann_taps sums it (sum.c:632-645)
	for (uint32_t k=0 ; k<W ; k++)
		{
		const uint32_t kk = (k <= (uint32_t) H)? k : W-1-k;
		total += (1 - cos (2*pi*((kk+1) / (double) M))) / 2;
		}
	K.scale = 0.5 / total;
	double2 h_rot[HN_RT_MAX_NT];
	for (int d=1 ; d<=K.NT+40 && d<=HN_RT_MAX_NT ; d++) cs (-(long) HN_G * d, &h_rot[d-1].x, &h_rot[d-1].y);
	GDSP_HIP_TRY (hipMalloc ((void**) &pl->d_rot, sizeof(h_rot)));
	GDSP_HIP_TRY (hipMemcpy (pl->d_rot, h_rot, sizeof(h_rot), hipMemcpyHostToDevice));
	hannPlanLen++;
	*out = pl;
	return GDSP_OK;
	}

bool gdsp_hann_blocks_batch_available (uint32_t W) { return W == 101; }

int gdsp_hann_blocks_apply_batch (const gdsp_batch_item* items, int nitems, uint32_t W, void* stream)
	{
	GDSP_REQUIRE (gdsp_hann_blocks_batch_available (W), "no one-launch block-sum kernel for this window");
	int rc = gdsp_batch_check (items, nitems, false);
	if (rc != GDSP_OK) return rc;
	const double* d_taps = NULL;
	rc = gdsp_smooth_taps_device (W, &d_taps);
	if (rc != GDSP_OK) return rc;
	typedef HannGeom<101> G;
	HannConsts<101> K;
	hann_consts<101> (K);
	hipStream_t s = gdsp_stream (stream);
	gdsp_batch_run (items, nitems, [] (uint32_t n) { return ((uint64_t) n + G::OUT - 1) / G::OUT; },
		[&] (const GdspBatch& B, uint32_t tiles)
			{ hipLaunchKernelGGL ((hann_blocks_batch_kernel<101>), dim3(tiles), dim3(HN_THREADS), 0, s, B, K, d_taps); });
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

// 1 when GDSP_FIR_HANN has a kernel for this window
bool gdsp_hann_blocks_available (uint32_t W)
	{ return (W & 1) && (W >= 81) && (W <= 4001) && (hann_direct_taps (W) > 0); }   // (below ~80 taps the direct kernel is faster)

int gdsp_hann_blocks_apply (const double* d_in, double* d_out, uint32_t n, uint32_t W, void* stream)
	{
	GDSP_REQUIRE (gdsp_hann_blocks_available (W), "no block-sum kernel for this window");
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_in != NULL) && (d_out != NULL) && (d_in != d_out), "vectors must be distinct and non-NULL");
	GDSP_REQUIRE (gdsp_aligned16 (d_in) && gdsp_aligned16 (d_out), "vectors must be 16-byte aligned");
	hipStream_t s = gdsp_stream (stream);
	const double* d_taps = NULL;                                   // the window as data, for tiles evaluated tap by tap
	int rc0 = gdsp_smooth_taps_device (W, &d_taps);
	if (rc0 != GDSP_OK) return rc0;
	if (W == 101) { hann_launch<101> (d_in, d_out, n, d_taps, s);  GDSP_LAUNCH_CHECK ();  return GDSP_OK; }
	HannPlanRT* pl = NULL;
	int rc = hann_plan_rt (W, &pl);
	if (rc != GDSP_OK) return rc;
	GDSP_REQUIRE ((pl->K.DQ >= 2) && (pl->K.NT + 40 <= HN_RT_MAX_NT) && (pl->K.OUT >= 512) && ((pl->K.LEAD & 1) == 0), "window outside the block-sum kernel's range");
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + pl->K.OUT - 1) / pl->K.OUT);
#define HN_RT_SMALL(EE) hipLaunchKernelGGL ((hann_blocks_rt_kernel<EE, HN_THREADS, false>), dim3(ntiles), dim3(HN_THREADS), 0, s, d_in, d_out, n, ntiles, pl->K, pl->d_rot, d_taps, (int) W)
#define HN_RT_MID(EE)   hipLaunchKernelGGL ((hann_blocks_rt_kernel<EE, HN_THREADS, true>),  dim3(ntiles), dim3(HN_THREADS), 0, s, d_in, d_out, n, ntiles, pl->K, pl->d_rot, d_taps, (int) W)
#define HN_RT_BIGK(EE)  hipLaunchKernelGGL ((hann_blocks_rt_kernel<EE, HN_RT_BIG, true>),   dim3(ntiles), dim3(HN_RT_BIG),  0, s, d_in, d_out, n, ntiles, pl->K, pl->d_rot, d_taps, (int) W)
	switch (pl->E)
		{
		case 8:  HN_RT_SMALL (8);  break;
		case 9:  HN_RT_SMALL (9);  break;
		case 12: HN_RT_SMALL (12); break;
		case 13: if (pl->scan) HN_RT_MID (13); else HN_RT_SMALL (13);  break;      // (windows of 1001 taps and more need 13 or more)
		case 16: if (pl->scan) HN_RT_MID (16); else HN_RT_SMALL (16);  break;
		case 17: if (pl->scan) HN_RT_MID (17); else HN_RT_SMALL (17);  break;
		case 20: HN_RT_BIGK (20);  break;
		case 21: HN_RT_BIGK (21);  break;
		case 28: HN_RT_BIGK (28);  break;
		default: HN_RT_BIGK (29);  break;
		}
#undef HN_RT_SMALL
#undef HN_RT_MID
#undef HN_RT_BIGK
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}
