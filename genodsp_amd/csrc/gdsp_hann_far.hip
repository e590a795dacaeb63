// gdsp_hann_far.hip -- `smooth` in GDSP_FIR_HANN mode for windows longer than one LDS tile can hold
// (3201 .. 50001 taps: the reference accepts up to 50001, sum.c:478).
//
// Same evaluation as gdsp_hann.hip: the window is c*(1 - cos(w k)); the E taps at either end are applied directly,
// the W-2E middle taps are S - C with S the plain sum under them and C the real part of a phase factor times the sum
// of x[e]*exp(j w e), both built from block sums by additions only.  What changes is where the blocks between a
// window's two ends come from.  A window of 50001 taps spans twelve 4096-element tiles, so:
//   pass A  (hf_totals_kernel, reads v once)  the (sum, cos-sum, sin-sum) of every 16-element block in the block's
//           own phase, of every aligned group of 16 blocks and of every aligned 4096-element tile, to HBM
//           (1.6 B/base), and one "holds inf/NaN/huge" flag per tile;
//   pass B  (hf_output_kernel)  a workgroup of 192 threads makes 3072 outputs.  It stages the 3072+E elements that
//           hold the right ends of its windows and the 3072+E+31 that hold the left ends (a window's other W-200
//           elements are never loaded), one thread owns one block as the right end of 16 windows exactly as in
//           gdsp_hann.hip, and the whole blocks between its two ends are three pieces:
//             - the blocks of its own tile before its own one: a prefix scan of the tile's block totals,
//             - the blocks of the left region after its left-end block: a suffix scan of that region's totals,
//             - the blocks between the two regions, the same for all 192 threads: summed once per workgroup
//               from the three levels pass A left in HBM (<= 15+15+13+15+15 terms),
//           every piece turned into the thread's phase by one constant rotation.  The scans are Hillis-Steele in LDS
//           with the rotation of each step a constant of W; nothing is a difference of running sums.
// 8 + 1.6 B/base for pass A and 8 + 8 + 8 B/base for pass B: 34 B/base whatever the window, against W multiply-adds
// per base (50001 taps: 0.75 Gbases/s) for direct evaluation.
//
// Tolerance, tests and the non-shift-invariance are those of gdsp_hann.hip.  A workgroup whose windows touch a tile
// that holds an infinity, a NaN or a magnitude >= 2^1017 writes nothing and marks itself; hf_direct_kernel then
// evaluates those outputs tap by tap (ascending fused multiply-adds with the taps as data = GDSP_FIR_FMA's bits).

#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <vector>
#include "gdsp_common.h"

#define HF_G        16
#define HF_PITCH    17
#define HF_TB       192                                  // threads = right-end blocks per workgroup of pass B
#define HF_OUT      (HF_TB * HF_G)
#define HF_EMAX     96
#define HF_HUGE_HI  0x7F800000u

struct HfConsts
	{
	int    W, E, H, DM, DQ, DR, BACK, LO;                // LO: offset in the left region of the first input of output s = 0, thread 0
	int    LB;                                           // blocks of the left region
	int    RN;                                           // elements of the right region
	double scale;
	double edge[HF_EMAX];                                // 1 - cos(w k), k = 1..E
	double ownC[HF_G], ownS[HF_G];                       // exp(+j w u)
	double leftC[2*HF_G], leftS[2*HF_G];                 // exp(+j w (u - DM)), u = 0..15+DR
	double demC[HF_G], demS[HF_G];                       // exp(+j w (W-E - s))
	double stepC[8], stepS[8];                           // exp(-j w 16 2^k): a step of the scans
	double backC, backS;                                 // exp(-j w 16 (DQ-1)): the left region's suffix into the owner's phase
	};

struct HfLevels                                          // pass A's output: three levels of block totals, structure of arrays
	{
	double *b0, *br, *bi;                                // per 16-element block, in the block's phase
	double *g0, *gr, *gi;                                // per 256 elements, in the group's first block's phase
	double *t0, *tr, *ti;                                // per 4096 elements, in the tile's first block's phase
	unsigned int* huge;                                  // per 4096-element tile
	long long nblocks;                                   // blocks covered (a whole number of tiles)
	};

__device__ __forceinline__ uint32_t hf_magnitude_hi (double x)
	{ return ((uint32_t) (__double_as_longlong (x) >> 32)) & 0x7FFFFFFFu; }

// ---------------------------------------------------------------- pass A ----
struct HfTotalsConsts { double ownC[HF_G], ownS[HF_G], blkC[16], blkS[16], grpC[16], grpS[16]; };   // exp(+j w u), exp(+j w 16 j), exp(+j w 256 g)

__global__ __launch_bounds__(256)
void hf_totals_kernel (const double* __restrict__ in, uint32_t n, HfLevels Lv, HfTotalsConsts K)
	{
	__shared__ __attribute__((aligned(16))) double lds[256 * HF_PITCH];
	__shared__ double tot[3][256], grp[3][16];
	__shared__ uint32_t hugeWave[4];
	const int     p  = threadIdx.x;
	const int64_t e0 = (int64_t) blockIdx.x * 4096;
	if (e0 + 4096 <= (int64_t) n)
		{
		const double2* src = reinterpret_cast<const double2*> (in + e0);
		double2 r[8];
#pragma unroll
		for (int u=0 ; u<8 ; u++) r[u] = gdsp_ld2 (&src[u*256 + p]);
#pragma unroll
		for (int u=0 ; u<8 ; u++)
			{ const int e = 2 * (u*256 + p);  double* d = lds + e + (e >> 4);  d[0] = r[u].x;  d[1] = r[u].y; }
		}
	else
		{
		for (int e=p ; e<4096 ; e+=256) lds[e + (e >> 4)] = (e0 + e < (int64_t) n)? in[e0 + e] : 0.0;
		}
	__syncthreads ();
	const double* xb = lds + p * HF_PITCH;
	double a0 = 0.0, ar = 0.0, ai = 0.0;
	uint32_t big = 0;
#pragma unroll
	for (int u=0 ; u<HF_G ; u++)
		{
		const double x = xb[u];
		big = max (big, hf_magnitude_hi (x));
		a0 += x;
		ar  = __builtin_fma (x, K.ownC[u], ar);
		ai  = __builtin_fma (x, K.ownS[u], ai);
		}
	const long long blk = (long long) blockIdx.x * 256 + p;
	Lv.b0[blk] = a0;  Lv.br[blk] = ar;  Lv.bi[blk] = ai;
	tot[0][p] = a0;  tot[1][p] = ar;  tot[2][p] = ai;
	const bool any = (__builtin_amdgcn_ballot_w64 (big >= HF_HUGE_HI) != 0);
	if ((p & 63) == 0) hugeWave[p >> 6] = any? 1u : 0u;
	__syncthreads ();
	if (p < 16)                                            // the group's 16 blocks, each turned forward into the first one's phase
		{
		double g0 = 0.0, gr = 0.0, gi = 0.0;
		for (int j=0 ; j<16 ; j++)
			{
			const double b0 = tot[0][16*p+j], br = tot[1][16*p+j], bi = tot[2][16*p+j];
			g0 += b0;
			gr += __builtin_fma (br, K.blkC[j], -(bi * K.blkS[j]));
			gi += __builtin_fma (br, K.blkS[j],   bi * K.blkC[j]);
			}
		grp[0][p] = g0;  grp[1][p] = gr;  grp[2][p] = gi;
		const long long g = (long long) blockIdx.x * 16 + p;
		Lv.g0[g] = g0;  Lv.gr[g] = gr;  Lv.gi[g] = gi;
		}
	__syncthreads ();
	if (p == 0)
		{
		double t0 = 0.0, tr = 0.0, ti = 0.0;
		for (int g=0 ; g<16 ; g++)
			{
			const double b0 = grp[0][g], br = grp[1][g], bi = grp[2][g];
			t0 += b0;
			tr += __builtin_fma (br, K.grpC[g], -(bi * K.grpS[g]));
			ti += __builtin_fma (br, K.grpS[g],   bi * K.grpC[g]);
			}
		Lv.t0[blockIdx.x] = t0;  Lv.tr[blockIdx.x] = tr;  Lv.ti[blockIdx.x] = ti;
		Lv.huge[blockIdx.x] = hugeWave[0] | hugeWave[1] | hugeWave[2] | hugeWave[3];
		}
	}

// ---------------------------------------------------------------- pass B ----
// element e of a staged region sits at e + (e >> 4)
__device__ __forceinline__ void hf_stage (double* lds, const double* __restrict__ in, uint32_t n, int64_t g0, int count)
	{
	if ((g0 >= 0) && (g0 + count <= (int64_t) n) && ((g0 & 1) == 0) && ((count & 1) == 0))
		{
		const double2* src = reinterpret_cast<const double2*> (in + g0);
		const int np = count / 2;
		for (int base=0 ; base<np ; base+=8*HF_TB)
			{
			double2 r[8];
#pragma unroll
			for (int u=0 ; u<8 ; u++) { const int q = base + u*HF_TB + (int) threadIdx.x;  r[u] = src[(q < np)? q : np-1]; }
#pragma unroll
			for (int u=0 ; u<8 ; u++)
				{
				const int q = base + u*HF_TB + (int) threadIdx.x;
				if (q < np) { const int e = 2*q;  lds[e + (e >> 4)] = r[u].x;  lds[(e+1) + ((e+1) >> 4)] = r[u].y; }
				}
			}
		}
	else
		{
		for (int e=threadIdx.x ; e<count ; e+=HF_TB)
			{
			const int64_t g = g0 + e;
			lds[e + (e >> 4)] = ((g >= 0) && (g < (int64_t) n))? in[g] : 0.0;
			}
		}
	}

__global__ __launch_bounds__(HF_TB)
void hf_output_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, HfConsts K, HfLevels Lv,
                       const double2* __restrict__ rotFar, unsigned int* __restrict__ tileFlag)
	{
	constexpr int RNMAX = HF_OUT + ((HF_EMAX + 2 + 1) & ~1);        // right region: the tile and E more
	constexpr int LBMAX = HF_TB + (HF_EMAX + 46) / 16;              // left region, in blocks
	const int E  = K.E;
	const int RN = K.RN;
	__shared__ __attribute__((aligned(16))) double Rg[RNMAX + RNMAX/16 + 2];
	__shared__ __attribute__((aligned(16))) double Lg[LBMAX * HF_PITCH + 2];
	__shared__ double totR[3][256], totL[3][256];
	__shared__ double gapT[3];

	const int       p    = threadIdx.x;
	const long long tb0  = (long long) blockIdx.x * HF_TB;          // first right-end block of this tile
	const int64_t   rb   = 16 * tb0;                                // its first element
	// first block of the left region (the block grid is global: floor division also below zero)
	const int64_t   lneed = rb - K.DM - K.E;
	const long long lb0  = (lneed >= 0)? lneed / 16 : -((-lneed + 15) / 16);
	const int64_t   lfirst = 16 * lb0;
	const int       LB   = K.LB;

	// ---- a tile whose windows touch inf / NaN / a huge magnitude is left to hf_direct_kernel
		{
		int64_t lo = (lfirst < 0)? 0 : lfirst, hi = rb + HF_OUT + K.E;
		const long long ntiles = Lv.nblocks / 256;
		long long t0 = lo >> 12, t1 = hi >> 12;
		if (t1 >= ntiles) t1 = ntiles - 1;
		unsigned int flagged = 0;
		for (long long t=t0 ; t<=t1 ; t++) flagged |= Lv.huge[t];
		if (flagged != 0) { if (p == 0) tileFlag[blockIdx.x] = 1;  return; }
		}

	hf_stage (Rg, in, n, rb, RN);
	hf_stage (Lg, in, n, lfirst, LB * 16);
	__syncthreads ();

	// ---- phase 0: the E taps at either end of each of the 16 windows, directly
	double acc[HF_G];
#pragma unroll
	for (int s=0 ; s<HF_G ; s++) acc[s] = 0.0;
	// (sixteen taps at a time over a sliding window of 31 staged values: a loop, not E x 16 unrolled multiply-adds --
	//  with 88 taps the unrolled form did not fit the scalar registers and ran five times slower)
		{
#pragma unroll 1
		for (int t0=0 ; t0<E ; t0+=HF_G)                            // left end: window s meets input s+t under tap t+1
			{
			double x[2*HF_G - 1];
#pragma unroll
			for (int i=0 ; i<2*HF_G-1 ; i++) { const int o = 16*p + K.LO + t0 + i;  x[i] = Lg[o + (o >> 4)]; }
#pragma unroll
			for (int tt=0 ; tt<HF_G ; tt++)
				{
				if (t0 + tt < E)                                    // (uniform)
					{
					const double w = K.edge[t0 + tt];
#pragma unroll
					for (int s=0 ; s<HF_G ; s++) acc[s] = __builtin_fma (w, x[s + tt], acc[s]);
					}
				}
			}
#pragma unroll 1
		for (int m0=1 ; m0<=E ; m0+=HF_G)                           // right end: window s meets input s+m under tap W+1-m = tap m from the far end
			{
			double x[2*HF_G - 1];
#pragma unroll
			for (int i=0 ; i<2*HF_G-1 ; i++) { const int o = 16*p + m0 + i;  x[i] = Rg[o + (o >> 4)]; }
#pragma unroll
			for (int mm=0 ; mm<HF_G ; mm++)
				{
				if (m0 + mm <= E)
					{
					const double w = K.edge[E - (m0 + mm)];
#pragma unroll
					for (int s=0 ; s<HF_G ; s++) acc[s] = __builtin_fma (w, x[s + mm], acc[s]);
					}
				}
			}
		}

	// ---- phase 1: prefix sums of the own block in the own phase; block totals of both regions (in registers: thread p holds
	//      right-region block p, left-region block p and, for p < 16, left-region block 192+p)
	double P0[HF_G], Pr[HF_G], Pi[HF_G];
	double myR0, myRr, myRi;
		{
		const double* xb = Rg + p * HF_PITCH;
		double a0 = 0.0, ar = 0.0, ai = 0.0;
#pragma unroll
		for (int u=0 ; u<HF_G ; u++)
			{
			const double x = xb[u];
			a0 += x;
			ar  = __builtin_fma (x, K.ownC[u], ar);
			ai  = __builtin_fma (x, K.ownS[u], ai);
			P0[u] = a0;  Pr[u] = ar;  Pi[u] = ai;
			}
		myR0 = a0;  myRr = ar;  myRi = ai;
		}
	// the suffix of the left region stops where the own tile begins (those blocks are the prefix's)
	const long long reach = tb0 - lb0;                              // left-region blocks before the own tile
	const int       Ldom  = (reach < (long long) LB)? (int) reach : LB;
	double myL0 = 0.0, myLr = 0.0, myLi = 0.0, tail0 = 0.0, tailr = 0.0, taili = 0.0;
	for (int h=0 ; h<2 ; h++)
		{
		const int q = p + h*HF_TB;
		double a0 = 0.0, ar = 0.0, ai = 0.0;
		if ((q < Ldom) && ((h == 0) || (p < 16)))
			{
			const double* xb = Lg + q * HF_PITCH;
#pragma unroll
			for (int u=0 ; u<HF_G ; u++)
				{
				const double x = xb[u];
				a0 += x;
				ar  = __builtin_fma (x, K.ownC[u], ar);
				ai  = __builtin_fma (x, K.ownS[u], ai);
				}
			}
		if (h == 0) { myL0 = a0;  myLr = ar;  myLi = ai; }
		else        { tail0 = a0;  tailr = ar;  taili = ai; }
		}
	// the blocks between the two regions, the same for every thread, in the phase of the tile's first block: single
	// blocks up to a group boundary, groups up to a tile boundary, whole tiles, groups, single blocks -- laid out in
	// closed form so that lane i of wave 0 fetches term i (no list is built)
	if (p < 64)
		{
		long long g0 = lb0 + LB, g1 = tb0;                          // blocks [g0, g1)
		if (g0 < 0)  g0 = 0;
		if (g1 < g0) g1 = g0;
		long long s1 = (g0 + 15) & ~15LL;    if (s1 > g1) s1 = g1;  // single blocks [g0, s1)
		long long e1 = g1 & ~15LL;           if (e1 < s1) e1 = s1;  // ... and [e1, g1)
		long long s2 = (s1 + 255) & ~255LL;  if (s2 > e1) s2 = e1;  // groups [s1, s2)
		long long e2 = e1 & ~255LL;          if (e2 < s2) e2 = s2;  // ... and [e2, e1); tiles [s2, e2)
		const int nA = (int) (s1 - g0), nB = (int) ((s2 - s1) >> 4), nC = (int) ((e2 - s2) >> 8), nD = (int) ((e1 - e2) >> 4), nE = (int) (g1 - e1);
		const int m  = nA + nB + nC + nD + nE;
		double v0 = 0.0, vr = 0.0, vi = 0.0;
		for (int base=0 ; base<m ; base+=64)
			{
			double c0 = 0.0, cr = 0.0, ci = 0.0;
			const int i = base + p;
			if (i < m)
				{
				double b0, br, bi;
				long long origin;                                       // the entry's first block
				if (i < nA)                          { origin = g0 + i;                               b0 = Lv.b0[origin];       br = Lv.br[origin];       bi = Lv.bi[origin]; }
				else if (i < nA + nB)                { origin = s1 + 16LL * (i - nA);                 b0 = Lv.g0[origin >> 4];  br = Lv.gr[origin >> 4];  bi = Lv.gi[origin >> 4]; }
				else if (i < nA + nB + nC)           { origin = s2 + 256LL * (i - nA - nB);           b0 = Lv.t0[origin >> 8];  br = Lv.tr[origin >> 8];  bi = Lv.ti[origin >> 8]; }
				else if (i < nA + nB + nC + nD)      { origin = e2 + 16LL * (i - nA - nB - nC);       b0 = Lv.g0[origin >> 4];  br = Lv.gr[origin >> 4];  bi = Lv.gi[origin >> 4]; }
				else                                 { origin = e1 + (i - nA - nB - nC - nD);         b0 = Lv.b0[origin];       br = Lv.br[origin];       bi = Lv.bi[origin]; }
				const double2 w = rotFar[tb0 - origin];                 // exp(-j w 16 d): d blocks behind the tile's first block
				c0 = b0;
				cr = __builtin_fma (br, w.x, -(bi * w.y));
				ci = __builtin_fma (br, w.y,   bi * w.x);
				}
			for (int k=0 ; k<64 ; k++)                                  // fixed order
				{
				v0 += __shfl (c0, k, 64);  vr += __shfl (cr, k, 64);  vi += __shfl (ci, k, 64);
				}
			}
		if (p == 0) { gapT[0] = v0;  gapT[1] = vr;  gapT[2] = vi; }
		}

	// ---- the two scans, by wave shuffles.  prefix: X(p) = sum_{q<p} R(p-q) totR[q] (thread p keeps it);
	//      suffix: V(j) = sum_{q>=j} Rc(q-j) totL[q], to LDS (thread p needs V of another block).  R(d) = exp(-j w 16 d)
	//      turns a block d behind into this one's phase, Rc its conjugate; each step's rotation is a constant of W.
	const int lane = p & 63, wave = p >> 6;
	double X0 = 0.0, Xr = 0.0, Xi = 0.0;
		{
		double i0 = myR0, ir = myRr, ii = myRi;                     // inclusive, own phase, within the wave
		double v0 = myL0, vr = myLr, vi = myLi;
		double t0 = tail0, tr = tailr, ti = taili;                  // left-region blocks 192 .. 207 (lanes 0..15 of wave 0)
#pragma unroll
		for (int k=0 ; k<6 ; k++)
			{
			const int    d  = 1 << k;
			const double cC = K.stepC[k], cS = K.stepS[k];
			const double u0 = __shfl_up (i0, d, 64), ur = __shfl_up (ir, d, 64), ui = __shfl_up (ii, d, 64);
			if (lane >= d) { i0 += u0;  ir += __builtin_fma (ur, cC, -(ui * cS));  ii += __builtin_fma (ur, cS, ui * cC); }
			const double d0 = __shfl_down (v0, d, 64), dr = __shfl_down (vr, d, 64), di = __shfl_down (vi, d, 64);
			if (lane + d < 64) { v0 += d0;  vr += __builtin_fma (dr, cC, di * cS);  vi += __builtin_fma (di, cC, -(dr * cS)); }
			if (k < 4)
				{
				const double e0 = __shfl_down (t0, d, 64), er = __shfl_down (tr, d, 64), ei = __shfl_down (ti, d, 64);
				if (lane + d < 16) { t0 += e0;  tr += __builtin_fma (er, cC, ei * cS);  ti += __builtin_fma (ei, cC, -(er * cS)); }
				}
			}
		// exclusive prefix inside the wave: the block before, one step behind
		const double b0 = __shfl_up (i0, 1, 64), br = __shfl_up (ir, 1, 64), bi = __shfl_up (ii, 1, 64);
		if (lane >= 1) { X0 = b0;  Xr = __builtin_fma (br, K.stepC[0], -(bi * K.stepS[0]));  Xi = __builtin_fma (br, K.stepS[0], bi * K.stepC[0]); }
		if (lane == 63) { totR[0][wave] = i0;  totR[1][wave] = ir;  totR[2][wave] = ii; }      // the wave's blocks, in its last block's phase
		if (lane == 0)  { totR[0][8 + wave] = v0;  totR[1][8 + wave] = vr;  totR[2][8 + wave] = vi; }   // ... and in its first block's
		if ((wave == 0) && (lane < 16)) { totL[0][HF_TB + lane] = t0;  totL[1][HF_TB + lane] = tr;  totL[2][HF_TB + lane] = ti; }
		__syncthreads ();
		for (int w=0 ; w<wave ; w++)                               // earlier waves: their last block is p - (64 w + 63) behind
			{
			const double2 r = rotFar[p - (64*w + 63)];
			const double  c0 = totR[0][w], cr = totR[1][w], ci = totR[2][w];
			X0 += c0;  Xr += __builtin_fma (cr, r.x, -(ci * r.y));  Xi += __builtin_fma (cr, r.y, ci * r.x);
			}
		for (int w=wave+1 ; w<=HF_TB/64 ; w++)                     // later waves and the tail: their first block is 64 w - p ahead
			{
			const double2 r = rotFar[64*w - p];
			double c0, cr, ci;
			if (w < HF_TB/64) { c0 = totR[0][8 + w];  cr = totR[1][8 + w];  ci = totR[2][8 + w]; }
			else              { c0 = totL[0][HF_TB];  cr = totL[1][HF_TB];  ci = totL[2][HF_TB]; }
			v0 += c0;  vr += __builtin_fma (cr, r.x, ci * r.y);  vi += __builtin_fma (ci, r.x, -(cr * r.y));
			}
		totL[0][p] = v0;  totL[1][p] = vr;  totL[2][p] = vi;
		}
	__syncthreads ();

	// ---- phase 2: the middle stretch of one window per left end
		{
		// whole blocks A+1 .. own-1 in the own phase: the prefix of the own tile, the gap, the suffix of the left region
		const int AL = (int) (tb0 - K.DQ - lb0) + p;               // the left-end block of outputs s >= DR, in the left region
		double T0 = X0, Tr = Xr, Ti = Xi;
			{
			const double2 w = rotFar[p];                            // the gap is in the phase of the tile's first block: p blocks behind
			const double  b0 = gapT[0], br = gapT[1], bi = gapT[2];
			T0 += b0;
			Tr += __builtin_fma (br, w.x, -(bi * w.y));
			Ti += __builtin_fma (br, w.y,   bi * w.x);
			}
		if (AL + 1 < 256)
			{
			const double b0 = totL[0][AL+1], br = totL[1][AL+1], bi = totL[2][AL+1];
			T0 += b0;
			Tr += __builtin_fma (br, K.backC, -(bi * K.backS));     // block A+1 is DQ-1 blocks behind the own one
			Ti += __builtin_fma (br, K.backS,   bi * K.backC);
			}
		const int base = 16 * AL - K.DR;                            // element of the left region under u = 0
		double s0 = 0.0, sr = 0.0, si = 0.0;
#pragma unroll
		for (int u=2*HF_G-2 ; u>=0 ; u--)
			{
			if (u >= HF_G + K.DR) continue;                         // (uniform) the walk starts at u = 15 + DR
			if (u == K.DR - 1) { T0 += s0;  Tr += sr;  Ti += si;  s0 = 0.0;  sr = 0.0;  si = 0.0; }
			const int    e = base + u;
			const double x = Lg[e + (e >> 4)];
			s0 += x;
			sr  = __builtin_fma (x, K.leftC[u], sr);
			si  = __builtin_fma (x, K.leftS[u], si);
			if (u < HF_G)
				{
				const double z0 = (s0 + T0) + P0[u];
				const double zr = (sr + Tr) + Pr[u];
				const double zi = (si + Ti) + Pi[u];
				const double c  = __builtin_fma (K.demC[u], zr, -(K.demS[u] * zi));
				acc[u] = K.scale * ((z0 - c) + acc[u]);
				}
			}
		}
	__syncthreads ();                                              // every read of the staged inputs is done

	// ---- results back through LDS: output o of the tile belongs to thread o/16
		{
		double* mine = Rg + p * HF_PITCH;
#pragma unroll
		for (int u=0 ; u<HF_G ; u++) mine[u] = acc[u];
		}
	__syncthreads ();
	const int64_t out0 = rb - (K.H - K.E);                          // output index of the tile's first result (even)
	if ((out0 >= 0) && (out0 + HF_OUT <= (int64_t) n))
		{
		double2* dst = reinterpret_cast<double2*> (out + out0);
		for (int q=p ; q<HF_OUT/2 ; q+=HF_TB)
			{
			const int o = 2*q;
			const double* src = Rg + o + (o >> 4);
			gdsp_st2 (&dst[q], make_double2 (src[0], src[1]));
			}
		}
	else
		{
		for (int o=p ; o<HF_OUT ; o+=HF_TB)
			{
			const int64_t i = out0 + o;
			if ((i >= 0) && (i < (int64_t) n)) out[i] = Rg[o + (o >> 4)];
			}
		}
	}

// the outputs of marked tiles, tap by tap (ascending fused multiply-adds: GDSP_FIR_FMA's bits); rare, not tuned
__global__ __launch_bounds__(HF_TB)
void hf_direct_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, int W, int H, int E,
                       const double* __restrict__ taps, const unsigned int* __restrict__ tileFlag)
	{
	if (tileFlag[blockIdx.x] == 0) return;
	const int64_t out0 = 16 * (int64_t) blockIdx.x * HF_TB - (H - E);
	double acc[HF_G];
#pragma unroll
	for (int m=0 ; m<HF_G ; m++) acc[m] = 0.0;
	for (int k=0 ; k<W ; k++)
		{
		const double w = taps[k];
#pragma unroll
		for (int m=0 ; m<HF_G ; m++)
			{
			const int64_t g = out0 + threadIdx.x + HF_TB*m - H + k;
			const double  x = ((g >= 0) && (g < (int64_t) n))? in[g] : 0.0;
			acc[m] = __builtin_fma (w, x, acc[m]);
			}
		}
#pragma unroll
	for (int m=0 ; m<HF_G ; m++)
		{
		const int64_t i = out0 + threadIdx.x + HF_TB*m;
		if ((i >= 0) && (i < (int64_t) n)) out[i] = acc[m];
		}
	}

// ------------------------------------------------------------------ host ----
struct HfPlan { int device;  uint32_t W;  HfConsts K;  HfTotalsConsts A;  double2* d_rotFar; };
struct HfWork { int device;  void* stream;  size_t nblocks;  double* d_levels;  unsigned int* d_flags;  size_t nflags; };
static std::vector<HfPlan*> hfPlans;
static std::vector<HfWork>  hfWork;
static std::mutex           hfLock;

static int hf_direct_taps (uint32_t W)                              // E: enough taps, and the parity that keeps a tile's first output even
	{
	const int H    = (int) (W - 1) / 2;
	const int need = (int) ceil (sqrt (0.15 * W));
	const int e    = need + (((H - need) & 1)? 1 : 0);
	return (e <= HF_EMAX)? e : -1;
	}

bool gdsp_hann_far_available (uint32_t W)
	{ return (W & 1) && (W >= 3201) && (W <= 50001) && (hf_direct_taps (W) > 0); }

static int hf_plan (uint32_t W, HfPlan** out)
	{
	int device = 0;
	GDSP_HIP_TRY (hipGetDevice (&device));
	for (HfPlan* pl : hfPlans) { if ((pl->device == device) && (pl->W == W)) { *out = pl;  return GDSP_OK; } }
	HfPlan* pl = new HfPlan;
	memset (pl, 0, sizeof(*pl));
	pl->device = device;  pl->W = W;
	HfConsts& K = pl->K;
	K.W = (int) W;  K.E = hf_direct_taps (W);  K.H = (int) (W - 1) / 2;
	K.DM = (int) W - 2*K.E - 1;  K.DQ = K.DM / HF_G;  K.DR = K.DM % HF_G;  K.BACK = K.DM + K.E;
	// left region: first block = floor((rb - DM - E)/16) with rb a multiple of 16
	const int lneed = -(K.DM + K.E);
	const int lb    = -((-lneed + 15) / 16);                        // floor division
	K.LO = -K.BACK - 16*lb;                                         // 0 .. 15
	const int aMax = HF_OUT - 1 - K.DM;                             // last left end, relative to rb
	const int lastBlock = (aMax >= 0)? aMax / 16 : -((-aMax + 15) / 16);
	K.LB = lastBlock - lb + 1;
	K.RN = HF_OUT + ((K.E + 2 + 1) & ~1);
	const double pi = 3.14159265358979323846264;
	const long   M  = (long) W + 1;
	auto cs = [&] (long m, double* c, double* sn)
		{
		long r = ((m % M) + M) % M;
		double x = r / (double) M;
		*c = cos (2*pi*x);  *sn = sin (2*pi*x);
		};
	for (int k=1 ; k<=K.E ; k++)          { double c, sn;  cs (k, &c, &sn);  K.edge[k-1] = 1 - c; }
	for (int u=0 ; u<HF_G ; u++)          cs (u, &K.ownC[u], &K.ownS[u]);
	for (int u=0 ; u<HF_G + K.DR ; u++)   cs ((long) u - K.DM, &K.leftC[u], &K.leftS[u]);
	for (int u=0 ; u<HF_G ; u++)          cs ((long) W - K.E - u, &K.demC[u], &K.demS[u]);
	for (int k=0 ; k<8 ; k++)             cs (-(long) HF_G * (1L << k), &K.stepC[k], &K.stepS[k]);
	cs (-(long) HF_G * (K.DQ - 1), &K.backC, &K.backS);
	double total = 0.0;                                            // as gdsp_hann_taps sums it (sum.c:632-645)
	for (uint32_t k=0 ; k<W ; k++)
		{
		const uint32_t kk = (k <= (uint32_t) K.H)? k : W-1-k;
		total += (1 - cos (2*pi*((kk+1) / (double) M))) / 2;
		}
	K.scale = 0.5 / total;
	for (int u=0 ; u<HF_G ; u++) { pl->A.ownC[u] = K.ownC[u];  pl->A.ownS[u] = K.ownS[u]; }
	for (int j=0 ; j<16 ; j++)   { cs ((long) HF_G * j, &pl->A.blkC[j], &pl->A.blkS[j]);  cs ((long) 256 * j, &pl->A.grpC[j], &pl->A.grpS[j]); }
	const int nrot = K.DQ + K.LB + 320;                             // exp(-j w 16 d) for every distance pass B looks up
	std::vector<double2> rot (nrot);
	for (int d=0 ; d<nrot ; d++) cs (-(long) HF_G * d, &rot[d].x, &rot[d].y);
	GDSP_HIP_TRY (hipMalloc ((void**) &pl->d_rotFar, nrot * sizeof(double2)));
	GDSP_HIP_TRY (hipMemcpy (pl->d_rotFar, rot.data (), nrot * sizeof(double2), hipMemcpyHostToDevice));
	hfPlans.push_back (pl);
	*out = pl;
	return GDSP_OK;
	}

// level arrays and tile flags, kept per (device, stream): calls on one stream follow one another, calls on different
// streams must not share them
static int hf_work (void* stream, size_t nblocks, size_t nflags, HfWork** out)
	{
	int device = 0;
	GDSP_HIP_TRY (hipGetDevice (&device));
	HfWork* w = NULL;
	for (HfWork& x : hfWork) { if ((x.device == device) && (x.stream == stream)) w = &x; }
	if (w == NULL) { hfWork.push_back (HfWork { device, stream, 0, NULL, NULL, 0 });  w = &hfWork.back (); }
	if (w->nblocks < nblocks)
		{
		if (w->d_levels != NULL) { GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (stream)));  GDSP_HIP_TRY (hipFree (w->d_levels)); }
		w->d_levels = NULL;  w->nblocks = 0;
		// 3 arrays per level, blocks + groups + tiles, then one flag word per tile
		const size_t doubles = 3 * (nblocks + nblocks/16 + nblocks/256) + 64;
		GDSP_HIP_TRY (hipMalloc ((void**) &w->d_levels, doubles * sizeof(double) + (nblocks/256 + 16) * sizeof(unsigned int)));
		w->nblocks = nblocks;
		}
	if (w->nflags < nflags)
		{
		if (w->d_flags != NULL) { GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (stream)));  GDSP_HIP_TRY (hipFree (w->d_flags)); }
		w->d_flags = NULL;  w->nflags = 0;
		GDSP_HIP_TRY (hipMalloc ((void**) &w->d_flags, (nflags + 16) * sizeof(unsigned int)));
		w->nflags = nflags;
		}
	*out = w;
	return GDSP_OK;
	}

int gdsp_hann_far_apply (const double* d_in, double* d_out, uint32_t n, uint32_t W, void* stream)
	{
	GDSP_REQUIRE (gdsp_hann_far_available (W), "no far block-sum kernel for this window");
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_in != NULL) && (d_out != NULL) && (d_in != d_out), "vectors must be distinct and non-NULL");
	GDSP_REQUIRE (gdsp_aligned16 (d_in) && gdsp_aligned16 (d_out), "vectors must be 16-byte aligned");
	const double* d_taps = NULL;
	int rc = gdsp_smooth_taps_device (W, &d_taps);
	if (rc != GDSP_OK) return rc;
	std::lock_guard<std::mutex> hold (hfLock);
	HfPlan* pl = NULL;
	rc = hf_plan (W, &pl);
	if (rc != GDSP_OK) return rc;
	const HfConsts& K = pl->K;
	GDSP_REQUIRE ((K.DQ >= HF_TB + 8) && (K.E <= HF_EMAX) && (K.LB <= 256) && (K.DR + HF_G <= 2*HF_G), "window outside the far kernel's range");
	// outputs i = b' - (H-E), b' in tiles of 3072: the last output n-1 sits at b' = n-1 + H-E
	const uint32_t ntilesB = (uint32_t) (((uint64_t) n + (K.H - K.E) + HF_OUT - 1) / HF_OUT);
	// pass A covers every block a window can touch: up to the last right end plus E
	const size_t ntilesA = ((size_t) ntilesB * HF_OUT + K.E + 4096 + 4095) / 4096;
	const size_t nblocks = ntilesA * 256;
	HfWork* wk = NULL;
	rc = hf_work (stream, nblocks, ntilesB, &wk);
	if (rc != GDSP_OK) return rc;
	HfLevels Lv;
	double* base = wk->d_levels;
	Lv.b0 = base;                    Lv.br = Lv.b0 + wk->nblocks;       Lv.bi = Lv.br + wk->nblocks;
	Lv.g0 = Lv.bi + wk->nblocks;     Lv.gr = Lv.g0 + wk->nblocks/16;    Lv.gi = Lv.gr + wk->nblocks/16;
	Lv.t0 = Lv.gi + wk->nblocks/16;  Lv.tr = Lv.t0 + wk->nblocks/256;   Lv.ti = Lv.tr + wk->nblocks/256;
	Lv.huge = (unsigned int*) (Lv.ti + wk->nblocks/256 + 8);
	Lv.nblocks = (long long) nblocks;
	hipStream_t s = gdsp_stream (stream);
	GDSP_HIP_TRY (hipMemsetAsync (wk->d_flags, 0, (ntilesB + 16) * sizeof(unsigned int), s));
	hipLaunchKernelGGL (hf_totals_kernel, dim3((uint32_t) ntilesA), dim3(256), 0, s, d_in, n, Lv, pl->A);
	hipLaunchKernelGGL (hf_output_kernel, dim3(ntilesB), dim3(HF_TB), 0, s, d_in, d_out, n, K, Lv, pl->d_rotFar, wk->d_flags);
	hipLaunchKernelGGL (hf_direct_kernel, dim3(ntilesB), dim3(HF_TB), 0, s, d_in, d_out, n, (int) W, K.H, K.E, d_taps, wk->d_flags);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}
