// gdsp_hann_tile.h -- the block-sum evaluation of the Hann window for one tile, shared by gdsp_hann.hip (`smooth
// --smooth=hann`) and gdsp_peaks.hip (the filter in front of `smooth = localmax|localmin` in the reference's
// arithmetic).  See the head of gdsp_hann.hip for the method.
#pragma once
#include <math.h>
#include <float.h>
#include "gdsp_common.h"

#define HN_THREADS 256
#define HN_G       16
#define HN_PITCH   17
#define HN_ELEMS   (HN_THREADS * HN_G)
#define HN_E       8                                     // direct taps at either end of the window

// exponent field of 2^1017 = DBL_MAX/128 in the high word of a double (sign stripped): at or above it the
// unweighted block sums could overflow; infinities and NaNs (exponent 0x7FF) are above it too
#define HN_HUGE_HI 0x7F800000u

__device__ __forceinline__ uint32_t hann_magnitude_hi (double x)
	{ return ((uint32_t) (__double_as_longlong (x) >> 32)) & 0x7FFFFFFFu; }

// A tile the block sums must not touch: out[o] = sum_k taps[k] * x[first + o + k], ascending k, one fused
// multiply-add per tap (what fir_*_kernel<.., FMA> does).  Output o = p + 256*i; results go back into the
// LDS image like the block-sum results.  Rare (a tile holding inf / NaN / |x| >= 2^1017), so not tuned.
__attribute__((unused)) static __device__ __noinline__ void hann_direct_tile (double* lds, const double* __restrict__ taps, int W, int first, int nout,
                                               int nthreads = HN_THREADS)
	{
	const int p = threadIdx.x;
	const int lastElem = nthreads * HN_G - 1;
	double acc[HN_G];
#pragma unroll
	for (int i=0 ; i<HN_G ; i++) acc[i] = 0.0;
	for (int k=0 ; k<W ; k++)
		{
		const double w = taps[k];
#pragma unroll
		for (int i=0 ; i<HN_G ; i++)
			{
			int e = first + p + nthreads*i + k;
			if (e > lastElem) e = lastElem;                        // (outputs past nout are computed and dropped)
			acc[i] = __builtin_fma (w, lds[e + (e >> 4)], acc[i]);
			}
		}
	__syncthreads ();                                              // every read of the staged inputs is done
#pragma unroll
	for (int i=0 ; i<HN_G ; i++)
		{
		const int o = p + nthreads*i;
		if (o < nout) lds[o + (o >> 4)] = acc[i];
		}
	__syncthreads ();
	}

// E direct taps at either end (HN_E for the smoothed track itself: the cancellation in S - C where the taps are small must
// stay inside W 2^-52 sum|w x| per window; 0 for the peaks filter, gdsp_peaks.hip, which only needs every value to a few
// parts in 2^40 of the tile's largest input and saves the 256 multiply-adds per thread the direct taps cost)
template <int W, int EE = HN_E> struct HannGeom
	{
	static constexpr int H      = (W - 1) / 2;
	static constexpr int E      = EE;
	static constexpr int DM     = W - 2*E - 1;           // a' = b' - DM: ends of the middle stretch
	static constexpr int DQ     = DM / HN_G, DR = DM % HN_G;
	static constexpr int BACK   = DM + E;                // first element of the window = b' - BACK
	static constexpr int HALO_L = (BACK + HN_G - 1) / HN_G;   // leading blocks that only feed
	static constexpr int HALO_R = (E + HN_G - 1) / HN_G;      // trailing ones
	static constexpr int OUT    = (HN_THREADS - HALO_L - HALO_R) * HN_G;
	static constexpr int NLEFT  = HN_G + DR;             // elements that hold the 16 left ends a'
	static constexpr int NT     = DQ - 1;                // whole blocks always between
	static constexpr int LO     = HALO_L * HN_G - BACK;  // offset of element b'-BACK in block p-HALO_L (s = 0)
	static constexpr int LEAD   = HALO_L * HN_G - (H - E);    // staged elements before the first output
	static constexpr int NEDGE  = HN_G + E - 1;          // inputs under the E direct taps of 16 outputs
	static_assert (DQ >= 2, "window shorter than two blocks");
	static_assert ((LEAD & 1) == 0, "tile start must stay 16-byte aligned");
	static_assert (LO + NEDGE <= 2 * HN_G, "left edge spans more than two blocks");
	};

template <int W, int EE = HN_E> struct HannConsts
	{
	double edge[(EE > 0)? EE : 1];                                 // 1 - cos(w k),        k = 1..E
	double ownC[HN_G], ownS[HN_G];                                 // exp(+j w u),         u = 0..15
	double leftC[HannGeom<W, EE>::NLEFT], leftS[HannGeom<W, EE>::NLEFT];   // exp(+j w (u - DM)),  u = 0..NLEFT-1
	double rotC[HannGeom<W, EE>::NT], rotS[HannGeom<W, EE>::NT];   // exp(-j w 16 d),      d = 1..NT
	double demC[HN_G], demS[HN_G];                                 // exp(+j w (W-E - s)), s = 0..15
	double scale;                                                  // c = 1 / (2 * sum of raw taps)
	};

// Staging and phases 0-2 for one tile whose first staged element is e0 (even): on return acc[u] of a live thread p holds
// output (p - HALO_L)*16 + u of the tile, unless the tile must not go through the block sums (the return value: uniform
// over the workgroup), and the staged inputs are still in the LDS image (no barrier after the last read).
// With STATS, stats[wave] = { largest magnitude's high word, bit 0: a sign bit is set, bit 1: 0 < |x| < 2^-500 } of the wave's
// 64 blocks (visible on return).
// The LDS image has a pitch of 17 doubles per block of 16: the 17th slot of every block is never staged, read or written
// by the block sums.  A kernel that must fit its LDS into a quarter of a CU's (gdsp_peaks.hip: image + block totals are
// 40 KiB to the byte) keeps its few words of bookkeeping there: pad word i = the low word of block i's 17th slot.
__device__ __forceinline__ uint32_t* hann_pad_word (double* lds, int i)
	{ return reinterpret_cast<uint32_t*> (lds + i * HN_PITCH + HN_G); }

// -DPK_STAMPS (experiments only, tools/exp_peaks_stamps.sh): thread 0 of every workgroup adds the core-clock cycles since its
// entry to pkStamps[i] at point i (one workgroup in 128: every workgroup's seven atomics on one line made the kernel four
// times slower); the entry time waits in pad slot 210 (the plain form only: CWM keeps its change bits in every pad)
#ifdef PK_STAMPS
__device__ unsigned long long pkStamps[16];
// (the entry time in the LOW words of pads 210 and 211: CWM keeps its change bits in every pad's high word)
__device__ __forceinline__ void pk_stamp_set (double* lds)
	{ const long long t = clock64 ();  *hann_pad_word (lds, 210) = (uint32_t) t;  *hann_pad_word (lds, 211) = (uint32_t) (t >> 32); }
__device__ __forceinline__ long long pk_stamp_get (double* lds)
	{ return (long long) (((unsigned long long) *hann_pad_word (lds, 211) << 32) | *hann_pad_word (lds, 210)); }
#define PK_STAMP0(lds)   do { if ((threadIdx.x == 0) && ((blockIdx.x & 127) == 0)) { pk_stamp_set (lds);  atomicAdd (&pkStamps[15], 1ULL); } } while (0)
#define PK_STAMP(lds, i) do { if ((threadIdx.x == 0) && ((blockIdx.x & 127) == 0)) atomicAdd (&pkStamps[i], (unsigned long long) (clock64 () - pk_stamp_get (lds))); } while (0)
#else
#define PK_STAMP0(lds)   do { } while (0)
#define PK_STAMP(lds, i) do { } while (0)
#endif

// PADS: the four waves' verdicts live in pad words 0..3 instead of `huge` (which is then not used)
// SSTATS (with PADS; gdsp_peaks.hip): what the tile's inputs are like is found while they are STAGED, on the values in
// the loading registers -- the largest magnitude's verdict (pad words 0..3, instead of phase 1's look at every element)
// and, per wave, pad word HN_PAD_STATS + wave = { bit 0: a sign bit is set, bit 1: 0 < |x| < 2^-500 }.  The same elements
// as the 256 own blocks: the whole staged tile.
#define HN_PAD_STATS 4
#ifndef HN_REDO
#define HN_REDO 1
#endif
// CHG (gdsp_peaks.hip, the form for flat stretches): phase 1 leaves in the HIGH word of the own pad which of the block's
// elements differ from the element before them (bit u), found on the values it reads anyway
// RAW (gdsp_peaks.hip): the results are left without the window's normalisation (acc = S - C, not scale x (S - C)): the
// interval test on high words compares, and a positive factor changes no comparison; with EE = 0 there is no direct-tap
// sum to add either -- 32 vector instructions a thread less
template <int W, bool STATS = false, int EE = HN_E, bool PADS = false, bool SSTATS = false, bool RAW = false, bool CHG = false, bool LATE0 = false>
__device__ __forceinline__ bool hann_tile_sums (double* lds, double (*tot)[HN_THREADS], uint32_t* huge,
                                                const double* __restrict__ in, uint32_t n, int64_t e0,
                                                const HannConsts<W, EE>& K, double (&acc)[HN_G], uint32_t (*stats)[2] = NULL)
	{
	typedef HannGeom<W, EE> G;
	const int  p    = threadIdx.x;
	const bool live = (p >= G::HALO_L) && (p < HN_THREADS - G::HALO_R);

	// ---- stage 4096 elements, zero outside the chromosome
	uint32_t sSigns = 0, sSmallest = 0xFFFFFFFFu, sBig = 0;       // (SSTATS) of the elements this thread stages
	auto look = [&] (double x)
		{
		const uint32_t hi = (uint32_t) (__double_as_longlong (x) >> 32), lo = (uint32_t) __double_as_longlong (x);
		const uint32_t m  = hi & 0x7FFFFFFFu;
		sSigns |= hi;
		sBig = max (sBig, m);
		sSmallest = min (sSmallest, (m | min (lo, 1u)) - 1u);         // 0 only for a zero, which wraps to the top: ignored; a denormal's key is >= 1
		};
	if ((e0 >= 0) && (e0 + HN_ELEMS <= (int64_t) n))
		{
		const double2* src = reinterpret_cast<const double2*> (in + e0);
		double2 r[HN_G/2];
#pragma unroll
		for (int u=0 ; u<HN_G/2 ; u++) r[u] = gdsp_ld2 (&src[u*HN_THREADS + p]);
#pragma unroll
		for (int u=0 ; u<HN_G/2 ; u++)
			{
			const int e = 2 * (u*HN_THREADS + p);
			double* dst = lds + e + (e >> 4);
			dst[0] = r[u].x;  dst[1] = r[u].y;
			if (SSTATS) { look (r[u].x);  look (r[u].y); }
			}
		}
	else
		{
		for (int e=p ; e<HN_ELEMS ; e+=HN_THREADS)
			{
			const int64_t g = e0 + e;
			const double  x = ((g >= 0) && (g < (int64_t) n))? in[g] : 0.0;
			lds[e + (e >> 4)] = x;
			if (SSTATS) look (x);
			}
		}
	if (SSTATS)
		{
		const uint32_t flags = ((__builtin_amdgcn_ballot_w64 ((sSigns >> 31) != 0) != 0)? 1u : 0u)
		                     | ((__builtin_amdgcn_ballot_w64 (sSmallest < 0x20B00000u - 1u) != 0)? 2u : 0u);
		const bool any = (__builtin_amdgcn_ballot_w64 (sBig >= HN_HUGE_HI) != 0);
		if ((p & 63) == 0) { *hann_pad_word (lds, HN_PAD_STATS + (p >> 6)) = flags;  *hann_pad_word (lds, p >> 6) = any? 1u : 0u; }
		}
	__syncthreads ();
	if (PADS) PK_STAMP (lds, 1);                                   // staged (the tile's loads have landed)

	// ---- phase 0: the E taps at either end of each of the 16 windows, directly.  LATE0: behind phase 2 -- the same
	//      operations in the same order, added to the middle stretch's sum as before; only their sixteen sums do not sit
	//      in registers through phases 1 and 2 (with them the block sums need 146 registers: three waves per SIMD)
	auto edge_taps = [&] (double (&e)[HN_G])
		{
#pragma unroll
		for (int s=0 ; s<HN_G ; s++) e[s] = 0.0;
		if ((G::E > 0) && live)
			{
			const double* xl = lds + (p - G::HALO_L) * HN_PITCH;        // element b'-BACK of s=0 is xl[LO]
#pragma unroll
			for (int j=0 ; j<G::NEDGE ; j++)                            // window s meets input j under tap k = j-s+1
				{
				const int    o = G::LO + j;
				const double x = xl[o + (o >> 4)];
#pragma unroll
				for (int s=0 ; s<HN_G ; s++)
					{ if ((j - s >= 0) && (j - s < G::E)) e[s] = __builtin_fma (K.edge[j-s], x, e[s]); }
				}
			const double* xr = lds + p * HN_PITCH;                      // element b'+m of window s is xr[s+m]
#pragma unroll
			for (int j=1 ; j<=G::NEDGE ; j++)                           // tap W+1-m = tap m from the far end
				{
				const double x = xr[j + (j >> 4)];
#pragma unroll
				for (int s=0 ; s<HN_G ; s++)
					{ if ((j - s >= 1) && (j - s <= G::E)) e[s] = __builtin_fma (K.edge[G::E - (j-s)], x, e[s]); }
				}
			}
		};
	static_assert (!(LATE0 && RAW), "RAW leaves no direct taps to add");
	if (!LATE0) edge_taps (acc);

	// ---- phase 1: prefix sums of the own block in the own phase
	double P0[HN_G], Pr[HN_G], Pi[HN_G];
		{
		const double* xb = lds + p * HN_PITCH;
		double a0 = 0.0, ar = 0.0, ai = 0.0;
		uint32_t big = 0;                                          // largest exponent seen in the own block
		uint32_t signs = 0;  bool tiny = false;
		uint32_t chg = 0;                                          // (CHG) bit u: element u of the block differs from the one before it
		long long before = (CHG && (p > 0))? __double_as_longlong (lds[(p - 1) * HN_PITCH + HN_G - 1]) : 0;
#pragma unroll
		for (int u=0 ; u<HN_G ; u++)
			{
			const double x = xb[u];
			if (CHG)
				{
				chg |= ((__double_as_longlong (x) != before) || ((p == 0) && (u == 0)))? (1u << u) : 0u;     // (the tile's first element: a change)
				before = __double_as_longlong (x);
				}
			if (!SSTATS) big = max (big, hann_magnitude_hi (x));
			if (STATS)
				{
				const uint32_t hi = (uint32_t) (__double_as_longlong (x) >> 32), lo = (uint32_t) __double_as_longlong (x);
				signs |= hi;
				tiny = tiny || (((hi & 0x7FFFFFFFu) < 0x20B00000u) && (((hi & 0x7FFFFFFFu) | lo) != 0));
				}
			a0 += x;
			ar  = __builtin_fma (x, K.ownC[u], ar);
			ai  = __builtin_fma (x, K.ownS[u], ai);
			P0[u] = a0;  Pr[u] = ar;  Pi[u] = ai;
			}
		tot[0][p] = a0;  tot[1][p] = ar;  tot[2][p] = ai;
		if (CHG) hann_pad_word (lds, p)[1] = chg;
		if (!SSTATS)
			{
			const bool any = (__builtin_amdgcn_ballot_w64 (big >= HN_HUGE_HI) != 0);   // the 256 blocks are the whole tile
			if ((p & 63) == 0) { if (PADS) *hann_pad_word (lds, p >> 6) = any? 1u : 0u;  else huge[p >> 6] = any? 1u : 0u; }
			}
		if (STATS)
			{
			const uint32_t flags = ((__builtin_amdgcn_ballot_w64 ((signs >> 31) != 0) != 0)? 1u : 0u)
			                     | ((__builtin_amdgcn_ballot_w64 (tiny) != 0)? 2u : 0u);
			for (int off=32 ; off>0 ; off>>=1) big = max (big, (uint32_t) __shfl_xor ((int) big, off, 64));
			if ((p & 63) == 0) { stats[p >> 6][0] = big;  stats[p >> 6][1] = flags; }
			}
		}
	__syncthreads ();
	if (PADS) PK_STAMP (lds, 2);                                   // phase 1 (own block's prefix sums), its barrier
	bool direct;                                                   // uniform over the workgroup
	if (PADS) direct = ((*hann_pad_word (lds, 0) | *hann_pad_word (lds, 1) | *hann_pad_word (lds, 2) | *hann_pad_word (lds, 3)) != 0);
	else
		{
		const uint4 hg = *reinterpret_cast<const uint4*> (huge);
		direct = ((hg.x | hg.y | hg.z | hg.w) != 0);
		}

	// ---- phase 2: the middle stretch of one window per left end
	if (live && !direct)
		{
		double T0 = 0.0, Tr = 0.0, Ti = 0.0;                       // whole blocks p-NT .. p-1
#pragma unroll
		for (int d=G::NT ; d>=1 ; d--)
			{
			const double b0 = tot[0][p-d], br = tot[1][p-d], bi = tot[2][p-d];
			T0 += b0;
			Tr += __builtin_fma (br, K.rotC[d-1], -(bi * K.rotS[d-1]));
			Ti += __builtin_fma (br, K.rotS[d-1],   bi * K.rotC[d-1]);
			}
		// the own block's prefix sums, as phase 1 left them -- but the last ones are the block's totals, read back from the
		// totals, and the first HN_REDO are made again from the block's first elements (the same operations on the same
		// operands) where the last windows want them: 48 doubles alive from phase 1 to the end of phase 2 are more than the
		// 128 registers of four waves per SIMD hold beside the sums in flight, and what the compiler spilled (5-7 doubles
		// a lane, written and read back through scratch memory) went to memory with the tile's outputs
		auto prefix = [&] (int u, double& q0, double& qr, double& qi)
			{
			if (u == HN_G - 1) { q0 = tot[0][p];  qr = tot[1][p];  qi = tot[2][p]; }
			else if (u < HN_REDO)
				{
				const double* xb = lds + p * HN_PITCH;
				q0 = 0.0;  qr = 0.0;  qi = 0.0;
#pragma unroll
				for (int v=0 ; v<HN_REDO ; v++)
					{
					if (v > u) continue;
					const double x = xb[v];
					q0 += x;
					qr  = __builtin_fma (x, K.ownC[v], qr);
					qi  = __builtin_fma (x, K.ownS[v], qi);
					}
				}
			else { q0 = P0[u];  qr = Pr[u];  qi = Pi[u]; }
			};
		const double* lb = lds + (p - G::DQ) * HN_PITCH;            // block of the left ends of s >= DR
		const double* la = lb - HN_PITCH + (HN_G - G::DR);          // last DR elements of the block before
		double s0 = 0.0, sr = 0.0, si = 0.0;
#pragma unroll
		for (int u=G::NLEFT-1 ; u>=G::DR ; u--)
			{
			const double x = lb[u - G::DR];
			s0 += x;
			sr  = __builtin_fma (x, K.leftC[u], sr);
			si  = __builtin_fma (x, K.leftS[u], si);
			if (u < HN_G)                                           // left end of the stretch whose right end is own[u]
				{
				double q0, qr, qi;
				prefix (u, q0, qr, qi);
				const double z0 = (s0 + T0) + q0;
				const double zr = (sr + Tr) + qr;
				const double zi = (si + Ti) + qi;
				const double c  = __builtin_fma (K.demC[u], zr, -(K.demS[u] * zi));
				acc[u] = RAW? ((EE == 0)? (z0 - c) : ((z0 - c) + acc[u])) : LATE0? (z0 - c) : K.scale * ((z0 - c) + acc[u]);
				}
			}
		T0 += s0;  Tr += sr;  Ti += si;                            // that block is whole for the remaining windows
		s0 = 0.0;  sr = 0.0;  si = 0.0;
#pragma unroll
		for (int u=G::DR-1 ; u>=0 ; u--)
			{
			const double x = la[u];
			s0 += x;
			sr  = __builtin_fma (x, K.leftC[u], sr);
			si  = __builtin_fma (x, K.leftS[u], si);
			double q0, qr, qi;
			prefix (u, q0, qr, qi);
			const double z0 = (s0 + T0) + q0;
			const double zr = (sr + Tr) + qr;
			const double zi = (si + Ti) + qi;
			const double c  = __builtin_fma (K.demC[u], zr, -(K.demS[u] * zi));
			acc[u] = RAW? ((EE == 0)? (z0 - c) : ((z0 - c) + acc[u])) : LATE0? (z0 - c) : K.scale * ((z0 - c) + acc[u]);
			}
		if (LATE0)
			{
			__builtin_amdgcn_sched_barrier (0);                      // (the direct taps behind the walk, not among its last steps: registers)
			double e[HN_G];
			edge_taps (e);
#pragma unroll
			for (int u=0 ; u<HN_G ; u++) acc[u] = K.scale * (acc[u] + e[u]);
			}
		}
	return direct;
	}

template <int W, int EE = HN_E>
static void hann_consts (HannConsts<W, EE>& K)
	{
	typedef HannGeom<W, EE> G;
	const double pi = 3.14159265358979323846264;
	const int    M  = W + 1;                                       // the window's period
	auto cs = [&] (long m, double* c, double* sn)                   // exp(j*2*pi*m/M), argument reduced first
		{
		long r = ((m % M) + M) % M;
		double x = r / (double) M;
		*c = cos (2*pi*x);  *sn = sin (2*pi*x);
		};
	for (int k=1 ; k<=G::E ; k++)    { double c, sn;  cs (k, &c, &sn);  K.edge[k-1] = 1 - c; }
	for (int u=0 ; u<HN_G ; u++)     cs (u, &K.ownC[u], &K.ownS[u]);
	for (int u=0 ; u<G::NLEFT ; u++) cs ((long) u - G::DM, &K.leftC[u], &K.leftS[u]);
	for (int d=1 ; d<=G::NT ; d++)   cs (-(long) HN_G * d, &K.rotC[d-1], &K.rotS[d-1]);
	for (int u=0 ; u<HN_G ; u++)     cs ((long) W - G::E - u, &K.demC[u], &K.demS[u]);
	double total = 0.0;                                            // as gdsp_hann_taps sums it (sum.c:632-645)
	for (int k=0 ; k<W ; k++)
		{
		const int kk = (k <= G::H)? k : W-1-k;
		total += (1 - cos (2*pi*((kk+1) / (double) M))) / 2;
		}
	K.scale = 0.5 / total;
	}

