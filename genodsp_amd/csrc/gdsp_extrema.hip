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
// 16-byte loads; every lane then produces two adjacent outputs and stores them
// as one 16-byte word.  Small windows are scanned directly from LDS; large
// windows first build range extremes over power-of-two spans by doubling
// (log2(window) LDS sweeps), after which every window is two lookups.

#include <float.h>
#include "gdsp_common.h"

#define EX_THREADS 256
#define EX_TILE    4096              // outputs per workgroup
#define EX_DIRECT_MAX_SPAN 32        // windows up to this many bases are scanned directly
#define EX_LDS_DOUBLES 18432         // 144 KiB of the 160 KiB LDS

template <bool MAX> __device__ __forceinline__ bool ex_beats (double a, double b)
	{ return MAX? (a > b) : (a < b); }
// extreme of two values with NaN never winning: v_max_f64 / v_min_f64 (IEEE maxNum/minNum)
template <bool MAX> __device__ __forceinline__ double ex_pick (double a, double b)
	{ return MAX? fmax (a, b) : fmin (a, b); }

// LOCAL: localmin/localmax semantics; otherwise bestmin/bestmax
template <bool MAX, bool LOCAL, bool DIRECT>
__global__ __launch_bounds__(EX_THREADS)
void extrema_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t ntiles,
                     uint32_t lft, uint32_t rgt, double fill, int tile)
	{
	extern __shared__ __attribute__((aligned(16))) double lds[];
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

	if (DIRECT)
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
			if (g + 1 < (int64_t) n) *reinterpret_cast<double2*> (out + g) = make_double2 (e0, e1);
			else if (g < (int64_t) n) out[g] = e0;
			}
		return;
		}

	// doubling: after step s, cur[i] = extreme of x[i .. i+2^s); ping-pong between two LDS arrays
	double* cur = lds;
	double* nxt = lds + L;
	int     lg  = 0;
	while ((2 << lg) <= span) lg++;              // 2^lg <= span < 2^(lg+1)
	for (int s=0 ; s<lg ; s++)
		{
		const int half = 1 << s;
		for (int p=threadIdx.x ; p+2*half<=L ; p+=EX_THREADS) nxt[p] = ex_pick<MAX> (cur[p], cur[p+half]);
		__syncthreads ();
		double* swap = cur;  cur = nxt;  nxt = swap;
		}
	const int     w2 = 1 << lg;
	const double* mm = cur + sh;
	for (int o = 2*threadIdx.x ; o < tile ; o += 2*EX_THREADS)
		{
		int64_t g = tileStart + o;
		if (g >= (int64_t) n) break;
		double e0 = ex_pick<MAX> (mm[o],   mm[o   + span - w2]);
		double e1 = ex_pick<MAX> (mm[o+1], mm[o+1 + span - w2]);
		if (LOCAL)
			{
			// the staged copy was consumed by the ping-pong; centres come back from L2
			double c0 = in[g], c1 = (g + 1 < (int64_t) n)? in[g+1] : 0.0;
			e0 = ex_beats<MAX> (e0, c0)? fill : c0;
			e1 = ex_beats<MAX> (e1, c1)? fill : c1;
			}
		if (g + 1 < (int64_t) n) *reinterpret_cast<double2*> (out + g) = make_double2 (e0, e1);
		else                     out[g] = e0;
		}
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
	const bool     direct = (span <= EX_DIRECT_MAX_SPAN);
	int            tile   = EX_TILE;
	size_t         ldsDoubles;
	if (direct) ldsDoubles = (size_t) tile + span + 2;
	else
		{
		// two ping-pong arrays of tile+span; the tile is the smallest that keeps the halo <= tile/2
		// that still fits
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
		hipLaunchKernelGGL ((extrema_kernel<MAX, LOCAL, true>),  dim3(ntiles), dim3(EX_THREADS), bytes, s,
		                    d_in, d_out, n, ntiles, lft, rgt, fill, tile);
	else
		{
		if (bytes > 64*1024)          // more than 64 KiB of dynamic LDS has to be asked for
			GDSP_HIP_TRY (hipFuncSetAttribute ((const void*) extrema_kernel<MAX, LOCAL, false>,
			                                   hipFuncAttributeMaxDynamicSharedMemorySize,
			                                   (int) (EX_LDS_DOUBLES*sizeof(double))));
		hipLaunchKernelGGL ((extrema_kernel<MAX, LOCAL, false>), dim3(ntiles), dim3(EX_THREADS), bytes, s,
		                    d_in, d_out, n, ntiles, lft, rgt, fill, tile);
		}
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

extern "C" {

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
