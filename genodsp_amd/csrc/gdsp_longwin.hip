// gdsp_longwin.hip -- windowed operators for windows too long for one LDS tile.
//
// The tiled kernels (gdsp_extrema.hip, gdsp_sums.hip) keep a window's worth of halo in LDS and
// therefore stop at a few thousand bases.  The reference accepts any window up to the
// chromosome length (u32), so these entry points finish the range with whole-vector passes
// through caller-provided HBM workspace:
//   * bestmin/bestmax, localmin/localmax: van Herk / Gil-Werman over the whole chromosome --
//     prefix extremes G and suffix extremes Hs over segments of `span` bases (one workgroup per
//     segment, walking it in 2048-base chunks with a carry), then out = pick(Hs[a], G[b]) for the
//     window [a,b]; 48-56 B/base of traffic instead of 16, but any window.
//   * slidingsum: whole-vector cumulative sum, then the difference of two prefix values per base
//     (bit-identical on exactly summable signals, like the tiled form).
// The *_any entry points pick the tiled kernel whenever the window fits it.

#include <math.h>
#include "gdsp_common.h"

#define LW_THREADS 256
#define LW_PER     8
#define LW_CHUNK   (LW_THREADS * LW_PER)

template <bool MAX> __device__ __forceinline__ double lw_pick (double a, double b)
	{ return MAX? fmax (a, b) : fmin (a, b); }

// prefix (REVERSE=false) or suffix (REVERSE=true) extremes inside segment blockIdx.x
template <bool MAX, bool REVERSE>
__global__ __launch_bounds__(LW_THREADS)
void segment_extremes_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t span)
	{
	__shared__ double waveTot[LW_THREADS/64];
	__shared__ double carryShared;
	const double   pad  = MAX? -INFINITY : INFINITY;
	const uint64_t s0   = (uint64_t) blockIdx.x * span;
	const uint64_t s1   = (s0 + span < n)? s0 + span : n;
	const uint64_t len  = s1 - s0;
	const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double carry = pad;

	for (uint64_t c0=0 ; c0<len ; c0+=LW_CHUNK)
		{
		// element k of the segment in scan order sits at s0+k (prefix) or s1-1-k (suffix)
		double x[LW_PER];
		double run = pad;
#pragma unroll
		for (int i=0 ; i<LW_PER ; i++)
			{
			const uint64_t k = c0 + (uint64_t) threadIdx.x * LW_PER + i;
			x[i] = (k < len)? in[REVERSE? s1 - 1 - k : s0 + k] : pad;
			run  = lw_pick<MAX> (run, x[i]);
			x[i] = run;
			}
		// exclusive scan of the per-thread extremes across the workgroup
		double incl = run;
		for (int d=1 ; d<64 ; d*=2)
			{
			const double up = __shfl_up (incl, d, 64);
			if (lane >= d) incl = lw_pick<MAX> (up, incl);
			}
		double excl = __shfl_up (incl, 1, 64);
		if (lane == 0) excl = pad;
		__syncthreads ();                              // waveTot / carryShared free again
		if (lane == 63) waveTot[wave] = incl;
		__syncthreads ();
		double before = carry;
		for (int w=0 ; w<wave ; w++) before = lw_pick<MAX> (before, waveTot[w]);
		before = lw_pick<MAX> (before, excl);
#pragma unroll
		for (int i=0 ; i<LW_PER ; i++)
			{
			const uint64_t k = c0 + (uint64_t) threadIdx.x * LW_PER + i;
			if (k < len) out[REVERSE? s1 - 1 - k : s0 + k] = lw_pick<MAX> (before, x[i]);
			}
		if (threadIdx.x == LW_THREADS-1) carryShared = lw_pick<MAX> (before, run);
		__syncthreads ();
		carry = carryShared;
		}
	}

template <bool MAX, bool LOCAL>
__global__ __launch_bounds__(LW_THREADS)
void window_extreme_kernel (const double* __restrict__ in, const double* __restrict__ G, const double* __restrict__ Hs,
                            double* __restrict__ out, uint32_t n, uint32_t lft, uint32_t rgt, uint32_t span, double fill)
	{
	const size_t stride = (size_t) gridDim.x * LW_THREADS;
	for (size_t i = (size_t) blockIdx.x * LW_THREADS + threadIdx.x ; i < n ; i += stride)
		{
		const size_t a = (i >= lft)? i - lft : 0;
		const size_t b = (i + rgt < n)? i + rgt : (size_t) n - 1;
		// a full window either fills one segment exactly or spans two; a window cut short by an end
		// of the vector starts at a segment start (a == 0) or ends at a segment end (b == n-1)
		const size_t sa = a / span, sb = b / span;
		double e;
		if (sa != sb)                e = lw_pick<MAX> (Hs[a], G[b]);
		else if (a == sa * span)     e = G[b];
		else                         e = Hs[a];
		if (LOCAL)
			{
			const double c = in[i];
			e = (MAX? (e > c) : (e < c))? fill : c;
			}
		out[i] = e;
		}
	}

__global__ __launch_bounds__(LW_THREADS)
void prefix_difference_kernel (const double* __restrict__ S, double* __restrict__ out, uint32_t n,
                               uint32_t lft, uint32_t rgt, double denom)
	{
	const size_t stride = (size_t) gridDim.x * LW_THREADS;
	for (size_t i = (size_t) blockIdx.x * LW_THREADS + threadIdx.x ; i < n ; i += stride)
		{
		const size_t b  = (i + rgt < n)? i + rgt : (size_t) n - 1;
		const double hi = S[b];
		const double lo = (i > lft)? S[i - lft - 1] : 0.0;
		out[i] = (hi - lo) / denom;
		}
	}

template <bool MAX, bool LOCAL>
static int long_extrema (const double* d_in, double* d_out, uint32_t n, uint32_t lft, uint32_t rgt, double fill,
                         void* d_work, size_t workBytes, void* stream)
	{
	GDSP_REQUIRE (d_work != NULL, "this window needs workspace (gdsp_long_window_work)");
	GDSP_REQUIRE (workBytes >= 2 * (size_t) n * sizeof(double), "workspace too small (gdsp_long_window_work)");
	if (lft > n) lft = n;
	if (rgt > n) rgt = n;
	uint64_t span = (uint64_t) lft + rgt + 1;
	if (span > n) span = n;
	double*        G     = (double*) d_work;
	double*        Hs    = G + n;
	const uint32_t nsegs = (uint32_t) (((uint64_t) n + span - 1) / span);
	hipStream_t    s     = gdsp_stream (stream);
	hipLaunchKernelGGL ((segment_extremes_kernel<MAX, false>), dim3(nsegs), dim3(LW_THREADS), 0, s, d_in, G,  n, (uint32_t) span);
	hipLaunchKernelGGL ((segment_extremes_kernel<MAX, true>),  dim3(nsegs), dim3(LW_THREADS), 0, s, d_in, Hs, n, (uint32_t) span);
	size_t   want   = ((size_t) n + LW_THREADS*4 - 1) / (LW_THREADS*4);
	uint32_t blocks = (uint32_t) (want > 4096? 4096 : (want < 1? 1 : want));
	// a window [a,b] never spans more than two segments because b-a+1 <= span
	hipLaunchKernelGGL ((window_extreme_kernel<MAX, LOCAL>), dim3(blocks), dim3(LW_THREADS), 0, s,
	                    d_in, G, Hs, d_out, n, lft, rgt, (uint32_t) span, fill);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

extern "C" {

/* workspace the *_any entry points may need for this vector length: two vectors of n doubles
 * (extrema) or one plus the scan totals (slidingsum) -- the larger is returned */
size_t gdsp_long_window_work (uint32_t n)
	{ return 2 * ((size_t) n + 2) * sizeof(double) + gdsp_cumulative_sum_work (n) + 1024; }   // (+ the fixed part of the morphology tables)

int gdsp_best_extrema_any (const double* d_in, double* d_out, uint32_t n, uint32_t W, int wantMax,
                           void* d_work, size_t workBytes, void* stream)
	{
	int rc = gdsp_best_extrema (d_in, d_out, n, W, wantMax, stream);
	if ((rc != GDSP_EINVAL) || (n == 0) || (d_in == NULL) || (d_out == NULL) || (d_in == d_out) || (W < 1)) return rc;
	const uint32_t lft = (W - 1) / 2, rgt = (W - 1) - lft;
	if (wantMax) return long_extrema<true,  false> (d_in, d_out, n, lft, rgt, 0.0, d_work, workBytes, stream);
	return              long_extrema<false, false> (d_in, d_out, n, lft, rgt, 0.0, d_work, workBytes, stream);
	}

int gdsp_local_extrema_any (const double* d_in, double* d_out, uint32_t n, uint32_t N, int wantMax, double fill,
                            void* d_work, size_t workBytes, void* stream)
	{
	int rc = gdsp_local_extrema (d_in, d_out, n, N, wantMax, fill, stream);
	if ((rc != GDSP_EINVAL) || (n == 0) || (d_in == NULL) || (d_out == NULL) || (d_in == d_out) || (N < 1)) return rc;
	const uint32_t h = (N - 1) / 2;
	if (wantMax) return long_extrema<true,  true> (d_in, d_out, n, h, h, fill, d_work, workBytes, stream);
	return              long_extrema<false, true> (d_in, d_out, n, h, h, fill, d_work, workBytes, stream);
	}

int gdsp_sliding_sum_any (const double* d_in, double* d_out, uint32_t n, uint32_t W, double denom,
                          void* d_work, size_t workBytes, void* stream)
	{
	int rc = gdsp_sliding_sum (d_in, d_out, n, W, denom, stream);
	if ((rc != GDSP_EINVAL) || (n == 0) || (d_in == NULL) || (d_out == NULL) || (d_in == d_out) || (W < 1) || (denom == 0.0)) return rc;
	GDSP_REQUIRE (d_work != NULL, "this window needs workspace (gdsp_long_window_work)");
	GDSP_REQUIRE (workBytes >= ((size_t) n + 2) * sizeof(double) + gdsp_cumulative_sum_work (n),
	              "workspace too small (gdsp_long_window_work)");
	double* S      = (double*) d_work;
	void*   totals = (void*) (S + (((size_t) n + 2) & ~(size_t) 1));
	GDSP_HIP_TRY (hipMemcpyAsync (S, d_in, (size_t) n * sizeof(double), hipMemcpyDeviceToDevice, gdsp_stream (stream)));
	rc = gdsp_cumulative_sum (S, n, totals, stream);
	if (rc != GDSP_OK) return rc;
	uint32_t hOff = (W - 1) / 2, lft = W - 1 - hOff;           // sum.c:436-455
	size_t   want   = ((size_t) n + LW_THREADS*4 - 1) / (LW_THREADS*4);
	uint32_t blocks = (uint32_t) (want > 4096? 4096 : (want < 1? 1 : want));
	hipLaunchKernelGGL (prefix_difference_kernel, dim3(blocks), dim3(LW_THREADS), 0, gdsp_stream (stream),
	                    S, d_out, n, lft, hOff, denom);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

} // extern "C"
