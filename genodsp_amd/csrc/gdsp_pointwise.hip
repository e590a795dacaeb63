// gdsp_pointwise.hip -- per-base operators: binarize, clip, erase, addconst, abs,
// invert, fill, and the min/max/count reduction behind invert and percentile 0/100.
//
// All of these are one load and one store per base (16 B/base) and therefore HBM
// bound on MI355X: each lane moves 16 bytes per access, eight accesses in flight.
// Comparisons are written in the same form as the reference's so that NaN and
// signed-zero inputs take the same branch.

#include <float.h>
#include "gdsp_common.h"

#define PW_THREADS 256
#define PW_UNROLL  8                                  // 16-byte accesses in flight per lane
#define PW_TILE    (PW_THREADS * PW_UNROLL * 2)       // 4096 bases = 32 KiB per workgroup
#define PW_MAX_BLOCKS (256 * 8)

// One workgroup = one contiguous 32 KiB tile: all of its loads are issued before the first
// store, and tiles are dealt so that each XCD walks a contiguous eighth of the vector
// (gdsp_xcd_tile).  Measured on MI355X this streams ~25 % faster than a grid-stride loop over
// the same vector (profiles/r01_ops_throughput.txt).
template <class F>
__device__ __forceinline__ void pointwise_tile (double* __restrict__ v, uint32_t n, uint32_t tile, const F& f)
	{
	const size_t   base = (size_t) tile * PW_TILE;
	if (base + PW_TILE <= (size_t) n)
		{
		double2* p = reinterpret_cast<double2*> (v + base) + threadIdx.x;
		double2  d[PW_UNROLL];
#pragma unroll
		for (int u=0 ; u<PW_UNROLL ; u++) d[u] = gdsp_ld2 (&p[u*PW_THREADS]);
#pragma unroll
		for (int u=0 ; u<PW_UNROLL ; u++) { d[u].x = f (d[u].x);  d[u].y = f (d[u].y); }
#pragma unroll
		for (int u=0 ; u<PW_UNROLL ; u++) gdsp_st2 (&p[u*PW_THREADS], d[u]);
		}
	else
		{
		for (size_t i = base + threadIdx.x ; i < (size_t) n ; i += PW_THREADS) v[i] = f (v[i]);
		}
	}

template <class F>
__global__ __launch_bounds__(PW_THREADS)
void pointwise_kernel (double* __restrict__ v, uint32_t n, uint32_t ntiles, F f)
	{ pointwise_tile (v, n, gdsp_xcd_tile (blockIdx.x, ntiles), f); }

template <class F>                                    // one grid over every vector of the table (gdsp_common.h), in place on B.out
__global__ __launch_bounds__(PW_THREADS)
void pointwise_batch_kernel (GdspBatch B, F f)
	{
	const double* in;  double* v;  uint32_t n;
	const uint32_t tile = gdsp_batch_tile (B, in, v, n);
	pointwise_tile (v, n, tile, f);
	}

template <class F>
static int pointwise_batch_launch (const gdsp_batch_item* items, int nitems, F f, void* stream)
	{
	int rc = gdsp_batch_check (items, nitems, true);
	if (rc != GDSP_OK) return rc;
	hipStream_t s = gdsp_stream (stream);
	gdsp_batch_run (items, nitems, [] (uint32_t n) { return ((uint64_t) n + PW_TILE - 1) / PW_TILE; },
		[&] (const GdspBatch& B, uint32_t tiles)
			{ hipLaunchKernelGGL ((pointwise_batch_kernel<F>), dim3(tiles), dim3(PW_THREADS), 0, s, B, f); });
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

template <class F>
static int pointwise_launch (double* d_v, uint32_t n, F f, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE (d_v != NULL, "NULL vector");
	GDSP_REQUIRE (gdsp_aligned16 (d_v), "vector must be 16-byte aligned");
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + PW_TILE - 1) / PW_TILE);
	hipLaunchKernelGGL ((pointwise_kernel<F>), dim3(ntiles), dim3(PW_THREADS), 0, gdsp_stream (stream), d_v, n, ntiles, f);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

// logical.c:247-257
struct BinarizeAbove { double T, one, zero;  __device__ double operator() (double x) const { return (x >= T)? one : zero; } };
struct BinarizeBelow { double T, one, zero;  __device__ double operator() (double x) const { return (x >  T)? one : zero; } };
// mask.c:893-911
struct ClipMin  { double lo;      __device__ double operator() (double x) const { return (x < lo)? lo : x; } };
struct ClipMax  { double hi;      __device__ double operator() (double x) const { return (x > hi)? hi : x; } };
struct ClipBoth { double lo, hi;  __device__ double operator() (double x) const { return (x < lo)? lo : ((x > hi)? hi : x); } };
// mask.c:1187-1227 (kind: 0 below-min, 1 above-max, 2 outside, 3 >=min, 4 <=max, 5 inside)
template <int KIND>
struct Erase
	{
	double lo, hi, zero;
	__device__ double operator() (double x) const
		{
		bool hit;
		if      (KIND == 0) hit = (x < lo);
		else if (KIND == 1) hit = (x > hi);
		else if (KIND == 2) hit = (x < lo) || (x > hi);
		else if (KIND == 3) hit = (x >= lo);
		else if (KIND == 4) hit = (x <= hi);
		else                hit = (x >= lo) && (x <= hi);
		return hit? zero : x;
		}
	};
struct AddConst { double c;       __device__ double operator() (double x) const { return x + c; } };          // add.c:738-739
struct AbsVal   {                 __device__ double operator() (double x) const { return (x < 0)? -x : x; } }; // add.c:1046-1047
struct Invert   { double twoMid;  __device__ double operator() (double x) const { return twoMid - x; } };      // add.c:935-936
struct FillVal  { double c;       __device__ double operator() (double)   const { return c; } };

// ---------------------------------------------------------------- reduction ----
// d_minmax[0] = min, [1] = max, [2] = count of the sampled values (every window-th
// element with lo <= v <= hi; the two rejection tests are the reference's
// `v < min -> skip`, `v > max -> skip`, percentile.c:447-448).
__device__ __forceinline__ void atomic_min_f64 (double* addr, double val)
	{
	unsigned long long* a = reinterpret_cast<unsigned long long*> (addr);
	unsigned long long  old = *a, assumed;
	do  {
		assumed = old;
		if (!(val < __longlong_as_double ((long long) assumed))) break;
		old = atomicCAS (a, assumed, (unsigned long long) __double_as_longlong (val));
		} while (assumed != old);
	}
__device__ __forceinline__ void atomic_max_f64 (double* addr, double val)
	{
	unsigned long long* a = reinterpret_cast<unsigned long long*> (addr);
	unsigned long long  old = *a, assumed;
	do  {
		assumed = old;
		if (!(val > __longlong_as_double ((long long) assumed))) break;
		old = atomicCAS (a, assumed, (unsigned long long) __double_as_longlong (val));
		} while (assumed != old);
	}

__global__ __launch_bounds__(PW_THREADS)
void minmax_kernel (const double* __restrict__ v, uint32_t n, uint32_t window, double lo, double hi,
                    double* __restrict__ result)
	{
	const size_t nsamp  = ((size_t) n + window - 1) / window;
	const size_t stride = (size_t) gridDim.x * PW_THREADS;
	double   mn = DBL_MAX, mx = -DBL_MAX;
	uint32_t cnt = 0;

	for (size_t s = (size_t) blockIdx.x * PW_THREADS + threadIdx.x ; s < nsamp ; s += stride)
		{
		double x = v[s * window];
		if (x < lo) continue;
		if (x > hi) continue;
		if (x < mn) mn = x;
		if (x > mx) mx = x;
		cnt++;
		}

	// wave reduce (64 lanes), then one set of atomics per wave
	for (int off=32 ; off>0 ; off>>=1)
		{
		double   omn = __shfl_down (mn, off, 64);
		double   omx = __shfl_down (mx, off, 64);
		uint32_t oc  = __shfl_down (cnt, off, 64);
		if (omn < mn) mn = omn;
		if (omx > mx) mx = omx;
		cnt += oc;
		}
	if ((threadIdx.x & 63) == 0)
		{
		if (cnt != 0)
			{
			atomic_min_f64 (&result[0], mn);
			atomic_max_f64 (&result[1], mx);
			atomicAdd (&result[2], (double) cnt);
			}
		}
	}

// window 1, 16-byte aligned: a workgroup walks 32 KiB tiles (grid stride) with the tile's eight
// 16-byte loads per lane issued together; the grid-stride loop above reads one value per lane at a time
__global__ __launch_bounds__(PW_THREADS)
void minmax_dense_kernel (const double* __restrict__ v, uint32_t n, double lo, double hi, double* __restrict__ result)
	{
	double   mn = DBL_MAX, mx = -DBL_MAX;
	uint32_t cnt = 0;
	auto take = [&] (double x)
		{
		const bool in = !(x < lo) && !(x > hi);
		if (in && (x < mn)) mn = x;
		if (in && (x > mx)) mx = x;
		cnt += in;
		};
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + PW_TILE - 1) / PW_TILE);
	for (uint32_t tile=blockIdx.x ; tile<ntiles ; tile+=gridDim.x)
		{
		const size_t base = (size_t) tile * PW_TILE;
		if (base + PW_TILE <= (size_t) n)
			{
			const double2* p = reinterpret_cast<const double2*> (v + base) + threadIdx.x;
			double2 d[PW_UNROLL];
#pragma unroll
			for (int u=0 ; u<PW_UNROLL ; u++) d[u] = gdsp_ld2 (&p[u*PW_THREADS]);
#pragma unroll
			for (int u=0 ; u<PW_UNROLL ; u++) { take (d[u].x);  take (d[u].y); }
			}
		else
			{
			for (size_t i = base + threadIdx.x ; i < (size_t) n ; i += PW_THREADS) take (v[i]);
			}
		}
	for (int off=32 ; off>0 ; off>>=1)
		{
		double   omn = __shfl_down (mn, off, 64);
		double   omx = __shfl_down (mx, off, 64);
		uint32_t oc  = __shfl_down (cnt, off, 64);
		if (omn < mn) mn = omn;
		if (omx > mx) mx = omx;
		cnt += oc;
		}
	if (((threadIdx.x & 63) == 0) && (cnt != 0))
		{
		atomic_min_f64 (&result[0], mn);
		atomic_max_f64 (&result[1], mx);
		atomicAdd (&result[2], (double) cnt);
		}
	}

// ------------------------------------------------------------------- map ----
// op_map_apply, map.c:194-381: piecewise-linear mapping through a table of (in, out) knots
// sorted by `in`.  Values at or below the first knot / at or above the last take that knot's
// output; in between, the piece is found by the reference's binary search (map.c:300-315,
// including its skip over repeated knots) and the value is out_lo + (v - in_lo) * out_diff /
// in_diff with the reference's operation order, each operation rounded on its own.  The
// reference also remembers the last piece between bases; for strictly increasing knots that
// shortcut picks the same piece as the search, so the result is bit-identical.  The table
// sits in LDS (up to MAP_LDS_KNOTS knots) or is searched in global memory.
#define MAP_LDS_KNOTS 2048
template <bool IN_LDS>
__global__ __launch_bounds__(PW_THREADS)
void map_kernel (double* __restrict__ v, uint32_t n, uint32_t ntiles,
                 const double* __restrict__ knotIn, const double* __restrict__ knotOut, uint32_t nknots)
	{
	__shared__ double ldsIn[IN_LDS? MAP_LDS_KNOTS : 1], ldsOut[IN_LDS? MAP_LDS_KNOTS : 1];
	const double* tin  = knotIn;
	const double* tout = knotOut;
	if (IN_LDS)
		{
		for (uint32_t k=threadIdx.x ; k<nknots ; k+=PW_THREADS) { ldsIn[k] = knotIn[k];  ldsOut[k] = knotOut[k]; }
		__syncthreads ();
		tin = ldsIn;  tout = ldsOut;
		}
	const uint32_t maxIx = nknots - 1;
	const double   minIn = tin[0], maxIn = tin[maxIx], outForMin = tout[0], outForMax = tout[maxIx];

	auto mapOne = [&] (double x) -> double
		{
		if (x <= minIn) return outForMin;
		if (x >= maxIn) return outForMax;
		if (x != x)     return x;                       // NaN: the reference's search would not terminate meaningfully
		uint32_t lo = 0, hi = maxIx;
		while (lo + 1 < hi)
			{
			const uint32_t mid = (lo + hi) / 2;
			const double   m   = tin[mid];
			if      (x < m) hi = mid;
			else if (x > m) lo = mid;
			else          { lo = mid;  break; }
			}
		while ((lo < maxIx) && (tin[lo] == tin[lo+1])) lo++;
		const double pieceLo = tin[lo], pieceHi = tin[lo+1], outLo = tout[lo], outHi = tout[lo+1];
		if (x == pieceLo) return outLo;
		if (x == pieceHi) return outHi;
		return outLo + (x - pieceLo) * (outHi - outLo) / (pieceHi - pieceLo);
		};

	const uint32_t tile = gdsp_xcd_tile (blockIdx.x, ntiles);
	const size_t   base = (size_t) tile * PW_TILE;
	if (base + PW_TILE <= (size_t) n)
		{
		double2* p = reinterpret_cast<double2*> (v + base) + threadIdx.x;
		double2  d[PW_UNROLL];
#pragma unroll
		for (int u=0 ; u<PW_UNROLL ; u++) d[u] = gdsp_ld2 (&p[u*PW_THREADS]);
#pragma unroll
		for (int u=0 ; u<PW_UNROLL ; u++) { d[u].x = mapOne (d[u].x);  d[u].y = mapOne (d[u].y); }
#pragma unroll
		for (int u=0 ; u<PW_UNROLL ; u++) gdsp_st2 (&p[u*PW_THREADS], d[u]);
		}
	else
		{
		for (size_t i = base + threadIdx.x ; i < (size_t) n ; i += PW_THREADS) v[i] = mapOne (v[i]);
		}
	}

extern "C" {

int gdsp_map (double* d_v, uint32_t n, const double* d_knotIn, const double* d_knotOut, uint32_t nknots, void* stream)
	{
	GDSP_REQUIRE (nknots >= 1, "the mapping needs at least one knot");
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_v != NULL) && (d_knotIn != NULL) && (d_knotOut != NULL), "NULL pointer");
	GDSP_REQUIRE (gdsp_aligned16 (d_v), "vector must be 16-byte aligned");
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + PW_TILE - 1) / PW_TILE);
	if (nknots <= MAP_LDS_KNOTS)
		hipLaunchKernelGGL ((map_kernel<true>),  dim3(ntiles), dim3(PW_THREADS), 0, gdsp_stream (stream), d_v, n, ntiles, d_knotIn, d_knotOut, nknots);
	else
		hipLaunchKernelGGL ((map_kernel<false>), dim3(ntiles), dim3(PW_THREADS), 0, gdsp_stream (stream), d_v, n, ntiles, d_knotIn, d_knotOut, nknots);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

int gdsp_binarize (double* d_v, uint32_t n, double T, int tiesAbove, double one, double zero, void* stream)
	{
	if (tiesAbove) return pointwise_launch (d_v, n, BinarizeAbove {T, one, zero}, stream);
	return pointwise_launch (d_v, n, BinarizeBelow {T, one, zero}, stream);
	}

int gdsp_binarize_batch (const gdsp_batch_item* items, int nitems, double T, int tiesAbove, double one, double zero, void* stream)
	{
	if (tiesAbove) return pointwise_batch_launch (items, nitems, BinarizeAbove {T, one, zero}, stream);
	return pointwise_batch_launch (items, nitems, BinarizeBelow {T, one, zero}, stream);
	}

int gdsp_clip_batch (const gdsp_batch_item* items, int nitems, int haveMin, double minVal, int haveMax, double maxVal, void* stream)
	{
	GDSP_REQUIRE (haveMin || haveMax, "clip needs a minimum or a maximum");
	if (!haveMax) return pointwise_batch_launch (items, nitems, ClipMin {minVal}, stream);
	if (!haveMin) return pointwise_batch_launch (items, nitems, ClipMax {maxVal}, stream);
	return pointwise_batch_launch (items, nitems, ClipBoth {minVal, maxVal}, stream);
	}

int gdsp_erase_batch (const gdsp_batch_item* items, int nitems, int haveMin, double minVal, int haveMax, double maxVal,
                      int keepInside, double zero, void* stream)
	{
	GDSP_REQUIRE (haveMin || haveMax, "erase needs a minimum or a maximum");
	if (keepInside)
		{
		if (!haveMax) return pointwise_batch_launch (items, nitems, Erase<0> {minVal, maxVal, zero}, stream);
		if (!haveMin) return pointwise_batch_launch (items, nitems, Erase<1> {minVal, maxVal, zero}, stream);
		return pointwise_batch_launch (items, nitems, Erase<2> {minVal, maxVal, zero}, stream);
		}
	if (!haveMax) return pointwise_batch_launch (items, nitems, Erase<3> {minVal, maxVal, zero}, stream);
	if (!haveMin) return pointwise_batch_launch (items, nitems, Erase<4> {minVal, maxVal, zero}, stream);
	return pointwise_batch_launch (items, nitems, Erase<5> {minVal, maxVal, zero}, stream);
	}

int gdsp_add_constant_batch (const gdsp_batch_item* items, int nitems, double c, void* stream)
	{
	if (c == 0.0) return GDSP_OK;                       // add.c:736
	return pointwise_batch_launch (items, nitems, AddConst {c}, stream);
	}

int gdsp_abs_batch (const gdsp_batch_item* items, int nitems, void* stream)
	{ return pointwise_batch_launch (items, nitems, AbsVal {}, stream); }

int gdsp_clip (double* d_v, uint32_t n, int haveMin, double minVal, int haveMax, double maxVal, void* stream)
	{
	GDSP_REQUIRE (haveMin || haveMax, "clip needs a minimum or a maximum");
	if (!haveMax) return pointwise_launch (d_v, n, ClipMin {minVal}, stream);
	if (!haveMin) return pointwise_launch (d_v, n, ClipMax {maxVal}, stream);
	return pointwise_launch (d_v, n, ClipBoth {minVal, maxVal}, stream);
	}

int gdsp_erase (double* d_v, uint32_t n, int haveMin, double minVal, int haveMax, double maxVal,
                int keepInside, double zero, void* stream)
	{
	GDSP_REQUIRE (haveMin || haveMax, "erase needs a minimum or a maximum");
	if (keepInside)
		{
		if (!haveMax) return pointwise_launch (d_v, n, Erase<0> {minVal, maxVal, zero}, stream);
		if (!haveMin) return pointwise_launch (d_v, n, Erase<1> {minVal, maxVal, zero}, stream);
		return pointwise_launch (d_v, n, Erase<2> {minVal, maxVal, zero}, stream);
		}
	if (!haveMax) return pointwise_launch (d_v, n, Erase<3> {minVal, maxVal, zero}, stream);
	if (!haveMin) return pointwise_launch (d_v, n, Erase<4> {minVal, maxVal, zero}, stream);
	return pointwise_launch (d_v, n, Erase<5> {minVal, maxVal, zero}, stream);
	}

int gdsp_add_constant (double* d_v, uint32_t n, double c, void* stream)
	{
	if (c == 0.0) return GDSP_OK;                       // add.c:736
	return pointwise_launch (d_v, n, AddConst {c}, stream);
	}

int gdsp_abs (double* d_v, uint32_t n, void* stream)
	{ return pointwise_launch (d_v, n, AbsVal {}, stream); }

int gdsp_invert (double* d_v, uint32_t n, double mid, void* stream)
	{ return pointwise_launch (d_v, n, Invert {2*mid}, stream); }

int gdsp_fill (double* d_v, uint32_t n, double val, void* stream)
	{ return pointwise_launch (d_v, n, FillVal {val}, stream); }

int gdsp_minmax_init (double* d_minmax, void* stream)
	{
	GDSP_REQUIRE (d_minmax != NULL, "NULL result");
	static const double init[3] = { DBL_MAX, -DBL_MAX, 0.0 };
	GDSP_HIP_TRY (hipMemcpyAsync (d_minmax, init, sizeof(init), hipMemcpyHostToDevice, gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_minmax_update (const double* d_v, uint32_t n, uint32_t window, double lo, double hi,
                        double* d_minmax, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_v != NULL) && (d_minmax != NULL), "NULL pointer");
	if (window == 0) window = 1;
	if ((window == 1) && gdsp_aligned16 (d_v))
		{
		size_t   tiles  = ((size_t) n + PW_TILE - 1) / PW_TILE;
		uint32_t blocks = (uint32_t) (tiles > PW_MAX_BLOCKS? PW_MAX_BLOCKS : tiles);
		hipLaunchKernelGGL (minmax_dense_kernel, dim3(blocks), dim3(PW_THREADS), 0, gdsp_stream (stream), d_v, n, lo, hi, d_minmax);
		GDSP_LAUNCH_CHECK ();
		return GDSP_OK;
		}
	size_t   nsamp  = ((size_t) n + window - 1) / window;
	size_t   want   = (nsamp + (size_t) PW_THREADS*4 - 1) / ((size_t) PW_THREADS*4);
	uint32_t blocks = (uint32_t) (want < 1? 1 : (want > PW_MAX_BLOCKS? PW_MAX_BLOCKS : want));
	hipLaunchKernelGGL (minmax_kernel, dim3(blocks), dim3(PW_THREADS), 0, gdsp_stream (stream),
	                    d_v, n, window, lo, hi, d_minmax);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

} // extern "C"
