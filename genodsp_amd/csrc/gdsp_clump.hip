// gdsp_clump.hip -- clump / anticlump: intervals whose average is on the right side of a threshold.
//
// Reference: clump_search, clump.c:494-736 (Allison's maximal-average scan).  For every base ix the
// reference keeps one running sum P of (v - avg) and a monotone list of its record minima and marks
// the longest stretch ending at ix whose sum is >= 0 when it is at least minLength long; marked
// stretches are merged and each merged run is then trimmed to its first and last base on the right
// side of the threshold (clump.c:668-716).  That walk is sequential, but what it marks is a pure
// function of the prefix sums:
//     base t is marked  <=>  there are j < t <= i with i-j >= L and P[j] <= P[i]      (P[-1] = 0)
//                       <=>  Q[t-L-1] <= R[t]   or   good[j] for some j in [t-L, t-1]
// with Q = running minimum of P from the left, R = running maximum of P from the right and
// good[j] = (P[j] <= R[j+L]).  Every piece is a scan or a per-base test, so the operator becomes six
// whole-vector scans (sum, min, max, a count for the window-any, and two "last event" scans for the
// trimming) plus per-base kernels.  Comparisons are exact; the one rounding-sensitive part is P
// itself, which the reference sums left to right: results are bit-identical whenever P is exactly
// representable (read depth against an integer or dyadic threshold), like slidingsum.
// HBM-bound, ~200 B/base over all passes.

#include <math.h>
#include "gdsp_common.h"

#define SC_THREADS 256
#define SC_PER     16
#define SC_CHUNK   (SC_THREADS * SC_PER)

enum { SC_ADD = 0, SC_MIN = 1, SC_MAX = 2 };

template <int OP> __device__ __forceinline__ double sc_op (double a, double b)
	{ return (OP == SC_ADD)? a + b : ((OP == SC_MIN)? fmin (a, b) : fmax (a, b)); }
template <int OP> __device__ __forceinline__ double sc_identity ()
	{ return (OP == SC_ADD)? 0.0 : ((OP == SC_MIN)? INFINITY : -INFINITY); }

// logical element k of the scan is physical element k (forward) or n-1-k (reverse)
template <bool REVERSE> __device__ __forceinline__ size_t sc_at (size_t k, size_t n) { return REVERSE? n - 1 - k : k; }

template <int OP>
__device__ __forceinline__ double sc_block_reduce (double x, double* part)
	{
	for (int off=32 ; off>0 ; off>>=1) x = sc_op<OP> (x, __shfl_down (x, off, 64));
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = x;
	__syncthreads ();
	return sc_op<OP> (sc_op<OP> (part[0], part[1]), sc_op<OP> (part[2], part[3]));
	}

// Both passes read and write lane-consecutive elements (a chunk of a reversed scan simply runs down in
// memory), sixteen accesses per lane in flight; the apply pass turns the chunk in LDS (pitch 17 per 16) so
// that a thread scans sixteen consecutive elements.
template <int OP, bool REVERSE>
__global__ __launch_bounds__(SC_THREADS)
void scan_totals_kernel (const double* __restrict__ v, size_t n, double* __restrict__ totals)
	{
	__shared__ double part[SC_THREADS/64];
	const size_t k0 = (size_t) blockIdx.x * SC_CHUNK;
	double x[SC_PER];
#pragma unroll
	for (int i=0 ; i<SC_PER ; i++)
		{
		const size_t k = k0 + (size_t) i * SC_THREADS + threadIdx.x;
		x[i] = v[sc_at<REVERSE> ((k < n)? k : n-1, n)];
		}
	double acc = sc_identity<OP> ();
#pragma unroll
	for (int i=0 ; i<SC_PER ; i++)
		{
		const size_t k = k0 + (size_t) i * SC_THREADS + threadIdx.x;
		if (k < n) acc = sc_op<OP> (acc, x[i]);
		}
	acc = sc_block_reduce<OP> (acc, part);
	if (threadIdx.x == 0) totals[blockIdx.x] = acc;
	}

template <int OP>
__global__ __launch_bounds__(1024)
void scan_offsets_kernel (double* __restrict__ totals, uint32_t nchunks)
	{
	__shared__ double sums[1024];
	const uint32_t per = (nchunks + 1023) / 1024;
	const uint32_t a = threadIdx.x * per, b = (a + per < nchunks)? a + per : ((a < nchunks)? nchunks : a);
	double acc = sc_identity<OP> ();
	for (uint32_t i=a ; i<b ; i++) acc = sc_op<OP> (acc, totals[i]);
	sums[threadIdx.x] = acc;
	__syncthreads ();
	for (int d=1 ; d<1024 ; d*=2)
		{
		double up = ((int) threadIdx.x >= d)? sums[threadIdx.x - d] : sc_identity<OP> ();
		__syncthreads ();
		sums[threadIdx.x] = sc_op<OP> (up, sums[threadIdx.x]);
		__syncthreads ();
		}
	double run = (threadIdx.x > 0)? sums[threadIdx.x - 1] : sc_identity<OP> ();   // exclusive
	for (uint32_t i=a ; i<b ; i++) { double t = totals[i];  totals[i] = run;  run = sc_op<OP> (run, t); }
	}

template <int OP, bool REVERSE>
__global__ __launch_bounds__(SC_THREADS)
void scan_apply_kernel (double* __restrict__ v, size_t n, const double* __restrict__ offsets)
	{
	__shared__ double turn[SC_THREADS * (SC_PER + 1)];
	__shared__ double waveTot[SC_THREADS/64];
	const size_t k0   = (size_t) blockIdx.x * SC_CHUNK;
	const int    lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double x[SC_PER];
#pragma unroll
	for (int i=0 ; i<SC_PER ; i++)
		{
		const size_t k = k0 + (size_t) i * SC_THREADS + threadIdx.x;
		x[i] = v[sc_at<REVERSE> ((k < n)? k : n-1, n)];
		}
#pragma unroll
	for (int i=0 ; i<SC_PER ; i++)
		{
		const int    kk = i * SC_THREADS + (int) threadIdx.x;         // place in the chunk, scan order
		const size_t k  = k0 + kk;
		turn[kk + (kk >> 4)] = (k < n)? x[i] : sc_identity<OP> ();
		}
	__syncthreads ();

	double* mine = turn + threadIdx.x * (SC_PER + 1);
	double  run  = sc_identity<OP> ();
#pragma unroll
	for (int i=0 ; i<SC_PER ; i++) { run = sc_op<OP> (run, mine[i]);  x[i] = run; }
	double incl = run;
	for (int d=1 ; d<64 ; d*=2)
		{
		const double up = __shfl_up (incl, d, 64);
		if (lane >= d) incl = sc_op<OP> (up, incl);
		}
	double excl = __shfl_up (incl, 1, 64);
	if (lane == 0) excl = sc_identity<OP> ();
	if (lane == 63) waveTot[wave] = incl;
	__syncthreads ();
	double before = offsets[blockIdx.x];
	for (int w=0 ; w<wave ; w++) before = sc_op<OP> (before, waveTot[w]);
	before = sc_op<OP> (before, excl);
#pragma unroll
	for (int i=0 ; i<SC_PER ; i++) mine[i] = sc_op<OP> (before, x[i]);
	__syncthreads ();
#pragma unroll
	for (int i=0 ; i<SC_PER ; i++)
		{
		const int    kk = i * SC_THREADS + (int) threadIdx.x;
		const size_t k  = k0 + kk;
		if (k < n) v[sc_at<REVERSE> (k, n)] = turn[kk + (kk >> 4)];
		}
	}

// in-place inclusive scan of v[0..n) with OP, left to right or right to left
template <int OP, bool REVERSE>
static void scan_inplace (double* v, size_t n, double* totals, hipStream_t s)
	{
	const uint32_t nchunks = (uint32_t) ((n + SC_CHUNK - 1) / SC_CHUNK);
	hipLaunchKernelGGL ((scan_totals_kernel<OP, REVERSE>), dim3(nchunks), dim3(SC_THREADS), 0, s, v, n, totals);
	hipLaunchKernelGGL ((scan_offsets_kernel<OP>),         dim3(1),       dim3(1024),       0, s, totals, nchunks);
	hipLaunchKernelGGL ((scan_apply_kernel<OP, REVERSE>),  dim3(nchunks), dim3(SC_THREADS), 0, s, v, n, totals);
	}

// ------------------------------------------------------------------ per-base passes ----
__global__ void clump_values_kernel (const double* __restrict__ v, double* __restrict__ P, size_t n, double avg, int above)
	{
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x ; i < n ; i += stride)
		P[i] = above? v[i] - avg : avg - v[i];                       // clump.c:583-584
	}

// Q = min(running minimum of P, 0) ; R = running maximum of P from the right
__global__ void clump_copy2_kernel (const double* __restrict__ P, double* __restrict__ Q, double* __restrict__ R, size_t n)
	{
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x ; i < n ; i += stride) { Q[i] = P[i];  R[i] = P[i]; }
	}

// G[k], k in [0,n]: good[j] for j = k-1 in [-1, n-1]:  P[j] <= R[j+L]  (P[-1] = 0)
__global__ void clump_good_kernel (const double* __restrict__ P, const double* __restrict__ R, double* __restrict__ G,
                                   size_t n, uint64_t L)
	{
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x ; k <= n ; k += stride)
		{
		const uint64_t idx = (uint64_t) k + L - 1;                   // j + L with j = k-1
		const double   pj  = (k == 0)? 0.0 : P[k-1];
		G[k] = ((idx < n) && (pj <= R[idx]))? 1.0 : 0.0;
		}
	}

// "last event" keys: not marked -> even key, marked and on the right side of the threshold -> odd key,
// marked but on the wrong side -> no event.  Kf is scanned left to right, Kr right to left.
__global__ void clump_keys_kernel (const double* __restrict__ v, const double* __restrict__ Q, const double* __restrict__ R,
                                   const double* __restrict__ C, double* __restrict__ Kf, double* __restrict__ Kr,
                                   size_t n, uint64_t L, double avg, int above)
	{
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for (size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x ; t < n ; t += stride)
		{
		bool marked = false;
		if (t + 1 >= L + 1)                                          // t-L-1 >= -1
			{
			const double q = (t == L)? 0.0 : fmin (Q[t - L - 1], 0.0);   // running minimum including P[-1] = 0
			marked = (q <= R[t]);
			}
		if (!marked)
			{
			const double hi = C[t], lo = (t >= L)? C[t - L] : 0.0;     // good[j], j in [max(-1,t-L), t-1]
			marked = (hi - lo > 0.0);
			}
		const bool onSide = above? (v[t] >= avg) : (v[t] <= avg);      // clump.c:687-688
		const double tf = 2.0 * (double) t, tr = 2.0 * (double) (n - 1 - t);
		Kf[t] = !marked? tf : (onSide? tf + 1.0 : -1.0);
		Kr[t] = !marked? tr : (onSide? tr + 1.0 : -1.0);
		}
	}

__global__ void clump_write_kernel (double* __restrict__ v, const double* __restrict__ Kf, const double* __restrict__ Kr,
                                    size_t n, double one, double zero)
	{
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for (size_t t = (size_t) blockIdx.x * blockDim.x + threadIdx.x ; t < n ; t += stride)
		{
		const double a = Kf[t], b = Kr[t];
		const bool in = (a >= 0.0) && (b >= 0.0) && (fmod (a, 2.0) == 1.0) && (fmod (b, 2.0) == 1.0);
		v[t] = in? one : zero;
		}
	}

extern "C" {

size_t gdsp_clump_work (uint32_t n)
	{ return (5 * ((size_t) n + 4) + ((size_t) n + SC_CHUNK) / SC_CHUNK + 8) * sizeof(double); }

/* clump (above != 0) / anticlump: in place; d_work >= gdsp_clump_work(n) bytes */
int gdsp_clump (double* d_v, uint32_t n, double average, uint32_t minLength, int above,
                double one, double zero, void* d_work, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_v != NULL) && (d_work != NULL), "NULL pointer");
	hipStream_t   s = gdsp_stream (stream);
	const size_t  N = n, pitch = (N + 4) & ~(size_t) 1;
	const uint64_t L = (minLength < 1)? 1 : minLength;
	double* P  = (double*) d_work;          // later Kf
	double* Q  = P + pitch;
	double* R  = Q + pitch;
	double* G  = R + pitch;                 // n+1 entries; becomes C
	double* Kr = G + pitch;
	double* totals = Kr + pitch;
	const uint32_t blocks = (uint32_t) ((N + 1023) / 1024 > 4096? 4096 : (N + 1023) / 1024);

	hipLaunchKernelGGL (clump_values_kernel, dim3(blocks), dim3(256), 0, s, d_v, P, N, average, above);
	scan_inplace<SC_ADD, false> (P, N, totals, s);                        // P = prefix sums
	hipLaunchKernelGGL (clump_copy2_kernel, dim3(blocks), dim3(256), 0, s, P, Q, R, N);
	scan_inplace<SC_MIN, false> (Q, N, totals, s);                        // running minimum from the left
	scan_inplace<SC_MAX, true>  (R, N, totals, s);                        // running maximum from the right
	hipLaunchKernelGGL (clump_good_kernel, dim3(blocks), dim3(256), 0, s, P, R, G, N, L);
	scan_inplace<SC_ADD, false> (G, N + 1, totals, s);                    // C = running count of good[]
	double* Kf = P;
	hipLaunchKernelGGL (clump_keys_kernel, dim3(blocks), dim3(256), 0, s, d_v, Q, R, G, Kf, Kr, N, L, average, above);
	scan_inplace<SC_MAX, false> (Kf, N, totals, s);                       // last event at or before t
	scan_inplace<SC_MAX, true>  (Kr, N, totals, s);                       // first event at or after t
	hipLaunchKernelGGL (clump_write_kernel, dim3(blocks), dim3(256), 0, s, d_v, Kf, Kr, N, one, zero);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

} // extern "C"
