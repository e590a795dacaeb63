// gdsp_report.hip -- run-length encoding of a chromosome vector on the device.
//
// Reference: report_intervals, genodsp.c:1561-1691 (also behind the `output`
// operator, opio.c:454-482).  The reference walks every base on the host and
// prints "chrom start end value" whenever the value changes.  With the signal in
// HBM, copying 8 B/base back to format a few MB of text would dominate the whole
// pipeline, so the runs are found here and only (start, end, value) triples cross
// PCIe.
//
// The reference's state machine reduces to two per-base predicates:
//   reportable(j) : uncovered==show, or v[j] != 0            (genodsp.c:1597)
//   startsRun(j)  : reportable(j) and (j==0 or !reportable(j-1) or !collapse
//                   or !(v[j]==v[j-1]))                      (genodsp.c:1624-1656)
//   endsRun(j)    : reportable(j) and (j==n-1 or !reportable(j+1) or !collapse
//                   or !(v[j+1]==v[j]))
// The k-th start pairs with the k-th end (+1, half-open).  Equality is exact
// `==` on doubles as in the reference (NaN never collapses, -0.0 is a zero).
// Three launches: per-tile counts, a scan of the counts, and the compaction.
// HBM-bound: 2 x 8 B/base read (the second pass mostly from the Infinity Cache
// for chromosomes below 256 MB), a few bytes per run written.

#include "gdsp_common.h"

#define RP_THREADS 256
#define RP_PER     16
#define RP_TILE    (RP_THREADS * RP_PER)      // 4096 bases per workgroup

struct RpFlags { uint32_t starts, ends; };    // bit k = base (first + k)

// flags for the RP_PER consecutive bases owned by this thread
__device__ __forceinline__ RpFlags rp_flags (const double* __restrict__ v, uint32_t n, uint64_t first,
                                             int collapse, int show, double (&x)[RP_PER+2])
	{
	// x[0] = v[first-1], x[1..RP_PER] = owned bases, x[RP_PER+1] = v[first+RP_PER]
	if ((first >= 2) && (first + RP_PER + 2 <= n) && ((first & 1) == 0))
		{
		// aligned interior: nine 16-byte loads starting at first-2
		const double2* p = reinterpret_cast<const double2*> (v + first - 2);
		double2 d = p[0];                                       // (plain loads: a lane walks its own strip, the rest of each line has to wait in the cache)
		x[0] = d.y;
#pragma unroll
		for (int k=0 ; k<RP_PER/2 ; k++) { d = p[k+1];  x[1+2*k] = d.x;  x[2+2*k] = d.y; }
		d = p[RP_PER/2 + 1];
		x[RP_PER+1] = d.x;
		}
	else
		{
#pragma unroll
		for (int k=0 ; k<RP_PER+2 ; k++)
			{
			int64_t g = (int64_t) first - 1 + k;
			x[k] = ((g >= 0) && (g < (int64_t) n))? v[g] : 0.0;
			}
		}
	RpFlags f = { 0, 0 };
#pragma unroll
	for (int k=0 ; k<RP_PER ; k++)
		{
		const uint64_t j = first + k;
		if (j >= n) break;
		const double c = x[k+1];
		const bool rep  = show || !(c == 0);
		if (!rep) continue;
		const bool repL = (j > 0)     && (show || !(x[k]   == 0));
		const bool repR = (j + 1 < n) && (show || !(x[k+2] == 0));
		if (!repL || !collapse || !(c == x[k]))   f.starts |= 1u << k;
		if (!repR || !collapse || !(x[k+2] == c)) f.ends   |= 1u << k;
		}
	return f;
	}

// exclusive scan of (a,b) pairs over the workgroup; returns this thread's offsets and the totals
__device__ __forceinline__ void rp_block_scan (uint32_t a, uint32_t b, uint32_t& offA, uint32_t& offB,
                                               uint32_t& totA, uint32_t& totB, uint32_t (*waveTot)[2])
	{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t ia = a, ib = b;
	for (int d=1 ; d<64 ; d*=2)
		{
		uint32_t ua = __shfl_up (ia, d, 64), ub = __shfl_up (ib, d, 64);
		if (lane >= d) { ia += ua;  ib += ub; }
		}
	if (lane == 63) { waveTot[wave][0] = ia;  waveTot[wave][1] = ib; }
	__syncthreads ();
	offA = ia - a;  offB = ib - b;  totA = 0;  totB = 0;
	for (int w=0 ; w<RP_THREADS/64 ; w++)
		{
		if (w < wave) { offA += waveTot[w][0];  offB += waveTot[w][1]; }
		totA += waveTot[w][0];  totB += waveTot[w][1];
		}
	}

__global__ __launch_bounds__(RP_THREADS)
void report_count_kernel (const double* __restrict__ v, uint32_t n, int collapse, int show,
                          uint32_t* __restrict__ counts)
	{
	__shared__ uint32_t waveTot[RP_THREADS/64][2];
	double x[RP_PER+2];
	const uint64_t first = (uint64_t) blockIdx.x * RP_TILE + (uint64_t) threadIdx.x * RP_PER;
	RpFlags f = { 0, 0 };
	if (first < n) f = rp_flags (v, n, first, collapse, show, x);
	uint32_t offA, offB, totA, totB;
	rp_block_scan (__builtin_popcount (f.starts), __builtin_popcount (f.ends), offA, offB, totA, totB, waveTot);
	if (threadIdx.x == 0) { counts[2*blockIdx.x] = totA;  counts[2*blockIdx.x+1] = totB; }
	}

// exclusive scan of the per-tile (starts, ends) counts; one workgroup walking tiles of 4096 count pairs, four consecutive
// pairs per thread (32 contiguous bytes per lane: the loads coalesce; a thread walking its own far-apart slice of the array
// took 79 us for 35 k tiles), lanes, then waves, then a carry from tile to tile
__global__ __launch_bounds__(1024)
void report_scan_kernel (uint32_t* __restrict__ counts, uint32_t ntiles, uint32_t* __restrict__ total)
	{
	__shared__ uint32_t partA[16], partB[16];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t carryA = 0, carryB = 0;
	for (uint32_t t0=0 ; t0<ntiles ; t0+=4096)
		{
		const uint32_t c0 = t0 + threadIdx.x * 4;
		uint32_t a[4], b[4];
		if (c0 + 4 <= ntiles)
			{
			const uint4 lo = *reinterpret_cast<const uint4*> (counts + 2*c0), hi = *reinterpret_cast<const uint4*> (counts + 2*c0 + 4);
			a[0] = lo.x;  b[0] = lo.y;  a[1] = lo.z;  b[1] = lo.w;  a[2] = hi.x;  b[2] = hi.y;  a[3] = hi.z;  b[3] = hi.w;
			}
		else
			{
#pragma unroll
			for (int i=0 ; i<4 ; i++) { const bool in = (c0 + i < ntiles);  a[i] = in? counts[2*(c0+i)] : 0;  b[i] = in? counts[2*(c0+i)+1] : 0; }
			}
		const uint32_t sumA = a[0] + a[1] + a[2] + a[3], sumB = b[0] + b[1] + b[2] + b[3];
		uint32_t inclA = sumA, inclB = sumB;
		for (int d=1 ; d<64 ; d*=2)
			{
			const uint32_t ua = __shfl_up (inclA, d, 64), ub = __shfl_up (inclB, d, 64);
			if (lane >= d) { inclA += ua;  inclB += ub; }
			}
		if (lane == 63) { partA[wave] = inclA;  partB[wave] = inclB; }
		__syncthreads ();
		uint32_t ra = carryA + inclA - sumA, rb = carryB + inclB - sumB, allA = carryA, allB = carryB;
		for (int w=0 ; w<16 ; w++)
			{
			if (w < wave) { ra += partA[w];  rb += partB[w]; }
			allA += partA[w];  allB += partB[w];
			}
		__syncthreads ();
		carryA = allA;  carryB = allB;
#pragma unroll
		for (int i=0 ; i<4 ; i++)
			{
			if (c0 + i < ntiles) { counts[2*(c0+i)] = ra;  counts[2*(c0+i)+1] = rb; }
			ra += a[i];  rb += b[i];
			}
		}
	if (threadIdx.x == 0) *total = carryA;
	}

__global__ __launch_bounds__(RP_THREADS)
void report_write_kernel (const double* __restrict__ v, uint32_t n, int collapse, int show,
                          const uint32_t* __restrict__ offsets,
                          uint32_t* __restrict__ runStart, uint32_t* __restrict__ runEnd,
                          double* __restrict__ runVal, uint32_t cap)
	{
	__shared__ uint32_t waveTot[RP_THREADS/64][2];
	double x[RP_PER+2];
	const uint64_t first = (uint64_t) blockIdx.x * RP_TILE + (uint64_t) threadIdx.x * RP_PER;
	RpFlags f = { 0, 0 };
	if (first < n) f = rp_flags (v, n, first, collapse, show, x);
	uint32_t offA, offB, totA, totB;
	rp_block_scan (__builtin_popcount (f.starts), __builtin_popcount (f.ends), offA, offB, totA, totB, waveTot);
	uint32_t ka = offsets[2*blockIdx.x] + offA, kb = offsets[2*blockIdx.x+1] + offB;
#pragma unroll
	for (int k=0 ; k<RP_PER ; k++)
		{
		if (f.starts & (1u << k))
			{
			if (ka < cap) { runStart[ka] = (uint32_t) (first + k);  runVal[ka] = x[k+1]; }
			ka++;
			}
		if (f.ends & (1u << k))
			{
			if (kb < cap) runEnd[kb] = (uint32_t) (first + k + 1);
			kb++;
			}
		}
	}

extern "C" {

size_t gdsp_report_runs_work (uint32_t n)
	{ return (2 * (((size_t) n + RP_TILE - 1) / RP_TILE) + 2) * sizeof(uint32_t); }

int gdsp_report_runs (const double* d_v, uint32_t n, int collapse, int uncovered,
                      uint32_t* d_runStart, uint32_t* d_runEnd, double* d_runVal, uint32_t cap,
                      uint32_t* d_count, void* d_work, void* stream)
	{
	GDSP_REQUIRE (d_count != NULL, "NULL count");
	hipStream_t s = gdsp_stream (stream);
	if (n == 0) { GDSP_HIP_TRY (hipMemsetAsync (d_count, 0, sizeof(uint32_t), s));  return GDSP_OK; }
	GDSP_REQUIRE ((d_v != NULL) && (d_work != NULL), "NULL pointer");
	GDSP_REQUIRE (gdsp_aligned16 (d_v), "vector must be 16-byte aligned");
	GDSP_REQUIRE ((cap == 0) || ((d_runStart != NULL) && (d_runEnd != NULL) && (d_runVal != NULL)), "NULL run arrays");
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + RP_TILE - 1) / RP_TILE);
	const int      show   = (uncovered == 1);            // NA (-1) hides zeros too, genodsp.c:1597
	uint32_t*      counts = (uint32_t*) d_work;
	hipLaunchKernelGGL (report_count_kernel, dim3(ntiles), dim3(RP_THREADS), 0, s, d_v, n, collapse, show, counts);
	hipLaunchKernelGGL (report_scan_kernel,  dim3(1),      dim3(1024),       0, s, counts, ntiles, d_count);
	if (cap != 0)
		hipLaunchKernelGGL (report_write_kernel, dim3(ntiles), dim3(RP_THREADS), 0, s, d_v, n, collapse, show,
		                    counts, d_runStart, d_runEnd, d_runVal, cap);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

} // extern "C"
