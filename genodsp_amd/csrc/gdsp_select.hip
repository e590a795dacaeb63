// gdsp_select.hip -- exact order statistics for the `percentile` operator.
//
// Reference: op_percentile_apply, percentile.c:392-751.  The reference shuffles
// the sampled values to the front of the genome and sorts them in place with
// qsort (destroying the signal, percentile.c:34-36) to read off the k-th
// smallest.  Here the k-th smallest is found by radix select on the
// order-preserving 64-bit image of the doubles, leaving the signal untouched:
// each pass histograms one digit of the keys that still match the digits chosen
// so far; the histograms of all chromosomes (and all GPUs -- the path's only
// collective, a sum over ranks) are added, and the host walks the counts to the
// bucket that holds rank k.  The sample and its filter are the reference's:
// every window-th value with  !(v < lo) && !(v > hi)  (percentile.c:559-561).
//
// One pass reads the sample once (8 B per sampled base): HBM-bound.  Read depth
// is piecewise constant and 35 % zeros, so a naive LDS histogram would serialise
// on a handful of hot bins.  The dense kernel (window 1) gives every lane eight
// consecutive bases and one LDS atomic per run of equal digits; the strided
// kernel (window > 1) peels each wave's two most common digits with a ballot.
// Along with the counts a pass records the smallest and largest matching key:
// when they coincide every remaining candidate is the same value and the later
// passes are skipped (typically after three of five passes on integer depth).
// Measured: 4.8 TB/s per pass on read depth, 6.4 TB/s on real-valued signals.

#include "gdsp_common.h"

#define SE_THREADS    256
#define SE_MAX_BLOCKS 2048
#define SE_MAX_BITS   13

__global__ __launch_bounds__(SE_THREADS)
void select_hist_kernel (const double* __restrict__ v, uint32_t n, uint32_t window, double lo, double hi,
                         int shift, int bits, uint64_t prefix, unsigned long long* __restrict__ hist)
	{
	__shared__ uint32_t bins[1 << SE_MAX_BITS];
	const int      nbins = 1 << bits;
	const uint64_t mask  = (uint64_t) nbins - 1;
	const int      above = shift + bits;                 // key bits [above,64) must equal the prefix's
	for (int b=threadIdx.x ; b<nbins ; b+=SE_THREADS) bins[b] = 0;
	__syncthreads ();

	const size_t nsamp  = ((size_t) n + window - 1) / window;
	const size_t stride = (size_t) gridDim.x * SE_THREADS;
	const int    lane   = threadIdx.x & 63;
	uint64_t     kmin = ~0ULL, kmax = 0;

	// every lane of a wave runs the same number of trips so the ballots stay convergent
	const size_t first = (size_t) blockIdx.x * SE_THREADS + threadIdx.x;
	for (size_t s0 = first - lane ; s0 < nsamp ; s0 += stride)
		{
		const size_t s = s0 + lane;
		bool     live = false;
		uint32_t bin  = 0;
		if (s < nsamp)
			{
			const double x = v[s * window];
			if (!(x < lo) && !(x > hi))
				{
				const uint64_t key = gdsp_key_of (x);
				if ((above >= 64) || ((key >> above) == (prefix >> above)))
					{
					live = true;
					bin  = (uint32_t) ((key >> shift) & mask);
					if (key < kmin) kmin = key;
					if (key > kmax) kmax = key;
					}
				}
			}
		// peel the two most common digits of this wave
#pragma unroll
		for (int round=0 ; round<2 ; round++)
			{
			const uint64_t active = __ballot (live);
			if (active == 0) break;
			const int      leader = __builtin_ctzll (active);
			const uint32_t b0     = __shfl (bin, leader, 64);
			const uint64_t same   = __ballot (live && (bin == b0));
			if (lane == leader) atomicAdd (&bins[b0], (uint32_t) __builtin_popcountll (same));
			if (bin == b0) live = false;
			}
		if (live) atomicAdd (&bins[bin], 1u);
		}
	__syncthreads ();

	for (int b=threadIdx.x ; b<nbins ; b+=SE_THREADS)
		{ uint32_t c = bins[b];  if (c) atomicAdd (&hist[b], (unsigned long long) c); }

	for (int off=32 ; off>0 ; off>>=1)
		{
		uint64_t a = __shfl_down ((unsigned long long) kmin, off, 64);
		uint64_t b = __shfl_down ((unsigned long long) kmax, off, 64);
		if (a < kmin) kmin = a;
		if (b > kmax) kmax = b;
		}
	if ((lane == 0) && (kmin <= kmax))
		{
		atomicMin (&hist[nbins],   (unsigned long long) kmin);
		atomicMax (&hist[nbins+1], (unsigned long long) kmax);
		}
	}

// Dense form (window == 1, the usual case): every lane takes 8 CONSECUTIVE bases per step
// (four 16-byte loads), so on read-depth-like signals -- runs of ~100 equal values -- a
// lane's eight digits are almost always one run, and the lane issues ONE LDS atomic for
// all eight.  That removes the same-address serialisation of the LDS histogram without any
// cross-lane work; on random digits it degenerates to one atomic per base on random banks.
#define SE_PER 8
__global__ __launch_bounds__(SE_THREADS)
void select_hist_dense_kernel (const double* __restrict__ v, uint32_t n, double lo, double hi,
                               int shift, int bits, uint64_t prefix, unsigned long long* __restrict__ hist)
	{
	__shared__ uint32_t bins[1 << SE_MAX_BITS];
	const int      nbins = 1 << bits;
	const uint64_t mask  = (uint64_t) nbins - 1;
	const int      above = shift + bits;
	for (int b=threadIdx.x ; b<nbins ; b+=SE_THREADS) bins[b] = 0;
	__syncthreads ();

	const int    lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t step = (size_t) gridDim.x * SE_THREADS * SE_PER;
	uint64_t     kmin = ~0ULL, kmax = 0;

	for (size_t wbase = ((size_t) blockIdx.x * (SE_THREADS/64) + wave) * 64 * SE_PER ; wbase < n ; wbase += step)
		{
		const size_t g = wbase + (size_t) lane * SE_PER;
		double x[SE_PER];
		if (wbase + 64*SE_PER <= (size_t) n)                 // wave-uniform: whole chunk inside
			{
			const double2* p = reinterpret_cast<const double2*> (v + g);
#pragma unroll
			for (int i=0 ; i<SE_PER/2 ; i++) { double2 d = p[i];  x[2*i] = d.x;  x[2*i+1] = d.y; }   // (plain loads: a lane walks its own strip, the rest of each line has to wait in the cache)
			}
		else
			{
#pragma unroll
			for (int i=0 ; i<SE_PER ; i++) x[i] = (g + i < (size_t) n)? v[g+i] : 0.0;
			}

		uint32_t runBin = 0xFFFFFFFFu, runCnt = 0;
#pragma unroll
		for (int i=0 ; i<SE_PER ; i++)
			{
			if (g + i >= (size_t) n) break;
			const double xi = x[i];
			if ((xi < lo) || (xi > hi)) continue;
			const uint64_t key = gdsp_key_of (xi);
			if ((above < 64) && ((key >> above) != (prefix >> above))) continue;
			const uint32_t bin = (uint32_t) ((key >> shift) & mask);
			if (key < kmin) kmin = key;
			if (key > kmax) kmax = key;
			if (bin == runBin) runCnt++;
			else
				{
				if (runCnt) atomicAdd (&bins[runBin], runCnt);
				runBin = bin;  runCnt = 1;
				}
			}
		if (runCnt) atomicAdd (&bins[runBin], runCnt);
		}
	__syncthreads ();

	for (int b=threadIdx.x ; b<nbins ; b+=SE_THREADS)
		{ uint32_t c = bins[b];  if (c) atomicAdd (&hist[b], (unsigned long long) c); }
	for (int off=32 ; off>0 ; off>>=1)
		{
		uint64_t a = __shfl_down ((unsigned long long) kmin, off, 64);
		uint64_t b = __shfl_down ((unsigned long long) kmax, off, 64);
		if (a < kmin) kmin = a;
		if (b > kmax) kmax = b;
		}
	if ((lane == 0) && (kmin <= kmax))
		{
		atomicMin (&hist[nbins],   (unsigned long long) kmin);
		atomicMax (&hist[nbins+1], (unsigned long long) kmax);
		}
	}

extern "C" {

// d_hist layout: (1<<bits) counts, then the smallest and the largest matching key
int gdsp_select_hist_init (uint64_t* d_hist, int bits, void* stream)
	{
	GDSP_REQUIRE (d_hist != NULL, "NULL histogram");
	GDSP_REQUIRE ((bits >= 1) && (bits <= SE_MAX_BITS), "bits must be 1..13");
	const size_t nbins = (size_t) 1 << bits;
	GDSP_HIP_TRY (hipMemsetAsync (d_hist, 0, (nbins + 2) * sizeof(uint64_t), gdsp_stream (stream)));
	GDSP_HIP_TRY (hipMemsetAsync (d_hist + nbins, 0xFF, sizeof(uint64_t), gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_select_histogram (const double* d_v, uint32_t n, uint32_t window, double lo, double hi,
                           int shift, int bits, uint64_t prefix, uint64_t* d_hist, void* stream)
	{
	GDSP_REQUIRE (d_hist != NULL, "NULL histogram");
	GDSP_REQUIRE ((bits >= 1) && (bits <= SE_MAX_BITS), "bits must be 1..13");
	GDSP_REQUIRE ((shift >= 0) && (shift + bits <= 64), "digit outside the 64-bit key");
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE (d_v != NULL, "NULL vector");
	if (window == 0) window = 1;
	if ((window == 1) && gdsp_aligned16 (d_v))
		{
		size_t   want   = ((size_t) n + (size_t) SE_THREADS*SE_PER*4 - 1) / ((size_t) SE_THREADS*SE_PER*4);
		uint32_t blocks = (uint32_t) (want < 1? 1 : (want > SE_MAX_BLOCKS? SE_MAX_BLOCKS : want));
		hipLaunchKernelGGL (select_hist_dense_kernel, dim3(blocks), dim3(SE_THREADS), 0, gdsp_stream (stream),
		                    d_v, n, lo, hi, shift, bits, prefix, (unsigned long long*) d_hist);
		GDSP_LAUNCH_CHECK ();
		return GDSP_OK;
		}
	size_t   nsamp  = ((size_t) n + window - 1) / window;
	size_t   want   = (nsamp + (size_t) SE_THREADS*8 - 1) / ((size_t) SE_THREADS*8);
	uint32_t blocks = (uint32_t) (want < 1? 1 : (want > SE_MAX_BLOCKS? SE_MAX_BLOCKS : want));
	hipLaunchKernelGGL (select_hist_kernel, dim3(blocks), dim3(SE_THREADS), 0, gdsp_stream (stream),
	                    d_v, n, window, lo, hi, shift, bits, prefix, (unsigned long long*) d_hist);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

// Host: walk the summed counts to the bucket holding 0-based rank k.
int gdsp_select_pick (const uint64_t* h_hist, int bits, uint64_t k, uint32_t* bucket, uint64_t* kWithin)
	{
	GDSP_REQUIRE ((h_hist != NULL) && (bucket != NULL) && (kWithin != NULL), "NULL pointer");
	GDSP_REQUIRE ((bits >= 1) && (bits <= SE_MAX_BITS), "bits must be 1..13");
	const uint32_t nbins = 1u << bits;
	uint64_t seen = 0;
	for (uint32_t b=0 ; b<nbins ; b++)
		{
		if (k < seen + h_hist[b]) { *bucket = b;  *kWithin = k - seen;  return GDSP_OK; }
		seen += h_hist[b];
		}
	gdsp_set_error ("gdsp_select_pick: rank %llu is beyond the %llu values counted",
	                (unsigned long long) k, (unsigned long long) seen);
	return GDSP_EINVAL;
	}

double   gdsp_key_to_double (uint64_t key) { return gdsp_value_of (key); }
uint64_t gdsp_double_to_key (double v)     { return gdsp_key_of (v); }

// percentile.c:587-589 / :681:  k = (u32) ((u64) numValues * pt / (100.0*1000)),
// and a rank equal to numValues means the largest value (:688-710)
uint32_t gdsp_percentile_rank (uint32_t numValues, uint32_t pThousandths)
	{
	uint32_t k = (uint32_t) (((uint64_t) numValues) * pThousandths / (100.0*1000));
	if ((numValues != 0) && (k >= numValues)) k = numValues - 1;
	return k;
	}

} // extern "C"
