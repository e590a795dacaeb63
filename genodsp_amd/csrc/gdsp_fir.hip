// gdsp_fir.hip -- zero-padded direct FIR over one chromosome (the `smooth` operator).
//
// Replaces the hot loop of op_smooth_apply, sum.c:651-663 in the reference:
//     out[ix] = sum over taps k (ascending) of taps[k] * v[ix - hOff + k]
// with taps outside [0,n) skipped.  Skipping a tap and adding taps[k]*(+0.0) give
// the same bits (the running sum starts at +0.0 and can never become -0.0), so
// the kernels stage a zero-padded tile and run the full tap range everywhere.
//
// Shape of the work on MI355X.  W=101 taps on f64 is 101 multiply-adds against
// 16 bytes per base: the FP64 vector pipe (16 lanes/clk/SIMD), not HBM, is the
// first limit, so the kernel is organised around keeping that pipe issuing:
//   * one workgroup = 256 threads = one tile of 256*R consecutive outputs,
//     staged once into LDS with its (W-1) halo by 16-byte coalesced loads;
//   * each thread owns R *consecutive* outputs and walks its R+W-1 inputs once:
//     one ds_read_b64 feeds up to R multiply-adds, so LDS traffic is ~1/R of the
//     arithmetic.  R is odd: lane stride = R doubles = 2R dwords, and 2R*t mod 64
//     is distinct for the 32 lanes of a ds_read_b64 group only when R is odd --
//     no padding, no swizzle, the LDS image stays linear;
//   * taps are wave-uniform: they sit in the kernarg segment and reach the VALU
//     as SGPR operands, costing no vector registers and no LDS reads;
//   * results go back through LDS so that the global store is 16 bytes per lane,
//     fully coalesced, instead of R strided 8-byte stores per lane.
// Measured on MI355X (profiles/r01_*): the FP64 pipe is busy 86 % of the kernel's
// cycles in FMA mode and the chip holds ~2.0 GHz under this load, i.e. the
// kernel runs at the FP64 vector rate the part sustains, at half the HBM peak.
// Persistent forms (taps in VGPRs at 3 waves/SIMD, or taps in SGPRs with register
// prefetch of the next tile at 5 waves/SIMD) were 7-20 % slower: the FMA stream
// wants all 8 waves/SIMD; 7, 9 or 11 outputs per thread and 128-, 256- or
// 512-thread workgroups perform alike (profiles/r01_fir_variants_ab.txt).
// EXACT mode issues v_mul_f64 + v_add_f64 per tap (bit-identical to the
// reference's unfused x86 loop); FMA mode issues one v_fma_f64 per tap.
// This translation unit is compiled with -ffp-contract=off so that only the
// explicit __builtin_fma is ever fused.

#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include "gdsp_common.h"

#define FIR_THREADS 256
#define FIR_R       9            // outputs per thread (odd, see above)
#define FIR_KC_MAX  1026         // taps per LDS stage in the generic kernel (multiple of FIR_R)

template <int W> struct FirTaps { double w[W]; };

// ------------------------------------------------------------------ store ----
// Tile results: registers -> LDS (lane-strided, conflict-free for odd R) ->
// coalesced 16-byte global stores.  Caller has already synchronised after the
// last read of the staged inputs.
template <int R>
__device__ __forceinline__ void fir_store_tile (double* lds, const double (&acc)[R],
                                                double* __restrict__ out, int64_t tileStart, uint32_t n)
	{
	constexpr int T = FIR_THREADS * R;
	double* mine = lds + threadIdx.x * R;
#pragma unroll
	for (int r=0 ; r<R ; r++) mine[r] = acc[r];
	__syncthreads ();

	if (tileStart + T <= (int64_t) n)
		{
		double2*       dst = reinterpret_cast<double2*> (out + tileStart);
		const double2* src = reinterpret_cast<const double2*> (lds);
#pragma unroll
		for (int i=0 ; i<(T/2 + FIR_THREADS - 1)/FIR_THREADS ; i++)
			{
			int p = threadIdx.x + i*FIR_THREADS;
			if (p < T/2) gdsp_st2 (&dst[p], src[p]);
			}
		}
	else
		{
		for (int p=threadIdx.x ; p<T ; p+=FIR_THREADS)
			{ if (tileStart + p < (int64_t) n) out[tileStart+p] = lds[p]; }
		}
	}

// ---------------------------------------------------- compile-time W kernel ----
template <int W, int R, bool FMA>
__device__ __forceinline__ void fir_fixed_tile (const double* __restrict__ in, double* __restrict__ out,
                                                uint32_t n, uint32_t tile, const FirTaps<W>& taps)
	{
	constexpr int H  = (W - 1) / 2;
	constexpr int T  = FIR_THREADS * R;
	constexpr int SH = H & 1;                 // keeps the first staged index even
	constexpr int L  = T + W - 1 + SH;
	constexpr int LP = (L + 1) & ~1;
	__shared__ __attribute__((aligned(16))) double lds[LP];

	const int64_t  tileStart = (int64_t) tile * T;
	const int64_t  g0        = tileStart - H - SH;

	// stage tile + halo
	if ((g0 >= 0) && (g0 + LP <= (int64_t) n))
		{
		const double2* src = reinterpret_cast<const double2*> (in + g0);
		double2*       dst = reinterpret_cast<double2*> (lds);
#pragma unroll
		for (int i=0 ; i<(LP/2 + FIR_THREADS - 1)/FIR_THREADS ; i++)
			{
			int p = threadIdx.x + i*FIR_THREADS;
			if (p < LP/2) dst[p] = src[p];
			}
		}
	else
		{
		for (int p=threadIdx.x ; p<LP ; p+=FIR_THREADS)
			{
			int64_t g = g0 + p;
			lds[p] = ((g >= 0) && (g < (int64_t) n))? in[g] : 0.0;
			}
		}
	__syncthreads ();

	// R outputs per thread; input j of this thread meets tap k=j-r of output r
	const double* x = lds + SH + threadIdx.x * R;
	double acc[R];
#pragma unroll
	for (int r=0 ; r<R ; r++) acc[r] = 0.0;

#pragma unroll
	for (int j=0 ; j<R+W-1 ; j++)
		{
		const double xv = x[j];
#pragma unroll
		for (int r=0 ; r<R ; r++)
			{
			const int k = j - r;
			if ((k >= 0) && (k < W))
				{
				if (FMA) acc[r] = __builtin_fma (taps.w[k], xv, acc[r]);
				else     acc[r] = acc[r] + taps.w[k] * xv;
				}
			}
		}

	__syncthreads ();
	fir_store_tile<R> (lds, acc, out, tileStart, n);
	}

template <int W, int R, bool FMA>
__global__ __launch_bounds__(FIR_THREADS)
void fir_fixed_kernel (const double* __restrict__ in, double* __restrict__ out,
                       uint32_t n, uint32_t ntiles, FirTaps<W> taps)
	{ fir_fixed_tile<W, R, FMA> (in, out, n, gdsp_xcd_tile (blockIdx.x, ntiles), taps); }

template <int W, int R, bool FMA>                        // one grid over every vector of the table (gdsp_common.h)
__global__ __launch_bounds__(FIR_THREADS)
void fir_fixed_batch_kernel (GdspBatch B, FirTaps<W> taps)
	{
	const double* in;  double* out;  uint32_t n;
	const uint32_t tile = gdsp_batch_tile (B, in, out, n);
	fir_fixed_tile<W, R, FMA> (in, out, n, tile, taps);
	}

// ------------------------------------------- fixed W fused with localmin/localmax ----
// `= smooth W=101 = localmax N=11` (BASELINE configs[2]) in one pass: the smoothed tile is
// already in LDS on its way out, so the neighbourhood test runs there and only the peaks
// track is written -- 16 B/base for the pair instead of 32.  A workgroup still computes
// 256*R smoothed values but keeps the inner 256*R - 2h (h = neighbourhood half width); the
// h values on either side are recomputed by the neighbouring tiles (0.4 % extra FP64 work
// at N=11).  Smoothed values and comparisons are exactly those of the two separate
// kernels, so the result is bit-identical to running them one after the other.
#define FIR_FUSE_MAX_HALF 64
template <int W, int R, bool FMA, bool MAX>
__device__ __forceinline__ void fir_fixed_extrema_tile (const double* __restrict__ in, double* __restrict__ out,
                                                        uint32_t n, uint32_t tile, const FirTaps<W>& taps, int h, double fill)
	{
	constexpr int H  = (W - 1) / 2;
	constexpr int T  = FIR_THREADS * R;
	constexpr int LP = (T + W - 1 + 1 + 1) & ~1;           // room for the alignment shift
	__shared__ __attribute__((aligned(16))) double lds[LP];

	const int      sh        = (h + H) & 1;                // keeps the first staged index even
	const int      stride    = T - 2*h;                    // outputs kept per tile (even)
	const int64_t  keepStart = (int64_t) tile * stride;    // first output this tile stores
	const int64_t  compStart = keepStart - h;              // first smoothed value it computes
	const int64_t  g0        = compStart - H - sh;

	gdsp_stage_f64<FIR_THREADS> (lds, in, n, g0, LP, 0.0);
	__syncthreads ();

	const double* x = lds + sh + threadIdx.x * R;
	double acc[R];
#pragma unroll
	for (int r=0 ; r<R ; r++) acc[r] = 0.0;
#pragma unroll
	for (int j=0 ; j<R+W-1 ; j++)
		{
		const double xv = x[j];
#pragma unroll
		for (int r=0 ; r<R ; r++)
			{
			const int k = j - r;
			if ((k >= 0) && (k < W))
				{
				if (FMA) acc[r] = __builtin_fma (taps.w[k], xv, acc[r]);
				else     acc[r] = acc[r] + taps.w[k] * xv;
				}
			}
		}
	__syncthreads ();

	// smoothed values into LDS; positions outside the vector can never beat anything
	const double never = MAX? -INFINITY : INFINITY;
	double* s = lds;
#pragma unroll
	for (int r=0 ; r<R ; r++)
		{
		const int64_t g = compStart + (int64_t) threadIdx.x * R + r;
		s[threadIdx.x * R + r] = ((g >= 0) && (g < (int64_t) n))? acc[r] : never;
		}
	__syncthreads ();

	for (int o = 2*threadIdx.x ; o < stride ; o += 2*FIR_THREADS)
		{
		const int64_t g = keepStart + o;
		if (g >= (int64_t) n) break;
		const double* w = s + o;                           // w[0..2h] = neighbourhood of output o
		double e0 = w[0], e1 = w[2*h + 1], mid = w[1];
		for (int k=2 ; k<=2*h ; k++) mid = MAX? fmax (mid, w[k]) : fmin (mid, w[k]);
		if (h > 0) { e0 = MAX? fmax (e0, mid) : fmin (e0, mid);  e1 = MAX? fmax (mid, e1) : fmin (mid, e1); }
		else       { e1 = w[1]; }
		const double c0 = w[h], c1 = w[h+1];
		const double r0 = (MAX? (e0 > c0) : (e0 < c0))? fill : c0;
		const double r1 = (MAX? (e1 > c1) : (e1 < c1))? fill : c1;
		if (g + 1 < (int64_t) n) gdsp_st2 (reinterpret_cast<double2*> (out + g), make_double2 (r0, r1));
		else                     out[g] = r0;
		}
	}

template <int W, int R, bool FMA, bool MAX>
__global__ __launch_bounds__(FIR_THREADS)
void fir_fixed_extrema_kernel (const double* __restrict__ in, double* __restrict__ out,
                               uint32_t n, uint32_t ntiles, FirTaps<W> taps, int h, double fill)
	{ fir_fixed_extrema_tile<W, R, FMA, MAX> (in, out, n, gdsp_xcd_tile (blockIdx.x, ntiles), taps, h, fill); }

template <int W, int R, bool FMA, bool MAX>
__global__ __launch_bounds__(FIR_THREADS)
void fir_fixed_extrema_batch_kernel (GdspBatch B, FirTaps<W> taps, int h, double fill)
	{
	const double* in;  double* out;  uint32_t n;
	const uint32_t tile = gdsp_batch_tile (B, in, out, n);
	fir_fixed_extrema_tile<W, R, FMA, MAX> (in, out, n, tile, taps, h, fill);
	}

// the same over a table of vectors, gated (gdsp_peaks.hip): a block works only when its vector takes the direct route --
// the probe of the filtered route chose it, or the filter's queue overflowed; otherwise it leaves at once
// A block answers for FIR_GATED_TILES consecutive tiles of its vector (the table B counts such groups): on the filtered
// route every block of this launch leaves at once, and leaving costs what dispatching it costs -- 1.35 M single-tile
// blocks per genome were 0.66 ms of nothing; groups of sixteen are 84 k blocks.
#define FIR_GATED_TILES 16
template <int W, int R, bool FMA, bool MAX>
__global__ __launch_bounds__(FIR_THREADS)
void fir_fixed_extrema_gated_kernel (GdspBatch B, const GdspPeaksCtl* __restrict__ ctl, FirTaps<W> taps, int h, double fill)
	{
	const double* in;  double* out;  uint32_t n, v;
	const uint32_t group = gdsp_batch_tile (B, in, out, n, &v);
	const GdspPeaksCtl c = ctl[v];
	if (!gdsp_peaks_takes_direct (c) && (c.overflow == 0)) return;
	const uint32_t stride = FIR_THREADS * R - 2 * (uint32_t) h;
	for (uint32_t k=0 ; k<FIR_GATED_TILES ; k++)
		{
		const uint32_t tile = group * FIR_GATED_TILES + k;
		if ((uint64_t) tile * stride >= n) break;
		if (k != 0) __syncthreads ();                              // (the previous tile's last reads of the LDS image)
		fir_fixed_extrema_tile<W, R, FMA, MAX> (in, out, n, tile, taps, h, fill);
		}
	}

// --------------------------------------------------------- run-time W kernel ----
// Any odd W (up to the reference's 50001, sum.c:478).  Taps are walked in stages
// of at most KC so the LDS image stays small; the accumulators live in registers
// across stages, so every output still sums its taps in ascending order.
template <int R, bool FMA>
__device__ __forceinline__ void fir_generic_tile (const double* __restrict__ in, double* __restrict__ out,
                                                  uint32_t n, uint32_t tile,
                                                  const double* __restrict__ taps, uint32_t W, uint32_t KC)
	{
	extern __shared__ __attribute__((aligned(16))) double ldsDyn[];
	constexpr int T = FIR_THREADS * R;

	const int64_t  tileStart = (int64_t) tile * T;
	const int64_t  H         = (W - 1) / 2;

	double acc[R];
#pragma unroll
	for (int r=0 ; r<R ; r++) acc[r] = 0.0;

	for (uint32_t k0=0 ; k0<W ; k0+=KC)
		{
		const uint32_t kc  = (W - k0 < KC)? (W - k0) : KC;
		const uint32_t kcR = ((kc + R - 1) / R) * R;
		const int64_t  gA  = tileStart - H + k0;          // global index of this stage's first input
		const int      sh  = (int) (gA & 1);               // (gA may be negative: & 1 is still its parity)
		const int64_t  g0  = gA - sh;
		const int      L   = (T + (int) kcR + R + sh + 1) & ~1;

		if (k0 != 0) __syncthreads ();
		gdsp_stage_f64<FIR_THREADS> (ldsDyn, in, n, g0, L, 0.0);
		__syncthreads ();

		const double* x = ldsDyn + sh + threadIdx.x * R;
		double xw[2*R];
#pragma unroll
		for (int r=0 ; r<R ; r++) xw[r] = x[r];

		for (uint32_t kb=0 ; kb<kc ; kb+=R)
			{
#pragma unroll
			for (int r=0 ; r<R ; r++) xw[R+r] = x[kb+R+r];
#pragma unroll
			for (int kk=0 ; kk<R ; kk++)
				{
				if (kb + kk < kc)                          // wave-uniform
					{
					const double w = taps[k0+kb+kk];
#pragma unroll
					for (int r=0 ; r<R ; r++)
						{
						if (FMA) acc[r] = __builtin_fma (w, xw[r+kk], acc[r]);
						else     acc[r] = acc[r] + w * xw[r+kk];
						}
					}
				}
#pragma unroll
			for (int r=0 ; r<R ; r++) xw[r] = xw[R+r];
			}
		}

	__syncthreads ();
	fir_store_tile<R> (ldsDyn, acc, out, tileStart, n);
	}

template <int R, bool FMA>
__global__ __launch_bounds__(FIR_THREADS)
void fir_generic_kernel (const double* __restrict__ in, double* __restrict__ out,
                         uint32_t n, uint32_t ntiles,
                         const double* __restrict__ taps, uint32_t W, uint32_t KC)
	{ fir_generic_tile<R, FMA> (in, out, n, gdsp_xcd_tile (blockIdx.x, ntiles), taps, W, KC); }

template <int R, bool FMA>
__global__ __launch_bounds__(FIR_THREADS)
void fir_generic_batch_kernel (GdspBatch B, const double* __restrict__ taps, uint32_t W, uint32_t KC)
	{
	const double* in;  double* out;  uint32_t n;
	const uint32_t tile = gdsp_batch_tile (B, in, out, n);
	fir_generic_tile<R, FMA> (in, out, n, tile, taps, W, KC);
	}

// ------------------------------------------------------------------- host ----

struct gdsp_fir_plan
	{
	uint32_t W;
	double*  h_taps;     // W values
	double*  d_taps;     // W values + padding, device
	};

extern "C" {

// Hann taps exactly as the reference builds them (sum.c:632-645): mirrored
// store of (1-cos(2*pi*x))/2 for x=(k+1)/(W+1), ascending sum, divide.
int gdsp_hann_taps (uint32_t W, double* h_taps)
	{
	GDSP_REQUIRE (h_taps != NULL, "h_taps is NULL");
	GDSP_REQUIRE ((W >= 3) && (W & 1), "W must be odd and >= 3");
	const uint32_t hOff = (W - 1) / 2;
	const double   pi   = 3.14159265358979323846264;
	for (uint32_t k=0 ; k<=hOff ; k++)
		{
		double x = (k+1) / (double) (W+1);
		h_taps[k] = h_taps[W-1-k] = (1 - cos (2*pi*x)) / 2;
		}
	double total = 0.0;
	for (uint32_t k=0 ; k<W ; k++) total += h_taps[k];
	for (uint32_t k=0 ; k<W ; k++) h_taps[k] /= total;
	return GDSP_OK;
	}

int gdsp_fir_plan_create (gdsp_fir_plan** plan, const double* h_taps, uint32_t W)
	{
	GDSP_REQUIRE (plan != NULL, "plan is NULL");
	GDSP_REQUIRE (h_taps != NULL, "h_taps is NULL");
	GDSP_REQUIRE ((W >= 1) && (W & 1), "W must be odd");
	gdsp_fir_plan* p = (gdsp_fir_plan*) calloc (1, sizeof(gdsp_fir_plan));
	if (p == NULL) { gdsp_set_error ("out of host memory");  return GDSP_ENOMEM; }
	p->W = W;
	p->h_taps = (double*) malloc ((size_t) W * sizeof(double));
	if (p->h_taps == NULL) { free (p);  gdsp_set_error ("out of host memory");  return GDSP_ENOMEM; }
	memcpy (p->h_taps, h_taps, (size_t) W * sizeof(double));
	size_t padded = (size_t) W + 2*FIR_R;
	hipError_t e = hipMalloc ((void**) &p->d_taps, padded * sizeof(double));
	if (e != hipSuccess) { free (p->h_taps);  free (p);  GDSP_HIP_TRY (e); }
	e = hipMemset (p->d_taps, 0, padded * sizeof(double));
	if (e == hipSuccess) e = hipMemcpy (p->d_taps, h_taps, (size_t) W * sizeof(double), hipMemcpyHostToDevice);
	if (e != hipSuccess) { (void) hipFree (p->d_taps);  free (p->h_taps);  free (p);  GDSP_HIP_TRY (e); }
	*plan = p;
	return GDSP_OK;
	}

int gdsp_fir_plan_destroy (gdsp_fir_plan* plan)
	{
	if (plan == NULL) return GDSP_OK;
	hipError_t e = hipFree (plan->d_taps);
	free (plan->h_taps);
	free (plan);
	GDSP_HIP_TRY (e);
	return GDSP_OK;
	}

int gdsp_fir_apply (const gdsp_fir_plan* plan, const double* d_in, double* d_out,
                    uint32_t n, int mode, void* stream)
	{
	GDSP_REQUIRE (plan != NULL, "plan is NULL");
	GDSP_REQUIRE ((mode == GDSP_FIR_EXACT) || (mode == GDSP_FIR_FMA), "unknown mode");
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_in != NULL) && (d_out != NULL), "NULL vector");
	GDSP_REQUIRE (d_in != d_out, "FIR is out-of-place: d_out must not alias d_in");
	GDSP_REQUIRE (gdsp_aligned16 (d_in) && gdsp_aligned16 (d_out), "vectors must be 16-byte aligned");

	constexpr int  T = FIR_THREADS * FIR_R;
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + T - 1) / T);
	hipStream_t    s = gdsp_stream (stream);

	if ((plan->W == 101) && (mode == GDSP_FIR_EXACT) && gdsp_fir_slide_wanted (n))
		{
		const gdsp_batch_item one = { d_in, d_out, n };
		return gdsp_fir_slide_batch (&one, 1, plan->h_taps, stream);
		}
	if (plan->W == 101)
		{
		FirTaps<101> taps;
		memcpy (taps.w, plan->h_taps, sizeof(taps.w));
		if (mode == GDSP_FIR_FMA)
			hipLaunchKernelGGL ((fir_fixed_kernel<101, FIR_R, true>),  dim3(ntiles), dim3(FIR_THREADS), 0, s,
			                    d_in, d_out, n, ntiles, taps);
		else
			hipLaunchKernelGGL ((fir_fixed_kernel<101, FIR_R, false>), dim3(ntiles), dim3(FIR_THREADS), 0, s,
			                    d_in, d_out, n, ntiles, taps);
		}
	else
		{
		uint32_t KC = ((plan->W + FIR_R - 1) / FIR_R) * FIR_R;
		if (KC > FIR_KC_MAX) KC = FIR_KC_MAX;
		size_t ldsBytes = ((size_t) T + KC + 2*FIR_R + 4) * sizeof(double);
		if (mode == GDSP_FIR_FMA)
			hipLaunchKernelGGL ((fir_generic_kernel<FIR_R, true>),  dim3(ntiles), dim3(FIR_THREADS), ldsBytes, s,
			                    d_in, d_out, n, ntiles, plan->d_taps, plan->W, KC);
		else
			hipLaunchKernelGGL ((fir_generic_kernel<FIR_R, false>), dim3(ntiles), dim3(FIR_THREADS), ldsBytes, s,
			                    d_in, d_out, n, ntiles, plan->d_taps, plan->W, KC);
		}
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

} // extern "C"

static int smooth_plan (uint32_t W, gdsp_fir_plan** out);
int gdsp_smooth_taps_device (uint32_t W, const double** d_taps)
	{
	gdsp_fir_plan* plan = NULL;
	int rc = smooth_plan (W, &plan);
	if (rc == GDSP_OK) *d_taps = plan->d_taps;
	return rc;
	}

template <bool FMA, bool MAX>
static void fir_extrema_launch (const double* d_in, double* d_out, uint32_t n, const double* h_taps,
                                int h, double fill, hipStream_t s)
	{
	FirTaps<101> taps;
	memcpy (taps.w, h_taps, sizeof(taps.w));
	const int      stride = FIR_THREADS*FIR_R - 2*h;
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + stride - 1) / stride);
	hipLaunchKernelGGL ((fir_fixed_extrema_kernel<101, FIR_R, FMA, MAX>), dim3(ntiles), dim3(FIR_THREADS), 0, s,
	                    d_in, d_out, n, ntiles, taps, h, fill);
	}

extern "C" {

// Hann plans cached per (device, W): `smooth` rebuilds its window for every
// chromosome in the reference (sum.c:632-645); here it is built once.
// Plans are never evicted (an entry is W taps; a caller or a kernel in flight on another
// device may still hold a plan's taps), so a pointer handed out stays valid for the process.
struct smooth_cache_entry { int device; uint32_t W; gdsp_fir_plan* plan; };
static std::vector<smooth_cache_entry> smoothCache;
static std::mutex smoothCacheLock;

// 1 when gdsp_smooth_local_extrema has a fused kernel for this pair of parameters
int gdsp_smooth_local_extrema_fusable (uint32_t W, uint32_t N)
	{ return (W == 101) && (N >= 1) && ((N - 1) / 2 <= FIR_FUSE_MAX_HALF); }

// `= smooth W = localmax|localmin N` in one pass (sum.c:616-676 then minmax.c:981-1022 /
// :1183-1227); bit-identical to gdsp_smooth followed by gdsp_local_extrema.
int gdsp_smooth_local_extrema (const double* d_in, double* d_out, uint32_t n, uint32_t W, int mode,
                               uint32_t N, int wantMax, double fill, void* stream)
	{
	GDSP_REQUIRE (gdsp_smooth_local_extrema_fusable (W, N), "no fused kernel for this window/neighborhood");
	if (mode == GDSP_FIR_HANN) mode = GDSP_FIR_FMA;            // ties must stay ties under the strict comparisons: direct taps
	GDSP_REQUIRE ((mode == GDSP_FIR_EXACT) || (mode == GDSP_FIR_FMA), "unknown mode");
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_in != NULL) && (d_out != NULL) && (d_in != d_out), "vectors must be distinct and non-NULL");
	GDSP_REQUIRE (gdsp_aligned16 (d_in) && gdsp_aligned16 (d_out), "vectors must be 16-byte aligned");
	gdsp_fir_plan* plan = NULL;
	int rc = smooth_plan (W, &plan);
	if (rc != GDSP_OK) return rc;
	const int   h = (int) ((N - 1) / 2);
	hipStream_t s = gdsp_stream (stream);
	if (gdsp_peaks_filter_available (W, N) && ((mode == GDSP_FIR_EXACT) || gdsp_peaks_filter_wanted_for_fma ()))   // the filtered route, see gdsp_smooth_local_extrema_batch
		{
		const gdsp_batch_item one = { d_in, d_out, n };
		return gdsp_peaks_filter_batch (&one, 1, plan->h_taps, mode == GDSP_FIR_FMA, N, wantMax, fill, stream);
		}
	if (mode == GDSP_FIR_FMA)
		{ if (wantMax) fir_extrema_launch<true, true>  (d_in, d_out, n, plan->h_taps, h, fill, s);
		  else         fir_extrema_launch<true, false> (d_in, d_out, n, plan->h_taps, h, fill, s); }
	else
		{ if (wantMax) fir_extrema_launch<false, true>  (d_in, d_out, n, plan->h_taps, h, fill, s);
		  else         fir_extrema_launch<false, false> (d_in, d_out, n, plan->h_taps, h, fill, s); }
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

int gdsp_smooth (const double* d_in, double* d_out, uint32_t n, uint32_t W, int mode, void* stream)
	{
	if (mode == GDSP_FIR_HANN)
		{
		GDSP_REQUIRE ((W >= 3) && (W & 1), "W must be odd and >= 3");
		GDSP_REQUIRE (W <= 50001, "W exceeds 50001");
		if (gdsp_hann_blocks_available (W)) return gdsp_hann_blocks_apply (d_in, d_out, n, W, stream);
		if (gdsp_hann_far_available (W))    return gdsp_hann_far_apply (d_in, d_out, n, W, stream);
		mode = GDSP_FIR_FMA;                                   // same tolerance class, direct evaluation
		}
	gdsp_fir_plan* plan = NULL;
	int rc = smooth_plan (W, &plan);
	if (rc != GDSP_OK) return rc;
	return gdsp_fir_apply (plan, d_in, d_out, n, mode, stream);
	}

} // extern "C"

// every vector of a batch: non-NULL, distinct input and output, 16-byte aligned
static int batch_check (const gdsp_batch_item* items, int nitems, bool inPlace)
	{
	GDSP_REQUIRE ((nitems == 0) || (items != NULL), "items is NULL");
	GDSP_REQUIRE (nitems >= 0, "negative item count");
	for (int i=0 ; i<nitems ; i++)
		{
		if (items[i].n == 0) continue;
		GDSP_REQUIRE (items[i].d_out != NULL, "NULL vector");
		GDSP_REQUIRE (gdsp_aligned16 (items[i].d_out), "vectors must be 16-byte aligned");
		if (inPlace) continue;
		GDSP_REQUIRE ((items[i].d_in != NULL) && (items[i].d_in != items[i].d_out), "vectors must be distinct and non-NULL");
		GDSP_REQUIRE (gdsp_aligned16 (items[i].d_in), "vectors must be 16-byte aligned");
		}
	return GDSP_OK;
	}
int gdsp_batch_check (const gdsp_batch_item* items, int nitems, bool inPlace) { return batch_check (items, nitems, inPlace); }

// gdsp_smooth for every vector of a device in one launch (sum.c:616-676 applied per chromosome by genodsp.c:909-921)
extern "C" int gdsp_smooth_batch (const gdsp_batch_item* items, int nitems, uint32_t W, int mode, void* stream)
	{
	GDSP_REQUIRE ((W >= 3) && (W & 1), "W must be odd and >= 3");
	GDSP_REQUIRE (W <= 50001, "W exceeds 50001");
	int rc = batch_check (items, nitems, false);
	if (rc != GDSP_OK) return rc;
	if (mode == GDSP_FIR_HANN)
		{
		if (gdsp_hann_blocks_batch_available (W)) return gdsp_hann_blocks_apply_batch (items, nitems, W, stream);
		if (gdsp_hann_blocks_available (W) || gdsp_hann_far_available (W))
			{
			for (int i=0 ; i<nitems ; i++)
				{ rc = gdsp_smooth (items[i].d_in, items[i].d_out, items[i].n, W, mode, stream);  if (rc != GDSP_OK) return rc; }
			return GDSP_OK;
			}
		mode = GDSP_FIR_FMA;
		}
	GDSP_REQUIRE ((mode == GDSP_FIR_EXACT) || (mode == GDSP_FIR_FMA), "unknown mode");
	gdsp_fir_plan* plan = NULL;
	rc = smooth_plan (W, &plan);
	if (rc != GDSP_OK) return rc;
	constexpr int T = FIR_THREADS * FIR_R;
	hipStream_t   s = gdsp_stream (stream);
	auto tilesOf = [] (uint32_t n) { return ((uint64_t) n + T - 1) / T; };
	uint64_t total = 0;
	for (int i=0 ; i<nitems ; i++) total += items[i].n;
	bool aligned = true;                                           // (the slide forms load and store 16 bytes a lane; stretches of --sharding=bases may start on 8)
	for (int i=0 ; i<nitems ; i++) aligned = aligned && gdsp_aligned16 (items[i].d_in) && gdsp_aligned16 (items[i].d_out);
	if ((W == 101) && (mode == GDSP_FIR_EXACT) && aligned && gdsp_fir_slide_wanted (total))
		return gdsp_fir_slide_batch (items, nitems, plan->h_taps, stream);
	if (W == 101)
		{
		FirTaps<101> taps;
		memcpy (taps.w, plan->h_taps, sizeof(taps.w));
		gdsp_batch_run (items, nitems, tilesOf, [&] (const GdspBatch& B, uint32_t tiles)
			{
			if (mode == GDSP_FIR_FMA) hipLaunchKernelGGL ((fir_fixed_batch_kernel<101, FIR_R, true>),  dim3(tiles), dim3(FIR_THREADS), 0, s, B, taps);
			else                      hipLaunchKernelGGL ((fir_fixed_batch_kernel<101, FIR_R, false>), dim3(tiles), dim3(FIR_THREADS), 0, s, B, taps);
			});
		}
	else
		{
		uint32_t KC = ((W + FIR_R - 1) / FIR_R) * FIR_R;
		if (KC > FIR_KC_MAX) KC = FIR_KC_MAX;
		const size_t ldsBytes = ((size_t) T + KC + 2*FIR_R + 4) * sizeof(double);
		gdsp_batch_run (items, nitems, tilesOf, [&] (const GdspBatch& B, uint32_t tiles)
			{
			if (mode == GDSP_FIR_FMA) hipLaunchKernelGGL ((fir_generic_batch_kernel<FIR_R, true>),  dim3(tiles), dim3(FIR_THREADS), ldsBytes, s, B, plan->d_taps, W, KC);
			else                      hipLaunchKernelGGL ((fir_generic_batch_kernel<FIR_R, false>), dim3(tiles), dim3(FIR_THREADS), ldsBytes, s, B, plan->d_taps, W, KC);
			});
		}
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

template <bool FMA, bool MAX>
static void fir_extrema_batch_launch (const gdsp_batch_item* items, int nitems, const double* h_taps, int h, double fill, hipStream_t s)
	{
	FirTaps<101> taps;
	memcpy (taps.w, h_taps, sizeof(taps.w));
	const int stride = FIR_THREADS*FIR_R - 2*h;
	gdsp_batch_run (items, nitems, [=] (uint32_t n) { return ((uint64_t) n + stride - 1) / stride; },
		[&] (const GdspBatch& B, uint32_t tiles)
			{ hipLaunchKernelGGL ((fir_fixed_extrema_batch_kernel<101, FIR_R, FMA, MAX>), dim3(tiles), dim3(FIR_THREADS), 0, s, B, taps, h, fill); });
	}

int gdsp_fir_extrema_gated_launch (const gdsp_batch_item* items, int count, const GdspPeaksCtl* d_ctl, const double* h_taps,
                                   int fma, int h, int wantMax, double fill, void* stream)
	{
	FirTaps<101> taps;
	memcpy (taps.w, h_taps, sizeof(taps.w));
	const int stride = FIR_THREADS*FIR_R - 2*h;
	GdspBatch B;
	gdsp_batch_make (B, items, count, [=] (uint32_t n) { return (((uint64_t) n + stride - 1) / stride + FIR_GATED_TILES - 1) / FIR_GATED_TILES; });
	hipStream_t s = gdsp_stream (stream);
	const dim3 grid (B.tile0[GDSP_BATCH_MAX]), block (FIR_THREADS);
	if (fma) { if (wantMax) hipLaunchKernelGGL ((fir_fixed_extrema_gated_kernel<101, FIR_R, true,  true>),  grid, block, 0, s, B, d_ctl, taps, h, fill);
	           else         hipLaunchKernelGGL ((fir_fixed_extrema_gated_kernel<101, FIR_R, true,  false>), grid, block, 0, s, B, d_ctl, taps, h, fill); }
	else     { if (wantMax) hipLaunchKernelGGL ((fir_fixed_extrema_gated_kernel<101, FIR_R, false, true>),  grid, block, 0, s, B, d_ctl, taps, h, fill);
	           else         hipLaunchKernelGGL ((fir_fixed_extrema_gated_kernel<101, FIR_R, false, false>), grid, block, 0, s, B, d_ctl, taps, h, fill); }
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

extern "C" int gdsp_smooth_local_extrema_batch (const gdsp_batch_item* items, int nitems, uint32_t W, int mode,
                                                uint32_t N, int wantMax, double fill, void* stream)
	{
	GDSP_REQUIRE (gdsp_smooth_local_extrema_fusable (W, N), "no fused kernel for this window/neighborhood");
	if (mode == GDSP_FIR_HANN) mode = GDSP_FIR_FMA;
	GDSP_REQUIRE ((mode == GDSP_FIR_EXACT) || (mode == GDSP_FIR_FMA), "unknown mode");
	int rc = batch_check (items, nitems, false);
	if (rc != GDSP_OK) return rc;
	gdsp_fir_plan* plan = NULL;
	rc = smooth_plan (W, &plan);
	if (rc != GDSP_OK) return rc;
	const int   h = (int) ((N - 1) / 2);
	hipStream_t s = gdsp_stream (stream);
	// the filtered route (gdsp_peaks.hip): block sums rule out ~99.5 % of the bases, the rest are evaluated tap by tap in
	// the reference's arithmetic -- the same bits, off the FP64 pipe; tie-heavy vectors fall to the direct kernel on the
	// device.  EXACT: 278 against 133 Gbases/s over the genome; FMA (one fused multiply-add per tap in the exact chains and
	// in the direct kernel) takes it as well since round 4 (gdsp_peaks_filter_wanted_for_fma)
	if (gdsp_peaks_filter_available (W, N) && ((mode == GDSP_FIR_EXACT) || gdsp_peaks_filter_wanted_for_fma ()))
		return gdsp_peaks_filter_batch (items, nitems, plan->h_taps, mode == GDSP_FIR_FMA, N, wantMax, fill, stream);
	if (mode == GDSP_FIR_FMA)
		{ if (wantMax) fir_extrema_batch_launch<true, true>  (items, nitems, plan->h_taps, h, fill, s);
		  else         fir_extrema_batch_launch<true, false> (items, nitems, plan->h_taps, h, fill, s); }
	else
		{ if (wantMax) fir_extrema_batch_launch<false, true>  (items, nitems, plan->h_taps, h, fill, s);
		  else         fir_extrema_batch_launch<false, false> (items, nitems, plan->h_taps, h, fill, s); }
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

extern "C" {

static int smooth_plan (uint32_t W, gdsp_fir_plan** out)
	{
	GDSP_REQUIRE ((W >= 3) && (W & 1), "W must be odd and >= 3");
	GDSP_REQUIRE (W <= 50001, "W exceeds 50001");          // sum.c:478, :557-558
	int device = 0;
	GDSP_HIP_TRY (hipGetDevice (&device));

	gdsp_fir_plan* plan = NULL;
	std::lock_guard<std::mutex> hold (smoothCacheLock);
	for (const smooth_cache_entry& e : smoothCache)
		{ if ((e.device == device) && (e.W == W)) { plan = e.plan;  break; } }
	if (plan == NULL)
		{
		double* taps = (double*) malloc ((size_t) W * sizeof(double));
		if (taps == NULL) { gdsp_set_error ("out of host memory");  return GDSP_ENOMEM; }
		gdsp_hann_taps (W, taps);
		int rc = gdsp_fir_plan_create (&plan, taps, W);
		free (taps);
		if (rc != GDSP_OK) return rc;
		smoothCache.push_back (smooth_cache_entry { device, W, plan });
		}
	*out = plan;
	return GDSP_OK;
	}

} // extern "C"
