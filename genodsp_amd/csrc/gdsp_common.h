// gdsp_common.h -- shared by the HIP translation units of libgenodsp_hip.so
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include "genodsp_hip.h"

// error plumbing: the C ABI returns codes; the message is kept per thread
void gdsp_set_error (const char* fmt, ...);

#define GDSP_HIP_TRY(call)                                                        \
	do {                                                                          \
		hipError_t e_ = (call);                                                   \
		if (e_ != hipSuccess) {                                                   \
			gdsp_set_error ("%s:%d: %s -> %s", __FILE__, __LINE__, #call,         \
			                hipGetErrorString (e_));                              \
			return GDSP_EHIP;                                                     \
		}                                                                         \
	} while (0)

#define GDSP_LAUNCH_CHECK()  GDSP_HIP_TRY (hipGetLastError ())

#define GDSP_REQUIRE(cond, msg)                                                   \
	do {                                                                          \
		if (!(cond)) {                                                            \
			gdsp_set_error ("%s: %s", __func__, msg);                             \
			return GDSP_EINVAL;                                                   \
		}                                                                         \
	} while (0)

// gdsp_hann.hip: `smooth` through block sums of the Hann window's constant and cosine parts
bool gdsp_hann_blocks_available (uint32_t W);
int  gdsp_hann_blocks_apply (const double* d_in, double* d_out, uint32_t n, uint32_t W, void* stream);
bool gdsp_hann_blocks_batch_available (uint32_t W);                   // (the W=101 kernel has a one-launch-per-device form)
int  gdsp_hann_blocks_apply_batch (const gdsp_batch_item* items, int nitems, uint32_t W, void* stream);
int  gdsp_batch_check (const gdsp_batch_item* items, int nitems, bool inPlace);   // gdsp_fir.hip: argument checks shared by the *_batch calls

// gdsp_hann_far.hip: the same for windows longer than one LDS tile can hold (3201 .. 50001 taps), block totals in HBM
bool gdsp_hann_far_available (uint32_t W);
int  gdsp_hann_far_apply (const double* d_in, double* d_out, uint32_t n, uint32_t W, void* stream);

// gdsp_fir.hip: the device copy of the reference's Hann taps for W (cached per device, never freed)
int  gdsp_smooth_taps_device (uint32_t W, const double** d_taps);

// gdsp_extrema.hip: dilate / erode as window-any / window-all in the block form
bool gdsp_morph_blocks_available (uint32_t left, uint32_t right);
void gdsp_morph_blocks (const double* d_in, double* d_out, uint32_t n, uint32_t left, uint32_t right, int erode,
                        double T, double one, double zero, void* stream);

bool gdsp_morph_blocks_batch (const gdsp_batch_item* items, int nitems, uint32_t left, uint32_t right, int erode,
                              double T, double one, double zero, void* stream);      // false: some vector needs another kernel

// gdsp_fir_slide.hip: `smooth W=101` in the reference's arithmetic with every product computed once (sliding accumulators)
bool gdsp_fir_slide_wanted (uint64_t bases);
int  gdsp_fir_slide_batch (const gdsp_batch_item* items, int nitems, const double* h_taps, void* stream);

static inline hipStream_t gdsp_stream (void* s) { return (hipStream_t) s; }

__host__ __device__ static inline bool gdsp_aligned16 (const void* p) { return (((uintptr_t) p) & 15) == 0; }

// 16-byte accesses of data this kernel touches once: non-temporal, so the streams do not sweep the L2 for each other
// (measured on hann_blocks_kernel<101>: 383 -> 400 Gbases/s with both; loads alone +2.5 %, stores alone +0 %).  Only for
// accesses where the lanes of a wave cover whole lines between them: a lane that walks a strip of its own (the select
// histogram, the report's count pass) needs the rest of each line to wait in the cache -- non-temporal loads there cost 1.7-3x.
// GDSP_STREAMING=0 builds the plain accesses (A/B).
#ifndef GDSP_STREAMING
#define GDSP_STREAMING 1
#endif
typedef double gdsp_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 gdsp_ld2 (const double2* p)
	{
#if GDSP_STREAMING
	const gdsp_v2d t = __builtin_nontemporal_load (reinterpret_cast<const gdsp_v2d*> (p));
	return make_double2 (t.x, t.y);
#else
	return *p;
#endif
	}
__device__ __forceinline__ void gdsp_st2 (double2* p, double2 v)
	{
#if GDSP_STREAMING
	gdsp_v2d t;  t.x = v.x;  t.y = v.y;
	__builtin_nontemporal_store (t, reinterpret_cast<gdsp_v2d*> (p));
#else
	*p = v;
#endif
	}

// MI355X: 256 CUs in 8 XCDs; workgroups are dealt round-robin over the XCDs
// (block b and b+8 share an L2).  Remap a linear block id so that each XCD
// walks one contiguous eighth of the tiles: neighbouring tiles (which share
// their window halo) then meet in the same L2.  Speed only, never correctness.
#define GDSP_NUM_XCD 8
__device__ __forceinline__ uint32_t gdsp_xcd_tile (uint32_t b, uint32_t nblocks)
	{
	uint32_t per  = nblocks / GDSP_NUM_XCD;        // tiles per XCD in the even part
	uint32_t even = per * GDSP_NUM_XCD;
	if (b >= even) return b;                       // ragged tail keeps its id
	return (b % GDSP_NUM_XCD) * per + (b / GDSP_NUM_XCD);
	}

// ---- one launch for many vectors (genodsp.c:909-921 applies an operator chromosome by chromosome; a device that owns
// several chromosomes, or several stretches, gets ONE grid over all of them: no ramp and drain between 24 launches).
// The table travels in the kernarg segment (no device allocation, no copy to wait for): vector s owns the tiles
// [tile0[s], tile0[s+1]) of the grid.  A block finds its vector by a short scalar search; XCD-contiguous tile order
// is applied to the whole grid, so each XCD still walks one contiguous stretch of (concatenated) tiles.
#define GDSP_BATCH_MAX 32
struct GdspBatch
	{
	const double* in[GDSP_BATCH_MAX];
	double*       out[GDSP_BATCH_MAX];
	uint32_t      n[GDSP_BATCH_MAX];
	uint32_t      tile0[GDSP_BATCH_MAX + 1];
	uint32_t      nvec;
	};

// -> tile index inside the vector; in / out / n are set to the vector's.  The whole offset table comes in with two
// scalar loads issued together and the search runs on registers: five compares, each followed by selects that keep the
// half of the table the answer lies in (31 s_cselect in all; unused entries hold the grid size, which no tile id
// reaches) -- two scalar-load latencies per block instead of the seven of a search that loads as it goes, which cost the
// hann kernel 6 % of its time.
__device__ __forceinline__ uint32_t gdsp_batch_tile (const GdspBatch& B, const double*& in, double*& out, uint32_t& n, uint32_t* vector = NULL)
	{
	static_assert (GDSP_BATCH_MAX == 32, "the search below is five levels deep");
	const uint32_t g = gdsp_xcd_tile (blockIdx.x, B.tile0[GDSP_BATCH_MAX]);
	uint32_t t32[32], t16[16], t8[8], t4[4], t2[2];
#pragma unroll
	for (int i=0 ; i<32 ; i++)
		{
		t32[i] = B.tile0[i];
		asm ("" : "+s" (t32[i]));                              // the table is loaded whole (else the selects below turn into loads from selected addresses)
		}
	const bool c16 = (t32[16] <= g);
#pragma unroll
	for (int i=0 ; i<16 ; i++) t16[i] = c16? t32[16+i] : t32[i];
	const bool c8 = (t16[8] <= g);
#pragma unroll
	for (int i=0 ; i<8 ; i++) t8[i] = c8? t16[8+i] : t16[i];
	const bool c4 = (t8[4] <= g);
#pragma unroll
	for (int i=0 ; i<4 ; i++) t4[i] = c4? t8[4+i] : t8[i];
	const bool c2 = (t4[2] <= g);
	t2[0] = c2? t4[2] : t4[0];  t2[1] = c2? t4[3] : t4[1];
	const bool c1 = (t2[1] <= g);
	const uint32_t first = c1? t2[1] : t2[0];                  // = tile0[v]
	const uint32_t v = (c16? 16u : 0u) + (c8? 8u : 0u) + (c4? 4u : 0u) + (c2? 2u : 0u) + (c1? 1u : 0u);
	in = B.in[v];  out = B.out[v];  n = B.n[v];
	if (vector != NULL) *vector = v;
	return g - first;
	}

// host: the table for `count` (<= GDSP_BATCH_MAX) non-empty vectors
template <typename TilesOf>
static inline void gdsp_batch_make (GdspBatch& B, const gdsp_batch_item* items, int count, TilesOf tilesOf)
	{
	B.tile0[0] = 0;
	for (int k=0 ; k<GDSP_BATCH_MAX ; k++)
		{
		if (k < count) { B.in[k] = items[k].d_in;  B.out[k] = items[k].d_out;  B.n[k] = items[k].n;  B.tile0[k+1] = B.tile0[k] + (uint32_t) tilesOf (items[k].n); }
		else           { B.in[k] = NULL;  B.out[k] = NULL;  B.n[k] = 0;  B.tile0[k+1] = B.tile0[k]; }
		}
	B.nvec = (uint32_t) count;
	}

// ---- `smooth = localmax|localmin` in the reference's arithmetic, filtered (gdsp_peaks.hip): what its launches share.
// Per vector: how many bases the filter queued for exact evaluation, whether the queue overflowed, what a probe of a few
// tiles counted, and how many bases the probe looked at.  A vector whose probe found more than PK_DIRECT_NUM/256 of its
// bases undecided (piecewise-constant depth: every base of a flat run ties) is evaluated base by base by the direct
// kernel instead -- decided on the device, the same way by every block of every launch.
struct GdspPeaksCtl { uint32_t count, overflow, probe, sampled, flat; };   // flat: bases of the probed tiles that would be written their run's value (gdsp_peaks.hip, CWM)
#define GDSP_PEAKS_DIRECT_NUM 3u
// (host and device decide by the same rule.  flat: more than an eighth of the probed bases would be written their run's
//  value -- sampled counts quarter bases -- and the vector takes the filter's form for flat stretches; in a vector that is
//  not flat those bases tie and are queued like any undecided base, four quarter bases each, so they count here)
__host__ __device__ __forceinline__ bool gdsp_peaks_flat_form (const GdspPeaksCtl& c) { return (uint64_t) c.flat * 32u > (uint64_t) c.sampled; }
__host__ __device__ __forceinline__ bool gdsp_peaks_takes_direct (const GdspPeaksCtl& c)
	{
	const uint64_t open = (uint64_t) c.probe + (gdsp_peaks_flat_form (c)? 0u : 4u * (uint64_t) c.flat);
	return open * 256u > (uint64_t) c.sampled * GDSP_PEAKS_DIRECT_NUM;
	}

// gdsp_fir.hip: the direct fused kernel over a table of vectors, gated: a block works only when its vector takes the direct
// route (probe) or its queue overflowed
int gdsp_fir_extrema_gated_launch (const gdsp_batch_item* items, int count, const GdspPeaksCtl* d_ctl, const double* h_taps,
                                   int fma, int h, int wantMax, double fill, void* stream);
// gdsp_peaks.hip: the filtered route for every vector of a table (W = 101, neighbourhoods of 3..15 bases)
bool gdsp_peaks_filter_available (uint32_t W, uint32_t N);
bool gdsp_peaks_filter_wanted_for_fma (void);                      // GDSP_PEAKS_FILTER=fma
int  gdsp_peaks_filter_batch (const gdsp_batch_item* items, int nitems, const double* h_taps, int fma, uint32_t N, int wantMax,
                              double fill, void* stream);

// host: launch(B, tiles) over tables of up to GDSP_BATCH_MAX vectors each until every item has been covered (empty
// vectors are skipped).  tilesOf(n) = tiles the kernel needs for a vector of n elements.
template <typename TilesOf, typename Launch>
static inline void gdsp_batch_run (const gdsp_batch_item* items, int nitems, TilesOf tilesOf, Launch launch)
	{
	int i = 0;
	while (i < nitems)
		{
		GdspBatch B;
		int k = 0;
		B.tile0[0] = 0;
		for ( ; (i<nitems) && (k<GDSP_BATCH_MAX) ; i++)
			{
			if (items[i].n == 0) continue;
			const uint64_t t = (uint64_t) B.tile0[k] + tilesOf (items[i].n);
			if ((t > 0x7FFFFFFFull) && (k > 0)) break;             // grid limit: the rest goes into the next launch
			B.in[k] = items[i].d_in;  B.out[k] = items[i].d_out;  B.n[k] = items[i].n;
			B.tile0[++k] = (uint32_t) t;
			}
		for (int j=k ; j<GDSP_BATCH_MAX ; j++) { B.in[j] = NULL;  B.out[j] = NULL;  B.n[j] = 0;  B.tile0[j+1] = B.tile0[k]; }
		B.nvec = (uint32_t) k;
		if (k > 0) launch (B, B.tile0[k]);
		}
	}

// Stage v[g0 .. g0+L) into LDS (zero / `pad` outside [0,n)).  g0 and L are even and v is
// 16-byte aligned, so interior tiles move as 16-byte words; up to STAGE_DEPTH loads per lane
// are issued before the first LDS store so that a workgroup keeps ~32 KiB in flight.
#define GDSP_STAGE_DEPTH 8
template <int THREADS>
__device__ __forceinline__ void gdsp_stage_f64 (double* lds, const double* __restrict__ v, uint32_t n,
                                                int64_t g0, int L, double pad)
	{
	if ((g0 >= 0) && (g0 + L <= (int64_t) n))
		{
		const double2* src = reinterpret_cast<const double2*> (v + g0);
		double2*       dst = reinterpret_cast<double2*> (lds);
		const int      np  = L / 2;
		for (int base=0 ; base<np ; base+=GDSP_STAGE_DEPTH*THREADS)
			{
			double2 r[GDSP_STAGE_DEPTH];
			// every lane loads unconditionally (index clamped): a predicated load would get its
			// own branch and an s_waitcnt vmcnt(0), serialising the whole batch
#pragma unroll
			for (int u=0 ; u<GDSP_STAGE_DEPTH ; u++)
				{ int p = base + u*THREADS + (int) threadIdx.x;  r[u] = gdsp_ld2 (&src[(p < np)? p : np-1]); }
#pragma unroll
			for (int u=0 ; u<GDSP_STAGE_DEPTH ; u++)
				{ int p = base + u*THREADS + (int) threadIdx.x;  if (p < np) dst[p] = r[u]; }
			}
		}
	else
		{
		for (int p=threadIdx.x ; p<L ; p+=THREADS)
			{
			int64_t g = g0 + p;
			lds[p] = ((g >= 0) && (g < (int64_t) n))? v[g] : pad;
			}
		}
	}

// order-preserving image of a double (radix select): -0.0 folded onto +0.0
__host__ __device__ __forceinline__ uint64_t gdsp_key_of (double v)
	{
	union { double d; uint64_t u; } c;
	c.d = v;
	if (c.u == 0x8000000000000000ULL) c.u = 0;
	return (c.u & 0x8000000000000000ULL)? ~c.u : (c.u | 0x8000000000000000ULL);
	}
__host__ __device__ __forceinline__ double gdsp_value_of (uint64_t key)
	{
	union { double d; uint64_t u; } c;
	c.u = (key & 0x8000000000000000ULL)? (key & 0x7FFFFFFFFFFFFFFFULL) : ~key;
	return c.d;
	}
