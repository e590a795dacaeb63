// gdsp_fir_slide.hip -- `smooth W=101` in the reference's arithmetic with every product computed ONCE.
//
// Reference: op_smooth_apply, sum.c:616-676.  out[i] = ((0 + w[0] x[i-50]) + w[1] x[i-49]) + ... + w[100] x[i+50], every
// product and every sum rounded (sum.c:655-663); the window is stored mirrored (sum.c:632-645), so w[k] and w[100-k] are
// the same double.  The direct kernel (gdsp_fir.hip) spends 101 multiplies and 101 adds per base and runs at the rate the
// FP64 pipe sustains: 0.29 of HBM.  But the ROUNDED PRODUCT w[m] x[j] (m < 50) appears twice: as term m of out[j+50-m]
// and as term 100-m of out[j-50+m].  Nothing about the reference's bits depends on who computes it, only on the order in
// which each output adds its terms.  So: walk the INPUTS in ascending order.  At input j, for every m <= 50, p = w[m] x[j]
// is added to the accumulator of out[j+50-m] (its term m: terms 0..m-1 came with inputs j-m..j-1) and, for m < 50, to
// the accumulator of out[j-50+m] (its term 100-m: the earlier ones came with the earlier inputs).  Every output receives
// its 101 terms in the reference's order, one per input step; 51 multiplies + 101 adds per base instead of 202 operations.
//
// A thread therefore owns a STRIP of consecutive outputs and slides along it with the 101 live accumulators in
// registers.  Two refinements make that fit a wave:
//   * parity classes.  out[j+50-m] and out[j-50+m] differ by 100-2m: same parity.  A thread of class PI keeps only the
//     outputs o = PI (mod 2) of the strip (51 live accumulators instead of 101) and at input t uses the taps
//     m = t-PI (mod 2): 26 or 25 multiplies and 51 or 50 adds.  The two classes of a strip are two waves, which read the
//     same inputs and share nothing else.
//   * blocks of FS_B inputs, fully unrolled with the accumulators as a register array indexed by compile-time numbers;
//     after a block the FS_B/2 finished outputs are stored and the array is shifted down by as many places (50 64-bit
//     moves per 16 inputs: 4 % on top of the arithmetic).  The 51 distinct taps sit in vector registers (as scalar
//     operands their 102 words do not fit the scalar file next to everything else, and the compiler then spills them
//     through v_writelane).
// Memory: a lane reads its own strip (16 bytes per load, the line's first touch issued a block ahead) and writes its own
// outputs 8 bytes at a time; lines are completed in the L2.  At one load per ~600 cycles of arithmetic per wave this is
// nowhere near a limit: the kernel is bound by the FP64 pipe, as the direct kernel is, with a quarter fewer operations.
// A strip needs its 100 inputs of run-in like any tile needs its halo: strips of 4096 outputs pay 2.7 % for it.
//
// Bit-identical to gdsp_fir.hip's exact mode, and so to the reference, on every input (zero padding adds +0.0 terms:
// the running sum starts at +0.0 and can never become -0.0; NaN and infinities propagate through the same operations).

#include <string.h>
#include <stdlib.h>
#include "gdsp_common.h"

#define FS_THREADS 256
#define FS_B       16                            // inputs per unrolled block
#define FS_NA      (50 + FS_B/2)                 // accumulators of a parity class alive during a block
#define FS_LEAD    50                            // inputs ahead of a strip's first output: the half window
#define FS_SHIFT   14                            // strips start FS_SHIFT outputs before multiples of S: their first INPUT (s0 - 50) is then a multiple of 16

struct FsTaps { double w[51]; };                 // w[0..50]; w[100-m] = w[m]

// one vector, or a table of them: strips are numbered through the whole table
struct FsTable
	{
	const double* in[GDSP_BATCH_MAX];
	double*       out[GDSP_BATCH_MAX];
	uint32_t      n[GDSP_BATCH_MAX];
	uint32_t      strip0[GDSP_BATCH_MAX + 1];    // first strip of vector v
	uint32_t      nvec, S, NB;                   // outputs per strip (a multiple of 16), blocks per strip
	};

// the block: inputs x[0..15] (t = t0 .. t0+15, t0 a multiple of 16), accumulators a <-> output o = t0 - 100 + PI + 2a
// taps m < FS_MS are scalar operands (kernel arguments, kept in scalar registers), the rest vector registers.
// The operations are written as instructions: left to itself the compiler hoists the 26 products of an input above
// their 51 adds (to hide the multiplier's latency) and then has nowhere to keep them -- 110 to 430 registers spilled in
// every arrangement tried.  Here a product is issued two places ahead of its adds and lives in one of three registers.
#ifndef FS_MS
#define FS_MS 16
#endif
#define FS_AHEAD 2
__device__ __forceinline__ double fs_mul_v (double w, double x)
	{ double p;  asm volatile ("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(w), "v"(x));  return p; }
__device__ __forceinline__ double fs_mul_s (double w, double x)
	{ double p;  asm volatile ("v_mul_f64 %0, %1, %2" : "=v"(p) : "s"(w), "v"(x));  return p; }
__device__ __forceinline__ void fs_add (double& acc, double p)
	{ asm volatile ("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(p)); }

template <int PI>
__device__ __forceinline__ void fs_pair (double (&acc)[FS_NA], const FsTaps& ws, const double (&w)[51 - FS_MS], double x0, double x1, const int d0)
	{
#pragma unroll
	for (int e=0 ; e<2 ; e++)
		{
		const int    d = d0 + e;
		const double x = e? x1 : x0;
		const int    first = (d - PI) & 1, cnt = (50 - first) / 2 + 1;     // taps first, first+2, .. <= 50
		double p[FS_AHEAD + 1];
		auto mul = [&] (int idx)
			{
			const int m = first + 2*idx;
			p[idx % (FS_AHEAD + 1)] = (m < FS_MS)? fs_mul_s (ws.w[(m < FS_MS)? m : 0], x) : fs_mul_v (w[(m < FS_MS)? 0 : m - FS_MS], x);
			};
#pragma unroll
		for (int idx=0 ; idx<FS_AHEAD ; idx++) { if (idx < cnt) mul (idx); }
#pragma unroll
		for (int idx=0 ; idx<cnt ; idx++)
			{
			if (idx + FS_AHEAD < cnt) mul (idx + FS_AHEAD);
			const int m  = first + 2*idx;
			const int a1 = (100 + d - m - PI) / 2;                 // out[t - m]: its term m
			fs_add (acc[a1], p[idx % (FS_AHEAD + 1)]);
			if (m < 50) { const int a2 = (d + m - PI) / 2;  fs_add (acc[a2], p[idx % (FS_AHEAD + 1)]); }      // out[t - 100 + m]: its term 100 - m
			}
		}
	}

// one strip, one parity class.  in/out/n: the strip's vector; s0: its first output (even; s0 - FS_LEAD is a multiple of 16).
// Input t of the strip (t = 0 .. 16 NB - 1) is in[s0 - 50 + t]: term 0 of output o = t, term 100 of output o = t - 100.
template <int PI, bool GUARD>
__device__ __forceinline__ void fs_strip (const double* __restrict__ in, double* __restrict__ out, const int64_t n, const int64_t s0,
                                          const int S, const int NB, const FsTaps& ws, const double* __restrict__ wLds)
	{
	double w[51 - FS_MS];
#pragma unroll
	for (int m=FS_MS ; m<51 ; m++) w[m - FS_MS] = wLds[m];         // (LDS reads land in vector registers)
	double acc[FS_NA];
#pragma unroll
	for (int a=0 ; a<FS_NA ; a++) acc[a] = 0.0;

	// !GUARD: every input of the strip lies inside the vector (and with them every output): plain 16-byte loads, and the
	// only question at a store is whether the output belongs to this strip -- the same answer in every lane
	const int64_t g0 = s0 - FS_LEAD;
	const double* __restrict__ inp  = in + g0;                     // input t of the strip
	double* __restrict__       outp = out + s0;                    // output o of the strip
	auto load2 = [&] (int t) -> double2
		{
		if (!GUARD) return *reinterpret_cast<const double2*> (inp + t);
		const int64_t g = g0 + t;
		double2 r;
		r.x = ((g     >= 0) && (g     < n))? in[g]     : 0.0;
		r.y = ((g + 1 >= 0) && (g + 1 < n))? in[g + 1] : 0.0;
		return r;
		};
	double2 cur = load2 (0);
	for (int b=0 ; b<NB ; b++)
		{
		const int tb = b * FS_B;
#pragma unroll
		for (int dp=0 ; dp<FS_B/2 ; dp++)
			{
			// the next pair is on its way while this one is worked on (past the strip's last block: the last pair again)
			const int tn = (dp + 1 < FS_B/2)? tb + 2*(dp + 1) : ((b + 1 < NB)? tb + FS_B : tb);
			const double2 nxt = load2 (tn);
			fs_pair<PI> (acc, ws, w, cur.x, cur.y, 2*dp);
			cur = nxt;
			}
		// outputs finished by this block: accumulator a = 0 .. 7 <-> o = 16 b + PI + 2 a - 100 (its term 100 came with
		// input t = o + 100 <= 16 b + 15)
#pragma unroll
		for (int a=0 ; a<FS_B/2 ; a++)
			{
			const int o = tb + PI + 2*a - 50 - FS_LEAD;
			if ((o >= 0) && (o < S) && (!GUARD || ((s0 + o >= 0) && (s0 + o < n)))) outp[o] = acc[a];
			}
#pragma unroll
		for (int a=0 ; a<FS_NA - FS_B/2 ; a++) acc[a] = acc[a + FS_B/2];
#pragma unroll
		for (int a=FS_NA - FS_B/2 ; a<FS_NA ; a++) acc[a] = 0.0;
		}
	}

// waves 0 and 1 of a workgroup are the two classes of 64 strips, waves 2 and 3 of the next 64
__global__ __launch_bounds__(FS_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fir_slide_kernel (FsTable T, FsTaps taps)
	{
	__shared__ double wLds[51];
	if (threadIdx.x < 51) wLds[threadIdx.x] = taps.w[threadIdx.x];
	__syncthreads ();
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t strip = (blockIdx.x * (FS_THREADS / 128) + (wave >> 1)) * 64 + lane;
	if (strip >= T.strip0[T.nvec]) return;
	uint32_t v = 0;
	while ((v + 1 < T.nvec) && (T.strip0[v + 1] <= strip)) v++;  // (per lane; once per strip)
	const double* in  = T.in[v];
	double*       out = T.out[v];
	const int64_t n   = (int64_t) T.n[v];
	const int64_t s0   = (int64_t) (strip - T.strip0[v]) * T.S - FS_SHIFT;
	const int64_t lo   = s0 - FS_LEAD, hi = lo + (int64_t) T.NB * FS_B + 2;
	const bool    edge = !((lo >= 0) && (hi <= n));
	const int S = (int) T.S, NB = (int) T.NB;
	if (wave & 1) { if (edge) fs_strip<1, true> (in, out, n, s0, S, NB, taps, wLds);  else fs_strip<1, false> (in, out, n, s0, S, NB, taps, wLds); }
	else          { if (edge) fs_strip<0, true> (in, out, n, s0, S, NB, taps, wLds);  else fs_strip<0, false> (in, out, n, s0, S, NB, taps, wLds); }
	}

// ---- the same walk with the memory side taken off the lanes' critical path (GDSP_FIR_SLIDE=2).  In the kernel above a
// lane fetches its own strip 16 bytes at a time, one pair ahead of the arithmetic: every 128-byte line is a miss the
// wave waits for (90 Gbases/s, and slower the further apart the strips lie).  Here a workgroup is ONE pair of waves --
// the two parity classes of 64 strips -- and a block of 16 inputs per strip comes in as eight LDS-DMA loads
// (global_load_lds_dwordx4: per-lane source address, lane-linear destination: image [pair][lane], read back with
// conflict-free ds_read_b128), four issued by each wave, a whole block ahead of its use: two input buffers, one barrier
// per block.  The 16 outputs a block finishes per strip (8 from each class) meet in LDS as 8 pairs and leave as 16-byte
// stores, four pairs per wave.  Strips at a vector's ends fill their LDS slots through guarded register loads instead.
#define FS2_THREADS 128
template <int PI>
__device__ __forceinline__ void fs2_strip (const double* __restrict__ in, double* __restrict__ out, const int64_t n, const int64_t s0,
                                           const int S, const int NB, const bool edge, const bool idle, const FsTaps& ws,
                                           const double* __restrict__ wLds, double2 (*inBuf)[FS_B/2][64], double2 (*outBuf)[FS_B/2][64])
	{
	const int lane = threadIdx.x & 63;
	double w[51 - FS_MS];
#pragma unroll
	for (int m=FS_MS ; m<51 ; m++) w[m - FS_MS] = wLds[m];
	double acc[FS_NA];
#pragma unroll
	for (int a=0 ; a<FS_NA ; a++) acc[a] = 0.0;
	const int64_t g0 = s0 - FS_LEAD;
	const double* __restrict__ inp  = in + g0;                     // input t of the strip
	double* __restrict__       outp = out + s0;                    // output o of the strip

	// this wave's half of block b's eight pairs, into buffer `buf`
	auto fetch = [&] (int b, int buf)
		{
#pragma unroll
		for (int kk=0 ; kk<FS_B/4 ; kk++)
			{
			const int k = PI * (FS_B/4) + kk, t = b * FS_B + 2*k;
			if (!edge)
				__builtin_amdgcn_global_load_lds ((const __attribute__((address_space(1))) void*) (inp + t),
				                                  (__attribute__((address_space(3))) void*) &inBuf[buf][k][0], 16, 0, 0);
			else
				{
				const int64_t g = g0 + t;
				double2 r;
				r.x = ((g     >= 0) && (g     < n))? in[g]     : 0.0;
				r.y = ((g + 1 >= 0) && (g + 1 < n))? in[g + 1] : 0.0;
				inBuf[buf][k][lane] = r;
				}
			}
		};
	fetch (0, 0);
	// The LDS-DMA loads land in LDS behind this wave's vmcnt: the OTHER wave reads those slots after the barrier, and a
	// workgroup barrier does not wait for one wave's outstanding loads on another's behalf -- the issuing wave drains them
	// itself before it joins (here and before the barrier that publishes block b+1, below)
	__builtin_amdgcn_s_waitcnt (0x0f70);                           // vmcnt(0), the other counters left alone (gfx9 encoding)
	__syncthreads ();                                              // (the taps too)
	if (NB > 1) fetch (1, 1);
	for (int b=0 ; b<NB ; b++)
		{
		const int buf = b & 1, tb = b * FS_B;
		double2 cur = inBuf[buf][0][lane];
#pragma unroll
		for (int dp=0 ; dp<FS_B/2 ; dp++)
			{
			const double2 nxt = inBuf[buf][(dp + 1 < FS_B/2)? dp + 1 : dp][lane];
			fs_pair<PI> (acc, ws, w, cur.x, cur.y, 2*dp);
			cur = nxt;
			}
		// finished: accumulator a <-> o = tb + PI + 2a - 100: element PI of the strip's pair a
#pragma unroll
		for (int a=0 ; a<FS_B/2 ; a++) reinterpret_cast<double*> (&outBuf[buf][a][lane])[PI] = acc[a];
#pragma unroll
		for (int a=0 ; a<FS_NA - FS_B/2 ; a++) acc[a] = acc[a + FS_B/2];
#pragma unroll
		for (int a=FS_NA - FS_B/2 ; a<FS_NA ; a++) acc[a] = 0.0;
		__builtin_amdgcn_s_waitcnt (0x0f70);                       // vmcnt(0): this wave's half of block b+1 has landed (see above)
		__syncthreads ();                                          // block b's inputs are consumed and its outputs paired; block b+1's inputs have landed
		if (b + 2 < NB) fetch (b + 2, buf);
#pragma unroll
		for (int kk=0 ; kk<FS_B/4 ; kk++)
			{
			const int a = PI * (FS_B/4) + kk, o = tb + 2*a - 50 - FS_LEAD;
			if ((o < 0) || (o >= S) || idle) continue;
			const double2 r = outBuf[buf][a][lane];
			if (!edge) *reinterpret_cast<double2*> (outp + o) = r;
			else
				{
				if ((s0 + o     >= 0) && (s0 + o     < n)) outp[o]     = r.x;
				if ((s0 + o + 1 >= 0) && (s0 + o + 1 < n)) outp[o + 1] = r.y;
				}
			}
		}
	}

__global__ __launch_bounds__(FS2_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fir_slide_lds_kernel (FsTable T, FsTaps taps)
	{
	__shared__ __attribute__((aligned(16))) double2 inBuf[2][FS_B/2][64];
	__shared__ __attribute__((aligned(16))) double2 outBuf[2][FS_B/2][64];
	__shared__ double wLds[51];
	if (threadIdx.x < 51) wLds[threadIdx.x] = taps.w[threadIdx.x];
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	uint32_t   strip = blockIdx.x * 64 + lane;
	const bool idle  = (strip >= T.strip0[T.nvec]);                // (such a lane walks the table's last strip and stores nothing)
	if (idle) strip = T.strip0[T.nvec] - 1;
	uint32_t v = 0;
	while ((v + 1 < T.nvec) && (T.strip0[v + 1] <= strip)) v++;
	const double* in  = T.in[v];
	double*       out = T.out[v];
	const int64_t n   = (int64_t) T.n[v];
	const int64_t s0   = (int64_t) (strip - T.strip0[v]) * T.S - FS_SHIFT;
	const int64_t lo   = s0 - FS_LEAD, hi = lo + (int64_t) T.NB * FS_B;
	const bool    edge = !((lo >= 0) && (hi <= n));
	if (wave) fs2_strip<1> (in, out, n, s0, (int) T.S, (int) T.NB, edge, idle, taps, wLds, inBuf, outBuf);
	else      fs2_strip<0> (in, out, n, s0, (int) T.S, (int) T.NB, edge, idle, taps, wLds, inBuf, outBuf);
	}

// ------------------------------------------------------------------- host ----
// GDSP_FIR_SLIDE=1: this kernel for exact W=101 (A/B against the direct kernel; opt-in until measured)
bool gdsp_fir_slide_wanted (uint64_t bases)
	{
	const char* e = getenv ("GDSP_FIR_SLIDE");
	return (e != NULL) && (strcmp (e, "0") != 0) && (bases >= 1);
	}

// strips: long enough that the 100 inputs of run-in are a few per cent, short enough that the launch fills the chip a
// few times over (1024 wave pairs of 64 strips are in flight at once)
static uint32_t fs_strip_length (uint64_t totalBases)
	{
	const char* e = getenv ("GDSP_FIR_SLIDE_STRIP");
	if ((e != NULL) && (atoll (e) >= 16)) return (uint32_t) ((atoll (e) + 15) / 16 * 16);
	const uint64_t inFlight = 1024ull * 64;
	uint64_t S = totalBases / (inFlight * 6);                      // six rounds at least
	if (S > 4096) S = 4096;
	if (S < 512)  S = 512;
	return (uint32_t) ((S + 15) / 16 * 16);
	}

int gdsp_fir_slide_batch (const gdsp_batch_item* items, int nitems, const double* h_taps, void* stream)
	{
	FsTaps taps;
	memcpy (taps.w, h_taps, sizeof(taps.w));                       // w[0..50]
	uint64_t total = 0;
	for (int i=0 ; i<nitems ; i++) total += items[i].n;
	const uint32_t S  = fs_strip_length (total);
	const uint32_t NB = (S + (FS_LEAD + 50) + FS_B - 1) / FS_B;     // the last output o = S-1 is finished by input t = S-1 + FS_LEAD + 50
	hipStream_t s = gdsp_stream (stream);
	int i = 0;
	while (i < nitems)
		{
		FsTable T;
		memset (&T, 0, sizeof(T));
		T.S = S;  T.NB = NB;
		int k = 0;
		for ( ; (i<nitems) && (k<GDSP_BATCH_MAX) ; i++)
			{
			if (items[i].n == 0) continue;
			const uint64_t strips = ((uint64_t) items[i].n + FS_SHIFT + S - 1) / S;
			T.in[k] = items[i].d_in;  T.out[k] = items[i].d_out;  T.n[k] = items[i].n;
			T.strip0[k+1] = T.strip0[k] + (uint32_t) strips;
			k++;
			}
		for (int j=k ; j<GDSP_BATCH_MAX ; j++) T.strip0[j+1] = T.strip0[k];
		T.nvec = (uint32_t) k;
		if (k == 0) break;
		const char* form = getenv ("GDSP_FIR_SLIDE");
		if ((form != NULL) && (strcmp (form, "2") == 0))
			hipLaunchKernelGGL (fir_slide_lds_kernel, dim3((T.strip0[k] + 63) / 64), dim3(FS2_THREADS), 0, s, T, taps);
		else
			hipLaunchKernelGGL (fir_slide_kernel, dim3((T.strip0[k] + 127) / 128), dim3(FS_THREADS), 0, s, T, taps);
		}
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}
