// tools/stream_shapes.hip -- what shape of a 1:1 read/write stream reaches which rate on this part.
//
// `smooth --smooth=hann` moves its 16 B/base at 6.2-6.4 TB/s, the in-place pointwise operators at 5.8-6.0, the fused
// `percentile = binarize` pass at 5.1-5.6 with a tenth of the arithmetic.  This program runs the same trivial body
// (y = x > T ? 1 : 0 on doubles, 16-byte accesses) in the skeletons those kernels use, on one 249 Mbp vector:
//   tile    one 4096-element tile per workgroup, all eight loads of a lane up front, eight stores at the end
//           (hann_blocks_kernel's traffic pattern), tiles in XCD-contiguous order or in launch order;
//           with `lds` KiB of LDS per workgroup to hold the occupancy where hann's is (44 KiB: three per CU);
//           optionally through an LDS image and two barriers like the real kernel
//   walk    a workgroup walks tiles b, b+G, b+2G, ... in two halves with loads always in flight and a store per
//           pair as it is computed (pc_partition_tab_kernel's pattern), five workgroups per CU
// each out of place and in place, with non-temporal or plain accesses.
//   hipcc --offload-arch=gfx950 -O3 -o stream_shapes tools/stream_shapes.hip && ./stream_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf (stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString (e_)); exit (1); } } while (0)
#define THREADS 256
#define TILE    4096

template <bool NT> __device__ __forceinline__ double2 ld2 (const double2* p)
	{ return NT? make_double2 (__builtin_nontemporal_load (&p->x), __builtin_nontemporal_load (&p->y)) : *p; }
template <bool NT> __device__ __forceinline__ void st2 (double2* p, double2 v)
	{ if (NT) { __builtin_nontemporal_store (v.x, &p->x);  __builtin_nontemporal_store (v.y, &p->y); } else *p = v; }
__device__ __forceinline__ double2 body (double2 v, double T) { return make_double2 ((v.x > T)? 1.0 : 0.0, (v.y > T)? 1.0 : 0.0); }

__device__ __forceinline__ uint32_t xcd_tile (uint32_t b, uint32_t nblocks)
	{
	const uint32_t per = nblocks / 8, even = per * 8;
	if (b >= even) return b;
	return (b % 8) * per + (b / 8);
	}

template <bool NT, bool XCD, bool STAGE>
__global__ __launch_bounds__(THREADS)
void tile_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t ntiles, double T)
	{
	extern __shared__ double lds[];
	const uint32_t tile = XCD? xcd_tile (blockIdx.x, ntiles) : blockIdx.x;
	const double2* src = reinterpret_cast<const double2*> (in + (size_t) tile * TILE) + threadIdx.x;
	double2*       dst = reinterpret_cast<double2*> (out + (size_t) tile * TILE) + threadIdx.x;
	double2 r[8];
#pragma unroll
	for (int u=0 ; u<8 ; u++) r[u] = ld2<NT> (&src[u*THREADS]);
	if (STAGE)
		{
#pragma unroll
		for (int u=0 ; u<8 ; u++) { const int e = 2 * (u*THREADS + (int) threadIdx.x);  lds[e + (e >> 4)] = r[u].x;  lds[e + 1 + ((e + 1) >> 4)] = r[u].y; }
		__syncthreads ();
		double a[16];
#pragma unroll
		for (int i=0 ; i<16 ; i++) a[i] = lds[threadIdx.x * 17 + i];
		__syncthreads ();
#pragma unroll
		for (int i=0 ; i<16 ; i++) lds[threadIdx.x * 17 + i] = (a[i] > T)? 1.0 : 0.0;
		__syncthreads ();
#pragma unroll
		for (int u=0 ; u<8 ; u++) { const int e = 2 * (u*THREADS + (int) threadIdx.x);  r[u] = make_double2 (lds[e + (e >> 4)], lds[e + 1 + ((e + 1) >> 4)]); }
		}
	else
		{
#pragma unroll
		for (int u=0 ; u<8 ; u++) r[u] = body (r[u], T);
		}
#pragma unroll
	for (int u=0 ; u<8 ; u++) st2<NT> (&dst[u*THREADS], r[u]);
	}

// the tile pattern with ELEMS elements per workgroup (ELEMS/512 loads per lane) and what a counting kernel would do at
// the end of every workgroup: EPI 0 nothing; 1 one returning atomic on ONE address; 2 on one of eight addresses;
// 3 a record of nine words stored; 4 = 2 + 3
template <int ELEMS, int EPI, int WGS>
__global__ __launch_bounds__(THREADS, WGS)
void big_tile_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t ntiles, double T, unsigned long long* ctr, uint32_t* rec)
	{
	constexpr int L = ELEMS / 512;
	const uint32_t tile = xcd_tile (blockIdx.x, ntiles);
	const double2* src = reinterpret_cast<const double2*> (in + (size_t) tile * ELEMS) + threadIdx.x;
	double2*       dst = reinterpret_cast<double2*> (out + (size_t) tile * ELEMS) + threadIdx.x;
	double2 r[L];
#pragma unroll
	for (int u=0 ; u<L ; u++) r[u] = ld2<true> (&src[u*THREADS]);
	uint32_t cnt = 0;
#pragma unroll
	for (int u=0 ; u<L ; u++) { r[u] = body (r[u], T);  cnt += (r[u].x != 0.0) + (r[u].y != 0.0); }
	unsigned long long got = 0;
	if ((EPI == 1) && (threadIdx.x == 0)) got = atomicAdd (&ctr[0], (unsigned long long) (cnt & 7));
	if (((EPI == 2) || (EPI == 4)) && (threadIdx.x == 0)) got = atomicAdd (&ctr[16 * (blockIdx.x & 7)], (unsigned long long) (cnt & 7));
	if (((EPI == 3) || (EPI == 4)) && (threadIdx.x < 9)) rec[(size_t) tile * 9 + threadIdx.x] = cnt;
#pragma unroll
	for (int u=0 ; u<L ; u++) st2<true> (&dst[u*THREADS], r[u]);
	if ((EPI != 0) && (EPI != 3) && (threadIdx.x == 0) && (got == 0xFFFFFFFFFFFFull)) out[0] = 1.0;      // (the atomic's value is waited for)
	}

template <bool NT>
__global__ __launch_bounds__(THREADS, 5)
void walk_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t ntiles, double T)
	{
	const uint32_t step = gridDim.x;
	uint32_t tile = blockIdx.x;
	auto load_half = [&] (uint32_t t, int half, double2 (&d)[4])
		{
		const double2* p = reinterpret_cast<const double2*> (in + (size_t) t * TILE) + threadIdx.x + half * 4 * THREADS;
#pragma unroll
		for (int u=0 ; u<4 ; u++) d[u] = ld2<NT> (&p[u*THREADS]);
		};
	auto work_half = [&] (uint32_t t, int half, double2 (&d)[4])
		{
		double2* q = reinterpret_cast<double2*> (out + (size_t) t * TILE) + threadIdx.x + half * 4 * THREADS;
#pragma unroll
		for (int u=0 ; u<4 ; u++) st2<NT> (&q[u*THREADS], body (d[u], T));
		};
	double2 A[4], B[4];
	if (tile < ntiles) load_half (tile, 0, A);
	while (tile < ntiles)
		{
		load_half (tile, 1, B);
		work_half (tile, 0, A);
		const uint32_t next = tile + step;
		if (next < ntiles) load_half (next, 0, A);
		work_half (tile, 1, B);
		tile = next;
		}
	}

// a workgroup takes PER tiles, whole tiles at a time: eight loads, the body, eight stores, then the next tile.
//   STRIDED: tiles b, b+G, ...;  else a contiguous stretch of PER tiles, the stretches in XCD-contiguous order
//   AHEAD:   the next tile's loads are issued before the current tile's stores (a second set of registers)
template <bool NT, bool STRIDED, bool AHEAD, int WGS>
__global__ __launch_bounds__(THREADS, WGS)
void loop_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t ntiles, uint32_t per, double T)
	{
	const uint32_t first = STRIDED? blockIdx.x : xcd_tile (blockIdx.x, gridDim.x) * per;
	const uint32_t step  = STRIDED? gridDim.x : 1;
	auto load = [&] (uint32_t t, double2 (&d)[8])
		{
		const double2* p = reinterpret_cast<const double2*> (in + (size_t) t * TILE) + threadIdx.x;
#pragma unroll
		for (int u=0 ; u<8 ; u++) d[u] = ld2<NT> (&p[u*THREADS]);
		};
	auto store = [&] (uint32_t t, double2 (&d)[8])
		{
		double2* q = reinterpret_cast<double2*> (out + (size_t) t * TILE) + threadIdx.x;
#pragma unroll
		for (int u=0 ; u<8 ; u++) st2<NT> (&q[u*THREADS], body (d[u], T));
		};
	double2 A[8], B[8];
	uint32_t t = first;
	if (t < ntiles) load (t, A);
	for (uint32_t k=0 ; (k<per) && (t<ntiles) ; k++)
		{
		const uint32_t next = t + step;
		const bool more = (k + 1 < per) && (next < ntiles);
		if (AHEAD) { if (more) load (next, B);  store (t, A);  if (more) { for (int u=0 ; u<8 ; u++) A[u] = B[u]; } }
		else       { store (t, A);  if (more) load (next, A); }
		t = next;
		}
	}

// persistent workgroups that TAKE their tiles from a counter, GRAB consecutive tiles at a time (whole tiles: eight loads,
// the body, eight stores): the tiles in flight form one compact advancing window, as with one short workgroup per tile
template <bool NT, int WGS, bool AHEAD>
__global__ __launch_bounds__(THREADS, WGS)
void take_kernel (const double* __restrict__ in, double* __restrict__ out, uint32_t ntiles, uint32_t grab, unsigned int* next, double T)
	{
	__shared__ uint32_t sFirst;
	auto load = [&] (uint32_t t, double2 (&d)[8])
		{
		const double2* p = reinterpret_cast<const double2*> (in + (size_t) t * TILE) + threadIdx.x;
#pragma unroll
		for (int u=0 ; u<8 ; u++) d[u] = ld2<NT> (&p[u*THREADS]);
		};
	auto store = [&] (uint32_t t, double2 (&d)[8])
		{
		double2* q = reinterpret_cast<double2*> (out + (size_t) t * TILE) + threadIdx.x;
#pragma unroll
		for (int u=0 ; u<8 ; u++) st2<NT> (&q[u*THREADS], body (d[u], T));
		};
	for ( ; ; )
		{
		if (threadIdx.x == 0) sFirst = atomicAdd (next, grab);
		__syncthreads ();
		const uint32_t first = sFirst;
		__syncthreads ();
		if (first >= ntiles) break;
		const uint32_t last = (first + grab < ntiles)? first + grab : ntiles;
		double2 A[8], B[8];
		load (first, A);
		for (uint32_t t=first ; t<last ; t++)
			{
			const bool more = (t + 1 < last);
			if (AHEAD) { if (more) load (t + 1, B);  store (t, A);  if (more) { for (int u=0 ; u<8 ; u++) A[u] = B[u]; } }
			else       { store (t, A);  if (more) load (t + 1, A); }
			}
		}
	}

static double timed (void (*launch) (void), int reps)
	{
	hipEvent_t e0, e1;
	CHECK (hipEventCreate (&e0));  CHECK (hipEventCreate (&e1));
	launch ();
	CHECK (hipDeviceSynchronize ());
	double best = 1e30;
	for (int r=0 ; r<reps ; r++)
		{
		CHECK (hipEventRecord (e0, 0));
		for (int k=0 ; k<5 ; k++) launch ();
		CHECK (hipEventRecord (e1, 0));
		CHECK (hipEventSynchronize (e1));
		float ms;
		CHECK (hipEventElapsedTime (&ms, e0, e1));
		if (ms / 5 < best) best = ms / 5;
		}
	return best;
	}

static double *A, *B;
static uint32_t NT_;
static size_t   LDS;
static bool     INPLACE;
#define TILE_LAUNCH(NTv, XCDv, STAGEv) [] () { hipLaunchKernelGGL ((tile_kernel<NTv, XCDv, STAGEv>), dim3(NT_), dim3(THREADS), LDS, 0, A, INPLACE? A : B, NT_, 3.0); }
static uint32_t WALK_BLOCKS;
static uint32_t LOOP_PER;
#define LOOP_LAUNCH(STRv, AHv, WGSv) [] () { hipLaunchKernelGGL ((loop_kernel<true, STRv, AHv, WGSv>), dim3((NT_ + LOOP_PER - 1) / LOOP_PER), dim3(THREADS), 0, 0, A, INPLACE? A : B, NT_, LOOP_PER, 3.0); }
static unsigned long long* CTR;
static uint32_t* REC;
#define BIG_LAUNCH(ELv, EPIv, WGSv) [] () { hipLaunchKernelGGL ((big_tile_kernel<ELv, EPIv, WGSv>), dim3((uint32_t) ((size_t) NT_ * TILE / ELv)), dim3(THREADS), 0, 0, A, INPLACE? A : B, (uint32_t) ((size_t) NT_ * TILE / ELv), 3.0, CTR, REC); }
static unsigned int* NEXT;
static uint32_t GRAB, TAKE_WGS;
#define TAKE_LAUNCH(WGSv, AHv) [] () { CHECK (hipMemsetAsync (NEXT, 0, 4, 0));  hipLaunchKernelGGL ((take_kernel<true, WGSv, AHv>), dim3(TAKE_WGS), dim3(THREADS), 0, 0, A, INPLACE? A : B, NT_, GRAB, NEXT, 3.0); }
#define WALK_LAUNCH(NTv) [] () { hipLaunchKernelGGL ((walk_kernel<NTv>), dim3(WALK_BLOCKS), dim3(THREADS), 0, 0, A, INPLACE? A : B, NT_, 3.0); }

int main (void)
	{
	const size_t n = (size_t) 60784 * TILE;                        // 248 958 976 elements: 1.99 GB a side
	NT_ = (uint32_t) (n / TILE);
	CHECK (hipMalloc ((void**) &A, n * 8));  CHECK (hipMalloc ((void**) &B, n * 8));
	CHECK (hipMemset (A, 0x3f, n * 8));  CHECK (hipMemset (B, 0, n * 8));
	CHECK (hipMalloc ((void**) &NEXT, 64));
	CHECK (hipMalloc ((void**) &CTR, 4096));  CHECK (hipMemset (CTR, 0, 4096));
	CHECK (hipMalloc ((void**) &REC, (size_t) NT_ * 9 * 4 + 64));
	CHECK (hipFuncSetAttribute ((const void*) tile_kernel<true, true, true>,   hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
	CHECK (hipFuncSetAttribute ((const void*) tile_kernel<true, true, false>,  hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
	CHECK (hipFuncSetAttribute ((const void*) tile_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
	CHECK (hipFuncSetAttribute ((const void*) tile_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
	const double bytes = 16.0 * n;
	auto say = [&] (const char* what, double ms) { printf ("%-86s %7.3f ms  %6.0f GB/s  %.3f of 8 TB/s\n", what, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000); fflush (stdout); };
	for (int inplace=0 ; inplace<2 ; inplace++)
		{
		INPLACE = (inplace != 0);
		printf ("---- %s\n", INPLACE? "in place" : "out of place");
		const size_t ldsKiB[] = { 0, 20, 26, 35, 44, 60 };
		for (size_t k : ldsKiB)
			{
			LDS = k * 1024;
			char what[160];
			snprintf (what, sizeof(what), "tile, xcd order, non-temporal, %2zu KiB LDS per workgroup (%s per CU)", k, k == 0? "registers decide" : k == 20? "8" : k == 26? "6" : k == 35? "4" : k == 44? "3" : "2");
			say (what, timed (TILE_LAUNCH (true, true, false), 5));
			}
		LDS = 44 * 1024;
		say ("tile, LAUNCH order, non-temporal, 44 KiB (3 per CU)", timed (TILE_LAUNCH (true, false, false), 5));
		say ("tile, xcd order, PLAIN accesses, 44 KiB (3 per CU)", timed (TILE_LAUNCH (false, true, false), 5));
		say ("tile, xcd order, non-temporal, through an LDS image and three barriers, 44 KiB (3 per CU)", timed (TILE_LAUNCH (true, true, true), 5));
		const uint32_t perWG[] = { 4, 16, 64 };
		for (uint32_t p : perWG)
			{
			WALK_BLOCKS = (NT_ + p - 1) / p;
			char what[160];
			snprintf (what, sizeof(what), "walk, %2u tiles per workgroup, halves in flight, non-temporal, 5 per CU", p);
			say (what, timed (WALK_LAUNCH (true), 5));
			}
		WALK_BLOCKS = (NT_ + 15) / 16;
		say ("walk, 16 tiles per workgroup, halves in flight, PLAIN accesses, 5 per CU", timed (WALK_LAUNCH (false), 5));
		if (!INPLACE)
			{
			say ("tile of 4096, 5 per CU, no epilogue", timed (BIG_LAUNCH (4096, 0, 5), 5));
			say ("tile of 4096, 5 per CU, one returning atomic per workgroup on ONE address", timed (BIG_LAUNCH (4096, 1, 5), 5));
			say ("tile of 4096, 5 per CU, one returning atomic per workgroup on one of EIGHT addresses", timed (BIG_LAUNCH (4096, 2, 5), 5));
			say ("tile of 4096, 5 per CU, a nine-word record stored per workgroup", timed (BIG_LAUNCH (4096, 3, 5), 5));
			say ("tile of 4096, 5 per CU, atomic on one of eight + record", timed (BIG_LAUNCH (4096, 4, 5), 5));
			say ("tile of 8192, 5 per CU, no epilogue", timed (BIG_LAUNCH (8192, 0, 5), 5));
			say ("tile of 8192, 5 per CU, atomic on one of eight + record", timed (BIG_LAUNCH (8192, 4, 5), 5));
			say ("tile of 8192, 3 per CU, no epilogue", timed (BIG_LAUNCH (8192, 0, 3), 5));
			say ("tile of 16384, 3 per CU, no epilogue", timed (BIG_LAUNCH (16384, 0, 3), 5));
			say ("tile of 16384, 3 per CU, atomic on one of eight + record", timed (BIG_LAUNCH (16384, 4, 3), 5));
			}
		const uint32_t grabs[] = { 1, 2, 4 };
		for (uint32_t g : grabs)
			{
			GRAB = g;
			char what[160];
			TAKE_WGS = 256 * 5;  snprintf (what, sizeof(what), "take, %u tile(s) per grab, 5 persistent workgroups per CU, loads after stores", g);   say (what, timed (TAKE_LAUNCH (5, false), 5));
			TAKE_WGS = 256 * 5;  snprintf (what, sizeof(what), "take, %u tile(s) per grab, 5 persistent workgroups per CU, next loads before stores", g);   say (what, timed (TAKE_LAUNCH (5, true), 5));
			TAKE_WGS = 256 * 3;  snprintf (what, sizeof(what), "take, %u tile(s) per grab, 3 persistent workgroups per CU, loads after stores", g);   say (what, timed (TAKE_LAUNCH (3, false), 5));
			TAKE_WGS = 256 * 8;  snprintf (what, sizeof(what), "take, %u tile(s) per grab, 8 persistent workgroups per CU, loads after stores", g);   say (what, timed (TAKE_LAUNCH (8, false), 5));
			}
		const uint32_t pers[] = { 4, 16 };
		for (uint32_t p : pers)
			{
			LOOP_PER = p;
			char what[160];
			snprintf (what, sizeof(what), "loop, %2u whole tiles per workgroup, strided,     loads after stores,  5 per CU", p);   say (what, timed (LOOP_LAUNCH (true, false, 5), 5));
			snprintf (what, sizeof(what), "loop, %2u whole tiles per workgroup, strided,     next loads before stores, 5 per CU", p);   say (what, timed (LOOP_LAUNCH (true, true, 5), 5));
			snprintf (what, sizeof(what), "loop, %2u whole tiles per workgroup, contiguous (xcd order), loads after stores,  5 per CU", p);   say (what, timed (LOOP_LAUNCH (false, false, 5), 5));
			snprintf (what, sizeof(what), "loop, %2u whole tiles per workgroup, contiguous (xcd order), next loads before stores, 5 per CU", p);   say (what, timed (LOOP_LAUNCH (false, true, 5), 5));
			snprintf (what, sizeof(what), "loop, %2u whole tiles per workgroup, strided,     loads after stores,  3 per CU", p);   say (what, timed (LOOP_LAUNCH (true, false, 3), 5));
			snprintf (what, sizeof(what), "loop, %2u whole tiles per workgroup, strided,     loads after stores,  8 per CU", p);   say (what, timed (LOOP_LAUNCH (true, false, 8), 5));
			}
		}
	return 0;
	}
