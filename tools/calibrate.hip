// calibrate.hip -- what this MI355X actually sustains, for pricing the kernels:
//   * FP64 vector rate (v_fma_f64, and v_mul_f64+v_add_f64) at 1..8 waves per SIMD,
//     with the in-kernel shader clock (s_memtime / s_memrealtime) under that load;
//   * HBM streaming bandwidth of a 16-byte-per-lane copy.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/calibrate.hip -o build/calibrate
// Not part of the product; numbers go into DESIGN.md.

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf (stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString (e_)); exit (1); } } while (0)

#define ACCS 12

template <bool FMA>
__global__ __launch_bounds__(256)
void fp64_rate_kernel (double* out, double a, double b, int iters, unsigned long long* clocks)
	{
	double acc[ACCS];
#pragma unroll
	for (int i=0 ; i<ACCS ; i++) acc[i] = threadIdx.x * 1e-9 + i;
	unsigned long long c0 = __builtin_amdgcn_s_memtime (), r0 = __builtin_amdgcn_s_memrealtime ();
	for (int it=0 ; it<iters ; it++)
		{
#pragma unroll
		for (int u=0 ; u<8 ; u++)
#pragma unroll
			for (int i=0 ; i<ACCS ; i++)
				{
				if (FMA) acc[i] = __builtin_fma (a, acc[i], b);
				else     acc[i] = acc[i] * a + b;
				}
		}
	unsigned long long c1 = __builtin_amdgcn_s_memtime (), r1 = __builtin_amdgcn_s_memrealtime ();
	double s = 0;
#pragma unroll
	for (int i=0 ; i<ACCS ; i++) s += acc[i];
	out[(size_t) blockIdx.x * 256 + threadIdx.x] = s;
	if (threadIdx.x == 0) { clocks[2*blockIdx.x] = c1 - c0;  clocks[2*blockIdx.x+1] = r1 - r0; }
	}

__global__ __launch_bounds__(256)
void copy_kernel (const double2* __restrict__ in, double2* __restrict__ out, size_t n)
	{
	size_t stride = (size_t) gridDim.x * 256;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x ; i < n ; i += stride) out[i] = in[i];
	}

// copy variants: UNROLL loads in flight per lane, grid-stride or one contiguous chunk per block,
// plain or non-temporal accesses
template <int UNROLL, bool CHUNKED, bool NT>
__global__ __launch_bounds__(256)
void copy_var_kernel (const double2* __restrict__ in, double2* __restrict__ out, size_t n)
	{
	size_t stride, i, end;
	if (CHUNKED)
		{
		size_t per = (n + gridDim.x - 1) / gridDim.x;
		per = (per + 256*UNROLL - 1) / (256*UNROLL) * (256*UNROLL);
		i = (size_t) blockIdx.x * per + threadIdx.x;
		end = (i - threadIdx.x + per < n)? i - threadIdx.x + per : n;
		stride = 256;
		}
	else { stride = (size_t) gridDim.x * 256;  i = (size_t) blockIdx.x * 256 + threadIdx.x;  end = n; }
	for ( ; i + (UNROLL-1)*stride < end ; i += UNROLL*stride)
		{
		double2 d[UNROLL];
#pragma unroll
		for (int u=0 ; u<UNROLL ; u++)
			{
			if (NT) { d[u].x = __builtin_nontemporal_load (&in[i+u*stride].x);  d[u].y = __builtin_nontemporal_load (&in[i+u*stride].y); }
			else    d[u] = in[i+u*stride];
			}
#pragma unroll
		for (int u=0 ; u<UNROLL ; u++)
			{
			if (NT) { __builtin_nontemporal_store (d[u].x, &out[i+u*stride].x);  __builtin_nontemporal_store (d[u].y, &out[i+u*stride].y); }
			else    out[i+u*stride] = d[u];
			}
		}
	for ( ; i < end ; i += stride) out[i] = in[i];
	}

template <int UNROLL, bool CHUNKED, bool NT>
static void run_copy_var (const double2* a, double2* b, size_t n, int cus, hipEvent_t e0, hipEvent_t e1)
	{
	for (int blocksPer : { 2, 4, 8, 16 })
		{
		float best = 1e30f;
		for (int rep=0 ; rep<4 ; rep++)
			{
			CHECK (hipEventRecord (e0));
			hipLaunchKernelGGL ((copy_var_kernel<UNROLL, CHUNKED, NT>), dim3(cus * blocksPer), dim3(256), 0, 0, a, b, n);
			CHECK (hipEventRecord (e1));
			CHECK (hipEventSynchronize (e1));
			float ms;  CHECK (hipEventElapsedTime (&ms, e0, e1));
			if (rep >= 1) best = std::min (best, ms);
			}
		printf ("copy unroll %d %s %s, %2d blocks/CU: %.3f ms  %.1f GB/s\n", UNROLL, CHUNKED? "chunked    " : "grid-stride",
		        NT? "nt   " : "plain", blocksPer, best, 2.0 * n * sizeof(double2) / best / 1e6);
		}
	}

int main ()
	{
	hipDeviceProp_t prop;
	CHECK (hipGetDeviceProperties (&prop, 0));
	printf ("device: %s, CUs %d, clock %d kHz, mem clock %d kHz\n", prop.name, prop.multiProcessorCount,
	        prop.clockRate, prop.memoryClockRate);
	const int cus = prop.multiProcessorCount;
	hipEvent_t e0, e1;
	CHECK (hipEventCreate (&e0));  CHECK (hipEventCreate (&e1));

	double* out;  unsigned long long* clocks;
	CHECK (hipMalloc (&out, (size_t) cus * 8 * 256 * sizeof(double)));
	CHECK (hipMalloc (&clocks, (size_t) cus * 8 * 2 * sizeof(unsigned long long)));
	std::vector<unsigned long long> h (cus * 8 * 2);

	for (int fma=1 ; fma>=0 ; fma--)
		for (int perCU=1 ; perCU<=8 ; perCU*=2)
			{
			const int blocks = cus * perCU, iters = 20000;
			float best = 1e30f;
			for (int rep=0 ; rep<6 ; rep++)      // the first repetitions also warm the clocks up
				{
				CHECK (hipEventRecord (e0));
				if (fma) hipLaunchKernelGGL (fp64_rate_kernel<true>,  dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters, clocks);
				else     hipLaunchKernelGGL (fp64_rate_kernel<false>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters, clocks);
				CHECK (hipEventRecord (e1));
				CHECK (hipEventSynchronize (e1));
				float ms;  CHECK (hipEventElapsedTime (&ms, e0, e1));
				if (rep >= 3) best = std::min (best, ms);
				}
			CHECK (hipMemcpy (h.data (), clocks, (size_t) blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
			std::vector<double> ghz;
			for (int b=0 ; b<blocks ; b++) ghz.push_back ((double) h[2*b] / (double) h[2*b+1] * 0.1);
			std::sort (ghz.begin (), ghz.end ());
			double instr = (double) blocks * 256 * iters * 8 * ACCS * (fma? 1 : 2);
			double flops = (double) blocks * 256 * iters * 8 * ACCS * 2;
			printf ("%s  %d waves/SIMD: %8.3f ms  %6.2f TFLOP/s  %6.2f T lane-instr/s  clock median %.3f GHz  -> %.2f cycles per wave-instr per SIMD\n",
			        fma? "v_fma_f64    " : "v_mul+v_add  ", perCU, best, flops / best / 1e9, instr / best / 1e9,
			        ghz[ghz.size()/2],
			        ghz[ghz.size()/2] * 1e9 * (best * 1e-3) / ((double) iters * 8 * ACCS * (fma? 1 : 2) * perCU));
			}

	// HBM copy, 4 GiB each way
	size_t n = (size_t) 1 << 28;                       // double2 elements = 4 GiB
	double2 *a, *b;
	CHECK (hipMalloc (&a, n * sizeof(double2)));  CHECK (hipMalloc (&b, n * sizeof(double2)));
	CHECK (hipMemset (a, 1, n * sizeof(double2)));
	for (int blocksPer : { 4, 8, 16, 32 })
		{
		float best = 1e30f;
		for (int rep=0 ; rep<4 ; rep++)
			{
			CHECK (hipEventRecord (e0));
			hipLaunchKernelGGL (copy_kernel, dim3(cus * blocksPer), dim3(256), 0, 0, a, b, n);
			CHECK (hipEventRecord (e1));
			CHECK (hipEventSynchronize (e1));
			float ms;  CHECK (hipEventElapsedTime (&ms, e0, e1));
			if (rep >= 1) best = std::min (best, ms);
			}
		printf ("copy 16 B/lane, %2d blocks/CU: %.3f ms  %.1f GB/s (read+write)\n", blocksPer, best,
		        2.0 * n * sizeof(double2) / best / 1e6);
		}
	run_copy_var<4, false, false> (a, b, n, cus, e0, e1);
	run_copy_var<4, false, true>  (a, b, n, cus, e0, e1);
	run_copy_var<4, true,  false> (a, b, n, cus, e0, e1);
	run_copy_var<4, true,  true>  (a, b, n, cus, e0, e1);
	run_copy_var<8, false, false> (a, b, n, cus, e0, e1);
	run_copy_var<2, false, false> (a, b, n, cus, e0, e1);
	// read-only and write-only rates
	return 0;
	}
