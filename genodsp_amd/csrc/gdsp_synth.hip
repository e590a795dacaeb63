// gdsp_synth.hip -- synthetic coverage-like signal, generated in HBM.
//
// Not part of the reference: it stands in for the text ingest when benchmarking
// (BASELINE.json: "24-chrom 3.1 Gbp synthetic signal").  The generator is
// counter-based -- value(position) depends only on (seed, chromosome index,
// position) -- so the CPU checker regenerates any sub-range of any chromosome
// (oracle/gdsp_oracle.c:orc_synth_coverage holds the same integer recipe).
//
// Signal: the chromosome is cut into 128-base cells; cell c has one breakpoint
// b(c) in [0,128); positions left of it continue cell c-1's depth, the others
// take cell c's depth.  Depths are 0 (35 %) or a bell-shaped 1..61, so runs
// average ~128 bases like read-depth tracks.  Mode 1 multiplies each base by
// its own factor in [0.5,1.5) to give a real-valued signal.

#include "gdsp_common.h"

__device__ __forceinline__ uint64_t synth_mix64 (uint64_t x)
	{
	x += 0x9E3779B97F4A7C15ULL;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
	return x ^ (x >> 31);
	}

__device__ __forceinline__ uint32_t synth_cell_depth (uint64_t key, int64_t cell)
	{
	uint64_t h = synth_mix64 (key ^ ((uint64_t) cell * 0xD1342543DE82EF95ULL));
	if ((h >> 8) % 100 < 35) return 0;
	return 1 + (uint32_t) ((h >> 16) & 15) + (uint32_t) ((h >> 20) & 15)
	         + (uint32_t) ((h >> 24) & 15) + (uint32_t) ((h >> 28) & 15);
	}

__global__ __launch_bounds__(256)
void synth_coverage_kernel (double* __restrict__ out, uint64_t key, uint32_t start, uint32_t count, int mode)
	{
	const size_t stride = (size_t) gridDim.x * 256;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x ; i < count ; i += stride)
		{
		uint64_t pos  = (uint64_t) start + i;
		int64_t  cell = (int64_t) (pos >> 7);
		uint32_t off  = (uint32_t) (pos & 127);
		uint64_t hb   = synth_mix64 (key ^ 0xA5A5A5A5ULL ^ ((uint64_t) cell * 0x9E3779B97F4A7C15ULL));
		uint32_t brk  = (uint32_t) (hb & 127);
		uint32_t d    = (off >= brk)? synth_cell_depth (key, cell) : synth_cell_depth (key, cell-1);
		double   x    = (double) d;
		if (mode == 1)
			{
			uint64_t hp = synth_mix64 (key ^ 0x5bd1e995ULL ^ (pos * 0xC2B2AE3D27D4EB4FULL));
			// (hp>>11)*2^-53 is exact, so the sum rounds once; then one rounded product
			double   u  = 0.5 + (double) (hp >> 11) * (1.0 / 9007199254740992.0);
			x = x * u;
			}
		out[i] = x;
		}
	}

static uint64_t synth_host_mix64 (uint64_t x)
	{
	x += 0x9E3779B97F4A7C15ULL;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
	return x ^ (x >> 31);
	}

extern "C" int gdsp_synth_coverage (double* d_out, uint64_t seed, uint32_t chromIndex, uint32_t start,
                                    uint32_t count, int mode, void* stream)
	{
	if (count == 0) return GDSP_OK;
	GDSP_REQUIRE (d_out != NULL, "NULL vector");
	GDSP_REQUIRE ((mode == 0) || (mode == 1), "mode must be 0 or 1");
	uint64_t key    = synth_host_mix64 (seed ^ ((uint64_t) (chromIndex+1) << 40));
	size_t   want   = ((size_t) count + 1023) / 1024;
	uint32_t blocks = (uint32_t) (want > 4096? 4096 : want);
	hipLaunchKernelGGL (synth_coverage_kernel, dim3(blocks), dim3(256), 0, gdsp_stream (stream),
	                    d_out, key, start, count, mode);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}
