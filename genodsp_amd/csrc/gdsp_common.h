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

static inline hipStream_t gdsp_stream (void* s) { return (hipStream_t) s; }

static inline bool gdsp_aligned16 (const void* p) { return (((uintptr_t) p) & 15) == 0; }

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
