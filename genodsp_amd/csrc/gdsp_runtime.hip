// gdsp_runtime.hip -- device memory, streams, events, errors.
//
// What the reference does with calloc/free for chromosome and scratch vectors
// (genodsp.c:865-878, :1890-2037) the driver does here with HBM allocations.

#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include "gdsp_common.h"

static thread_local char gdsp_error_text[512] = "";

void gdsp_set_error (const char* fmt, ...)
	{
	va_list ap;
	va_start (ap, fmt);
	vsnprintf (gdsp_error_text, sizeof(gdsp_error_text), fmt, ap);
	va_end (ap);
	}

extern "C" {

const char* gdsp_last_error (void) { return gdsp_error_text; }
// "<library version> (gfx950) <git hash of the tree the library was built from>[-dirty]"; the hash comes from
// the Makefile (GDSP_BUILD_ID), "unknown" where the sources were not under git when it was built
#ifndef GDSP_BUILD_ID
#define GDSP_BUILD_ID "unknown"
#endif
const char* gdsp_version    (void) { return "genodsp_hip 0.2 (gfx950) " GDSP_BUILD_ID; }

int gdsp_device_count (int* count)
	{
	GDSP_REQUIRE (count != NULL, "count is NULL");
	GDSP_HIP_TRY (hipGetDeviceCount (count));
	return GDSP_OK;
	}

int gdsp_set_device (int device)
	{
	GDSP_HIP_TRY (hipSetDevice (device));
	return GDSP_OK;
	}

int gdsp_get_device (int* device)
	{
	GDSP_REQUIRE (device != NULL, "NULL pointer");
	GDSP_HIP_TRY (hipGetDevice (device));
	return GDSP_OK;
	}

// GDSP_POISON=<double|nan>: a debugging aid.  Every device allocation is filled with that value before it is handed
// out (a fresh box hands out zeros, which hides a kernel that reads memory nobody wrote: with a poison that wins every
// comparison -- 1e300 for the maxima, -1e300 for the minima, nan for arithmetic -- such a read changes the output).
// The driver poisons a vector's partner after every flip as well (genodsp_hip.c: flip_vector).
int gdsp_poison (double* value)
	{
	static int    known = -1;
	static double pattern = 0;
	if (known < 0)
		{
		const char* e = getenv ("GDSP_POISON");
		known = ((e != NULL) && (e[0] != 0))? 1 : 0;
		if (known) pattern = (strcmp (e, "nan") == 0)? __builtin_nan ("") : strtod (e, NULL);
		}
	if (value != NULL) *value = pattern;
	return known;
	}
} // extern "C"

__global__ void poison_kernel (uint64_t* __restrict__ p, size_t words, uint64_t bits)
	{
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x ; i < words ; i += (size_t) gridDim.x * blockDim.x) p[i] = bits;
	}

extern "C" {

int gdsp_malloc (void** d_ptr, size_t bytes)
	{
	GDSP_REQUIRE (d_ptr != NULL, "d_ptr is NULL");
	hipError_t e = hipMalloc (d_ptr, bytes? bytes : 16);
	if (e == hipErrorOutOfMemory) { gdsp_set_error ("hipMalloc(%zu) out of memory", bytes);  return GDSP_ENOMEM; }
	GDSP_HIP_TRY (e);
	double pattern;
	if (gdsp_poison (&pattern) && (bytes >= 8))
		{
		uint64_t bits;  memcpy (&bits, &pattern, 8);
		hipLaunchKernelGGL (poison_kernel, dim3(1024), dim3(256), 0, 0, (uint64_t*) *d_ptr, bytes / 8, bits);
		GDSP_HIP_TRY (hipDeviceSynchronize ());
		}
	return GDSP_OK;
	}

int gdsp_free (void* d_ptr)
	{
	GDSP_HIP_TRY (hipFree (d_ptr));
	return GDSP_OK;
	}

int gdsp_host_alloc (void** h_ptr, size_t bytes)
	{
	GDSP_REQUIRE (h_ptr != NULL, "h_ptr is NULL");
	GDSP_HIP_TRY (hipHostMalloc (h_ptr, bytes? bytes : 16, hipHostMallocDefault));
	return GDSP_OK;
	}

int gdsp_host_free (void* h_ptr)
	{
	GDSP_HIP_TRY (hipHostFree (h_ptr));
	return GDSP_OK;
	}

int gdsp_memcpy_h2d (void* d_dst, const void* h_src, size_t bytes, void* stream)
	{
	GDSP_HIP_TRY (hipMemcpyAsync (d_dst, h_src, bytes, hipMemcpyHostToDevice, gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_memcpy_d2h (void* h_dst, const void* d_src, size_t bytes, void* stream)
	{
	GDSP_HIP_TRY (hipMemcpyAsync (h_dst, d_src, bytes, hipMemcpyDeviceToHost, gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_memcpy_d2d (void* d_dst, const void* d_src, size_t bytes, void* stream)
	{
	GDSP_HIP_TRY (hipMemcpyAsync (d_dst, d_src, bytes, hipMemcpyDeviceToDevice, gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_memcpy_peer (void* d_dst, int dstDevice, const void* d_src, int srcDevice, size_t bytes, void* stream)
	{
	if (dstDevice == srcDevice) GDSP_HIP_TRY (hipMemcpyAsync (d_dst, d_src, bytes, hipMemcpyDeviceToDevice, gdsp_stream (stream)));
	else                        GDSP_HIP_TRY (hipMemcpyPeerAsync (d_dst, dstDevice, d_src, srcDevice, bytes, gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_memset (void* d_dst, int byte, size_t bytes, void* stream)
	{
	GDSP_HIP_TRY (hipMemsetAsync (d_dst, byte, bytes, gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_stream_create (void** stream)
	{
	GDSP_REQUIRE (stream != NULL, "stream is NULL");
	hipStream_t s;
	GDSP_HIP_TRY (hipStreamCreateWithFlags (&s, hipStreamNonBlocking));
	*stream = (void*) s;
	return GDSP_OK;
	}

int gdsp_stream_destroy (void* stream)
	{
	GDSP_HIP_TRY (hipStreamDestroy (gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_stream_sync (void* stream)
	{
	GDSP_HIP_TRY (hipStreamSynchronize (gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_device_sync (void)
	{
	GDSP_HIP_TRY (hipDeviceSynchronize ());
	return GDSP_OK;
	}

int gdsp_event_create (void** event)
	{
	GDSP_REQUIRE (event != NULL, "event is NULL");
	hipEvent_t e;
	GDSP_HIP_TRY (hipEventCreate (&e));
	*event = (void*) e;
	return GDSP_OK;
	}

int gdsp_event_destroy (void* event)
	{
	GDSP_HIP_TRY (hipEventDestroy ((hipEvent_t) event));
	return GDSP_OK;
	}

int gdsp_event_record (void* event, void* stream)
	{
	GDSP_HIP_TRY (hipEventRecord ((hipEvent_t) event, gdsp_stream (stream)));
	return GDSP_OK;
	}

int gdsp_stream_wait_event (void* stream, void* event)
	{
	GDSP_HIP_TRY (hipStreamWaitEvent (gdsp_stream (stream), (hipEvent_t) event, 0));
	return GDSP_OK;
	}

int gdsp_event_elapsed_ms (void* start, void* stop, float* ms)
	{
	GDSP_REQUIRE (ms != NULL, "ms is NULL");
	GDSP_HIP_TRY (hipEventSynchronize ((hipEvent_t) stop));
	GDSP_HIP_TRY (hipEventElapsedTime (ms, (hipEvent_t) start, (hipEvent_t) stop));
	return GDSP_OK;
	}

} // extern "C"
