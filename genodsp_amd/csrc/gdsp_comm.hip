// gdsp_comm.hip -- the path's only collective: all-reduce of a few KiB of counters across the GPUs
// of one node, RCCL over xGMI.
//
// The reference is one thread on one host; what it does with a whole-genome view -- percentile's
// sort of the sampled genome (percentile.c:547-683), invert's global min/max (add.c:909-923) --
// becomes, with chromosomes dealt over GPUs, a per-device count/histogram/extreme that must be
// combined: u64 sum / min / max, f64 min / max.  Messages are <= 64 KiB, so they are latency
// bound on the fully connected xGMI mesh; stock RCCL, one communicator per device of this
// process (ncclCommInitAll), every all-reduce issued for all devices inside one group call on
// each device's own stream, in place.
//
// RCCL is opened at run time (dlopen of librccl.so.1) the first time a communicator is asked
// for: a one-GPU run never loads it, and the library has no link-time dependency on it.  There
// is no fallback inside this file: if RCCL cannot be loaded gdsp_comm_create fails and says so.

#include <dlfcn.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>
#include "gdsp_common.h"

struct gdsp_comm
	{
	int         ndev;
	int*        devices;      // HIP device of rank r
	ncclComm_t* comms;        // communicator of rank r
	};

namespace {

struct RcclApi
	{
	void*        handle;
	ncclResult_t (*CommInitAll)    (ncclComm_t*, int, const int*);
	ncclResult_t (*CommDestroy)    (ncclComm_t);
	ncclResult_t (*AllReduce)      (const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
	ncclResult_t (*GroupStart)     (void);
	ncclResult_t (*GroupEnd)       (void);
	const char*  (*GetErrorString) (ncclResult_t);
	ncclResult_t (*GetVersion)     (int*);
	};

RcclApi    rccl = {};
std::mutex rcclLock;

int rccl_load (void)
	{
	std::lock_guard<std::mutex> hold (rcclLock);
	if (rccl.handle != NULL) return GDSP_OK;
	// RCCL writes its version banner and any NCCL_DEBUG output to stdout, which for the driver is the data
	// channel (interval text): send it to stderr unless the user has chosen a file
	setenv ("NCCL_DEBUG_FILE", "/dev/stderr", 0);
	const char* names[] = { getenv ("GDSP_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	void* h = NULL;
	for (const char* name : names)
		{
		if ((name == NULL) || (name[0] == 0)) continue;
		h = dlopen (name, RTLD_NOW | RTLD_LOCAL);
		if (h != NULL) break;
		}
	if (h == NULL) { gdsp_set_error ("RCCL cannot be loaded (librccl.so.1): %s", dlerror ());  return GDSP_EHIP; }
#define RCCL_SYM(field, name)                                                                          \
	do { *(void**) &rccl.field = dlsym (h, name);                                                      \
	     if (rccl.field == NULL) { gdsp_set_error ("RCCL lacks %s", name);  dlclose (h);  return GDSP_EHIP; } } while (0)
	RCCL_SYM (CommInitAll,    "ncclCommInitAll");
	RCCL_SYM (CommDestroy,    "ncclCommDestroy");
	RCCL_SYM (AllReduce,      "ncclAllReduce");
	RCCL_SYM (GroupStart,     "ncclGroupStart");
	RCCL_SYM (GroupEnd,       "ncclGroupEnd");
	RCCL_SYM (GetErrorString, "ncclGetErrorString");
	RCCL_SYM (GetVersion,     "ncclGetVersion");
#undef RCCL_SYM
	rccl.handle = h;
	return GDSP_OK;
	}

#define RCCL_TRY(call)                                                                                 \
	do { ncclResult_t r_ = (call);                                                                     \
	     if (r_ != ncclSuccess) { gdsp_set_error ("%s:%d: %s -> %s", __FILE__, __LINE__, #call, rccl.GetErrorString (r_));  \
	                              return GDSP_EHIP; } } while (0)

int comm_allreduce (gdsp_comm* comm, void* const* d_bufs, size_t count, ncclDataType_t type, int op, void* const* streams)
	{
	GDSP_REQUIRE ((comm != NULL) && (d_bufs != NULL), "NULL communicator or buffers");
	GDSP_REQUIRE ((op >= 0) && (op <= 2), "op must be 0 (sum), 1 (min) or 2 (max)");
	if (count == 0) return GDSP_OK;
	const ncclRedOp_t how = (op == 0)? ncclSum : (op == 1)? ncclMin : ncclMax;
	int home = 0;
	GDSP_HIP_TRY (hipGetDevice (&home));
	RCCL_TRY (rccl.GroupStart ());
	for (int r=0 ; r<comm->ndev ; r++)
		{
		if (d_bufs[r] == NULL) { (void) rccl.GroupEnd ();  gdsp_set_error ("gdsp_comm_allreduce: rank %d has no buffer", r);  return GDSP_EINVAL; }
		hipStream_t s = gdsp_stream ((streams != NULL)? streams[r] : NULL);
		ncclResult_t e = rccl.AllReduce (d_bufs[r], d_bufs[r], count, type, how, comm->comms[r], s);
		if (e != ncclSuccess)
			{ (void) rccl.GroupEnd ();  gdsp_set_error ("ncclAllReduce (rank %d) -> %s", r, rccl.GetErrorString (e));  return GDSP_EHIP; }
		}
	RCCL_TRY (rccl.GroupEnd ());
	GDSP_HIP_TRY (hipSetDevice (home));
	return GDSP_OK;
	}

} // namespace

extern "C" {

int gdsp_comm_create (gdsp_comm** out, const int* devices, int ndevices)
	{
	GDSP_REQUIRE ((out != NULL) && (devices != NULL) && (ndevices >= 1) && (ndevices <= 64), "bad arguments");
	for (int i=0 ; i<ndevices ; i++)
		for (int j=0 ; j<i ; j++) GDSP_REQUIRE (devices[i] != devices[j], "a device appears twice (RCCL wants one rank per GPU)");
	int rc = rccl_load ();
	if (rc != GDSP_OK) return rc;
	int home = 0;
	GDSP_HIP_TRY (hipGetDevice (&home));
	gdsp_comm* c = (gdsp_comm*) calloc (1, sizeof(gdsp_comm));
	if (c != NULL) { c->devices = (int*) calloc (ndevices, sizeof(int));  c->comms = (ncclComm_t*) calloc (ndevices, sizeof(ncclComm_t)); }
	if ((c == NULL) || (c->devices == NULL) || (c->comms == NULL))
		{
		if (c != NULL) { free (c->devices);  free (c->comms);  free (c); }
		gdsp_set_error ("out of host memory");
		return GDSP_ENOMEM;
		}
	c->ndev = ndevices;
	memcpy (c->devices, devices, ndevices * sizeof(int));
	// With NCCL_DEBUG set (VERSION on this pool's machines) RCCL prints a banner with printf while it
	// initialises.  stdout is the driver's data channel (interval text), so for the length of the call file
	// descriptor 1 points where 2 does; the stdio buffer is flushed on both sides of the switch.
	fflush (stdout);
	const int keep = dup (STDOUT_FILENO);
	if (keep >= 0) (void) dup2 (STDERR_FILENO, STDOUT_FILENO);
	ncclResult_t e = rccl.CommInitAll (c->comms, ndevices, c->devices);
	fflush (stdout);
	if (keep >= 0) { (void) dup2 (keep, STDOUT_FILENO);  close (keep); }
	(void) hipSetDevice (home);
	if (e != ncclSuccess)
		{
		gdsp_set_error ("ncclCommInitAll over %d device(s) -> %s", ndevices, rccl.GetErrorString (e));
		free (c->devices);  free (c->comms);  free (c);
		return GDSP_EHIP;
		}
	*out = c;
	return GDSP_OK;
	}

int gdsp_comm_destroy (gdsp_comm* comm)
	{
	if (comm == NULL) return GDSP_OK;
	for (int r=0 ; r<comm->ndev ; r++) { if (comm->comms[r] != NULL) (void) rccl.CommDestroy (comm->comms[r]); }
	free (comm->devices);  free (comm->comms);  free (comm);
	return GDSP_OK;
	}

int gdsp_comm_size (const gdsp_comm* comm) { return (comm == NULL)? 0 : comm->ndev; }

int gdsp_comm_device (const gdsp_comm* comm, int rank)
	{ return ((comm == NULL) || (rank < 0) || (rank >= comm->ndev))? -1 : comm->devices[rank]; }

int gdsp_comm_rccl_version (int* version)
	{
	GDSP_REQUIRE (version != NULL, "NULL pointer");
	int rc = rccl_load ();
	if (rc != GDSP_OK) return rc;
	RCCL_TRY (rccl.GetVersion (version));
	return GDSP_OK;
	}

int gdsp_comm_allreduce_u64 (gdsp_comm* comm, uint64_t* const* d_bufs, size_t count, int op, void* const* streams)
	{ return comm_allreduce (comm, (void* const*) d_bufs, count, ncclUint64, op, streams); }

int gdsp_comm_allreduce_f64 (gdsp_comm* comm, double* const* d_bufs, size_t count, int op, void* const* streams)
	{ return comm_allreduce (comm, (void* const*) d_bufs, count, ncclFloat64, op, streams); }

} // extern "C"
