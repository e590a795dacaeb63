// gdsp_intervals.hip -- interval-driven writes into a chromosome vector in HBM.
//
// Reference: read_intervals genodsp.c:1187-1350 (main ingest and the `input`
// operator, opio.c:225-256), op_add_apply / op_subtract_apply add.c:191-306,
// :484-599, op_multiply_apply / op_divide_apply multiply.c:193-393, :586-787.
// The reference walks the interval file and loops `for ix in [start,end)` per
// interval, so a base covered by several intervals receives them in FILE ORDER.
//
// Device formulation.  The host (which parses the text anyway) routes intervals
// to chromosomes, origin-shifts and clips them, and bins their indices into
// fixed tiles of IV_TILE bases, keeping file order inside every tile (a CSR:
// d_tileOffsets[ntiles+1], d_tileList[...]).  One workgroup owns one tile: its
// 1024 bases sit in registers (four per lane), the tile's interval list is
// walked once in file order, every lane applies the intervals that cover its
// bases, and the tile is written back with one coalesced store per base.
// Because each base sees its intervals in file order, sums are bit-identical to
// the reference for any values (not just integer depth), and so are the
// min/max/first-touch ("clear") rules.  Tiles without intervals are skipped,
// or just filled when the vector is being cleared.  Traffic: 16 B/base for
// touched tiles (8 B when clearing) plus 16 B per (interval, tile) pair.
// `clear` carries two bits so that a long interval stream can be applied in
// batches: GDSP_CLEAR_FILL starts every base from the missing value (first
// batch), GDSP_CLEAR_FIRST_TOUCH keeps the reference's "a base still holding the
// missing value is assigned, not accumulated" rule (every batch).

#include "gdsp_common.h"

#define IV_THREADS 256
#define IV_PER     4
#define IV_TILE    (IV_THREADS * IV_PER)      // 1024 bases; the host bins with the same number

enum { IV_SUM = 0, IV_MIN = 1, IV_MAX = 2, IV_MUL = 3, IV_DIV = 4, IV_SET = 5, IV_KEEP = 6 };

template <int OP>
__global__ __launch_bounds__(IV_THREADS)
void intervals_kernel (double* __restrict__ v, uint32_t n,
                       const uint32_t* __restrict__ start, const uint32_t* __restrict__ end,
                       const double* __restrict__ val,
                       const uint32_t* __restrict__ tileOffsets, const uint32_t* __restrict__ tileList,
                       int clear, double missingVal, double infinityVal)
	{
	const uint32_t tile  = blockIdx.x;
	const uint32_t lo    = tileOffsets[tile], hi = tileOffsets[tile+1];
	const bool     scale = (OP == IV_MUL) || (OP == IV_DIV) || (OP == IV_KEEP);   // ops with a rule for uncovered bases
	const bool     nonzeroToOne = (OP == IV_SET || OP == IV_KEEP) && (clear & GDSP_MASK_BINARIZE_FIRST);
	const bool     touch = (clear & GDSP_CLEAR_FIRST_TOUCH) != 0;   // "still missing -> assign"
	const bool     fill  = (clear & GDSP_CLEAR_FILL) != 0;          // start from the missing value
	if ((lo == hi) && !fill && !scale && !nonzeroToOne) return;   // untouched tile

	const uint64_t base = (uint64_t) tile * IV_TILE;
	uint32_t pos[IV_PER];
	double   x[IV_PER];
	bool     covered[IV_PER];
#pragma unroll
	for (int k=0 ; k<IV_PER ; k++)
		{
		pos[k]     = (uint32_t) (base + (uint64_t) k*IV_THREADS + threadIdx.x);   // coalesced per k
		covered[k] = false;
		x[k]       = fill? missingVal : ((pos[k] < n)? v[pos[k]] : 0.0);
		if (nonzeroToOne && (x[k] != 0.0)) x[k] = 1.0;                  // logical.c:471-472, :766-767
		}

	for (uint32_t j=lo ; j<hi ; j++)
		{
		const uint32_t i = tileList[j];                   // wave-uniform: scalar loads
		const uint32_t s = start[i], e = end[i];
		const double   a = val[i];
#pragma unroll
		for (int k=0 ; k<IV_PER ; k++)
			{
			if ((pos[k] >= s) && (pos[k] < e))
				{
				if (OP == IV_KEEP) covered[k] = true;                          // mask.c:593-596, logical.c:862-865
				else if (OP == IV_SET) x[k] = a;                               // mask.c:283-284, logical.c:535-536
				else if (scale)                                                // multiply.c:340-341, :735-736
					{ x[k] = (OP == IV_MUL)? x[k] * a : x[k] / a;  covered[k] = true; }
				else if (touch && (x[k] == missingVal)) x[k] = a;             // genodsp.c:1311,1319,1327
				else if (OP == IV_SUM) x[k] = x[k] + a;                        // genodsp.c:1328
				else if (OP == IV_MIN) { if (a < x[k]) x[k] = a; }             // genodsp.c:1312
				else                   { if (a > x[k]) x[k] = a; }             // genodsp.c:1320
				}
			}
		}

#pragma unroll
	for (int k=0 ; k<IV_PER ; k++)
		{
		if (pos[k] >= n) continue;
		double r = x[k];
		if (scale && !covered[k])                         // multiply.c:330-331, divide :711, masknot/and: the fill value
			r = (OP == IV_MUL)? 0.0 : ((OP == IV_KEEP)? infinityVal : ((x[k] >= 0)? infinityVal : -infinityVal));
		v[pos[k]] = r;
		}
	}

template <int OP>
static int intervals_launch (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end,
                             const double* d_val, const uint32_t* d_tileOffsets, const uint32_t* d_tileList,
                             int clear, double missingVal, double infinityVal, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE (d_v != NULL, "NULL vector");
	GDSP_REQUIRE (d_tileOffsets != NULL, "NULL tile offsets");
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + IV_TILE - 1) / IV_TILE);
	hipLaunchKernelGGL ((intervals_kernel<OP>), dim3(ntiles), dim3(IV_THREADS), 0, gdsp_stream (stream),
	                    d_v, n, d_start, d_end, d_val, d_tileOffsets, d_tileList, clear, missingVal, infinityVal);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

extern "C" {

uint32_t gdsp_interval_tile (void) { return IV_TILE; }

int gdsp_apply_intervals (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end,
                          const double* d_val, const uint32_t* d_tileOffsets, const uint32_t* d_tileList,
                          int overlapOp, int clear, double missingVal, void* stream)
	{
	switch (overlapOp)
		{
		case GDSP_OVERLAP_SUM: return intervals_launch<IV_SUM> (d_v, n, d_start, d_end, d_val, d_tileOffsets, d_tileList, clear, missingVal, 0.0, stream);
		case GDSP_OVERLAP_MIN: return intervals_launch<IV_MIN> (d_v, n, d_start, d_end, d_val, d_tileOffsets, d_tileList, clear, missingVal, 0.0, stream);
		case GDSP_OVERLAP_MAX: return intervals_launch<IV_MAX> (d_v, n, d_start, d_end, d_val, d_tileOffsets, d_tileList, clear, missingVal, 0.0, stream);
		}
	gdsp_set_error ("gdsp_apply_intervals: unknown overlap operator %d", overlapOp);
	return GDSP_EINVAL;
	}

int gdsp_scale_intervals (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end,
                          const double* d_val, const uint32_t* d_tileOffsets, const uint32_t* d_tileList,
                          int divide, double infinityVal, void* stream)
	{
	if (divide) return intervals_launch<IV_DIV> (d_v, n, d_start, d_end, d_val, d_tileOffsets, d_tileList, 0, 0.0, infinityVal, stream);
	return             intervals_launch<IV_MUL> (d_v, n, d_start, d_end, d_val, d_tileOffsets, d_tileList, 0, 0.0, infinityVal, stream);
	}

} // extern "C"

// ------------------------------------------------------- minover / maxover ----
// op_min_in_interval_apply minmax.c:193-390, op_max_in_interval_apply :596-793: inside every
// (sorted, non-overlapping) interval only the extreme value survives, at the position nearest
// the interval's centre among the tied ones (largest inset = min(ix-start, end-ix), then the
// lowest ix); everything else, inside or outside intervals, becomes the fill value.
// Three tile passes sharing the ingest kernel's tile lists:
//   phase 0: per interval, atomic min/max of the order-preserving key of the values;
//   phase 1: among bases equal to that extreme, atomic max of (inset << 32 | ~ix);
//   phase 2: rewrite the tile: the winner keeps its value, the rest is filled.
template <bool MAX, int PHASE>
__global__ __launch_bounds__(IV_THREADS)
void over_kernel (double* __restrict__ v, uint32_t n,
                  const uint32_t* __restrict__ start, const uint32_t* __restrict__ end,
                  const uint32_t* __restrict__ tileOffsets, const uint32_t* __restrict__ tileList,
                  unsigned long long* __restrict__ bestKey, unsigned long long* __restrict__ bestPos, double fill)
	{
	const uint32_t tile = blockIdx.x;
	const uint32_t lo   = tileOffsets[tile], hi = tileOffsets[tile+1];
	if ((PHASE < 2) && (lo == hi)) return;

	const uint64_t base = (uint64_t) tile * IV_TILE;
	uint32_t pos[IV_PER];
	double   x[IV_PER];
	bool     keep[IV_PER];
#pragma unroll
	for (int k=0 ; k<IV_PER ; k++)
		{
		pos[k]  = (uint32_t) (base + (uint64_t) k*IV_THREADS + threadIdx.x);
		x[k]    = (pos[k] < n)? v[pos[k]] : 0.0;
		keep[k] = false;
		}

	for (uint32_t j=lo ; j<hi ; j++)
		{
		const uint32_t i = tileList[j];
		const uint32_t s = start[i], e = (end[i] < n)? end[i] : n;
		if (PHASE == 0)
			{
			unsigned long long best = MAX? 0ULL : ~0ULL;
			bool any = false;
#pragma unroll
			for (int k=0 ; k<IV_PER ; k++)
				{
				if ((pos[k] >= s) && (pos[k] < e))
					{
					const unsigned long long key = gdsp_key_of (x[k]);
					if (MAX? (key > best) : (key < best)) best = key;
					any = true;
					}
				}
			for (int off=32 ; off>0 ; off>>=1)
				{
				const unsigned long long o = __shfl_down (best, off, 64);
				if (MAX? (o > best) : (o < best)) best = o;
				}
			if (__any (any) && ((threadIdx.x & 63) == 0))
				{ if (MAX) atomicMax (&bestKey[i], best);  else atomicMin (&bestKey[i], best); }
			}
		else if (PHASE == 1)
			{
			const unsigned long long target = bestKey[i];
			unsigned long long best = 0;
#pragma unroll
			for (int k=0 ; k<IV_PER ; k++)
				{
				if ((pos[k] >= s) && (pos[k] < e) && (gdsp_key_of (x[k]) == target))
					{
					const uint32_t a = pos[k] - s, b = e - pos[k];
					const unsigned long long packed = ((unsigned long long) ((a < b)? a : b) << 32) | (0xFFFFFFFFu - pos[k]);
					if (packed > best) best = packed;
					}
				}
			for (int off=32 ; off>0 ; off>>=1)
				{
				const unsigned long long o = __shfl_down (best, off, 64);
				if (o > best) best = o;
				}
			if ((best != 0) && ((threadIdx.x & 63) == 0)) atomicMax (&bestPos[i], best);
			}
		else
			{
			const uint32_t winner = 0xFFFFFFFFu - (uint32_t) (bestPos[i] & 0xFFFFFFFFu);
#pragma unroll
			for (int k=0 ; k<IV_PER ; k++) { if ((pos[k] >= s) && (pos[k] < e) && (pos[k] == winner)) keep[k] = true; }
			}
		}
	if (PHASE == 2)
		{
#pragma unroll
		for (int k=0 ; k<IV_PER ; k++) { if (pos[k] < n) v[pos[k]] = keep[k]? x[k] : fill; }
		}
	}

__global__ void over_init_kernel (unsigned long long* bestKey, unsigned long long* bestPos, uint32_t count, int wantMax)
	{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < count) { bestKey[i] = wantMax? 0ULL : ~0ULL;  bestPos[i] = 0ULL; }
	}

template <bool MAX>
static int over_launch (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end, uint32_t count,
                        const uint32_t* d_tileOffsets, const uint32_t* d_tileList, double fill,
                        unsigned long long* key, unsigned long long* pos, hipStream_t s)
	{
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + IV_TILE - 1) / IV_TILE);
	if (count != 0)
		{
		hipLaunchKernelGGL (over_init_kernel, dim3((count + 255) / 256), dim3(256), 0, s, key, pos, count, MAX? 1 : 0);
		hipLaunchKernelGGL ((over_kernel<MAX, 0>), dim3(ntiles), dim3(IV_THREADS), 0, s, d_v, n, d_start, d_end, d_tileOffsets, d_tileList, key, pos, fill);
		hipLaunchKernelGGL ((over_kernel<MAX, 1>), dim3(ntiles), dim3(IV_THREADS), 0, s, d_v, n, d_start, d_end, d_tileOffsets, d_tileList, key, pos, fill);
		}
	hipLaunchKernelGGL ((over_kernel<MAX, 2>), dim3(ntiles), dim3(IV_THREADS), 0, s, d_v, n, d_start, d_end, d_tileOffsets, d_tileList, key, pos, fill);
	GDSP_LAUNCH_CHECK ();
	return GDSP_OK;
	}

extern "C" {

size_t gdsp_extreme_in_intervals_work (uint32_t count) { return ((size_t) count + 1) * 2 * sizeof(uint64_t); }

/* minover (wantMax=0, fill = the infinity value) / maxover (wantMax=1, fill = the zero value) */
int gdsp_extreme_in_intervals (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end, uint32_t count,
                               const uint32_t* d_tileOffsets, const uint32_t* d_tileList,
                               int wantMax, double fill, void* d_work, void* stream)
	{
	if (n == 0) return GDSP_OK;
	GDSP_REQUIRE ((d_v != NULL) && (d_tileOffsets != NULL) && (d_work != NULL), "NULL pointer");
	unsigned long long* key = (unsigned long long*) d_work;
	unsigned long long* pos = key + count + 1;
	if (wantMax) return over_launch<true>  (d_v, n, d_start, d_end, count, d_tileOffsets, d_tileList, fill, key, pos, gdsp_stream (stream));
	return             over_launch<false> (d_v, n, d_start, d_end, count, d_tileOffsets, d_tileList, fill, key, pos, gdsp_stream (stream));
	}

/* mask.c:187-300 (mask), :483-640 (masknot), logical.c:439-560 (or), :737-880 (and):
 *   inside  = 1: bases under an interval become d_val[i] (mask: the mask value; or: 1.0)
 *   inside  = 0: bases under NO interval become outsideVal (masknot: the mask value; and: 0.0)
 *   binarizeFirst: every nonzero base becomes 1.0 before anything else (or, and) */
int gdsp_mask_intervals (double* d_v, uint32_t n, const uint32_t* d_start, const uint32_t* d_end,
                         const double* d_val, const uint32_t* d_tileOffsets, const uint32_t* d_tileList,
                         int inside, double outsideVal, int binarizeFirst, void* stream)
	{
	const int flags = binarizeFirst? GDSP_MASK_BINARIZE_FIRST : 0;
	if (inside) return intervals_launch<IV_SET>  (d_v, n, d_start, d_end, d_val, d_tileOffsets, d_tileList, flags, 0.0, 0.0, stream);
	return             intervals_launch<IV_KEEP> (d_v, n, d_start, d_end, d_val, d_tileOffsets, d_tileList, flags, 0.0, outsideVal, stream);
	}

// Host helper: bin `count` intervals [start,end) (already clipped to [0,n]) into
// tiles of gdsp_interval_tile() bases, file order kept inside each tile.
// h_tileOffsets has ntiles+1 entries; call once with h_tileList==NULL to learn
// the list length (returned in *listLen), then again with room for it.
int gdsp_bin_intervals (uint32_t n, const uint32_t* h_start, const uint32_t* h_end, uint32_t count,
                        uint32_t* h_tileOffsets, uint32_t* h_tileList, uint64_t* listLen)
	{
	GDSP_REQUIRE ((h_tileOffsets != NULL) && (listLen != NULL), "NULL pointer");
	GDSP_REQUIRE ((count == 0) || ((h_start != NULL) && (h_end != NULL)), "NULL interval arrays");
	const uint32_t ntiles = (uint32_t) (((uint64_t) n + IV_TILE - 1) / IV_TILE);
	for (uint32_t t=0 ; t<=ntiles ; t++) h_tileOffsets[t] = 0;
	uint64_t total = 0;
	for (uint32_t i=0 ; i<count ; i++)
		{
		uint32_t s = h_start[i], e = h_end[i];
		if (e > n) e = n;
		if (s >= e) continue;
		uint32_t t0 = s / IV_TILE, t1 = (e - 1) / IV_TILE;
		for (uint32_t t=t0 ; t<=t1 ; t++) h_tileOffsets[t+1]++;
		total += (uint64_t) t1 - t0 + 1;
		}
	*listLen = total;
	if (total > 0xFFFFFFFFULL) { gdsp_set_error ("gdsp_bin_intervals: more than 2^32 (interval, tile) pairs");  return GDSP_EINVAL; }
	for (uint32_t t=0 ; t<ntiles ; t++) h_tileOffsets[t+1] += h_tileOffsets[t];
	if (h_tileList == NULL) return GDSP_OK;

	// fill in file order; a moving cursor per tile (kept in the offsets, restored after)
	for (uint32_t i=0 ; i<count ; i++)
		{
		uint32_t s = h_start[i], e = h_end[i];
		if (e > n) e = n;
		if (s >= e) continue;
		uint32_t t0 = s / IV_TILE, t1 = (e - 1) / IV_TILE;
		for (uint32_t t=t0 ; t<=t1 ; t++) h_tileList[h_tileOffsets[t]++] = i;
		}
	for (uint32_t t=ntiles ; t>0 ; t--) h_tileOffsets[t] = h_tileOffsets[t-1];
	h_tileOffsets[0] = 0;
	return GDSP_OK;
	}

} // extern "C"
