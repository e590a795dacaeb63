/* ops_fused.c -- operator chains that run as one kernel.
 *
 * The reference runs a maximal run of per-chromosome operators back to back on each
 * chromosome "for cache performance" (genodsp.c:895-921).  On the GPU the analogue is not to
 * let the intermediate signal leave the chip at all: for the chains the device library fuses
 * (`= smooth = localmax|localmin`, `= dilate = erode [= binarize]`) the driver makes one
 * call, with results bit-identical to the separate operators (tests/test_hip_parity.py). */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

dspprototypes(op_smooth)  dspprototypes(op_local_minima)  dspprototypes(op_local_maxima)
dspprototypes(op_dilate)  dspprototypes(op_erode)         dspprototypes(op_binarize)

dspprototypes(op_best_local_min)  dspprototypes(op_best_local_max)  dspprototypes(op_close)  dspprototypes(op_open)
dspprototypes(op_clip)  dspprototypes(op_erase)  dspprototypes(op_add_constant)  dspprototypes(op_absolute_value)  dspprototypes(op_map)

int op_reach (dspop* op, u32* left, u32* right)
	{
	opfunc_apply f = op->funcApply;
	*left = *right = 0;
	if (f == op_smooth_apply)                              /* sum.c:647-663: taps -h..+h */
		{ *left = *right = (op_smooth_window (op) - 1) / 2;  return true; }
	if ((f == op_local_maxima_apply) || (f == op_local_minima_apply))      /* minmax.c:1195-1216 */
		{ u32 N;  int wantMax;  valtype fill;  op_local_describe (op, &N, &wantMax, &fill);  *left = *right = (N - 1) / 2;  return true; }
	if ((f == op_best_local_max_apply) || (f == op_best_local_min_apply))  /* minmax.c:1636-1640: [i-wLft, i+wRgt] */
		{ u32 W = op_best_window (op);  *left = (W - 1) / 2;  *right = W - 1 - *left;  return true; }
	if ((f == op_dilate_apply) || (f == op_erode_apply) || (f == op_close_apply) || (f == op_open_apply))
		{ op_morph_reach (op, left, right);  return (*left != u32Max); }
	if ((f == op_binarize_apply) || (f == op_clip_apply) || (f == op_erase_apply) || (f == op_add_constant_apply)
	 || (f == op_absolute_value_apply) || (f == op_map_apply))
		return true;
	return false;                                          /* sum, slidingsum, cumulativesum, clump, anticlump, plugins */
	}

int try_fused_apply (dspop* op, dspop* stopOp, spec* s)
	{
	dspop* next = op->next;
	if ((next == NULL) || (next == stopOp)) return 0;

	if ((op->funcApply == op_smooth_apply)
	 && ((next->funcApply == op_local_maxima_apply) || (next->funcApply == op_local_minima_apply)))
		{
		u32 N;  int wantMax;  valtype fill;
		op_local_describe (next, &N, &wantMax, &fill);
		if (!gdsp_smooth_local_extrema_fusable (op_smooth_window (op), N)) return 0;
		check_gdsp (gdsp_smooth_local_extrema (s->valVector, partner_vector (s->chrom), s->length,
		                                       op_smooth_window (op), firMode, N, wantMax, fill, op_stream ()), op->name);
		flip_vector (s->chrom);
		return 2;
		}

	if ((op->funcApply == op_dilate_apply) && (next->funcApply == op_erode_apply))
		{
		u32 dl, dr, el, er;  valtype dT, dOne, dZero, eT, eOne, eZero;
		valtype bT = 0, bOne = 1, bZero = 0;  int bTies = false, withBinarize = false;
		op_morph_describe (op,   &dl, &dr, &dT, &dOne, &dZero);
		op_morph_describe (next, &el, &er, &eT, &eOne, &eZero);
		dspop* third = next->next;
		if ((third != NULL) && (third != stopOp) && (third->funcApply == op_binarize_apply))
			{ op_binarize_describe (third, &bT, &bTies, &bOne, &bZero);  withBinarize = true; }
		int rc = gdsp_dilate_erode (s->valVector, partner_vector (s->chrom), s->length,
		                            dl, dr, dT, dOne, dZero, el, er, eT, eOne, eZero,
		                            withBinarize, bT, bTies, bOne, bZero, op_stream ());
		if (rc == GDSP_EINVAL) return 0;          /* reach too long for one tile: run them separately */
		check_gdsp (rc, op->name);
		flip_vector (s->chrom);
		return withBinarize? 3 : 2;
		}
	return 0;
	}
