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
