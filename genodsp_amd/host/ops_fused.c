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

/* ---- one launch per operator per device (gdsp_*_batch): the chromosome loop of genodsp.c:909-921 turned inside out.
 * Chromosomes are independent for every per-chromosome operator, so "every operator of the run on chromosome 1, then
 * on chromosome 2, ..." and "operator 1 on every chromosome, then operator 2, ..." give the same signal; the second
 * order lets one grid cover all the vectors a device owns (no ramp and drain between 24 short kernels). */
int op_batchable (dspop* op)
	{
	opfunc_apply f = op->funcApply;
	return (f == op_smooth_apply) || (f == op_local_maxima_apply) || (f == op_local_minima_apply)
	    || (f == op_best_local_max_apply) || (f == op_best_local_min_apply)
	    || (f == op_dilate_apply) || (f == op_erode_apply)
	    || (f == op_binarize_apply) || (f == op_clip_apply) || (f == op_erase_apply)
	    || (f == op_add_constant_apply) || (f == op_absolute_value_apply);
	}

/* apply op -- and the operators fused behind it -- to units[0..nunits), all of them on the current device;
 * returns how many operators were consumed (>= 1; op must be batchable) */
int batch_apply_on_device (dspop* op, dspop* stopOp, spec** units, int nunits, int allowFusion)
	{
	opfunc_apply f = op->funcApply;
	dspop* next = op->next;
	if (next == stopOp) next = NULL;
	gdsp_batch_item* items = (gdsp_batch_item*) calloc (nunits? nunits : 1, sizeof(gdsp_batch_item));
	if (items == NULL) { fprintf (stderr, "out of memory\n");  exit (EXIT_FAILURE); }
	/* (the in-place operators never ask for a partner: a pipeline of them alone runs without the partners' arena) */
	const int inPlace = (f == op_binarize_apply) || (f == op_clip_apply) || (f == op_erase_apply)
	                 || (f == op_add_constant_apply) || (f == op_absolute_value_apply);
	for (int i=0 ; i<nunits ; i++)
		{ items[i].d_in = units[i]->valVector;  items[i].d_out = inPlace? NULL : partner_of (units[i]);  items[i].n = units[i]->length; }
	void* st = op_stream ();
	int   consumed = 1, outOfPlace = true, rc = GDSP_OK;

	if (f == op_smooth_apply)
		{
		int mode = firMode;
		int feedsLocal = (next != NULL) && ((next->funcApply == op_local_maxima_apply) || (next->funcApply == op_local_minima_apply));
		if ((mode == GDSP_FIR_HANN) && feedsLocal) mode = GDSP_FIR_FMA;             /* as op_smooth_apply / try_fused_apply */
		u32 N = 0;  int wantMax = 0;  valtype fill = 0;
		if (feedsLocal) op_local_describe (next, &N, &wantMax, &fill);
		if (allowFusion && feedsLocal && gdsp_smooth_local_extrema_fusable (op_smooth_window (op), N))
			{ rc = gdsp_smooth_local_extrema_batch (items, nunits, op_smooth_window (op), mode, N, wantMax, fill, st);  consumed = 2; }
		else rc = gdsp_smooth_batch (items, nunits, op_smooth_window (op), mode, st);
		}
	else if ((f == op_local_maxima_apply) || (f == op_local_minima_apply))
		{
		u32 N;  int wantMax;  valtype fill;
		op_local_describe (op, &N, &wantMax, &fill);
		rc = gdsp_local_extrema_batch (items, nunits, N, wantMax, fill, st);
		if (rc == GDSP_EINVAL) goto one_by_one;              /* neighbourhood beyond one LDS tile */
		}
	else if ((f == op_best_local_max_apply) || (f == op_best_local_min_apply))
		{
		rc = gdsp_best_extrema_batch (items, nunits, op_best_window (op), f == op_best_local_max_apply, st);
		if (rc == GDSP_EINVAL) goto one_by_one;              /* window beyond one LDS tile */
		}
	else if ((f == op_dilate_apply) || (f == op_erode_apply))
		{
		u32 l, r;  valtype T, one, zero;
		op_morph_describe (op, &l, &r, &T, &one, &zero);
		rc = GDSP_EINVAL;
		if (allowFusion && (f == op_dilate_apply) && (next != NULL) && (next->funcApply == op_erode_apply))
			{
			u32 el, er;  valtype eT, eOne, eZero, bT = 0, bOne = 1, bZero = 0;  int bTies = false, withBinarize = false;
			op_morph_describe (next, &el, &er, &eT, &eOne, &eZero);
			if (gdsp_dilate_erode_fusable (l, r, el, er))          /* (a property of the reaches alone: every device decides alike) */
				{
				dspop* third = next->next;
				if ((third != NULL) && (third != stopOp) && (third->funcApply == op_binarize_apply))
					{ op_binarize_describe (third, &bT, &bTies, &bOne, &bZero);  withBinarize = true; }
				rc = gdsp_dilate_erode_batch (items, nunits, l, r, T, one, zero, el, er, eT, eOne, eZero,
				                              withBinarize, bT, bTies, bOne, bZero, st);
				consumed = withBinarize? 3 : 2;
				}
			}
		if ((rc == GDSP_EINVAL) && (consumed == 1))           /* not fused (or the combined reach is beyond one tile) */
			{
			rc = (f == op_dilate_apply)? gdsp_dilate_batch (items, nunits, l, r, T, one, zero, st)
			                           : gdsp_erode_batch  (items, nunits, l, r, T, one, zero, st);
			if (rc == GDSP_EINVAL) goto one_by_one;            /* reach beyond one LDS tile */
			}
		}
	else
		{
		outOfPlace = false;
		for (int i=0 ; i<nunits ; i++) { items[i].d_in = NULL;  items[i].d_out = units[i]->valVector; }    /* in place */
		if (f == op_binarize_apply)
			{
			valtype T, one, zero;  int ties;
			op_binarize_describe (op, &T, &ties, &one, &zero);
			rc = gdsp_binarize_batch (items, nunits, T, ties, one, zero, st);
			}
		else if ((f == op_clip_apply) || (f == op_erase_apply))
			{
			int haveMin, haveMax, keepInside;  valtype lo, hi, zero;
			op_limits_describe (op, &haveMin, &lo, &haveMax, &hi, &keepInside, &zero);
			rc = (f == op_clip_apply)? gdsp_clip_batch  (items, nunits, haveMin, lo, haveMax, hi, st)
			                         : gdsp_erase_batch (items, nunits, haveMin, lo, haveMax, hi, keepInside, zero, st);
			}
		else if (f == op_add_constant_apply) rc = gdsp_add_constant_batch (items, nunits, op_add_constant_value (op), st);
		else                                 rc = gdsp_abs_batch (items, nunits, st);
		}
	free (items);
	check_gdsp (rc, op->name);
	if (outOfPlace) { for (int i=0 ; i<nunits ; i++) flip_spec (units[i]); }
	return consumed;

	/* no tiled kernel takes this window: the operator's own apply, vector by vector, which goes on to its
	 * whole-vector route (gdsp_*_any); nothing has been flipped, whatever a refused batch call wrote went to partners */
one_by_one:
	free (items);
	for (int i=0 ; i<nunits ; i++) apply_to_unit (op, units[i]);
	return 1;
	}
