/* ops_sum.c -- operators sum, slidingsum, smooth, cumulativesum (device shims).
 *
 * Plugin surface and argument rules of the reference's sum.c (parse :81-198,
 * :302-396, :521-603, :720-763); each apply launches the matching kernel of
 * libgenodsp_hip.so on the chromosome's stream instead of looping on the host. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <float.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

typedef struct dspop_sum
	{
	dspop   common;
	u32     windowSize;
	valtype denominator;
	valtype zeroVal;
	int     windowIsChromosome, useActualDenom, denomIsWindowSize;
	} dspop_sum;

/* ------------------------------------------------------------------ sum ---- */
OP_SHORT (op_window_sum, "sum over non-overlapping windows")

void op_window_sum_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sReplace the signal by its sum over non-overlapping windows: the sum lands on the\n", indent);
	fprintf (f, "%swindow's first base, the window's other bases get the zero value.\n\n", indent);
	fprintf (f, "%susage: %s [options]\n", indent, name);
	fprintf (f, "%s  --window=<length>|chromosome   (W=) window size (default: global window, else 100)\n", indent);
	fprintf (f, "%s  --denom=<value>|window|actual  (D=) divide the sum (default 1)\n", indent);
	fprintf (f, "%s  --zero=<value>                 (Z=) fill for the rest of the window (default 0)\n", indent);
	}

dspop* op_window_sum_parse (char* name, int argc, char** argv)
	{
	dspop_sum* op = (dspop_sum*) new_op (name, sizeof(dspop_sum), false);
	op->windowSize  = (u32) get_named_global ("windowSize", 100);
	op->denominator = 1.0;
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if (strcmp (arg, "--window=chromosome") == 0) { op->windowIsChromosome = true;  continue; }
		if (is_opt3 (arg, "window", "W"))
			{ op->windowSize = window_arg (name, arg, argVal, "window size");  op->windowIsChromosome = false;  continue; }
		if (is_opt3 (arg, "denom", "D") || (strcmp_prefix (arg, "--denominator=") == 0))
			{
			op->denominator = 1.0;  op->useActualDenom = false;  op->denomIsWindowSize = false;
			if (strcmp (argVal, "actual") == 0) { op->useActualDenom = true;  continue; }
			if ((strcmp (argVal, "window") == 0) || (strcmp (argVal, "W") == 0)) { op->denomIsWindowSize = true;  continue; }
			op->denominator = string_to_valtype (argVal);
			if (op->denominator == 0) chastise ("[%s] denominator can't be zero (\"%s\")\n", name, arg);
			continue;
			}
		if (is_opt3 (arg, "zero", "Z")) { op->zeroVal = string_to_valtype (argVal);  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	if (!op->windowIsChromosome && (op->windowSize < 3))
		{ fprintf (stderr, "[%s] WARNING: raising window size from %d to %d\n", name, op->windowSize, 3);  op->windowSize = 3; }
	return (dspop*) op;
	}

void op_window_sum_free (dspop* op) { free (op); }

void op_window_sum_apply (dspop* _op, arg_dont_complain(char* vName), u32 vLen, valtype* v)
	{
	dspop_sum* op = (dspop_sum*) _op;
	u32     w = op->windowIsChromosome? vLen : op->windowSize;              /* sum.c:225-226 */
	valtype d = op->denomIsWindowSize? (valtype) w : op->denominator;
	check_gdsp (gdsp_window_sum (v, vLen, w, d, op->useActualDenom, op->zeroVal, op_stream ()), _op->name);
	}

/* ----------------------------------------------------------- slidingsum ---- */
OP_SHORT (op_sliding_sum, "continuous sum over overlapping windows")

void op_sliding_sum_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sReplace every base by the sum over the window centred on it; bases beyond the\n", indent);
	fprintf (f, "%sends of the chromosome count as zero.\n\n", indent);
	fprintf (f, "%susage: %s [options]\n", indent, name);
	fprintf (f, "%s  --window=<length>        (W=) window size (default: global window, else 100)\n", indent);
	fprintf (f, "%s  --denom=<value>|window   (D=) divide the sum (default 1)\n", indent);
	}

dspop* op_sliding_sum_parse (char* name, int argc, char** argv)
	{
	dspop_sum* op = (dspop_sum*) new_op (name, sizeof(dspop_sum), false);
	op->windowSize  = (u32) get_named_global ("windowSize", 100);
	op->denominator = 1.0;
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if (is_opt3 (arg, "window", "W")) { op->windowSize = window_arg (name, arg, argVal, "window size");  continue; }
		if (is_opt3 (arg, "denom", "D") || (strcmp_prefix (arg, "--denominator=") == 0))
			{
			if ((strcmp (argVal, "window") == 0) || (strcmp (argVal, "W") == 0)) { op->denominator = op->windowSize;  continue; }
			op->denominator = string_to_valtype (argVal);
			if (op->denominator == 0) chastise ("[%s] denominator can't be zero (\"%s\")\n", name, arg);
			continue;
			}
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	if (op->windowSize < 3)
		{ fprintf (stderr, "[%s] WARNING: raising window size from %d to %d\n", name, op->windowSize, 3);  op->windowSize = 3; }
	return (dspop*) op;
	}

void op_sliding_sum_free (dspop* op) { free (op); }

void op_sliding_sum_apply (dspop* _op, char* vName, u32 vLen, valtype* v)
	{
	dspop_sum* op = (dspop_sum*) _op;
	int rc = gdsp_sliding_sum (v, partner_vector (vName), vLen, op->windowSize, op->denominator, op_stream ());
	if (rc == GDSP_EINVAL)                       /* window beyond one LDS tile: prefix-difference form */
		{
		size_t bytes;  void* work = long_window_workspace (&bytes);
		rc = gdsp_sliding_sum_any (v, partner_vector (vName), vLen, op->windowSize, op->denominator, work, bytes, op_stream ());
		}
	check_gdsp (rc, _op->name);
	flip_vector (vName);
	}

/* --------------------------------------------------------------- smooth ---- */
#define maxWindowSize ((50*1000)+1)              /* sum.c:478 */

typedef struct dspop_smooth { dspop common;  u32 windowSize; } dspop_smooth;

OP_SHORT (op_smooth, "apply a smoothing filter (Hann window)")

void op_smooth_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sSmooth with a Hann window: every base becomes the weighted sum over the window\n", indent);
	fprintf (f, "%scentred on it; bases beyond the ends of the chromosome count as zero.\n\n", indent);
	fprintf (f, "%susage: %s [options]\n", indent, name);
	fprintf (f, "%s  --window=<length>        (W=) window size, raised by one if even (default 101)\n", indent);
	}

dspop* op_smooth_parse (char* name, int argc, char** argv)
	{
	dspop_smooth* op = (dspop_smooth*) new_op (name, sizeof(dspop_smooth), false);
	op->windowSize = (u32) get_named_global ("windowSize", 101);
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if (is_opt3 (arg, "window", "W"))
			{
			int w = string_to_unitized_int (argVal, /*thousands*/ true);
			if (w == 0) chastise ("[%s] window size can't be zero (\"%s\")\n", name, arg);
			if (w < 0)  chastise ("[%s] window size can't be negative (\"%s\")\n", name, arg);
			if (w > maxWindowSize) chastise ("[%s] window size exceeds %u (\"%s\")\n", name, maxWindowSize, arg);
			if (w < 3) { fprintf (stderr, "[%s] WARNING: raising window size from %d to %d\n", name, w, 3);  w = 3; }
			if ((w & 1) == 0) { fprintf (stderr, "[%s] WARNING: raising window size from %d to %d\n", name, w, w+1);  w++; }
			op->windowSize = (u32) w;
			continue;
			}
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	if ((op->windowSize & 1) == 0)
		{ fprintf (stderr, "[%s] WARNING: raising window size from %d to %d\n", name, op->windowSize, op->windowSize+1);  op->windowSize++; }
	return (dspop*) op;
	}

void op_smooth_free (dspop* op) { free (op); }

u32 op_smooth_window (dspop* op) { return ((dspop_smooth*) op)->windowSize; }

dspprototypes(op_local_minima)  dspprototypes(op_local_maxima)

void op_smooth_apply (dspop* _op, char* vName, u32 vLen, valtype* v)
	{
	dspop_smooth* op = (dspop_smooth*) _op;
	int mode = firMode;
	/* --smooth=hann is not shift invariant (a flat stretch comes out with last-bit differences), and
	 * localmin / localmax compare neighbours strictly: a smooth that feeds one of them is evaluated with
	 * direct taps (fma), fused or not, so the peaks do not depend on --nofuse (ops_fused.c does the same) */
	if ((mode == GDSP_FIR_HANN) && (_op->next != NULL)
	 && ((_op->next->funcApply == op_local_maxima_apply) || (_op->next->funcApply == op_local_minima_apply)))
		mode = GDSP_FIR_FMA;
	check_gdsp (gdsp_smooth (v, partner_vector (vName), vLen, op->windowSize, mode, op_stream ()), _op->name);
	flip_vector (vName);                              /* no copy-back pass (sum.c:672-673) */
	}

/* -------------------------------------------------------- cumulativesum ---- */
OP_SHORT (op_cumulative_sum, "compute the cumulative sum of the current set of interval values")

void op_cumulative_sum_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sReplace the signal by its running total, chromosome by chromosome.\n\n", indent);
	fprintf (f, "%susage: %s\n", indent, name);
	}

dspop* op_cumulative_sum_parse (char* name, int argc, char** argv)
	{
	dspop* op = (dspop*) new_op (name, sizeof(dspop), false);
	if (argc > 0) chastise ("[%s] Can't understand \"%s\"\n", name, argv[0]);
	return op;
	}

void op_cumulative_sum_free (dspop* op) { free (op); }

void op_cumulative_sum_apply (dspop* _op, arg_dont_complain(char* vName), u32 vLen, valtype* v)
	{
	valtype* work = get_scratch_vector ();            /* chunk totals fit easily */
	check_gdsp (gdsp_cumulative_sum (v, vLen, work, op_stream ()), _op->name);
	release_scratch_vector (work);
	}
