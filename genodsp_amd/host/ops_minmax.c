/* ops_minmax.c -- localmin, localmax, bestmin, bestmax (device shims).
 * Argument rules: minmax.c:883-967 (localmin), :1085-1167 (localmax), :1280-1356 (bestmin),
 * :1527-1603 (bestmax) in the reference. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <float.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

typedef struct dspop_local { dspop common;  u32 neighborhood;  valtype fill;  int wantMax; } dspop_local;
typedef struct dspop_best  { dspop common;  u32 windowSize;    int wantMax; } dspop_best;

static dspop* local_parse (char* name, int argc, char** argv, int wantMax)
	{
	dspop_local* op = (dspop_local*) new_op (name, sizeof(dspop_local), false);
	op->neighborhood = 3;
	op->wantMax      = wantMax;
	op->fill         = wantMax? 0.0 : valtypeMax;       /* minmax.c:1098, :901 */
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if (is_opt3 (arg, "neighborhood", "N"))
			{
			u32 n = window_arg (name, arg, argVal, "neighborhood");
			if ((n & 1) == 0) { fprintf (stderr, "[%s] WARNING: raising neighborhood from %d to %d\n", name, n, n+1);  n++; }
			op->neighborhood = n;
			continue;
			}
		if (wantMax  && is_opt3 (arg, "zero", "Z"))                  { op->fill = string_to_valtype (argVal);  continue; }
		if (!wantMax && (strcmp_prefix (arg, "--infinity=") == 0))   { op->fill = string_to_valtype (argVal);  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	return (dspop*) op;
	}

static void local_usage (char* name, FILE* f, char* indent, int wantMax)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sKeep a base only if no other base within the neighborhood is strictly %s;\n", indent, wantMax? "greater" : "smaller");
	fprintf (f, "%severything else becomes the %s value. Plateaus survive whole.\n\n", indent, wantMax? "zero" : "infinity");
	fprintf (f, "%susage: %s [options]\n", indent, name);
	fprintf (f, "%s  --neighborhood=<length>  (N=) odd neighborhood size (default 3)\n", indent);
	if (wantMax) fprintf (f, "%s  --zero=<value>           (Z=) value for non-maxima (default 0.0)\n", indent);
	else         fprintf (f, "%s  --infinity=<value>       value for non-minima (default: largest double)\n", indent);
	}

void op_local_describe (dspop* _op, u32* neighborhood, int* wantMax, valtype* fill)
	{
	dspop_local* op = (dspop_local*) _op;
	*neighborhood = op->neighborhood;  *wantMax = op->wantMax;  *fill = op->fill;
	}

static void local_apply (dspop* _op, char* vName, u32 vLen, valtype* v)
	{
	dspop_local* op = (dspop_local*) _op;
	int rc = gdsp_local_extrema (v, partner_vector (vName), vLen, op->neighborhood, op->wantMax, op->fill, op_stream ());
	if (rc == GDSP_EINVAL)                       /* neighbourhood beyond one LDS tile: whole-vector form */
		{
		size_t bytes;  void* work = long_window_workspace (&bytes);
		rc = gdsp_local_extrema_any (v, partner_vector (vName), vLen, op->neighborhood, op->wantMax, op->fill, work, bytes, op_stream ());
		}
	check_gdsp (rc, _op->name);
	flip_vector (vName);
	}

OP_SHORT (op_local_minima, "find local minima")
void   op_local_minima_usage (char* name, FILE* f, char* indent) { local_usage (name, f, indent, false); }
dspop* op_local_minima_parse (char* name, int argc, char** argv) { return local_parse (name, argc, argv, false); }
void   op_local_minima_free  (dspop* op) { free (op); }
void   op_local_minima_apply (dspop* op, char* vName, u32 vLen, valtype* v) { local_apply (op, vName, vLen, v); }

OP_SHORT (op_local_maxima, "find local maxima")
void   op_local_maxima_usage (char* name, FILE* f, char* indent) { local_usage (name, f, indent, true); }
dspop* op_local_maxima_parse (char* name, int argc, char** argv) { return local_parse (name, argc, argv, true); }
void   op_local_maxima_free  (dspop* op) { free (op); }
void   op_local_maxima_apply (dspop* op, char* vName, u32 vLen, valtype* v) { local_apply (op, vName, vLen, v); }

static dspop* best_parse (char* name, int argc, char** argv, int wantMax)
	{
	dspop_best* op = (dspop_best*) new_op (name, sizeof(dspop_best), false);
	op->windowSize = (u32) get_named_global ("windowSize", 100);
	op->wantMax    = wantMax;
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if (is_opt3 (arg, "window", "W")) { op->windowSize = window_arg (name, arg, argVal, "window size");  continue; }
		if (strcmp (arg, "--debug") == 0) continue;
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	return (dspop*) op;
	}

u32 op_best_window (dspop* op) { return ((dspop_best*) op)->windowSize; }

static void best_usage (char* name, FILE* f, char* indent, int wantMax)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sReplace every base by the %s over the window centred on it.\n\n", indent, wantMax? "maximum" : "minimum");
	fprintf (f, "%susage: %s [options]\n", indent, name);
	fprintf (f, "%s  --window=<length>        (W=) window size (default: global window, else 100)\n", indent);
	}

static void best_apply (dspop* _op, char* vName, u32 vLen, valtype* v)
	{
	dspop_best* op = (dspop_best*) _op;
	int rc = gdsp_best_extrema (v, partner_vector (vName), vLen, op->windowSize, op->wantMax, op_stream ());
	if (rc == GDSP_EINVAL)
		{
		size_t bytes;  void* work = long_window_workspace (&bytes);
		rc = gdsp_best_extrema_any (v, partner_vector (vName), vLen, op->windowSize, op->wantMax, work, bytes, op_stream ());
		}
	check_gdsp (rc, _op->name);
	flip_vector (vName);
	}

OP_SHORT (op_best_local_min, "find the minimum in a sliding window")
void   op_best_local_min_usage (char* name, FILE* f, char* indent) { best_usage (name, f, indent, false); }
dspop* op_best_local_min_parse (char* name, int argc, char** argv) { return best_parse (name, argc, argv, false); }
void   op_best_local_min_free  (dspop* op) { free (op); }
void   op_best_local_min_apply (dspop* op, char* vName, u32 vLen, valtype* v) { best_apply (op, vName, vLen, v); }

OP_SHORT (op_best_local_max, "find the maximum in a sliding window")
void   op_best_local_max_usage (char* name, FILE* f, char* indent) { best_usage (name, f, indent, true); }
dspop* op_best_local_max_parse (char* name, int argc, char** argv) { return best_parse (name, argc, argv, true); }
void   op_best_local_max_free  (dspop* op) { free (op); }
void   op_best_local_max_apply (dspop* op, char* vName, u32 vLen, valtype* v) { best_apply (op, vName, vLen, v); }
