/* ops_morphology.c -- close, open, dilate, erode (device shims).
 * Argument rules: morphology.c:96-215 (close), :396-512 (open), :696-866 (dilate),
 * :1163-1315 (erode) in the reference. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <float.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

enum { M_CLOSE, M_OPEN, M_DILATE, M_ERODE };
static const char* lengthWord[] = { "closing", "opening", "dilation", "erosion" };

typedef struct dspop_morph
	{
	dspop   common;
	int     kind;
	valtype length;                 /* <length> argument */
	u32     left, right;            /* --left / --right (dilate, erode) */
	int     haveThreshold;
	char*   thresholdVarName;
	valtype threshold, oneVal, zeroVal;
	} dspop_morph;

static dspop* morph_parse (char* name, int argc, char** argv, int kind)
	{
	dspop_morph* op = (dspop_morph*) new_op (name, sizeof(dspop_morph), false);
	int haveLength = false;
	op->kind   = kind;
	op->oneVal = 1.0;
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if (is_opt3 (arg, "threshold", "T"))
			{
			if (op->haveThreshold)
				{ fprintf (stderr, "[%s] threshold specified more than once (at \"%s\")\n", name, arg);  exit (EXIT_FAILURE); }
			value_or_variable (argVal, &op->threshold, &op->thresholdVarName);
			op->haveThreshold = true;
			continue;
			}
		if (is_opt3 (arg, "one", "O"))  { op->oneVal  = string_to_valtype (argVal);  continue; }
		if (is_opt3 (arg, "zero", "Z")) { op->zeroVal = string_to_valtype (argVal);  continue; }
		if (((kind == M_DILATE) || (kind == M_ERODE))
		 && ((strcmp_prefix (arg, "--left=") == 0) || (strcmp_prefix (arg, "--right=") == 0)))
			{
			u32 amount;
			if (kind == M_ERODE) amount = (u32) string_to_valtype (argVal);            /* plain number, :1238-1250 */
			else if (strcmp_suffix (argVal, "-1") == 0)                               /* "<n>-1", :773-779 */
				{
				char* t = copy_string (argVal);
				t[strlen (t) - 2] = 0;
				amount = (u32) (string_to_unitized_int (t, /*thousands*/ true) - 1);
				free (t);
				}
			else amount = (u32) string_to_unitized_int (argVal, /*thousands*/ true);
			if (arg[2] == 'l') op->left = amount;  else op->right = amount;
			continue;
			}
		if (strcmp (arg, "--debug") == 0) continue;
		if (strcmp_prefix (arg, "--") == 0) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if (!haveLength)
			{ op->length = (u32) string_to_unitized_int (arg, /*thousands*/ true);  haveLength = true;  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	if (haveLength && ((op->left != 0) || (op->right != 0)))
		{ fprintf (stderr, "[%s] %s length was provided in more than one way\n", name, lengthWord[kind]);  exit (EXIT_FAILURE); }
	if (!haveLength && (op->left == 0) && (op->right == 0))
		{ fprintf (stderr, "[%s] %s length was not provided\n", name, lengthWord[kind]);  exit (EXIT_FAILURE); }
	return (dspop*) op;
	}

static void morph_free (dspop* _op)
	{
	dspop_morph* op = (dspop_morph*) _op;
	if (op->thresholdVarName != NULL) free (op->thresholdVarName);
	free (op);
	}

/* parameters as morph_apply would use them (resolves a threshold variable, with the usual note) */
/* how far an output looks: dilate / erode test [i-right, i+left]; close / open decide a run from its length, at most
 * <length>+1 bases either way (no variable is resolved here: this is asked before the pipeline runs) */
void op_morph_reach (dspop* _op, u32* left, u32* right)
	{
	dspop_morph* op = (dspop_morph*) _op;
	if ((op->kind == M_DILATE) || (op->kind == M_ERODE))
		{
		u32 l = op->left, r = op->right;
		if ((l == 0) && (r == 0)) { l = (u32) (op->length / 2);  r = (u32) (op->length - l); }
		*left = r;  *right = l;
		return;
		}
	*left = *right = (op->length < 4.0e9)? (u32) op->length + 1 : u32Max;
	}

void op_morph_describe (dspop* _op, u32* left, u32* right, valtype* T, valtype* one, valtype* zero)
	{
	dspop_morph* op = (dspop_morph*) _op;
	resolve_variable (_op, &op->thresholdVarName, &op->threshold, "threshold");
	*left = op->left;  *right = op->right;
	if ((*left == 0) && (*right == 0))
		{ *left = (u32) (op->length / 2);  *right = (u32) (op->length - *left); }
	*T = op->threshold;  *one = op->oneVal;  *zero = op->zeroVal;
	}

static void morph_apply (dspop* _op, char* vName, u32 vLen, valtype* v)
	{
	dspop_morph* op = (dspop_morph*) _op;
	valtype* out = partner_vector (vName);
	void*    st  = op_stream ();
	int      rc;
	resolve_variable (_op, &op->thresholdVarName, &op->threshold, "threshold");
	u32 left = op->left, right = op->right;
	if ((left == 0) && (right == 0))
		{ left = (u32) (op->length / 2);  right = (u32) (op->length - left); }            /* morphology.c:917-920 */
	switch (op->kind)
		{
		case M_CLOSE:  rc = gdsp_close  (v, out, vLen, op->length, op->threshold, op->oneVal, op->zeroVal, st);  break;
		case M_OPEN:   rc = gdsp_open   (v, out, vLen, op->length, op->threshold, op->oneVal, op->zeroVal, st);  break;
		case M_DILATE: rc = gdsp_dilate (v, out, vLen, left, right, op->threshold, op->oneVal, op->zeroVal, st); break;
		default:       rc = gdsp_erode  (v, out, vLen, left, right, op->threshold, op->oneVal, op->zeroVal, st); break;
		}
	if (rc == GDSP_EINVAL)                       /* reach beyond one LDS tile: the set as bits in HBM workspace, any length */
		{
		size_t bytes;  void* work = long_window_workspace (&bytes);
		switch (op->kind)
			{
			case M_CLOSE:  rc = gdsp_close_any  (v, out, vLen, op->length, op->threshold, op->oneVal, op->zeroVal, work, bytes, st);  break;
			case M_OPEN:   rc = gdsp_open_any   (v, out, vLen, op->length, op->threshold, op->oneVal, op->zeroVal, work, bytes, st);  break;
			case M_DILATE: rc = gdsp_dilate_any (v, out, vLen, left, right, op->threshold, op->oneVal, op->zeroVal, work, bytes, st); break;
			default:       rc = gdsp_erode_any  (v, out, vLen, left, right, op->threshold, op->oneVal, op->zeroVal, work, bytes, st); break;
			}
		}
	check_gdsp (rc, _op->name);
	flip_vector (vName);
	}

static void morph_usage (char* name, FILE* f, char* indent, int kind)
	{
	static const char* what[] =
		{ "Fill gaps (runs at or below the threshold) no longer than the given length that lie\nbetween two intervals; the signal is binarised.",
		  "Remove intervals (runs above the threshold) no longer than the given length; the\nsignal is binarised.",
		  "Widen intervals (runs above the threshold) by the given length, split between the\ntwo sides; the signal is binarised.",
		  "Shrink intervals (runs above the threshold) by the given length, split between the\ntwo sides; the signal is binarised." };
	if (indent == NULL) indent = "";
	char* text = copy_string (what[kind]);
	for (char* line = strtok (text, "\n") ; line != NULL ; line = strtok (NULL, "\n")) fprintf (f, "%s%s\n", indent, line);
	free (text);
	fprintf (f, "%s\n%susage: %s <length> [options]\n", indent, indent, name);
	fprintf (f, "%s  --threshold=<value|variable>  (T=) values above this are \"in\" (default 0.0)\n", indent);
	fprintf (f, "%s  --one=<value>  --zero=<value>  (O= Z=) output values (default 1.0 and 0.0)\n", indent);
	if ((kind == M_DILATE) || (kind == M_ERODE))
		fprintf (f, "%s  --left=<length> --right=<length>  one-sided amounts instead of <length>\n", indent);
	}

OP_SHORT (op_close, "fill short gaps between intervals (and binarize)")
void   op_close_usage (char* name, FILE* f, char* indent) { morph_usage (name, f, indent, M_CLOSE); }
dspop* op_close_parse (char* name, int argc, char** argv) { return morph_parse (name, argc, argv, M_CLOSE); }
void   op_close_free  (dspop* op) { morph_free (op); }
void   op_close_apply (dspop* op, char* vName, u32 vLen, valtype* v) { morph_apply (op, vName, vLen, v); }

OP_SHORT (op_open, "remove short intervals (and binarize)")
void   op_open_usage (char* name, FILE* f, char* indent) { morph_usage (name, f, indent, M_OPEN); }
dspop* op_open_parse (char* name, int argc, char** argv) { return morph_parse (name, argc, argv, M_OPEN); }
void   op_open_free  (dspop* op) { morph_free (op); }
void   op_open_apply (dspop* op, char* vName, u32 vLen, valtype* v) { morph_apply (op, vName, vLen, v); }

OP_SHORT (op_dilate, "widen intervals (and binarize)")
void   op_dilate_usage (char* name, FILE* f, char* indent) { morph_usage (name, f, indent, M_DILATE); }
dspop* op_dilate_parse (char* name, int argc, char** argv) { return morph_parse (name, argc, argv, M_DILATE); }
void   op_dilate_free  (dspop* op) { morph_free (op); }
void   op_dilate_apply (dspop* op, char* vName, u32 vLen, valtype* v) { morph_apply (op, vName, vLen, v); }

OP_SHORT (op_erode, "shrink intervals (and binarize)")
void   op_erode_usage (char* name, FILE* f, char* indent) { morph_usage (name, f, indent, M_ERODE); }
dspop* op_erode_parse (char* name, int argc, char** argv) { return morph_parse (name, argc, argv, M_ERODE); }
void   op_erode_free  (dspop* op) { morph_free (op); }
void   op_erode_apply (dspop* op, char* vName, u32 vLen, valtype* v) { morph_apply (op, vName, vLen, v); }
