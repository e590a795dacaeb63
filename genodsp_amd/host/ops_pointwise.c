/* ops_pointwise.c -- binarize, clip, erase, addconst, abs, invert, variables (device shims).
 * Argument rules: logical.c:82-200 (binarize), mask.c:716-833 (clip), :1003-1131 (erase),
 * add.c:655-708 (addconst), :795-873 (invert), :976-1020 (abs), variables.c in the reference. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <float.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

/* ------------------------------------------------------------- binarize ---- */
typedef struct dspop_binarize
	{ dspop common;  char* thresholdVarName;  valtype threshold;  int tiesAbove;  valtype oneVal, zeroVal; } dspop_binarize;

OP_SHORT (op_binarize, "binarize the current set of interval values")

void op_binarize_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sSet every base to one of two values by comparing it with a threshold.\n\n", indent);
	fprintf (f, "%susage: %s [<threshold>] [options]\n", indent, name);
	fprintf (f, "%s  <threshold>              numeric threshold (default 0.0)\n", indent);
	fprintf (f, "%s  --threshold=<variable>   (T=) threshold from a named variable, e.g. percentile99\n", indent);
	fprintf (f, "%s  --ties:below|above       whether values equal to the threshold count as below (default) or above\n", indent);
	fprintf (f, "%s  --one=<value>  --zero=<value>  (O= Z=) output values (default 1.0 and 0.0)\n", indent);
	}

dspop* op_binarize_parse (char* name, int argc, char** argv)
	{
	dspop_binarize* op = (dspop_binarize*) new_op (name, sizeof(dspop_binarize), false);
	int haveThreshold = false;
	op->oneVal = 1.0;
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if (is_opt3 (arg, "threshold", "T"))                 /* a variable NAME only, logical.c:116-124 */
			{
			if (haveThreshold) { fprintf (stderr, "[%s] threshold specified more than once (at \"%s\")\n", name, arg);  exit (EXIT_FAILURE); }
			op->thresholdVarName = copy_string (argVal);
			haveThreshold = true;
			continue;
			}
		if ((strcmp (arg, "--ties:below") == 0) || (strcmp (arg, "--ties=below") == 0)) { op->tiesAbove = false;  continue; }
		if ((strcmp (arg, "--ties:above") == 0) || (strcmp (arg, "--ties=above") == 0)) { op->tiesAbove = true;   continue; }
		if (is_opt3 (arg, "one", "O"))  { op->oneVal  = string_to_valtype (argVal);  continue; }
		if (is_opt3 (arg, "zero", "Z")) { op->zeroVal = string_to_valtype (argVal);  continue; }
		if (strcmp_prefix (arg, "--") == 0) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if (!haveThreshold) { op->threshold = string_to_valtype (arg);  haveThreshold = true;  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	return (dspop*) op;
	}

void op_binarize_free (dspop* _op)
	{
	dspop_binarize* op = (dspop_binarize*) _op;
	if (op->thresholdVarName != NULL) free (op->thresholdVarName);
	free (op);
	}

/* what binarize will do once its threshold variable exists (the variable's name: NULL when the threshold is a number) */
const char* op_binarize_pending (dspop* _op, int* tiesAbove, valtype* one, valtype* zero)
	{
	dspop_binarize* op = (dspop_binarize*) _op;
	*tiesAbove = op->tiesAbove;  *one = op->oneVal;  *zero = op->zeroVal;
	return op->thresholdVarName;
	}

void op_binarize_describe (dspop* _op, valtype* T, int* tiesAbove, valtype* one, valtype* zero)
	{
	dspop_binarize* op = (dspop_binarize*) _op;
	resolve_variable (_op, &op->thresholdVarName, &op->threshold, "threshold");
	*T = op->threshold;  *tiesAbove = op->tiesAbove;  *one = op->oneVal;  *zero = op->zeroVal;
	}

void op_binarize_apply (dspop* _op, arg_dont_complain(char* vName), u32 vLen, valtype* v)
	{
	dspop_binarize* op = (dspop_binarize*) _op;
	resolve_variable (_op, &op->thresholdVarName, &op->threshold, "threshold");
	check_gdsp (gdsp_binarize (v, vLen, op->threshold, op->tiesAbove, op->oneVal, op->zeroVal, op_stream ()), _op->name);
	}

/* ----------------------------------------------------------- clip, erase ---- */
typedef struct dspop_limits
	{
	dspop   common;
	int     haveMinVal, haveMaxVal, keepInside;
	char   *minValVarName, *maxValVarName;
	valtype minVal, maxVal, zeroVal;
	} dspop_limits;

static dspop* limits_parse (char* name, int argc, char** argv, int isErase)
	{
	dspop_limits* op = (dspop_limits*) new_op (name, sizeof(dspop_limits), false);
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if ((strcmp_prefix (arg, "--min=") == 0) || (strcmp_prefix (arg, "--minimum=") == 0))
			{ value_or_variable (argVal, &op->minVal, &op->minValVarName);  op->haveMinVal = true;  continue; }
		if ((strcmp_prefix (arg, "--max=") == 0) || (strcmp_prefix (arg, "--maximum=") == 0))
			{ value_or_variable (argVal, &op->maxVal, &op->maxValVarName);  op->haveMaxVal = true;  continue; }
		if (isErase && ((strcmp (arg, "--keep:inside") == 0)  || (strcmp (arg, "--keep=inside") == 0)))  { op->keepInside = true;   continue; }
		if (isErase && ((strcmp (arg, "--keep:outside") == 0) || (strcmp (arg, "--keep=outside") == 0))) { op->keepInside = false;  continue; }
		if (isErase && is_opt3 (arg, "zero", "Z")) { op->zeroVal = string_to_valtype (argVal);  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	if (!op->haveMinVal && !op->haveMaxVal) chastise ("[%s] neither minimum nor maximum limit was provided\n", name);
	return (dspop*) op;
	}

static void limits_free (dspop* _op)
	{
	dspop_limits* op = (dspop_limits*) _op;
	if (op->minValVarName != NULL) free (op->minValVarName);
	if (op->maxValVarName != NULL) free (op->maxValVarName);
	free (op);
	}

OP_SHORT (op_clip, "clip the current set of interval values to specified limits")

void op_clip_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sClamp values to a minimum and/or a maximum.\n\n", indent);
	fprintf (f, "%susage: %s [options]\n", indent, name);
	fprintf (f, "%s  --min=<value|variable>   lower limit\n", indent);
	fprintf (f, "%s  --max=<value|variable>   upper limit\n", indent);
	}

dspop* op_clip_parse (char* name, int argc, char** argv) { return limits_parse (name, argc, argv, false); }
void   op_clip_free  (dspop* op) { limits_free (op); }

void op_limits_describe (dspop* _op, int* haveMin, valtype* lo, int* haveMax, valtype* hi, int* keepInside, valtype* zero)
	{
	dspop_limits* op = (dspop_limits*) _op;
	resolve_variable (_op, &op->minValVarName, &op->minVal, "minimum limit");
	resolve_variable (_op, &op->maxValVarName, &op->maxVal, "maximum limit");
	*haveMin = op->haveMinVal;  *lo = op->minVal;  *haveMax = op->haveMaxVal;  *hi = op->maxVal;
	*keepInside = op->keepInside;  *zero = op->zeroVal;
	}

void op_clip_apply (dspop* _op, arg_dont_complain(char* vName), u32 vLen, valtype* v)
	{
	dspop_limits* op = (dspop_limits*) _op;
	resolve_variable (_op, &op->minValVarName, &op->minVal, "minimum limit");
	resolve_variable (_op, &op->maxValVarName, &op->maxVal, "maximum limit");
	check_gdsp (gdsp_clip (v, vLen, op->haveMinVal, op->minVal, op->haveMaxVal, op->maxVal, op_stream ()), _op->name);
	}

OP_SHORT (op_erase, "erase any of the current set of interval values that are outside (or inside) specified limits")

void op_erase_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sReplace values inside (or outside) a range by the zero value.\n\n", indent);
	fprintf (f, "%susage: %s [options]\n", indent, name);
	fprintf (f, "%s  --min=<value|variable>   lower end of the range (inclusive)\n", indent);
	fprintf (f, "%s  --max=<value|variable>   upper end of the range (inclusive)\n", indent);
	fprintf (f, "%s  --keep:outside           erase what is inside the range (default)\n", indent);
	fprintf (f, "%s  --keep:inside            erase what is outside the range\n", indent);
	fprintf (f, "%s  --zero=<value>           (Z=) replacement value (default 0.0)\n", indent);
	}

dspop* op_erase_parse (char* name, int argc, char** argv) { return limits_parse (name, argc, argv, true); }
void   op_erase_free  (dspop* op) { limits_free (op); }

void op_erase_apply (dspop* _op, arg_dont_complain(char* vName), u32 vLen, valtype* v)
	{
	dspop_limits* op = (dspop_limits*) _op;
	resolve_variable (_op, &op->minValVarName, &op->minVal, "minimum limit");
	resolve_variable (_op, &op->maxValVarName, &op->maxVal, "maximum limit");
	check_gdsp (gdsp_erase (v, vLen, op->haveMinVal, op->minVal, op->haveMaxVal, op->maxVal, op->keepInside, op->zeroVal, op_stream ()), _op->name);
	}

/* ------------------------------------------------------------- addconst ---- */
typedef struct dspop_addconst { dspop common;  valtype val; } dspop_addconst;

OP_SHORT (op_add_constant, "add a constant to the current set of interval values")

void op_add_constant_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sAdd a constant to every base.\n\n%susage: %s <value>\n", indent, indent, name);
	}

dspop* op_add_constant_parse (char* name, int argc, char** argv)
	{
	dspop_addconst* op = (dspop_addconst*) new_op (name, sizeof(dspop_addconst), false);
	int haveVal = false;
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		if ((strcmp_prefix (arg, "--") == 0) || haveVal) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		op->val = string_to_valtype (arg);
		haveVal = true;
		}
	if (!haveVal) chastise ("[%s] no constant value was provided\n", name);
	return (dspop*) op;
	}

void op_add_constant_free (dspop* op) { free (op); }

valtype op_add_constant_value (dspop* op) { return ((dspop_addconst*) op)->val; }

void op_add_constant_apply (dspop* _op, arg_dont_complain(char* vName), u32 vLen, valtype* v)
	{ check_gdsp (gdsp_add_constant (v, vLen, ((dspop_addconst*) _op)->val, op_stream ()), _op->name); }

/* ------------------------------------------------------------------ abs ---- */
OP_SHORT (op_absolute_value, "take the absolute value of the current set of interval values")

void op_absolute_value_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sReplace every value by its absolute value.\n\n%susage: %s\n", indent, indent, name);
	}

dspop* op_absolute_value_parse (char* name, int argc, char** argv)
	{
	dspop* op = (dspop*) new_op (name, sizeof(dspop), false);
	if (argc > 0) chastise ("[%s] Can't understand \"%s\"\n", name, argv[0]);
	return op;
	}

void op_absolute_value_free (dspop* op) { free (op); }

void op_absolute_value_apply (dspop* _op, arg_dont_complain(char* vName), u32 vLen, valtype* v)
	{ check_gdsp (gdsp_abs (v, vLen, op_stream ()), _op->name); }

/* --------------------------------------------------------------- invert ---- */
typedef struct dspop_invert { dspop common;  int haveMidVal;  valtype midVal; } dspop_invert;

OP_SHORT (op_invert, "invert the current set of interval values (reflect them about a middle value)")

void op_invert_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sReflect every value about a middle value: v becomes 2*mid - v.\n\n", indent);
	fprintf (f, "%susage: %s [<value>|zero|negate|one|1/2|binary]\n", indent, name);
	fprintf (f, "%s  (default: the middle of the genome-wide minimum and maximum)\n", indent);
	}

dspop* op_invert_parse (char* name, int argc, char** argv)      /* add.c:795-873 */
	{
	dspop_invert* op = (dspop_invert*) new_op (name, sizeof(dspop_invert), true);
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		if ((strcmp_prefix (arg, "--") == 0) || op->haveMidVal) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if      ((strcmp (arg, "zero") == 0) || (strcmp (arg, "negate") == 0)) op->midVal = 0.0;
		else if  (strcmp (arg, "one") == 0)                                    op->midVal = 1.0;
		else if ((strcmp (arg, "1/2") == 0)  || (strcmp (arg, "binary") == 0)) op->midVal = 0.5;
		else op->midVal = string_to_valtype (arg);
		op->haveMidVal = true;
		}
	return (dspop*) op;
	}

void op_invert_free (dspop* op) { free (op); }

/* whole-genome: min and max over every device's chromosomes (one scalar pair per device
 * comes back to the host), then the reflection (add.c:890-939) */
void op_invert_apply (dspop* _op, arg_dont_complain(char* vName), arg_dont_complain(u32 vLen), arg_dont_complain(valtype* v))
	{
	dspop_invert* op = (dspop_invert*) _op;
	valtype mid = op->midVal;
	if (!op->haveMidVal)
		{
		valtype lo, hi;
		genome_extremes (&lo, &hi);                          /* add.c:909-923, over every device */
		mid = (lo + hi) / 2.0;                               /* add.c:925 */
		}
	sigpart* parts;
	int nparts = signal_parts (&parts);
	for (int i=0 ; i<nparts ; i++)                           /* (a stretch is inverted halo and all: the halo stays its neighbours' bases) */
		{
		select_device_of (parts[i].s);
		check_gdsp (gdsp_invert (parts[i].base, parts[i].baseLen, mid, op_stream ()), _op->name);
		}
	}

/* ------------------------------------------------------------ variables ---- */
OP_SHORT (op_show_variables, "show the current named variables")

void op_show_variables_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sPrint the named variables (e.g. those set by percentile) to stderr.\n\n%susage: %s\n", indent, indent, name);
	}

dspop* op_show_variables_parse (char* name, int argc, char** argv)
	{
	dspop* op = (dspop*) new_op (name, sizeof(dspop), true);
	if (argc > 0) chastise ("[%s] Can't understand \"%s\"\n", name, argv[0]);
	return op;
	}

void op_show_variables_free (dspop* op) { free (op); }

void op_show_variables_apply (arg_dont_complain(dspop* op), arg_dont_complain(char* vName), arg_dont_complain(u32 vLen), arg_dont_complain(valtype* v))
	{ fprintf (stderr, "variables:\n");  report_named_globals (stderr, "  "); }          /* variables.c:114-122 */
