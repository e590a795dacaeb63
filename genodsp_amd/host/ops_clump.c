/* ops_clump.c -- clump, anticlump (device shims).
 * Argument rules: clump.c:127-252 (op_clump_parse), :353-485 (parse_min_length); the per-chromosome
 * length rule is clump.c:512-518.  The search itself is gdsp_clump (gdsp_clump.hip). */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

typedef struct dspop_clump
	{
	dspop   common;
	int     above;                  /* clump: average at least T; anticlump: at most T */
	char*   averageVarName;
	valtype average;
	u32     minLength;
	double  relativeLength;         /* > 0: fraction of the chromosome length */
	valtype oneVal, zeroVal;
	} dspop_clump;

static void bad_length (char* name, char* arg, const char* why)
	{ fprintf (stderr, "[%s] %s (at \"%s\")\n", name, why, arg);  exit (EXIT_FAILURE); }

static double fraction_of (char* name, char* arg, char* text)
	{
	double f = string_to_double (text);
	if (f <= 0.0) bad_length (name, arg, "relative length has to be positive");
	if (f >  1.0) bad_length (name, arg, "relative length can't be more than 1");
	return f;
	}

/* <n> | CL | CL*<f> | <f>*CL | CL/<k> | max(<one of those>,<n>) */
static void length_arg (char* name, char* arg, char* text, u32* minLength, double* relative, int maxOk)
	{
	if (maxOk && (strcmp_prefix (text, "max(") == 0) && (strcmp_suffix (text, ")") == 0))
		{
		char* a = copy_string (text + 4);
		a[strlen (a) - 1] = 0;
		char* b = strchr (a, ',');
		if (b == NULL) bad_length (name, arg, "can't parse relative length");
		*(b++) = 0;
		u32 lenA = 0, lenB = 0;  double relA = 0.0, relB = 0.0;
		length_arg (name, arg, a, &lenA, &relA, false);
		length_arg (name, arg, b, &lenB, &relB, false);
		if ((relA > 0) == (relB > 0)) bad_length (name, arg, "can't parse relative length");
		*relative  = (relA > 0)? relA : relB;
		*minLength = (relA > 0)? lenB : lenA;
		free (a);
		return;
		}
	*minLength = 0;
	if (strcmp (text, "CL") == 0) { *relative = 1.0;  return; }
	if (strcmp_prefix (text, "CL*") == 0) { *relative = fraction_of (name, arg, text + 3);  return; }
	if (strcmp_suffix (text, "*CL") == 0)
		{
		char* t = copy_string (text);
		t[strlen (t) - 3] = 0;
		*relative = fraction_of (name, arg, t);
		free (t);
		return;
		}
	if (strcmp_prefix (text, "CL/") == 0)
		{
		double k = string_to_double (text + 3);
		if (k < 0.0) bad_length (name, arg, "relative length has to be positive");
		if (k < 1.0) bad_length (name, arg, "relative length can't be more than 1");
		*relative = 1.0 / k;
		return;
		}
	int n = string_to_unitized_int (text, /*thousands*/ true);
	if (n == 0) chastise ("[%s] minimum length can't be zero (\"%s\")\n", name, arg);
	if (n < 0)  chastise ("[%s] minimum length can't be negative (\"%s\")\n", name, arg);
	*minLength = (u32) n;
	*relative  = 0.0;
	}

static dspop* clump_parse (char* name, int argc, char** argv, int above)
	{
	dspop_clump* op = (dspop_clump*) new_op (name, sizeof(dspop_clump), false);
	int haveAverage = false;
	op->above     = above;
	op->minLength = 100;
	op->oneVal    = 1.0;
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if (is_opt3 (arg, "average", "T"))
			{
			if (haveAverage)
				{ fprintf (stderr, "[%s] average threshold specified more than once (at \"%s\")\n", name, arg);  exit (EXIT_FAILURE); }
			op->averageVarName = copy_string (argVal);
			haveAverage = true;
			continue;
			}
		if (is_opt3 (arg, "length", "L")) { length_arg (name, arg, argVal, &op->minLength, &op->relativeLength, true);  continue; }
		if (is_opt3 (arg, "one", "O"))    { op->oneVal  = string_to_valtype (argVal);  continue; }
		if (is_opt3 (arg, "zero", "Z"))   { op->zeroVal = string_to_valtype (argVal);  continue; }
		if ((strcmp (arg, "--debug") == 0) || (strcmp (arg, "--debug=detail") == 0)) continue;
		if (strcmp_prefix (arg, "--progress=") == 0) continue;
		if (strcmp_prefix (arg, "--") == 0) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if (!haveAverage) { op->average = string_to_valtype (arg);  haveAverage = true;  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	return (dspop*) op;
	}

static void clump_free (dspop* _op)
	{
	dspop_clump* op = (dspop_clump*) _op;
	if (op->averageVarName != NULL) free (op->averageVarName);
	free (op);
	}

static void clump_apply (dspop* _op, char* vName, u32 vLen, valtype* v)
	{
	dspop_clump* op = (dspop_clump*) _op;
	u32 minLength = op->minLength;
	if (op->relativeLength > 0.0)
		{
		u32 relLength = (u32) (op->relativeLength * vLen);
		if (relLength > minLength) minLength = relLength;
		}
	resolve_variable (_op, &op->averageVarName, &op->average, "threshold");
	void* work = device_workspace (gdsp_clump_work (vLen));
	check_gdsp (gdsp_clump (v, vLen, op->average, minLength, op->above, op->oneVal, op->zeroVal, work, op_stream ()), _op->name);
	}

static void clump_usage (char* name, FILE* f, char* indent, int above)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sFind intervals at least L long whose average is at %s a threshold T; bases in such\n", indent, above? "least" : "most");
	fprintf (f, "%sintervals become 1, other bases 0.  The ends of each interval are then trimmed of signal\n", indent);
	fprintf (f, "%s%s than T, so the reported runs may be shorter than L.\n", indent, above? "lower" : "higher");
	fprintf (f, "%s\n%susage: %s [<average>] [options]\n", indent, indent, name);
	fprintf (f, "%s  <average>                the threshold (default is 0.0)\n", indent);
	fprintf (f, "%s  --average=<variable>     (T=) get the threshold from a named variable\n", indent);
	fprintf (f, "%s  --length=<length>        (L=) minimum length of a qualifying interval (default is 100);\n", indent);
	fprintf (f, "%s                           CL, <scale>*CL, CL/<n> and max(<relative>,<n>) are relative to\n", indent);
	fprintf (f, "%s                           the chromosome length\n", indent);
	fprintf (f, "%s  --one=<value>  --zero=<value>  (O= Z=) output values (default 1.0 and 0.0)\n", indent);
	}

OP_SHORT (op_clump, "find intervals with an average above some threshold")
void   op_clump_usage (char* name, FILE* f, char* indent) { clump_usage (name, f, indent, true); }
dspop* op_clump_parse (char* name, int argc, char** argv) { return clump_parse (name, argc, argv, true); }
void   op_clump_free  (dspop* op) { clump_free (op); }
void   op_clump_apply (dspop* op, char* vName, u32 vLen, valtype* v) { clump_apply (op, vName, vLen, v); }

OP_SHORT (op_skimp, "find intervals with an average below some threshold")
void   op_skimp_usage (char* name, FILE* f, char* indent) { clump_usage (name, f, indent, false); }
dspop* op_skimp_parse (char* name, int argc, char** argv) { return clump_parse (name, argc, argv, false); }
void   op_skimp_free  (dspop* op) { clump_free (op); }
void   op_skimp_apply (dspop* op, char* vName, u32 vLen, valtype* v) { clump_apply (op, vName, vLen, v); }
