/* ops_common.c -- helpers shared by the operator files. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "genodsp_interface.h"
#include "utilities.h"
#include "host_services.h"

void* new_op (char* name, size_t bytes, int atRandom)
	{
	dspop* op = (dspop*) calloc (1, bytes);
	if (op == NULL) { fprintf (stderr, "[%s] failed to allocate control record (%d bytes)\n", name, (int) bytes);  exit (EXIT_FAILURE); }
	op->atRandom = atRandom;
	return op;
	}

/* --window=<n> and friends: zero and negatives are errors, 1-2 are raised to 3 with a
 * warning (sum.c:121-132, minmax.c:1116-1126) */
u32 window_arg (char* name, char* arg, char* argVal, const char* what)
	{
	int w = string_to_unitized_int (argVal, /*thousands*/ true);
	if (w == 0) chastise ("[%s] %s can't be zero (\"%s\")\n", name, what, arg);
	if (w < 0)  chastise ("[%s] %s can't be negative (\"%s\")\n", name, what, arg);
	if (w < 3) { fprintf (stderr, "[%s] WARNING: raising %s from %d to %d\n", name, what, w, 3);  w = 3; }
	return (u32) w;
	}

/* morphology.c:137-140, mask.c: a value that does not parse as a number is the name of
 * a variable some earlier operator (percentile) will have set by the time we run */
void value_or_variable (char* argVal, valtype* val, char** varName)
	{
	if (!try_string_to_valtype (argVal, val)) *varName = copy_string (argVal);
	}

/* logical.c:234-243: fetch the variable at first apply, say so, then forget the name */
void resolve_variable (dspop* op, char** varName, valtype* val, const char* role)
	{
	if (*varName == NULL) return;
	if (!named_global_exists (*varName, val))
		{
		fprintf (stderr, "[%s] attempt to use %s as %s failed (no such variable)\n", op->name, *varName, role);
		exit (EXIT_FAILURE);
		}
	fprintf (stderr, "[%s] using %s = " valtypeFmt " as %s\n", op->name, *varName, *val, role);
	free (*varName);
	*varName = NULL;
	}
