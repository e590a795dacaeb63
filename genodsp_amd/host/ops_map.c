/* ops_map.c -- map: piecewise-linear value mapping read from a two-column file (device shim).
 * Argument rules and file format: map.c:60-190 (parse), :414-540 (read_mapping) in the reference. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

typedef struct dspop_map { dspop common;  char* filename;  int destroyFile; } dspop_map;
typedef struct knot { valtype vIn, vOut; } knot;

OP_SHORT (op_map, "map the current set of interval values to new values")

void op_map_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sMap every value through a piecewise-linear function given as a file of\n", indent);
	fprintf (f, "%s\"<in> <out>\" pairs; values beyond the ends take the end pairs' outputs.\n\n", indent);
	fprintf (f, "%susage: %s <filename> [options]\n", indent, name);
	fprintf (f, "%s  --destroy                delete the file after reading it\n", indent);
	}

dspop* op_map_parse (char* name, int argc, char** argv)
	{
	dspop_map* op = (dspop_map*) new_op (name, sizeof(dspop_map), false);
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		if (strcmp (arg, "--destroy") == 0) { op->destroyFile = true;  continue; }
		if (strcmp (arg, "--debug") == 0) continue;
		if (strcmp_prefix (arg, "--") == 0) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if (op->filename == NULL) { op->filename = copy_string (arg);  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	if (op->filename == NULL) { fprintf (stderr, "[%s] no filename was provided\n", name);  exit (EXIT_FAILURE); }
	return (dspop*) op;
	}

void op_map_free (dspop* _op)
	{
	dspop_map* op = (dspop_map*) _op;
	if (op->filename != NULL) free (op->filename);
	free (op);
	}

static int knot_ascending (const void* a, const void* b)       /* map.c:386-392 */
	{
	valtype x = ((const knot*) a)->vIn, y = ((const knot*) b)->vIn;
	return (x > y) - (x < y);
	}

/* The reference re-reads the file for every chromosome (map.c:209-213); the table is the
 * same each time, so it is read once, sorted the same way, and kept on each device. */
void op_map_apply (dspop* _op, arg_dont_complain(char* vName), u32 vLen, valtype* v)
	{
	dspop_map* op = (dspop_map*) _op;
	static dspop_map* loadedFor = NULL;
	static valtype*   d_in[64], *d_out[64];
	static u32        nknots = 0;
	static valtype   *h_in = NULL, *h_out = NULL;

	if (loadedFor != op)
		{
		char  line[1000];
		knot* k = NULL;
		u32   cap = 0, lineNumber = 0;
		FILE* f = fopen (op->filename, "rt");
		if (f == NULL) { fprintf (stderr, "[%s] can't open \"%s\" for reading\n", _op->name, op->filename);  exit (EXIT_FAILURE); }
		nknots = 0;
		while (fgets (line, sizeof(line), f) != NULL)
			{
			lineNumber++;
			char* scan = skip_whitespace (line);
			if ((*scan == 0) || (*scan == '#')) continue;
			if (line[0] == ' ') { fprintf (stderr, "problem at line %u, line contains no first value\n", lineNumber);  exit (EXIT_FAILURE); }
			char* field = line;
			char* mark = skip_darkspace (field);
			scan = skip_whitespace (mark);
			if (*mark != 0) *mark = 0;
			if (*scan == 0) { fprintf (stderr, "problem at line %u, line contains no second value\n", lineNumber);  exit (EXIT_FAILURE); }
			char* field2 = scan;
			mark = skip_darkspace (scan);
			if (*mark != 0) *mark = 0;
			if (nknots == cap) { cap = cap? 2*cap : 64;  k = (knot*) realloc (k, cap * sizeof(knot)); }
			k[nknots].vIn  = string_to_valtype (field);
			k[nknots].vOut = string_to_valtype (field2);
			nknots++;
			}
		fclose (f);
		if (nknots == 0) { fprintf (stderr, "[%s] problem with mapping file \"%s\"\n", _op->name, op->filename);  exit (EXIT_FAILURE); }
		if (op->destroyFile) remove (op->filename);
		qsort (k, nknots, sizeof(knot), knot_ascending);
		free (h_in);  free (h_out);
		h_in  = (valtype*) malloc (nknots * sizeof(valtype));
		h_out = (valtype*) malloc (nknots * sizeof(valtype));
		for (u32 i=0 ; i<nknots ; i++) { h_in[i] = k[i].vIn;  h_out[i] = k[i].vOut; }
		free (k);
		for (int d=0 ; d<64 ; d++) { d_in[d] = NULL;  d_out[d] = NULL; }
		loadedFor = op;
		}

	spec* s = find_chromosome_spec (vName);
	int   d = (s != NULL)? device_index_of (s) : 0;
	if (d_in[d] == NULL)
		{
		check_gdsp (gdsp_malloc ((void**) &d_in[d],  nknots * sizeof(valtype)), _op->name);
		check_gdsp (gdsp_malloc ((void**) &d_out[d], nknots * sizeof(valtype)), _op->name);
		check_gdsp (gdsp_memcpy_h2d (d_in[d],  h_in,  nknots * sizeof(valtype), op_stream ()), _op->name);
		check_gdsp (gdsp_memcpy_h2d (d_out[d], h_out, nknots * sizeof(valtype), op_stream ()), _op->name);
		check_gdsp (gdsp_stream_sync (op_stream ()), _op->name);
		}
	check_gdsp (gdsp_map (v, vLen, d_in[d], d_out[d], nknots, op_stream ()), _op->name);
	}
