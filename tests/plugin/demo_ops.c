// demo_ops.c-- a small operator group written in the reference's style, against the
// reference's names only (utilities.h + genodsp_interface.h); test asset, see demo_ops.h.
//
// What differs from an operator of the reference is the one thing DESIGN.md says must: v is a
// device pointer, so the per-base work is a call into genodsp_hip.h instead of a loop.

#include <stdlib.h>
#define  true  1
#define  false 0
#include <stdio.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>
#include <float.h>
#include "utilities.h"
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "demo_ops.h"

//----------
// op_demo_lift-- v = |v| + amount, amount a number or the name of a variable
//----------

typedef struct dspop_demolift
	{
	dspop		common;			// common elements shared with all operators
	char*		amountVarName;
	valtype		amount;
	} dspop_demolift;

void op_demo_lift_short (char* name, int nameWidth, FILE* f, char* indent)
	{
	int nameFill = nameWidth-2 - strlen(name);
	if (indent == NULL) indent = "";
	fprintf (f, "%s%s:%*s", indent, name, nameFill+1, " ");
	fprintf (f, "absolute value plus a constant\n");
	}

void op_demo_lift_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%susage: %s <amount|variable>\n", indent, name);
	}

dspop* op_demo_lift_parse (char* name, int _argc, char** _argv)
	{
	dspop_demolift*	op;
	int				argc = _argc;
	char**			argv = _argv;
	int				haveAmount = false;

	op = (dspop_demolift*) malloc (sizeof(dspop_demolift));
	if (op == NULL) goto cant_allocate;
	op->common.atRandom = false;
	op->amountVarName   = NULL;
	op->amount          = get_named_global ("demoAmount", 1.0);

	while (argc > 0)
		{
		char* arg = argv[0];
		if (strcmp_prefix (arg, "--") == 0)
			chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if (haveAmount)
			chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if (!try_string_to_valtype (arg, &op->amount))
			op->amountVarName = copy_string (arg);
		haveAmount = true;
		argv++;  argc--;
		}
	return (dspop*) op;

cant_allocate:
	fprintf (stderr, "[%s] failed to allocate control record (%d bytes)\n", name, (int) sizeof(dspop_demolift));
	exit (EXIT_FAILURE);
	return NULL;
	}

void op_demo_lift_free (dspop* _op)
	{
	dspop_demolift*	op = (dspop_demolift*) _op;
	if (op->amountVarName != NULL) free (op->amountVarName);
	free (op);
	}

void op_demo_lift_apply (dspop* _op, char* vName, u32 vLen, valtype* v)
	{
	dspop_demolift*	op = (dspop_demolift*) _op;
	valtype*		keep   = get_scratch_vector ();
	s32*			marks  = get_scratch_ints ();
	s32*			marks2 = get_scratch_ints ();
	s32				seen[2];
	char			varName[200];

	if (op->amountVarName != NULL)
		{
		if (!named_global_exists (op->amountVarName, &op->amount))
			{
			fprintf (stderr, "[%s] attempt to use %s failed (no such variable)\n", op->common.name, op->amountVarName);
			exit (EXIT_FAILURE);
			}
		free (op->amountVarName);
		op->amountVarName = NULL;
		}

	// the scratch vectors are device memory of (at least) vLen entries: the signal survives a
	// trip through one, and the last of vLen ints can be written and read back
	if (marks == marks2)
		{ fprintf (stderr, "[%s] the same scratch ints were handed out twice\n", op->common.name);  exit (EXIT_FAILURE); }
	check_gdsp (gdsp_memcpy_d2d (keep, v, vLen * sizeof(valtype), op_stream ()), "copy to scratch");
	check_gdsp (gdsp_fill (v, vLen, -1.0, op_stream ()), "fill");
	check_gdsp (gdsp_memcpy_d2d (v, keep, vLen * sizeof(valtype), op_stream ()), "copy from scratch");
	check_gdsp (gdsp_memset (marks, 0xFF, vLen * sizeof(s32), op_stream ()), "mark");
	check_gdsp (gdsp_memcpy_d2h (&seen[0], &marks[vLen-1], sizeof(s32), op_stream ()), "fetch mark");
	check_gdsp (gdsp_memcpy_d2h (&seen[1], &marks2[vLen-1], sizeof(s32), op_stream ()), "fetch mark");
	check_gdsp (gdsp_stream_sync (op_stream ()), "synchronise");
	if ((seen[0] != -1) || (seen[1] != 0))
		{ fprintf (stderr, "[%s] scratch ints read back %d and %d\n", op->common.name, seen[0], seen[1]);  exit (EXIT_FAILURE); }
	check_gdsp (gdsp_memset (marks, 0, vLen * sizeof(s32), op_stream ()), "unmark");
	release_scratch_ints (marks2);
	release_scratch_ints (marks);
	release_scratch_vector (keep);

	check_gdsp (gdsp_abs (v, vLen, op_stream ()), "abs");
	check_gdsp (gdsp_add_constant (v, vLen, op->amount, op_stream ()), "add");

	sprintf (varName, "demoLength_%s", vName);
	set_named_global (varName, (valtype) find_chromosome_spec (vName)->length);
	}

//----------
// op_demo_snapshot-- write the genome to a file and read it back (whole-genome operator)
//----------

typedef struct dspop_demosnapshot
	{
	dspop		common;			// common elements shared with all operators
	char*		filename;
	} dspop_demosnapshot;

void op_demo_snapshot_short (char* name, int nameWidth, FILE* f, char* indent)
	{
	int nameFill = nameWidth-2 - strlen(name);
	if (indent == NULL) indent = "";
	fprintf (f, "%s%s:%*s", indent, name, nameFill+1, " ");
	fprintf (f, "save the signal to a file and restore it\n");
	}

void op_demo_snapshot_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%susage: %s <filename>\n", indent, name);
	}

dspop* op_demo_snapshot_parse (char* name, int argc, char** argv)
	{
	dspop_demosnapshot*	op;

	if (argc != 1) chastise ("[%s] needs exactly one filename\n", name);
	op = (dspop_demosnapshot*) malloc (sizeof(dspop_demosnapshot));
	if (op == NULL) { fprintf (stderr, "[%s] failed to allocate control record\n", name);  exit (EXIT_FAILURE); }
	op->common.atRandom = true;
	op->filename        = copy_string (argv[0]);
	return (dspop*) op;
	}

void op_demo_snapshot_free (dspop* _op)
	{
	dspop_demosnapshot*	op = (dspop_demosnapshot*) _op;
	free (op->filename);
	free (op);
	}

void op_demo_snapshot_apply
   (dspop*						_op,
	arg_dont_complain(char*		vName),
	arg_dont_complain(u32		vLen),
	arg_dont_complain(valtype*	v))
	{
	dspop_demosnapshot*	op = (dspop_demosnapshot*) _op;
	valtype				lengths[100];
	u32					n = 0, chromIx;

	for (chromIx=0 ; (chromsSorted[chromIx]!=NULL)&&(n<100) ; chromIx++)
		lengths[n++] = (valtype) chromsSorted[chromIx]->length;
	qsort (lengths, n, sizeof(valtype), valtype_ascending);
	set_named_global ("demoShortest", lengths[0]);
	set_named_global ("demoLongest",  lengths[n-1]);

	tracking_report ("%s(%s)\n", op->common.name, op->filename);
	write_all_chromosomes (op->filename);
	for (chromIx=0 ; chromsSorted[chromIx]!=NULL ; chromIx++)
		{
		select_device_of (chromsSorted[chromIx]);
		check_gdsp (gdsp_fill (chromsSorted[chromIx]->valVector, chromsSorted[chromIx]->length, 123.0, op_stream ()), "scribble");
		}
	read_all_chromosomes (op->filename);
	}
