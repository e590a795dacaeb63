// ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// In-process harness around the *unmodified* reference sources, compiled from
// where they lie under /root/reference (see oracle/Makefile, target `ref`).
// It exists so that full-precision f64 vectors can be pushed through the
// reference's own op_*_parse / op_*_apply functions (text I/O at "%.*f" cannot
// carry every bit, genodsp_interface.h:24).  Build products go to oracle/_ref/
// only; no reference source text is copied into this repository.
//
// The harness mirrors what the reference's main() does around the operators
// (genodsp.c:835-840 named-global defaults, :848-878 sort + allocate,
// :900-936 pipeline batching loop, :950-971 teardown) but takes chromosomes
// and signals from memory instead of argv/stdin.

#define main genodsp_reference_main
#include "genodsp.c"
#undef main

#include <stdint.h>

static int refh_live = 0;

// forget every chromosome, operator, scratch vector and named global
void refh_reset (void)
	{
	spec   *s, *nextS;
	dspop  *op, *nextOp;

	if (!refh_live) return;

	free_scratch_vectors ();
	free_named_globals ();

	for (s=chromsOfInterest ; s!=NULL ; s=nextS)
		{
		nextS = s->next;
		if (s->chrom     != NULL) free (s->chrom);
		if (s->valVector != NULL) free (s->valVector);
		free (s);
		}
	chromsOfInterest = NULL;
	if (chromsSorted != NULL) free (chromsSorted);
	chromsSorted = NULL;

	for (op=pipeline ; op!=NULL ; op=nextOp)
		{
		nextOp = op->next;
		if (op->name != NULL) free (op->name);
		(*op->funcFree) (op);
		}
	pipeline = tailOp = NULL;
	refh_live = 0;
	}

// declare one chromosome (call before refh_begin); returns 0 on duplicate
int refh_add_chrom (const char* name, uint32_t start, uint32_t length)
	{
	char nameCopy[1001];
	if (!refh_live)
		{
		init_named_globals ();
		set_named_global ("valColumn",     (valtype) valColumn);
		set_named_global ("valPrecision",  (valtype) valPrecision);
		set_named_global ("collapseRuns",  (valtype) collapseRuns);
		set_named_global ("showUncovered", (valtype) showUncovered);
		set_named_global ("originOne",     (valtype) originOne);
		refh_live = 1;
		}
	safe_strncpy (nameCopy, (char*) name, sizeof(nameCopy)-1);
	return add_chromosome_spec (nameCopy, start, length);
	}

// sort chromosomes, set up scratch pool, allocate zeroed vectors
void refh_begin (void)
	{
	u32 maxLength = 0, chromIx;

	sort_chromosomes_by_length ();
	for (chromIx=0 ; chromsSorted[chromIx]!=NULL ; chromIx++)
		{ if (chromsSorted[chromIx]->length > maxLength) maxLength = chromsSorted[chromIx]->length; }
	init_scratch_vectors (maxLength);
	for (chromIx=0 ; chromsSorted[chromIx]!=NULL ; chromIx++)
		{
		spec* s = chromsSorted[chromIx];
		s->valVector = (valtype*) calloc (s->length, sizeof(valtype));
		if (s->valVector == NULL) { fprintf (stderr, "refh: out of memory\n");  exit (EXIT_FAILURE); }
		}
	}

double* refh_vector (const char* name)
	{
	spec* s = find_chromosome_spec ((char*) name);
	return (s == NULL)? NULL : s->valVector;
	}

uint32_t refh_length (const char* name)
	{
	spec* s = find_chromosome_spec ((char*) name);
	return (s == NULL)? 0 : s->length;
	}

// name of the i-th chromosome in the reference's processing order (longest
// first, genodsp.c:1113-1145); NULL past the end
const char* refh_sorted_name (int i)
	{
	int k;
	for (k=0 ; chromsSorted[k]!=NULL ; k++)
		{ if (k == i) return chromsSorted[k]->chrom; }
	return NULL;
	}

// parse "= op args = op args ..." exactly as the reference's command line
// parser would (genodsp.c:317-329 -> :634-723), then run the batching loop
// (genodsp.c:900-936), then free the operators.
void refh_run (int argc, char** argv)
	{
	dspop  *firstOp, *stopOp, *op, *nextOp;
	u32     chromIx, maxLength = 0;
	int     consumed;

	while (argc > 0)
		{
		if (argv[0][0] != specialPipeChar)
			{ fprintf (stderr, "refh_run: expected '=' token, got \"%s\"\n", argv[0]);  exit (EXIT_FAILURE); }
		consumed = process_operator_options (argc, argv);
		argv += consumed;  argc -= consumed;
		}

	for (chromIx=0 ; chromsSorted[chromIx]!=NULL ; chromIx++)
		{ if (chromsSorted[chromIx]->length > maxLength) maxLength = chromsSorted[chromIx]->length; }

	firstOp = pipeline;
	while (firstOp != NULL)
		{
		for (stopOp=firstOp ; stopOp!=NULL ; stopOp=stopOp->next)
			{ if (stopOp->atRandom) break; }

		if (stopOp != firstOp)
			{
			for (chromIx=0 ; chromsSorted[chromIx]!=NULL ; chromIx++)
				{
				spec* s = chromsSorted[chromIx];
				for (op=firstOp ; op!=stopOp ; op=op->next)
					(*op->funcApply) (op, s->chrom, s->length, s->valVector);
				}
			}

		if (stopOp == NULL) firstOp = NULL;
		else
			{
			(*stopOp->funcApply) (stopOp, "*", maxLength, NULL);
			firstOp = stopOp->next;
			}
		}

	for (op=pipeline ; op!=NULL ; op=nextOp)
		{
		nextOp = op->next;
		if (op->name != NULL) free (op->name);
		(*op->funcFree) (op);
		}
	pipeline = tailOp = NULL;
	}

int refh_get_global (const char* name, double* val)
	{ return named_global_exists ((char*) name, val); }

void refh_set_global (const char* name, double val)
	{ set_named_global ((char*) name, val); }

// interval ingest from a text file through the reference's read_intervals
// (genodsp.c:1187-1350)
int refh_read_intervals_file (const char* path, int valCol, int origin1,
                              int overlapOp, int clear, double missingVal)
	{
	FILE* f = fopen (path, "rt");
	if (f == NULL) return 0;
	read_intervals (f, valCol, origin1, overlapOp, clear, missingVal);
	fclose (f);
	return 1;
	}

// interval report to a text file through the reference's report_intervals
// (genodsp.c:1561-1691)
int refh_report_file (const char* path, int precision, int noValues,
                      int collapse, int uncovered, int origin1)
	{
	FILE* f = fopen (path, "wt");
	if (f == NULL) return 0;
	report_intervals (f, precision, noValues, collapse, uncovered, origin1);
	fclose (f);
	return 1;
	}
