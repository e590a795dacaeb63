/* ops_intervals.c -- add, subtract, multiply, divide, input, output (device shims).
 *
 * Argument rules: add.c:80-175 / :373-468, multiply.c:80-175 / :465-568, opio.c:88-205 /
 * :320-435 in the reference.  The file is read on the host with the driver's
 * read_interval (same text rules as the main ingest); intervals are routed to their
 * chromosome and handed to the device in file order (host_services.h, ib_*), where
 * gdsp_apply_intervals / gdsp_scale_intervals do the per-base work. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <float.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

enum { K_ADD, K_SUBTRACT, K_MULTIPLY, K_DIVIDE, K_MASK, K_MASKNOT, K_OR, K_AND, K_MINWITH, K_MAXWITH, K_MINOVER, K_MAXOVER };

typedef struct dspop_fileop
	{
	dspop   common;
	int     kind;
	char*   filename;
	int     valColumn, originOne, destroyFile;
	valtype infinityVal;
	valtype maskVal;                /* mask, masknot */
	char*   maskValVarName;
	int     haveMaskVal;
	} dspop_fileop;

static dspop* fileop_parse (char* name, int argc, char** argv, int kind)
	{
	dspop_fileop* op = (dspop_fileop*) new_op (name, sizeof(dspop_fileop), true);
	op->kind        = kind;
	/* mask, masknot, minover and maxover read no value column and never ask for it (mask.c:92,400 minmax.c:102,503) */
	const int noValues = (kind == K_MASK) || (kind == K_MASKNOT) || (kind == K_MINOVER) || (kind == K_MAXOVER);
	op->valColumn   = noValues? -1 : (int) get_named_global ("valColumn", 4-1);
	op->originOne   = (int) get_named_global ("originOne", false);
	op->infinityVal = valtypeMax;
	if (kind == K_MINOVER) op->maskVal = valtypeMax;     /* minmax.c:103; maxover's zero value is 0.0 (:504) */
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		const int isOver = (kind == K_MINOVER) || (kind == K_MAXOVER);
		const int isMask = (kind == K_MASK) || (kind == K_MASKNOT) || isOver;     /* no value column either */
		const int isWith = (kind == K_MINWITH) || (kind == K_MAXWITH);
		if (!isMask && !isWith
		 && ((strcmp (arg, "--novalue") == 0) || (strcmp (arg, "--novalues") == 0) || (strcmp (arg, "--value=none") == 0)))
			{ op->valColumn = -1;  continue; }
		if (!isMask && (strcmp_prefix (arg, "--value=") == 0))
			{
			int col = string_to_int (argVal) - 1;
			if (col == -1) chastise ("[%s] value column can't be 0 (\"%s\")\n", name, arg);
			if (col < 0)   chastise ("[%s] value column can't be negative (\"%s\")\n", name, arg);
			if (col < 3)   chastise ("[%s] value column can't be 1, 2 or 3 (\"%s\")\n", name, arg);
			op->valColumn = col;
			continue;
			}
		if ((kind == K_MINOVER) && (strcmp_prefix (arg, "--infinity=") == 0)) { op->maskVal = string_to_valtype (argVal);  continue; }
		if ((kind == K_MAXOVER) && is_opt3 (arg, "zero", "Z"))                { op->maskVal = string_to_valtype (argVal);  continue; }
		if (isMask && !isOver && is_opt3 (arg, "mask", "M"))
			{
			if ((kind == K_MASK) && op->haveMaskVal)
				{ fprintf (stderr, "[%s] mask value specified more than once (at \"%s\")\n", name, arg);  exit (EXIT_FAILURE); }
			if (kind == K_MASK) value_or_variable (argVal, &op->maskVal, &op->maskValVarName);   /* mask.c:104-116 */
			else                op->maskVal = string_to_valtype (argVal);                          /* mask.c:412-418 */
			op->haveMaskVal = true;
			continue;
			}
		if ((strcmp (arg, "--origin=one") == 0)  || (strcmp (arg, "--origin=1") == 0)) { op->originOne = true;   continue; }
		if ((strcmp (arg, "--origin=zero") == 0) || (strcmp (arg, "--origin=0") == 0)) { op->originOne = false;  continue; }
		if (((kind == K_ADD) || (kind == K_SUBTRACT) || isWith) && (strcmp (arg, "--destroy") == 0)) { op->destroyFile = true;  continue; }
		if ((kind == K_DIVIDE) && (strcmp_prefix (arg, "--infinity=") == 0)) { op->infinityVal = string_to_valtype (argVal);  continue; }
		if (strcmp (arg, "--debug") == 0) continue;
		if (strcmp_prefix (arg, "--") == 0) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if (op->filename == NULL) { op->filename = copy_string (arg);  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	if (op->filename == NULL) { fprintf (stderr, "[%s] no filename was provided\n", name);  exit (EXIT_FAILURE); }
	return (dspop*) op;
	}

static void fileop_free (dspop* _op)
	{
	dspop_fileop* op = (dspop_fileop*) _op;
	if (op->filename != NULL) free (op->filename);
	if (op->maskValVarName != NULL) free (op->maskValVarName);
	free (op);
	}

/* add.c:191-306, :484-599, multiply.c:193-393, :586-787, mask.c:187-300, :483-640,
 * logical.c:439-560, :737-880, minmax.c:1893-2010, :2179-2294 */
static void fileop_apply (dspop* _op)
	{
	dspop_fileop* op = (dspop_fileop*) _op;
	char    line[1001], prevChrom[1001];
	char*   chrom;
	spec*   s = NULL;
	u32     start, end, o = op->originOne? 1 : 0, prevEnd = 0;
	valtype val;
	/* "sorted" kinds say what happens to bases under NO interval, so their intervals must be
	 * sorted, non-overlapping and grouped by chromosome, and every chromosome is visited */
	int     scaling  = (op->kind == K_MULTIPLY) || (op->kind == K_DIVIDE) || (op->kind == K_MASKNOT) || (op->kind == K_AND)
	                || (op->kind == K_MINOVER) || (op->kind == K_MAXOVER);
	int     skipZero = (op->kind != K_MASK) && (op->kind != K_MASKNOT) && (op->kind != K_MINWITH) && (op->kind != K_MAXWITH);
	int     valCol   = ((op->kind == K_MASK) || (op->kind == K_MASKNOT) || (op->kind == K_MINOVER) || (op->kind == K_MAXOVER))
	                   ? -1 : op->valColumn;                                             /* mask.c:237, minmax.c:227 */
	resolve_variable (_op, &op->maskValVarName, &op->maskVal, "mask value");

	FILE* f = fopen (op->filename, "rt");
	if (f == NULL) { fprintf (stderr, "[%s] can't open \"%s\" for reading\n", _op->name, op->filename);  exit (EXIT_FAILURE); }
	for (int i=0 ; chromsSorted[i]!=NULL ; i++) chromsSorted[i]->flag = false;

	ib_begin ();
	prevChrom[0] = 0;
	while (read_interval (f, line, sizeof(line), valCol, &chrom, &start, &end, &val))
		{
		if (skipZero && (val == 0.0)) continue;            /* add.c:235, multiply.c:236, logical.c:489 */
		if (strcmp (chrom, prevChrom) != 0)
			{
			s = find_chromosome_spec (chrom);
			if (scaling && (s != NULL))
				{
				prevEnd = 0;
				if (s->flag)
					{
					fprintf (stderr, "[%s] in \"%s\", not all intervals on %s are together (%d..%d begins new group)\n",
					                 _op->name, op->filename, chrom, start, end);
					exit (EXIT_FAILURE);
					}
				}
			safe_strncpy (prevChrom, chrom, sizeof(prevChrom)-1);
			}
		if (s == NULL) continue;
		if (!s->flag) { if (trackOperations) fprintf (stderr, "%s(%s)\n", _op->name, chrom);  s->flag = true; }

		start -= o;
		u32 adjStart = start, adjEnd = end;
		if (s->start == 0)
			{
			if (end > s->length)
				{
				fprintf (stderr, "[%s] in \"%s\", %s %d %d is beyond the end of the chromosome (L=%d)\n",
				                 _op->name, op->filename, chrom, start, end, s->length);
				exit (EXIT_FAILURE);
				}
			}
		else
			{
			if (end <= s->start) continue;
			adjEnd   = end - s->start;
			adjStart = (start <= s->start)? 0 : start - s->start;
			if (adjStart >= s->length) continue;
			if (adjEnd   >= s->length) adjEnd = s->length;
			}
		if (scaling)
			{
			if (adjStart < prevEnd)
				{
				fprintf (stderr, "[%s] in \"%s\", intervals on %s are not sorted (%d..%d after %d)\n",
				                 _op->name, op->filename, chrom, start, end, s->start + prevEnd);
				exit (EXIT_FAILURE);
				}
			prevEnd = adjEnd;
			}
		if      (op->kind == K_SUBTRACT) val = -val;
		else if (op->kind == K_MASK)     val = op->maskVal;
		else if (op->kind == K_OR)       val = 1.0;
		ib_add (s, adjStart, adjEnd, val);
		if ((op->kind <= K_SUBTRACT) && (ib_pending () >= ib_batch_limit ())) ib_flush_apply (ri_overlapSum, 0, 0.0, false);
		}
	fclose (f);

	if (scaling)
		{
		if (trackOperations)
			for (int i=0 ; chromsSorted[i]!=NULL ; i++)
				{ if (!chromsSorted[i]->flag) fprintf (stderr, "%s(%s,absent)\n", _op->name, chromsSorted[i]->chrom); }
		if      (op->kind == K_MINOVER) ib_flush_over (false, op->maskVal);
		else if (op->kind == K_MAXOVER) ib_flush_over (true,  op->maskVal);
		else if (op->kind == K_MASKNOT) ib_flush_mask (false, op->maskVal, false);
		else if (op->kind == K_AND)     ib_flush_mask (false, 0.0, true);
		else                            ib_flush_scale (op->kind == K_DIVIDE, op->infinityVal);
		}
	else
		{
		if      (op->kind == K_MASK)    ib_flush_mask (true, 0.0, false);
		else if (op->kind == K_OR)      ib_flush_mask (true, 0.0, true);
		else if (op->kind == K_MINWITH) ib_flush_apply (ri_overlapMin, 0, 0.0, false);
		else if (op->kind == K_MAXWITH) ib_flush_apply (ri_overlapMax, 0, 0.0, false);
		else                            ib_flush_apply (ri_overlapSum, 0, 0.0, false);
		if (op->destroyFile) remove (op->filename);
		}
	}

static void fileop_usage (char* name, FILE* f, char* indent, int kind)
	{
	static const char* what[] =
		{ "Add the values of the intervals in a file to the signal.",
		  "Subtract the values of the intervals in a file from the signal.",
		  "Multiply the signal by the values of the (sorted, non-overlapping) intervals in a\nfile; bases under no interval become zero.",
		  "Divide the signal by the values of the (sorted, non-overlapping) intervals in a\nfile; bases under no interval become +/- infinity.",
		  "Set the signal to the mask value under the intervals in a file.",
		  "Set the signal to the mask value everywhere EXCEPT under the (sorted, non-overlapping)\nintervals in a file.",
		  "Binarise the signal (non-zero -> 1), then OR it with the intervals in a file.",
		  "Binarise the signal (non-zero -> 1), then AND it with the (sorted, non-overlapping)\nintervals in a file.",
		  "Replace each base by the minimum of itself and the values of the intervals covering it.",
		  "Replace each base by the maximum of itself and the values of the intervals covering it.",
		  "Keep, inside each (sorted, non-overlapping) interval of a file, only the minimum (the most\ncentral one on ties); everything else becomes the infinity value.",
		  "Keep, inside each (sorted, non-overlapping) interval of a file, only the maximum (the most\ncentral one on ties); everything else becomes the zero value." };
	if (indent == NULL) indent = "";
	char* text = copy_string (what[kind]);
	for (char* line = strtok (text, "\n") ; line != NULL ; line = strtok (NULL, "\n")) fprintf (f, "%s%s\n", indent, line);
	free (text);
	fprintf (f, "%s\n%susage: %s <filename> [options]\n", indent, indent, name);
	if (kind == K_MINOVER)      fprintf (f, "%s  --infinity=<value>       value for non-minima (default: largest double)\n", indent);
	else if (kind == K_MAXOVER) fprintf (f, "%s  --zero=<value>           (Z=) value for non-maxima (default 0.0)\n", indent);
	else if ((kind == K_MASK) || (kind == K_MASKNOT))
		fprintf (f, "%s  --mask=<value%s>     (M=) the mask value (default 0.0)\n", indent, (kind == K_MASK)? "|variable" : "");
	else
		fprintf (f, "%s  --value=<col>            column of the interval value (default: the global one)\n", indent);
	if ((kind <= K_DIVIDE) || (kind == K_OR) || (kind == K_AND))
		fprintf (f, "%s  --novalue                intervals carry no value (each counts 1)\n", indent);
	fprintf (f, "%s  --origin=one|zero        coordinate convention of the file\n", indent);
	if ((kind <= K_SUBTRACT) || (kind >= K_MINWITH)) fprintf (f, "%s  --destroy                delete the file afterwards\n", indent);
	if (kind == K_DIVIDE)   fprintf (f, "%s  --infinity=<value>       value standing for infinity (default: largest double)\n", indent);
	}

#define FILEOP_GROUP(fn, kind, text)                                                            \
OP_SHORT (fn, text)                                                                             \
void   fn##_usage (char* name, FILE* f, char* indent) { fileop_usage (name, f, indent, kind); } \
dspop* fn##_parse (char* name, int argc, char** argv) { return fileop_parse (name, argc, argv, kind); } \
void   fn##_free  (dspop* op) { fileop_free (op); }                                             \
void   fn##_apply (dspop* op, arg_dont_complain(char* vName), arg_dont_complain(u32 vLen),      \
                   arg_dont_complain(valtype* v)) { fileop_apply (op); }

FILEOP_GROUP (op_add,      K_ADD,      "add interval values (read from a file) to the current set of interval values")
FILEOP_GROUP (op_subtract, K_SUBTRACT, "subtract interval values (read from a file) from the current set of interval values")
FILEOP_GROUP (op_multiply, K_MULTIPLY, "multiply the current set of interval values by interval values read from a file")
FILEOP_GROUP (op_divide,   K_DIVIDE,   "divide the current set of interval values by interval values read from a file")
FILEOP_GROUP (op_mask,     K_MASK,     "mask the current set of interval values, over intervals read from a file")
FILEOP_GROUP (op_mask_not, K_MASKNOT,  "mask the current set of interval values, outside of intervals read from a file")
FILEOP_GROUP (op_or,       K_OR,       "logical-OR the current set of intervals with intervals read from a file")
FILEOP_GROUP (op_and,      K_AND,      "logical-AND the current set of intervals with intervals read from a file")
FILEOP_GROUP (op_min_with, K_MINWITH,  "take the minimum of the current values and interval values read from a file")
FILEOP_GROUP (op_max_with, K_MAXWITH,  "take the maximum of the current values and interval values read from a file")
FILEOP_GROUP (op_min_in_interval, K_MINOVER, "find the minimum value in each of a set of intervals read from a file")
FILEOP_GROUP (op_max_in_interval, K_MAXOVER, "find the maximum value in each of a set of intervals read from a file")

/* ---------------------------------------------------------------- input ---- */
typedef struct dspop_input
	{ dspop common;  char* filename;  int valColumn;  int missingVal;  int overlapOp, originOne, destroyFile; } dspop_input;

OP_SHORT (op_input, "read intervals from a file (replacing the current set)")

void op_input_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sReplace the signal by the intervals read from a file.\n\n", indent);
	fprintf (f, "%susage: %s <filename> [options]\n", indent, name);
	fprintf (f, "%s  --value=<col>  --novalue  --origin=one|zero   as for the main input\n", indent);
	fprintf (f, "%s  --missing=<value>        value of bases under no interval (default 0)\n", indent);
	fprintf (f, "%s  --overlap=sum|min|max    how overlapping intervals combine (default sum)\n", indent);
	fprintf (f, "%s  --destroy                delete the file afterwards\n", indent);
	}

dspop* op_input_parse (char* name, int argc, char** argv)       /* opio.c:88-205 */
	{
	dspop_input* op = (dspop_input*) new_op (name, sizeof(dspop_input), true);
	op->valColumn = (int) get_named_global ("valColumn", 4-1);
	op->originOne = (int) get_named_global ("originOne", false);
	op->overlapOp = ri_overlapSum;
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if ((strcmp (arg, "--novalue") == 0) || (strcmp (arg, "--novalues") == 0) || (strcmp (arg, "--value=none") == 0))
			{ op->valColumn = -1;  continue; }
		if (strcmp_prefix (arg, "--value=") == 0)
			{
			int col = string_to_int (argVal) - 1;
			if (col == -1) chastise ("[%s] value column can't be 0 (\"%s\")\n", name, arg);
			if (col < 0)   chastise ("[%s] value column can't be negative (\"%s\")\n", name, arg);
			if (col < 3)   chastise ("[%s] value column can't be 1, 2 or 3 (\"%s\")\n", name, arg);
			op->valColumn = col;
			continue;
			}
		if (strcmp_prefix (arg, "--missing=") == 0)
			{ op->missingVal = (int) string_to_valtype (argVal);  continue; }     /* an int in the reference too, opio.c:35 */
		if (strcmp (arg, "--overlap=sum") == 0) { op->overlapOp = ri_overlapSum;  continue; }
		if ((strcmp (arg, "--overlap=minimum") == 0) || (strcmp (arg, "--overlap=min") == 0)) { op->overlapOp = ri_overlapMin;  continue; }
		if ((strcmp (arg, "--overlap=maximum") == 0) || (strcmp (arg, "--overlap=max") == 0)) { op->overlapOp = ri_overlapMax;  continue; }
		if ((strcmp (arg, "--origin=one") == 0)  || (strcmp (arg, "--origin=1") == 0)) { op->originOne = true;   continue; }
		if ((strcmp (arg, "--origin=zero") == 0) || (strcmp (arg, "--origin=0") == 0)) { op->originOne = false;  continue; }
		if (strcmp (arg, "--destroy") == 0) { op->destroyFile = true;  continue; }
		if (strcmp_prefix (arg, "--") == 0) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if (op->filename == NULL) { op->filename = copy_string (arg);  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	if (op->filename == NULL) { fprintf (stderr, "[%s] no filename was provided\n", name);  exit (EXIT_FAILURE); }
	return (dspop*) op;
	}

void op_input_free (dspop* _op)
	{
	dspop_input* op = (dspop_input*) _op;
	if (op->filename != NULL) free (op->filename);
	free (op);
	}

void op_input_apply (dspop* _op, arg_dont_complain(char* vName), arg_dont_complain(u32 vLen), arg_dont_complain(valtype* v))
	{
	dspop_input* op = (dspop_input*) _op;
	FILE* f = fopen (op->filename, "rt");
	if (f == NULL) { fprintf (stderr, "[%s] can't open \"%s\" for reading\n", _op->name, op->filename);  exit (EXIT_FAILURE); }
	read_intervals (f, op->valColumn, op->originOne, op->overlapOp, /*clear*/ true, op->missingVal);
	fclose (f);
	if (op->destroyFile) remove (op->filename);
	}

/* --------------------------------------------------------------- output ---- */
typedef struct dspop_output
	{ dspop common;  char* filename;  int noOutputValues, valPrecision, collapseRuns, showUncovered, originOne; } dspop_output;

OP_SHORT (op_output, "write the current set of intervals to a file")

void op_output_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sWrite the signal, as intervals, to a file (the pipeline continues).\n\n", indent);
	fprintf (f, "%susage: %s <filename> [options]\n", indent, name);
	fprintf (f, "%s  --nooutputvalue  --precision=<n>  --nocollapse  --uncovered:hide|show|NA\n", indent);
	fprintf (f, "%s  --origin=one|zero        (each defaults to the global setting)\n", indent);
	}

dspop* op_output_parse (char* name, int argc, char** argv)      /* opio.c:320-435 */
	{
	dspop_output* op = (dspop_output*) new_op (name, sizeof(dspop_output), true);
	op->noOutputValues = (int) get_named_global ("noOutputValues", false);
	op->valPrecision   = (int) get_named_global ("valPrecision",   0);
	op->collapseRuns   = (int) get_named_global ("collapseRuns",   true);
	op->showUncovered  = (int) get_named_global ("showUncovered",  uncovered_hide);
	op->originOne      = (int) get_named_global ("originOne",      false);
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		if ((strcmp (arg, "--nooutputvalue") == 0) || (strcmp (arg, "--nooutputvalues") == 0)) { op->noOutputValues = true;  continue; }
		if (strcmp_prefix (arg, "--precision=") == 0)
			{
			op->valPrecision = string_to_int (argVal);
			if (op->valPrecision < 0) chastise ("[%s] precision can't be negative (\"%s\")\n", name, arg);
			continue;
			}
		if (strcmp (arg, "--nocollapse") == 0) { op->collapseRuns = false;  continue; }
		if ((strcmp (arg, "--uncovered:hide") == 0) || (strcmp (arg, "--hide:uncovered") == 0)) { op->showUncovered = uncovered_hide;  continue; }
		if ((strcmp (arg, "--uncovered:show") == 0) || (strcmp (arg, "--show:uncovered") == 0)) { op->showUncovered = uncovered_show;  continue; }
		if ((strcmp (arg, "--uncovered:NA") == 0) || (strcmp (arg, "--uncovered:mark") == 0)
		 || (strcmp (arg, "--mark:uncovered") == 0) || (strcmp (arg, "--markgaps") == 0)) { op->showUncovered = uncovered_NA;  continue; }
		if ((strcmp (arg, "--origin=one") == 0)  || (strcmp (arg, "--origin=1") == 0)) { op->originOne = true;   continue; }
		if ((strcmp (arg, "--origin=zero") == 0) || (strcmp (arg, "--origin=0") == 0)) { op->originOne = false;  continue; }
		if (strcmp_prefix (arg, "--") == 0) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		if (op->filename == NULL) { op->filename = copy_string (arg);  continue; }
		chastise ("[%s] Can't understand \"%s\"\n", name, arg);
		}
	if (op->filename == NULL) { fprintf (stderr, "[%s] no filename was provided\n", name);  exit (EXIT_FAILURE); }
	return (dspop*) op;
	}

void op_output_free (dspop* _op)
	{
	dspop_output* op = (dspop_output*) _op;
	if (op->filename != NULL) free (op->filename);
	free (op);
	}

void op_output_apply (dspop* _op, arg_dont_complain(char* vName), arg_dont_complain(u32 vLen), arg_dont_complain(valtype* v))
	{
	dspop_output* op = (dspop_output*) _op;
	FILE* f = fopen (op->filename, "wt");
	if (f == NULL) { fprintf (stderr, "[%s] can't open \"%s\" for writing\n", _op->name, op->filename);  exit (EXIT_FAILURE); }
	report_intervals (f, op->valPrecision, op->noOutputValues, op->collapseRuns, op->showUncovered, op->originOne);
	fclose (f);
	}
