/* ops_percentile.c -- percentile (device shim): exact order statistics of the sampled
 * genome, published as named variables percentile<p>.
 *
 * Argument rules, rank formula, variable names and messages follow percentile.c:131-375,
 * :587-589, :657-710, :756-780 in the reference.  The reference sorts the genome in place
 * (destroying it, percentile.c:34-36, hence its --preserve option); here the order statistics
 * are found on the device (gdsp_percentiles) and the signal is left untouched.  --preserve=<file>
 * still does what it does in the reference -- the signal goes to the file as text with ten decimals
 * and is read back afterwards (percentile.c:532-535, :716-724), so values come back rounded to ten
 * decimals and the (emptied) file is left behind -- because a pipeline's output must not depend on
 * which of the two programs ran it.  With several GPUs every device counts its own chromosomes and
 * the counts are summed by the library's reduction hook (reduce_over_devices below: RCCL
 * all-reduce over the devices of this process, or a host sum with --reduce=host). */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <float.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

#define percentileStepUnits 1000                  /* resolution 0.001 %, percentile.c:15 */

typedef struct dspop_percentile
	{
	dspop   common;
	u32     percentileLo, percentileHi, percentileStep;
	valtype minAllowed, maxAllowed;
	u32     windowSize;
	int     valPrecision, quiet, reportForBash;
	char   *preserveFilename, *mapFilename;
	} dspop_percentile;

OP_SHORT (op_percentile, "compute percentiles of the current set of interval values")

void op_percentile_usage (char* name, FILE* f, char* indent)
	{
	if (indent == NULL) indent = "";
	fprintf (f, "%sCompute percentiles over all chromosomes; each result is stored in a named\n", indent);
	fprintf (f, "%svariable (percentile99, percentile99.5, ...) that later operators can use as a\n", indent);
	fprintf (f, "%sthreshold. The signal is not modified.\n\n", indent);
	fprintf (f, "%susage: %s <percentile> [options]\n", indent, name);
	fprintf (f, "%s  <low>[..<high>][by<step>]  one percentile or a range of them\n", indent);
	fprintf (f, "%s  <low>,<high>             exactly two percentiles\n", indent);
	fprintf (f, "%s  --step=<value>           step through a range (default 1)\n", indent);
	fprintf (f, "%s  --window=<length>        (W=) look at one base per window\n", indent);
	fprintf (f, "%s  --min=<value> --max=<value>  ignore values outside this range\n", indent);
	fprintf (f, "%s  --precision=<number>     digits when reporting values\n", indent);
	fprintf (f, "%s  --map=<filename>         write \"value percentile\" lines to a file\n", indent);
	fprintf (f, "%s  --report:bash            print results as shell assignments on stdout\n", indent);
	fprintf (f, "%s  --quiet                  do not report results on stderr\n", indent);
	fprintf (f, "%s  --preserve=<filename>    write the signal to this file first and read it back afterwards, as\n", indent);
	fprintf (f, "%s                           genodsp does (values return rounded to ten decimals); without it\n", indent);
	fprintf (f, "%s                           the signal is simply left as it is\n", indent);
	}

static u32 to_thousandths (valtype pct)           /* percentile.c:296-302 */
	{
	if (pct <   0.0) return 0;
	if (pct > 100.0) return 100*percentileStepUnits;
	return (u32) (int) (percentileStepUnits*pct + .5);
	}

dspop* op_percentile_parse (char* name, int argc, char** argv)
	{
	dspop_percentile* op = (dspop_percentile*) new_op (name, sizeof(dspop_percentile), true);
	int haveRange = false;
	op->percentileStep = percentileStepUnits;
	op->minAllowed     = -valtypeMax;
	op->maxAllowed     =  valtypeMax;
	op->windowSize     = (u32) get_named_global ("windowSize", 1);      /* percentile.c:153: the global --window= */
	if (op->windowSize == 0) op->windowSize = 1;
	op->valPrecision   = (int) get_named_global ("valPrecision", 0);

	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg = argv[0];
		char* argVal = strchr (arg, '=');  if (argVal != NULL) argVal++;
		char* stepText = NULL;

		if (strcmp_prefix (arg, "--step=") == 0) { stepText = argVal;  goto set_step; }
		if (is_opt3 (arg, "window", "W"))
			{
			int w = string_to_unitized_int (argVal, /*thousands*/ true);
			if (w == 0) w = 1;
			if (w < 0) chastise ("[%s] window size can't be negative (\"%s\")\n", name, arg);
			op->windowSize = (u32) w;
			continue;
			}
		if (strcmp_prefix (arg, "--min=") == 0) { op->minAllowed = string_to_valtype (argVal);  continue; }
		if (strcmp_prefix (arg, "--max=") == 0) { op->maxAllowed = string_to_valtype (argVal);  continue; }
		if (strcmp_prefix (arg, "--precision=") == 0)
			{
			op->valPrecision = string_to_int (argVal);
			if (op->valPrecision < 0) chastise ("[%s] precision can't be negative (\"%s\")\n", name, arg);
			continue;
			}
		if (strcmp_prefix (arg, "--preserve=") == 0)
			{ if (op->preserveFilename == NULL) op->preserveFilename = copy_string (argVal);  continue; }
		if ((strcmp_prefix (arg, "--map=") == 0) || (strcmp_prefix (arg, "--mapping=") == 0))
			{ if (op->mapFilename == NULL) op->mapFilename = copy_string (argVal);  continue; }
		if ((strcmp (arg, "--report:bash") == 0) || (strcmp (arg, "--bash") == 0)) { op->reportForBash = true;  continue; }
		if ((strcmp (arg, "--quiet") == 0) || (strcmp (arg, "--silent") == 0))     { op->quiet = true;  continue; }
		if (strcmp_prefix (arg, "--debug") == 0) continue;
		if (strcmp_prefix (arg, "--") == 0) chastise ("[%s] Can't understand \"%s\"\n", name, arg);

		if (haveRange) chastise ("[%s] Can't understand \"%s\"\n", name, arg);
			{
			valtype lo, hi;
			char*   text  = copy_string (arg);
			char*   comma = strchr (text, ',');
			if (comma != NULL)                                 /* <low>,<high> */
				{
				*(comma++) = 0;
				lo = string_to_valtype (text);  hi = string_to_valtype (comma);
				if (lo > hi) { valtype t = hi;  hi = lo;  lo = t; }
				op->percentileLo = to_thousandths (lo);  op->percentileHi = to_thousandths (hi);
				op->percentileStep = (op->percentileLo < op->percentileHi)? op->percentileHi - op->percentileLo : 1;
				}
			else                                               /* <low>[..<high>][by<step>] */
				{
				char* dots = strstr (text, "..");
				if (dots != NULL)
					{
					*dots = 0;  dots += 2;
					char* by = strstr (dots, "by");
					if (by != NULL) { *by = 0;  stepText = arg + (by + 2 - text); }
					lo = string_to_valtype (text);  hi = string_to_valtype (dots);
					}
				else lo = hi = string_to_valtype (text);
				if (lo > hi) { valtype t = hi;  hi = lo;  lo = t; }
				op->percentileLo = to_thousandths (lo);  op->percentileHi = to_thousandths (hi);
				}
			free (text);
			haveRange = true;
			if (stepText == NULL) continue;
			}
	set_step:
			{
			valtype st = string_to_valtype (stepText);
			if (st == 0) chastise ("[%s] step can't be zero (\"%s\")\n", name, arg);
			if (st < 0)  chastise ("[%s] step can't be negative (\"%s\")\n", name, arg);
			if (st < .001) st = .001;
			op->percentileStep = (u32) (percentileStepUnits*st + .5);
			}
		}
	if (!haveRange) { fprintf (stderr, "[%s] no range of percentiles was provided\n", name);  exit (EXIT_FAILURE); }
	if (op->reportForBash && op->quiet) chastise ("[%s] Can't use both --report:bash and --quiet\n", name);
	return (dspop*) op;
	}

void op_percentile_free (dspop* _op)
	{
	dspop_percentile* op = (dspop_percentile*) _op;
	if (op->preserveFilename != NULL) free (op->preserveFilename);
	if (op->mapFilename      != NULL) free (op->mapFilename);
	free (op);
	}

static void percentile_name (char* varName, u32 percentile);

static void percentile_name (char* varName, u32 percentile)     /* percentile.c:756-780 */
	{
	if (percentile % percentileStepUnits == 0)
		{ sprintf (varName, "percentile%d", percentile / percentileStepUnits);  return; }
	float pPct = percentile / ((float) percentileStepUnits);
	int   precision = 1;
	for (u32 denom=percentileStepUnits/10 ; denom>=1 ; precision++, denom/=10)
		{ if (percentile % denom == 0) { sprintf (varName, "percentile%.*f", precision, pPct);  return; } }
	sprintf (varName, "percentile%f", pPct);
	}

static int percentile_run (dspop* _op, dspop* binarize);

void op_percentile_apply (dspop* _op, arg_dont_complain(char* vName), arg_dont_complain(u32 vLen), arg_dont_complain(valtype* v))
	{ percentile_run (_op, NULL); }

dspprototypes(op_binarize)

/* The percentile, and -- when `next` is a binarize whose threshold is one of this operator's percentiles, the signal lives
 * in whole chromosomes and nothing asks for the reference's detour through a file -- that binarize in the same read of
 * the signal (gdsp_percentiles_binarize: the counting pass writes one / zero wherever its bracket already decides).
 * The outputs land in the partner vectors, which then become the signal.  Same values, same messages in the same
 * order, same signal as the two operators one after the other. */
static int percentile_fusable (dspop* _op, dspop* next)
	{
	dspop_percentile* op = (dspop_percentile*) _op;
	char varName[100];
	int  ties;  valtype one, zero;
	if ((next == NULL) || (next->funcApply != op_binarize_apply)) return false;
	const char* want = op_binarize_pending (next, &ties, &one, &zero);
	if ((want == NULL) || (op->preserveFilename != NULL) || !signal_in_whole_chromosomes ()) return false;
	if ((op->mapFilename == NULL) && ((op->percentileLo == 0) || (op->percentileLo == 100*percentileStepUnits))
	 && ((op->percentileHi == 0) || (op->percentileHi == 100*percentileStepUnits))) return false;        /* answered from the extremes alone */
	for (u32 pt=op->percentileLo ; pt<=op->percentileHi ; pt+=op->percentileStep)
		{
		percentile_name (varName, pt);
		if (strcmp (varName, want) == 0) return true;
		if (op->percentileStep == 0) break;
		}
	return false;
	}

int percentile_with_binarize (dspop* _op, dspop* next)
	{
	if (percentile_fusable (_op, next)) return percentile_run (_op, next)? 2 : 1;
	percentile_run (_op, NULL);
	return 1;
	}

/* returns true when `binarize` was run with it (false: the percentile alone has run -- or, asked to fuse what cannot be
 * fused, nothing has) */
static int percentile_run (dspop* _op, dspop* binarize)
	{
	dspop_percentile* op = (dspop_percentile*) _op;
	char  varName[100];

	/* percentile.c:432-530: `0`, `100` and any range from 0 to 100 are answered from the minimum and maximum of
	 * the sample alone -- only percentile0 / percentile100 are set, nothing is reported, nothing is preserved */
	int onlyMin = (op->percentileLo == 0) && (op->percentileHi == 0);
	int onlyMax = (op->percentileLo == 100*percentileStepUnits) && (op->percentileHi == 100*percentileStepUnits);
	int minAndMax = (op->percentileLo == 0) && (op->percentileHi == 100*percentileStepUnits);
	int extremesOnly = (op->mapFilename == NULL) && (onlyMin || onlyMax || minAndMax);

	if ((op->preserveFilename != NULL) && !extremesOnly) write_all_chromosomes (op->preserveFilename);
	FILE* mapF = (op->mapFilename != NULL)? fopen (op->mapFilename, "wt") : NULL;

	/* every chromosome on every GPU is one source of the population; the counts of the devices of this
	 * process are added through the driver's reduction (RCCL over the devices in use, see genodsp_hip.c) */
	int npct = 0;
	sigpart* parts;
	int nsrc = signal_parts (&parts);
	if (extremesOnly) npct = minAndMax? 2 : 1;
	else
		{
		for (u32 pt=op->percentileLo ; pt<=op->percentileHi ; pt+=op->percentileStep)
			{ npct++;  if (op->percentileStep == 0) break; }
		}
	gdsp_select_source* src = (gdsp_select_source*) calloc (nsrc? nsrc : 1, sizeof(gdsp_select_source));
	u32*     pts  = (u32*)     calloc (npct, sizeof(u32));
	valtype* vals = (valtype*) calloc (npct, sizeof(valtype));
	if ((src == NULL) || (pts == NULL) || (vals == NULL))
		{ fprintf (stderr, "[%s] out of memory\n", _op->name);  exit (EXIT_FAILURE); }
	sync_all_devices ();
	for (int i=0 ; i<nsrc ; i++)
		{
		/* the sample is every windowSize-th base counted from the chromosome's first (percentile.c:560): a stretch
		 * that starts at base `first` begins with the next multiple */
		u32 skip = (op->windowSize - parts[i].first % op->windowSize) % op->windowSize;
		select_device_of (parts[i].s);
		src[i].d_v = (skip < parts[i].n)? parts[i].v + skip : NULL;
		src[i].n   = (skip < parts[i].n)? parts[i].n - skip : 0;
		src[i].device = physical_device_of (parts[i].s);  src[i].stream = op_stream ();
		}
	npct = 0;
	if (extremesOnly)
		{
		if (onlyMax || minAndMax) pts[npct++] = 100*percentileStepUnits;      /* percentile100 is set first (:524-527) */
		if (onlyMin || minAndMax) pts[npct++] = 0;
		}
	else
		{
		for (u32 pt=op->percentileLo ; pt<=op->percentileHi ; pt+=op->percentileStep)
			{ pts[npct++] = pt;  if (op->percentileStep == 0) break; }
		}
	u64 numValues = 0;
	void* reduceCtx = NULL;
	gdsp_reduce_fn reduce = reduce_over_devices (&reduceCtx);
	int fused = false;
	if (binarize != NULL)
		{
		/* which of this operator's percentiles is the binarize's threshold? */
		gdsp_percentile_binarize fuse;
		const char* want = op_binarize_pending (binarize, &fuse.tiesAbove, &fuse.one, &fuse.zero);
		fuse.which = -1;
		for (int ip=0 ; (want != NULL) && (ip<npct) ; ip++)
			{ percentile_name (varName, pts[ip]);  if (strcmp (varName, want) == 0) fuse.which = ip; }
		if (fuse.which < 0) { fprintf (stderr, "[%s] internal error: nothing to fuse\n", _op->name);  exit (EXIT_FAILURE); }   /* (percentile_fusable said yes) */
		double** outs = (double**) calloc (nsrc? nsrc : 1, sizeof(double*));
		if (outs == NULL) { fprintf (stderr, "[%s] out of memory\n", _op->name);  exit (EXIT_FAILURE); }
		for (int i=0 ; i<nsrc ; i++)
			{
			/* the fused kernels index the output like the source: both must start at the vector's first base
			 * (percentile_fusable asks for whole chromosomes and a window of one base, so nothing is skipped) */
			if ((src[i].n != 0) && ((src[i].d_v != parts[i].base) || (parts[i].first != 0)))
				{ fprintf (stderr, "[%s] internal error: a fused binarize over a stretch\n", _op->name);  exit (EXIT_FAILURE); }
			outs[i] = partner_of (parts[i].s);
			}
		fuse.d_out = outs;
		int onePass = 0;
		check_gdsp (gdsp_percentiles_binarize (src, nsrc, op->windowSize, op->minAllowed, op->maxAllowed, pts, npct,
		                                       selectStrategy, 0, reduce, reduceCtx, vals, &numValues, &fuse, &onePass), "percentile");
		free (outs);
		fused = (numValues != 0);
		}
	else
		check_gdsp (gdsp_percentiles (src, nsrc, op->windowSize, op->minAllowed, op->maxAllowed, pts, npct,
		                              selectStrategy, 0, reduce, reduceCtx, vals, &numValues), "percentile");
	if (nsrc > 0) select_device_of (parts[0].s);
	free (src);
	if (getenv ("GDSP_PERCENTILE_REPORT") != NULL)               /* how the call was answered (tests; gdsp_percentiles_stats) */
		{
		uint64_t st[8];
		gdsp_percentiles_stats (st);
		fprintf (stderr, "[%s] route=%s resident=%d readbacks=%s population=%llu sample=%llu candidates=%llu fallbacks=%llu fused=%d\n",
		         _op->name, (st[0] == GDSP_SELECT_BRACKET)? "bracket" : "radix", (int) st[7], (st[7] != 0)? "1" : "many",
		         (unsigned long long) st[1], (unsigned long long) st[2], (unsigned long long) st[3], (unsigned long long) st[4], (int) st[6]);
		}
	if (numValues == 0)
		{
		fprintf (stderr, "[%s] percentile can't be computed;  no input values meet the criteria\n", _op->name);
		if (mapF != NULL) fclose (mapF);
		free (pts);  free (vals);
		return false;                                    /* (a binarize that was to run with it runs on its own, and says what the reference says) */
		}

	for (int ip=0 ; ip<npct ; ip++)
		{
		u32     pt   = pts[ip];
		valtype pVal = vals[ip];

		float pPct = pt / ((float) percentileStepUnits);
		percentile_name (varName, pt);
		set_named_global (varName, pVal);
		if (extremesOnly) continue;
		if (op->reportForBash)
			fprintf (stdout, "%s=" valtypeFmtPrec " # bash command\n", varName, op->valPrecision, pVal);
		else if (!op->quiet)
			fprintf (stderr, "percentile %.3f is " valtypeFmtPrec "\n", pPct, op->valPrecision, pVal);
		if (mapF != NULL) fprintf (mapF, valtypeFmtPrec " %.3f\n", op->valPrecision, pVal, pPct);
		}
	if (mapF != NULL) fclose (mapF);
	free (pts);  free (vals);

	if ((op->preserveFilename != NULL) && !extremesOnly)     /* percentile.c:716-724: restore, then empty the file */
		{
		read_all_chromosomes (op->preserveFilename);
		FILE* f = fopen (op->preserveFilename, "wb");
		if (f != NULL) fclose (f);
		}
	if (fused)
		{
		/* the binarize's own words when it picks its threshold up (logical.c:234-243), then its output becomes the signal */
		valtype T, one, zero;  int ties;
		op_binarize_describe (binarize, &T, &ties, &one, &zero);
		for (spec** c=chromsSorted ; *c!=NULL ; c++) flip_spec (*c);
		}
	return fused;
	}
