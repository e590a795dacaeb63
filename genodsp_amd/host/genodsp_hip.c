/* genodsp_hip.c -- host driver of the MI355X genodsp: the `= operator` pipeline CLI.
 *
 *   genodsp_hip --chromosomes=<file> [options] (= <operator> [args])*   < intervals > intervals
 *
 * Command line, interval text in/out, named variables and the operator table are
 * those of the reference driver (rsharris/genodsp genodsp.c; lines cited inline),
 * so a pipeline written for `genodsp` runs unchanged.  What differs is underneath:
 * every chromosome is a pair of f64 arrays in HBM (the vector and its partner, so
 * out-of-place operators finish with a pointer flip instead of the reference's
 * copy-back pass, e.g. sum.c:672-673), operators launch kernels of
 * libgenodsp_hip.so on a stream, intervals are parsed into pinned staging buffers
 * and applied on the device in file order, and output runs are found on the device
 * so that only (start,end,value) triples come back over PCIe.  With --gpus=N whole
 * chromosomes are dealt to N devices longest-first (LPT); no operator except
 * percentile/invert needs anything from another device.
 *
 * There is no CPU compute path here: if the HIP library or a GPU is missing the
 * program stops with a message.
 */
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <stdarg.h>
#include <float.h>
#include <time.h>
#include <pthread.h>
#include <unistd.h>
#include "genodsp_interface.h"
#include "genodsp_hip.h"
#include "utilities.h"
#include "host_services.h"

#define programName            "genodsp_hip"
#define programVersion         "0.1 (MI355X/gfx950; genodsp 0.0.10 command set)"
#define specialPipeChar        '='

/* ------------------------------------------------------------ operator table */
/* same names, aliases and order as the reference's dspTable (genodsp.c:117-174): all 37
 * operators.  As in the reference a "plugin" is a link-time function group: a further group
 * is added by compiling the driver with -DGDSP_EXTRA_OPERATORS='"my_ops.h"', a header that
 * declares the groups (dspprototypes) and defines GDSP_EXTRA_DSPTABLE_ROWS as the rows to append
 * (dspinforecord("name", op_x), ...), and by linking the group's object file (INTEGRATION.md;
 * tests/test_plugin_boundary.py builds and runs such a driver). */
dspprototypes(op_window_sum)     dspprototypes(op_sliding_sum)   dspprototypes(op_smooth)
dspprototypes(op_cumulative_sum) dspprototypes(op_percentile)    dspprototypes(op_add)
dspprototypes(op_subtract)       dspprototypes(op_add_constant)  dspprototypes(op_invert)
dspprototypes(op_multiply)       dspprototypes(op_divide)        dspprototypes(op_absolute_value)
dspprototypes(op_clip)           dspprototypes(op_erase)         dspprototypes(op_binarize)
dspprototypes(op_local_minima)   dspprototypes(op_local_maxima)  dspprototypes(op_best_local_min)
dspprototypes(op_best_local_max) dspprototypes(op_close)         dspprototypes(op_open)
dspprototypes(op_dilate)         dspprototypes(op_erode)         dspprototypes(op_input)
dspprototypes(op_output)         dspprototypes(op_show_variables)
dspprototypes(op_mask)           dspprototypes(op_mask_not)      dspprototypes(op_or)
dspprototypes(op_and)            dspprototypes(op_min_with)      dspprototypes(op_max_with)
dspprototypes(op_map)            dspprototypes(op_min_in_interval) dspprototypes(op_max_in_interval)
dspprototypes(op_clump)          dspprototypes(op_skimp)
#ifdef GDSP_EXTRA_OPERATORS
#include GDSP_EXTRA_OPERATORS
#endif

static dspinfo dspTable[] =
	{dspinforecord("sum"           , op_window_sum)     , dspinfoalias ("window_sum")     ,
	 dspinforecord("slidingsum"    , op_sliding_sum)    , dspinfoalias ("sliding_sum")    ,
	 dspinforecord("smooth"        , op_smooth)         ,
	 dspinforecord("cumulativesum" , op_cumulative_sum) , dspinfoalias ("cumulative")     , dspinfoalias ("integrate"),
	 dspinforecord("clump"         , op_clump)          ,
	 dspinforecord("anticlump"     , op_skimp)          , dspinfoalias ("anti_clump")     , dspinfoalias ("skimp"),
	 dspinforecord("percentile"    , op_percentile)     ,
	 dspinforecord("add"           , op_add)            ,
	 dspinforecord("subtract"      , op_subtract)       ,
	 dspinforecord("addconst"      , op_add_constant)   , dspinfoalias ("add_const")      ,
	 dspinforecord("invert"        , op_invert)         ,
	 dspinforecord("multiply"      , op_multiply)       ,
	 dspinforecord("divide"        , op_divide)         ,
	 dspinforecord("abs"           , op_absolute_value) ,
	 dspinforecord("mask"          , op_mask)           ,
	 dspinforecord("masknot"       , op_mask_not)       , dspinfoalias ("mask_not")       ,
	 dspinforecord("clip"          , op_clip)           ,
	 dspinforecord("erase"         , op_erase)          ,
	 dspinforecord("binarize"      , op_binarize)       ,
	 dspinforecord("or"            , op_or)             ,
	 dspinforecord("and"           , op_and)            ,
	 dspinforecord("maxover"       , op_max_in_interval), dspinfoalias ("max_over")       ,
	 dspinforecord("minover"       , op_min_in_interval), dspinfoalias ("min_over")       ,
	 dspinforecord("localmin"      , op_local_minima)   , dspinfoalias ("local_min")      ,
	 dspinforecord("localmax"      , op_local_maxima)   , dspinfoalias ("local_max")      ,
	 dspinforecord("bestmin"       , op_best_local_min) , dspinfoalias ("best_min")       ,
	 dspinfoalias ("bestlocalmin")                      , dspinfoalias ("best_local_min") ,
	 dspinforecord("bestmax"       , op_best_local_max) , dspinfoalias ("best_max")       ,
	 dspinfoalias ("bestlocalmax")                      , dspinfoalias ("best_local_max") ,
	 dspinforecord("minwith"       , op_min_with)       , dspinfoalias ("min_with")       ,
	 dspinforecord("maxwith"       , op_max_with)       , dspinfoalias ("max_with")       ,
	 dspinforecord("close"         , op_close)          ,
	 dspinforecord("open"          , op_open)           ,
	 dspinforecord("dilate"        , op_dilate)         ,
	 dspinforecord("erode"         , op_erode)          ,
	 dspinforecord("map"           , op_map)            ,
	 dspinforecord("input"         , op_input)          ,
	 dspinforecord("output"        , op_output)         ,
	 dspinforecord("variables"     , op_show_variables)
#ifdef GDSP_EXTRA_DSPTABLE_ROWS
	 , GDSP_EXTRA_DSPTABLE_ROWS
#endif
	};
#define dspTableLen (sizeof(dspTable)/sizeof(dspinfo))

static const char* notInThisBuild[] = { NULL };    /* every operator of the reference's table is built */

/* ------------------------------------------------------------------- globals */
spec*  chromsOfInterest = NULL;
spec** chromsSorted     = NULL;
int    trackOperations  = false;
int    dbgInput         = false;     /* --debug=input, --debug=pipe, --debug=globals: genodsp.c:61-63 */
int    dbgPipe          = false;
int    dbgGlobals       = false;
int    reportComments   = false;
u32    reportInputProgress = 0;

static dspop* pipeline = NULL, *tailOp = NULL;
static int valColumn      = 4-1;                 /* genodsp.c:48-56 defaults */
static int noOutputValues = false;
static int valPrecision   = 0;
static int collapseRuns   = true;
static int showUncovered  = uncovered_hide;
static int clipToLength   = false;
static int originOne      = false;
static int inhibitOutput  = false;
static int numDevices     = 1;
int        firMode        = GDSP_FIR_EXACT;      /* --smooth=exact|fma|hann (see ops_sum.c) */
int        selectStrategy = GDSP_SELECT_AUTO;    /* --percentile=auto|radix|bracket (see ops_percentile.c) */
static int fuseChains     = true;                /* --nofuse: one kernel per operator          */
static int batchLaunches  = -1;                  /* --nobatch: one launch per operator and chromosome, chromosome by chromosome; --batch: one per operator and device (the default, except under --progress=operations) */
enum { reduce_auto, reduce_rccl, reduce_host };
static int reduceHow      = reduce_auto;         /* --reduce=rccl|host: how whole-genome operators combine the devices */
static gdsp_comm* deviceComm = NULL;             /* RCCL communicator over the devices in use (NULL: host sums)        */
static u64 intervalsRead  = 0, linesWritten = 0; /* (what --report=gpu prints beside the times of ingest and output) */
static int reportGpu      = false;               /* --report=gpu: per-operator device time, Gbases/s and GB/s on stderr */

/* a chromosome as the driver sees it: the public spec first, device state after */
typedef struct xspec
	{
	spec     pub;
	valtype* partner;     /* second HBM buffer, same length */
	int      device;
	} xspec;

typedef struct devstate
	{
	void*    stream;
	valtype* scratch[4];  /* lazily allocated, longest-local-chromosome sized */
	int      scratchInUse[4];
	s32*     scratchInts[4];
	int      scratchIntsInUse[4];
	u32      maxLength;   /* longest chromosome on this device */
	} devstate;

/* --sharding=bases: between ingest and report the signal may live as STRETCHES of chromosomes, equal
 * shares of the genome's bases per device, each stretch with haloCap bases of its neighbours either side.  A stretch
 * is a spec like any other -- name of its chromosome, `start` = where its vector begins, the vector in HBM -- so the
 * operators are applied to it unchanged. */
typedef struct piece
	{
	xspec x;                /* what an operator is handed: x.pub.valVector[0] is base extStart of the chromosome */
	spec* whole;            /* the chromosome this is a stretch of */
	u32   extStart;         /* first base held (halo included) */
	u32   ownStart, ownEnd; /* the bases this stretch answers for: [ownStart, ownEnd) */
	} piece;
static int    shardBases     = false;            /* --sharding=bases */
static int    showShards     = false;            /* --shards=show: print the plan and stop */
static piece* pieces         = NULL;
static int    numPieces      = 0;
static int    signalInPieces = false;            /* where the signal is current: the stretches, or the whole chromosomes */
static int    halosFresh     = false;            /* every stretch's halo equals its neighbours' bases */
static u32    haloCap        = 0;                /* halo bases per side: the longest reach of a sharded run */
static spec*  activeSpec     = NULL;             /* the stretch an operator is being applied to */

static devstate devs[64];
static int      currentDevice = 0;
static int      physicalDevices = 1;   /* logical device d runs on GPU d % physicalDevices (see --gpus) */

static int use_device (int logical)
	{ return gdsp_set_device (logical % physicalDevices); }

void check_gdsp (int status, const char* what)
	{
	if (status == GDSP_OK) return;
	fprintf (stderr, "[%s] %s: %s\n", programName, what, gdsp_last_error ());
	exit (EXIT_FAILURE);
	}

/* ------------------------------------------------------------------ chastise */
static opfunc_usage chastiseUsage     = NULL;
static char*        chastiseUsageName = NULL;

static void usage (void)
	{
	fprintf (stderr,
	"usage: [cat <file>] | %s --chromosomes=<filename> [options] [operations]\n\n"
	"  --chromosomes=<filename>  (required) chromosome names and lengths, two columns\n"
	"  <name>:<length>           ... or give chromosomes directly (also <name>:<start>:<end>)\n"
	"  --value=<col>             column of the interval value (default 4)\n"
	"  --novalue                 intervals carry no value (each counts 1)\n"
	"  --nooutputvalue           write intervals without their value\n"
	"  --precision=<number>      digits after the decimal point in output (default 0)\n"
	"  --nocollapse              one output line per base, no run collapsing\n"
	"  --uncovered:hide|show|NA  how zero-valued stretches are written (default hide)\n"
	"  --cliptochromosome        clip intervals to the chromosome instead of failing\n"
	"  --origin=one|zero         interval coordinate convention (default zero, half-open)\n"
	"  --nooutput                do not write the resulting signal\n"
	"  --window=<length>         (W=) default window size for windowed operators\n"
	"  --gpus=<n>                shard whole chromosomes over n GPUs (default 1)\n"
	"  --sharding=chromosomes|bases  with --gpus=n: whole chromosomes dealt longest-first (default), or equal\n"
	"                            shares of the genome's bases, chromosomes cut where needed (stretches carry\n"
	"                            the halo their operators reach into; same output either way)\n"
	"  --shards=show             print which device gets what (and the makespan efficiency) and stop\n"
	"  --reduce=rccl|host        how percentile / invert combine the GPUs' counts: an RCCL all-reduce in\n"
	"                            HBM (default with --gpus > 1) or sums on the host\n"
	"  --nobatch                 apply operators chromosome by chromosome, one launch each, in the reference's order\n"
	"                            (implied by --progress=operations, whose lines then come in the reference's order;\n"
	"                            --batch keeps one launch per operator and device there too)\n"
	"                            (default: an operator covers all the chromosomes of a device in one launch)\n"
	"  --nofuse                  run every operator as its own kernel (default: the chains\n"
	"                            smooth=localmax|localmin and dilate=erode[=binarize] are fused)\n"
	"  --smooth=exact|fma|hann   arithmetic of `smooth`: exact = bit-identical to genodsp\n"
	"                            (default); fma = fused multiply-add, one rounding per tap;\n"
	"                            hann = block sums of the window (fastest, same tolerance,\n"
	"                            not shift invariant: a smooth feeding localmin/localmax is\n"
	"                            evaluated as fma instead; see DESIGN.md)\n"
	"  --percentile=auto|radix|bracket  how `percentile` finds its order statistics (same\n"
	"                            values either way; auto brackets them from a subsample\n"
	"                            when the genome is large)\n"
	"  --help[=<operator>]  ?  ?<operator>   operator help\n"
	"  --report=gpu              after the run, print what each operator cost on the GPU(s): calls, HIP-event\n"
	"                            milliseconds (slowest device), Gbases/s and the GB/s of SURVEY 8d's bytes per base\n"
	"  --report=comments  --progress=input:<n>  --progress=operations  --version\n\n"
	"Overlapping input intervals are summed. Input comes from stdin unless the first\n"
	"operator is \"input\". Operations have the form  = <operator> [arguments].\n", programName);
	exit (EXIT_FAILURE);
	}

void chastise (const char* format, ...)          /* genodsp.c:193-209 */
	{
	va_list args;
	va_start (args, format);
	if (format != NULL) vfprintf (stderr, format, args);
	va_end (args);
	if (chastiseUsage != NULL)
		{
		(*chastiseUsage) (chastiseUsageName, stderr, "  ");
		exit (EXIT_FAILURE);
		}
	usage ();
	}

static void usage_operations (void)
	{
	fprintf (stderr, "Operations (general form is %c <operator> [arguments]):\n", specialPipeChar);
	for (u32 i=0 ; i<dspTableLen ; i++)
		{ if (dspTable[i].funcShort != NULL) (*dspTable[i].funcShort) (dspTable[i].name, 12, stderr, "  "); }
	exit (EXIT_FAILURE);
	}

static dspinfo* find_operator (const char* name)  /* alias rows resolve to the row above, genodsp.c:666-673 */
	{
	dspinfo* real = NULL;
	for (u32 i=0 ; i<dspTableLen ; i++)
		{
		if (dspTable[i].funcShort != NULL) real = &dspTable[i];
		if (strcmp (name, dspTable[i].name) == 0) return real;
		}
	return NULL;
	}

/* ------------------------------------------------------------- named globals */
typedef struct namedglobal { struct namedglobal* next;  char* name;  valtype v; } namedglobal;
static namedglobal* namedGlobalHead = NULL;

static namedglobal* find_named_global (const char* name)
	{
	for (namedglobal* g=namedGlobalHead ; g!=NULL ; g=g->next)
		{ if (strcmp (name, g->name) == 0) return g; }
	return NULL;
	}

void set_named_global (char* name, valtype val)   /* newest first, like genodsp.c:2093-2104 */
	{
	if (dbgGlobals) fprintf (stderr, "set_named_global(%s," valtypeFmt ")\n", name, val);
	namedglobal* g = find_named_global (name);
	if (g == NULL)
		{
		g = (namedglobal*) malloc (sizeof(namedglobal));
		if (g == NULL) { fprintf (stderr, "out of memory for named global \"%s\"\n", name);  exit (EXIT_FAILURE); }
		g->name = copy_string (name);
		g->next = namedGlobalHead;
		namedGlobalHead = g;
		}
	g->v = val;
	}

valtype get_named_global (char* name, valtype defaultVal)
	{
	namedglobal* g = find_named_global (name);
	if (dbgGlobals)                                /* genodsp.c:2110-2131 */
		{
		fprintf (stderr, "get_named_global(%s) = ", name);
		if (g == NULL) fprintf (stderr, valtypeFmt " (default)\n", defaultVal);
		else           fprintf (stderr, valtypeFmt "\n", g->v);
		}
	return (g == NULL)? defaultVal : g->v;
	}

int named_global_exists (char* name, valtype* val)
	{
	namedglobal* g = find_named_global (name);
	if (dbgGlobals)                                /* genodsp.c:2143-2163 */
		{
		fprintf (stderr, "named_global_exists(%s) = ", name);
		if (g == NULL) fprintf (stderr, " (not found)\n");
		else           fprintf (stderr, valtypeFmt "\n", g->v);
		}
	if (g == NULL) return false;
	if (val != NULL) *val = g->v;
	return true;
	}

void report_named_globals (FILE* f, char* indent)  /* genodsp.c:2176-2199 */
	{
	int w = 1;
	if (indent == NULL) indent = "";
	for (namedglobal* g=namedGlobalHead ; g!=NULL ; g=g->next)
		{ int n = (int) strlen (g->name);  if (n > w) w = n; }
	if (w > 20) w = 20;
	for (namedglobal* g=namedGlobalHead ; g!=NULL ; g=g->next)
		fprintf (f, "%s%*s = " valtypeFmt "\n", indent, w, g->name, g->v);
	}

/* progress line that overwrites itself unless it ends in a newline (genodsp.c:2219-2238) */
void tracking_report (const char* format, ...)
	{
	static int prevLen = 0;
	char    line[1001];
	va_list args;
	va_start (args, format);
	line[0] = 0;
	if (format != NULL) vsnprintf (line, sizeof(line), format, args);
	va_end (args);
	int len = (int) strlen (line);
	int nl  = (len > 0) && (line[len-1] == '\n');
	if (nl) line[--len] = 0;
	fprintf (stderr, "%s", line);
	if (prevLen > len) fprintf (stderr, "%*s", prevLen - len, "");
	if (nl) { fprintf (stderr, "\n");  prevLen = 0; }
	else    { fprintf (stderr, "\r");  prevLen = len; }
	}

/* --------------------------------------------------------------- chromosomes */
static int add_chromosome_spec (char* name, u32 chromStart, u32 chromLength)   /* genodsp.c:1014-1060 */
	{
	spec* tail = NULL;
	if (chromLength == 0) return true;
	for (spec* s=chromsOfInterest ; s!=NULL ; s=s->next)
		{ tail = s;  if (strcmp (name, s->chrom) == 0) return false; }
	xspec* x = (xspec*) calloc (1, sizeof(xspec));
	if (x == NULL) { fprintf (stderr, "out of memory for chromosome \"%s\"\n", name);  exit (EXIT_FAILURE); }
	x->pub.chrom  = copy_string (name);
	x->pub.start  = chromStart;
	x->pub.length = chromLength;
	if (tail == NULL) chromsOfInterest = &x->pub;  else tail->next = &x->pub;
	return true;
	}

spec* find_chromosome_spec (char* chrom)
	{
	for (spec* s=chromsOfInterest ; s!=NULL ; s=s->next)
		{ if (strcmp (chrom, s->chrom) == 0) return s; }
	return NULL;
	}

static void read_chromosome_lengths (char* filename)      /* genodsp.c:728-814 */
	{
	char line[1001];
	u32  lineNumber = 0;
	FILE* f = fopen (filename, "rt");
	if (f == NULL) { fprintf (stderr, "can't open \"%s\" for reading\n", filename);  exit (EXIT_FAILURE); }
	while (fgets (line, sizeof(line), f) != NULL)
		{
		lineNumber++;
		size_t len = strlen (line);
		if ((len == sizeof(line)-1) && (line[len-1] != '\n'))
			{ fprintf (stderr, "problem at line %u, line is longer than internal buffer\n", lineNumber);  exit (EXIT_FAILURE); }
		char* scan = skip_whitespace (line);
		if ((*scan == 0) || (*scan == '#')) continue;
		char* chrom = line;
		char* mark = skip_darkspace (chrom);
		scan = skip_whitespace (mark);
		if (*mark != 0) *mark = 0;
		if (*scan == 0)
			{ fprintf (stderr, "problem at line %u, line contains no chromosome length\n", lineNumber);  exit (EXIT_FAILURE); }
		char* field = scan;
		mark = skip_darkspace (scan);
		if (*mark != 0) *mark = 0;
		if (!add_chromosome_spec (chrom, 0, (u32) string_to_u32 (field)))
			{ fprintf (stderr, "problem at line %u, chromosome \"%s\" appears more than once\n", lineNumber, chrom);  exit (EXIT_FAILURE); }
		}
	fclose (f);
	}

static int longest_first (const void* a, const void* b)
	{
	u32 x = (*(spec* const*) a)->length, y = (*(spec* const*) b)->length;
	return (x < y) - (x > y);
	}

static void sort_chromosomes_by_length (void)             /* genodsp.c:1113-1145 */
	{
	int n = 0;
	for (spec* s=chromsOfInterest ; s!=NULL ; s=s->next) n++;
	chromsSorted = (spec**) malloc ((n+1) * sizeof(spec*));
	if (chromsSorted == NULL) { fprintf (stderr, "out of memory\n");  exit (EXIT_FAILURE); }
	n = 0;
	for (spec* s=chromsOfInterest ; s!=NULL ; s=s->next) chromsSorted[n++] = s;
	chromsSorted[n] = NULL;
	qsort (chromsSorted, n, sizeof(spec*), longest_first);
	}

/* ------------------------------------------------------------ device services */
void select_device_of (spec* s)
	{
	int d = ((xspec*) s)->device;
	if (d != currentDevice) { check_gdsp (use_device (d), "select device");  currentDevice = d; }
	}

void* op_stream (void) { return devs[currentDevice].stream; }

static spec* vector_spec (char* vName)           /* the stretch being processed, else the chromosome of that name */
	{
	if ((activeSpec != NULL) && (strcmp (activeSpec->chrom, vName) == 0)) return activeSpec;
	return find_chromosome_spec (vName);
	}

static void ensure_partners (void);   /* the partners' arena: waits for the helper thread, or allocates now (below) */

valtype* partner_vector (char* vName)
	{
	spec* s = vector_spec (vName);
	if (s == NULL) return NULL;
	ensure_partners ();
	return ((xspec*) s)->partner;
	}

valtype* partner_of (spec* s) { ensure_partners ();  return ((xspec*) s)->partner; }

/* the output an operator wrote into the partner becomes the signal; what was the signal is nobody's data any more
 * (GDSP_POISON: and is overwritten to prove it -- a later kernel that still reads it changes the output) */
void flip_spec (spec* s)
	{
	ensure_partners ();
	valtype* t = s->valVector;  s->valVector = ((xspec*) s)->partner;  ((xspec*) s)->partner = t;
	valtype poison;
	if (gdsp_poison (&poison))
		{
		select_device_of (s);
		check_gdsp (gdsp_fill (t, s->length, poison, op_stream ()), "poison the partner");
		}
	}

void flip_vector (char* vName)
	{
	spec* s = vector_spec (vName);
	if (s != NULL) flip_spec (s);
	}

valtype* get_scratch_vector (void)                        /* genodsp.c:1904-1940, on the current device */
	{
	devstate* d = &devs[currentDevice];
	for (int i=0 ; i<4 ; i++)
		{
		if (d->scratchInUse[i]) continue;
		if (d->scratch[i] == NULL)
			{
			/* longest local chromosome, but never less than what the select histograms need */
			size_t n = (d->maxLength > 16384)? d->maxLength : 16384;
			check_gdsp (gdsp_malloc ((void**) &d->scratch[i], (n + 2) * sizeof(valtype)), "allocate scratch vector");
			}
		d->scratchInUse[i] = true;
		return d->scratch[i];
		}
	fprintf (stderr, "[%s] internal error: out of scratch vectors\n", programName);
	exit (EXIT_FAILURE);
	}

/* workspace for the long-window forms: gdsp_long_window_work(maxLength) bytes, once per device */
void* long_window_workspace (size_t* bytes)
	{
	static void* work[64];
	devstate* d = &devs[currentDevice];
	*bytes = gdsp_long_window_work (d->maxLength);
	if (work[currentDevice] == NULL) check_gdsp (gdsp_malloc (&work[currentDevice], *bytes), "allocate long-window workspace");
	return work[currentDevice];
	}

/* general per-device workspace that only ever grows (clump's scans) */
void* device_workspace (size_t bytes)
	{
	static void*  work[64];
	static size_t have[64];
	if (have[currentDevice] < bytes)
		{
		if (work[currentDevice] != NULL)
			{
			check_gdsp (gdsp_stream_sync (devs[currentDevice].stream), "synchronise");
			gdsp_free (work[currentDevice]);
			}
		check_gdsp (gdsp_malloc (&work[currentDevice], bytes), "allocate operator workspace");
		have[currentDevice] = bytes;
		}
	return work[currentDevice];
	}

void release_scratch_vector (valtype* v)
	{
	devstate* d = &devs[currentDevice];
	for (int i=0 ; i<4 ; i++) { if (d->scratch[i] == v) { d->scratchInUse[i] = false;  return; } }
	}

s32* get_scratch_ints (void)                              /* genodsp.c:1943-1979, on the current device */
	{
	devstate* d = &devs[currentDevice];
	for (int i=0 ; i<4 ; i++)
		{
		if (d->scratchIntsInUse[i]) continue;
		if (d->scratchInts[i] == NULL)
			{
			size_t n = (d->maxLength > 16384)? d->maxLength : 16384;
			check_gdsp (gdsp_malloc ((void**) &d->scratchInts[i], (n + 4) * sizeof(s32)), "allocate scratch ints");
			/* the reference callocs: a fresh vector reads as zeros */
			check_gdsp (gdsp_memset (d->scratchInts[i], 0, (n + 4) * sizeof(s32), d->stream), "clear scratch ints");
			}
		d->scratchIntsInUse[i] = true;
		return d->scratchInts[i];
		}
	fprintf (stderr, "[%s] internal error: out of scratch int vectors\n", programName);
	exit (EXIT_FAILURE);
	}

void release_scratch_ints (s32* v)
	{
	devstate* d = &devs[currentDevice];
	for (int i=0 ; i<4 ; i++) { if (d->scratchInts[i] == v) { d->scratchIntsInUse[i] = false;  return; } }
	}

int valtype_ascending (const void* v1, const void* v2)   /* genodsp.c:2262-2270; host arrays only */
	{
	const valtype a = *(const valtype*) v1, b = *(const valtype*) v2;
	return (a > b) - (a < b);
	}

void sync_all_devices (void)
	{
	for (int d=0 ; d<numDevices ; d++)
		{
		check_gdsp (use_device (d), "select device");
		check_gdsp (gdsp_stream_sync (devs[d].stream), "synchronise");
		}
	check_gdsp (use_device (currentDevice), "select device");
	}

/* ---- the one collective: percentile's counts and invert's extremes over the devices in use.
 * One process drives all GPUs, so the communicator is RCCL's single-process clique (ncclCommInitAll inside
 * gdsp_comm_create) and the all-reduces run in HBM on each device's stream.  --reduce=host keeps the sums on the
 * host instead; that is also what happens when shards share a GPU (GDSP_OVERSUBSCRIBE_GPUS: RCCL wants one
 * rank per GPU) and, by default, with a single device (nothing to reduce; --reduce=rccl still goes through a
 * one-rank communicator, which is how the one-GPU test box exercises this path).  The communicator is made on the
 * first percentile / invert, so pipelines without them never load RCCL; when RCCL cannot be loaded or initialised the
 * default (--reduce not given) falls back to host sums with a warning, an explicit --reduce=rccl stops. */
static void ensure_device_comm (void)
	{
	static int tried = false;
	if (tried) return;                               /* made on the first percentile / invert: a pipeline without them never loads RCCL */
	tried = true;
	int wantRccl = (reduceHow == reduce_rccl) || ((reduceHow == reduce_auto) && (numDevices > 1));
	if (!wantRccl) return;
	if (numDevices > physicalDevices)
		{
		if (reduceHow == reduce_rccl)
			{ fprintf (stderr, "[%s] --reduce=rccl needs one GPU per shard (%d shards, %d GPUs)\n", programName, numDevices, physicalDevices);  exit (EXIT_FAILURE); }
		return;
		}
	int devices[64];
	for (int d=0 ; d<numDevices ; d++) devices[d] = d % physicalDevices;
	int rc = gdsp_comm_create (&deviceComm, devices, numDevices);
	if ((rc != GDSP_OK) && (reduceHow == reduce_auto))
		{
		/* no loadable RCCL, or its initialisation failed: the counts are a few KiB, the host can add them */
		fprintf (stderr, "[%s] warning: RCCL is not usable (%s); percentile / invert add their counts on the host (--reduce=host)\n",
		         programName, gdsp_last_error ());
		deviceComm = NULL;
		return;
		}
	check_gdsp (rc, "create the RCCL communicator");
	check_gdsp (gdsp_percentiles_use_comm (deviceComm), "hand the communicator to percentile");
	if (trackOperations)
		{
		int version = 0;
		check_gdsp (gdsp_comm_rccl_version (&version), "RCCL version");
		fprintf (stderr, "reduce(rccl %d over %d device%s)\n", version, numDevices, (numDevices == 1)? "" : "s");
		}
	}

gdsp_reduce_fn reduce_over_devices (void** ctx) { ensure_device_comm ();  *ctx = NULL;  return NULL; }   /* (the communicator, when there is one, is inside the library) */

/* smallest and largest value of the whole genome (invert, add.c:909-923): every device folds its chromosomes into
 * three doubles of its own, the devices' results meet in an RCCL all-reduce (min / max) or on the host */
void genome_extremes (valtype* lo, valtype* hi)
	{
	static valtype* acc[64];
	valtype*        accs[64];
	void*           streams[64];
	ensure_device_comm ();
	for (int d=0 ; d<numDevices ; d++)
		{
		check_gdsp (use_device (d), "select device");
		if (acc[d] == NULL) check_gdsp (gdsp_malloc ((void**) &acc[d], 4 * sizeof(valtype)), "allocate accumulators");
		check_gdsp (gdsp_minmax_init (acc[d], devs[d].stream), "genome extremes");
		accs[d] = acc[d];  streams[d] = devs[d].stream;
		}
	sigpart* parts;
	int nparts = signal_parts (&parts);
	for (int i=0 ; i<nparts ; i++)
		{
		int d = ((xspec*) parts[i].s)->device;
		check_gdsp (use_device (d), "select device");
		check_gdsp (gdsp_minmax_update (parts[i].v, parts[i].n, 1, -DBL_MAX, DBL_MAX, acc[d], devs[d].stream), "genome extremes");
		}
	*lo = DBL_MAX;  *hi = -DBL_MAX;
	if (deviceComm != NULL)
		{
		valtype* his[64];
		for (int d=0 ; d<numDevices ; d++) his[d] = accs[d] + 1;
		check_gdsp (gdsp_comm_allreduce_f64 (deviceComm, accs, 1, /*min*/ 1, streams), "all-reduce the minimum");
		check_gdsp (gdsp_comm_allreduce_f64 (deviceComm, his,  1, /*max*/ 2, streams), "all-reduce the maximum");
		}
	for (int d=0 ; d<numDevices ; d++)
		{
		valtype r[3];
		check_gdsp (use_device (d), "select device");
		check_gdsp (gdsp_memcpy_d2h (r, acc[d], sizeof(r), devs[d].stream), "fetch extremes");
		check_gdsp (gdsp_stream_sync (devs[d].stream), "synchronise");
		if ((deviceComm != NULL) && (d > 0)) continue;                 /* device 0 holds the reduced pair already */
		if (r[0] < *lo) *lo = r[0];
		if (r[1] > *hi) *hi = r[1];
		}
	check_gdsp (use_device (currentDevice), "select device");
	}

int device_count_in_use (void) { return numDevices; }
int device_index_of (spec* s)  { return ((xspec*) s)->device; }
int physical_device_of (spec* s) { return ((xspec*) s)->device % physicalDevices; }   /* the GPU a logical shard runs on */

/* deal chromosomes to devices longest-first onto the least loaded (LPT), then allocate */
static void deal_chromosomes (void)
	{
	u64 load[64] = { 0 };
	for (int i=0 ; chromsSorted[i]!=NULL ; i++)
		{
		int best = 0;
		for (int d=1 ; d<numDevices ; d++) { if (load[d] < load[best]) best = d; }
		((xspec*) chromsSorted[i])->device = best;
		load[best] += chromsSorted[i]->length;
		if (chromsSorted[i]->length > devs[best].maxLength) devs[best].maxLength = chromsSorted[i]->length;
		}
	}

/* ---- allocation: ONE arena per device for the vectors and one for their partners, each a
 * single gdsp_malloc, every vector sub-allocated at 256 bytes.  The arenas are made by a helper thread while the main
 * thread reads and parses stdin (host/ingest.c produces records that do not depend on the device; the first batch of
 * intervals to be applied waits for the vectors): start-up of the HIP runtime and the allocation -- seconds when the
 * process before has just given 49 GB back -- run beside the parse instead of in front of it.  The partners' arena is
 * made only when the parsed pipeline holds an out-of-place operator (pipeline_wants_partners), behind the vectors', and is
 * waited for by the first operator that asks for a partner; an operator that asks without having been foreseen (a
 * plug-in) gets it made on the spot.  Reference: genodsp.c:865-893 (allocate every vector, then read stdin). */
#define ARENA_ALIGN 256
static size_t arena_room (u32 length)
	{ return ((((size_t) length + 2) * sizeof(valtype)) + ARENA_ALIGN - 1) & ~(size_t) (ARENA_ALIGN - 1); }

static pthread_t       allocThread;
static pthread_mutex_t allocLock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t  allocCond = PTHREAD_COND_INITIALIZER;
static int    allocThreaded  = false;                 /* a helper thread was started (and not yet joined) */
static int    vectorsReady   = false, partnersReady = false, partnersPlanned = false;
static double allocStartupMs = 0, allocVectorsMs = 0, allocPartnersMs = 0, allocWaitedMs = 0;
static double now_ms (void);

/* one arena on every device for `which` (0: the vectors, 1: the partners) of the whole chromosomes and of the stretches */
static void make_arenas (int which)
	{
	for (int d=0 ; d<numDevices ; d++)
		{
		size_t bytes = 0;
		for (int i=0 ; chromsSorted[i]!=NULL ; i++)
			{ if (((xspec*) chromsSorted[i])->device == d) bytes += arena_room (chromsSorted[i]->length); }
		for (int i=0 ; i<numPieces ; i++)
			{ if (pieces[i].x.device == d) bytes += arena_room (pieces[i].x.pub.length); }
		if (bytes == 0) continue;
		char* base = NULL;
		check_gdsp (use_device (d), "select device");
		check_gdsp (gdsp_malloc ((void**) &base, bytes), which? "allocate the partners' arena" : "allocate the vectors' arena");
		for (int i=0 ; chromsSorted[i]!=NULL ; i++)
			{
			spec*  s = chromsSorted[i];
			xspec* x = (xspec*) s;
			if (x->device != d) continue;
			if (which) x->partner = (valtype*) base;
			else
				{
				if (trackOperations) tracking_report ("allocate(%s / %s bytes)\n", s->chrom, ucommatize (s->length));
				s->valVector = (valtype*) base;
				check_gdsp (gdsp_fill (s->valVector, s->length, 0.0, devs[d].stream), "clear chromosome vector");
				}
			base += arena_room (s->length);
			}
		for (int i=0 ; i<numPieces ; i++)
			{
			piece* p = &pieces[i];
			if (p->x.device != d) continue;
			if (which) p->x.partner = (valtype*) base;  else p->x.pub.valVector = (valtype*) base;
			base += arena_room (p->x.pub.length);
			}
		}
	}

static void* allocate_worker (void* arg)
	{
	(void) arg;
	const double t0 = now_ms ();
	int available = 0;
	check_gdsp (gdsp_device_count (&available), "count GPUs");
	if (available < 1) { fprintf (stderr, "[%s] no GPU visible\n", programName);  exit (EXIT_FAILURE); }
	physicalDevices = available;
	if ((available < numDevices) && (getenv ("GDSP_OVERSUBSCRIBE_GPUS") == NULL))
		{
		fprintf (stderr, "[%s] %d GPU(s) requested, %d visible (set GDSP_OVERSUBSCRIBE_GPUS=1 to run the\n"
		                 "%d shards on the visible GPUs anyway)\n", programName, numDevices, available, numDevices);
		exit (EXIT_FAILURE);
		}
	if ((reduceHow == reduce_rccl) && (numDevices > physicalDevices))
		{ fprintf (stderr, "[%s] --reduce=rccl needs one GPU per shard (%d shards, %d GPUs)\n", programName, numDevices, physicalDevices);  exit (EXIT_FAILURE); }
	for (int d=0 ; d<numDevices ; d++)
		{
		check_gdsp (use_device (d), "select device");
		check_gdsp (gdsp_stream_create (&devs[d].stream), "create stream");
		}
	const double t1 = now_ms ();
	make_arenas (0);
	if (trackOperations) tracking_report ("allocate(--done--)\n");
	for (int d=0 ; d<numDevices ; d++)                        /* (the vectors read as zeros before anybody is told they exist) */
		{ check_gdsp (use_device (d), "select device");  check_gdsp (gdsp_stream_sync (devs[d].stream), "synchronise"); }
	const double t2 = now_ms ();
	pthread_mutex_lock (&allocLock);
	allocStartupMs = t1 - t0;  allocVectorsMs = t2 - t1;
	vectorsReady = true;
	pthread_cond_broadcast (&allocCond);
	pthread_mutex_unlock (&allocLock);
	if (partnersPlanned)
		{
		make_arenas (1);
		const double t3 = now_ms ();
		pthread_mutex_lock (&allocLock);
		allocPartnersMs = t3 - t2;
		partnersReady = true;
		pthread_cond_broadcast (&allocCond);
		pthread_mutex_unlock (&allocLock);
		}
	check_gdsp (use_device (0), "select device");
	return NULL;
	}

/* the main thread may touch the devices from here on (it has not so far: its current device is the runtime's default, 0) */
static int vectorsSeen = false, partnersSeen = false;   /* the main thread's own copies: the shared flags are read under the lock */

static void wait_for_vectors (void)
	{
	if (vectorsSeen) return;
	const double t0 = now_ms ();
	pthread_mutex_lock (&allocLock);
	while (!vectorsReady) pthread_cond_wait (&allocCond, &allocLock);
	pthread_mutex_unlock (&allocLock);
	vectorsSeen = true;
	allocWaitedMs += now_ms () - t0;
	}

static void ensure_partners (void)
	{
	if (partnersSeen) return;
	wait_for_vectors ();
	if (partnersPlanned)
		{
		const double t0 = now_ms ();
		pthread_mutex_lock (&allocLock);
		while (!partnersReady) pthread_cond_wait (&allocCond, &allocLock);
		pthread_mutex_unlock (&allocLock);
		partnersSeen = true;
		allocWaitedMs += now_ms () - t0;
		return;
		}
	/* nobody foresaw an out-of-place operator (a plug-in's): the helper thread has finished with the devices, make them now */
	if (allocThreaded) { pthread_join (allocThread, NULL);  allocThreaded = false; }
	const double t0 = now_ms ();
	make_arenas (1);
	check_gdsp (use_device (currentDevice), "select device");
	allocPartnersMs = now_ms () - t0;
	partnersPlanned = partnersReady = partnersSeen = true;
	}

/* does the parsed pipeline hold an operator that writes into a partner?  The built-in in-place operators are named;
 * anything else (the out-of-place ones, operators linked in through GDSP_EXTRA_OPERATORS) counts as wanting one */
static int pipeline_wants_partners (void)
	{
	static const opfunc_apply inPlace[] =
		{ op_window_sum_apply, op_cumulative_sum_apply, op_add_apply, op_subtract_apply, op_add_constant_apply, op_invert_apply,
		  op_multiply_apply, op_divide_apply, op_absolute_value_apply, op_clip_apply, op_erase_apply, op_binarize_apply,
		  op_input_apply, op_output_apply, op_show_variables_apply, op_mask_apply, op_mask_not_apply, op_or_apply, op_and_apply,
		  op_min_with_apply, op_max_with_apply, op_map_apply, op_min_in_interval_apply, op_max_in_interval_apply,
		  op_clump_apply, op_skimp_apply };
	if (shardBases) return true;                               /* (stretches and their runs: not worth a second rule) */
	for (dspop* op=pipeline ; op!=NULL ; op=op->next)
		{
		int known = false;
		for (size_t i=0 ; i<sizeof(inPlace)/sizeof(inPlace[0]) ; i++) { if (op->funcApply == inPlace[i]) known = true; }
		if (op->funcApply == op_percentile_apply)                  /* (fused with the binarize behind it, it writes partners) */
			known = !(fuseChains && (op->next != NULL) && (op->next->funcApply == op_binarize_apply));
		if (!known) return true;
		}
	return false;
	}

/* a complaint about the input ends the run (exit) while the helper thread may still be inside the HIP runtime, whose own
 * exit handlers would then tear down what that thread is using: this handler, registered after the runtime's and therefore
 * run before them, lets the helper finish first (when it is the helper itself that gives up -- no GPU -- there is nobody
 * to wait for) */
static void join_allocation_at_exit (void)
	{
	if (allocThreaded && !pthread_equal (pthread_self (), allocThread)) { pthread_join (allocThread, NULL);  allocThreaded = false; }
	}

/* start the allocation; beside the parse of stdin unless the progress lines are wanted in the reference's order */
static void allocate_vectors (void)
	{
	partnersPlanned = pipeline_wants_partners ();
	const char* e = getenv ("GDSP_ALLOC_THREAD");
	if (trackOperations || ((e != NULL) && (e[0] == '0')))
		{ allocate_worker (NULL);  return; }
	gdsp_version ();                                           /* (the library, and with it the runtime's exit handlers, is loaded by now) */
	atexit (join_allocation_at_exit);
	if (pthread_create (&allocThread, NULL, allocate_worker, NULL) != 0)
		{ allocate_worker (NULL);  return; }
	allocThreaded = true;
	}

/* ------------------------------------------------- --sharding=bases: stretches ---- */
#define HALO_LIMIT (1u << 20)       /* a run reaching further than this is given whole chromosomes */

/* reach of the maximal run [firstOp, stopOp) when every operator in it can work on a stretch: the reaches add up
 * along the chain, plus one base per operator (dilate treats position 0 of its vector specially, morphology.c:925) */
static int run_reach (dspop* firstOp, dspop* stopOp, u32* left, u32* right, int* pointwise)
	{
	u64 L = 0, R = 0;
	int flat = true;                                  /* every operator of the run looks at its own base only */
	for (dspop* op=firstOp ; op!=stopOp ; op=op->next)
		{
		u32 l, r;
		if (!op_reach (op, &l, &r)) return false;
		if ((l | r) != 0) flat = false;
		L += (u64) l + 1;  R += (u64) r + 1;
		}
	if ((L > HALO_LIMIT) || (R > HALO_LIMIT)) return false;
	*left = (u32) L;  *right = (u32) R;
	if (pointwise != NULL) *pointwise = flat;
	return true;
	}

/* equal shares of the genome's bases per device: the chromosomes, longest first, laid end to end and cut at
 * total*d/N; a cut that would leave a stub is moved to the chromosome's end */
static void plan_pieces (void)
	{
	u64 total = 0;
	int nchrom = 0;
	for (dspop* firstOp=pipeline ; firstOp!=NULL ; )
		{
		dspop* stopOp;
		u32 l, r;
		for (stopOp=firstOp ; stopOp!=NULL ; stopOp=stopOp->next) { if (stopOp->atRandom) break; }
		if ((stopOp != firstOp) && run_reach (firstOp, stopOp, &l, &r, NULL))
			{ if (l > haloCap) haloCap = l;  if (r > haloCap) haloCap = r; }
		firstOp = (stopOp == NULL)? NULL : stopOp->next;
		}
	for (int i=0 ; chromsSorted[i]!=NULL ; i++) { total += chromsSorted[i]->length;  nchrom++; }
	const u64 stub = 4 * (u64) haloCap + 64;
	pieces = (piece*) calloc (nchrom + numDevices, sizeof(piece));
	if (pieces == NULL) { fprintf (stderr, "out of memory\n");  exit (EXIT_FAILURE); }
	u64 at = 0;                                   /* bases before the current chromosome */
	int d  = 0;
	for (int i=0 ; chromsSorted[i]!=NULL ; i++)
		{
		spec* s = chromsSorted[i];
		u32   a = 0;
		while (a < s->length)
			{
			u64 cut = (d+1 < numDevices)? total * (u64) (d+1) / (u64) numDevices : total;    /* end of device d's share */
			u32 b;
			if (cut >= at + s->length) b = s->length;                  /* the share runs past this chromosome */
			else if (cut <= at + a)    { d++;  continue; }             /* the share is full */
			else
				{
				b = (u32) (cut - at) & ~1u;                            /* cuts at even bases: what a stretch answers for starts 16-byte aligned (the dense kernels) */
				if (b - a < stub)         { d++;  continue; }          /* only a stub left of this share: next device */
				if (s->length - b < stub) b = s->length;               /* only a stub left of the chromosome: take it too */
				}
			piece* p = &pieces[numPieces++];
			p->whole = s;  p->ownStart = a;  p->ownEnd = b;
			p->extStart = ((a > haloCap)? a - haloCap : 0) & ~1u;
			u32 extEnd  = (s->length - b > haloCap)? b + haloCap : s->length;
			p->x.pub.chrom  = s->chrom;
			p->x.pub.start  = s->start + p->extStart;
			p->x.pub.length = extEnd - p->extStart;
			p->x.device     = d;
			if (p->x.pub.length > devs[d].maxLength) devs[d].maxLength = p->x.pub.length;
			a = b;
			if ((at + b >= cut) && (d+1 < numDevices)) d++;
			}
		at += s->length;
		}
	}

/* --shards=show: who gets what, and how even that is (no GPU needed: printed before any device is touched) */
static void show_shards (void)
	{
	u64 load[64] = { 0 }, total = 0, most = 0;
	for (int d=0 ; d<numDevices ; d++)
		{
		fprintf (stderr, "device %d:", d);
		if (shardBases)
			{
			for (int i=0 ; i<numPieces ; i++)
				{
				if (pieces[i].x.device != d) continue;
				fprintf (stderr, " %s:%u-%u", pieces[i].whole->chrom, pieces[i].ownStart, pieces[i].ownEnd);
				load[d] += pieces[i].x.pub.length;                 /* what it computes: halos included */
				total   += pieces[i].ownEnd - pieces[i].ownStart;
				}
			}
		else
			{
			for (int i=0 ; chromsSorted[i]!=NULL ; i++)
				{
				if (((xspec*) chromsSorted[i])->device != d) continue;
				fprintf (stderr, " %s", chromsSorted[i]->chrom);
				load[d] += chromsSorted[i]->length;  total += chromsSorted[i]->length;
				}
			}
		fprintf (stderr, " = %s bases\n", ucommatize (load[d]));
		if (load[d] > most) most = load[d];
		}
	fprintf (stderr, "sharding=%s halo=%u makespan efficiency %.4f\n", shardBases? "bases" : "chromosomes", haloCap,
	         (most == 0)? 1.0 : (double) total / ((double) most * numDevices));
	}

static void copy_bases (valtype* dst, int dstDev, const valtype* src, int srcDev, u32 count)
	{
	if (count == 0) return;
	check_gdsp (use_device (dstDev), "select device");
	check_gdsp (gdsp_memcpy_peer (dst, dstDev % physicalDevices, src, srcDev % physicalDevices, (size_t) count * sizeof(valtype),
	                              devs[dstDev].stream), "copy between devices");
	}

/* whole chromosomes -> stretches (halos included, so they are fresh) */
static void to_pieces (void)
	{
	if (signalInPieces) return;
	sync_all_devices ();
	for (int i=0 ; i<numPieces ; i++)
		{
		piece* p = &pieces[i];
		copy_bases (p->x.pub.valVector, p->x.device, p->whole->valVector + p->extStart, ((xspec*) p->whole)->device, p->x.pub.length);
		}
	sync_all_devices ();
	signalInPieces = true;  halosFresh = true;
	}

int signal_in_whole_chromosomes (void) { return !shardBases && !signalInPieces; }

/* stretches -> whole chromosomes: every stretch returns the bases it answers for */
void to_whole (void)
	{
	if (!signalInPieces) return;
	sync_all_devices ();
	for (int i=0 ; i<numPieces ; i++)
		{
		piece* p = &pieces[i];
		copy_bases (p->whole->valVector + p->ownStart, ((xspec*) p->whole)->device,
		            p->x.pub.valVector + (p->ownStart - p->extStart), p->x.device, p->ownEnd - p->ownStart);
		}
	sync_all_devices ();
	signalInPieces = false;
	}

/* every halo base is fetched from the stretch that answers for it (the only GPU-to-GPU traffic of a sharded run) */
static void refresh_halos (void)
	{
	if (halosFresh) return;
	sync_all_devices ();
	for (int i=0 ; i<numPieces ; i++)
		{
		piece* p = &pieces[i];
		u32 extEnd = p->extStart + p->x.pub.length;
		for (int j=0 ; j<numPieces ; j++)
			{
			piece* q = &pieces[j];
			if ((q == p) || (q->whole != p->whole)) continue;
			for (int side=0 ; side<2 ; side++)
				{
				u32 lo = side? p->ownEnd : p->extStart, hi = side? extEnd : p->ownStart;
				if (q->ownStart > lo) lo = q->ownStart;
				if (q->ownEnd   < hi) hi = q->ownEnd;
				if (lo >= hi) continue;
				copy_bases (p->x.pub.valVector + (lo - p->extStart), p->x.device,
				            q->x.pub.valVector + (lo - q->extStart), q->x.device, hi - lo);
				}
			}
		}
	sync_all_devices ();
	halosFresh = true;
	}

/* the signal as the stretches of it that someone answers for, wherever it lives now (percentile, invert) */
int signal_parts (sigpart** out)
	{
	static sigpart* parts = NULL;
	int n = 0;
	if (parts == NULL)
		{
		int cap = numPieces;
		for (int i=0 ; chromsSorted[i]!=NULL ; i++) cap++;
		parts = (sigpart*) calloc (cap + 1, sizeof(sigpart));
		if (parts == NULL) { fprintf (stderr, "out of memory\n");  exit (EXIT_FAILURE); }
		}
	if (signalInPieces)
		{
		for (int i=0 ; i<numPieces ; i++)
			{
			piece* p = &pieces[i];
			parts[n].s = &p->x.pub;  parts[n].v = p->x.pub.valVector + (p->ownStart - p->extStart);
			parts[n].n = p->ownEnd - p->ownStart;  parts[n].first = p->ownStart;
			parts[n].base = p->x.pub.valVector;  parts[n].baseLen = p->x.pub.length;
			n++;
			}
		}
	else
		{
		for (int i=0 ; chromsSorted[i]!=NULL ; i++)
			{
			spec* s = chromsSorted[i];
			parts[n].s = s;  parts[n].v = s->valVector;  parts[n].n = s->length;  parts[n].first = 0;
			parts[n].base = s->valVector;  parts[n].baseLen = s->length;
			n++;
			}
		}
	*out = parts;
	return n;
	}

/* ---------------------------------------------------------------- text ingest */
/* (read_interval, the line reader, is ingest.c) */

/* Pending intervals per chromosome, in file order, flushed to the device in
 * batches: pinned staging -> hipMemcpyAsync -> gdsp_apply_intervals. */
#define BATCH_INTERVALS (8*1024*1024)
u64 ib_batch_limit (void)                     /* GDSP_BATCH_INTERVALS=<n>: a test hook that forces many small batches */
	{
	static u64 limit = 0;
	if (limit == 0)
		{
		const char* e = getenv ("GDSP_BATCH_INTERVALS");
		limit = ((e != NULL) && (atoll (e) > 0))? (u64) atoll (e) : BATCH_INTERVALS;
		}
	return limit;
	}

typedef struct pending { u32* start;  u32* end;  valtype* val;  u32 count, cap; } pending;
static pending* pend = NULL;          /* indexed like chromsSorted */
static int      numChroms = 0;
static u64      pendTotal = 0;

/* staging shared by all flushes (pinned) and its device mirror, per device */
typedef struct staging
	{
	size_t   ivCap, offCap, listCap;
	u32     *h_start, *h_end, *h_off, *h_list;   valtype* h_val;
	u32     *d_start, *d_end, *d_off, *d_list;   valtype* d_val;
	void*    d_over;   size_t overCap;            /* per-interval records of minover / maxover */
	} staging;
static staging stage[64];

static int sorted_index (spec* s)
	{ for (int i=0 ; chromsSorted[i]!=NULL ; i++) { if (chromsSorted[i] == s) return i; }  return -1; }

void ib_begin (void)
	{
	if (pend == NULL)
		{
		for (numChroms=0 ; chromsSorted[numChroms]!=NULL ; numChroms++) ;
		pend = (pending*) calloc (numChroms, sizeof(pending));
		}
	for (int i=0 ; i<numChroms ; i++) pend[i].count = 0;
	pendTotal = 0;
	}

void ib_add (spec* s, u32 start, u32 end, valtype val)
	{
	pending* p = &pend[sorted_index (s)];
	if (p->count == p->cap)
		{
		p->cap   = (p->cap == 0)? 1024 : 2*p->cap;
		p->start = (u32*)     realloc (p->start, p->cap * sizeof(u32));
		p->end   = (u32*)     realloc (p->end,   p->cap * sizeof(u32));
		p->val   = (valtype*) realloc (p->val,   p->cap * sizeof(valtype));
		if ((p->start == NULL) || (p->end == NULL) || (p->val == NULL))
			{ fprintf (stderr, "out of memory buffering intervals\n");  exit (EXIT_FAILURE); }
		}
	p->start[p->count] = start;  p->end[p->count] = end;  p->val[p->count] = val;
	p->count++;
	pendTotal++;
	intervalsRead++;
	}

u64 ib_pending (void) { return pendTotal; }

static void grow (void** h, void** d, size_t* cap, size_t want, size_t elem)
	{
	if (want <= *cap) return;
	size_t n = (*cap == 0)? 4096 : *cap;
	while (n < want) n *= 2;
	if (*h != NULL) check_gdsp (gdsp_host_free (*h), "free staging");
	if (*d != NULL) check_gdsp (gdsp_free (*d), "free staging");
	check_gdsp (gdsp_host_alloc (h, n * elem), "allocate pinned staging");
	check_gdsp (gdsp_malloc (d, n * elem), "allocate device staging");
	*cap = n;
	}

/* stage one chromosome's pending intervals on its device and return the device arrays */
static void stage_chromosome (int ci, staging** out)
	{
	spec*    s  = chromsSorted[ci];
	pending* p  = &pend[ci];
	wait_for_vectors ();                                       /* (the first batch of a run: the arenas may still be in the making) */
	select_device_of (s);
	staging* st = &stage[currentDevice];
	void*    stream = op_stream ();
	u32      ntiles = (u32) (((u64) s->length + gdsp_interval_tile () - 1) / gdsp_interval_tile ());
	u64      listLen = 0;

	check_gdsp (gdsp_stream_sync (stream), "synchronise before restaging");   /* the previous flush is done with the buffers */
	/* three arrays share ivCap: grow them together */
	if ((size_t) p->count + 1 > st->ivCap)
		{
		size_t cap = st->ivCap, c2 = st->ivCap, c3 = st->ivCap;
		grow ((void**) &st->h_start, (void**) &st->d_start, &cap, (size_t) p->count + 1, sizeof(u32));
		grow ((void**) &st->h_end,   (void**) &st->d_end,   &c2,  (size_t) p->count + 1, sizeof(u32));
		grow ((void**) &st->h_val,   (void**) &st->d_val,   &c3,  (size_t) p->count + 1, sizeof(valtype));
		st->ivCap = cap;
		}
	grow ((void**) &st->h_off, (void**) &st->d_off, &st->offCap, (size_t) ntiles + 1, sizeof(u32));

	if (p->count != 0)                                     /* (a chromosome no interval has named has no arrays yet) */
		{
		memcpy (st->h_start, p->start, (size_t) p->count * sizeof(u32));
		memcpy (st->h_end,   p->end,   (size_t) p->count * sizeof(u32));
		memcpy (st->h_val,   p->val,   (size_t) p->count * sizeof(valtype));
		}
	check_gdsp (gdsp_bin_intervals (s->length, st->h_start, st->h_end, p->count, st->h_off, NULL, &listLen), "bin intervals");
	grow ((void**) &st->h_list, (void**) &st->d_list, &st->listCap, (size_t) listLen + 1, sizeof(u32));
	check_gdsp (gdsp_bin_intervals (s->length, st->h_start, st->h_end, p->count, st->h_off, st->h_list, &listLen), "bin intervals");

	if (p->count != 0)
		{
		check_gdsp (gdsp_memcpy_h2d (st->d_start, st->h_start, (size_t) p->count * sizeof(u32),     stream), "stage intervals");
		check_gdsp (gdsp_memcpy_h2d (st->d_end,   st->h_end,   (size_t) p->count * sizeof(u32),     stream), "stage intervals");
		check_gdsp (gdsp_memcpy_h2d (st->d_val,   st->h_val,   (size_t) p->count * sizeof(valtype), stream), "stage intervals");
		}
	check_gdsp (gdsp_memcpy_h2d (st->d_off, st->h_off, ((size_t) ntiles + 1) * sizeof(u32), stream), "stage intervals");
	if (listLen != 0)
		check_gdsp (gdsp_memcpy_h2d (st->d_list, st->h_list, (size_t) listLen * sizeof(u32), stream), "stage intervals");
	*out = st;
	}

/* apply and forget the pending intervals; `everyChromosome` also visits chromosomes
 * without intervals (needed when clearing, or for multiply/divide's gap rule) */
void ib_flush_apply (int overlapOp, int clearFlags, valtype missingVal, int everyChromosome)
	{
	for (int ci=0 ; ci<numChroms ; ci++)
		{
		if ((pend[ci].count == 0) && !everyChromosome) continue;
		staging* st;
		stage_chromosome (ci, &st);
		spec* s = chromsSorted[ci];
		check_gdsp (gdsp_apply_intervals (s->valVector, s->length, st->d_start, st->d_end, st->d_val, st->d_off, st->d_list,
		                                  overlapOp, clearFlags, missingVal, op_stream ()), "apply intervals");
		pend[ci].count = 0;
		}
	pendTotal = 0;
	}

void ib_flush_scale (int divide, valtype infinityVal)
	{
	for (int ci=0 ; ci<numChroms ; ci++)
		{
		staging* st;
		stage_chromosome (ci, &st);
		spec* s = chromsSorted[ci];
		check_gdsp (gdsp_scale_intervals (s->valVector, s->length, st->d_start, st->d_end, st->d_val, st->d_off, st->d_list,
		                                  divide, infinityVal, op_stream ()), "scale by intervals");
		pend[ci].count = 0;
		}
	pendTotal = 0;
	}

/* mask / masknot / or / and: every chromosome is visited (the binarise pass and the
 * outside-value rule apply to chromosomes the file never mentions too) */
void ib_flush_mask (int inside, valtype outsideVal, int binarizeFirst)
	{
	for (int ci=0 ; ci<numChroms ; ci++)
		{
		if ((pend[ci].count == 0) && inside && !binarizeFirst) continue;
		staging* st;
		stage_chromosome (ci, &st);
		spec* s = chromsSorted[ci];
		check_gdsp (gdsp_mask_intervals (s->valVector, s->length, st->d_start, st->d_end, st->d_val, st->d_off, st->d_list,
		                                 inside, outsideVal, binarizeFirst, op_stream ()), "mask by intervals");
		pend[ci].count = 0;
		}
	pendTotal = 0;
	}

/* minover / maxover: every chromosome is visited (bases under no interval are filled too) */
void ib_flush_over (int wantMax, valtype fillVal)
	{
	for (int ci=0 ; ci<numChroms ; ci++)
		{
		staging* st;
		stage_chromosome (ci, &st);
		spec* s = chromsSorted[ci];
		size_t need = gdsp_extreme_in_intervals_work (pend[ci].count);
		if (need > st->overCap)
			{
			if (st->d_over != NULL) check_gdsp (gdsp_free (st->d_over), "free workspace");
			check_gdsp (gdsp_malloc (&st->d_over, 2*need), "allocate workspace");
			st->overCap = 2*need;
			}
		check_gdsp (gdsp_extreme_in_intervals (s->valVector, s->length, st->d_start, st->d_end, pend[ci].count,
		                                       st->d_off, st->d_list, wantMax, fillVal, st->d_over, op_stream ()), "extreme in intervals");
		pend[ci].count = 0;
		}
	pendTotal = 0;
	}

/* read_intervals, genodsp.c:1187-1350: same routing, origin, clipping and failure rules;
 * the per-base loops run on the device in file order */
void read_intervals (FILE* f, int valCol, int origin1, int overlapOp, int clear, valtype missingVal)
	{
	char     line[1001], prevChrom[1001];
	char*    chrom;
	spec*    s = NULL;
	u32      start, end, o = origin1? 1 : 0;
	valtype  val;
	int      clearFlags = clear? GDSP_CLEAR_BOTH : 0;

	to_whole ();                                               /* intervals are applied to whole chromosomes */
	if (trackOperations) { for (int i=0 ; chromsSorted[i]!=NULL ; i++) chromsSorted[i]->flag = false; }
	ib_begin ();
	prevChrom[0] = 0;
	while (read_interval (f, line, sizeof(line), valCol, &chrom, &start, &end, &val))
		{
		if (strcmp (chrom, prevChrom) != 0)
			{ s = find_chromosome_spec (chrom);  safe_strncpy (prevChrom, chrom, sizeof(prevChrom)-1); }
		if (s == NULL) continue;
		if (trackOperations && !s->flag) { tracking_report ("input(%s)\n", chrom);  s->flag = true; }

		start -= o;
		u32 adjStart = start, adjEnd = end;
		if (clipToLength)
			{
			if (start > s->start + s->length) adjStart = start = s->start + s->length;
			if (end   > s->start + s->length) adjEnd   = end   = s->start + s->length;
			}
		if (s->start == 0)
			{
			if (end > s->length)
				{
				fprintf (stderr, "%s %d %d is beyond the end of the chromosome (L=%d)\n", chrom, start, end, s->length);
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
		ib_add (s, adjStart, adjEnd, val);
		if (ib_pending () >= ib_batch_limit ())
			{
			ib_flush_apply (overlapOp, clearFlags, missingVal, (clearFlags & GDSP_CLEAR_FILL) != 0);
			clearFlags &= ~GDSP_CLEAR_FILL;            /* later batches keep only the first-touch rule */
			}
		}
	ib_flush_apply (overlapOp, clearFlags, missingVal, (clearFlags & GDSP_CLEAR_FILL) != 0);
	if (trackOperations) tracking_report ("input(--done--)\n");
	}

/* ---------------------------------------------------------------- text output */
/* Output lines are formatted by hand into large buffers: with millions of runs the report is bound
 * by fprintf otherwise.  Same characters as the reference's "%s\t%d\t%d\t%.*f\n" (genodsp.c:1640-1668). */
static char* put_int (char* p, int v)                                   /* %d */
	{
	char digits[12];
	int  n = 0;
	unsigned int u = (v < 0)? 0u - (unsigned int) v : (unsigned int) v;
	if (v < 0) *(p++) = '-';
	do { digits[n++] = (char) ('0' + u % 10);  u /= 10; } while (u != 0);
	while (n > 0) *(p++) = digits[--n];
	return p;
	}

static char* put_u64 (char* p, unsigned long long u, int minDigits)
	{
	char digits[24];
	int  n = 0;
	do { digits[n++] = (char) ('0' + u % 10);  u /= 10; } while (u != 0);
	while (n < minDigits) digits[n++] = '0';
	while (n > 0) *(p++) = digits[--n];
	return p;
	}

static char* put_fixed (char* p, valtype v, int precision)              /* %.*f */
	{
	static const unsigned long long pow10[] = { 1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull,
	                                            100000000ull, 1000000000ull };
	if ((precision >= 0) && (precision <= 9) && (fabs (v) * (double) pow10[precision] < 9.0e18))    /* (a NaN fails) */
		{
		/* v = m * 2^e exactly, so v * 10^precision = m * 10^precision / 2^-e in 128 bits: round that to the
		 * nearest integer, ties to even, which is how printf rounds the exact decimal expansion
		 * (held to snprintf on 20 M values of every kind, tools/check_output_format.c) */
		int e;
		const double    a = fabs (v);
		const long long m = (long long) ldexp (frexp (a, &e), 53);
		e -= 53;
		if (e > -120)
			{
			unsigned __int128 scaled = (unsigned __int128) (unsigned long long) m * pow10[precision];
			unsigned long long q;
			if (e >= 0) q = (unsigned long long) (scaled << e);
			else
				{
				const int shift = -e;
				const unsigned __int128 rem  = scaled & ((((unsigned __int128) 1) << shift) - 1);
				const unsigned __int128 half = ((unsigned __int128) 1) << (shift - 1);
				q = (unsigned long long) (scaled >> shift);
				if ((rem > half) || ((rem == half) && (q & 1))) q++;
				}
			if (signbit (v)) *(p++) = '-';
			p = put_u64 (p, q / pow10[precision], 1);
			if (precision > 0) { *(p++) = '.';  p = put_u64 (p, q % pow10[precision], precision); }
			return p;
			}
		}
	return NULL;                                       /* not a case for the fast path: the caller prints through printf */
	}

/* Millions of runs (a smoothed genome prints one line per covered base: 132 M lines for BASELINE configs[1] on the
 * 12 M-read genome) are formatted by a team of threads: the runs of a chromosome are cut into one stretch per thread,
 * every thread formats its stretch into a buffer of its own with the very functions above, and the buffers are written
 * in order.  An NA line depends only on the run before it, so stretches are independent.  (One thread: 9.1 s for those
 * 132 M lines; the reference's fprintf: minutes.) */
typedef struct fmtjob
	{
	const char* chrom;  u32 chromStart, chromLength, o;
	const u32 *runStart, *runEnd;  const valtype* runVal;
	u32 from, to, count;
	int withVal, precision, uncovered;
	char* buf;  size_t len, cap;  u64 lines;
	} fmtjob;

static void fmt_room (fmtjob* j, size_t need)
	{
	if (j->cap - j->len >= need) return;
	size_t cap = (j->cap == 0)? (1u << 20) : j->cap;
	while (cap - j->len < need) cap *= 2;
	j->buf = (char*) realloc (j->buf, cap);
	if (j->buf == NULL) { fprintf (stderr, "out of memory formatting the output\n");  exit (EXIT_FAILURE); }
	j->cap = cap;
	}

static void fmt_line (fmtjob* j, size_t chromLen, int start, int end, int withVal, valtype v, int na)
	{
	fmt_room (j, chromLen + 96);
	char* p = j->buf + j->len;
	memcpy (p, j->chrom, chromLen);  p += chromLen;  *(p++) = '\t';
	p = put_int (p, start);  *(p++) = '\t';
	p = put_int (p, end);
	if (na) { *(p++) = '\t';  *(p++) = 'N';  *(p++) = 'A'; }
	else if (withVal)
		{
		*(p++) = '\t';
		char* q = put_fixed (p, v, j->precision);
		if (q == NULL)                                 /* any length: measured, then printed in place */
			{
			j->len = (size_t) (p - j->buf);
			int need = snprintf (NULL, 0, valtypeFmtPrec, j->precision, v);
			fmt_room (j, (size_t) need + 8);
			p = j->buf + j->len;
			q = p + snprintf (p, (size_t) need + 1, valtypeFmtPrec, j->precision, v);
			}
		p = q;
		}
	*(p++) = '\n';
	j->len = (size_t) (p - j->buf);
	j->lines++;
	}

static void* fmt_worker (void* arg)
	{
	fmtjob* j = (fmtjob*) arg;
	const size_t chromLen = strlen (j->chrom);
	for (u32 r=j->from ; r<j->to ; r++)
		{
		u32 prevOutputEnd = (r == 0)? 0 : j->chromStart + j->runEnd[r-1];
		u32 outputStart = j->chromStart + j->runStart[r], outputEnd = j->chromStart + j->runEnd[r];
		if ((j->uncovered == uncovered_NA) && (outputStart != prevOutputEnd))
			fmt_line (j, chromLen, (int) (prevOutputEnd + j->o), (int) outputStart, false, 0.0, true);
		fmt_line (j, chromLen, (int) (outputStart + j->o), (int) outputEnd, j->withVal, j->runVal[r], false);
		}
	if ((j->to == j->count) && (j->uncovered == uncovered_NA))                 /* the stretch that ends the chromosome closes it */
		{
		u32 prevOutputEnd = (j->count == 0)? 0 : j->chromStart + j->runEnd[j->count-1];
		if (j->chromStart + j->chromLength != prevOutputEnd)
			fmt_line (j, chromLen, (int) (prevOutputEnd + j->o), (int) (j->chromStart + j->chromLength), false, 0.0, true);
		}
	return NULL;
	}

#define FMT_MAX_THREADS 32
static int fmt_team_size (void)                          /* GDSP_OUTPUT_THREADS=<n>; default: the cores at hand, at most 16 */
	{
	static int n = 0;
	if (n == 0)
		{
		const char* e = getenv ("GDSP_OUTPUT_THREADS");
		long cores = sysconf (_SC_NPROCESSORS_ONLN);
		n = (e != NULL)? atoi (e) : (int) ((cores > 16)? 16 : cores);
		if (n < 1) n = 1;
		if (n > FMT_MAX_THREADS) n = FMT_MAX_THREADS;
		}
	return n;
	}

static void format_runs (FILE* f, spec* s, const u32* runStart, const u32* runEnd, const valtype* runVal, u32 count,
                         int withVal, int precision, int uncovered, u32 o)
	{
	static fmtjob jobs[FMT_MAX_THREADS];                   /* (the buffers are kept from one chromosome to the next) */
	int T = fmt_team_size ();
	if (count < 65536) T = 1;
	pthread_t tid[FMT_MAX_THREADS];
	for (int t=0 ; t<T ; t++)
		{
		fmtjob* j = &jobs[t];
		j->chrom = s->chrom;  j->chromStart = s->start;  j->chromLength = s->length;  j->o = o;
		j->runStart = runStart;  j->runEnd = runEnd;  j->runVal = runVal;  j->count = count;
		j->from = (u32) ((u64) count * (u64) t / (u64) T);  j->to = (u32) ((u64) count * (u64) (t + 1) / (u64) T);
		j->withVal = withVal;  j->precision = precision;  j->uncovered = uncovered;
		j->len = 0;  j->lines = 0;
		if ((t > 0) && (pthread_create (&tid[t], NULL, fmt_worker, j) != 0))
			{ fprintf (stderr, "can't start an output thread\n");  exit (EXIT_FAILURE); }
		}
	fmt_worker (&jobs[0]);
	for (int t=0 ; t<T ; t++)
		{
		if (t > 0) pthread_join (tid[t], NULL);
		if (jobs[t].len != 0) fwrite (jobs[t].buf, 1, jobs[t].len, f);
		linesWritten += jobs[t].lines;
		}
	}

typedef struct reportbuf
	{
	u32*     d_count;  void* d_work;
	u32      cap;
	u32     *d_start, *d_end, *h_start, *h_end;
	valtype *d_val, *h_val;
	} reportbuf;
static reportbuf reportBufs[64];

/* report_intervals, genodsp.c:1561-1691: runs come from the device
 * (gdsp_report_runs), the NA bookkeeping and formatting are done here */
void report_intervals (FILE* f, int precision, int noValues, int collapse, int uncovered, int origin1)
	{
	u32 o = origin1? 1 : 0;
	to_whole ();                                               /* runs are found chromosome by chromosome */
	for (spec* s=chromsOfInterest ; s!=NULL ; s=s->next)
		{
		if (trackOperations) tracking_report ("output(%s)\n", s->chrom);
		select_device_of (s);
		void*      stream = op_stream ();
		reportbuf* rb     = &reportBufs[currentDevice];
		u32        count  = 0;
		/* per device, kept for the run: the count word, the workspace (longest local chromosome), and run arrays that
		 * only grow.  One pass finds and writes the runs when they fit what is there (count + scan + write = 16 B/base);
		 * only a chromosome with more runs than any before it costs a second call. */
		if (rb->d_count == NULL)
			{
			check_gdsp (gdsp_malloc ((void**) &rb->d_count, 16), "allocate run count");
			check_gdsp (gdsp_malloc (&rb->d_work, gdsp_report_runs_work (devs[currentDevice].maxLength)), "allocate run workspace");
			}
		for (int attempt=0 ; attempt<2 ; attempt++)
			{
			if (rb->cap < ((attempt == 0)? 1u << 16 : count))
				{
				u32 want = (attempt == 0)? 1u << 16 : count + count/8 + 1024;
				if (rb->d_start != NULL)
					{
					gdsp_free (rb->d_start);  gdsp_free (rb->d_end);  gdsp_free (rb->d_val);
					gdsp_host_free (rb->h_start);  gdsp_host_free (rb->h_end);  gdsp_host_free (rb->h_val);
					}
				check_gdsp (gdsp_malloc ((void**) &rb->d_start, (size_t) want * sizeof(u32)), "allocate runs");
				check_gdsp (gdsp_malloc ((void**) &rb->d_end,   (size_t) want * sizeof(u32)), "allocate runs");
				check_gdsp (gdsp_malloc ((void**) &rb->d_val,   (size_t) want * sizeof(valtype)), "allocate runs");
				check_gdsp (gdsp_host_alloc ((void**) &rb->h_start, (size_t) want * sizeof(u32)), "allocate runs");
				check_gdsp (gdsp_host_alloc ((void**) &rb->h_end,   (size_t) want * sizeof(u32)), "allocate runs");
				check_gdsp (gdsp_host_alloc ((void**) &rb->h_val,   (size_t) want * sizeof(valtype)), "allocate runs");
				rb->cap = want;
				}
			check_gdsp (gdsp_report_runs (s->valVector, s->length, collapse, uncovered, rb->d_start, rb->d_end, rb->d_val, rb->cap,
			                              rb->d_count, rb->d_work, stream), "find runs");
			check_gdsp (gdsp_memcpy_d2h (&count, rb->d_count, sizeof(u32), stream), "fetch run count");
			check_gdsp (gdsp_stream_sync (stream), "synchronise");
			if (count <= rb->cap) break;                               /* (else: the arrays hold only the first cap runs) */
			}
		u32 *runStart = rb->h_start, *runEnd = rb->h_end;  valtype* runVal = rb->h_val;
		if (count != 0)
			{
			check_gdsp (gdsp_memcpy_d2h (runStart, rb->d_start, (size_t) count * sizeof(u32), stream), "fetch runs");
			check_gdsp (gdsp_memcpy_d2h (runEnd,   rb->d_end,   (size_t) count * sizeof(u32), stream), "fetch runs");
			check_gdsp (gdsp_memcpy_d2h (runVal,   rb->d_val,   (size_t) count * sizeof(valtype), stream), "fetch runs");
			check_gdsp (gdsp_stream_sync (stream), "synchronise");
			}

		format_runs (f, s, runStart, runEnd, runVal, count, !noValues, precision, uncovered, o);
		}
	if (trackOperations) tracking_report ("output(--done--)\n");
	}

/* read_all_chromosomes / write_all_chromosomes, genodsp.c:1718-1792: the signal as text, ten decimals, runs
 * collapsed, zero stretches left out; read back over a cleared genome.  (percentile --preserve: values come
 * back rounded to ten decimals, exactly as they do in the reference.) */
void write_all_chromosomes (char* filename)
	{
	FILE* f = fopen (filename, "wt");
	if (f == NULL) { fprintf (stderr, "can't open \"%s\" for writing\n", filename);  exit (EXIT_FAILURE); }
	if (trackOperations) fprintf (stderr, "write_all(%s)\n", filename);
	int saveTrack = trackOperations;
	trackOperations = false;
	report_intervals (f, /*precision*/ 10, /*no values*/ false, /*collapse*/ true, uncovered_hide, /*origin one*/ false);
	trackOperations = saveTrack;
	fclose (f);
	}

void read_all_chromosomes (char* filename)
	{
	FILE* f = fopen (filename, "rt");
	if (f == NULL) { fprintf (stderr, "can't open \"%s\" for reading\n", filename);  exit (EXIT_FAILURE); }
	if (trackOperations) fprintf (stderr, "read_all(%s)\n", filename);
	int saveTrack = trackOperations;
	trackOperations = false;
	read_intervals (f, /*value column*/ 4-1, /*origin one*/ false, ri_overlapSum, /*clear*/ true, 0.0);
	trackOperations = saveTrack;
	fclose (f);
	}

/* ------------------------------------------------ --report=gpu: what each step cost ---- */
/* per-operator HIP-event time and the rates that follow from it, from the product binary itself.
 * A per-chromosome operator (or fused chain) is bracketed by two events on its chromosome's stream; its time is the
 * sum over its calls on the slowest device (devices work side by side).  Whole-genome steps -- ingest, file-driven
 * operators, percentile, output -- mix host and device work and are timed on the wall clock with the devices
 * drained.  GB/s credits SURVEY 8d's algorithmic bytes: 16 per base and operator, 8 for percentile and the report. */
typedef struct phase
	{
	char   label[160];
	dspop* op;  int nops;
	u32    calls;  u64 units;  const char* unitName;
	double bytesPerBase, wallMs, devMs[64];
	int    onWall;
	} phase;
typedef struct span { int phase, device;  void *ev0, *ev1; } span;
static phase  phases[512];
static int    numPhases = 0;
static span*  spans = NULL;
static u32    numSpans = 0, capSpans = 0;

static double now_ms (void)
	{ struct timespec t;  clock_gettime (CLOCK_MONOTONIC, &t);  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

static int phase_for (dspop* op, int nops, const char* fixedLabel)
	{
	for (int i=0 ; i<numPhases ; i++)
		{ if ((op != NULL) && (phases[i].op == op) && (phases[i].nops == nops)) return i; }
	if (numPhases == (int) (sizeof(phases)/sizeof(phases[0]))) return numPhases-1;
	phase* ph = &phases[numPhases];
	memset (ph, 0, sizeof(*ph));
	ph->op = op;  ph->nops = nops;  ph->unitName = "bases";
	if (fixedLabel != NULL) snprintf (ph->label, sizeof(ph->label), "%s", fixedLabel);
	else
		{
		size_t at = 0;
		dspop* o = op;
		for (int k=0 ; (k<nops) && (o!=NULL) && (at+2<sizeof(ph->label)) ; k++, o=o->next)
			at += (size_t) snprintf (ph->label + at, sizeof(ph->label) - at, "%s%s", (k == 0)? "" : "=", o->name);
		}
	ph->bytesPerBase = 16.0 * ((nops < 1)? 1 : nops);
	return numPhases++;
	}

static u32 span_open (void)                      /* on the current device's operator stream */
	{
	if (numSpans == capSpans)
		{
		capSpans = (capSpans == 0)? 1024 : 2*capSpans;
		spans = (span*) realloc (spans, capSpans * sizeof(span));
		if (spans == NULL) { fprintf (stderr, "out of memory\n");  exit (EXIT_FAILURE); }
		}
	span* sp = &spans[numSpans];
	sp->phase = -1;  sp->device = currentDevice;
	check_gdsp (gdsp_event_create (&sp->ev0), "create event");
	check_gdsp (gdsp_event_create (&sp->ev1), "create event");
	check_gdsp (gdsp_event_record (sp->ev0, op_stream ()), "record event");
	return numSpans++;
	}

static void span_close (u32 ix, dspop* op, int nops, u64 bases)
	{
	span* sp = &spans[ix];
	check_gdsp (gdsp_event_record (sp->ev1, op_stream ()), "record event");
	sp->phase = phase_for (op, nops, NULL);
	phases[sp->phase].calls++;
	phases[sp->phase].units += bases;
	}

static void wall_phase (dspop* op, const char* label, double ms, u64 units, const char* unitName, double bytesPerBase)
	{
	int i = phase_for (op, 1, label);
	phases[i].onWall = true;  phases[i].calls++;  phases[i].wallMs += ms;  phases[i].units += units;
	phases[i].unitName = unitName;  phases[i].bytesPerBase = bytesPerBase;
	}

static void report_gpu_times (void)
	{
	sync_all_devices ();
	for (u32 i=0 ; i<numSpans ; i++)
		{
		float ms = 0;
		check_gdsp (use_device (spans[i].device), "select device");
		check_gdsp (gdsp_event_elapsed_ms (spans[i].ev0, spans[i].ev1, &ms), "event time");
		if (spans[i].phase >= 0) phases[spans[i].phase].devMs[spans[i].device] += ms;
		gdsp_event_destroy (spans[i].ev0);  gdsp_event_destroy (spans[i].ev1);
		}
	check_gdsp (use_device (currentDevice), "select device");
	fprintf (stderr, "[%s] --report=gpu: %d device%s (%s); event = HIP-event ms summed over a step's calls on the slowest device,\n"
	                 "  wall = host clock with the devices drained (steps that mix host and device work)\n",
	         programName, numDevices, (numDevices == 1)? "" : "s", gdsp_version ());
	if (!batchLaunches)
		fprintf (stderr, "  launches: one per operator and chromosome%s; --batch gives one per operator and device\n",
		         trackOperations? " (--progress=operations keeps the reference's order)" : " (--nobatch)");
	fprintf (stderr, "  %-34s %6s %16s %-9s %10s %5s %10s %9s %6s\n", "step", "calls", "units", "", "ms", "clock", "Gbases/s", "GB/s", "B/base");
	for (int i=0 ; i<numPhases ; i++)
		{
		phase* ph = &phases[i];
		double ms = ph->wallMs;
		if (!ph->onWall) { for (int d=0 ; d<numDevices ; d++) { if (ph->devMs[d] > ms) ms = ph->devMs[d]; } }
		fprintf (stderr, "  %-34s %6u %16llu %-9s %10.3f %5s", ph->label, ph->calls, (unsigned long long) ph->units, ph->unitName,
		         ms, ph->onWall? "wall" : "event");
		if ((strcmp (ph->unitName, "bases") == 0) && (ms > 0))
			fprintf (stderr, " %10.2f %9.1f %6.0f", ph->units / ms * 1e-6, ph->units * ph->bytesPerBase / ms * 1e-6, ph->bytesPerBase);
		fprintf (stderr, "\n");
		}
	}

/* ------------------------------------------------------------- option parsing */
static int process_operator_options (int argc, char** argv)    /* genodsp.c:634-723 */
	{
	int   consumed = 0;
	char* arg = argv[0];
	char* dspName;

	if ((arg[0] == specialPipeChar) && (arg[1] == 0)) { argv++;  argc--;  consumed++;  dspName = argv[0]; }
	else dspName = skip_whitespace (arg+1);
	if (dbgPipe) fprintf (stderr, "  dspName=\"%s\"\n", dspName);

	dspinfo* info = find_operator (dspName);
	if (info == NULL)
		{
		for (int i=0 ; notInThisBuild[i]!=NULL ; i++)
			{
			if (strcmp (dspName, notInThisBuild[i]) == 0)
				{
				fprintf (stderr, "\"%s\" is a genodsp operation outside this build's GPU hot path (see DESIGN.md)\n", dspName);
				exit (EXIT_FAILURE);
				}
			}
		chastise ("\"%s\" is not a known operation\n", dspName);
		}
	argv++;  argc--;  consumed++;

	int dspArgC = 0;
	while ((dspArgC < argc) && (argv[dspArgC][0] != specialPipeChar)) { dspArgC++;  consumed++; }
	if (dbgPipe)                                   /* genodsp.c:691-697: every argument left, not only this operator's */
		{
		fprintf (stderr, "  args:");
		for (int i=0 ; i<argc ; i++) fprintf (stderr, " %s", argv[i]);
		fprintf (stderr, "\n");
		}

	chastiseUsage = info->funcUsage;  chastiseUsageName = info->name;
	dspop* op = (*info->funcParse) (info->name, dspArgC, argv);
	chastiseUsage = NULL;  chastiseUsageName = NULL;

	op->name      = copy_string (info->name);
	op->funcApply = info->funcApply;
	op->funcFree  = info->funcFree;
	op->next      = NULL;
	if (tailOp == NULL) pipeline = op;  else tailOp->next = op;
	tailOp = op;
	if (dbgPipe) fprintf (stderr, "  argsConsumed=%d\n", consumed);
	return consumed;
	}

static void help_for (char* name)
	{
	dspinfo* info = find_operator (name);
	if (info == NULL) { fprintf (stderr, "\"%s\" is not a known operation\n", name);  exit (EXIT_FAILURE); }
	fprintf (stderr, "=== %s ===\n", info->name);
	(*info->funcUsage) (info->name, stderr, "  ");
	exit (EXIT_SUCCESS);
	}

static void parse_options (int _argc, char** _argv)            /* genodsp.c:284-629 */
	{
	int    argc = _argc - 1;
	char** argv = _argv + 1;
	char*  chromsFilename = NULL;

	if (argc == 0) chastise (NULL);
	for ( ; argc > 0 ; argv++, argc--)
		{
		char* arg    = argv[0];
		char* argVal = strchr (arg, '=');
		if (argVal != NULL) argVal++;

		if (arg[0] == specialPipeChar)
			{
			if (dbgPipe) fprintf (stderr, "pipe char in arg[%d]\n", (int) (argv - _argv));
			if ((argc == 1) && (arg[1] == 0))
				chastise ("%c at end of command line, with no operation\n", specialPipeChar);
			int consumed = process_operator_options (argc, argv);
			argv += consumed - 1;  argc -= consumed - 1;
			continue;
			}
		if ((strcmp_prefix (arg, "--chromosomes=") == 0) || (strcmp_prefix (arg, "--chroms=") == 0))
			{ chromsFilename = argVal;  continue; }
		if ((strcmp (arg, "--novalue") == 0) || (strcmp (arg, "--novalues") == 0) || (strcmp (arg, "--value=none") == 0))
			{ valColumn = -1;  set_named_global ("valColumn", (valtype) valColumn);  continue; }
		if (strcmp_prefix (arg, "--value=") == 0)
			{
			valColumn = string_to_int (argVal) - 1;
			if (valColumn == -1) chastise ("value column can't be 0 (\"%s\")\n", arg);
			if (valColumn < 0)   chastise ("value column can't be negative (\"%s\")\n", arg);
			if (valColumn < 3)   chastise ("value column can't be 1, 2 or 3 (\"%s\")\n", arg);
			set_named_global ("valColumn", (valtype) valColumn);
			continue;
			}
		if ((strcmp (arg, "--nooutputvalue") == 0) || (strcmp (arg, "--nooutputvalues") == 0))
			{ noOutputValues = true;  set_named_global ("noOutputValues", (valtype) noOutputValues);  continue; }
		if (strcmp_prefix (arg, "--precision=") == 0)
			{
			valPrecision = string_to_int (argVal);
			if (valPrecision < 0) chastise ("precision can't be negative (\"%s\")\n", arg);
			set_named_global ("valPrecision", (valtype) valPrecision);
			continue;
			}
		if (strcmp (arg, "--nocollapse") == 0)
			{ collapseRuns = false;  set_named_global ("collapseRuns", (valtype) collapseRuns);  continue; }
		if ((strcmp (arg, "--uncovered:hide") == 0) || (strcmp (arg, "--hide:uncovered") == 0))
			{ showUncovered = uncovered_hide;  set_named_global ("showUncovered", (valtype) showUncovered);  continue; }
		if ((strcmp (arg, "--uncovered:show") == 0) || (strcmp (arg, "--show:uncovered") == 0))
			{ showUncovered = uncovered_show;  set_named_global ("showUncovered", (valtype) showUncovered);  continue; }
		if ((strcmp (arg, "--uncovered:NA") == 0) || (strcmp (arg, "--uncovered:mark") == 0)
		 || (strcmp (arg, "--mark:uncovered") == 0) || (strcmp (arg, "--markgaps") == 0))
			{ showUncovered = uncovered_NA;  set_named_global ("showUncovered", (valtype) showUncovered);  continue; }
		if ((strcmp (arg, "--cliptochromosome") == 0) || (strcmp (arg, "--cliptochrom") == 0)
		 || (strcmp (arg, "--cliptolength") == 0) || (strcmp (arg, "--clip") == 0))
			{ clipToLength = true;  continue; }
		if ((strcmp (arg, "--origin=one") == 0) || (strcmp (arg, "--origin=1") == 0))
			{ originOne = true;  set_named_global ("originOne", (valtype) originOne);  continue; }
		if ((strcmp (arg, "--origin=zero") == 0) || (strcmp (arg, "--origin=0") == 0))
			{ originOne = false;  set_named_global ("originOne", (valtype) originOne);  continue; }
		if (strcmp (arg, "--nooutput") == 0) { inhibitOutput = true;  continue; }
		if ((strcmp_prefix (arg, "--window=") == 0) || (strcmp_prefix (arg, "W=") == 0) || (strcmp_prefix (arg, "--W=") == 0))
			{
			int w = string_to_unitized_int (argVal, /*thousands*/ true);
			if (w == 0) chastise ("window size can't be zero (\"%s\")\n", arg);
			if (w < 0)  chastise ("window size can't be negative (\"%s\")\n", arg);
			set_named_global ("windowSize", (valtype) w);
			continue;
			}
		if (strcmp_prefix (arg, "--gpus=") == 0)
			{
			numDevices = string_to_int (argVal);
			if ((numDevices < 1) || (numDevices > 64)) chastise ("--gpus must be between 1 and 64 (\"%s\")\n", arg);
			continue;
			}
		if (strcmp (arg, "--nofuse") == 0) { fuseChains = false;  continue; }
		if (strcmp (arg, "--nobatch") == 0) { batchLaunches = false;  continue; }
		if (strcmp (arg, "--batch") == 0)   { batchLaunches = true;   continue; }
		if (strcmp (arg, "--shards=show") == 0)          { showShards = true;   continue; }
		if (strcmp (arg, "--sharding=bases") == 0)       { shardBases = true;   continue; }
		if (strcmp (arg, "--sharding=chromosomes") == 0) { shardBases = false;  continue; }
		if (strcmp (arg, "--reduce=rccl") == 0) { reduceHow = reduce_rccl;  continue; }
		if (strcmp (arg, "--reduce=host") == 0) { reduceHow = reduce_host;  continue; }
		if (strcmp (arg, "--smooth=exact") == 0) { firMode = GDSP_FIR_EXACT;  continue; }
		if (strcmp (arg, "--smooth=fma")   == 0) { firMode = GDSP_FIR_FMA;    continue; }
		if (strcmp (arg, "--smooth=hann")  == 0) { firMode = GDSP_FIR_HANN;   continue; }
		if (strcmp (arg, "--percentile=auto")    == 0) { selectStrategy = GDSP_SELECT_AUTO;     continue; }
		if (strcmp (arg, "--percentile=radix")   == 0) { selectStrategy = GDSP_SELECT_RADIX;    continue; }
		if (strcmp (arg, "--percentile=bracket") == 0) { selectStrategy = GDSP_SELECT_BRACKET;  continue; }
		if (strcmp (arg, "?") == 0) usage_operations ();
		if ((strcmp_prefix (arg, "--help=") == 0) || (strcmp_prefix (arg, "?=") == 0))
			{
			if (strcmp (argVal, "*") != 0) help_for (argVal);
			goto help_for_all;
			}
		if (strcmp_prefix (arg, "?") == 0) help_for (arg+1);
		if (strcmp (arg, "--help") == 0)
			{
		help_for_all:
			for (u32 i=0 ; i<dspTableLen ; i++)
				{
				if (dspTable[i].funcShort == NULL) continue;
				fprintf (stderr, "=== %s ===\n", dspTable[i].name);
				(*dspTable[i].funcUsage) (dspTable[i].name, stderr, "  ");
				}
			exit (EXIT_SUCCESS);
			}
		if ((strcmp (arg, "--report=comments") == 0) || (strcmp (arg, "--report:comments") == 0))
			{ reportComments = true;  continue; }
		if ((strcmp (arg, "--report=gpu") == 0) || (strcmp (arg, "--report:gpu") == 0))
			{ reportGpu = true;  continue; }
		if ((strcmp_prefix (arg, "--progress=input:") == 0) || (strcmp_prefix (arg, "--progress:input=") == 0)
		 || (strcmp_prefix (arg, "--progress:input:") == 0))
			{
			if (strcmp_prefix (argVal, "input:") == 0) argVal = strchr (arg, ':') + 1;
			reportInputProgress = string_to_unitized_int (argVal, /*thousands*/ true);
			continue;
			}
		if ((strcmp (arg, "--progress=operations") == 0) || (strcmp (arg, "--progress:operations") == 0)
		 || (strcmp (arg, "--debug=operations") == 0))
			{ trackOperations = true;  continue; }
		if (strcmp (arg, "--version") == 0)
			{ fprintf (stderr, "%s (version %s; %s)\n", programName, programVersion, gdsp_version ());  exit (EXIT_SUCCESS); }
		if (strcmp (arg, "--debug=input") == 0)   { dbgInput   = true;  continue; }
		if (strcmp (arg, "--debug=pipe") == 0)    { dbgPipe    = true;  continue; }
		if (strcmp (arg, "--debug=globals") == 0) { dbgGlobals = true;  continue; }
		if (strcmp_prefix (arg, "--") == 0) chastise ("Can't understand \"%s\"\n", arg);

		/* <chromosome>:<length> or <chromosome>:<start>:<end> */
		char* c1 = strchr (arg, ':');
		if (c1 == NULL)
			{
			fprintf (stderr, "\"%s\" contains no chromosome length\n(expected \"chromosome:length\" or \"chromosome:start:end\")\n", arg);
			exit (EXIT_FAILURE);
			}
		char* c2 = strchr (c1+1, ':');
		u32 chromStart = 0, chromLength;
		*(c1++) = 0;
		if (c2 == NULL) chromLength = (u32) string_to_u32 (c1);
		else { *(c2++) = 0;  chromStart = (u32) string_to_u32 (c1);  chromLength = (u32) string_to_u32 (c2) - chromStart; }
		if (!add_chromosome_spec (arg, chromStart, chromLength)) chastise ("can't specify %s more than once\n", arg);
		}
	if (chromsFilename != NULL) read_chromosome_lengths (chromsFilename);
	if (chromsOfInterest == NULL) chastise ("gotta give me some chromosome names\n");
	}

/* what --progress=operations calls a unit of work: the chromosome, or chromosome:start-end for a stretch */
static int runSharded = false;
static const char* unit_label (spec* s)
	{
	static char where[200];
	if (!runSharded) return s->chrom;
	piece* p = (piece*) s;                                     /* (a stretch's spec is the first member of its piece) */
	snprintf (where, sizeof(where), "%.100s:%u-%u", s->chrom, p->ownStart, p->ownEnd);
	return where;
	}

void apply_to_unit (dspop* op, spec* s)
	{
	select_device_of (s);
	activeSpec = runSharded? s : NULL;
	(*op->funcApply) (op, s->chrom, s->length, s->valVector);
	activeSpec = NULL;
	}

/* ----------------------------------------------------------------------- main */
int main (int argc, char** argv)
	{
	set_named_global ("valColumn",     (valtype) valColumn);       /* genodsp.c:835-840 */
	set_named_global ("valPrecision",  (valtype) valPrecision);
	set_named_global ("collapseRuns",  (valtype) collapseRuns);
	set_named_global ("showUncovered", (valtype) showUncovered);
	set_named_global ("originOne",     (valtype) originOne);
	parse_options (argc, argv);
	/* --progress=operations prints a line per operator and chromosome in the order they are applied; the reference applies
	 * them chromosome by chromosome (genodsp.c:909-921), so that order is kept there unless --batch asks otherwise */
	if (batchLaunches < 0) batchLaunches = !trackOperations;

	if (showShards)
		{
		sort_chromosomes_by_length ();
		deal_chromosomes ();
		if (shardBases) plan_pieces ();
		show_shards ();
		return EXIT_SUCCESS;
		}

	sort_chromosomes_by_length ();
	deal_chromosomes ();
	if (shardBases) plan_pieces ();                            /* (before allocation: stretches count towards scratch sizes) */
	const double tStart = now_ms ();
	allocate_vectors ();                                       /* HIP start-up, one arena per device: beside the parse of stdin */

	/* stdin is the signal unless the first operator is `input` (genodsp.c:891-893) */
	if ((pipeline == NULL) || (strcmp (pipeline->name, "input") != 0))
		{
		double t0 = now_ms ();
		read_intervals (stdin, valColumn, originOne, ri_overlapSum, /*clear*/ false, 0.0);
		wait_for_vectors ();
		if (reportGpu) { sync_all_devices ();  wall_phase (NULL, "input (stdin: parse, stage, apply)", now_ms () - t0, intervalsRead, "intervals", 0); }
		}
	wait_for_vectors ();
	if (reportGpu)                                             /* what a run spends before its first operator; the helper thread's clock */
		{
		u64 bases = 0;
		for (int i=0 ; chromsSorted[i]!=NULL ; i++) bases += chromsSorted[i]->length;
		wall_phase (NULL, "start-up (HIP runtime, devices)", allocStartupMs, (u64) numDevices, "devices", 0);
		wall_phase (NULL, allocThreaded? "allocate vectors (one arena per device; beside the input)" : "allocate vectors (one arena per device)",
		            allocVectorsMs, bases, "values", 0);
		wall_phase (NULL, "ready for the first operator (since start)", now_ms () - tStart, bases, "values", 0);
		}

	/* batching loop, genodsp.c:900-936: maximal runs of per-chromosome operators go
	 * chromosome by chromosome (longest first); whole-genome operators get one call */
	u32 maxLength = 0;
	for (int i=0 ; chromsSorted[i]!=NULL ; i++) { if (chromsSorted[i]->length > maxLength) maxLength = chromsSorted[i]->length; }
	dspop* firstOp = pipeline;
	while (firstOp != NULL)
		{
		dspop* stopOp;
		for (stopOp=firstOp ; stopOp!=NULL ; stopOp=stopOp->next) { if (stopOp->atRandom) break; }
		if (stopOp != firstOp)
			{
			u32 reachL = 0, reachR = 0;
			int pointwise = false;                    /* a run of per-base operators neither needs fresh halos nor spoils them */
			int sharded = shardBases && run_reach (firstOp, stopOp, &reachL, &reachR, &pointwise);
			int nunits = 0;
			if (sharded) { to_pieces ();  if (!pointwise) refresh_halos ();  nunits = numPieces; }
			else         { to_whole ();  while (chromsSorted[nunits] != NULL) nunits++; }
			spec** units = (spec**) malloc ((nunits + 1) * sizeof(spec*));
			if (units == NULL) { fprintf (stderr, "out of memory\n");  exit (EXIT_FAILURE); }
			for (int i=0 ; i<nunits ; i++) units[i] = sharded? &pieces[i].x.pub : chromsSorted[i];
			runSharded = sharded;
			/* the run as stretches of operators: those with a one-launch-per-device form go operator by operator, each
			 * covering all the units of a device at once; the others chromosome by chromosome as in genodsp.c:909-921
			 * (chromosomes are independent for every operator of a run, so the order cannot matter) */
			for (dspop* op=firstOp ; op!=stopOp ; )
				{
				if (batchLaunches && op_batchable (op))
					{
					int consumed = 1;
					for (int d=0 ; d<numDevices ; d++)
						{
						int     m = 0;
						u64     bases = 0;
						spec**  mine = (spec**) malloc ((nunits + 1) * sizeof(spec*));
						if (mine == NULL) { fprintf (stderr, "out of memory\n");  exit (EXIT_FAILURE); }
						for (int i=0 ; i<nunits ; i++)
							{ if (((xspec*) units[i])->device == d) { mine[m++] = units[i];  bases += units[i]->length; } }
						if (m > 0)
							{
							select_device_of (mine[0]);
							u32 sp = reportGpu? span_open () : 0;
							consumed = batch_apply_on_device (op, stopOp, mine, m, fuseChains);
							if (reportGpu) span_close (sp, op, consumed, bases);
							}
						free (mine);
						}
					for (int k=0 ; k<consumed ; k++, op=op->next)
						{
						if (!trackOperations) continue;
						for (int i=0 ; i<nunits ; i++) fprintf (stderr, "%s(%s)\n", op->name, unit_label (units[i]));
						}
					continue;
					}
				dspop* runEnd = op;
				while ((runEnd != stopOp) && !(batchLaunches && op_batchable (runEnd))) runEnd = runEnd->next;
				for (int i=0 ; i<nunits ; i++)
					{
					spec* s = units[i];
					const char* where = unit_label (s);
					select_device_of (s);
					activeSpec = sharded? s : NULL;
					for (dspop* o=op ; o!=runEnd ; o=o->next)
						{
						if (trackOperations) fprintf (stderr, "%s(%s)\n", o->name, where);
						dspop* first = o;
						u32 sp = reportGpu? span_open () : 0;
						int fused = fuseChains? try_fused_apply (o, runEnd, s) : 0;
						if (fused == 0) (*o->funcApply) (o, s->chrom, s->length, s->valVector);
						if (reportGpu) span_close (sp, first, (fused == 0)? 1 : fused, s->length);
						for ( ; fused > 1 ; fused--)           /* the chain ran as one kernel */
							{
							o = o->next;
							if (trackOperations) fprintf (stderr, "%s(%s)\n", o->name, where);
							}
						}
					activeSpec = NULL;
					}
				op = runEnd;
				}
			free (units);
			if (sharded && !pointwise) halosFresh = false;     /* (per-base operators transform halo and owner alike) */
			}
		if (stopOp == NULL) firstOp = NULL;
		else
			{
			if (trackOperations) tracking_report ("%s(*)\n", stopOp->name);
			if ((stopOp->funcApply != op_percentile_apply) && (stopOp->funcApply != op_invert_apply)
			 && (stopOp->funcApply != op_show_variables_apply))
				to_whole ();                                   /* file-driven operators address whole chromosomes */
			double t0 = now_ms ();
			u64 ivBefore = intervalsRead;
			if (reportGpu) sync_all_devices ();
			int ran = 1;
			if (fuseChains && (stopOp->funcApply == op_percentile_apply))
				{
				/* `percentile P = binarize --threshold=percentileP`: the binarize in the percentile's own read of the signal */
				ran = percentile_with_binarize (stopOp, stopOp->next);
				if ((ran == 2) && trackOperations)
					{ for (int i=0 ; chromsSorted[i]!=NULL ; i++) fprintf (stderr, "%s(%s)\n", stopOp->next->name, chromsSorted[i]->chrom); }
				}
			else (*stopOp->funcApply) (stopOp, "*", maxLength, NULL);
			if (reportGpu)
				{
				u64 total = 0;
				for (int i=0 ; chromsSorted[i]!=NULL ; i++) total += chromsSorted[i]->length;
				sync_all_devices ();
				if (stopOp->funcApply == op_show_variables_apply) ;
				else if ((intervalsRead != ivBefore) && (stopOp->funcApply != op_percentile_apply))
					{
					char label[160];
					snprintf (label, sizeof(label), "%s (file: parse, stage, apply)", stopOp->name);
					wall_phase (stopOp, label, now_ms () - t0, intervalsRead - ivBefore, "intervals", 0);
					}
				else if (ran == 2) wall_phase (stopOp, "percentile=binarize", now_ms () - t0, total, "bases", 24);
				else wall_phase (stopOp, stopOp->name, now_ms () - t0, total, "bases", (stopOp->funcApply == op_percentile_apply)? 8 : 16);
				}
			firstOp = (ran == 2)? stopOp->next->next : stopOp->next;
			}
		}

	if (!inhibitOutput)
		{
		double t0 = now_ms ();
		u64 linesBefore = linesWritten;
		if (reportGpu) sync_all_devices ();
		report_intervals (stdout, valPrecision, noOutputValues, collapseRuns, showUncovered, originOne);
		if (reportGpu) { fflush (stdout);  wall_phase (NULL, "output (find runs, fetch, format)", now_ms () - t0, linesWritten - linesBefore, "lines", 0); }
		}
	sync_all_devices ();
	if (allocThreaded) { pthread_join (allocThread, NULL);  allocThreaded = false; }
	if (reportGpu)
		{
		u64 bases = 0;
		for (int i=0 ; chromsSorted[i]!=NULL ; i++) bases += chromsSorted[i]->length;
		if (partnersPlanned) wall_phase (NULL, "allocate partners (one arena per device; helper thread)", allocPartnersMs, bases, "values", 0);
		else               wall_phase (NULL, "allocate partners (none: every operator works in place)", 0, 0, "values", 0);
		wall_phase (NULL, "waited for the helper thread's allocations", allocWaitedMs, 0, "values", 0);
		report_gpu_times ();
		}

	for (dspop* op=pipeline, *next ; op!=NULL ; op=next)
		{ next = op->next;  free (op->name);  (*op->funcFree) (op); }
	if (deviceComm != NULL) { gdsp_percentiles_use_comm (NULL);  gdsp_comm_destroy (deviceComm); }
	return EXIT_SUCCESS;
	}

/* every exit() of the host sources arrives here (-Dexit=gdsp_exit, host/Makefile): a complaint about the input may end
 * the run while the allocation helper thread is still inside the HIP runtime, and exit() runs the runtime's handlers
 * under it.  Let it finish first (the helper itself -- no GPU visible -- has nobody to wait for). */
#undef exit
extern void exit (int status) __attribute__((noreturn));
void gdsp_exit (int status)
	{
	join_allocation_at_exit ();
	exit (status);
	}
