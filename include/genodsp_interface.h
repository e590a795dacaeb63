/* genodsp_interface.h -- operator ("dspop") plugin surface of the MI355X host driver.
 *
 * This is an independent, ABI-compatible re-declaration of the reference's
 * operator interface (rsharris/genodsp genodsp_interface.h:20-190): the same type
 * names, struct layouts, macro names and ALL sixteen host-service prototypes
 * (:167-190), so that an operator group written against the reference's header
 * (five functions X_short / X_usage / X_parse / X_free / X_apply registered with
 * dspinforecord(name, X)) compiles against this one unchanged.  Held by
 * tests/test_plugin_boundary.py: every operator file of the reference passes
 * `gcc -fsyntax-only -Wall -Wextra -Werror` against include/ (where /root/reference
 * exists), and a group written in the reference's style is linked into the driver's
 * table (GDSP_EXTRA_OPERATORS) and run on the GPU.
 *
 * The one semantic difference is where the values live: `valVector` and the `v`
 * handed to X_apply are DEVICE pointers (f64 arrays in HBM).  An operator body
 * therefore never dereferences v; it calls a kernel entry point of
 * include/genodsp_hip.h on the driver's stream.  The extra services at the end of
 * this file (stream, partner vector, device selection) are what such a shim needs
 * and have no counterpart in the reference.
 */
#ifndef genodsp_interface_H
#define genodsp_interface_H

#include <stdio.h>
#include <inttypes.h>
#include "utilities.h"                  /* s32/u32/s64/u64, arg_dont_complain and the string helpers; the reference's
                                           operator files include it themselves, before this header */

#ifdef __cplusplus
extern "C" {
#endif

#ifndef true
#define true  1
#define false 0
#endif

/* ---- host services an operator may call: same names and argument meaning as the reference's
 *      (:167-190).  Declared first because they only need the scalar types. ---- */
typedef double valtype;                            /* what a chromosome holds per base (:20) */
struct spec;
struct dspop;

void chastise (const char* format, ...);           /* message, the operator's usage text, exit */
void tracking_report (const char* format, ...);    /* --progress=operations                    */
struct spec* find_chromosome_spec (char* chrom);

valtype* get_scratch_vector (void);                /* DEVICE scratch: longest-chromosome many valtype, on the current GPU */
void release_scratch_vector (valtype* v);
s32*  get_scratch_ints (void);                     /* DEVICE scratch: longest-chromosome many s32 (:182, genodsp.c:1943-1979) */
void  release_scratch_ints (s32* v);

/* the whole signal to / from a text file (genodsp.c:1718-1792): what percentile --preserve uses.
 * Written with 10 decimals, runs collapsed, zero stretches left out; read back over a cleared genome. */
void read_all_chromosomes  (char* filename);
void write_all_chromosomes (char* filename);

int  valtype_ascending (const void* v1, const void* v2);      /* qsort comparator for HOST arrays of valtype (:190) */

int named_global_exists (char* name, valtype* val);           /* the percentile -> threshold channel */
valtype get_named_global (char* name, valtype defaultVal);
void set_named_global (char* name, valtype val);
void report_named_globals (FILE* f, char* indent);

int read_interval (FILE* f, char* buffer, int bufferLen, int valCol, char** chrom, uint32_t* start, uint32_t* end, valtype* val);
void read_intervals (FILE* f, int valCol, int originOne, int overlapOp, int clear, valtype missingVal);
void report_intervals (FILE* f, int precision, int noOutputValues, int collapseRuns, int showUncovered, int originOne);

/* what read_intervals does where intervals overlap, and how report_intervals writes zero stretches */
enum { ri_overlapSum = 0, ri_overlapMin = 1, ri_overlapMax = 2 };
enum { uncovered_hide = 0, uncovered_show = 1, uncovered_NA = -1 };

extern int trackOperations, reportComments;
extern u32 reportInputProgress;

/* ---- values (:20-26): text <-> valtype goes through utilities.h's double conversions ---- */
#define valtypeMax  DBL_MAX
#define valtypePuny DBL_MIN
#define valtypeFmt "%f"
#define valtypeFmtPrec "%.*f"
#define string_to_valtype(text) ((valtype) string_to_double (text))
#define try_string_to_valtype(text,out) try_string_to_double (text, (valtype*) (out))
#ifndef M_PI
#define M_PI 3.14159265358979323846264
#endif

/* ---- chromosomes of interest (:37-57).  Field order and types are the reference's. ---- */
typedef struct spec
	{
	struct spec* next;        /* list in chromosome-file order (= output order)     */
	char*        chrom;
	int          flag;        /* scratch for operators (e.g. "seen in this file")   */
	u32          start;       /* bases before the vector, for name:start:end specs  */
	u32          length;      /* entries in valVector; never zero                   */
	valtype*     valVector;   /* DEVICE pointer: `length` f64 values in HBM         */
	} spec;

extern spec*  chromsOfInterest;   /* file order                                       */
extern spec** chromsSorted;       /* longest first, NULL terminated (processing order) */

/* ---- operator function groups (:76-125): five functions per operator X --
 *      X_short, X_usage, X_parse, X_free, X_apply -- declared by dspprototypes(X) and
 *      entered into the operator table by dspinforecord("name", X) ---- */
typedef void (*opfunc_short) (char* name, int nameWidth, FILE* f, char* indent);
typedef void (*opfunc_usage) (char* name, FILE* f, char* indent);
typedef struct dspop* (*opfunc_parse) (char* name, int argc, char** argv);
typedef void (*opfunc_free) (struct dspop* op);
typedef void (*opfunc_apply) (struct dspop* op, char* vName, u32 vLen, valtype* v);

#define dspprototypes(X)                                                       \
	void X##_short (char*, int, FILE*, char*);  void X##_usage (char*, FILE*, char*);  \
	struct dspop* X##_parse (char*, int, char**);  void X##_free (struct dspop*);       \
	void X##_apply (struct dspop*, char*, u32, valtype*);

typedef struct dspop              /* every operator's private record starts with this */
	{
	struct dspop* next;
	char*         name;
	opfunc_apply  funcApply;
	opfunc_free   funcFree;
	int           atRandom;   /* true: one call with vName="*", v=NULL; the operator
	                             walks chromsSorted itself                            */
	} dspop;

typedef struct dspinfo            /* a row of the operator table; an alias row has only its name */
	{
	char*        name;
	opfunc_short funcShort;
	opfunc_usage funcUsage;
	opfunc_parse funcParse;
	opfunc_free  funcFree;
	opfunc_apply funcApply;
	} dspinfo;

#define dspinforecord(name,X) { name, X##_short, X##_usage, X##_parse, X##_free, X##_apply }
#define dspinfoalias(name)    { name, 0, 0, 0, 0, 0 }

/* ---- device-side additions (no counterpart in the reference) ---- */
void*    op_stream              (void);          /* hipStream_t of the chromosome being processed  */
valtype* partner_vector         (char* vName);   /* second HBM buffer of that chromosome, same length */
void     flip_vector            (char* vName);   /* the partner becomes valVector (no copy-back pass) */
void     select_device_of       (spec* chromSpec); /* make its GPU current (multi-GPU sharding)    */
void     interval_ops_from_file (char* opName, char* filename, int valCol, int originOne,
                                 int kind, valtype infinityVal); /* add/subtract/multiply/divide */
void     check_gdsp             (int status, const char* what);  /* exit with a message unless 0  */

#ifdef __cplusplus
}
#endif
#endif /* genodsp_interface_H */
