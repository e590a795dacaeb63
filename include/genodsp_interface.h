/* genodsp_interface.h -- operator ("dspop") plugin surface of the MI355X host driver.
 *
 * This is an independent, ABI-compatible re-declaration of the reference's
 * operator interface (rsharris/genodsp genodsp_interface.h:20-190): the same type
 * names, struct layouts, macro names and host-service prototypes, so that an
 * operator group written against the reference's header (five functions
 * X_short / X_usage / X_parse / X_free / X_apply registered with
 * dspinforecord(name, X)) compiles against this one unchanged.
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

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t  s32;                   /* utilities.h:4-8 in the reference */
typedef uint32_t u32;
typedef int64_t  s64;
typedef uint64_t u64;

#ifdef __GNUC__
#define arg_dont_complain(arg) arg __attribute__ ((unused))
#else
#define arg_dont_complain(arg) arg
#endif

#ifndef true
#define true  1
#define false 0
#endif

/* ---- values along the chromosomes (reference :20-26) ---- */
typedef double valtype;
#define string_to_valtype(s)       ((valtype) string_to_double(s))
#define try_string_to_valtype(s,v) try_string_to_double(s,(valtype*)v)
#define valtypeFmt     "%f"
#define valtypeFmtPrec "%.*f"
#define valtypeMax     DBL_MAX
#define valtypePuny    DBL_MIN

/* ---- chromosomes of interest (reference :37-57) ---- */
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

/* ---- operator function groups (reference :76-125) ---- */
struct dspop;
#define opfuncargs_short (char*,int,FILE*,char*)
#define opfuncargs_usage (char*,FILE*,char*)
#define opfuncargs_parse (char*,int,char**)
#define opfuncargs_free  (struct dspop*)
#define opfuncargs_apply (struct dspop*,char*,u32,valtype*)

typedef void          (*opfunc_short) opfuncargs_short;
typedef void          (*opfunc_usage) opfuncargs_usage;
typedef struct dspop* (*opfunc_parse) opfuncargs_parse;
typedef void          (*opfunc_free)  opfuncargs_free;
typedef void          (*opfunc_apply) opfuncargs_apply;

#define dspprototypes(funcName) \
void          funcName##_short opfuncargs_short; \
void          funcName##_usage opfuncargs_usage; \
struct dspop* funcName##_parse opfuncargs_parse; \
void          funcName##_free  opfuncargs_free;  \
void          funcName##_apply opfuncargs_apply;

/* every operator's private record starts with this */
typedef struct dspop
	{
	struct dspop* next;
	char*         name;
	opfunc_apply  funcApply;
	opfunc_free   funcFree;
	int           atRandom;   /* true: one call with vName="*", v=NULL; the operator
	                             walks chromsSorted itself                            */
	} dspop;

typedef struct dspinfo
	{
	char*        name;
	opfunc_short funcShort;
	opfunc_usage funcUsage;
	opfunc_parse funcParse;
	opfunc_free  funcFree;
	opfunc_apply funcApply;
	} dspinfo;

#define dspinforecord(name,funcName) \
	{ name, funcName##_short, funcName##_usage, funcName##_parse, funcName##_free, funcName##_apply }
#define dspinfoalias(name) \
	{ name, NULL, NULL, NULL, NULL, NULL }

/* ---- miscellany (reference :133-159) ---- */
#ifndef M_PI
#define M_PI 3.14159265358979323846264
#endif

extern int trackOperations;
extern int reportComments;
extern u32 reportInputProgress;

#define uncovered_NA   -1
#define uncovered_show 1
#define uncovered_hide 0

#define ri_overlapSum 0
#define ri_overlapMin 1
#define ri_overlapMax 2

/* ---- host services (reference :167-190), same names and argument meaning ---- */
void     chastise               (const char* format, ...);
spec*    find_chromosome_spec   (char* chrom);
void     read_intervals         (FILE* f, int valCol, int originOne,
                                 int overlapOp, int clear, valtype missingVal);
int      read_interval          (FILE* f, char* buffer, int bufferLen, int valCol,
                                 char** chrom, u32* start, u32* end, valtype* val);
void     report_intervals       (FILE* f, int precision, int noOutputValues, int collapseRuns,
                                 int showUncovered, int originOne);
valtype* get_scratch_vector     (void);          /* DEVICE scratch, longest-chromosome sized */
void     release_scratch_vector (valtype* v);
void     set_named_global       (char* name, valtype val);
valtype  get_named_global       (char* name, valtype defaultVal);
int      named_global_exists    (char* name, valtype* val);
void     report_named_globals   (FILE* f, char* indent);
void     tracking_report        (const char* format, ...);

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
