/* host_services.h -- driver internals shared with the operator files (not part of the
 * plugin surface in include/genodsp_interface.h). */
#ifndef host_services_H
#define host_services_H

#include "genodsp_interface.h"
#include "genodsp_hip.h"

extern int selectStrategy;                  /* GDSP_SELECT_* (--percentile=) */
extern int firMode;                         /* GDSP_FIR_EXACT or GDSP_FIR_FMA (--smooth=) */
extern int dbgInput;                        /* --debug=input: echo every input line (genodsp.c:1422) */

/* pending-interval batches: collect in file order, apply on the owning device */
void ib_begin       (void);
void ib_add         (spec* s, u32 start, u32 end, valtype val);
u64  ib_pending     (void);
void ib_flush_apply (int overlapOp, int clearFlags, valtype missingVal, int everyChromosome);
void ib_flush_scale (int divide, valtype infinityVal);
void ib_flush_mask  (int inside, valtype outsideVal, int binarizeFirst);
void ib_flush_over  (int wantMax, valtype fillVal);

void sync_all_devices    (void);
/* the signal as the stretches someone answers for, wherever it lives now: whole chromosomes, or (--sharding=bases,
 * between ingest and report) stretches of them.  v[0] is base `first` of chromosome s->chrom; base/baseLen is the
 * 16-byte aligned vector v lies in (an elementwise operator may as well run over all of it) */
typedef struct sigpart { spec* s;  valtype* v;  u32 n;  u32 first;  valtype* base;  u32 baseLen; } sigpart;
int  signal_parts (sigpart** parts);
int  signal_in_whole_chromosomes (void);       /* false under --sharding=bases (the signal may live in stretches) */
void to_whole (void);                          /* make the whole chromosomes current (file-driven operators, report) */
/* how whole-genome operators (percentile, invert) combine what the devices of this process found: the
 * reduction hook for gdsp_percentiles (NULL: the library adds its devices' counts on the host) */
gdsp_reduce_fn reduce_over_devices (void** ctx);
void genome_extremes (valtype* lo, valtype* hi);     /* min / max of the whole genome over all devices (invert) */
u64   ib_batch_limit (void);                   /* intervals buffered before they are applied (8 M; GDSP_BATCH_INTERVALS) */
void* device_workspace (size_t bytes);         /* per-device, grows on demand, kept for the run */
void* long_window_workspace (size_t* bytes);   /* lazily allocated, for windows beyond one LDS tile */
int  device_count_in_use (void);
int  device_index_of     (spec* s);
int  physical_device_of  (spec* s);

/* ops_common.c: helpers shared by the operator files */
void* new_op           (char* name, size_t bytes, int atRandom);   /* zeroed control record */
u32   window_arg       (char* name, char* arg, char* argVal, const char* what);
/* "--x=<value|variable>": number now, or a named variable resolved at first apply */
void  value_or_variable (char* argVal, valtype* val, char** varName);
void  resolve_variable  (dspop* op, char** varName, valtype* val, const char* role);

/* ops_fused.c: run op (and the operators after it, up to stopOp) as one fused kernel when
 * the chain is one the device library fuses; returns how many operators were consumed (0 = none) */
int   try_fused_apply   (dspop* op, dspop* stopOp, spec* s);
/* one launch per operator per device (see ops_fused.c) */
int   op_batchable          (dspop* op);
int   batch_apply_on_device (dspop* op, dspop* stopOp, spec** units, int nunits, int allowFusion);
valtype* partner_of    (spec* s);              /* the second HBM buffer of a chromosome or stretch */
void     flip_spec     (spec* s);              /* swap vector and partner */
void     apply_to_unit (dspop* op, spec* s);   /* the operator's own apply on one chromosome or stretch */
void  op_limits_describe    (dspop* op, int* haveMin, valtype* lo, int* haveMax, valtype* hi, int* keepInside, valtype* zero);
valtype op_add_constant_value (dspop* op);
u32   op_smooth_window    (dspop* op);
u32   op_best_window      (dspop* op);
void  op_morph_reach      (dspop* op, u32* left, u32* right);
/* --sharding=bases: true when output i of this per-chromosome operator depends on inputs [i-left, i+right] only
 * and not on where i lies in the vector (so a stretch of a chromosome with that much halo computes what the whole
 * chromosome would); false for running sums, window grids and clump, which need the chromosome in one piece */
int   op_reach            (dspop* op, u32* left, u32* right);
void  op_local_describe   (dspop* op, u32* neighborhood, int* wantMax, valtype* fill);
void  op_morph_describe   (dspop* op, u32* left, u32* right, valtype* T, valtype* one, valtype* zero);
void  op_binarize_describe (dspop* op, valtype* T, int* tiesAbove, valtype* one, valtype* zero);
const char* op_binarize_pending (dspop* op, int* tiesAbove, valtype* one, valtype* zero);
/* `= percentile P = binarize --threshold=percentileP` in one read of the signal: runs the percentile operator and, when the
 * operator after it is that binarize, the binarize with it; returns how many operators ran (1 or 2) */
int   percentile_with_binarize (dspop* percentile, dspop* next);

/* argument helpers used by every operator's parse function */
#define OP_SHORT(fn, text)                                                            \
void fn##_short (char* name, int nameWidth, FILE* f, char* indent)                    \
	{                                                                                 \
	int fillW = nameWidth-2 - (int) strlen (name);                                    \
	if (indent == NULL) indent = "";                                                  \
	if (fillW > 0) fprintf (f, "%s%s:%*s", indent, name, fillW+1, " ");               \
	else           fprintf (f, "%s%s: ", indent, name);                               \
	fprintf (f, text "\n");                                                           \
	}

#define is_opt3(arg, long_, short_)                                                   \
	((strcmp_prefix (arg, "--" long_ "=") == 0) || (strcmp_prefix (arg, short_ "=") == 0) \
	 || (strcmp_prefix (arg, "--" short_ "=") == 0))

#endif
