/* utilities.h -- scalar types and the small string / number helpers operators use.
 *
 * Same names, argument meaning and behaviour as the reference's helpers
 * (rsharris/genodsp utilities.h:4-40), so an operator file written for the reference --
 * which includes "utilities.h" and then "genodsp_interface.h" -- finds here what it
 * found there.  Implemented independently in genodsp_amd/host/utilities.c. */
#ifndef utilities_H
#define utilities_H

#include <stddef.h>
#include <inttypes.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t  s32;                   /* utilities.h:4-8 in the reference */
typedef uint32_t u32;
typedef int64_t  s64;
typedef uint64_t u64;

#define u32Max ((u32) -1)

#ifdef __GNUC__
#define arg_dont_complain(arg) arg __attribute__ ((unused))
#else
#define arg_dont_complain(arg) arg
#endif

/* strings */
char* copy_string (const char* s);
void safe_strncpy (char* dest, const char* src, size_t n);
int strcmp_prefix (const char* str, const char* prefix);      /* 0 when str starts with prefix */
int strcmp_suffix (const char* str, const char* suffix);      /* 0 when str ends with suffix   */
int strncmp_suffix (const char* str, const char* suffix, size_t n);  /* ... looking at no more than n characters of str */
char* skip_whitespace (char* s);
char* skip_darkspace (char* s);

/* text -> number; the string_to_* forms stop the program on anything that is not a number */
int string_to_int (const char* s);
int string_to_u32 (const char* s);
int string_to_unitized_int (const char* s, int byThousands);  /* 10K, 1.5M, 2G                 */
double string_to_double (const char* s);                      /* also inf, -inf, 1/inf         */
int try_string_to_double (const char* s, double* v);
int try_string_to_u32 (const char* s, u32* v);                /* (not in the reference)        */

/* number -> text; both return a buffer private to the function */
char* ucommatize (const u64 v);                               /* 1234567 -> "1,234,567"        */
char* duration_to_string (float seconds);                     /* 12.345s, 3m07.250s, 1h02m03.000s */

#define round_up_16(b)  ((((u64) (b))+15)&(~15))

#ifdef __cplusplus
}
#endif
#endif /* utilities_H */
