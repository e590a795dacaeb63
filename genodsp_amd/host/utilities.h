/* utilities.h -- small string / number helpers of the host driver.
 * Same names and behaviour as the reference's utilities.h:23-36 (operators written
 * against the reference call these), implemented independently in utilities.c. */
#ifndef utilities_H
#define utilities_H

#include <stddef.h>
#include "genodsp_interface.h"

#define u32Max ((u32) -1)

char*  copy_string            (const char* s);
int    strcmp_prefix          (const char* str, const char* prefix);  /* 0 when str starts with prefix */
int    strcmp_suffix          (const char* str, const char* suffix);  /* 0 when str ends with suffix   */
int    string_to_int          (const char* s);
int    string_to_u32          (const char* s);
int    string_to_unitized_int (const char* s, int byThousands);       /* 10K, 1.5M, 2G                 */
double string_to_double       (const char* s);                        /* also inf, -inf, 1/inf         */
int    try_string_to_u32      (const char* s, u32* v);
int    try_string_to_double   (const char* s, double* v);
char*  skip_whitespace        (char* s);
char*  skip_darkspace         (char* s);
char*  ucommatize             (const u64 v);                          /* 1234567 -> "1,234,567"        */
void   safe_strncpy           (char* dest, const char* src, size_t n);

#endif
