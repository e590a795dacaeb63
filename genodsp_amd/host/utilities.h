/* utilities.h -- small string / number helpers of the host driver.
 * Same names and behaviour as the reference's helpers (utilities.h:23-36): operators written
 * against the reference call these.  Implemented independently in utilities.c. */
#ifndef utilities_H
#define utilities_H

#include <stddef.h>
#include "genodsp_interface.h"

#define u32Max ((u32) -1)

/* strings */
char* copy_string (const char* s);
void safe_strncpy (char* dest, const char* src, size_t n);
int strcmp_prefix (const char* str, const char* prefix);      /* 0 when str starts with prefix */
int strcmp_suffix (const char* str, const char* suffix);      /* 0 when str ends with suffix   */
char* skip_whitespace (char* s);
char* skip_darkspace (char* s);

/* text -> number; the string_to_* forms stop the program on anything that is not a number */
int string_to_int (const char* s);
int string_to_u32 (const char* s);
int try_string_to_u32 (const char* s, u32* v);
int string_to_unitized_int (const char* s, int byThousands);  /* 10K, 1.5M, 2G                 */
double string_to_double (const char* s);                      /* also inf, -inf, 1/inf         */
int try_string_to_double (const char* s, double* v);

/* number -> text */
char* ucommatize (const u64 v);                               /* 1234567 -> "1,234,567"        */

#endif
