/* utilities.c -- string / number helpers of the host driver (see utilities.h).
 *
 * Number syntax follows the reference so that command lines mean the same thing
 * (utilities.c:236-309 unitized integers, :334-355 named constants): an optional
 * K/M/G suffix scales by 10^3/10^6/10^9 (or powers of 1024), a fractional mantissa
 * is allowed with a suffix ("1.5K"), and the words inf / -inf / 1/inf stand for
 * DBL_MAX / -DBL_MAX / DBL_MIN -- not IEEE infinity. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <limits.h>
#include <float.h>
#include "utilities.h"

#define true  1
#define false 0

static void die (const char* what, const char* s)
	{
	fprintf (stderr, "\"%s\" %s\n", s, what);
	exit (EXIT_FAILURE);
	}

char* copy_string (const char* s)
	{
	if (s == NULL) return NULL;
	size_t n = strlen (s) + 1;
	char*  d = (char*) malloc (n);
	if (d == NULL) { fprintf (stderr, "out of memory copying \"%s\"\n", s);  exit (EXIT_FAILURE); }
	memcpy (d, s, n);
	return d;
	}

int strcmp_prefix (const char* str, const char* prefix)
	{ return strncmp (str, prefix, strlen (prefix)); }

int strcmp_suffix (const char* str, const char* suffix)
	{
	size_t n = strlen (str), m = strlen (suffix);
	if (m > n) return -1;
	return strcmp (str + n - m, suffix);
	}

int strncmp_suffix (const char* str, const char* suffix, size_t n)   /* utilities.c:104-116 in the reference */
	{
	size_t len = strlen (str), m = strlen (suffix);
	if (len > n) len = n;
	if (m > len) return strcmp (str, suffix);
	return strcmp (str + len - m, suffix);
	}

static const char* first_dark (const char* s)        /* the conversions skip leading blanks, tabs and newlines (utilities.c:143-146) */
	{ while ((*s == ' ') || (*s == '\t') || (*s == '\n')) s++;  return s; }

int string_to_int (const char* s)                    /* utilities.c:136-178: value, or the reference's three complaints */
	{
	int  v;
	char extra;
	const char* ss = first_dark (s);
	if (*ss == 0) { fprintf (stderr, "an empty string is not an integer\n");  exit (EXIT_FAILURE); }
	if (sscanf (ss, "%d%c", &v, &extra) != 1) die ("is not an integer", s);
	if (((v < 0) && (*ss != '-')) || ((v > 0) && (*ss == '-'))) die ("is outside the range of a signed integer", s);
	return v;
	}

/* 1 to 9 decimal digits and nothing else: what nearly every field of an interval file is.  Same
 * value as the general conversions below give, without the cost of sscanf (ingest is bound by it). */
static int plain_digits (const char* s, u32* v)
	{
	u32 x = 0;
	int n = 0;
	for ( ; (s[n] >= '0') && (s[n] <= '9') ; n++) { if (n == 9) return false;  x = 10*x + (u32) (s[n] - '0'); }
	if ((n == 0) || (s[n] != 0)) return false;
	*v = x;
	return true;
	}

int try_string_to_u32 (const char* s, u32* out)      /* string_to_u32 (utilities.c:181-213) without the exit */
	{
	u32  v;
	char extra;
	if (plain_digits (s, &v)) { *out = v;  return true; }
	const char* ss = first_dark (s);
	if ((*ss == 0) || (*ss == '-') || (sscanf (ss, "%u%c", &v, &extra) != 1)) return false;
	*out = v;
	return true;
	}

int string_to_u32 (const char* s)
	{
	u32 v;
	if (try_string_to_u32 (s, &v)) return (int) v;
	if (*first_dark (s) == 0) { fprintf (stderr, "an empty string is not an unsigned integer\n");  exit (EXIT_FAILURE); }
	die ("is not an unsigned integer", s);
	return 0;
	}

int string_to_unitized_int (const char* s, int byThousands)
	{
	char   body[20];                                   /* (the reference's buffer: longer strings are parsed as they are) */
	size_t len = strlen (s);
	long   mult = 1;
	int    v;
	float  vf;
	char   extra;

	if ((len > 0) && (len < sizeof(body)))
		{
		switch (s[len-1])
			{
			case 'K': case 'k': mult = byThousands? 1000L       : 1024L;                 break;
			case 'M': case 'm': mult = byThousands? 1000000L    : 1024L * 1024L;         break;
			case 'G': case 'g': mult = byThousands? 1000000000L : 1024L * 1024L * 1024L; break;
			}
		}
	if (mult != 1) { memcpy (body, s, len-1);  body[len-1] = 0; }
	const char* digits = (mult != 1)? body : s;

	if (sscanf (digits, "%d%c", &v, &extra) == 1)
		{
		if (mult != 1)
			{
			if ((v > 0) && ( v > INT_MAX / mult)) die ("is out of range for an integer", s);
			if ((v < 0) && (-v > INT_MAX / mult)) die ("is out of range for an integer", s);
			v *= (int) mult;
			}
		return v;
		}
	if (sscanf (digits, "%f%c", &vf, &extra) != 1) die ("is not an integer", s);
	if ((vf > 0) && ( vf*mult > INT_MAX)) die ("is out of range for an integer", s);
	if ((vf < 0) && (-vf*mult > INT_MAX)) die ("is out of range for an integer", s);
	return (int) ((vf * mult) + .5);
	}

int try_string_to_double (const char* s, double* v)
	{
	static const struct { const char* word;  int sign;  int puny; } named[] =
		{ {"inf",1,0}, {"+inf",1,0}, {"-inf",-1,0}, {"1/inf",1,1}, {"+1/inf",1,1}, {"-1/inf",-1,1} };
	const char* t = s;
	double      x;
	char        extra;

	u32 whole;
	if (plain_digits (s, &whole)) { if (v != NULL) *v = (double) whole;  return true; }   /* exact: < 2^53 */
	while ((*t == ' ') || (*t == '\t') || (*t == '\n')) t++;
	if (*t == 0) return false;
	for (size_t i=0 ; i<sizeof(named)/sizeof(named[0]) ; i++)
		{
		if (strcmp (s, named[i].word) != 0) continue;
		x = named[i].sign * (named[i].puny? DBL_MIN : DBL_MAX);
		if (v != NULL) *v = x;
		return true;
		}
	if (sscanf (s, "%lf%c", &x, &extra) != 1) return false;
	if (v != NULL) *v = x;
	return true;
	}

double string_to_double (const char* s)
	{
	double v;
	const char* t = s;
	while ((*t == ' ') || (*t == '\t') || (*t == '\n')) t++;
	if (*t == 0) { fprintf (stderr, "an empty string is not a number\n");  exit (EXIT_FAILURE); }
	if (!try_string_to_double (s, &v)) die ("is not a number", s);
	return v;
	}

char* skip_whitespace (char* s)
	{ while ((*s != 0) && ((*s == ' ') || (*s == '\t') || (*s == '\n') || (*s == '\r') || (*s == '\f') || (*s == '\v'))) s++;  return s; }

char* skip_darkspace (char* s)
	{ while ((*s != 0) && !((*s == ' ') || (*s == '\t') || (*s == '\n') || (*s == '\r') || (*s == '\f') || (*s == '\v'))) s++;  return s; }

/* a small ring of buffers so several calls can sit in one printf */
char* ucommatize (const u64 v)
	{
	static char ring[5][32];
	static int  next = 0;
	char  digits[24];
	char* out = ring[next];
	next = (next + 1) % 5;
	int n = snprintf (digits, sizeof(digits), "%" PRIu64, v);
	int o = 0;
	for (int i=0 ; i<n ; i++)
		{
		out[o++] = digits[i];
		if (((n - 1 - i) % 3 == 0) && (i != n-1)) out[o++] = ',';
		}
	out[o] = 0;
	return out;
	}

char* duration_to_string (float seconds)                           /* utilities.c:449-473 in the reference */
	{
	static char text[64];
	int whole = (int) (seconds / 60);
	if (seconds < 60)   { snprintf (text, sizeof(text), "%.3fs", seconds);  return text; }
	seconds -= 60 * whole;
	if (whole < 60) snprintf (text, sizeof(text), "%dm%06.3fs", whole, seconds);
	else            snprintf (text, sizeof(text), "%dh%02dm%06.3fs", whole / 60, whole % 60, seconds);
	return text;
	}

void safe_strncpy (char* dest, const char* src, size_t n)
	{
	strncpy (dest, src, n);
	dest[n] = 0;
	}
