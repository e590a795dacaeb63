/* ingest.c -- interval text in, a block at a time, parsed by a team of threads.
 *
 * read_interval (genodsp.c:1384-1534 in the reference) hands the driver and the interval-file
 * operators one interval per call.  Here it is a cursor over records that were parsed ahead:
 * the stream is read in blocks of 16 MiB, a block is cut at line ends into one stretch per
 * thread, every thread applies the reference's per-line rules to its stretch (track lines,
 * blank lines, comments, fields cut in place, the value column, the 1000-character line
 * buffer of the reference), and the records come back in file order.  A line the reference
 * would have stopped at becomes a record carrying the message; it is printed, and the process
 * ends, when the cursor reaches it -- so what is reported first is what the reference reports
 * first.  Line numbers are global across files, as the reference's static counter is.
 * With --progress=input:<n> or --report=comments, whose output is interleaved with the lines,
 * or GDSP_INGEST_THREADS=1, lines are read one at a time as in the reference. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
#include <unistd.h>
#include "genodsp_interface.h"
#include "utilities.h"
#include "host_services.h"

#define BLOCK_BYTES   (16u << 20)
#define LINE_LIMIT    1000                     /* the reference reads lines with fgets into char[1001] */
#define MAX_THREADS   32

typedef struct record
	{
	char*   chrom;                             /* in the block buffer, NUL terminated */
	u32     start, end;
	valtype val;
	u64     line;                              /* local to the stretch until the block is stitched */
	char*   problem;                           /* malloc'ed message: print, exit */
	int     numbered;                          /* the message starts with "problem at line <n>" */
	int     lineIsGlobal;                      /* `line` is owedLine, not a stretch-local number */
	} record;

typedef struct stretch
	{
	char*   from;  char* to;                   /* [from,to): whole lines; the last may lack its '\n' */
	int     valCol;
	record* recs;  size_t count, cap;
	u64     lines;
	int     owesAtStart, owesAtEnd;            /* a line without its '\n' before / at the end of this stretch (see parse_stretch) */
	u64     owedAt;
	} stretch;

typedef struct stream
	{
	FILE*   f;
	char*   block;     size_t carried;         /* bytes of an unfinished line kept from the last read */
	int     atEof;
	stretch part[MAX_THREADS];  int parts;
	int     cur;  size_t at;                   /* cursor: stretch and record */
	} stream;

static stream* open_streams[8];
static u64     lineNumber = 0;                 /* lines handed out so far, all files (reference: a static) */
static int     missingEol = false;             /* serial reader only */

/* the complaints of read_interval, word for word (genodsp.c:1511-1533) */
#define NO_CHROM "line contains no chromosome or begins with whitespace"
#define NO_START "line contains no interval start\n(expected \"chromosome start end ...\", but there are fewer than 2 fields)"
#define NO_END   "line contains no interval end\n(expected \"chromosome start end ...\", but there are fewer than 3 fields)"
#define NO_VALUE "line contains no interval value\n(expected \"chromosome start end value\", but there are fewer than 4 fields)"

static char* format_problem (const char* fmt, const char* a)
	{
	size_t n = strlen (fmt) + (a? strlen (a) : 0) + 8;
	char*  m = (char*) malloc (n);
	if (m == NULL) { fprintf (stderr, "out of memory\n");  exit (EXIT_FAILURE); }
	snprintf (m, n, fmt, a);
	return m;
	}

static record* new_record (stretch* st)
	{
	if (st->count == st->cap)
		{
		st->cap  = (st->cap == 0)? 4096 : 2*st->cap;
		st->recs = (record*) realloc (st->recs, st->cap * sizeof(record));
		if (st->recs == NULL) { fprintf (stderr, "out of memory buffering intervals\n");  exit (EXIT_FAILURE); }
		}
	record* r = &st->recs[st->count++];
	memset (r, 0, sizeof(*r));
	r->val = 1.0;
	return r;
	}

static void line_problem (stretch* st, u64 line, const char* what)       /* "problem at line <n>, <what>\n" */
	{
	record* r = new_record (st);
	r->line = line;  r->numbered = true;
	r->problem = format_problem ("%s", what);
	}

/* one line [s, s+len), NUL at s[len]: the rules of read_interval's tail */
static void parse_line (stretch* st, char* s, u64 line)
	{
	char *scan, *mark, *field;
	if (strcmp_prefix (s, "track ") == 0) return;
	scan = skip_whitespace (s);
	if ((*scan == 0) || (*scan == '#')) return;

	char* chrom = scan = s;
	if (*scan == ' ') { line_problem (st, line, NO_CHROM);  return; }
	mark = skip_darkspace (scan);  scan = skip_whitespace (mark);  if (*mark != 0) *mark = 0;
	if (*scan == 0) { line_problem (st, line, NO_START);  return; }
	field = scan;
	mark = skip_darkspace (scan);  scan = skip_whitespace (mark);  if (*mark != 0) *mark = 0;
	u32 start, end;
	if (!try_string_to_u32 (field, &start))
		{ record* r = new_record (st);  r->line = line;  r->problem = format_problem ("\"%s\" is not an unsigned integer\n", field);  return; }
	if (*scan == 0) { line_problem (st, line, NO_END);  return; }
	field = scan;
	mark = skip_darkspace (scan);  scan = skip_whitespace (mark);  if (*mark != 0) *mark = 0;
	if (!try_string_to_u32 (field, &end))
		{ record* r = new_record (st);  r->line = line;  r->problem = format_problem ("\"%s\" is not an unsigned integer\n", field);  return; }

	valtype val = 1.0;
	if (st->valCol != -1)
		{
		for (int col=3 ; col<=st->valCol ; col++)
			{
			if (*scan == 0) { line_problem (st, line, NO_VALUE);  return; }
			field = scan;
			mark = skip_darkspace (scan);  scan = skip_whitespace (mark);
			}
		if (*mark != 0) *mark = 0;
		if (!try_string_to_double (field, &val))
			{ record* r = new_record (st);  r->line = line;  r->problem = format_problem ("\"%s\" is not a number\n", field);  return; }
		}
	record* r = new_record (st);
	r->chrom = chrom;  r->start = start;  r->end = end;  r->val = val;  r->line = line;
	}

/* The reference reads with fgets into char[1001] and takes a line's length from strlen (genodsp.c:1407-1419): a line
 * whose visible text does not end in '\n' -- 1000 characters or more, or cut short by a NUL byte -- is parsed as it
 * stands, and the NEXT read, if there is one, stops with "line is longer than internal buffer" under this line's
 * number.  `owed` carries that debt from one line to the next (across blocks too; a block holding a NUL byte is
 * parsed by one thread so that the debt always has one owner). */
static int owedEol  = false;
static u64 owedLine = 0;                                  /* global line number of the line that owes its '\n' */

static void* parse_stretch (void* arg)
	{
	stretch* st = (stretch*) arg;
	char* s = st->from;
	int  owed = st->owesAtStart;
	u64  owedAt = 0;                                          /* local line number; 0 = the debt came in from the previous block */
	st->count = 0;  st->lines = 0;
	while (s < st->to)
		{
		char* nl  = (char*) memchr (s, '\n', (size_t) (st->to - s));
		char* end = (nl != NULL)? nl : st->to;             /* the line's characters are [s,end) */
		size_t len = (size_t) (end - s);
		st->lines++;
		if (owed)
			{
			record* r = new_record (st);
			r->numbered = true;  r->problem = format_problem ("%s", "line is longer than internal buffer");
			r->line = owedAt;  r->lineIsGlobal = (owedAt == 0);
			owed = false;
			}
		size_t chunk   = (len >= LINE_LIMIT)? LINE_LIMIT : len;        /* what fgets hands over first */
		size_t visible = strnlen (s, chunk);
		char   keep    = s[chunk];
		s[chunk] = 0;                                        /* (the byte after the last stretch belongs to the buffer) */
		parse_line (st, s, st->lines);
		s[chunk] = keep;
		if (len >= LINE_LIMIT)
			{
			/* the rest of the line (or its '\n') is the next read, unless the file ends right here */
			if ((len > LINE_LIMIT) || (nl != NULL))
				{
				if ((st->count == 0) || (st->recs[st->count-1].problem == NULL) || (st->recs[st->count-1].line != st->lines))
					line_problem (st, st->lines, "line is longer than internal buffer");
				}
			}
		else if ((visible < chunk) && (visible != 0))       /* cut short by a NUL byte: strlen sees no '\n' */
			{ owed = true;  owedAt = st->lines; }
		s = end + 1;
		}
	st->owesAtEnd = owed;  st->owedAt = owedAt;
	return NULL;
	}

static int team_size (void)
	{
	static int n = 0;
	if (n == 0)
		{
		const char* e = getenv ("GDSP_INGEST_THREADS");
		long cores = sysconf (_SC_NPROCESSORS_ONLN);
		n = (e != NULL)? atoi (e) : (int) ((cores > 16)? 16 : cores);
		if (n < 1) n = 1;
		if (n > MAX_THREADS) n = MAX_THREADS;
		}
	return n;
	}

/* read and parse the next block; false at the end of the file */
static int next_block (stream* sm, int valCol)
	{
	if (sm->atEof && (sm->carried == 0)) return false;
	size_t have = sm->carried;
	while (!sm->atEof && (have < BLOCK_BYTES))
		{
		size_t got = fread (sm->block + have, 1, BLOCK_BYTES - have, sm->f);
		if (got == 0) { sm->atEof = true;  break; }
		have += got;
		}
	if (have == 0) return false;
	/* whole lines only: what follows the last '\n' waits for the next read (unless this is the end) */
	size_t use = have;
	if (!sm->atEof)
		{
		char* last = NULL;
		for (size_t i=have ; i>0 ; i--) { if (sm->block[i-1] == '\n') { last = sm->block + i - 1;  break; } }
		if (last != NULL) use = (size_t) (last - sm->block) + 1;         /* (no '\n' at all: one over-long line) */
		}
	sm->block[have] = 0;

	int T = team_size ();
	if (use < (size_t) T * 65536) T = 1;
	if (memchr (sm->block, 0, use) != NULL) T = 1;             /* NUL bytes change what the reference takes for a line: one owner */
	sm->parts = 0;
	char* at = sm->block;  char* stop = sm->block + use;
	for (int t=0 ; (t<T) && (at<stop) ; t++)
		{
		char* to = stop;
		if (t < T-1)
			{
			char* aim = sm->block + (use / T) * (size_t) (t + 1);
			if (aim < at) aim = at;
			char* nl = (char*) memchr (aim, '\n', (size_t) (stop - aim));
			to = (nl != NULL)? nl + 1 : stop;
			}
		stretch* st = &sm->part[sm->parts++];
		st->from = at;  st->to = to;  st->valCol = valCol;
		st->owesAtStart = (t == 0)? owedEol : false;
		at = to;
		}
	pthread_t tid[MAX_THREADS];
	for (int t=1 ; t<sm->parts ; t++)
		{
		if (pthread_create (&tid[t], NULL, parse_stretch, &sm->part[t]) != 0)
			{ fprintf (stderr, "can't start an ingest thread\n");  exit (EXIT_FAILURE); }
		}
	parse_stretch (&sm->part[0]);
	for (int t=1 ; t<sm->parts ; t++) pthread_join (tid[t], NULL);

	/* stitch: line numbers become global.  Bytes past `use` are the start of an unfinished line: they
	 * move to the front once this block's records are spent (chrom pointers live in the block) */
	u64 base = lineNumber;
	for (int t=0 ; t<sm->parts ; t++)
		{
		stretch* st = &sm->part[t];
		for (size_t i=0 ; i<st->count ; i++)
			{
			if (st->recs[i].lineIsGlobal) st->recs[i].line = owedLine;
			else                          st->recs[i].line += base;
			}
		if (t == sm->parts-1) { owedEol = st->owesAtEnd;  if (owedEol && (st->owedAt != 0)) owedLine = base + st->owedAt; }
		base += st->lines;
		}
	lineNumber = base;
	sm->carried = have - use;
	if (sm->carried != 0) memmove (sm->block + BLOCK_BYTES + 8, sm->block + use, sm->carried);   /* parked behind the block */
	sm->cur = 0;  sm->at = 0;
	return true;
	}

static stream* stream_of (FILE* f)
	{
	int free_slot = -1;
	for (int i=0 ; i<8 ; i++)
		{
		if ((open_streams[i] != NULL) && (open_streams[i]->f == f)) return open_streams[i];
		if ((open_streams[i] == NULL) && (free_slot < 0)) free_slot = i;
		}
	if (free_slot < 0) { fprintf (stderr, "too many interval files open at once\n");  exit (EXIT_FAILURE); }
	stream* sm = (stream*) calloc (1, sizeof(stream));
	if (sm != NULL) sm->block = (char*) malloc (2 * (size_t) BLOCK_BYTES + 16);
	if ((sm == NULL) || (sm->block == NULL)) { fprintf (stderr, "out of memory for the ingest buffer\n");  exit (EXIT_FAILURE); }
	sm->f = f;
	open_streams[free_slot] = sm;
	return sm;
	}

static void close_stream (stream* sm)
	{
	for (int i=0 ; i<8 ; i++) { if (open_streams[i] == sm) open_streams[i] = NULL; }
	for (int t=0 ; t<MAX_THREADS ; t++) free (sm->part[t].recs);
	free (sm->block);
	free (sm);
	}

static int read_interval_serial (FILE* f, char* buffer, int bufferLen, int valCol,
                                 char** _chrom, u32* _start, u32* _end, valtype* _val);

int read_interval (FILE* f, char* buffer, int bufferLen, int valCol,     /* genodsp.c:1384-1534 */
                   char** _chrom, u32* _start, u32* _end, valtype* _val)
	{
	if ((reportInputProgress != 0) || reportComments || dbgInput || (team_size () == 1))
		return read_interval_serial (f, buffer, bufferLen, valCol, _chrom, _start, _end, _val);

	stream* sm = stream_of (f);
	if ((_val == NULL) && (valCol != -1)) valCol = -1;
	for (;;)
		{
		while ((sm->cur < sm->parts) && (sm->at >= sm->part[sm->cur].count)) { sm->cur++;  sm->at = 0; }
		if (sm->cur < sm->parts) break;
		/* this block is spent: bring the unfinished line to the front and read on */
		if (sm->carried != 0) memmove (sm->block, sm->block + BLOCK_BYTES + 8, sm->carried);
		if (!next_block (sm, valCol)) { close_stream (sm);  return false; }
		}
	record* r = &sm->part[sm->cur].recs[sm->at++];
	if (r->problem != NULL)
		{
		if (r->numbered) fprintf (stderr, "problem at line %s, %s\n", ucommatize (r->line), r->problem);
		else             fprintf (stderr, "%s", r->problem);
		exit (EXIT_FAILURE);
		}
	if (_chrom != NULL) *_chrom = r->chrom;
	if (_start != NULL) *_start = r->start;
	if (_end   != NULL) *_end   = r->end;
	if (_val   != NULL) *_val   = r->val;
	return true;
	}

/* one line at a time, exactly the reference's loop */
static int read_interval_serial (FILE* f, char* buffer, int bufferLen, int valCol,
                                 char** _chrom, u32* _start, u32* _end, valtype* _val)
	{
	char *scan, *mark, *field;

	for (;;)
		{
		if (fgets (buffer, bufferLen, f) == NULL) return false;
		lineNumber++;
		if (missingEol)
			{ fprintf (stderr, "problem at line %s, line is longer than internal buffer\n", ucommatize (lineNumber-1));  exit (EXIT_FAILURE); }
		size_t len = strlen (buffer);
		if (len != 0) missingEol = (buffer[len-1] != '\n');
		if (dbgInput) fprintf (stderr, "input = \"%s\"\n", buffer);      /* genodsp.c:1422 */
		if (strcmp_prefix (buffer, "track ") == 0) continue;

		int progressNow = (reportInputProgress != 0)
		               && ((lineNumber == 1) || (lineNumber % reportInputProgress == 0));
		scan = skip_whitespace (buffer);
		if (*scan == 0)
			{ if (progressNow) fprintf (stderr, "progress: input line %s\n", ucommatize (lineNumber));  continue; }
		if (*scan == '#')
			{
			if (reportComments)   fprintf (stderr, "input line %s: %s", ucommatize (lineNumber), scan);
			else if (progressNow) fprintf (stderr, "progress: input line %s\n", ucommatize (lineNumber));
			continue;
			}
		if (progressNow) fprintf (stderr, "progress: input line %s\n", ucommatize (lineNumber));
		break;
		}

	char* chrom = scan = buffer;
	if (*scan == ' ')
		{ fprintf (stderr, "problem at line %s, " NO_CHROM "\n", ucommatize (lineNumber));  exit (EXIT_FAILURE); }
	mark = skip_darkspace (scan);  scan = skip_whitespace (mark);  if (*mark != 0) *mark = 0;
	if (*scan == 0)
		{ fprintf (stderr, "problem at line %s, " NO_START "\n", ucommatize (lineNumber));  exit (EXIT_FAILURE); }
	field = scan;
	mark = skip_darkspace (scan);  scan = skip_whitespace (mark);  if (*mark != 0) *mark = 0;
	u32 start = (u32) string_to_u32 (field);
	if (*scan == 0)
		{ fprintf (stderr, "problem at line %s, " NO_END "\n", ucommatize (lineNumber));  exit (EXIT_FAILURE); }
	field = scan;
	mark = skip_darkspace (scan);  scan = skip_whitespace (mark);  if (*mark != 0) *mark = 0;
	u32 end = (u32) string_to_u32 (field);

	valtype val = 1.0;
	if ((valCol != -1) && (_val != NULL))
		{
		for (int col=3 ; col<=valCol ; col++)
			{
			if (*scan == 0)
				{ fprintf (stderr, "problem at line %s, " NO_VALUE "\n", ucommatize (lineNumber));  exit (EXIT_FAILURE); }
			field = scan;
			mark = skip_darkspace (scan);  scan = skip_whitespace (mark);
			}
		if (*mark != 0) *mark = 0;
		val = string_to_valtype (field);
		}
	if (_chrom != NULL) *_chrom = chrom;
	if (_start != NULL) *_start = start;
	if (_end   != NULL) *_end   = end;
	if (_val   != NULL) *_val   = val;
	return true;
	}
