/* genome_reads.c -- seeded read intervals over the 24-chromosome, 3 088 269 832-base genome of BASELINE.json's
 * configs[1..4] (hg38-like lengths), as `chrom<TAB>start<TAB>end` lines for `--novalue`.
 *
 *   genome_reads <chromosomes file to write> [seed] > intervals
 *
 * The same bytes wherever it runs (counter-free splitmix64 streams keyed by seed and chromosome), so the build
 * container can record what the reference prints for this input (tools/make_genome_golden.py) and the GPU box can
 * regenerate the input instead of shipping it.  The signal is ChIP-like: clusters of 40..259 overlapping reads every
 * 8..72 kbp, single background reads in between; about 12 M reads, ~5 % of the genome covered.  Reads inside a
 * cluster are not sorted (the drivers must not care).
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

static const char*    names[24] = { "chr1","chr2","chr3","chr4","chr5","chr6","chr7","chr8","chr9","chr10","chr11","chr12",
                                    "chr13","chr14","chr15","chr16","chr17","chr18","chr19","chr20","chr21","chr22","chrX","chrY" };
static const uint32_t lens[24]  = { 248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                                    138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                                    83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415 };

static uint64_t state;
static uint64_t next64 (void)
	{
	uint64_t z = (state += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
	}
static uint32_t below (uint32_t n) { return (uint32_t) ((next64 () >> 32) * (uint64_t) n >> 32); }

static char  buf[(1 << 20) + 256];
static char* at = buf;

static char* put_u32 (char* p, uint32_t u)
	{
	char d[12];
	int  n = 0;
	do { d[n++] = (char) ('0' + u % 10);  u /= 10; } while (u != 0);
	while (n > 0) *(p++) = d[--n];
	return p;
	}

static void emit (const char* chrom, size_t chromLen, uint32_t start, uint32_t end)
	{
	memcpy (at, chrom, chromLen);  at += chromLen;
	*(at++) = '\t';  at = put_u32 (at, start);
	*(at++) = '\t';  at = put_u32 (at, end);
	*(at++) = '\n';
	if (at - buf > (1 << 20)) { fwrite (buf, 1, (size_t) (at - buf), stdout);  at = buf; }
	}

int main (int argc, char** argv)
	{
	if (argc < 2) { fprintf (stderr, "usage: genome_reads <chromosomes file to write> [seed] > intervals\n");  return 1; }
	uint64_t seed = (argc > 2)? strtoull (argv[2], NULL, 10) : 20240611ULL;
	FILE* cf = fopen (argv[1], "wt");
	if (cf == NULL) { fprintf (stderr, "can't write %s\n", argv[1]);  return 1; }
	for (int c=0 ; c<24 ; c++) fprintf (cf, "%s %u\n", names[c], lens[c]);
	fclose (cf);

	for (int c=0 ; c<24 ; c++)
		{
		const uint32_t L = lens[c];
		const size_t   nameLen = strlen (names[c]);
		state = seed * 0x2545F4914F6CDD1DULL + (uint64_t) (c + 1) * 0xD6E8FEB86659FD93ULL;
		uint32_t pos = 2000 + below (20000);
		while (pos + 2000 < L)
			{
			/* a cluster centred on pos */
			uint32_t reads = 40 + below (220);
			for (uint32_t r=0 ; r<reads ; r++)
				{
				uint32_t off   = below (300) + below (300) + below (300);        /* 0..897, bell shaped around 450 */
				uint32_t start = pos - 450 + off;
				uint32_t len   = 36 + below (115);
				uint32_t end   = (start + len > L)? L : start + len;
				emit (names[c], nameLen, start, end);
				}
			/* the gap to the next cluster, with background reads */
			uint32_t gap = 8000 + below (64000);
			uint32_t bg  = gap / 8000;
			for (uint32_t r=0 ; r<bg ; r++)
				{
				uint32_t start = pos + 1000 + below (gap - 1000);
				uint32_t len   = 36 + below (115);
				if (start >= L) continue;
				uint32_t end   = (start + len > L)? L : start + len;
				emit (names[c], nameLen, start, end);
				}
			pos += gap;
			}
		}
	fwrite (buf, 1, (size_t) (at - buf), stdout);
	return 0;
	}
