T=${TMPDIR:-/tmp}/gdsp_cli_bench
t() { local t0=$(date +%s.%N); "$@" > /dev/null 2>&1; local t1=$(date +%s.%N); python3 -c "print('%-40s %.3f s' % ('$LABEL', $t1 - $t0))"; }
LABEL="startup only (empty stdin)"; t genodsp_amd/genodsp_hip --chromosomes=$T/genome.chroms --novalue --nooutput < /dev/null
LABEL="ingest 5M lines, no output"; t genodsp_amd/genodsp_hip --chromosomes=$T/genome.chroms --novalue --nooutput < $T/reads.dat
LABEL="ingest + report (coverage)"; t genodsp_amd/genodsp_hip --chromosomes=$T/genome.chroms --novalue < $T/reads.dat
LABEL="cat reads.dat"; t cat $T/reads.dat
LABEL="wc -l reads.dat"; t wc -l $T/reads.dat
