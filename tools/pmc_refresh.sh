#!/bin/bash
# Retake the two PMC passes (FETCH_SIZE, WRITE_SIZE) on the bench process and rewrite profiles/round5_{conv,dsp}_pmc_traffic.* for the
# kernel sources in this tree (bench.py quotes traffic only while the hashes match):   gpurun -- bash tools/pmc_refresh.sh
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_refresh; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH_PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pop512 --no-roofline"
STITO_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $BENCH_PMC > $O/pmc_fetch.log 2>&1
STITO_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $BENCH_PMC > $O/pmc_write.log 2>&1
cd $R
python profiles/summarize_pmc_bench.py $O/pmc_fetch/*/*_results.db $O/pmc_write/*/*_results.db 512 $O/conv_pmc_traffic.json > $O/conv_pmc_traffic.txt 2>&1
python profiles/summarize_pmc_dsp.py $O/pmc_fetch/*/*_results.db $O/pmc_write/*/*_results.db 256 480000 $O/dsp_pmc_traffic.json > $O/dsp_pmc_traffic.txt 2>&1
cat $O/dsp_pmc_traffic.txt; tail -2 $O/conv_pmc_traffic.txt | cut -c1-40,92-170
rm -rf $O/pmc_fetch $O/pmc_write
