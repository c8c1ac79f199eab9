#!/bin/bash
# The measurement pass behind profiles/ (round 3): the two PMC passes over the bench process itself (separate, kernel-trace only;
# their summary goes into profiles/ FIRST so that the bench line of this very run carries roofline.traffic for this tree's kernel
# sources), the bench line, rocprofv3 kernel trace + stats of the same command, the micro-benchmarks behind the split-precision
# kernels, and the five BASELINE configs.
# Run on the GPU box:   gpurun -- bash tools/profile_round.sh     (outputs under gpurun_out/r3p: copy the summaries into profiles/)
set -x
rm -rf gpurun_out/r3p; mkdir -p gpurun_out/r3p
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# 1. PMC passes (separate), on the bench process
BENCH_PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pop512 --no-roofline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r3p/pmc_fetch -- $BENCH_PMC > $R/gpurun_out/r3p/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r3p/pmc_write -- $BENCH_PMC > $R/gpurun_out/r3p/pmc_write.log 2>&1
(cd $R && python profiles/summarize_pmc_bench.py gpurun_out/r3p/pmc_fetch/*/*_results.db gpurun_out/r3p/pmc_write/*/*_results.db 512 gpurun_out/r3p/conv_pmc_traffic.json > gpurun_out/r3p/conv_pmc_traffic.txt 2>&1 && cp gpurun_out/r3p/conv_pmc_traffic.json profiles/round3_conv_pmc_traffic.json)
cat $R/gpurun_out/r3p/conv_pmc_traffic.txt | cut -c1-60,92-170
# 2. bench line (with cpu baseline), plain
python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r3p/bench.json 2> $R/gpurun_out/r3p/bench.err
# 3. kernel trace + stats of the same command (shorter)
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3p/prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pop512 > $R/gpurun_out/r3p/bench_prof.json 2> $R/gpurun_out/r3p/prof.log
python $R/profiles/summarize_rocprof.py $R/gpurun_out/r3p/prof/*/*_results.db > $R/gpurun_out/r3p/kernel_stats.txt
cd $R
# 4. micro-benchmarks (numerics of the operand split, f16 MFMA rate, LDS fill under the sharing patterns)
(cd tools/ubench && [ -x split_mfma ] || hipcc --offload-arch=gfx950 -O3 split_mfma.hip -o split_mfma -w; timeout 200 ./split_mfma 7 > ../../gpurun_out/r3p/split_mfma_ubench.txt 2>&1)
# 5. all configs + trunk accuracy
timeout 900 python tools/run_configs.py --steps 3 > gpurun_out/r3p/run_configs.txt 2>&1
timeout 400 python tools/trunk_accuracy.py > gpurun_out/r3p/trunk_accuracy.txt 2>&1
cat gpurun_out/r3p/run_configs.txt
head -34 gpurun_out/r3p/kernel_stats.txt | cut -c1-70,100-170
tail -4 gpurun_out/r3p/kernel_stats.txt
tail -c 600 gpurun_out/r3p/bench.json
# the raw rocpd databases stay on the box's scratch (the summaries above are what travels)
rm -rf gpurun_out/r3p/prof gpurun_out/r3p/pmc_fetch gpurun_out/r3p/pmc_write
