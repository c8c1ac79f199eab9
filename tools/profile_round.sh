#!/bin/bash
# The measurement pass behind profiles/: the two PMC passes over the conv launches (separate, kernel-trace only; their
# summary goes into profiles/ FIRST so that the bench line of this very run carries roofline.traffic for this tree's kernel
# sources), the bench line, rocprofv3 kernel trace + stats of the same command, and the five BASELINE configs.
# Run on the GPU box:   gpurun -- bash tools/profile_round.sh     (outputs under gpurun_out/r2p: copy the summaries into profiles/)
set -x
rm -rf gpurun_out/r2p; mkdir -p gpurun_out/r2p
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# 1. PMC passes (separate), conv launches only
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r2p/pmc_fetch -- python $R/tools/conv_bench.py --streams 512 --reps 1 --modes 9 > $R/gpurun_out/r2p/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r2p/pmc_write -- python $R/tools/conv_bench.py --streams 512 --reps 1 --modes 9 > $R/gpurun_out/r2p/pmc_write.log 2>&1
(cd $R && python profiles/summarize_pmc.py gpurun_out/r2p/pmc_fetch/*/*_results.db gpurun_out/r2p/pmc_write/*/*_results.db 512 gpurun_out/r2p/conv_pmc_traffic.json > gpurun_out/r2p/conv_pmc_traffic.txt 2>&1 && cp gpurun_out/r2p/conv_pmc_traffic.json profiles/round2_conv_pmc_traffic.json)
cat $R/gpurun_out/r2p/conv_pmc_traffic.txt
# 2. bench line (with cpu baseline), plain
python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r2p/bench.json 2> $R/gpurun_out/r2p/bench.err
# 3. kernel trace + stats of the same command (shorter)
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2p/prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r2p/bench_prof.json 2> $R/gpurun_out/r2p/prof.log
python $R/profiles/summarize_rocprof.py $R/gpurun_out/r2p/prof/*/*_results.db > $R/gpurun_out/r2p/kernel_stats.txt
cd $R
# 4. all configs
timeout 600 python tools/run_configs.py --steps 3 > gpurun_out/r2p/run_configs.txt 2>&1
cat gpurun_out/r2p/run_configs.txt
head -30 gpurun_out/r2p/kernel_stats.txt | cut -c1-70,100-170
tail -1 gpurun_out/r2p/kernel_stats.txt
tail -c 400 gpurun_out/r2p/bench.json
# the raw rocpd databases stay on the box's scratch (the summaries above are what travels)
rm -rf gpurun_out/r2p/prof gpurun_out/r2p/pmc_fetch gpurun_out/r2p/pmc_write
