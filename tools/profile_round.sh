#!/bin/bash
# The measurement pass behind profiles/ (round 4): the two PMC passes over the bench process itself (separate, kernel-trace only;
# their summary goes into profiles/ FIRST so that the bench line of this very run carries roofline.traffic for this tree's kernel
# sources), the bench line, rocprofv3 kernel trace + stats of the same command, the micro-benchmarks behind the split-precision
# kernels, and the five BASELINE configs.
# Run on the GPU box:   gpurun -- bash tools/profile_round.sh     (outputs under gpurun_out/r4p: copy the summaries into profiles/)
set -x
rm -rf gpurun_out/r4p; mkdir -p gpurun_out/r4p
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# 1. PMC passes (separate), on the bench process
BENCH_PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pop512 --no-roofline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r4p/pmc_fetch -- $BENCH_PMC > $R/gpurun_out/r4p/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r4p/pmc_write -- $BENCH_PMC > $R/gpurun_out/r4p/pmc_write.log 2>&1
(cd $R && python profiles/summarize_pmc_bench.py gpurun_out/r4p/pmc_fetch/*/*_results.db gpurun_out/r4p/pmc_write/*/*_results.db 512 gpurun_out/r4p/conv_pmc_traffic.json > gpurun_out/r4p/conv_pmc_traffic.txt 2>&1 && cp gpurun_out/r4p/conv_pmc_traffic.json profiles/round4_conv_pmc_traffic.json)
cat $R/gpurun_out/r4p/conv_pmc_traffic.txt | cut -c1-60,92-170
# 2. bench line (with cpu baseline), plain
python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r4p/bench.json 2> $R/gpurun_out/r4p/bench.err
# 3. kernel trace + stats of the same command (shorter)
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4p/prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pop512 > $R/gpurun_out/r4p/bench_prof.json 2> $R/gpurun_out/r4p/prof.log
python $R/profiles/summarize_rocprof.py $R/gpurun_out/r4p/prof/*/*_results.db > $R/gpurun_out/r4p/kernel_stats.txt
cd $R
# 4. micro-benchmarks (numerics of the operand split, f16 MFMA rate, LDS fill under the sharing patterns)
(cd tools/ubench && [ -x split_mfma ] || hipcc --offload-arch=gfx950 -O3 split_mfma.hip -o split_mfma -w; timeout 200 ./split_mfma 7 > ../../gpurun_out/r4p/split_mfma_ubench.txt 2>&1)
(cd tools/ubench && [ -x s3_loop ] || hipcc --offload-arch=gfx950 -O3 s3_loop.hip -o s3_loop -w; (echo "# tools/ubench/s3_loop 2048; s3_loop 1024   (main-loop prototype of the 128 x 128 six-sweep tile; all-zero operands: the part holds ~2.15 GHz here, ~1.6 under real data)"; timeout 200 ./s3_loop 2048; timeout 200 ./s3_loop 1024) > ../../gpurun_out/r4p/s3_loop_ubench.txt 2>&1)
# 5. all configs (the five of BASELINE.json + the reference's own operating points) + trunk accuracy
timeout 1200 python tools/run_configs.py --steps 3 2>&1 | grep -v "^[A-Za-z0-9]*: [a-z_]* = \|^$" > gpurun_out/r4p/run_configs.txt
timeout 400 python tools/trunk_accuracy.py > gpurun_out/r4p/trunk_accuracy.txt 2>&1
# 6. race hunt on the shipped conv kernels (every layer shape at 512 streams, 10 launches per algorithm) and random shapes
(echo "# python tools/conv_stress.py --streams 512 --reps 10 --modes 2,3,4,5,8,9"; timeout 900 python tools/conv_stress.py --streams 512 --reps 10 --modes 2,3,4,5,8,9 2>&1 | grep -v amdgpu.ids) > gpurun_out/r4p/conv_stress.txt
(echo "# python tools/conv_fuzz.py --cases 150 --seed 4"; timeout 600 python tools/conv_fuzz.py --cases 150 --seed 4 2>&1 | grep -v amdgpu.ids) > gpurun_out/r4p/conv_fuzz.txt
# 7. LDS / issue counters of the conv kernels (three separate PMC passes, kernel trace only), production mix (conv_bench mode 100)
cd /tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  d=$R/gpurun_out/r4p/ldspmc_$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python $R/tools/conv_bench.py --streams 512 --modes 100 --reps 1 > $d.log 2>&1
  python $R/tools/pmc_sum.py $d k_conv >> $R/gpurun_out/r4p/stream_lds_pmc.txt
  rm -rf $d
done
# 8. the CLI-default operating point (pop 32) under the kernel trace: which kernel binds there
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4p/prof32 -- python $R/tools/run_configs.py --only 7 > $R/gpurun_out/r4p/pop32_run.txt 2> $R/gpurun_out/r4p/prof32.log
python $R/profiles/summarize_rocprof.py $R/gpurun_out/r4p/prof32/*/*_results.db > $R/gpurun_out/r4p/pop32_kernel_stats.txt
rm -rf $R/gpurun_out/r4p/prof32
cd $R
head -30 gpurun_out/r4p/pop32_kernel_stats.txt | cut -c1-70,100-170
cat gpurun_out/r4p/stream_lds_pmc.txt | cut -c1-130
tail -n 2 gpurun_out/r4p/conv_stress.txt; tail -n 2 gpurun_out/r4p/conv_fuzz.txt
cat gpurun_out/r4p/run_configs.txt
head -34 gpurun_out/r4p/kernel_stats.txt | cut -c1-70,100-170
tail -4 gpurun_out/r4p/kernel_stats.txt
tail -c 600 gpurun_out/r4p/bench.json
# the raw rocpd databases stay on the box's scratch (the summaries above are what travels)
rm -rf gpurun_out/r4p/prof gpurun_out/r4p/pmc_fetch gpurun_out/r4p/pmc_write
