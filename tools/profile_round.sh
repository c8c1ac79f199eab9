#!/bin/bash
# The measurement pass behind profiles/ (round 6; round 5's recipe + the small-population A/B and the chunked-schedule sweep).  One command on one box:   gpurun -- bash tools/profile_round.sh
# Outputs under gpurun_out/r6p: copy the summaries into profiles/.  Order: the PMC passes over the bench process itself first (separate
# passes, kernel trace only, eager launches so that every kernel is a dispatch the counters are attributed to) -- their summaries go
# into profiles/ BEFORE the bench line is taken, so that the line of this very run carries roofline.traffic / roofline_dsp.traffic for
# this tree's kernel sources -- then the bench line, rocprofv3 kernel trace + stats of the same command, the EA-side counters
# (requests, latency) of the conv launches with their Infinity-Cache / HBM calibration, the configs, accuracy, race hunt, LDS counters.
set -x
rm -rf gpurun_out/r6p; mkdir -p gpurun_out/r6p
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6p
cd /tmp && export TMPDIR=/tmp
# 1. PMC passes (separate), on the bench process, eager launches
BENCH_PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pop512 --no-roofline"
STITO_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -- $BENCH_PMC > $O/pmc_fetch.log 2>&1
STITO_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -- $BENCH_PMC > $O/pmc_write.log 2>&1
(cd $R && python profiles/summarize_pmc_bench.py gpurun_out/r6p/pmc_fetch/*/*_results.db gpurun_out/r6p/pmc_write/*/*_results.db 512 gpurun_out/r6p/conv_pmc_traffic.json > gpurun_out/r6p/conv_pmc_traffic.txt 2>&1 && cp gpurun_out/r6p/conv_pmc_traffic.json profiles/round6_conv_pmc_traffic.json)
(cd $R && python profiles/summarize_pmc_dsp.py gpurun_out/r6p/pmc_fetch/*/*_results.db gpurun_out/r6p/pmc_write/*/*_results.db 256 480000 gpurun_out/r6p/dsp_pmc_traffic.json > gpurun_out/r6p/dsp_pmc_traffic.txt 2>&1 && cp gpurun_out/r6p/dsp_pmc_traffic.json profiles/round6_dsp_pmc_traffic.json)
cat $O/conv_pmc_traffic.txt | cut -c1-60,92-170; cat $O/dsp_pmc_traffic.txt
# 2. bench line (with cpu baseline), plain
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
# 3. kernel trace + stats of the same command (shorter)
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pop512 > $O/bench_prof.json 2> $O/prof.log
python $R/profiles/summarize_rocprof.py $O/prof/*/*_results.db > $O/kernel_stats.txt
# 4. EA-side counters of the conv launches (requests, in-flight level -> latency) + the Infinity-Cache / HBM calibration curve under the same counters
STITO_GRAPH=0 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $O/ea1 -- $BENCH_PMC > $O/ea1.log 2>&1
STITO_GRAPH=0 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --kernel-trace -d $O/ea2 -- $BENCH_PMC > $O/ea2.log 2>&1
STITO_GRAPH=0 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum --kernel-trace -d $O/ea3 -- $BENCH_PMC > $O/ea3.log 2>&1
python $R/profiles/summarize_pmc_ea.py $O/ea1/*/*_results.db $O/ea2/*/*_results.db $O/ea3/*/*_results.db > $O/conv_ea_pmc.txt 2>&1
(cd $R/tools/ubench && [ -x mall_probe ] || hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe -w)
$R/tools/ubench/mall_probe curve > $O/mall_curve.txt 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $O/mp1 -- $R/tools/ubench/mall_probe curve > $O/mp1.log 2>&1
python $R/profiles/summarize_pmc_ea.py --all "k_probe<2>" $O/mp1/*/*_results.db > $O/mall_curve_pmc.txt 2>&1
cd $R
# 5. all configs (the five of BASELINE.json + the reference's own operating points) + trunk accuracy + small-population steady state
timeout 1200 python tools/run_configs.py --steps 3 2>&1 | grep -v "^[A-Za-z0-9]*: [a-z_]* = \|^$" > $O/run_configs.txt
timeout 400 python tools/trunk_accuracy.py > $O/trunk_accuracy.txt 2>&1
for pop in 32 64 128; do for gr in 1 0; do
  echo -n "pop $pop, 262144 samples, bench chain, STITO_GRAPH=$gr: cand/s, ms/step, host ms (ask + launch, sync, tell): " >> $O/small_pop.txt
  STITO_GRAPH=$gr python bench.py --pop-per-gpu $pop --seconds 5.4613 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-pop512 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stages']['evaluate_ms']['max'], d['stages']['gather_ms']['max'], d['stages']['tell_ms']['max'])" >> $O/small_pop.txt
done; done
# 5b. round 6: depth-first trunk schedule sweep + its EA counters are tools/chunk_sweep.py / tools/chunk_ea.sh (run on their own);
#     small populations: deep layers two-sweep / six-sweep, sweeps in one workgroup or split (tools/small_conv_ab.sh)
bash tools/small_conv_ab.sh > $O/small_conv_ab.txt 2>&1
# 6. race hunt on the shipped conv kernels (every layer shape at 512 streams, 10 launches per algorithm) and random shapes
(echo "# python tools/conv_stress.py --streams 512 --reps 10 --modes 2,3,4,5,8,9"; timeout 900 python tools/conv_stress.py --streams 512 --reps 10 --modes 2,3,4,5,8,9 2>&1 | grep -v amdgpu.ids) > $O/conv_stress.txt
(echo "# python tools/conv_fuzz.py --cases 150 --seed 5"; timeout 600 python tools/conv_fuzz.py --cases 150 --seed 5 2>&1 | grep -v amdgpu.ids) > $O/conv_fuzz.txt
# 7. graph soak: 12 fresh processes x 20 replays at pop 32, 4 x 6 at pop 256 (the in-suite test runs 50 smaller ones)
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 300 python tools/graph_soak.py --pop 32 --samples 96000 --seed $i 2>&1 | grep graph_soak >> $O/graph_soak.txt; done
for i in 1 2 3 4; do timeout 300 python tools/graph_soak.py --pop 256 --samples 262144 --replays 6 --seed $i 2>&1 | grep graph_soak >> $O/graph_soak.txt; done
# 8. the CLI-default operating point (pop 32) under the kernel trace: which kernel binds there
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof32 -- python $R/tools/run_configs.py --only 7 > $O/pop32_run.txt 2> $O/prof32.log
python $R/profiles/summarize_rocprof.py $O/prof32/*/*_results.db > $O/pop32_kernel_stats.txt
cd $R
head -30 $O/pop32_kernel_stats.txt | cut -c1-70,100-170
cat $O/small_pop.txt $O/graph_soak.txt
tail -n 2 $O/conv_stress.txt; tail -n 2 $O/conv_fuzz.txt
cat $O/run_configs.txt
head -34 $O/kernel_stats.txt | cut -c1-70,100-170
tail -4 $O/kernel_stats.txt
cat $O/mall_curve_pmc.txt | cut -c1-60,140-220
tail -c 700 $O/bench.json
# the raw rocpd databases stay on the box's scratch (the summaries above are what travels)
rm -rf $O/prof $O/prof32 $O/pmc_fetch $O/pmc_write $O/ea1 $O/ea2 $O/ea3 $O/mp1
