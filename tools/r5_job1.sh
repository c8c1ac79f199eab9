#!/bin/bash
# round 5, GPU job 1: (a) does the hipGraph corruption reproduce (fresh processes), (b) EA-side counters of the conv launches of the
# bench process, (c) the same counters on the MALL calibration micro-benchmark
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; rm -rf $O; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# (a)
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 300 python $R/tools/graph_soak.py --pop 32 --samples 96000 --seed $i 2>&1 | grep graph_soak >> $O/graph_soak_baseline.txt; done
for i in 1 2 3 4; do timeout 300 python $R/tools/graph_soak.py --pop 256 --samples 262144 --replays 6 --seed $i 2>&1 | grep graph_soak >> $O/graph_soak_baseline.txt; done
cat $O/graph_soak_baseline.txt
# (b)
BENCH_PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pop512 --no-roofline"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $O/ea1 -- $BENCH_PMC > $O/ea1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --kernel-trace -d $O/ea2 -- $BENCH_PMC > $O/ea2.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum --kernel-trace -d $O/ea3 -- $BENCH_PMC > $O/ea3.log 2>&1
python $R/profiles/summarize_pmc_ea.py $O/ea1/*/*_results.db $O/ea2/*/*_results.db $O/ea3/*/*_results.db > $O/conv_ea_pmc.txt 2>&1
cat $O/conv_ea_pmc.txt | cut -c1-400
# (c)
$R/tools/ubench/mall_probe > $O/mall_probe.txt 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $O/mp1 -- $R/tools/ubench/mall_probe > $O/mp1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --kernel-trace -d $O/mp2 -- $R/tools/ubench/mall_probe > $O/mp2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/mp3 -- $R/tools/ubench/mall_probe > $O/mp3.log 2>&1
python $R/profiles/summarize_pmc_ea.py --all "k_probe<1>" $O/mp1/*/*_results.db $O/mp2/*/*_results.db $O/mp3/*/*_results.db > $O/mall_probe_pmc.txt 2>&1
cat $O/mall_probe.txt $O/mall_probe_pmc.txt
tail -3 $O/ea1.log $O/mp1.log
rm -rf $O/ea1 $O/ea2 $O/ea3 $O/mp1 $O/mp2 $O/mp3
