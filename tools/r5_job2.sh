#!/bin/bash
# round 5, GPU job 2: graph soak on the kernel-zeroed library, new tests, bench line, s3 2-D XCD partition under the EA counters, MALL curve
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; rm -rf $O; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 300 python $R/tools/graph_soak.py --pop 32 --samples 96000 --seed $i 2>&1 | grep graph_soak >> $O/graph_soak.txt; done
for i in 1 2; do timeout 300 python $R/tools/graph_soak.py --pop 256 --samples 262144 --replays 6 --seed $i 2>&1 | grep graph_soak >> $O/graph_soak.txt; done
cat $O/graph_soak.txt
cd $R
timeout 1500 python -m pytest tests/test_gpu_es.py -x -q -m gpu 2>&1 | tail -15 > $O/test_gpu_es.txt; cat $O/test_gpu_es.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv" 2>&1 | tail -5 > $O/test_conv.txt; cat $O/test_conv.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -5 $O/bench.err
STITO_GRAPH=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pop512 > $O/bench_eager.json 2>> $O/bench.err; cut -c1-300 $O/bench_eager.json
cd /tmp
BENCH_PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pop512 --no-roofline"
STITO_GRAPH=0 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $O/ea1 -- $BENCH_PMC > $O/ea1.log 2>&1
python $R/profiles/summarize_pmc_ea.py $O/ea1/*/*_results.db > $O/conv_ea_pmc_2d.txt 2>&1
STITO_W43S3_XM=8 STITO_GRAPH=0 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $O/ea2 -- $BENCH_PMC > $O/ea2.log 2>&1
python $R/profiles/summarize_pmc_ea.py $O/ea2/*/*_results.db > $O/conv_ea_pmc_xm8.txt 2>&1
tail -6 $O/conv_ea_pmc_2d.txt | cut -c1-160; tail -6 $O/conv_ea_pmc_xm8.txt | cut -c1-160
$R/tools/ubench/mall_probe curve > $O/mall_curve.txt 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $O/mp1 -- $R/tools/ubench/mall_probe curve > $O/mp1.log 2>&1
python $R/profiles/summarize_pmc_ea.py --all "k_probe<2>" $O/mp1/*/*_results.db > $O/mall_curve_pmc.txt 2>&1
cat $O/mall_curve.txt; cut -c1-60,140-220 $O/mall_curve_pmc.txt
rm -rf $O/ea1 $O/ea2 $O/mp1
