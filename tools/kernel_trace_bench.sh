R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pop512 > $O/bench_prof.json 2> $O/prof.log
python $R/profiles/summarize_rocprof.py $O/prof/*/*_results.db > $O/kernel_stats.txt
rm -rf $O/prof
head -40 $O/kernel_stats.txt | cut -c1-64,100-165
cd $R; timeout 300 python -m pytest tests/test_gpu_es.py -m gpu -x -q -k "case_study" 2>&1 | tail -2
