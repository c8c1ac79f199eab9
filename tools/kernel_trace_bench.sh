R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pop512 > $O/bench_prof.json 2> $O/prof.log
python $R/profiles/summarize_rocprof.py $O/prof/*/*_results.db > $O/kernel_stats.txt
rm -rf $O/prof
tail -4 $O/kernel_stats.txt; python -c "
import json; d=json.load(open('$O/bench_prof.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], [o['avg_launch_ms'] for o in d['roofline_other']])"
