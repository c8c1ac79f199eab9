# pop-32 steady state under the kernel trace (bench chain, 262 144 samples): which kernels bind a small population
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pop32; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STITO_GRAPH=0 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --pop-per-gpu 32 --seconds 5.4613 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-pop512 > $O/bench.json 2> $O/prof.log
python $R/profiles/summarize_rocprof.py $O/prof/*/*_results.db > $O/pop32_kernel_stats.txt
rm -rf $O/prof
head -36 $O/pop32_kernel_stats.txt | cut -c1-60,100-160
