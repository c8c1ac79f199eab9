set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/chunk_ea; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in 0:2:6 16:2:6 32:2:6 32:5:5; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $O/ea_$cfg -- python $R/tools/chunk_sweep.py --once $cfg > $O/ea_$cfg.log 2>&1
  (echo "# schedule chunk:first:last = $cfg (0 = layer by layer), 512 streams"; python $R/profiles/summarize_pmc_ea_sum.py $O/ea_$cfg/*/*_results.db) >> $O/chunk_ea.txt 2>&1
  rm -rf $O/ea_$cfg
done
cat $O/chunk_ea.txt
