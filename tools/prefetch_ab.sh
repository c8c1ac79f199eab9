# next-item prefetch inside the persistent k_conv_wino43s (default build) against the same kernel without it (tools/ab_build.sh nopf -DS43_PREFETCH_NEXT=0)
cd $GRAFT_REPO_ROOT
for lib in st-ito_amd/st_ito/_lib/ab/libstito_hip_nopf.so ""; do
  echo "== lib=${lib:-default (prefetch)}"
  for rep in 1 2; do
  STITO_LIB_PATH=$lib python tools/conv_bench.py --streams 512 --modes 100 --reps 7 2>/dev/null | grep -E "117x32|58x16 256|total" | tr '\n' ' '; echo
  done
  STITO_LIB_PATH=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pop512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['last_fitness_sha16'])"
done
