# k_conv_wino43s (STITO_W43_QUEUES) / the f32 k_conv_wino43 (STITO_W43_QUEUES_F32) layers as N grids on N streams (1 = one grid): bench step, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for q in "1 1" "2 1" "1 2" "2 2"; do
  set -- $q
  echo -n "STITO_W43_QUEUES=$1 STITO_W43_QUEUES_F32=$2: "; STITO_W43_QUEUES=$1 STITO_W43_QUEUES_F32=$2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-pop512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['last_fitness_sha16'])"
done; done
