# persistent workgroups (k_conv_wino43s, the f32 k_conv_wino43, the transform passes) against one workgroup per item: per-layer times
# (conv_bench, production mix, random data) and the bench line
cd $GRAFT_REPO_ROOT
for p in "0 0 0" "1 1 0" "1 1 1"; do
  set -- $p
  echo "== STITO_W43S_PERSIST=$1 STITO_W43_PERSIST=$2 STITO_W43T_PERSIST=$3"
  STITO_W43S_PERSIST=$1 STITO_W43_PERSIST=$2 STITO_W43T_PERSIST=$3 python tools/conv_bench.py --streams 512 --modes 100 --reps 7 2>/dev/null | grep -vE "^layer|amdgpu"
  STITO_W43S_PERSIST=$1 STITO_W43_PERSIST=$2 STITO_W43T_PERSIST=$3 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pop512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['last_fitness_sha16'])"
done
