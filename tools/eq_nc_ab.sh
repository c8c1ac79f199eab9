# A/B of k_eq's chunk count (EQ_NC_BUILD): tools/ab_build.sh eq512 -DEQ_NC_BUILD=512 first
cd $GRAFT_REPO_ROOT
for lib in "" st-ito_amd/st_ito/_lib/ab/libstito_hip_eq512.so; do
  for cfg in "32 5.4613 2" "256 10 2" "512 10 2" "256 30 2"; do
    set -- $cfg
    echo -n "lib=${lib:-default} "; STITO_LIB_PATH=$lib python tools/fx_bench.py --chain ParametricEQ --pop $1 --seconds $2 --channels $3 2>/dev/null | tail -1
  done
  echo -n "lib=${lib:-default} bench: "; STITO_LIB_PATH=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pop512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['last_fitness_sha16'])"
done
