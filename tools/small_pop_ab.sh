cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sweep_split or six_sweeps or compressor or small_batches" 2>&1 | tail -5
for pop in 32 64 128; do
for mode in "0 8" "0 x" "1 x" "a x"; do
  set -- $mode
  export STITO_W43S2_SWSPLIT=$1; [ "$1" = "a" ] && unset STITO_W43S2_SWSPLIT
  export STITO_W43S2_XM=$2; [ "$2" = "x" ] && unset STITO_W43S2_XM
  echo -n "pop $pop, 262144 samples, bench chain, SWSPLIT=$1 XM=$2: cand/s, ms/step: "
  timeout 300 python bench.py --pop-per-gpu $pop --seconds 5.4613 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-pop512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['last_fitness_sha16'])"
done; done
unset STITO_W43S2_SWSPLIT STITO_W43S2_XM
timeout 600 python tools/run_configs.py --only 7 2>&1 | tail -2
timeout 300 python tools/soak_case.py --seed 0 --case 28 2>&1 | tail -3
