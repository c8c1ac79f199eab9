# steady-state step time of small populations (bench chain, 262 144 samples, 40 steps) + the default bench line's headline
cd $GRAFT_REPO_ROOT
for pop in 32 64 128; do
  echo -n "pop $pop, 262144 samples, bench chain: cand/s, ms/step, host ms (ask + launch, sync, tell), fitness hash: "
  python bench.py --pop-per-gpu $pop --seconds 5.4613 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-pop512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stages']['evaluate_ms']['max'], d['stages']['gather_ms']['max'], d['stages']['tell_ms']['max'], d['last_fitness_sha16'])"
done
