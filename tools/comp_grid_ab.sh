# compressor streaming passes: one tile per workgroup (STITO_COMP_WGS_PER_CU=0) against ~32 / 16 / 64 workgroups per CU walking tiles
cd $GRAFT_REPO_ROOT
for w in 0 16 32 64; do
  echo -n "STITO_COMP_WGS_PER_CU=$w: "; STITO_COMP_WGS_PER_CU=$w python tools/fx_bench.py --chain Compressor --pop 256 --seconds 10 --channels 2 --reps 7 2>/dev/null | tail -1
done
for w in 0 32; do
  echo -n "STITO_COMP_WGS_PER_CU=$w bench: "; STITO_COMP_WGS_PER_CU=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pop512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['last_fitness_sha16'])"
done
echo -n "pre-persistence library bench: "; STITO_LIB_PATH=st-ito_amd/st_ito/_lib/ab/libstito_hip_prepersist.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pop512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['last_fitness_sha16'])"
