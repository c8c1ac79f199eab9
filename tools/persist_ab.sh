# persistent k_conv_wino43s (this build) against the PRE-PERSISTENCE library (tools/build_rev_lib.sh 8fdea21 prepersist) and against this
# library with one workgroup per item: per-layer times (conv_bench, production mix, random data) and the bench line, same box
cd $GRAFT_REPO_ROOT
run() {
  echo "== $1"
  python tools/conv_bench.py --streams 512 --modes 100 --reps 7 2>/dev/null | grep -E "234x64 128|117x32|58x16 256|total" | tr '\n' ' '; echo
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pop512 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['last_fitness_sha16'])"
}
STITO_LIB_PATH=st-ito_amd/st_ito/_lib/ab/libstito_hip_prepersist.so run "pre-persistence library (8fdea21)"
STITO_W43S_PERSIST=0 run "this library, one workgroup per item"
run "this library, persistent (default)"
STITO_LIB_PATH=st-ito_amd/st_ito/_lib/ab/libstito_hip_prepersist.so run "pre-persistence library (8fdea21), again"
