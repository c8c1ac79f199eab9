#!/bin/bash
# Which role sets k_reverb's tile time?  Builds libstito_hip variants with one role switched off (RV_ABL) into tools/ab/ (they travel
# with gpurun; *.so is git-ignored) and times the Reverb-only chain with each:   bash tools/reverb_ablate.sh build   (here)
#                                                                                bash tools/reverb_ablate.sh run     (on the GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/ab
  for a in 0 1 2 3; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DRV_ABL=$a -c st-ito_amd/csrc/dsp.hip -o tools/ab/dsp_$a.o || exit 1
    objs=$(ls st-ito_amd/csrc/build/*.o | grep -v "/dsp.o")
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libstito_rvabl$a.so tools/ab/dsp_$a.o $objs || exit 1
  done
  rm -f tools/ab/*.o
else
  for a in 0 1 2 3; do
    echo -n "RV_ABL=$a (1 = no comb, 2 = no all-pass): "
    STITO_LIB_PATH=$PWD/tools/ab/libstito_rvabl$a.so python tools/fx_bench.py --chain Reverb --pop ${POP:-256} --reps 5 2>/dev/null | tail -1
  done
fi
