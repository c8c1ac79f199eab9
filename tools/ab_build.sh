#!/bin/bash
# A/B builds of libstito_hip.so for timing experiments: tools/ab_build.sh <tag> <extra hipcc flags for conv_wino43.hip, conv_wino23r.hip, cnn14.hip, frontend.hip and dsp.hip>
# -> st-ito_amd/st_ito/_lib/ab/libstito_hip_<tag>.so (use with STITO_LIB_PATH; the other objects come from the regular build)
set -e
cd "$(dirname "$0")/../st-ito_amd/csrc"
make -j4 > /dev/null
tag=$1; shift
mkdir -p build/ab ../st_ito/_lib/ab
for f in conv_wino43 conv_wino23r cnn14 frontend dsp; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $f.hip -o build/ab/${f}_$tag.o &
done
wait
objs=$(ls build/*.o | grep -v -e conv_wino43.o -e conv_wino23r.o -e cnn14.o -e frontend.o -e /dsp.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../st_ito/_lib/ab/libstito_hip_$tag.so $objs build/ab/conv_wino43_$tag.o build/ab/conv_wino23r_$tag.o build/ab/cnn14_$tag.o build/ab/frontend_$tag.o build/ab/dsp_$tag.o
echo built ../st_ito/_lib/ab/libstito_hip_$tag.so
