#!/bin/bash
# Samples the shader clock / power while a command runs (timing experiments: is a kernel slower because the part clocks down under it?)
#   tools/clock_probe.sh <outfile> <command ...>
out=$1; shift
( while true; do
    for f in /sys/class/drm/card*/device/pp_dpm_sclk; do grep '\*' $f 2>/dev/null | tr '\n' ' '; done
    for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_input; do [ -r $f ] && echo -n " P=$(cat $f)"; done
    for f in /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input; do [ -r $f ] && echo -n " F=$(cat $f)"; done
    echo
    sleep 0.05
  done ) > $out 2>&1 &
pid=$!
"$@"
kill $pid
