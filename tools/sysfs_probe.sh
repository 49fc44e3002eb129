#!/bin/bash
# what the box exposes for clock / power sampling without a GPU call (bench.py ClockSampler)
for d in /sys/class/drm/card*/device; do
  echo "== $d"; cat $d/pp_dpm_sclk 2>&1 | head -12; ls $d/hwmon 2>&1
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input freq1_input; do [ -r $h/$f ] && echo "$f=$(cat $h/$f)"; done; done
done
timeout 20 rocm-smi --showclocks --showpower --json 2>&1 | head -c 1500
