#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python tools/decode_rows_time.py 7 14 16 21 28 32
for R in 14 28; do
  rm -rf gpurun_out/dr_$R
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dr_$R -o dr -- python3 tools/decode_rows_time.py $R > /dev/null 2>&1
  f=$(find gpurun_out/dr_$R -name '*kernel_stats.csv' | head -1)
  echo "== rows $R"; head -8 "$f" | cut -d, -f1-4 | sed 's/(unsigned short const.*QkvRope)//' | cut -c1-150
  find gpurun_out/dr_$R -name '*trace.csv' -delete
done
