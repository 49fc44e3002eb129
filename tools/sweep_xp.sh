#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "--streams 3" "--streams 4" "--streams 8 --pools 2" "--streams 6 --pools 2" "--streams 10 --pools 2" "--streams 12 --pools 3"; do
  echo "== $args"
  timeout 300 python bench.py --steps 32 --warmup 8 --no-extras --no-cpu-baseline $args 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('decode'))
except Exception as e: print('ERR', e)"
done
