#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "--streams 14 --pools 2 --pool-rows 49" "--streams 14 --pools 2 --pool-rows 49" "--streams 14 --pools 2 --pool-rows 49" "--streams 18 --pools 2 --pool-rows 64" "--streams 18 --pools 2 --pool-rows 64"; do
  echo "== $args"
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 $args > /tmp/b.out 2>&1; echo "rc=$?"; tail -1 /tmp/b.out | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config'].get('decode')[-40:])
except Exception as e: print('ERR', e); print(open('/tmp/b.out').read()[-1500:])"
done
