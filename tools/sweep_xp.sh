#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "--prefill-batch 4" "--prefill-batch 4 --pool-rows 56 --streams 16" "--prefill-batch 8 --pool-rows 56 --streams 16" "--prefill-batch 8" "--prefill-batch 4 --pool-rows 56 --streams 24 --pools 3" "--prefill-batch 4 --pool-rows 56 --streams 16 --steps 40 --warmup 16"; do
  echo "== $args"
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 $args > /tmp/b.out 2>&1; echo "rc=$?"; tail -1 /tmp/b.out | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config'].get('decode')[-40:], d['config'].get('prefill')[-40:])
except Exception as e: print('ERR', e); print(open('/tmp/b.out').read()[-2500:])"
done
