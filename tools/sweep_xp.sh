#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "" "" "--eos" "--merge-decode 0 --streams 3" "--pools 1 --streams 4 --pool-rows 32 --prefill-batch 1" "--gpus 1 --steps 40 --warmup 16"; do
  echo "== $args"
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 $args > /tmp/b.out 2>&1; echo "rc=$?"; tail -1 /tmp/b.out | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config'].get('decode')[-40:], d['config'].get('prefill')[-60:])
except Exception as e: print('ERR', e); print(open('/tmp/b.out').read()[-2500:])"
done
