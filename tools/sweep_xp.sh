#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "--gemm-cus 224" "--gemm-cus 192" "--gemm-cus 160" "--gemm-cus 240"; do
  echo "== $args"
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 $args > /tmp/b.out 2>&1; echo "rc=$?"; tail -1 /tmp/b.out | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'])
except Exception as e: print('ERR', e); print(open('/tmp/b.out').read()[-2500:])"
done
