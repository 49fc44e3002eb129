#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for args in "--gemm-cus 0" "--gemm-cus 240" "--gemm-cus 224" "--gemm-cus 192" "--streams 3" "--streams 5" "--streams 6" "--streams 2"; do
  echo "== $args"
  python bench.py --steps 24 --warmup 4 --no-extras --no-cpu-baseline $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
