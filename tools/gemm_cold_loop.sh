#!/bin/bash
# N cold runs (fresh process each) of tools/gemm_cold.py; the log's distinct lines must be ONE per flavour.  usage: gemm_cold_loop.sh N out.log
N=${1:-100}; OUT=${2:-gpurun_out/gemm_cold.log}
: > $OUT
for i in $(seq 1 $N); do
  for f in bf16 f16; do timeout 120 python tools/gemm_cold.py $f >> $OUT 2>/dev/null || echo "$f RUN $i FAILED rc=$?" >> $OUT; done
done
echo "runs per flavour: $N" >> $OUT
echo "distinct result lines:" >> $OUT
grep -v "^runs\|^distinct" $OUT | sort | uniq -c >> $OUT
tail -n 6 $OUT
