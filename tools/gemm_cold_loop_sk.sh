#!/bin/bash
# N cold runs (fresh process each, both flavours) of tools/gemm_cold.py on stream-K shapes of the k-split ping-pong loop; one distinct line per (flavour, shape) expected.
# usage: gemm_cold_loop_sk.sh N out.log
N=${1:-40}; OUT=${2:-gpurun_out/gemm_cold_sk.log}
: > $OUT
for i in $(seq 1 $N); do
  for sh in 1005,22016,2048 4020,22016,4096 8040,4096,4096; do
    for f in bf16 f16; do SHAPE=$sh timeout 120 python tools/gemm_cold.py $f 2>/dev/null | sed "s/^/$sh /" >> $OUT || echo "$sh $f RUN $i FAILED" >> $OUT; done
  done
done
echo "runs per flavour and shape: $N" >> $OUT
echo "distinct result lines:" >> $OUT
grep -v "^runs\|^distinct" $OUT | sort | uniq -c >> $OUT
tail -n 8 $OUT
