#!/bin/bash
# the in-flight tests (no retry any more; since the recursion-tail change of round 4 also the two DecodeServer pipelines), N cold runs each (a fresh process per run) on one lease.
#   bash tools/inflight_20x.sh [N]  ->  gpurun_out/inflight_20x.log
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
N=${1:-20}
LOG=gpurun_out/inflight_20x.log
mkdir -p gpurun_out; : > $LOG
fail=0
for i in $(seq 1 $N); do
  for t in "tests/test_gpu_model.py::test_recursions_in_flight_on_two_streams_match_sequential" "tests/test_gpu_configs.py::test_eos_generates_interleave_on_two_streams" \
           "tests/test_gpu_merged_decode.py::test_recursions_through_the_decode_server_equal_sequential" "tests/test_gpu_merged_decode.py::test_one_row_generates_in_flight_share_prefill_passes_and_decode_steps"; do
    out=$(timeout 600 python3 -m pytest "$t" -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1)
    echo "run $i $t : $out" >> $LOG
    case "$out" in *passed*) ;; *) fail=$((fail+1));; esac
  done
done
echo "cold runs: $N per test, failures: $fail" >> $LOG
tail -3 $LOG
