# Round-4 profiles (run on the GPU box from the repo root: gpurun -- 'bash tools/profile_r4.sh'): kernel stats, HBM traffic (FETCH / WRITE in
# separate passes), SQ / GRBM counters of the MFMA stage, isolated merged decode steps (bf16 and FP8 weights), clocks / power next to the
# GEMM and the recursion.  Every rocprofv3 call sits under `timeout` and gets the program itself after `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
B="bench.py --steps 16 --warmup 0 --settle 0 --no-cpu-baseline --no-extras"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_stats -o bench -- python3 bench.py --steps 32 --warmup 16 --settle 0 --no-cpu-baseline --no-extras > gpurun_out/r4_stats.log 2>&1
find gpurun_out/r4_stats -name '*trace.csv' -delete
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r4_fetch -o bench -- python3 $B > gpurun_out/r4_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4_write -o bench -- python3 $B > gpurun_out/r4_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r4_sq -o bench -- python3 $B > gpurun_out/r4_sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/r4_grbm -o bench -- python3 $B > gpurun_out/r4_grbm.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r4_stats gpurun_out/r4_fetch gpurun_out/r4_write gpurun_out/r4 gpurun_out/r4_sq gpurun_out/r4_grbm
for d in r4_fetch r4_write r4_sq r4_grbm; do find gpurun_out/$d -name '*.csv' -size +4M -delete; done
# isolated merged decode steps
( cd /tmp; bash $GRAFT_REPO_ROOT/tools/decode_rows_prof.sh 14 28 56 70 112 140; bash $GRAFT_REPO_ROOT/tools/decode_rows_prof.sh 70 112 140 fp8 ) > gpurun_out/r4_decode_prof.log 2>&1
python3 tools/decode_rows_time.py 7 14 28 56 70 112 140 > gpurun_out/r4_decode_ms.log 2>&1
python3 tools/decode_rows_time.py 56 70 112 140 --fp8 >> gpurun_out/r4_decode_ms.log 2>&1
python3 tools/decode_rows_summary.py 14 28 56 70 112 140 70f8 112f8 140f8
# clocks / power: one sample per second next to (a) the 4096^3 ping-pong GEMM on random data, (b) the decode gate/up weight stream, (c) the bench
sample() { for i in $(seq 1 $2); do echo "== $1 t=$i"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" ; sleep 1; done; }
( python3 tools/clock_probe.py gemm_random 10 > gpurun_out/r4_clk_gemm.out 2>&1 & sleep 3; sample gemm 6; wait ) > gpurun_out/r4_clocks.log 2>&1
( python3 tools/clock_probe.py gemv 10 > gpurun_out/r4_clk_gemv.out 2>&1 & sleep 3; sample gemv 6; wait ) >> gpurun_out/r4_clocks.log 2>&1
( python3 bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r4_clk_bench.out 2>&1 & sleep 40; sample bench 6; wait ) >> gpurun_out/r4_clocks.log 2>&1
ls -la gpurun_out/r4_*.json gpurun_out/r4_*.csv 2>/dev/null
tail -2 gpurun_out/r4_sq.log
