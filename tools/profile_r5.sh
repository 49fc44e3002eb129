# Round-5 profiles (run on the GPU box from the repo root: gpurun -- 'bash tools/profile_r5.sh'): kernel stats, HBM traffic (FETCH / WRITE in
# separate passes), SQ / GRBM counters of the MFMA stage, isolated merged decode steps (bf16 and FP8 weights), clocks / power next to the
# GEMM and the recursion (the default build: fp16 operands).  Every rocprofv3 call sits under `timeout` and gets the program itself after `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
B="bench.py --steps 16 --warmup 0 --settle 0 --no-cpu-baseline --no-extras"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_stats -o bench -- python3 bench.py --steps 32 --warmup 16 --settle 0 --no-cpu-baseline --no-extras > gpurun_out/r5_stats.log 2>&1
find gpurun_out/r5_stats -name '*trace.csv' -delete
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r5_fetch -o bench -- python3 $B > gpurun_out/r5_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r5_write -o bench -- python3 $B > gpurun_out/r5_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r5_sq -o bench -- python3 $B > gpurun_out/r5_sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/r5_grbm -o bench -- python3 $B > gpurun_out/r5_grbm.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r5_stats gpurun_out/r5_fetch gpurun_out/r5_write gpurun_out/r5 gpurun_out/r5_sq gpurun_out/r5_grbm
for d in r5_fetch r5_write r5_sq r5_grbm; do find gpurun_out/$d -name '*.csv' -size +4M -delete; done
# isolated merged decode steps
( cd /tmp; bash $GRAFT_REPO_ROOT/tools/decode_rows_prof.sh 14 28 56 70 112 140; bash $GRAFT_REPO_ROOT/tools/decode_rows_prof.sh 70 112 140 fp8 ) > gpurun_out/r5_decode_prof.log 2>&1
python3 tools/decode_rows_time.py 7 14 28 56 70 112 140 > gpurun_out/r5_decode_ms.log 2>&1
python3 tools/decode_rows_time.py 56 70 112 140 --fp8 >> gpurun_out/r5_decode_ms.log 2>&1
python3 tools/decode_rows_summary.py 14 28 56 70 112 140 70f8 112f8 140f8
# one batched prefill pass at the headline's row count (4 x 1005 rows), per-kernel averages -> gpurun_out/r5_prefill_pass.json
bash tools/prefill_prof.sh 4 > gpurun_out/r5_prefill_prof.log 2>&1
# the stage-2 adapter alone (one recursion's 100 windows x 256 frames, 30 runs): per-kernel times when nothing else runs
rm -rf gpurun_out/r5_adapter
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_adapter -o adp -- python3 tools/adapter_prof.py 30 > gpurun_out/r5_adapter.log 2>&1
find gpurun_out/r5_adapter -name '*trace.csv' -delete
cp $(find gpurun_out/r5_adapter -name 'adp_kernel_stats.csv' | head -1) gpurun_out/r5_adapter_kernel_stats.csv 2>/dev/null
# (clocks / power: bench.py samples sclk and socket power itself now - `clocks_during_the_timed_region`, `extra_measurements.sustained.clocks`)
ls -la gpurun_out/r5_*.json gpurun_out/r5_*.csv 2>/dev/null
tail -2 gpurun_out/r5_sq.log
