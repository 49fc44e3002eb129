# Round-6 profiles (run on the GPU box from the repo root: gpurun -- 'bash tools/profile_r6.sh'): kernel stats, HBM traffic (FETCH / WRITE in
# separate passes), SQ / GRBM counters of the MFMA stage, isolated merged decode steps (bf16 and FP8 weights), clocks / power next to the
# GEMM and the recursion (the default build: fp16 operands).  Every rocprofv3 call sits under `timeout` and gets the program itself after `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
B="bench.py --steps 16 --warmup 0 --settle 0 --no-cpu-baseline --no-extras"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_stats -o bench -- python3 bench.py --steps 32 --warmup 16 --settle 0 --no-cpu-baseline --no-extras > gpurun_out/r6_stats.log 2>&1
find gpurun_out/r6_stats -name '*trace.csv' -delete
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r6_fetch -o bench -- python3 $B > gpurun_out/r6_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r6_write -o bench -- python3 $B > gpurun_out/r6_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r6_sq -o bench -- python3 $B > gpurun_out/r6_sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/r6_grbm -o bench -- python3 $B > gpurun_out/r6_grbm.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r6_stats gpurun_out/r6_fetch gpurun_out/r6_write gpurun_out/r6 gpurun_out/r6_sq gpurun_out/r6_grbm
for d in r6_fetch r6_write r6_sq r6_grbm; do find gpurun_out/$d -name '*.csv' -size +4M -delete; done
# isolated merged decode steps
( cd /tmp; bash $GRAFT_REPO_ROOT/tools/decode_rows_prof.sh 14 28 56 70 112 140; bash $GRAFT_REPO_ROOT/tools/decode_rows_prof.sh 70 112 140 fp8 ) > gpurun_out/r6_decode_prof.log 2>&1
python3 tools/decode_rows_time.py 7 14 28 56 70 112 140 > gpurun_out/r6_decode_ms.log 2>&1
python3 tools/decode_rows_time.py 56 70 112 140 --fp8 >> gpurun_out/r6_decode_ms.log 2>&1
python3 tools/decode_rows_summary.py 14 28 56 70 112 140 70f8 112f8 140f8
# one batched prefill pass at the headline's row count (8 x 1005 rows since the bench batches up to 8 prefills), per-kernel averages -> gpurun_out/r6_prefill_pass.json
bash tools/prefill_prof.sh 8 > gpurun_out/r6_prefill_prof.log 2>&1
# the stage-2 adapter alone (one recursion's 100 windows x 256 frames, 30 runs): per-kernel times when nothing else runs
rm -rf gpurun_out/r6_adapter
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_adapter -o adp -- python3 tools/adapter_prof.py 30 > gpurun_out/r6_adapter.log 2>&1
find gpurun_out/r6_adapter -name '*trace.csv' -delete
cp $(find gpurun_out/r6_adapter -name 'adp_kernel_stats.csv' | head -1) gpurun_out/r6_adapter_kernel_stats.csv 2>/dev/null
# the vendor yardstick on the prefill GEMM shapes in the build that ships (fp16 operands) and in the bf16 build
python3 tools/blas_yardstick.py f16 > gpurun_out/r6_blas_yardstick_f16.txt 2>&1
python3 tools/blas_yardstick.py bf16 > gpurun_out/r6_blas_yardstick_bf16.txt 2>&1
# (clocks / power: bench.py samples sclk and socket power itself now - `clocks_during_the_timed_region`, `extra_measurements.sustained.clocks`)
ls -la gpurun_out/r6_*.json gpurun_out/r6_*.csv 2>/dev/null
tail -2 gpurun_out/r6_sq.log
