# Round-2 profiles (run on the GPU box from the repo root): kernel stats, HBM traffic (FETCH / WRITE in separate passes) and
# the SQ counters of the MFMA stage.  Every rocprofv3 call sits under `timeout` (a counter set the hardware cannot collect makes
# the tool abort and then hang) and gets the program itself after `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# default pipeline (16 steps in flight, 56-row pools, up to 4 prefills per pass): 16 steps = 4 full prefill passes of 4020 rows, 2 gangs of 8
B="bench.py --steps 16 --warmup 0 --settle 0 --no-cpu-baseline --no-extras"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_stats -o bench -- python3 bench.py --steps 32 --warmup 16 --settle 0 --no-cpu-baseline --no-extras > gpurun_out/r2_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_fetch -o bench -- python3 $B > gpurun_out/r2_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2_write -o bench -- python3 $B > gpurun_out/r2_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r2_sq -o bench -- python3 $B > gpurun_out/r2_sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/r2_grbm -o bench -- python3 $B > gpurun_out/r2_grbm.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r2_stats gpurun_out/r2_fetch gpurun_out/r2_write gpurun_out/r2 gpurun_out/r2_sq gpurun_out/r2_grbm
ls -la gpurun_out/r2_*.json gpurun_out/r2_*.csv 2>/dev/null
tail -2 gpurun_out/r2_sq.log
