# A/B of the dense-projector GEMM (25600 x 4096 x 768): timing, then FETCH_SIZE / WRITE_SIZE in separate passes.
# Every rocprofv3 call sits under `timeout`: a counter set the hardware cannot collect makes the tool abort and then hang.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in ${MODES:-2 6 0}; do
timeout 120 python3 - <<PY
import sys
sys.path.insert(0,'.')
import torch
from revisionllm_amd import hip, ops
dev=torch.device('cuda:0'); opt=hip.Options(gemm_arows=$mode)
x=torch.randn(25600,768,device=dev).to(torch.bfloat16); w=ops.pack_fragments((torch.randn(4096,768,device=dev)*0.02).to(torch.bfloat16))
bias=torch.randn(4096,device=dev); out=torch.empty(25600,4096,dtype=torch.bfloat16,device=dev)
f=lambda: ops.gemm(x,w,bias=bias,out=out,w_packed=True,stream_k=False,ctx=opt)
for _ in range(5): f()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); torch.cuda.synchronize(); a.record()
for _ in range(30): f()
b.record(); torch.cuda.synchronize(); print('mode',$mode,'us',a.elapsed_time(b)/30*1e3, flush=True)
PY
done
for mode in ${PMC_MODES:-2 0}; do
timeout 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_arows_f$mode -o a -- python3 tools/arows_only.py $mode > gpurun_out/pmc_arows_f$mode.log 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_arows_w$mode -o a -- python3 tools/arows_only.py $mode >> gpurun_out/pmc_arows_f$mode.log 2>&1
done
python3 - <<'PY'
import csv, collections, glob
for d in sorted(glob.glob('gpurun_out/pmc_arows_[fw]*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm' in r['Kernel_Name']:
                acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()):
            print(d, k[0][-40:], k[1], sum(v)/len(v))
PY
