"""Which Python lines of the bench pipeline make the host WAIT for the device (``.item()`` / ``.cpu()`` / ``.tolist()`` / pageable copies):
runs bench.py under ``torch.cuda.set_sync_debug_mode("warn")`` and counts the warnings per source line.

    python3 tools/sync_audit.py [bench.py arguments]   ->  gpurun_out/r4_sync_audit.txt
"""
import collections
import os
import runpy
import sys
import warnings

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seen = collections.Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        seen[f"{os.path.relpath(filename, root)}:{lineno}"] += 1


warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
sys.argv = ["bench.py"] + (sys.argv[1:] or ["--steps", "20", "--warmup", "20", "--no-cpu-baseline", "--no-extras"])
try:
    runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
finally:
    torch.cuda.set_sync_debug_mode("default")
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "r4_sync_audit.txt"), "w") as f:
        for k, v in seen.most_common():
            f.write(f"{v:7d}  {k}\n")
    sys.stderr.write("".join(f"{v:7d}  {k}\n" for k, v in seen.most_common(40)))
