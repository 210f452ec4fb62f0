"""Which copies of the HIP / HSA / RCCL runtimes does a process map once libpenguin_hip.so AND torch are loaded, and does
it exit cleanly?  (VERDICT r02 item 3 / ADVICE: `double free or corruption` at interpreter exit.)
    python scripts/which_hip_runtime.py [order] [gpu]
order = "lib,torch" (default) or "torch,lib"; "gpu": both sides also touch the device.  The exit code of the process is
the experiment's result.  Findings (round 3): the abort needs librccl mapped BEFORE torch is imported -- any hipcc-built
shared object linked with -lrccl does it, an empty one included, with a single copy of every runtime mapped -- so the
library loads RCCL on first use (csrc/pg_rccl.h) and shares torch's HIP runtime (penguin/jl_amd/_lib.py)."""
import re
import sys

sys.path.insert(0, ".")
order = (sys.argv[1] if len(sys.argv) > 1 else "lib,torch").split(",")
use_gpu = len(sys.argv) > 2 and sys.argv[2] == "gpu"
for what in order:
    if what == "lib":
        import penguin.jl_amd as pj
        pj.lib()
        if use_gpu:
            pj.init(0)
    elif what == "torch":
        import torch
        if use_gpu:
            torch.zeros(4, device="cuda").sum().item()
pat = re.compile(r"(libamdhip64|libhsa-runtime64|librccl|libhiprtc|libroctracer|librocprofiler)[^/]*$")
seen = {}
for line in open("/proc/self/maps"):
    path = line.split()[-1] if "/" in line else ""
    if pat.search(path):
        seen[path] = seen.get(path, 0) + 1
print("order", order, "gpu" if use_gpu else "no gpu call")
for p in sorted(seen):
    print("   mapped:", p)
sys.stdout.flush()
