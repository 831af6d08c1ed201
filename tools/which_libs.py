"""Which HIP / RCCL libraries end up in a bench-like process (torch first, then the library, then
the library's RCCL look-up)?  Prints the mapped paths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
from esvio_amd import frontend as FE
ft = FE.FeatureTracker(FE.make_config(346, 260, device=0))
if "--dist" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.barrier()
FE.comm_unique_id()
seen = set()
for line in open("/proc/self/maps"):
    p = line.split()[-1]
    if ("rccl" in p or "amdhip64" in p or "hsa-runtime" in p or "esvio_fe" in p) and p not in seen:
        seen.add(p); print(p)
