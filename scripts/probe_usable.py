import os, sys
os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["BIST_SPLIT_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bist_amd import graphsplit as GS
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode in ("streams", "streams_nonnull"):
    keep = [torch.cuda.Stream() for _ in range(int(os.environ.get("NPRE", "9")))]     # what a trainer process has created before
    x = torch.zeros(1 << 20, device="cuda")
    for s in keep:
        with torch.cuda.stream(s):
            x.add_(1)
    torch.cuda.synchronize()
if mode == "streams_nonnull":
    with torch.cuda.stream(torch.cuda.Stream()):
        print(mode, "usable:", GS.usable(), "|", GS.WHY_NOT)
else:
    print(mode, "usable:", GS.usable(), "|", GS.WHY_NOT)
