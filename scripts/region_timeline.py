"""Timeline of ONE hipGraph replay of the roofline region (development aid): run under rocprofv3 --kernel-trace, then
python scripts/region_timeline.py --parse <kernel_trace.csv>"""
import argparse
import csv
import os
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--parse")
ap.add_argument("--B", type=int, default=64)
a = ap.parse_args()
if a.parse:
    rows = sorted(csv.DictReader(open(a.parse)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "spin" in r["Kernel_Name"].lower()]
    rows = rows[marks[-2] + 1:marks[-1]]
    t0 = int(rows[0]["Start_Timestamp"])
    end = 0
    for r in rows:
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        wgs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>3s} {wgs:6d}  {(m.group(1) if m else r['Kernel_Name'][:40])}")
        end = max(end, e)
    print("span", round(end, 1), "us")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bist_amd.model as M
from bist_amd.data.synthetic import synthetic_batch
import bench
args = bench.model_args(1, 512, 8, 0.0)
torch.manual_seed(1)
model = M.make_model(3000, 3000, args, ft_sizes=[2048]).cuda().to(torch.bfloat16).eval()
b = synthetic_batch(a.B, dtype=torch.bfloat16, seed=1)
with torch.no_grad():
    q = model.encode_text(b, {})["encoded_query"]
    vl = model.mutlimodal_decoder.v_layers[0]

    def run():
        f = model.vid_encoder(b, {})
        vl({"t2s": q, "s2t": q}, f, b)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        run()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    torch.cuda._sleep(1000)
    g.replay()
    torch.cuda.synchronize()
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
