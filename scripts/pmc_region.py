"""HBM bytes of one pass over the roofline region (scripts/prof_attn.py) from two rocprofv3 counter passes, per the HBM section of
/opt/skills/guides/MI355X_MICROARCH.md: separate --pmc passes, FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled on gfx950 (it
reports half the bytes of 16-byte-per-lane streaming reads -- what every kernel of this region issues).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_pmc_fetch -- python3 scripts/prof_attn.py --B 64 --iters 5
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_pmc_write -- python3 scripts/prof_attn.py --B 64 --iters 5
  python scripts/pmc_region.py <fetch counter_collection.csv> <write counter_collection.csv> 64 32 5 > profiles/r02_attn_fwd_B64_pmc.json
"""
import collections
import csv
import json
import re
import sys


def region(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "spin" in r["Kernel_Name"].lower()]
    assert len(marks) >= 2, f"{path}: the two spin-kernel markers are missing ({len(marks)})"
    return rows[marks[-2] + 1:marks[-1]]


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:48]


B, T, iters = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
S, C, d, Lq = 49, 2048, 512, 20
fetch, write = region(sys.argv[1], "FETCH_SIZE"), region(sys.argv[2], "WRITE_SIZE")
per = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in fetch:
    per[short(r["Kernel_Name"])][0] += float(r["Counter_Value"]) * 1024 * 2 / iters
    per[short(r["Kernel_Name"])][2] += 1
for r in write:
    per[short(r["Kernel_Name"])][1] += float(r["Counter_Value"]) * 1024 / iters
fb, wb = sum(v[0] for v in per.values()), sum(v[1] for v in per.values())
rows_v = B * T * S
# SURVEY.md 8(d) "Algorithmic bytes": features read once, projected video tensor written once and read once per direction-fused
# pass (lower bound) or once per direction (this build: two stage-1 launches), weights, stage-1 outputs written and read once
alg_lo = 2 * (rows_v * C + rows_v * d + rows_v * d + (C * d + 40 * d * d) + 2 * B * (S + T) * Lq * d)
alg_hi = alg_lo + 2 * rows_v * d
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # (region_sources_sha: the key bench.py checks before quoting these bytes)
print(json.dumps({
    "B": B, "T": T, "passes_averaged": iters, "region_sources_sha256": bench.region_sources_sha(),
    "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_bytes": fb + wb,
    "algorithmic_bytes_range": [alg_lo, alg_hi], "ratio_to_algorithmic": [(fb + wb) / alg_hi, (fb + wb) / alg_lo],
    "per_kernel_bytes": {k: {"fetch": round(v[0]), "write": round(v[1]), "launches_per_pass": v[2] / iters} for k, v in
                         sorted(per.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))},
    "method": "two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE, each with --kernel-trace only) over scripts/prof_attn.py; dispatches "
              "between its two spin-kernel markers; FETCH_SIZE x 2 (gfx950), KiB units; scripts/pmc_region.py",
}, indent=1))
