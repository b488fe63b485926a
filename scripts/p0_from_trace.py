"""Durations of the P0 GEMM launches in a rocprofv3 kernel trace (the `--stats` average of gemm_big_kernel mixes P0 with the
K = 512 products and the B = 64 launches of the attn_fwd section): python scripts/p0_from_trace.py <kernel_trace.csv>"""
import csv, statistics, sys

GRID = 196 * 512          # 196 tiles of 256x256 for M = 25088, N = 512
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[1]))
     if "gemm_big_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == GRID]
p0 = [x for x in d if x > 45]          # K = 2048 (P0); the K = 512 products on the same grid take ~25 us
print(f"{len(d)} gemm_big_kernel launches on 196 workgroups; {len(p0)} are P0 [25088x2048]x[2048x512]: mean {statistics.mean(p0):.1f} us, "
      f"min {min(p0):.1f}, max {max(p0):.1f}; first 12 (training passes): mean {statistics.mean(p0[:12]):.1f} us; the rest are the eval loops of attn_fwd")
