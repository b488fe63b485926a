"""Mean counter values per launch of the P0 GEMM (gemm_big_kernel on 196 workgroups, the largest-FETCH launches are told apart by
duration in the kernel trace of the same run) from rocprofv3 --pmc CSVs: python scripts/pmc_kernel.py <counter_collection.csv> ..."""
import csv, collections, sys
GRID = 196 * 512
for path in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "gemm_big_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == GRID:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if dur > 45:                      # K = 2048 (P0); the K = 512 products on this grid take ~25 us
                acc[r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
    for k, v in acc.items():
        print(f"{k}: mean {sum(x for x, _ in v) / len(v):.4g} per launch over {len(v)} P0 launches (mean {sum(d for _, d in v) / len(v):.1f} us under the counter pass)")
