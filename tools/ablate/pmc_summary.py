"""Per-kernel averages of the rocprofv3 --pmc counter CSVs written by tools/ablate/rpx_pmc.sh (developer tool)."""
import csv, collections, sys
filt = sys.argv[1] if len(sys.argv) > 1 else "relpos"
for d in sys.argv[2:] or ("gpurun_out/r2/rpx_pmc1", "gpurun_out/r2/rpx_pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"{d}/p_counter_collection.csv")):
        n = r["Kernel_Name"]
        if filt not in n:
            continue
        acc[n[:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, c in acc.items():
        print(n)
        for k, v in sorted(c.items()):
            print("   %-28s %14.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
