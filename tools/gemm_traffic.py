"""HBM traffic of the GEMM family from rocprofv3 PMC passes (developer tool).

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR_F -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d DIR_W -o w -- python bench.py ...   (separate pass: TCC has 4 slots)
  python tools/gemm_traffic.py DIR_F/f_counter_collection.csv DIR_W/w_counter_collection.csv profiles/r1_gemm_traffic.json

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
128-byte read requests as 64 bytes, so reads are doubled.  Averages are per launch over every gemm_nt* / gemm_tn* kernel launch."""
import csv, json, sys, collections


def load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not ("gemm_nt" in r["Kernel_Name"] or "gemm_tn" in r["Kernel_Name"]):
            continue
        k = r["Kernel_Name"].split("(")[0]
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    return per


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"unit": "bytes per launch", "correction": "reads = 2 x FETCH_SIZE KiB (gfx950), writes = WRITE_SIZE KiB", "kernels": {}}
tot_l = tot_b = 0
for k in sorted(set(f) | set(w)):
    n = max(f[k][0], w[k][0])
    rd = 2.0 * 1024 * f[k][1] / max(1, f[k][0])
    wr = 1024.0 * w[k][1] / max(1, w[k][0])
    out["kernels"][k] = {"launches": n, "read": round(rd), "write": round(wr)}
    tot_l += n
    tot_b += n * (rd + wr)
out["launches"] = tot_l
out["avg_bytes_per_launch"] = round(tot_b / max(1, tot_l))
out["commit"] = sys.argv[4] if len(sys.argv) > 4 else None      # the tree the counters were collected on
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
