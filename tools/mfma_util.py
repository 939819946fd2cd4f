"""MFMA-pipe utilisation of the GEMM family from one rocprofv3 PMC pass (developer tool).

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -o m -- python bench.py --steps 2 --warmup 1 \
            --no-cpu-baseline --no-kernel-timer
  python tools/mfma_util.py DIR/m_counter_collection.csv profiles/r1_gemm_mfma_busy.json

SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs (32 cycles per v_mfma_f32_32x32x16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed
over the 8 XCDs.  busy fraction = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8)."""
import csv, json, sys, collections

per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if not ("gemm_nt" in k or "gemm_tn" in k):
        continue
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
out = {"formula": "busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)", "kernels": {}}
tb = tg = 0.0
for k in sorted(per):
    b, g = per[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), per[k].get("GRBM_GUI_ACTIVE", 0.0)
    out["kernels"][k] = {"launches": cnt[k], "mfma_busy_cycles": round(b), "gui_active": round(g), "busy_fraction": round(b / (128.0 * g), 4) if g else None}
    tb += b
    tg += g
out["family_busy_fraction"] = round(tb / (128.0 * tg), 4) if tg else None
out["commit"] = sys.argv[3] if len(sys.argv) > 3 else None      # the tree the counters were collected on
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
