"""Per-kernel PMC summary from one rocprofv3 counter-collection CSV (developer tool).

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES \
            SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d DIR -o a -- python bench.py --steps 2 --warmup 1 ...
  python tools/pmc_summary.py DIR/a_counter_collection.csv out.json mhsa relpos logmel absmax layernorm

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts
cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs, SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT LDS-array cycles
summed over the 256 CUs.  Derived: mfma_busy = MFMA_BUSY / (1024 x GUI_ACTIVE / 8); lds_busy = LDS_IDX_ACTIVE / (256 x GUI_ACTIVE / 8);
valu_busy = 4 x ACTIVE_INST_VALU / (1024 x GUI_ACTIVE / 8) (quad-cycles -> cycles; one VALU issue port per SIMD)."""
import collections
import csv
import json
import sys

path, out_path, pats = sys.argv[1], sys.argv[2], sys.argv[3:]
per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0]
    if pats and not any(p in k for p in pats):
        continue
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[k] += 1
out = {"source": path.split("/")[-1], "units": "see tools/pmc_summary.py", "kernels": {}}
for k in sorted(per):
    c = per[k]
    g = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    d = {"launches": n[k], **{kk: round(v) for kk, v in sorted(c.items())}}
    if g:
        d["mfma_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * g), 4)
        d["lds_busy"] = round(c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256 * g), 4)
        d["valu_busy"] = round(4 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024 * g), 4)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("SQ_WAVE_CYCLES"):
        d["waves_parked_share"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        d["waves_issue_stalled_share"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
    out["kernels"][k] = d
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
