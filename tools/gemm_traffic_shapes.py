"""Fabric-side bytes of every GEMM launch of one train step, PER SHAPE (developer tool; round-6 review item 2: "break the traffic down
per shape, not per template instance").

  run   (under a counters-only rocprofv3 pass, once per counter group):
        SED_OVERLAP_TEACHER=0 SED_DW_STREAM=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d D_F -o f -- python tools/gemm_traffic_shapes.py run calls.json
        ... --pmc WRITE_SIZE ... -d D_W -o w ...          ... --pmc TCC_HIT_sum TCC_MISS_sum ... -d D_H -o h ...
        One instrumented step between two marker launches (colsum_f32_kernel, which no MAT-SED step uses); the ordered list of GEMM calls
        (entry point, M, N, K, epilogue, algorithmic bytes) goes to calls.json.
  parse python tools/gemm_traffic_shapes.py parse calls.json D_F/f_counter_collection.csv D_W/w_counter_collection.csv [D_H/h_counter_collection.csv] out.json
        aligns the i-th gemm_* dispatch between the markers with the i-th recorded call (single stream: dispatch order = call order).

Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE in KiB, gfx950 counts 128-byte reads as 64 -> reads doubled.  These are
L2 <-> fabric bytes: a line re-read from the Infinity Cache counts like one read from HBM."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(path, mode="finetune2"):
    import torch
    import bench
    from transformer4sed_amd import ops, synth
    dev = torch.device("cuda", 0)
    B = 32
    net, ema_net, opt, trainer, sd = bench.build(B, 12, dev, mode)
    sn = wn = (B * 4 + 11) // 12; un = B - sn - wn
    trainer.cfg = json.loads(json.dumps(bench.MODE_CFG[mode]))
    trainer.cfg["training"]["batch_size"] = [sn, 0, wn, un]
    wav = torch.from_numpy(synth.synth_wav(B, seed=1000)).to(dev)
    labels = torch.from_numpy(synth.synth_batch_labels(sn, wn, un, seed=1000)).to(dev)
    step = lambda: trainer.finetune_step(wav, labels.clone())
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    mark_in, mark_out = torch.ones(4, 64, device=dev), torch.zeros(64, device=dev)
    timer = ops.KernelTimer(bench.GEMM_KERNELS)
    ops.call("sed_colsum_f32", mark_in, mark_out, 4, 64, 64)
    ops.TIMER = timer
    step()
    ops.TIMER = None
    ops.call("sed_colsum_f32", mark_in, mark_out, 4, 64, 64)
    torch.cuda.synchronize()
    calls = [dict(name=name, key=list(key), alg_bytes=by, flops=fl, ms=e0.elapsed_time(e1)) for name, e0, e1, fl, by, fi, key in timer.records]
    json.dump(calls, open(path, "w"))
    print(f"{len(calls)} GEMM calls recorded")


def dispatches(path, counters):
    rows = collections.defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in counters:
            d = int(r["Dispatch_Id"])
            rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            names[d] = r["Kernel_Name"]
    order = sorted(rows)
    marks = [i for i, d in enumerate(order) if "colsum_f32" in names[d]]
    assert len(marks) == 2, f"expected two marker launches, found {len(marks)}"
    seg = [d for d in order[marks[0] + 1:marks[1]] if "gemm_nt" in names[d] or "gemm_tn" in names[d]]
    return [(names[d].split("(")[0], rows[d]) for d in seg]


def parse(calls_path, f_path, w_path, h_path, out_path):
    calls = json.load(open(calls_path))
    F, W = dispatches(f_path, ("FETCH_SIZE",)), dispatches(w_path, ("WRITE_SIZE",))
    H = dispatches(h_path, ("TCC_HIT_sum", "TCC_MISS_sum")) if h_path else None
    assert len(F) == len(W) == len(calls), (len(F), len(W), len(calls))
    tab = collections.OrderedDict()
    for i, c in enumerate(calls):
        k = (c["name"],) + tuple(c["key"])
        t = tab.setdefault(k, dict(n=0, alg=0.0, rd=0.0, wr=0.0, ms=0.0, hit=0.0, miss=0.0, kernel=F[i][0]))
        t["n"] += 1; t["alg"] += c["alg_bytes"]; t["ms"] += c["ms"]
        t["rd"] += 2.0 * 1024 * F[i][1]["FETCH_SIZE"]; t["wr"] += 1024.0 * W[i][1]["WRITE_SIZE"]
        if H:
            t["hit"] += H[i][1].get("TCC_HIT_sum", 0.0); t["miss"] += H[i][1].get("TCC_MISS_sum", 0.0)
    rows = []
    for k, t in tab.items():
        n = t["n"]
        rows.append(dict(entry=k[0], M=k[1], N=k[2], K=k[3], epi=k[4], launches=n, kernel=t["kernel"], alg_MB=round(t["alg"] / n / 1e6, 1),
                         read_MB=round(t["rd"] / n / 1e6, 1), write_MB=round(t["wr"] / n / 1e6, 1), ratio=round((t["rd"] + t["wr"]) / t["alg"], 3),
                         ms=round(t["ms"] / n, 4), GBps_fabric=round((t["rd"] + t["wr"]) / n / (t["ms"] / n * 1e-3) / 1e9, 1) if t["ms"] else None,
                         l2_hit_rate=round(t["hit"] / (t["hit"] + t["miss"]), 4) if (H and t["hit"] + t["miss"] > 0) else None,
                         excess_MB_per_step=round((t["rd"] + t["wr"] - t["alg"]) / 1e6, 1)))
    rows.sort(key=lambda r: -r["excess_MB_per_step"])
    tot_alg = sum(t["alg"] for t in tab.values()); tot = sum(t["rd"] + t["wr"] for t in tab.values())
    out = dict(unit="per launch; reads = 2 x FETCH_SIZE KiB (gfx950), writes = WRITE_SIZE KiB; L2 <-> fabric bytes", launches=len(calls),
               avg_bytes_per_launch=round(tot / len(calls)), alg_bytes_per_launch=round(tot_alg / len(calls)), ratio=round(tot / tot_alg, 3),
               commit=open(".gpurun_head").read().strip() if os.path.exists(".gpurun_head") else None, shapes=rows)
    json.dump(out, open(out_path, "w"), indent=1)
    print(f"{len(calls)} launches: fabric {tot / len(calls) / 1e6:.1f} MB vs algorithmic {tot_alg / len(calls) / 1e6:.1f} MB per launch = {tot / tot_alg:.3f}x")
    for r in rows[:40]:
        print("%-18s M=%7d N=%5d K=%5d %-8s n=%3d  alg %7.1f  read %7.1f  write %7.1f MB  x%5.2f  excess/step %8.1f MB  %7.4f ms  L2 hit %s" % (
            r["entry"], r["M"], r["N"], r["K"], r["epi"], r["launches"], r["alg_MB"], r["read_MB"], r["write_MB"], r["ratio"], r["excess_MB_per_step"],
            r["ms"], r["l2_hit_rate"]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        a = sys.argv[2:]
        parse(a[0], a[1], a[2], a[3] if len(a) > 4 else None, a[-1])
