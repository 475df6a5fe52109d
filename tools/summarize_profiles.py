"""Condense rocprofv3 outputs under gpurun_out/ into the small, committed summaries under profiles/.
usage: python tools/summarize_profiles.py <round tag, e.g. r02> <sub-directory of gpurun_out written by tools/measure_round.sh>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
SRC = sys.argv[2] if len(sys.argv) > 2 else "final2"  # sub-directory of gpurun_out holding the measurement pass


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:80]


def kernel_stats(src, dst):
    """rocprofv3's --stats table, one row per (kernel, launch shape): a bench run launches its kernels at the timed batch AND - in the
    self-check of bench.py - at batch 1; a mean over both is no launch time.  With the kernel trace next to the stats file the rows are
    rebuilt from it with the grid as part of the key (column Grid); without it the stats table is copied as is (Grid empty)."""
    trace = os.path.join(os.path.dirname(src), os.path.basename(src).replace("kernel_stats", "kernel_trace"))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Grid", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        if os.path.exists(trace):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(trace)):
                grid = "x".join(str(int(r["Grid_Size_" + a]) // max(1, int(r["Workgroup_Size_" + a]))) for a in "XYZ")
                acc[(short(r["Kernel_Name"]), grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            total = float(sum(sum(v) for v in acc.values())) or 1.0
            for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([name, grid, len(v), sum(v), "%.1f" % (sum(v) / len(v)), "%.4g" % (100.0 * sum(v) / total), min(v), max(v)])
        else:
            for r in csv.DictReader(open(src)):
                w.writerow([short(r["Name"]), "", r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc(dirs, dst):
    out = []
    for label, d in dirs:
        for path in glob.glob(os.path.join(G, d, "*counter_collection.csv")):
            acc = collections.defaultdict(list)
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(path)):
                if "shasta" not in r["Kernel_Name"]:
                    continue
                # one row per (kernel, launch shape, counter): the self-check of bench.py launches the same kernels at batch 1
                k = (short(r["Kernel_Name"]), str(r.get("Grid_Size", "")), r["Counter_Name"])
                acc[k].append(float(r["Counter_Value"]))
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for (k, g, c), v in sorted(acc.items()):
                out.append([label, k, g, c, len(v), sum(v) / len(v), sum(dur[(k, g, c)]) / len(v)])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["run", "kernel", "grid_size", "counter", "launches", "mean_value_per_launch", "mean_duration_ns"])
        w.writerows(out)


if __name__ == "__main__":
    os.makedirs(P, exist_ok=True)
    for name, dst in (("prof_default", "bench_default_b512"), ("prof_b1024", "bench_b1024"), ("prof_b128", "bench_b128"), ("prof_b64", "bench_b64"), ("prof_b1", "bench_b1"),
                      ("prof_pieces", "bench_pieces_b512"), ("prof_conv_b8", "shared_conv_b8"), ("prof_conv_b8_heads7", "shared_conv_b8_heads7"),
                      ("prof_pair320_car", "pair_f320_car_b512"), ("prof_pair320_n500", "pair_f320_n500_b256"),
                      ("prof_train_n500", "train_n500_b8"), ("prof_train_n90", "train_n90_b64"),
                      ("prof_convtrain_b8", "shared_conv_train_b8"), ("prof_voxelize", "voxelize")):
        src = os.path.join(G, SRC, name, "d_kernel_stats.csv")
        if os.path.exists(src):
            kernel_stats(src, os.path.join(P, "%s_kernel_stats_%s.csv" % (TAG, dst)))
    runs = []
    for d in sorted(os.listdir(os.path.join(G, SRC))):
        if d.startswith("pmc_") and os.path.isdir(os.path.join(G, SRC, d)):
            kind, b = d[4:].rsplit("_", 1)  # pmc_fetch_b512 -> (fetch, b512)
            runs.append(("%s_%s" % (b, kind), SRC + "/" + d))
    pmc(runs, os.path.join(P, TAG + "_pmc_summary.csv"))
    for f in sorted(os.listdir(os.path.join(G, SRC))):
        if f.startswith("bench_") and f.endswith(".json"):
            lines = [l for l in open(os.path.join(G, SRC, f)) if l.startswith("{")]
            if lines:
                json.dump(json.loads(lines[-1]), open(os.path.join(P, TAG + "_" + f), "w"), indent=1)
        if f in ("train_n500.log", "train_n90.log", "pair_mlp.log"):
            lines = [l.rstrip() for l in open(os.path.join(G, SRC, f)) if "ms/step" in l or "fwd0" in l or "adam_lowrank" in l]
            if lines:
                open(os.path.join(P, TAG + "_" + f.replace(".log", ".txt")), "w").write("\n".join(lines) + "\n")
        if f == "stage_power.log":
            lines = [l for l in open(os.path.join(G, SRC, f)) if l.startswith("{")]
            if lines:
                json.dump(json.loads(lines[-1]), open(os.path.join(P, TAG + "_stage_power.json"), "w"), indent=1)
        if f == "bench_extra.json":
            json.dump(json.load(open(os.path.join(G, SRC, f))), open(os.path.join(P, TAG + "_bench_extra.json"), "w"), indent=1)
        if f == "voxelize.log":
            shutil.copy(os.path.join(G, SRC, f), os.path.join(P, TAG + "_voxelize.txt"))
        if f in ("gather_bwd.log", "train_determinism.log", "train_soak_conv.log"):
            shutil.copy(os.path.join(G, SRC, f), os.path.join(P, TAG + "_" + f.replace(".log", ".txt")))
        if f in ("conv_check.jsonl", "pipeline.log", "pipeline_sync.log", "pair320_car.log", "pair320_n500.log", "conv_train.jsonl"):
            rows = [json.loads(l) for l in open(os.path.join(G, SRC, f)) if l.startswith("{")]
            json.dump(rows, open(os.path.join(P, TAG + "_" + f.split(".")[0] + ".json"), "w"), indent=1)
        if f.startswith("l1_check_") and f.endswith(".json"):
            rows = [json.loads(l) for l in open(os.path.join(G, SRC, f)) if l.startswith("{")]
            json.dump(rows, open(os.path.join(P, TAG + "_" + f), "w"), indent=1)
    # HBM bytes per launch of the two heaviest kernels (bench.py: roofline.traffic), FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
    rows = list(csv.DictReader(open(os.path.join(P, TAG + "_pmc_summary.csv"))))
    traffic = {}
    for b in (1, 32, 64, 128, 512, 1024):
        for key, kern in (("batch_%d", "anchor_l1"), ("pair_batch_%d", "::pair_")):
            f = [float(r["mean_value_per_launch"]) for r in rows if r["run"] == "b%d_fetch" % b and kern in r["kernel"] and r["counter"] == "FETCH_SIZE"
                 and int(r["launches"]) > 3]
            w = [float(r["mean_value_per_launch"]) for r in rows if r["run"] == "b%d_write" % b and kern in r["kernel"] and r["counter"] == "WRITE_SIZE"
                 and int(r["launches"]) > 3]
            if f and w:
                traffic[key % b] = int((2 * f[0] + w[0]) * 1024)
    kern_names = {}
    for b in (1, 32, 64, 128, 512, 1024):
        for key, kern in (("batch_%d", "anchor_l1"), ("pair_batch_%d", "::pair_")):
            names = sorted({r["kernel"] for r in rows if r["run"] == "b%d_fetch" % b and kern in r["kernel"] and int(r["launches"]) > 3})
            if names and (key % b) in traffic:
                kern_names[key % b] = names[0].replace("shasta::", "").replace("void ", "")
    # pipe counters of the same two kernels (bench.py: roofline.mfma_busy / valu_busy / valu_insts), full-batch launches only
    counters = {}
    for b in (512, 1024):
        for key, kern in (("batch_%d", "anchor_l1"), ("pair_batch_%d", "::pair_")):
            best = {}
            for r in rows:
                if r["run"] == "b%d_mfma" % b and kern in r["kernel"] and int(r["launches"]) > 3:
                    g = int(r["grid_size"] or 0)
                    if r["counter"] not in best or g > best[r["counter"]][0]:
                        best[r["counter"]] = (g, float(r["mean_value_per_launch"]))
            if best:
                counters[key % b] = {c: v[1] for c, v in best.items()}
    traffic["counters"] = counters
    traffic["_meta"] = {"pass": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round %s, profiles/%s_pmc_summary.csv" % (TAG, TAG), "kernels": kern_names}
    traffic["_note"] = ("batch_B / pair_batch_B: HBM-side bytes per launch of anchor_l1*_kernel / pair_mfma4_kernel at B frame-pairs per step, default "
                        "arithmetic, = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes (profiles/%s_pmc_summary.csv); "
                        "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-byte fabric requests of a wide coalesced stream at "
                        "64 bytes; confirmed here by TCC_EA0_RDREQ_sum x 128 B)" % TAG)
    json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    # the bench lines of this pass were printed with the PREVIOUS pmc_traffic.json (bench.py reads the committed file): give their
    # `traffic` fields the figures of the PMC passes of the same pass
    for f in sorted(os.listdir(P)):
        if f.startswith(TAG + "_bench_") and f.endswith(".json"):
            d = json.load(open(os.path.join(P, f)))
            if d.get("config", {}).get("arithmetic", "").startswith("f16x2") and d.get("config", {}).get("precut_weight_stream", True) and d.get("roofline"):
                b = d["config"]["frame_pairs_per_step_per_gpu"]
                for key in ("roofline", "roofline_second"):
                    if not d.get(key):
                        continue
                    k = ("pair_batch_%d" if d[key]["kernel"].startswith("pair") else "batch_%d") % b
                    if k in traffic:
                        d[key]["traffic"] = traffic[k]
                        d[key]["traffic_source"] = "rocprofv3 --pmc passes of the same measurement pass (profiles/%s_pmc_summary.csv)" % TAG
                json.dump(d, open(os.path.join(P, f), "w"), indent=1)
    print(traffic)
    print(sorted(os.listdir(P)))
