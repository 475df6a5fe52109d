"""Condense rocprofv3 outputs under gpurun_out/ into the small, committed summaries under profiles/ (round tag r01)."""
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
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc(dirs, dst):
    out = []
    for label, d in dirs:
        for path in glob.glob(os.path.join(G, d, "*counter_collection.csv")):
            acc = collections.defaultdict(list)
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(path)):
                if "shasta" not in r["Kernel_Name"]:
                    continue
                k = (short(r["Kernel_Name"]), r["Counter_Name"])
                acc[k].append(float(r["Counter_Value"]))
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for (k, c), v in sorted(acc.items()):
                out.append([label, k, c, len(v), sum(v) / len(v), sum(dur[(k, c)]) / len(v)])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["run", "kernel", "counter", "launches", "mean_value_per_launch", "mean_duration_ns"])
        w.writerows(out)


if __name__ == "__main__":
    os.makedirs(P, exist_ok=True)
    kernel_stats(os.path.join(G, SRC, "prof_default/d_kernel_stats.csv"), os.path.join(P, TAG + "_kernel_stats_bench_default_b512.csv"))
    kernel_stats(os.path.join(G, SRC, "prof_b128/d_kernel_stats.csv"), os.path.join(P, TAG + "_kernel_stats_bench_b128.csv"))
    kernel_stats(os.path.join(G, SRC, "prof_b64/d_kernel_stats.csv"), os.path.join(P, TAG + "_kernel_stats_bench_b64.csv"))
    kernel_stats(os.path.join(G, SRC, "prof_b1/d_kernel_stats.csv"), os.path.join(P, TAG + "_kernel_stats_bench_b1.csv"))
    pmc([("b1_fetch", SRC + "/pmc_fetch_b1"), ("b1_write", SRC + "/pmc_write_b1"), ("b32_fetch", SRC + "/pmc_fetch_b32"),
         ("b32_write", SRC + "/pmc_write_b32"), ("b64_fetch", SRC + "/pmc_fetch_b64"), ("b64_write", SRC + "/pmc_write_b64"),
         ("b128_fetch", SRC + "/pmc_fetch_b128"), ("b128_write", SRC + "/pmc_write_b128"), ("b512_fetch", SRC + "/pmc_fetch_b512"),
         ("b512_write", SRC + "/pmc_write_b512"), ("b512_mfma", SRC + "/pmc_mfma_b512")],
        os.path.join(P, TAG + "_pmc_summary.csv"))
    for b in ("default", "b1", "b32", "b64", "b64_f32", "b128"):
        src = os.path.join(G, SRC, "bench_%s.json" % b)
        if os.path.exists(src):
            line = [l for l in open(src) if l.startswith("{")][-1]
            json.dump(json.loads(line), open(os.path.join(P, TAG + "_bench_%s.json" % b), "w"), indent=1)
    # HBM bytes per launch of the dominant kernel (bench.py: roofline.traffic), FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
    rows = list(csv.DictReader(open(os.path.join(P, TAG + "_pmc_summary.csv"))))
    traffic = {}
    for b in (1, 32, 64, 128, 512):
        for key, kern in (("batch_%d", "anchor_l1"), ("pair_batch_%d", "pair_mfma4")):
            f = [float(r["mean_value_per_launch"]) for r in rows if r["run"] == "b%d_fetch" % b and kern in r["kernel"] and r["counter"] == "FETCH_SIZE"]
            w = [float(r["mean_value_per_launch"]) for r in rows if r["run"] == "b%d_write" % b and kern in r["kernel"] and r["counter"] == "WRITE_SIZE"]
            if f and w:
                traffic[key % b] = int((2 * f[0] + w[0]) * 1024)
    traffic["_note"] = ("batch_B / pair_batch_B: HBM bytes per launch of anchor_l1*_kernel / pair_mfma4_kernel at B frame-pairs per step "
                        "= (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc "
                        "passes (profiles/%s_pmc_summary.csv); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the "
                        "bytes of a wide coalesced stream)" % TAG)
    json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    print(traffic)
    print(os.listdir(P))
