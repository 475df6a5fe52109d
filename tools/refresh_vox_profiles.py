"""Merge a K1-only measurement (tools/gpu_vox_refresh.sh) into the committed summaries: the voxeliser's kernel stats, its timing log and
its rows of the PMC summary.  usage: python tools/refresh_vox_profiles.py <round tag> <sub-directory of gpurun_out>"""
import csv
import os
import shutil
import sys
import tempfile

sys.argv, argv = sys.argv[:1], sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_profiles as SP  # noqa: E402

tag, src = argv[0], argv[1]
SP.kernel_stats(os.path.join(SP.G, src, "prof_voxelize", "d_kernel_stats.csv"), os.path.join(SP.P, tag + "_kernel_stats_voxelize.csv"))
shutil.copy(os.path.join(SP.G, src, "voxelize.log"), os.path.join(SP.P, tag + "_voxelize.txt"))
tmp = tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False).name
SP.pmc([("b16_voxfetch", src + "/pmc_voxfetch_b16"), ("b16_voxwrite", src + "/pmc_voxwrite_b16")], tmp)
new = list(csv.reader(open(tmp)))[1:]
os.unlink(tmp)
path = os.path.join(SP.P, tag + "_pmc_summary.csv")
rows = list(csv.reader(open(path)))
keep = [r for r in rows[1:] if r[0] not in ("b16_voxfetch", "b16_voxwrite")]
with open(path, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    w.writerows(keep + new)
print("replaced %d voxeliser rows by %d" % (len(rows) - 1 - len(keep), len(new)))
