"""Merge a training-only measurement (tools/gpu_train_refresh.sh) into the committed summaries: the kernel stats of the two training
profiles and their timing lines.  usage: python tools/refresh_train_profiles.py <round tag> <sub-directory of gpurun_out>"""
import os
import shutil
import sys

sys.argv, argv = sys.argv[:1], sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_profiles as SP  # noqa: E402

tag, src = argv[0], argv[1]
for name, dst in (("prof_train_n500", "train_n500_b8"), ("prof_train_n90", "train_n90_b64")):
    SP.kernel_stats(os.path.join(SP.G, src, name, "d_kernel_stats.csv"), os.path.join(SP.P, "%s_kernel_stats_%s.csv" % (tag, dst)))
for f in ("train_n500.log", "train_n90.log"):
    lines = [l.rstrip() for l in open(os.path.join(SP.G, src, f)) if "ms/step" in l or "fwd0" in l or "adam_lowrank" in l]
    open(os.path.join(SP.P, tag + "_" + f.replace(".log", ".txt")), "w").write("\n".join(lines) + "\n")
for f, dst in (("gather_bwd.log", "_gather_bwd.txt"), ("train_determinism.log", "_train_determinism.txt"), ("train_soak_conv.log", "_train_soak_conv.txt")):
    if os.path.exists(os.path.join(SP.G, src, f)):
        shutil.copy(os.path.join(SP.G, src, f), os.path.join(SP.P, tag + dst))
print("ok")
