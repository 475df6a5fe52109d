#!/bin/bash
# energy / clock observation of a bench run: bash tools/gpu_energy.sh <subdir> <lib> [<lib> ...]
# reads the GPU's energy accumulator before and after every run and samples power / sclk while it runs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
rocm-smi --showenergycounter --showpower --showclocks > $O/smi_idle.txt 2>&1
for round in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    ( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/${name}_${round}_smi.txt &
    SMI=$!
    rocm-smi --showenergycounter 2>/dev/null | grep -i energy > $O/${name}_${round}_e0.txt
    SHASTA_HIP_LIB=$R/$lib python bench.py --no-cpu-baseline --steps 400 --warmup 20 > $O/${name}_$round.json 2> $O/${name}_$round.err
    rocm-smi --showenergycounter 2>/dev/null | grep -i energy > $O/${name}_${round}_e1.txt
    kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
    python - <<PY
import json,re
d=json.loads(open("$O/${name}_$round.json").read().strip().splitlines()[-1])
def e(f):
    t=open(f).read(); m=re.findall(r"([0-9.]+)\s*$", t.strip().splitlines()[-1]) if t.strip() else []
    return t.strip().replace("\n"," | ")
print("%-24s round $round: %.0f fp/s step %.3f ms | e0: %s | e1: %s" % ("$name", d["value"], d["ms_per_step"], e("$O/${name}_${round}_e0.txt")[-60:], e("$O/${name}_${round}_e1.txt")[-60:]))
PY
    sort $O/${name}_${round}_smi.txt | uniq -c | sort -rn | head -4
  done
done
cat $O/smi_idle.txt | head -30
