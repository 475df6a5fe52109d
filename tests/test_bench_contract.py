"""bench.py contract (task statement): one JSON line with the agreed keys, measured on the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "2",
                        "--cpu-sample", "1", "--extra-file", os.path.join(ROOT, "gpurun_out", "bench_extra_test.json")], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0]  # the JSON object is the LAST stdout line
    assert len(lines[0]) < 4096, len(lines[0])  # round 5's 22 KB line was not parsed by the driver: the long form goes to extra_file + stderr
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["dtype"].startswith("f32") and j["data"] == "synthetic" and j["vs_baseline"] is None and "workload" in j["config"]
    assert j["value"] > 0 and abs(j["value"] - 2 * 1e3 / j["ms_per_step"]) < 1e-6 * j["value"]
    ro = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    # "valu": the fp16-piece pair kernel is bound by vector issue, not by a matrix pipe; its frac stays the f32-equivalent figure
    assert ro["bound"] in ("hbm", "mfma", "valu") and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9
    assert "kernel" in ro and ro["avg_launch_ms"] > 0 and "roofline_second" in j
    assert not any(isinstance(v, str) and len(v) > 200 for v in ro.values())
    extra = j["extra_file"]
    extra = extra if os.path.isabs(extra) else os.path.join(ROOT, extra)
    with open(extra) as f:
        long_form = json.load(f)
    assert long_form["headline"]["roofline"]["kernel"].startswith(ro["kernel"]) and "extra" in long_form
    cb = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
