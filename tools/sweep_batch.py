"""Run bench.py over several batch sizes and print one line per size (GPU box helper, not part of the bench contract)."""
import json
import subprocess
import sys

sizes = [int(x) for x in sys.argv[1:]] or [1, 2, 8, 16, 32, 64]
for B in sizes:
    r = subprocess.run([sys.executable, "bench.py", "--batch", str(B), "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line:
        j = json.loads(line[-1])
        print("B=%-3d %9.1f fp/s  %8.3f ms/step  l1 %.3f ms (%d GB/s, %.0f%% of step)" % (
            B, j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["achieved"],
            100 * j["roofline"]["share_of_step"]), flush=True)
    else:
        print(B, "FAILED", r.stderr[-1500:], flush=True)
