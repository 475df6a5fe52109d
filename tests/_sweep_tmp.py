import subprocess, sys, json
for B in (1, 8, 16, 32):
    r = subprocess.run([sys.executable, "bench.py", "--batch", str(B), "--steps", "20", "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line:
        j = json.loads(line[-1]); print(B, round(j["value"],1), "fp/s", round(j["ms_per_step"],3), "ms/step; l1", round(j["roofline"]["avg_launch_ms"],3), "ms", round(j["roofline"]["achieved"]), "GB/s")
    else:
        print(B, "FAILED", r.stderr[-2000:])
