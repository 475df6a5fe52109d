#!/usr/bin/env python3
"""bench.py -- affinity frame-pairs/s at N=M=500, F=256 (BASELINE.json metric) on N MI355X of one node.

A "step" is one pass of the affinity hot path (BEV gather -> anchors -> pair residual ->
aff + softmaxes) over one batch of `--batch` synthetic frame-pairs whose inputs (NHWC BEV feature maps, box tables,
weights) are already resident in HBM.  Frame-pairs are independent, so with N GPUs every rank runs its own replica on
its own batch (weak scaling, no data-path collective); the only collectives are the barrier and the MAX over ranks of
the elapsed time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher environment starts its own N ranks (fresh child processes, one per
GPU, started before this process touches the GPU; rank 0's JSON line is the output).  Rank 0 prints ONE JSON line (see
DESIGN.md "Measurement" for the definition of every field).  On one GPU the same line carries, under `extra`, the other
operating points SURVEY.md 8(d) asks for, measured by the same process right after the headline run: frame-pairs per step
1 / 8 / 64 / 512 (the reference's eval loop runs batch 1; 512 was the headline batch of rounds 1 - 2), the strict-f32, bf16-piece and
opt-in fixed-grid arithmetic at 512 per step, and the reference's shipped car configuration (max_obj 90, num_point 5 -> F = 320,
num_feats 3).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_OBJ, NF, NPOINT, CH = 500, 7, 4, 64  # N=M=500, F=256, nf=7
HW = 180
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: dense f32-input MFMA peak (= f32 vector peak); no xf32 on gfx950
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 / f16 MFMA peak (no sparsity)
DENSE_GFLOP_PER_PAIR = 28.65            # whole forward at N=M=500, F=256, dense formulation: 26.45 pair MLPs + 2.05 anchors + 0.15 aff
HEADLINE = dict(max_obj=N_OBJ, num_feats=NF, num_point=NPOINT)
CAR = dict(max_obj=90, num_feats=3, num_point=5)  # configs/nusc/car.py:22-39 of the reference (BASELINE configs 2-3)


class Work:
    """Algorithmic work of one frame-pair for a configuration; T = D = N + 2, P = T * D pairs."""

    def __init__(self, max_obj, num_feats, num_point):
        self.N, self.nf, self.np = max_obj, num_feats, num_point
        F = self.F = CH * num_point
        self.K, self.H = max_obj * F, max_obj * F // 64
        self.pairs = (max_obj + 2) ** 2
        h1, h2, h3, r1, r2 = F // 8, F // 16, F // 32, 32 + F // 8, 8 + F // 32
        nf = num_feats
        # multiply-adds per pair: the reference's dense formulation of the three pair MLPs (first layers on the concatenated pair
        # tensor), what is left once the first layers are factorised over the table rows, and that with every layer width rounded up
        # to the 4-wide block of the 4x4x1 MFMA
        self.dense_pair_macs = (2 * F * h1 + h1 * h2 + h2 * h3 + h3) + (2 * nf * 32 + 32 * 8 + 8) + ((2 * F + 2 * nf) * r1 + r1 * r2 + r2 * 3)
        self.layer2_macs = h1 * h2 + r1 * r2 + 32 * 8
        self.useful_pair_macs = self.layer2_macs + h2 * h3 + h3 + r2 * 3 + 8
        r4 = lambda x: (x + 3) // 4 * 4  # noqa: E731
        layers = ((h1, h2), (h2, h3), (h3, 1), (r1, r2), (r2, 3), (32, 8), (8, 1))  # (in, out) of layers 2-4; + one bias slot per output
        self.executed_pair_macs = sum(r4(i) * r4(o) + r4(o) for i, o in layers)

    def l1_algorithmic_bytes(self, B):
        """aug_shape first layer (anchor_l1*_kernel).  ALGORITHMIC HBM bytes of one launch over B frame-pairs: every weight of the
        four (N*F/64, N*F) fp32 matrices ONCE (4.096 GB at the headline size, whatever the batch), the two (N*F) activation vectors
        of each batch item once, one (4*N*F/64) partial vector per batch item written.  (Weight passes beyond the first are executed
        traffic, not algorithmic - reported as `weight_passes_executed`.)"""
        return 4 * self.H * self.K * 4 + 2 * B * self.K * 4 + B * 4 * self.H * 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv, timeout_s=1800.0):
    """`python bench.py --gpus N` typed as is: start N fresh children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would), before this process has made any GPU call; rank 0's stdout (the JSON line) is passed through.
    All children are polled: the first one that fails (or the overall time limit) ends the others - they are fresh processes of
    ours, killed by PID - so a dead rank is an error code, not a hang in the other ranks' rendezvous."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + timeout_s
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0:
                    rc = abs(code)
                    print("bench.py: rank %d exited with code %d; stopping the other ranks" % (procs.index(p), code), file=sys.stderr)
        if rc == 0 and live:
            if time.monotonic() > deadline:
                rc = 124
                print("bench.py: ranks still running after %.0f s; stopping them" % timeout_s, file=sys.stderr)
            else:
                time.sleep(0.05)
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc


class EnergyCounter:
    """The GPU's accumulated-energy counter (librocm_smi64: rsmi_dev_energy_count_get) read around the timed loop: the step is
    power-limited, so joules per step is what its duration follows.  The rocm_smi device index is NOT the
    HIP ordinal under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES or a remapped lease: the device is matched by PCI bus id
    (rsmi_dev_pci_id_get against the HIP device's domain:bus:device); no match -> no energy figure (null), never another GPU's."""

    def __init__(self, torch_device):
        self.lib, self.index, self.pci = None, None, None
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch_device)
            want = tuple(int(getattr(pr, k, -1)) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
            lib = C.CDLL("librocm_smi64.so")
            if lib.rsmi_init(C.c_uint64(0)) != 0:
                return
            n = C.c_uint32()
            if lib.rsmi_num_monitor_devices(C.byref(n)) != 0:
                return
            if n.value == 1 and torch.cuda.device_count() == 1:  # nothing to confuse
                self.lib, self.index, self.pci = lib, 0, "%04x:%02x:%02x (only device)" % want
                return
            for i in range(n.value):
                bdf = C.c_uint64()
                if lib.rsmi_dev_pci_id_get(C.c_uint32(i), C.byref(bdf)) != 0:
                    continue
                v = bdf.value  # ((domain & 0xffffffff) << 32) | ((bus & 0xff) << 8) | ((device & 0x1f) << 3) | function
                if ((v >> 32) & 0xffffffff, (v >> 8) & 0xff, (v >> 3) & 0x1f) == want:
                    self.lib, self.index = lib, i
                    self.pci = "%04x:%02x:%02x" % want
                    break
        except (OSError, AttributeError, RuntimeError, AssertionError):
            pass

    def joules(self):
        if self.lib is None:
            return None
        cnt, res, ts = C.c_uint64(), C.c_float(), C.c_uint64()
        if self.lib.rsmi_dev_energy_count_get(C.c_uint32(self.index), C.byref(cnt), C.byref(res), C.byref(ts)) != 0:
            return None
        return cnt.value * res.value * 1e-6


class Bench:
    """One process = one GPU.  Holds the resident inputs (sized for the largest batch) and measures operating points."""

    def __init__(self, args, dev, rank, world, dist):
        import torch
        import shasta_amd
        from shasta_amd import hip
        self.torch, self.hip, self.shasta = torch, hip, shasta_amd
        self.args, self.dev, self.rank, self.world, self.dist = args, dev, rank, world, dist
        self.lib = hip.load()
        self.gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.models, self.inputs = {}, {}
        B = args.batch
        self.bev = torch.relu(torch.randn(B, HW, HW, CH, device=dev, generator=self.gen))
        self.pbev = torch.relu(torch.randn(B, HW, HW, CH, device=dev, generator=self.gen))

    def model(self, cfg):
        key = tuple(sorted(cfg.items()))
        if key not in self.models:
            self.torch.manual_seed(0)
            with self.torch.device(self.dev):  # random-init weights of the named architecture, created directly in HBM
                self.models[key] = self.shasta.build_simp_track(dict(
                    type="Shasta", reader=None, backbone=None, neck=None,
                    bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                    **cfg)).eval()
        return self.models[key]

    def boxes(self, cfg):
        """(det0, prev): (B, N, 11) rows [x, y, z, w, l, h, yaw, vx, vy, dt, score] as SURVEY.md 8(d) draws them"""
        key = cfg["max_obj"]
        if key not in self.inputs:
            torch, dev, g, B, n = self.torch, self.dev, self.gen, self.args.batch, cfg["max_obj"]

            def one():
                b = torch.zeros(B, n, 11, device=dev)
                b[..., 0:2] = torch.rand(B, n, 2, device=dev, generator=g) * 100 - 50
                b[..., 2] = torch.randn(B, n, device=dev, generator=g)
                b[..., 3:6] = torch.rand(B, n, 3, device=dev, generator=g) * 4 + 0.5
                b[..., 6] = (torch.rand(B, n, device=dev, generator=g) * 2 - 1) * 3.14159265
                b[..., 7:9] = torch.randn(B, n, 2, device=dev, generator=g)
                b[..., 9] = 0.5
                b[..., 10] = torch.rand(B, n, device=dev, generator=g)
                return b
            self.inputs[key] = (one(), one())
        return self.inputs[key]

    def sync_all(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def measure(self, cfg, B, steps, warmup, arithmetic, graph=False, energy=None, selfcheck=False):
        """W untimed + K timed steps of B frame-pairs (the first B of the resident inputs) bracketed by barrier + synchronize; HIP
        events around the two heaviest kernels of every timed step, on the launch stream (shasta_affinity_forward_timed_f32)."""
        torch, hip, lib = self.torch, self.hip, self.lib
        model = self.model(cfg)
        model.arithmetic = arithmetic
        model.precut_weight_stream = not self.args.no_precut
        det0, prev = self.boxes(cfg)
        det0, prev, bev, pbev = det0[:B], prev[:B], self.bev[:B], self.pbev[:B]
        det = det0.clone()
        # one-time set-up (the packed small weights, the companion image of the first aug_shape layers: 4 GB of allocation and one pass
        # over the weights) belongs in front of the warm-up steps, not inside the first of them
        w = model._weights()
        model._ensure_packed(w, self.dev)
        model._ensure_aux(w, B, self.dev)
        evs = []  # per step: (L1 start, L1 stop, pair start, pair stop)
        for _ in range(steps):
            four = tuple(C.c_void_p() for _ in range(4))
            for e in four:
                hip.check(lib.shasta_event_create(C.byref(e)), "event_create")
            evs.append(four)

        def step(ev=None):
            det.copy_(det0)  # forward back-projects det_boxes in place (shasta.py:270): restore the input
            return model.affinity_from_bev(bev, pbev, det, prev, l1_events=ev)

        g = None
        with torch.no_grad():
            if graph:
                # capture one step (all launches go through the C ABI on the capture stream; outputs are static tensors)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        m1, m2 = step()
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    m1, m2 = step()
            for _ in range(warmup):
                if g is not None:
                    g.replay()
                else:
                    m1, m2 = step()
            self.sync_all()
            e0 = energy.joules() if energy else None
            t0 = time.perf_counter()
            for i in range(steps):
                if g is not None:
                    g.replay()
                else:
                    m1, m2 = step(evs[i])
            self.sync_all()
            elapsed = time.perf_counter() - t0
            e1 = energy.joules() if energy else None
        joules = (e1 - e0) if (e0 is not None and e1 is not None and e1 > e0) else None
        if self.world > 1:
            t = torch.tensor([elapsed], device=self.dev, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert os.environ.get("SHASTA_BENCH_PROBE") == "1" or (bool(torch.isfinite(m1).all()) and abs(float(m1[0, 0].sum()) - 1.0) < 1e-4)
        check = None
        if selfcheck:
            # self-check of the operating point: three frame-pairs of the timed batch recomputed one at a time (batch 1 runs the VALU
            # weight-stream kernel and the small-batch tiles of every other stage) must reproduce the batched result
            check = 0.0
            with torch.no_grad():
                for i in sorted({0, B // 2, B - 1}):
                    s1, s2 = model.affinity_from_bev(bev[i:i + 1], pbev[i:i + 1], det0[i:i + 1].clone(), prev[i:i + 1])
                    check = max(check, float((s1 - m1[i:i + 1]).abs().max()), float((s2 - m2[i:i + 1]).abs().max()))
            # (SHASTA_BENCH_PROBE=1: a diagnostic library whose results are wrong on purpose - tools/build_variant.py ablations - is being timed)
            assert check <= 1e-6 or os.environ.get("SHASTA_BENCH_PROBE") == "1", "self-check failed: batched and one-at-a-time results differ by %.3e" % check
        ms = C.c_float()
        l1, pair = [], []
        if g is not None:  # the per-step events were not recorded under graph replay: time the two kernels separately
            with torch.no_grad():
                for i in range(steps):
                    step(evs[i])
            torch.cuda.synchronize()
        for four in evs:
            hip.check(lib.shasta_event_elapsed_ms(four[0], four[1], C.byref(ms)), "event_elapsed")
            l1.append(ms.value)
            hip.check(lib.shasta_event_elapsed_ms(four[2], four[3], C.byref(ms)), "event_elapsed")
            pair.append(ms.value)
            for e in four:
                lib.shasta_event_destroy(e)
        return dict(B=B, steps=steps, warmup=warmup, elapsed=elapsed, ms_per_step=elapsed / steps * 1e3,
                    value=self.world * B * steps / elapsed, l1_ms=sum(l1) / len(l1), pair_ms=sum(pair) / len(pair), joules=joules,
                    selfcheck=check)


def rooflines(cfg, arithmetic, r, with_traffic=True):
    """`roofline` objects of the two kernels that carry a step, from the HIP-event launch times of measurement r.
    Kernel 1: aug_shape first layer = the fp32 weight stream (4.096 GB at the headline size).
      B == 1: VALU GEMV; B <= 32: f32 MFMA kernel, 1024 matrix-pipe cycles per 4 KB weight tile against ~1300 of HBM  -> HBM-bound
      pieces: B <= 64: bf16-piece kernel (each fp32 product = 6 exact bf16 piece products), 768 cycles per tile        -> HBM-bound
              B  > 64: the same with 128 items per weight pass, 1536 cycles per tile -> bf16 matrix pipe: priced as EXECUTED bf16
                       flops (6 piece products per fp32 product) against the dense bf16 MFMA peak
      f16x2 (default): as pieces up to 64 frame-pairs; above, 3 fp16 piece products: 128 items per pass (768 cycles per tile,
                       HBM-bound) up to 128 frame-pairs, 256 items per pass (1536 cycles) above: EXECUTED f16 flops against the
                       dense f16 MFMA peak
      f32 and B > 32: f32 MFMA kernel, 64 items per pass, 2048 cycles per tile -> f32 matrix pipe
    Kernel 2: the pair kernel = layers 2-4 of fuse_shape / res_coeff / fuse_det for all (N+2)^2 pairs.  Three flop counts per
    launch: `dense` = the reference's formulation, `useful` = what is left once the first layers are factorised
    over the table rows, `executed`.  `achieved` prices the USEFUL flops against the f32 MFMA peak."""
    wk = Work(**cfg)
    B, l1_ms, pair_ms, step_ms = r["B"], r["l1_ms"], r["pair_ms"], r["ms_per_step"]
    headline = cfg == HEADLINE
    alg = wk.l1_algorithmic_bytes(B)
    hbm_gbs = alg / (l1_ms * 1e-3) / 1e9
    l1_flops = 2.0 * B * 4 * wk.H * wk.K  # dense fp32 flops of the four first layers for B frame-pairs (algorithmic = executed)
    l1_tflops = l1_flops / (l1_ms * 1e-3) / 1e12
    # the two-piece fp16 weight stream serves B > 64, and with the pre-cut weight image (default) every batch of at least 17
    f32_forced, f16x2 = arithmetic == "f32", arithmetic in ("f16x2", "f16grid") and (B > 64 or (PRECUT and B >= 17))
    wide = False
    if B == 1 or (B <= 32 and not f16x2):
        passes, nprod = 1, 1
    elif f32_forced:
        passes, nprod = -(-B // 64), 1
    elif f16x2:  # 32 / 64 / 128 items per weight pass up to 128 frame-pairs, 256 above - 512 with the pre-cut weight image when that
        # pads the batch no more (anchor_l1_wide_kernel, anchor_split.hip: split_wide); three fp16 piece products per fp32 product
        wide = PRECUT and B > 256 and -(-B // 512) * 512 <= -(-B // 256) * 256
        passes, nprod = (1 if B <= 128 else -(-B // 512) if wide else -(-B // 256)), 3
    else:        # 64 / 128 items per pass; six bf16 piece products
        passes, nprod = (1 if B <= 64 else -(-B // 128)), 6
    hbm_bound = B == 1 or (B <= 32 and not f16x2) or (not f32_forced and (B <= 128 if f16x2 else B <= 64))  # <= 1024 matrix cycles per 4 KB weight tile
    if hbm_bound:
        roof_l1 = {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS}
    elif f32_forced:
        roof_l1 = {"bound": "mfma", "achieved": l1_tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": l1_tflops / MFMA_F32_PEAK_TFLOPS, "mfma_dtype": "f32"}
    else:
        roof_l1 = {"bound": "mfma", "achieved": nprod * l1_tflops, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": nprod * l1_tflops / MFMA_BF16_PEAK_TFLOPS, "mfma_dtype": "f16" if f16x2 else "bf16",
                   "note": "executed %s MFMA flops = %d piece products per fp32 product, against the dense bf16 / f16 peak"
                           % ("f16" if f16x2 else "bf16", nprod)}
    l1_kernel = ("anchor_l1_kernel<1,8> (VALU GEMV)" if B == 1 else
                 "anchor_l1_wide_kernel (fp16 pieces, pre-cut weight image, 512 items per pass)" if (f16x2 and B > 128 and wide) else
                 ("anchor_l1_split_kernel (fp16 pieces%s)" % (", pre-cut weight image" if PRECUT else "")) if f16x2 else
                 "anchor_l1_mfma_kernel (f32 MFMA)" if (B <= 32 or f32_forced) else "anchor_l1_split_kernel (bf16 pieces)")
    tr, src = _pmc_traffic(B, "l1", l1_kernel) if (with_traffic and headline and arithmetic == "f16x2") else (None, None)
    roof_l1.update({"kernel": "%s: aug_shape.*.0, 4 x %d x %d fp32 weight stream" % (l1_kernel, wk.H, wk.K),
                    "traffic": tr, "traffic_source": src, "algorithmic_bytes_per_launch": alg, "weight_passes_executed": passes,
                    "executed_weight_bytes_per_launch": passes * 4 * wk.H * wk.K * 4,
                    "algorithmic_flops_per_launch": l1_flops, "executed_flops_per_launch": nprod * l1_flops,
                    "avg_launch_ms": l1_ms, "algorithmic_hbm_gbs": hbm_gbs, "fp32_tflops": l1_tflops,
                    "share_of_step": l1_ms / step_ms})
    pair_useful = 2.0 * wk.useful_pair_macs * wk.pairs * B
    pair_tflops = pair_useful / (pair_ms * 1e-3) / 1e12
    pair_f16 = arithmetic in ("f16x2", "f16grid") and wk.F in PAIR_F16_WIDTHS  # second layers as three fp16 piece products each
    pair_kernel = ("pair_f16w_kernel" if wk.F == 320 else "pair_f16_kernel") if pair_f16 else "pair_mfma4_kernel"
    tr, src = _pmc_traffic(B, "pair", pair_kernel) if (with_traffic and headline and arithmetic == "f16x2") else (None, None)
    roof_pair = {"bound": "mfma", "achieved": pair_tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": pair_tflops / MFMA_F32_PEAK_TFLOPS, "traffic": tr, "traffic_source": src,
                 "useful_flops_per_launch": pair_useful,
                 "dense_algorithmic_flops_per_launch": 2.0 * wk.dense_pair_macs * wk.pairs * B,
                 "dense_equivalent_tflops": 2.0 * wk.dense_pair_macs * wk.pairs * B / (pair_ms * 1e-3) / 1e12,
                 "avg_launch_ms": pair_ms, "share_of_step": pair_ms / step_ms}
    T = wk.N + 2
    if pair_f16:
        executed = 2.0 * (3 * PAIR_F16_LAYER2_SLOTS[wk.F] + (wk.executed_pair_macs - wk.layer2_macs)) * wk.pairs * B
        roof_pair.update({
            # what limits this kernel is VALU issue (cutting the hidden activations into fp16 pieces: ~2.5 vector slots per hidden value);
            # the matrix pipe idles most of the time (mfma_busy).  `frac` stays the f32-equivalent figure of rounds 1 - 5 (useful fp32
            # flops against the f32 MFMA peak - a yardstick, not this kernel's ceiling); `frac_executed` = executed matrix flops against
            # the peak of the pipe they run on (f16)
            "bound": "valu",
            "kernel": "%s: per-pair MLP tails (second layers on the f16 matrix path) + hand residual -> residual (B, %d, %d)" % (pair_kernel, T, T),
            "mfma_dtype": "f16 (layer 2: three piece products per fp32 product) + f32 (layers 3-4)",
            "executed_flops_per_launch": executed,
            "frac_f32_equiv": pair_tflops / MFMA_F32_PEAK_TFLOPS,
            "frac_executed": executed / (pair_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS})
        roof_pair.update(_pmc_counters(B, "pair", pair_kernel) if (with_traffic and headline and arithmetic == "f16x2") else {})
    else:
        roof_pair.update({"kernel": "pair_mfma4_kernel<%d,8>: per-pair MLP tails + hand residual -> residual (B, %d, %d)" % (wk.F, T, T),
                          "mfma_dtype": "f32", "executed_flops_per_launch": 2.0 * wk.executed_pair_macs * wk.pairs * B})
    if roof_l1["bound"] == "mfma" and with_traffic and headline and arithmetic == "f16x2":
        roof_l1.update(_pmc_counters(B, "l1", l1_kernel))
    return roof_l1, roof_pair


PRECUT = True                            # Shasta.precut_weight_stream of this run (main() clears it for --no-precut)
PAIR_F16_WIDTHS = (256, 320)             # feature widths with an fp16-piece pair kernel (pair_f16_kernel / pair_f16w_kernel)
PAIR_F16_LAYER2_SLOTS = {256: 2048, 320: 5120}  # multiply-add slots of the layer-2 MFMAs per pair and piece product (16 rows x 128 k; 10 fragments x 32 rows x 16 k)


def brief(cfg, arithmetic, r):
    """One extra operating point, condensed: throughput, step time, and the roofline of each of the two heavy kernels."""
    a, b = rooflines(cfg, arithmetic, r, with_traffic=False)
    keep = ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "share_of_step", "mfma_dtype")
    first, second = (a, b) if r["l1_ms"] >= r["pair_ms"] else (b, a)
    out = {"value": r["value"], "unit": "frame-pairs/s", "frame_pairs_per_step": r["B"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
           "warmup": r["warmup"], "arithmetic": arithmetic,
           "roofline": dict({k: first[k] for k in keep if k in first}, kernel=first["kernel"].split(":")[0]),
           "roofline_second": dict({k: second[k] for k in keep if k in second}, kernel=second["kernel"].split(":")[0])}
    if r["B"] == 1:
        out["latency_ms"] = r["ms_per_step"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024,
                    help="frame-pairs per step per GPU (SURVEY.md 8(d): headline = best batch; measured 512: 54.4 k, 768: 56.5 k, 1024: 57.3 k, "
                         "1536: 55.1 k, 2048: 54.4 k frame-pairs/s on one box - the same joules per frame-pair, the longer kernels hold more power)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline operating point only (no `extra` object)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (no HIP-event roofline)")
    ap.add_argument("--cpu-sample", type=int, default=16, help="frame-pairs timed on the host for cpu_baseline")
    ap.add_argument("--arithmetic", choices=["f16x2", "pieces", "f32", "f16grid"], default="f16x2",
                    help="Shasta.arithmetic: how fp32 products are formed on the matrix cores above the batch thresholds: two fp16 pieces for "
                         "the weight stream + three bf16 pieces elsewhere (default), three bf16 pieces everywhere, or f32 MFMA kernels only")
    ap.add_argument("--no-precut", action="store_true", help="Shasta.precut_weight_stream = False: cut the fp32 first-layer weights inside the "
                                                             "weight-stream kernel instead of streaming the pre-cut fp16 piece image (+4.1 GB resident)")
    ap.add_argument("--extra-file", default=os.path.join(ROOT, "bench_extra.json"),
                    help="where the long form goes (full roofline objects, the other operating points of SURVEY 8(d), notes); also printed to stderr")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction plumbing only, on the CPU with gloo (no GPU, no forward): for the CPU test suite")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run:
        return dry_run(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)
    assert args.gpus == world, "--gpus must equal the number of launched ranks (WORLD_SIZE)"

    global PRECUT
    PRECUT = not args.no_precut
    bench = Bench(args, dev, rank, world, dist)
    B = args.batch
    energy = EnergyCounter(dev)
    r = bench.measure(HEADLINE, B, args.steps, args.warmup, args.arithmetic, graph=args.graph, energy=energy, selfcheck=True)
    roof_l1, roof_pair = rooflines(HEADLINE, args.arithmetic, r)
    # `roofline` is the kernel with the longer average launch; the other one rides along as `roofline_second`
    roof, second = (roof_l1, roof_pair) if r["l1_ms"] >= r["pair_ms"] else (roof_pair, roof_l1)
    joules, elapsed = r["joules"], r["elapsed"]
    probe = os.environ.get("SHASTA_BENCH_PROBE") == "1"  # a diagnostic library (results wrong on purpose) is being timed: never a headline
    out = build_line(args, world, r, roof, second, probe)
    if probe:
        out["probe"] = True
        out["probe_value"] = r["value"]
    # everything else (the full roofline objects, the other operating points, notes) goes to a side file and to stderr: the last stdout
    # line stays a few hundred bytes per object (round 5's 22 KB line was not parsed by the driver)
    detail = {"headline": {"value": r["value"], "ms_per_step": r["ms_per_step"], "frame_pairs_per_step": B,
                           "dense_equivalent_tflops": DENSE_GFLOP_PER_PAIR * 1e9 * r["value"] / 1e12,
                           "hip_graph": bool(args.graph), "precut_weight_stream": not args.no_precut,
                           "arithmetic_note": ARITHMETIC_NOTES[args.arithmetic],
                           "energy": None if joules is None else {"joules_per_step": joules / args.steps, "avg_power_w": joules / elapsed,
                                                                  "millijoules_per_frame_pair": joules / args.steps / B * 1e3, "pci": energy.pci},
                           "roofline": roof, "roofline_second": second}}
    if rank == 0 and world == 1 and not args.no_extras and not args.graph:
        try:
            detail["extra"] = extras(bench, args)
            f32 = detail["extra"].get("arithmetic_f32", {})
            out["value_f32"] = f32.get("value")
        except Exception as err:  # noqa: BLE001  (the headline above is measured: never lose the line to an extra)
            detail["extra"] = {"error": "%s: %s" % (type(err).__name__, str(err)[:300])}
    if args.arithmetic == "f32":
        out["value_f32"] = out["value"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(bench.model(HEADLINE), args.cpu_sample)
        except Exception as err:  # noqa: BLE001
            out["cpu_baseline"] = {"error": "%s: %s" % (type(err).__name__, str(err)[:200]), "kind": "port"}
    if rank == 0:
        out["extra_file"] = write_extra(detail, args.extra_file)
        emit(out)
    if world > 1:
        dist.destroy_process_group()


def build_line(args, world, r, roof, second, probe=False):
    """The compact object of the one stdout line (every key the contract names, no prose); main() and --dry-run share it."""
    B, joules = r["B"], r.get("joules")
    return {
        "metric": "affinity frame-pairs/sec at N=M=500, F=256",
        "value": None if probe else r["value"],
        "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthetic N=M=500, F=256 (num_point 4 x C 64), nf=7: affinity forward from HBM-resident NHWC BEV maps "
                               "(SURVEY 8a rows 4-16)", "frame_pairs_per_step_per_gpu": B,
                   "max_obj": N_OBJ, "num_feats": NF, "num_point": NPOINT, "arithmetic": args.arithmetic,
                   "parallelism": "replica x%d" % world},
        "roofline": compact_roofline(roof),
        "roofline_second": compact_roofline(second),
        "cpu_baseline": None,
        "value_f32": None,
        "selfcheck_max_abs": r.get("selfcheck"),
        "mj_per_frame_pair": None if joules is None else joules / args.steps / B * 1e3,
        "extra_file": None,
    }


LINE_LIMIT = 4096  # bytes of the one stdout line (tests/test_bench_launch.py, tests/test_bench_contract.py hold it to this)

ARITHMETIC_NOTES = {
    "f16x2": "fp32 operands in HBM, fp32 accumulation throughout; the first aug_shape layer (from 17 frame-pairs per step with the pre-cut weight "
             "image, above 64 without) and the second layers of the pair MLPs form every fp32 product from three products of two range-scaled fp16 "
             "pieces per operand (max error vs float64: weight stream 5.5e-6 against 7.0e-6 for the f32 MFMA kernel, pair stage 1.8e-6 against "
             "2.3e-6), and so do the six aff layers from 8192 table rows; the row-embedding GEMMs use six products of three exact bf16 pieces",
    "pieces": "fp32 operands, fp32 accumulation; above 32 frame-pairs per step the first aug_shape layer, and from 8192 table rows the "
              "row-embedding GEMMs and the aff layers, form every fp32 product from six exact bf16 piece products",
    "f32": "fp32 operands, fp32 accumulation throughout, f32 MFMA kernels only",
    "f16grid": "as f16x2, but the pair kernel takes the fp16 pieces of its hidden activations from a fixed grid per MLP: NOT fp32-equivalent",
}


def compact_roofline(ro):
    """The judged keys of a roofline object (task statement 4) + the kernel's short name, its launch time and, where the committed
    counter pass holds them, the pipe counters; the long form is in the side file."""
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "share_of_step",
            "frac_f32_equiv", "frac_executed", "mfma_busy", "valu_busy", "valu_insts")
    c = {k: ro[k] for k in keep if k in ro}
    c["kernel"] = ro["kernel"].split(":")[0].split(" (")[0]
    return c


def write_extra(detail, path):
    """The long form: `path` (default bench_extra.json beside bench.py; the temp directory if that is read-only) and stderr."""
    text = json.dumps(detail)
    print("bench.py extra: " + text, file=sys.stderr, flush=True)
    import tempfile
    for p in (path, os.path.join(tempfile.gettempdir(), "shasta_bench_extra.json")):
        try:
            with open(p, "w") as f:
                f.write(text + "\n")
            return os.path.relpath(p, ROOT) if os.path.abspath(p).startswith(ROOT + os.sep) else p
        except OSError:
            continue
    return None


def emit(out):
    """The ONE stdout line, last thing printed; never longer than LINE_LIMIT (optional keys are dropped first, in this order)."""
    line = json.dumps(out)
    for k in ("mj_per_frame_pair", "selfcheck_max_abs", "roofline_second"):
        if len(line) <= LINE_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out)
    assert len(line) <= LINE_LIMIT, "bench line of %d bytes" % len(line)
    sys.stderr.flush()
    print(line, flush=True)


def extras(bench, args):
    """The other operating points of SURVEY.md 8(d), same process, same resident inputs (the first b frame-pairs), short runs:
    B in {1, 8, 64} at the headline size (B = 1 is the reference's eval batch, tools/nusc_shasta/eval.py:96-101: its step time is the
    latency), the other two arithmetic forms at the headline batch (`arithmetic_f32` = strict fp32 products on the f32 MFMA
    instructions), and the shipped car configuration (configs/nusc/car.py:22-39: max_obj 90, num_point 5 -> F = 320, num_feats 3)."""
    B = min(args.batch, 512)  # the other forms and the car configuration at 512 frame-pairs per step, as in rounds 1 - 2
    ex = {"note": "measured by the same process after the headline run; each entry: W warm-up + K timed steps, barrier + synchronize "
                  "on both sides; rooflines from HIP events on the launch stream as in the headline"}

    def point(cfg, b, steps, mode, note=None):
        """One extra operating point; a failure here must not cost the headline line: it is recorded in place of the entry."""
        try:
            # warm-up: the clock needs ~45 ms to settle after the lighter points before a heavy one (tools/thermal_probe.py)
            warm = 10 if b >= 256 else 5 if steps >= 30 else 3
            e = brief(cfg, mode, bench.measure(cfg, b, steps, warm, mode))
            if note:
                e["note"] = note
            return e
        except Exception as err:  # noqa: BLE001
            return {"error": "%s: %s" % (type(err).__name__, str(err)[:300]), "frame_pairs_per_step": b, "arithmetic": mode}

    sweep = {}
    for b, k in ((1, 200), (8, 100), (64, 50), (512, 30)):
        if b <= args.batch and b != args.batch:
            sweep["b%d" % b] = point(HEADLINE, b, k, args.arithmetic)
    ex["batch_sweep"] = sweep
    for mode in ("f32", "pieces"):
        if mode != args.arithmetic:
            ex["arithmetic_" + mode] = point(HEADLINE, B, 10, mode)
    if args.arithmetic == "f16x2":
        # the opt-in fixed-grid form of the pair kernel's fp16 pieces: NOT fp32-equivalent (errors of `residual` about 5x the fp32
        # kernels', 1e-6 of its range; inside BASELINE's 1e-4 with the arg-max unchanged) - reported next to the headline, never as it
        ex["arithmetic_f16grid"] = point(HEADLINE, B, 20, "f16grid",
                                         "opt-in Shasta.arithmetic = 'f16grid': fp16 pieces of the pair kernel's hidden activations on a fixed "
                                         "grid per MLP; not fp32-equivalent")
    car = {}
    for b, k in ((1, 200), (8, 200), (B, 30)):
        if b <= B:
            car["b%d" % b] = point(CAR, b, k, args.arithmetic)
    try:  # rows 4-16 of one frame pair replayed from a captured hipGraph: launch gaps out of the way
        model = bench.model(CAR)
        model.arithmetic = args.arithmetic
        det0, prev = bench.boxes(CAR)
        det0, prev, bev1, pbev1 = det0[:1], prev[:1], bench.bev[:1], bench.pbev[:1]
        work = det0.clone()

        def one():
            work.copy_(det0)
            return model.affinity_from_bev(bev1, pbev1, work, prev)
        with bench.torch.no_grad():
            ms = graph_ms(bench.torch, one, 500)
        car["b1_graph"] = {"ms_per_step": ms, "frame_pairs_per_s": 1e3 / ms, "arithmetic": args.arithmetic,
                           "note": "batch 1 replayed from a captured hipGraph (events around 500 replays); b1 is the eager loop"}
    except Exception as err:  # noqa: BLE001
        car["b1_graph"] = {"error": "%s: %s" % (type(err).__name__, str(err)[:300])}
    if B >= 64:
        car["b%d_pieces" % B] = point(CAR, B, 30, "pieces")
    car["config"] = "max_obj 90, num_point 5 (F = 320), num_feats 3: configs/nusc/car.py:22-39 of the reference; 180 x 180 x 64 BEV maps"
    ex["car_90_320_3"] = car
    bench.model(HEADLINE).arithmetic = args.arithmetic
    n5 = {}
    cfg5 = dict(max_obj=N_OBJ, num_feats=3, num_point=5)
    for b, k in ((1, 100), (64, 30), (B, 10)):
        if b <= B:
            n5["b%d" % b] = point(cfg5, b, k, args.arithmetic)
    n5["config"] = "max_obj 500, num_point 5 (F = 320), num_feats 3: BASELINE config 3's table shape (all 7 classes, N,M <= 500, configs/nusc/*.py)"
    ex["n500_f320_nf3"] = n5
    bench.models.pop(tuple(sorted(cfg5.items())), None)  # 6.4 GB of weights + their companion image: released before the next extras
    torch = bench.torch
    torch.cuda.empty_cache()
    for name, fn in (("pipeline", extra_pipeline), ("shared_conv", extra_shared_conv), ("voxelize", extra_voxelize),
                     ("train_step", extra_train_step)):
        try:
            fn(bench, args, ex)
        except Exception as err:  # noqa: BLE001
            ex[name] = {"error": "%s: %s" % (type(err).__name__, str(err)[:300])}
    return ex


def timed_ms(torch, fn, iters, warm=3):
    """mean wall time of fn() in ms over `iters` calls, synchronised on both sides, after `warm` untimed calls"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def extra_voxelize(bench, args, ex):
    """K1 (SURVEY.md 8(a) rows 1-2, det3d/ops/point_cloud/point_cloud_ops.py:112-184 + voxel_encoder.py:18-28): hard voxelisation + per-voxel
    mean of one nuScenes-sized cloud (3e5 points, 10 sweeps: loading.py:128-148), the configuration of configs/nusc/car.py:120-125.
    Algorithmic bytes: the points read once, the (V, 10, 5) voxels + coordinates + counts + means written once."""
    import numpy as np
    torch, dev = bench.torch, bench.dev
    from shasta_amd.voxel_generator import points_to_voxel_device
    VS, RG = np.array([0.075, 0.075, 0.2], np.float32), np.array([-54, -54, -5, 54, 54, 3], np.float32)
    rng = np.random.default_rng(0)
    P = 300000
    pts = np.zeros((P, 5), np.float32)
    r = np.abs(rng.normal(0, 18, size=P)).astype(np.float32)
    th = rng.uniform(0, 2 * np.pi, size=P).astype(np.float32)
    pts[:, 0], pts[:, 1] = r * np.cos(th), r * np.sin(th)
    pts[:, 2] = rng.normal(-1.5, 0.6, size=P)
    pts[:, 3] = rng.uniform(0, 255, size=P)
    d = torch.from_numpy(pts).to(dev)
    out = points_to_voxel_device(d, VS, RG, 10, 160000, with_mean=True)
    V = int(out[0].shape[0])
    ms = timed_ms(torch, lambda: points_to_voxel_device(d, VS, RG, 10, 160000, with_mean=True), 20)
    alg = P * 5 * 4 + V * (10 * 5 * 4 + 3 * 4 + 4 + 5 * 4)
    e = {"points": P, "voxels": V, "ms": ms, "note": "one call incl. output allocation and the host read of the voxel count",
         "algorithmic_bytes": alg, "algorithmic_gbs": alg / ms / 1e6, "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS, "clouds_per_s": 1e3 / ms}
    # the clouds of a batch in one chain of launches (current + previous cloud of 8 samples: preprocess.py:179-208 voxelises both), the voxel
    # counts left on the device: no host read inside the timed loop
    from shasta_amd.voxel_generator import points_to_voxel_batch_device
    n = 16
    allp = torch.cat([d[torch.randperm(P, device=dev, generator=bench.gen)] for _ in range(n)])
    offs = [P * i for i in range(n + 1)]
    nv = points_to_voxel_batch_device((allp, offs), VS, RG, 10, 160000, with_mean=True)[4]
    msb = timed_ms(torch, lambda: points_to_voxel_batch_device((allp, offs), VS, RG, 10, 160000, with_mean=True), 10)
    Vb = int(nv.sum())
    algb = n * P * 5 * 4 + Vb * (10 * 5 * 4 + 3 * 4 + 4 + 5 * 4)
    e["batch16"] = {"clouds": n, "points_per_cloud": P, "voxels": Vb, "ms": msb, "clouds_per_s": n / msb * 1e3, "algorithmic_bytes": algb,
                    "algorithmic_gbs": algb / msb / 1e6, "hbm_frac": algb / msb / 1e6 / HBM_PEAK_GBS,
                    "note": "shasta_voxelize_mean_batch_f32: one chain of 9 launches (3 memsets + 6 kernels) for 16 clouds, num_voxels stays on the device"}
    del allp
    if not args.no_cpu_baseline:
        from oracle import voxelize_oracle as VO  # the checker's C twin of the serial reference loop, timed on one host core
        VO.points_to_voxel(pts, VS, RG, 10, 160000)
        t0 = time.perf_counter()
        for _ in range(3):
            VO.points_to_voxel(pts, VS, RG, 10, 160000, with_mean=True)
        e["cpu_oracle_ms"] = (time.perf_counter() - t0) / 3 * 1e3
        e["cpu_oracle"] = "oracle/voxelize_oracle.c (restatement of the reference's numba loop), 1 thread"
    ex["voxelize"] = e


def extra_train_step(bench, args, ex):
    """Row 19 / BASELINE config 5 (tools/nusc_shasta/train.py:198-218): forward + masked NLL + HIP backward + fused Adam of the affinity
    network on one GPU, synthetic batch; fp32 and Shasta.train_precision = "bf16".  Split by torch events on the launch stream."""
    torch, dev = bench.torch, bench.dev
    from shasta_amd import training
    ts = {"note": "ms per step = forward (rows 4-16) + loss + backward + Adam; frame-pairs/s = B / step; fwd / bwd / adam from events on the stream"}
    ts["note"] += ("; *_dense: the four first-layer gradients of aug_shape written out and read back by Adam (36 B per parameter), the others: "
                   "Adam straight from their factors (FusedAdam(lowrank_first_layers=model), 24 B per parameter), fp32 / bf16: at steps of up to 16 frame-pairs "
                   "inside loss.backward(), in the pass that also forms dx = ghid W1 (in_backward=True: that time is then in bwd_ms, adam_ms holds "
                   "the other tensors); *_stepafter: the same with the update in step(); *_densepairs: the pair MLPs' "
                   "backward in the dense formulation of round 4 (hidden activations of every pair in HBM, strided GEMMs: "
                   "Shasta.dense_pair_backward), the others (fp32): recomputed and back-propagated per pair on chip (csrc/pair_bwd.hip); "
                   "bf16: the GEMMs around them (first-layer tables, aff) with bf16 operands; *_from_neck: the whole step of the reference "
                   "(Shasta.forward from two (B, 512, 180, 180) neck outputs: shared_conv in train() mode hand-written, forward + backward + "
                   "its four parameters in the update), *_from_neck_miopen: the same with shared_conv through nn.Sequential")
    for (cfg, B, steps) in ((CAR, 16, 10), (CAR, 64, 6), (HEADLINE, 8, 3)):
        for prec in ("fp32", "bf16", "fp32_stepafter", "fp32_dense", "fp32_densepairs"):
            if prec.startswith(("fp32_dense", "fp32_stepafter")) and cfg is not HEADLINE:
                continue
            key = "n%d_b%d_%s" % (cfg["max_obj"], B, prec)
            try:
                torch.manual_seed(0)
                with torch.device(dev):
                    model = bench.shasta.build_simp_track(dict(
                        type="Shasta", reader=None, backbone=None, neck=None,
                        bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8), **cfg)).train()
                model.train_precision = prec.split("_")[0]
                model.dense_pair_backward = prec.endswith("_densepairs")
                params = training.affinity_params(model)
                opt = training.FusedAdam(params, lr=1e-4, lowrank_first_layers=None if prec.endswith("_dense") else model,
                                         in_backward=prec in ("fp32", "bf16"))
                N = cfg["max_obj"]
                det0, prev = bench.boxes(cfg)
                det0, prev, bev, pbev = det0[:B], prev[:B], bench.bev[:B], bench.pbev[:B]
                gt = (torch.rand(B, N + 2, N + 2, device=dev, generator=bench.gen) < 0.02).float()
                gt[:, 0, 0] = 1
                evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]

                def one(ev=None):
                    opt.zero_grad(set_to_none=True)
                    if ev:
                        ev[0].record()
                    m1, m2 = training.affinity_train(model, bev, pbev, det0.clone(), prev.clone())
                    loss = training.affinity_loss(m1, m2, gt)
                    if ev:
                        ev[1].record()
                    loss.backward()
                    if ev:
                        ev[2].record()
                    opt.step()
                    if ev:
                        ev[3].record()
                for _ in range(2):
                    one()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    one(evs[i])
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / steps * 1e3
                ts[key] = {"ms_per_step": ms, "frame_pairs_per_s": B / ms * 1e3,
                           "fwd_ms": sum(e[0].elapsed_time(e[1]) for e in evs) / steps, "bwd_ms": sum(e[1].elapsed_time(e[2]) for e in evs) / steps,
                           "adam_ms": sum(e[2].elapsed_time(e[3]) for e in evs) / steps}
                if prec == "fp32":
                    # the step as the reference runs it (tools/nusc_shasta/train.py:198-218 through Shasta.forward): from the frozen neck's
                    # outputs, i.e. with shared_conv in train() mode - conv + batch-statistics BatchNorm + ReLU forward, BatchNorm / ReLU
                    # backward and the weight gradient - hand-written (csrc/shared_conv_train.hip) and, beside it, through the module's
                    # own nn.Sequential (MIOpen conv + wgrad, ATen BatchNorm, autograd)
                    x = torch.relu(torch.randn(B, 512, HW, HW, device=dev, generator=bench.gen))
                    xp = torch.relu(torch.randn(B, 512, HW, HW, device=dev, generator=bench.gen))
                    conv_params = list(model.shared_conv.parameters())
                    opt2 = training.FusedAdam(conv_params, lr=1e-4)
                    for hand in (True, False):
                        model.hand_written_train_conv = hand

                        def neck_step(ev=None):
                            opt.zero_grad(set_to_none=True)
                            opt2.zero_grad(set_to_none=True)
                            if ev:
                                ev[0].record()
                            m1, m2, _ = model(dict(det_boxes=det0.clone(), prev_det_boxes=prev.clone(), bev_map=x, prev_bev_map=xp), train_mode=True)
                            loss = training.affinity_loss(m1, m2, gt)
                            if ev:
                                ev[1].record()
                            loss.backward()
                            if ev:
                                ev[2].record()
                            opt.step()
                            opt2.step()
                            if ev:
                                ev[3].record()
                        for _ in range(2):
                            neck_step()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for i in range(steps):
                            neck_step(evs[i])
                        torch.cuda.synchronize()
                        ms = (time.perf_counter() - t0) / steps * 1e3
                        ts[key + ("_from_neck" if hand else "_from_neck_miopen")] = {
                            "ms_per_step": ms, "frame_pairs_per_s": B / ms * 1e3,
                            "fwd_ms": sum(e[0].elapsed_time(e[1]) for e in evs) / steps, "bwd_ms": sum(e[1].elapsed_time(e[2]) for e in evs) / steps,
                            "adam_ms": sum(e[2].elapsed_time(e[3]) for e in evs) / steps}
                    del x, xp, opt2
                del model, params, opt
                torch.cuda.empty_cache()
            except Exception as err:  # noqa: BLE001
                ts[key] = {"error": "%s: %s" % (type(err).__name__, str(err)[:200])}
    ex["train_step"] = ts


def extra_pipeline(bench, args, ex):
    """BASELINE configs 2-4 as one chain on a synthetic split (20 scenes x 40 frames, all seven class models, N = 20 .. 90, F = 320,
    nf = 3): per-frame files -> loader -> neck maps (device stand-in for the out-of-scope backbone) -> shared_conv of all class heads in
    one launch -> per-class affinity forward -> device decode -> merge -> PubTrackerMerged (shasta_amd.pipeline.run_split;
    tools/nusc_shasta/eval.py:96-181, merge_results.py:37-59, pub_test.py:88-162).  Frames/s of the whole chain, the per-stage split
    (device synchronised at every stage boundary, so a little slower than the un-instrumented total), and the reference-style
    frame-by-frame chain on the host (oracle/pipeline_oracle.py, features instead of neck maps: no K0 on the host side)."""
    import shutil
    import tempfile
    torch, dev = bench.torch, bench.dev
    from shasta_amd import pipeline, scenes
    root = tempfile.mkdtemp(prefix="shasta_split_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        import gc
        gc.collect()
        torch.cuda.empty_cache()  # the earlier extras leave GBs of cached blocks of other sizes: the chain's 2.7 GB map stacks start from a clean pool
        n_scenes, n_frames = 20, 40
        paths, sc = scenes.write_synthetic_split(root, n_scenes=n_scenes, frames_per_scene=n_frames, seed=3)
        models = {n: pipeline.build_class_model(n, dev, seed=1) for n in pipeline.CLASS_CONFIGS}
        for m in models.values():
            m.arithmetic = args.arithmetic
        neck = scenes.TokenNeck()
        wpaths, wsc = scenes.write_synthetic_split(os.path.join(root, "warm"), n_scenes=2, frames_per_scene=n_frames, seed=5)
        pipeline.run_split(models, wpaths, wsc, neck, dev, batch_pairs=n_frames)  # warm-up: weight packs, allocator
        reps = []
        for _ in range(3):  # best of three, all listed (the host side is Python: it varies with the box's CPU)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipeline.run_split(models, paths, sc, neck, dev, batch_pairs=n_frames)
            torch.cuda.synchronize()
            reps.append(time.perf_counter() - t0)
        total = min(reps)
        timer = pipeline.StageTimer(sync=True)
        pipeline.run_split(models, paths, sc, neck, dev, batch_pairs=n_frames, timer=timer)
        st = timer.seconds
        n = n_scenes * n_frames
        fwd_s = pipeline.forward_only_seconds(models, paths, sc, neck, dev, batch_pairs=n_frames)
        cpu = "?"
        try:
            with open("/proc/cpuinfo") as fh:
                cpu = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), "?")
        except OSError:
            pass
        e = {"frames": n, "classes": 7, "frames_per_run": n_frames, "seconds": total, "seconds_all_runs": [round(v, 4) for v in reps], "frames_per_s": n / total,
             "host_cpu": cpu,  # the chain is half host Python: 0.66 - 0.72 s on an EPYC 9575F box, 0.9 - 1.4 s on the slowest box seen
             "class_frame_pairs_per_s": 7 * n / total,
             "stages_s": {k: round(v, 4) for k, v in st.items()},
             "forward_only_s": fwd_s, "forward_only_frames_per_s": n / fwd_s, "chain_over_forward_only": fwd_s / total,
             "note": "frames_per_s: whole chain; forward_only = device time (HIP events) of K0 for all heads + the seven class forwards + decode kernels over the same "
                     "runs, nothing else; chain_over_forward_only = their ratio.  What separates the two is host Python on the "
                     "reference's formats (per-frame json files -> class dicts -> decode lists -> result rows; the tracker itself is one kernel launch per split).  stages_s: a separate pass with the "
                     "device synchronised at every stage boundary (slower than the un-instrumented total)"}
        if not args.no_cpu_baseline:
            from oracle import pipeline_oracle as PO
            small_root = os.path.join(root, "small")
            spaths, ssc = scenes.write_synthetic_split(small_root, n_scenes=1, frames_per_scene=4, seed=4)
            W = {nm: {k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for nm, m in models.items()}
            t0 = time.perf_counter()
            PO.run_split(W, pipeline.CLASS_CONFIGS, spaths, ssc, scenes.TokenBev())
            dt = time.perf_counter() - t0
            e["cpu_oracle_chain"] = {"frames": 4, "seconds": dt, "frames_per_s": 4 / dt, "threads": torch.get_num_threads(),
                                     "note": "oracle/pipeline_oracle.py: one frame pair at a time per class, torch-CPU forward from NHWC "
                                             "features (the host side never runs K0), reference-style decode loop and tracker"}
        ex["pipeline"] = e
    finally:
        shutil.rmtree(root, ignore_errors=True)


def graph_ms(torch, fn, iters, warm=5):
    """fn() captured into a hipGraph (after `warm` eager calls on a side stream), replayed `iters` times between two events on the current
    stream: mean device time of one replay in ms - what a batch-1 loop costs once the host's launch work is out of the way."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def extra_shared_conv(bench, args, ex):
    """K0 (SURVEY.md 8(f)-1, det3d/models/tracker/shasta.py:42-47,223-228): the 3x3 512 -> 64 convolution + BN + ReLU -> NHWC that a real
    nuScenes forward starts with, both maps of a frame pair per launch.  Arithmetic "f16x2" (csrc/shared_conv_f16.hip, incl. its
    max-reduction pass over the maps), strict "f32" (csrc/shared_conv.hip), MIOpen (the module's own nn.Sequential) beside them, and all
    seven class heads in one launch.  `from_neck_b1`: the shipped car configuration from the neck output (conv + rows 4-16), batch 1."""
    torch, dev = bench.torch, bench.dev
    from shasta_amd.shared_conv import SharedConvBank
    flop_pair = 2 * 2 * HW * HW * 64 * 512 * 9  # both maps
    car = bench.model(CAR)
    heads = [car]
    for i in range(6):
        torch.manual_seed(100 + i)
        with torch.device(dev):
            heads.append(bench.shasta.build_simp_track(dict(
                type="Shasta", reader=None, backbone=None, neck=None,
                bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                max_obj=4, num_feats=3, num_point=5)).eval())
    bank = SharedConvBank(heads)
    sc = {"note": "ms per call of B frame pairs (two (B,512,180,180) fp32 maps -> two (B,180,180,64) NHWC maps); tflops = fp32-equivalent "
                  "(38.2 GFLOP per frame pair and head); f16x2 includes the max-reduction pass over the maps"}
    with torch.no_grad():
        for B in (1, 8):
            x = torch.relu(torch.randn(B, 512, HW, HW, device=dev, generator=bench.gen))
            xp = torch.relu(torch.randn(B, 512, HW, HW, device=dev, generator=bench.gen))
            e = {}
            for mode in ("f16x2", "f32"):
                car.arithmetic = mode
                ms = timed_ms(torch, lambda: car.shared_conv_nhwc(x, xp), 30 if B == 1 else 10)
                e[mode] = {"ms": ms, "ms_per_frame_pair": ms / B, "tflops": B * flop_pair / ms / 1e9}
            ms = timed_ms(torch, lambda: (car.shared_conv(x).permute(0, 2, 3, 1).contiguous(), car.shared_conv(xp).permute(0, 2, 3, 1).contiguous()),
                          30 if B == 1 else 10)
            e["miopen"] = {"ms": ms, "ms_per_frame_pair": ms / B, "tflops": B * flop_pair / ms / 1e9}
            ms = timed_ms(torch, lambda: bank(x, xp), 20 if B == 1 else 5)
            e["f16x2_7_heads"] = {"ms": ms, "ms_per_frame_pair_per_head": ms / B / 7, "tflops": 7 * B * flop_pair / ms / 1e9}
            # the producer of the maps names their largest magnitude (relu of normals from 24-bit uniforms: < 5.8): no maxima pass
            one = SharedConvBank([car])
            ms = timed_ms(torch, lambda: one(x, xp, bound=8.0), 30 if B == 1 else 10)
            e["f16x2_bounded"] = {"ms": ms, "ms_per_frame_pair": ms / B, "tflops": B * flop_pair / ms / 1e9,
                                  "note": "shasta_shared_conv_multi_bounded_f32: the caller supplies max |x|, the maps are not read a second time"}
            ms = timed_ms(torch, lambda: bank(x, xp, bound=8.0), 20 if B == 1 else 5)
            e["f16x2_7_heads_bounded"] = {"ms": ms, "ms_per_frame_pair_per_head": ms / B / 7, "tflops": 7 * B * flop_pair / ms / 1e9}
            e["f16x2_over_miopen"] = e["miopen"]["ms"] / e["f16x2"]["ms"]
            sc["b%d" % B] = e
            if B == 1:
                car.arithmetic = args.arithmetic
                det0, prev = bench.boxes(CAR)
                det0, prev = det0[:1], prev[:1]

                def fwd():
                    return car(dict(det_boxes=det0.clone(), prev_det_boxes=prev, bev_map=x, prev_bev_map=xp), train_mode=False)
                ms = timed_ms(torch, fwd, 50, warm=5)
                ex["car_90_320_3"]["from_neck_b1"] = {"ms_per_step": ms, "frame_pairs_per_s": 1e3 / ms, "arithmetic": args.arithmetic,
                                                       "note": "Shasta.forward from the neck outputs: shared_conv (both maps) + rows 4-16, one frame pair"}
                ms = timed_ms(torch, lambda: car(dict(det_boxes=det0.clone(), prev_det_boxes=prev, bev_map=x, prev_bev_map=xp, bev_map_bound=8.0),
                                                 train_mode=False), 50, warm=5)
                ex["car_90_320_3"]["from_neck_b1_bounded"] = {"ms_per_step": ms, "frame_pairs_per_s": 1e3 / ms, "arithmetic": args.arithmetic,
                                                               "note": "example['bev_map_bound'] = the producer's max |x| of the maps: shared_conv skips its maxima pass"}
                try:  # the same step replayed from a captured hipGraph (tools/nusc_shasta/eval.py:96-101 is a batch-1 loop)
                    work = det0.clone()

                    def fwd_static():
                        work.copy_(det0)
                        return car(dict(det_boxes=work, prev_det_boxes=prev, bev_map=x, prev_bev_map=xp), train_mode=False)
                    ms = graph_ms(torch, fwd_static, 200)
                    ex["car_90_320_3"]["from_neck_b1_graph"] = {"ms_per_step": ms, "frame_pairs_per_s": 1e3 / ms, "arithmetic": args.arithmetic,
                                                                 "note": "the same step replayed from a captured hipGraph, events around 200 replays"}
                except Exception as err:  # noqa: BLE001
                    ex["car_90_320_3"]["from_neck_b1_graph"] = {"error": "%s: %s" % (type(err).__name__, str(err)[:300])}
            del x, xp
    car.arithmetic = args.arithmetic
    ex["shared_conv"] = sc


def _pmc_file():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def _same_kernel(recorded, running):
    return bool(recorded) and recorded.split("<")[0].split(" ")[0] == running.split("<")[0].split(" ")[0]


def _pmc_traffic(B, kernel, running):
    """(HBM bytes per launch, source) of one of the two heaviest kernels of the DEFAULT arithmetic.  The figure is NOT measured by
    this run: it comes from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json: (2 FETCH_SIZE + WRITE_SIZE) 1024,
    FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes), and is reported only when that file was recorded for the kernel
    this run launches; (None, reason) otherwise."""
    d = _pmc_file()
    if d is None:
        return None, "no profiles/pmc_traffic.json"
    meta = d.get("_meta", {})
    key = ("batch_%d" if kernel == "l1" else "pair_batch_%d") % B
    if d.get(key) is None:
        return None, "profiles/pmc_traffic.json holds no pass at %d frame-pairs per step" % B
    recorded = meta.get("kernels", {}).get(key, "")
    if not _same_kernel(recorded, running):
        return None, "profiles/pmc_traffic.json was recorded for %r, this run launches %r" % (recorded, running)
    return d[key], "static: profiles/pmc_traffic.json (%s; kernel %s), not measured by this run" % (meta.get("pass", "rocprofv3 --pmc pass"), recorded)


def _pmc_counters(B, kernel, running):
    """Pipe counters of the same committed pass (profiles/pmc_traffic.json `counters`: mean per launch of SQ_VALU_MFMA_BUSY_CYCLES,
    SQ_INSTS_VALU, GRBM_GUI_ACTIVE at B frame-pairs per step), as fractions of the launch (profiles/README.md has the formulas):
      mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
      valu_busy = 4 cycles x SQ_INSTS_VALU (wave-instructions, a 64-wide wave takes four passes of the 16-lane SIMD) / the same denominator
    Static like `traffic`; {} when the pass was recorded for another kernel."""
    d = _pmc_file() or {}
    key = ("batch_%d" if kernel == "l1" else "pair_batch_%d") % B
    c = d.get("counters", {}).get(key)
    if not c or not _same_kernel(d.get("_meta", {}).get("kernels", {}).get(key, ""), running):
        return {}
    simd_cycles = 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0
    return {"mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, "valu_busy": 4.0 * c["SQ_INSTS_VALU"] / simd_cycles,
            "valu_insts": c["SQ_INSTS_VALU"], "counters_source": "static: profiles/pmc_traffic.json `counters` (%s)" % d.get("_meta", {}).get("pass", "")}


def cpu_baseline(model, sample):
    """The CPU oracle (a restatement of the reference PyTorch forward, oracle/shasta_oracle.py) timed on this box's
    host cores on a bounded sample of the same workload: `sample` frame-pairs, one at a time (the reference's
    eval batch size, tools/nusc_shasta/eval.py:96-101), same timed region (rows 4-16, inputs resident in host memory)."""
    import torch

    from oracle import shasta_oracle as O
    w = {k: v.detach().cpu() for k, v in model.state_dict().items() if not k.startswith("shared_conv")}
    g = torch.Generator().manual_seed(99)
    bev = torch.relu(torch.randn(1, HW, HW, CH, generator=g))
    pbev = torch.relu(torch.randn(1, HW, HW, CH, generator=g))
    det, prev = O.synth_boxes(g, 1, N_OBJ), O.synth_boxes(g, 1, N_OBJ)
    O.forward_from_bev(w, bev, pbev, det.clone(), prev, NF, NPOINT)  # warm-up (first call pays allocator/oneDNN setup)
    # give the CPU its best thread count: all cores is not the fastest for these small ops on a many-core host
    all_threads = torch.get_num_threads()
    best_n, best_t = all_threads, None
    for n in sorted({min(all_threads, c) for c in (8, 16, 32, 64, all_threads)}):
        torch.set_num_threads(n)
        O.forward_from_bev(w, bev, pbev, det.clone(), prev, NF, NPOINT)
        t0 = time.perf_counter()
        for _ in range(2):
            O.forward_from_bev(w, bev, pbev, det.clone(), prev, NF, NPOINT)
        dt = (time.perf_counter() - t0) / 2
        if best_t is None or dt < best_t:
            best_n, best_t = n, dt
    torch.set_num_threads(best_n)
    t0 = time.perf_counter()
    for _ in range(sample):
        O.forward_from_bev(w, bev, pbev, det.clone(), prev, NF, NPOINT)
    dt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    host = os.cpu_count() or all_threads
    return {"value": sample / dt, "unit": "frame-pairs/s", "cores": host, "threads": best_n, "kind": "port",
            "sample": CPU_SAMPLE_TEXT % (sample, dt, host, all_threads, best_n)}


CPU_SAMPLE_TEXT = ("%d frame-pairs at N=M=500,F=256, batch 1, torch-CPU fp32 oracle, %.1f s on a host with %d logical CPUs; the fastest of "
                   "{8,16,32,64,%d} torch threads was used: %d")


def dry_run(args, rank, world):
    """CPU plumbing check of the multi-rank launch (no GPU, no forward): rendezvous over gloo, the barrier and the MAX
    reduction over ranks that bracket the timed region, and the rank-0 JSON line with `value` null."""
    import torch
    import torch.distributed as dist
    if os.environ.get("SHASTA_BENCH_DRY_FAIL_RANK") == str(rank):  # test hook: a rank that dies before the rendezvous
        sys.exit(7)
    if world > 1:
        dist.init_process_group(backend="gloo")
        dist.barrier()
    # the timed region of measure() in miniature: every rank "works" 1 ms per step between the barriers, the slowest rank's time is
    # the job's (SHASTA_BENCH_DRY_SLOW_RANK = r: that rank takes half a second longer - the MAX must show it, whichever rank it is)
    t0 = time.perf_counter()
    time.sleep(1e-3 * args.steps + (0.5 if os.environ.get("SHASTA_BENCH_DRY_SLOW_RANK") == str(rank) else 0.0))
    own = time.perf_counter() - t0
    t = torch.tensor([float(rank + 1), own], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    assert int(t[0].item()) == world and args.gpus == world
    if rank == 0:
        # the line has the shape of a measured one (same builder, rooflines priced from stand-in launch times, the longest cpu_baseline
        # sample text) with `value` null, so that the CPU suite can hold its length to LINE_LIMIT
        B = args.batch
        ms = t[1].item() / max(1, args.steps) * 1e3
        r = {"B": B, "value": None, "ms_per_step": ms, "l1_ms": 0.3 * ms, "pair_ms": 0.48 * ms, "joules": None, "selfcheck": None}
        roof_l1, roof_pair = rooflines(HEADLINE, args.arithmetic, r)
        out = build_line(args, world, r, roof_pair, roof_l1)
        out["cpu_baseline"] = {"value": None, "unit": "frame-pairs/s", "cores": os.cpu_count(), "threads": 0, "kind": "port",
                               "sample": CPU_SAMPLE_TEXT % (args.cpu_sample, 0.0, os.cpu_count() or 0, 0, 0)}
        out.update({"dry_run": True, "max_over_ranks": t[0].item(), "elapsed_max_over_ranks_s": t[1].item(), "elapsed_rank0_s": own})
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
