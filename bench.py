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
DESIGN.md "Measurement" for the definition of every field).
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
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (no sparsity)


# Algorithmic work per frame-pair at N=M=500, F=256, nf=7; P = (N+2)^2 = 252 004 pairs
PAIRS = (N_OBJ + 2) ** 2
DENSE_PAIR_MACS = 17032 + 712 + 34736   # fuse_shape + fuse_det + res_coeff per pair in the reference's dense formulation
USEFUL_PAIR_MACS = 1984                 # layers 2-4 of the three pair MLPs (the first layers are factorised over table rows)
EXECUTED_PAIR_MACS = 2108               # the same with every layer width rounded up to the 4-wide MFMA block
DENSE_GFLOP_PER_PAIR = 28.65            # whole forward, dense formulation: 26.45 pair MLPs + 2.05 anchors + 0.15 aff


def l1_algorithmic_bytes(B):
    """aug_shape first layer (anchor_l1*_kernel).  ALGORITHMIC HBM bytes of one launch over B frame-pairs: every weight of the
    four (N*F/64, N*F) fp32 matrices ONCE (4.096 GB, whatever the batch), the two (N*F) activation vectors of each batch item
    once, one (4*N*F/64) partial vector per batch item written.  (The kernel makes ceil(B/128) weight passes above 128
    frame-pairs per launch; those re-reads are executed traffic, not algorithmic - reported as `weight_passes_executed`.)"""
    K = N_OBJ * CH * NPOINT
    H = K // 64
    return 4 * H * K * 4 + 2 * B * K * 4 + B * 4 * H * 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` typed as is: start N fresh children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would), before this process has made any GPU call; rank 0's stdout (the JSON line) is passed through."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


class EnergyCounter:
    """The GPU's accumulated-energy counter (librocm_smi64: rsmi_dev_energy_count_get, 15.3 uJ units) read around the timed loop:
    the step is power-limited, so joules per step is what its duration follows.  None when unavailable."""

    def __init__(self, index):
        self.lib, self.index = None, index
        try:
            lib = C.CDLL("librocm_smi64.so")
            if lib.rsmi_init(C.c_uint64(0)) == 0:
                self.lib = lib
        except OSError:
            pass

    def joules(self):
        if self.lib is None:
            return None
        cnt, res, ts = C.c_uint64(), C.c_float(), C.c_uint64()
        if self.lib.rsmi_dev_energy_count_get(C.c_uint32(self.index), C.byref(cnt), C.byref(res), C.byref(ts)) != 0:
            return None
        return cnt.value * res.value * 1e-6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frame-pairs per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (no HIP-event roofline)")
    ap.add_argument("--cpu-sample", type=int, default=16, help="frame-pairs timed on the host for cpu_baseline")
    ap.add_argument("--arithmetic", choices=["f16x2", "pieces", "f32"], default="f16x2",
                    help="Shasta.arithmetic: how fp32 products are formed on the matrix cores above the batch thresholds: two fp16 pieces for "
                         "the weight stream + three bf16 pieces elsewhere (default), three bf16 pieces everywhere, or f32 MFMA kernels only")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction plumbing only, on the CPU with gloo (no GPU, no forward): for the CPU test suite")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    import shasta_amd
    from shasta_amd import hip

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

    lib = hip.load()
    B = args.batch
    torch.manual_seed(0)
    with torch.device(dev):  # random-init weights of the named architecture, created directly in HBM (4.1 GB)
        model = shasta_amd.build_simp_track(dict(
            type="Shasta", reader=None, backbone=None, neck=None,
            bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
            max_obj=N_OBJ, num_feats=NF, num_point=NPOINT)).eval()
    model.arithmetic = args.arithmetic
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    bev = torch.relu(torch.randn(B, HW, HW, CH, device=dev, generator=g))
    pbev = torch.relu(torch.randn(B, HW, HW, CH, device=dev, generator=g))

    def boxes():
        b = torch.zeros(B, N_OBJ, 11, device=dev)
        b[..., 0:2] = torch.rand(B, N_OBJ, 2, device=dev, generator=g) * 100 - 50
        b[..., 2] = torch.randn(B, N_OBJ, device=dev, generator=g)
        b[..., 3:6] = torch.rand(B, N_OBJ, 3, device=dev, generator=g) * 4 + 0.5
        b[..., 6] = (torch.rand(B, N_OBJ, device=dev, generator=g) * 2 - 1) * 3.14159265
        b[..., 7:9] = torch.randn(B, N_OBJ, 2, device=dev, generator=g)
        b[..., 9] = 0.5
        b[..., 10] = torch.rand(B, N_OBJ, device=dev, generator=g)
        return b

    det0, prev = boxes(), boxes()
    det = det0.clone()

    evs = []  # per step: (L1 start, L1 stop, pair start, pair stop)
    for _ in range(args.steps):
        four = tuple(C.c_void_p() for _ in range(4))
        for e in four:
            hip.check(lib.shasta_event_create(C.byref(e)), "event_create")
        evs.append(four)

    def step(ev=None):
        det.copy_(det0)  # forward back-projects det_boxes in place (shasta.py:270): restore the input
        return model.affinity_from_bev(bev, pbev, det, prev, l1_events=ev)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graph = None
    with torch.no_grad():
        if args.graph:
            # capture one step (all launches go through the C ABI on the capture stream; outputs are static tensors)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    m1, m2 = step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                m1, m2 = step()
        for _ in range(args.warmup):
            if graph is not None:
                graph.replay()
            else:
                m1, m2 = step()
        sync_all()
        energy = EnergyCounter(local_rank)
        e0 = energy.joules()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if graph is not None:
                graph.replay()
            else:
                m1, m2 = step(evs[i])
        sync_all()
        elapsed = time.perf_counter() - t0
        e1 = energy.joules()
    joules = (e1 - e0) if (e0 is not None and e1 is not None and e1 > e0) else None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(m1).all()) and abs(float(m1[0, 0].sum()) - 1.0) < 1e-4
    # self-check of the operating point: three frame-pairs of the timed batch recomputed one at a time (batch 1 runs the VALU
    # weight-stream kernel and the small-batch tiles of every other stage) must reproduce the batched result
    selfcheck = 0.0
    with torch.no_grad():
        for i in sorted({0, B // 2, B - 1}):
            s1, s2 = model.affinity_from_bev(bev[i:i + 1], pbev[i:i + 1], det0[i:i + 1].clone(), prev[i:i + 1])
            selfcheck = max(selfcheck, float((s1 - m1[i:i + 1]).abs().max()), float((s2 - m2[i:i + 1]).abs().max()))
    assert selfcheck <= 1e-6, "self-check failed: batched and one-at-a-time results differ by %.3e" % selfcheck

    ms = C.c_float()
    l1, pair = [], []
    if graph is not None:  # the per-step events were not recorded under graph replay: time the two kernels separately
        with torch.no_grad():
            for i in range(args.steps):
                step(evs[i])
        torch.cuda.synchronize()
    for four in evs:
        hip.check(lib.shasta_event_elapsed_ms(four[0], four[1], C.byref(ms)), "event_elapsed")
        l1.append(ms.value)
        hip.check(lib.shasta_event_elapsed_ms(four[2], four[3], C.byref(ms)), "event_elapsed")
        pair.append(ms.value)
        for e in four:
            lib.shasta_event_destroy(e)
    l1_ms, pair_ms = sum(l1) / len(l1), sum(pair) / len(pair)
    step_ms = elapsed / args.steps * 1e3
    # Kernel 1: aug_shape first layer = the 4.1 GB fp32 weight stream.
    #   B <= 32: f32 MFMA kernel, 1024 matrix-pipe cycles per 4 KB weight tile against ~1300 of HBM        -> HBM-bound
    #   pieces: B <= 64: bf16-piece kernel (each fp32 product = 6 exact bf16 piece products), 768 cycles per tile   -> HBM-bound
    #           B  > 64: the same with 128 items per weight pass, 1536 cycles per tile -> bound by the bf16 matrix pipe: priced as
    #                    EXECUTED bf16 flops (6 piece products per fp32 product) against the dense bf16 MFMA peak
    #   f16x2 (default): as pieces up to 64 frame-pairs; above, 3 fp16 piece products: 128 items per pass (768 cycles per tile,
    #                    HBM-bound) up to 128 frame-pairs, 256 items per pass (1536 cycles) above: EXECUTED f16 flops (3 per fp32
    #                    product) against the dense f16 MFMA peak
    #   --arithmetic f32 and B > 32: f32 MFMA kernel, 64 items per pass, 2048 cycles per tile -> f32 matrix pipe
    alg = l1_algorithmic_bytes(B)
    hbm_gbs = alg / (l1_ms * 1e-3) / 1e9
    K = N_OBJ * CH * NPOINT
    l1_flops = 2.0 * B * 4 * (K // 64) * K  # dense fp32 flops of the four first layers for B frame-pairs (algorithmic = executed)
    l1_tflops = l1_flops / (l1_ms * 1e-3) / 1e12
    f32_forced, f16x2 = args.arithmetic == "f32", args.arithmetic == "f16x2" and B > 64  # up to 64 frame-pairs the bf16-piece kernel serves
    if B <= 32:
        passes, nprod = 1, 1
    elif f32_forced:
        passes, nprod = -(-B // 64), 1
    elif f16x2:  # 128 items per weight pass up to 128 frame-pairs, 256 above; three fp16 piece products per fp32 product
        passes, nprod = (1 if B <= 128 else -(-B // 256)), 3
    else:        # 64 / 128 items per pass; six bf16 piece products
        passes, nprod = (1 if B <= 64 else -(-B // 128)), 6
    hbm_bound = B <= 32 or (not f32_forced and (B <= 128 if f16x2 else B <= 64))  # <= 768 matrix cycles per 4 KB weight tile
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
    roof_l1.update({"kernel": "anchor_l1_kernel (B=1) / anchor_l1_mfma_kernel (B<=32) / anchor_l1_split_kernel (B>32): "
                              "aug_shape.*.0, 4 x 2000 x 128000 fp32 weight stream",
                    "traffic": _pmc_traffic(B) if args.arithmetic == "f16x2" else None, "algorithmic_bytes_per_launch": alg, "weight_passes_executed": passes,
                    "executed_weight_bytes_per_launch": passes * 4 * (K // 64) * K * 4,
                    "algorithmic_flops_per_launch": l1_flops, "executed_flops_per_launch": nprod * l1_flops,
                    "avg_launch_ms": l1_ms, "algorithmic_hbm_gbs": hbm_gbs, "fp32_tflops": l1_tflops,
                    "share_of_step": l1_ms / step_ms})
    # Kernel 2: the pair kernel = layers 2-4 of fuse_shape / res_coeff / fuse_det for all (N+2)^2 pairs on the f32 matrix pipe.
    # Three flop counts per launch: `dense` = the reference's formulation of the three pair MLPs (first layers
    # on the concatenated pair tensor), `useful` = what is left for the pair kernel once the first layers are factorised over
    # the table rows (their GEMMs run in gemm_nt_*), `executed` = useful with every width rounded up to the MFMA block.
    # `achieved` prices the USEFUL flops against the f32 MFMA peak.
    pair_useful = 2.0 * USEFUL_PAIR_MACS * PAIRS * B
    pair_tflops = pair_useful / (pair_ms * 1e-3) / 1e12
    pair_f16 = args.arithmetic == "f16x2"  # F = 256: second layers (1792 of the 1984 MACs per pair) as three fp16 piece products each
    roof_pair = {"bound": "mfma", "achieved": pair_tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": pair_tflops / MFMA_F32_PEAK_TFLOPS, "traffic": _pmc_traffic(B, "pair") if args.arithmetic == "f16x2" else None,
                 "useful_flops_per_launch": pair_useful,
                 "dense_algorithmic_flops_per_launch": 2.0 * DENSE_PAIR_MACS * PAIRS * B,
                 "dense_equivalent_tflops": 2.0 * DENSE_PAIR_MACS * PAIRS * B / (pair_ms * 1e-3) / 1e12,
                 "avg_launch_ms": pair_ms, "share_of_step": pair_ms / step_ms}
    if pair_f16:
        roof_pair.update({
            "kernel": "pair_f16_kernel<8>: per-pair MLP tails (second layers on the f16 matrix path) + hand residual -> residual (B, 502, 502)",
            "mfma_dtype": "f16 (layer 2: three piece products per fp32 product) + f32 (layers 3-4)",
            "executed_flops_per_launch": 2.0 * (3 * 2048 + (EXECUTED_PAIR_MACS - 1792)) * PAIRS * B,
            "note": "useful fp32 flops priced against the f32 MFMA peak, the yardstick of the f32 form (pair_mfma4_kernel, --arithmetic pieces); "
                    "in this form the kernel is bound by VALU issue (cutting the activations) and LDS latency at 2 waves per SIMD, not by a matrix pipe"})
    else:
        roof_pair.update({"kernel": "pair_mfma4_kernel<256,8>: per-pair MLP tails + hand residual -> residual (B, 502, 502)",
                          "mfma_dtype": "f32", "executed_flops_per_launch": 2.0 * EXECUTED_PAIR_MACS * PAIRS * B})
    # `roofline` is the kernel with the longer average launch; the other one rides along as `roofline_second`
    roof, second = (roof_l1, roof_pair) if l1_ms >= pair_ms else (roof_pair, roof_l1)

    out = {
        "metric": "affinity frame-pairs/sec at N=M=500, F=256",
        "value": world * B * args.steps / elapsed,
        "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthetic N=M=500, F=256 (num_point=4, C=64), nf=7 affinity forward from HBM-resident "
                               "NHWC BEV features (SURVEY 8a rows 4-16)", "frame_pairs_per_step_per_gpu": B,
                   "max_obj": N_OBJ, "num_feats": NF, "num_point": NPOINT, "bev_hw": HW,
                   "parallelism": "replica x%d (frame-parallel, no data-path collective)" % world,
                   "hip_graph": bool(args.graph),
                   "arithmetic": {"f16x2": "fp32 operands in HBM, fp32 accumulation throughout; above 32 frame-pairs per step the first "
                                           "aug_shape layer (above 64 frame-pairs) and the second layers of the pair MLPs form every fp32 product from three products of "
                                           "two range-scaled fp16 pieces per operand (round to nearest; measured max error vs float64: weight "
                                           "stream 5.5e-6 against 7.0e-6 for the f32 MFMA kernel, pair stage 1.8e-6 against 2.3e-6); from 8192 "
                                           "table rows the row-embedding GEMMs and the aff layers use six products of three exact bf16 "
                                           "pieces (--arithmetic pieces / f32 select the other forms)",
                                  "pieces": "fp32 operands, fp32 accumulation throughout; above 32 frame-pairs per step the first aug_shape "
                                            "layer, and from 8192 table rows the row-embedding GEMMs and the aff layers, form every fp32 product "
                                            "from six exact bf16 piece products on the bf16 MFMA path",
                                  "f32": "fp32 operands, fp32 accumulation throughout, f32 MFMA kernels only"}[args.arithmetic]},
        "dense_equivalent_tflops": DENSE_GFLOP_PER_PAIR * 1e9 * world * B * args.steps / elapsed / 1e12,
        "selfcheck_max_abs": selfcheck,
        # rank 0's GPU over the timed loop, from the device's energy accumulator (null without librocm_smi64)
        "energy": None if joules is None else {"joules_per_step": joules / args.steps, "avg_power_w": joules / elapsed,
                                               "millijoules_per_frame_pair": joules / args.steps / B * 1e3},
        "roofline": roof,
        "roofline_second": second,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, args.cpu_sample)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _pmc_traffic(B, kernel="l1"):
    """HBM bytes per launch of one of the two heaviest kernels of the DEFAULT arithmetic from the committed rocprofv3 PMC passes
    (profiles/), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE doubled on gfx950); None when no profile for this batch size
    exists."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(("batch_%d" if kernel == "l1" else "pair_batch_%d") % B)
    except (OSError, ValueError):
        return None


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
            "sample": "%d frame-pairs at N=M=500,F=256, batch 1, torch-CPU fp32 oracle, %.1f s on a host with %d logical CPUs; the "
                      "fastest of {8,16,32,64,%d} torch threads was used: %d" % (sample, dt, host, all_threads, best_n)}


def dry_run(args, rank, world):
    """CPU plumbing check of the multi-rank launch (no GPU, no forward): rendezvous over gloo, the barrier and the MAX
    reduction over ranks that bracket the timed region, and the rank-0 JSON line with `value` null."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="gloo")
        dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    assert int(t.item()) == world and args.gpus == world
    if rank == 0:
        print(json.dumps({"metric": "affinity frame-pairs/sec at N=M=500, F=256", "value": None, "unit": "frame-pairs/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "max_over_ranks": t.item()}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
