#!/usr/bin/env python3
"""bench.py -- affinity frame-pairs/s at N=M=500, F=256 (BASELINE.json metric) on N MI355X of one node.

A "step" is one pass of the affinity hot path (BEV gather -> anchors -> pair residual ->
aff + softmaxes) over one batch of `--batch` synthetic frame-pairs whose inputs (NHWC BEV feature maps, box tables,
weights) are already resident in HBM.  Frame-pairs are independent, so with N GPUs every rank runs its own replica on
its own batch (weak scaling, no data-path collective); the only collectives are the barrier and the MAX over ranks of
the elapsed time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the definition of every field).
"""
import argparse
import ctypes as C
import json
import os
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


def l1_algorithmic_bytes(B):
    """aug_shape first layer (anchor_l1*_kernel).  Algorithmic HBM bytes of one launch: every weight of the four
    (N*F/64, N*F) fp32 matrices once per weight pass, the two (N*F) activation vectors of each of the B batch items
    once, and one (4*N*F/64) partial vector per batch item written.  One weight pass serves up to
    128 frame-pairs - from there the pass is bound by the matrix pipe, not by HBM, so larger batches take ceil(B/128)
    passes by design."""
    K = N_OBJ * CH * NPOINT
    H = K // 64
    passes = max(1, -(-B // 128))
    return passes * 4 * H * K * 4 + 2 * B * K * 4 + B * 4 * H * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frame-pairs per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (no HIP-event roofline)")
    ap.add_argument("--cpu-sample", type=int, default=16, help="frame-pairs timed on the host for cpu_baseline")
    args = ap.parse_args()

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
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)
    assert args.gpus == world, "--gpus must equal the number of launched ranks"

    lib = hip.load()
    B = args.batch
    torch.manual_seed(0)
    with torch.device(dev):  # random-init weights of the named architecture, created directly in HBM (4.1 GB)
        model = shasta_amd.build_simp_track(dict(
            type="Shasta", reader=None, backbone=None, neck=None,
            bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
            max_obj=N_OBJ, num_feats=NF, num_point=NPOINT)).eval()
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
        t0 = time.perf_counter()
        for i in range(args.steps):
            if graph is not None:
                graph.replay()
            else:
                m1, m2 = step(evs[i])
        sync_all()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(m1).all()) and abs(float(m1[0, 0].sum()) - 1.0) < 1e-4

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
    # Kernel 1: aug_shape first layer = the 4.1 GB fp32 weight stream, read once per pass of up to 128 frame-pairs.
    #   B <= 32: f32 MFMA kernel, 1024 matrix-pipe cycles per 4 KB weight tile against ~1300 of HBM        -> HBM-bound
    #   B <= 64: bf16-piece kernel (each fp32 product = 6 exact bf16 piece products), 768 cycles per tile   -> HBM-bound
    #   B  > 64: the same with 128 items per pass, 1536 cycles per tile -> bound by the bf16 matrix pipe; its peak for
    #            fp32-equivalent flops is the dense bf16 peak / 6 piece products
    #   SHASTA_L1_F32=1 and B > 32: f32 MFMA kernel, 64 items per pass, 2048 cycles per tile -> f32 matrix pipe
    alg = l1_algorithmic_bytes(B)
    hbm_gbs = alg / (l1_ms * 1e-3) / 1e9
    K = N_OBJ * CH * NPOINT
    l1_flops = 2.0 * B * 4 * (K // 64) * K  # dense fp32 flops of the four first layers for B frame-pairs
    l1_tflops = l1_flops / (l1_ms * 1e-3) / 1e12
    f32_forced = bool(os.environ.get("SHASTA_L1_F32")) or bool(os.environ.get("SHASTA_L1_VALU"))
    if B <= 32 or (B <= 64 and not f32_forced):
        roof_l1 = {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS}
    elif f32_forced:
        roof_l1 = {"bound": "mfma", "achieved": l1_tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": l1_tflops / MFMA_F32_PEAK_TFLOPS}
    else:
        peak = MFMA_BF16_PEAK_TFLOPS / 6.0
        roof_l1 = {"bound": "mfma", "achieved": l1_tflops, "peak": peak, "unit": "TFLOP/s", "frac": l1_tflops / peak,
                   "peak_note": "fp32-equivalent: dense bf16 MFMA peak 2500 TFLOP/s / 6 piece products per fp32 product"}
    roof_l1.update({"kernel": "anchor_l1_kernel (B=1) / anchor_l1_mfma_kernel (B<=32) / anchor_l1_split_kernel (B>32): "
                              "aug_shape.*.0, 4 x 2000 x 128000 fp32 weight stream",
                    "traffic": _pmc_traffic(B), "algorithmic_bytes_per_launch": alg, "algorithmic_flops_per_launch": l1_flops,
                    "avg_launch_ms": l1_ms, "hbm_gbs": hbm_gbs, "fp32_equivalent_tflops": l1_tflops,
                    "share_of_step": l1_ms / step_ms})
    # Kernel 2: the pair kernel (layers 2-4 of fuse_shape / res_coeff / fuse_det for all (N+2)^2 pairs on the f32 matrix
    # pipe): 1984 useful multiply-adds per pair; everything it reads is L2-resident.
    pair_flops = 2.0 * 1984 * (N_OBJ + 2) ** 2 * B
    pair_tflops = pair_flops / (pair_ms * 1e-3) / 1e12
    roof_pair = {"bound": "mfma", "achieved": pair_tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": pair_tflops / MFMA_F32_PEAK_TFLOPS, "traffic": _pmc_traffic(B, "pair"),
                 "kernel": "pair_mfma4_kernel<256,8>: per-pair MLP tails + hand residual -> residual (B, 502, 502)",
                 "algorithmic_flops_per_launch": pair_flops, "avg_launch_ms": pair_ms, "share_of_step": pair_ms / step_ms}
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
                   "arithmetic": "fp32 operands, fp32 accumulation throughout; above 32 frame-pairs per step the first aug_shape "
                                 "layer (and from 8192 table rows the row-embedding GEMMs) form every fp32 product from six exact "
                                 "bf16 piece products on the bf16 MFMA path (error below the fp32 FMA's rounding; SHASTA_L1_F32=1 / "
                                 "SHASTA_GEMM_F32=1 select the f32 MFMA kernels)"},
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
    """HBM bytes per launch of one of the two heaviest kernels from the committed rocprofv3 PMC passes (profiles/),
    corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE doubled on gfx950); None when no profile for this batch size
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
    return {"value": sample / dt, "unit": "frame-pairs/s", "cores": best_n, "kind": "port",
            "sample": "%d frame-pairs at N=M=500,F=256, batch 1, torch-CPU fp32 oracle, %.1f s; best of {8,16,32,64,%d} "
                      "threads = %d" % (sample, dt, all_threads, best_n)}


if __name__ == "__main__":
    main()
