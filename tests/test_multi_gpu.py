"""Multi-rank paths on REAL GPUs over RCCL (rows e / BASELINE configs 4-5).  Every other multi-rank test of the suite runs on
gloo / CPU; these run whenever the box has at least two GPUs and skip on a 1-GPU lease, so the first bigger box exercises them:
  (i)  bench.py --gpus N typed as is (N = 2, 4, 8 - whatever the box has): N replicas, one JSON line, n_gpus == N, throughput about N
       times one replica's;
  (ii) the data-parallel training step of tools/nusc_shasta/train.py:155-156,198-218 on N ranks: SyncBatchNorm statistics over
       both ranks, the rank-B factor all-gather inside the HIP backward, allreduce_gradients for the rest - equal to ONE process
       over the whole batch (the same reference computation as tests/test_training_ddp.py uses on gloo)."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_training_ddp import _ddp_case, _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need(n):
    if torch.cuda.device_count() < n:
        pytest.skip("needs %d GPUs, this box has %d" % (n, torch.cuda.device_count()))


def _bench(gpus):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "6", "--warmup", "2", "--batch", "64",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return lines[0]


@pytest.mark.parametrize("gpus", [2, 4, 8])
def test_bench_replicas_on_several_gpus(gpus):
    _need(gpus)
    one, many = _bench(1), _bench(gpus)
    assert one["n_gpus"] == 1 and many["n_gpus"] == gpus and many["scaling"] == "weak"
    assert many["config"]["frame_pairs_per_step_per_gpu"] == 64 and many["selfcheck_max_abs"] <= 1e-6
    # replicas, no data-path collective: the whole-job rate is the sum (MAX over ranks of the time; allow clock / box spread)
    assert 0.8 * gpus * one["value"] <= many["value"] <= 1.15 * gpus * one["value"], (one["value"], many["value"])


def _nccl_train_worker(rank, world, port, q, backend="nccl"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    gpu = rank if backend == "nccl" else 0  # gloo: the ranks share GPU 0 (1-GPU lease)
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from shasta_amd import training
    from shasta_amd.sync_bn import convert_syncbn_model
    model, bev, pbev, det, prev, gt = _ddp_case(2 * world)
    model = model.to(dev)
    convert_syncbn_model(model)
    model.train()
    sl = slice(rank * 2, rank * 2 + 2)
    ex = dict(det_boxes=det[sl].to(dev).contiguous(), prev_det_boxes=prev[sl].to(dev).contiguous(), bev_map=bev[sl].to(dev),
              prev_bev_map=pbev[sl].to(dev))
    m1, m2, _ = model(ex, train_mode=True)       # shared_conv in train() mode (SyncBN) -> affinity_train (HIP forward + backward)
    training.affinity_loss(m1, m2, gt[sl].to(dev)).backward()
    flagged = [bool(getattr(model.aug_shape[i][0].weight, "_shasta_grad_is_global", False)) for i in range(4)]
    training.allreduce_gradients([p for p in model.parameters() if p.grad is not None])
    out = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
    out["__running_mean"] = model.shared_conv[1].running_mean.cpu().numpy()
    out["__flagged"] = flagged
    q.put((rank, out))
    dist.destroy_process_group()


def test_two_process_train_step_on_one_gpu_equals_the_single_process_step():
    """The data-parallel step on a 1-GPU lease: two processes share GPU 0 and exchange over gloo - the hand-written train-mode
    shared_conv with a synchronised BatchNorm (statistics all-gathered in the forward, sums all-reduced in the backward), the factor
    all-gather inside the HIP backward, allreduce_gradients: everything of the multi-rank step except RCCL itself."""
    _need(1)
    _train_step_equals_single(2, "gloo")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_train_step_equals_the_single_process_step(world):
    _need(world)
    _train_step_equals_single(world, "nccl")


def _train_step_equals_single(world, backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_train_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert all(res[r]["__flagged"] == [True] * 4 for r in range(world))  # the factor exchange ran, not the dense all-reduce
    # one process over the whole batch: BatchNorm statistics of all four maps, mean of the two per-rank losses (CPU autograd of the oracle)
    from oracle import shasta_oracle as O
    model, bev, pbev, det, prev, gt = _ddp_case(2 * world)
    model.train()
    w = dict(model.named_parameters())
    w.update(dict(model.named_buffers()))
    a = model.shared_conv(bev).permute(0, 2, 3, 1).contiguous()
    b = model.shared_conv(pbev).permute(0, 2, 3, 1).contiguous()
    loss = 0
    for r in range(world):
        sl = slice(2 * r, 2 * r + 2)
        m1, m2 = O.forward_from_bev(w, a[sl], b[sl], det[sl].clone(), prev[sl].clone(), 3, 4, grad=True)
        loss = loss + O.affinity_loss(m1, m2, gt[sl]) / world
    loss.backward()
    want = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert len(want) > 60 and set(want) <= set(res[0])
    for n, g in want.items():
        for r in range(world):
            scale = max(float(g.abs().max()), 1e-8)
            assert float((torch.from_numpy(res[r][n]) - g).abs().max()) <= 2e-3 * scale + 1e-7, (n, r)  # HIP backward vs autograd: 2e-3 as in test_training.py
    assert torch.allclose(torch.from_numpy(res[0]["__running_mean"]), model.shared_conv[1].running_mean, rtol=1e-4, atol=1e-6)
    assert all((res[0]["__running_mean"] == res[r]["__running_mean"]).all() for r in range(1, world))


def _nccl_chain_worker(rank, world, port, root, work, backend="nccl"):
    """BASELINE config 4 on real GPUs: this rank's scenes through the HIP chain on its own GPU, per-class results gathered on rank 0
    over RCCL (replica.gather_decoded -> dist.gather_object), rank 0 merges, tracks and writes the reference CLIs' files."""
    from shasta_amd import pipeline, scenes
    from tests.test_pipeline import _models
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    gpu = rank if backend == "nccl" else 0  # gloo: the ranks share GPU 0 (1-GPU lease)
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    paths = scenes.split_paths(root)
    meta = json.load(open(paths["frames_meta_path"]))["frames"]
    sc, cur = [], None
    for fr in meta:
        if fr["first"]:
            cur = ("scene-%04d" % len(sc), [])
            sc.append(cur)
        cur[1].append(fr["token"])
    models = {n: m.to(dev) for n, m in _models().items()}
    res = pipeline.run_split(models, paths, sc, scenes.TokenBev(), dev, work_dir=work if rank == 0 else None, batch_pairs=3, rank=rank,
                             world=world)
    assert (res is None) == (rank != 0)
    dist.destroy_process_group()


def test_two_process_chain_on_one_gpu_equals_single_rank(tmp_path):
    """The same comparison on a 1-GPU lease: two processes share GPU 0 and exchange over gloo - everything of the multi-rank device
    chain (sharding, HIP forward + decode per rank, gather, merge, whole-scene tracker on rank 0, the files) except RCCL itself."""
    _need(1)
    _chain_equals_single_rank(2, tmp_path, "gloo")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_scene_sharded_chain_equals_single_rank(world, tmp_path):
    """pipeline.run_split(rank, world) over `nccl` - scenes sharded over the GPUs (uneven: 11 scenes of 2 to 5
    frames), decoded results gathered on rank 0, whose merged_cp_val.json / tracking_result.json are byte-equal to the files of the
    one-rank run on one GPU (tools/nusc_shasta/eval.py:175-181 is the post-pass the gather feeds, pub_test.py:88-162 the tracker)."""
    _need(world)
    _chain_equals_single_rank(world, tmp_path, "nccl")


def _chain_equals_single_rank(world, tmp_path, backend):
    from shasta_amd import pipeline, scenes
    from tests.test_pipeline import _models
    lengths = [3, 5, 2, 4, 3, 2, 5, 3, 4, 2, 3]
    root = str(tmp_path / "split")
    paths, sc = scenes.write_synthetic_split(root, n_scenes=len(lengths), frames_per_scene=lengths, seed=7)
    many, one = str(tmp_path / "many"), str(tmp_path / "one")
    mp.spawn(_nccl_chain_worker, args=(world, _free_port(), root, many, backend), nprocs=world, join=True)
    dev = torch.device("cuda", 0)
    models = {n: m.to(dev) for n, m in _models().items()}
    pipeline.run_split(models, paths, sc, scenes.TokenBev(), dev, work_dir=one, batch_pairs=3)
    for f in ("merged_cp_val.json", "tracking_result.json", "car/cp_val.json"):
        with open(os.path.join(many, f), "rb") as a, open(os.path.join(one, f), "rb") as b:
            assert a.read() == b.read(), f
