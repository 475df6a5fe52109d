"""Multi-GPU inference = frame-parallel replicas: one process per GPU, each with a full weight copy,
scenes sharded across ranks, NO collective in the forward.  The only exchange is at the end: every rank's decoded
per-token results are gathered on rank 0, which runs the `dead` post-pass of tools/nusc_shasta/eval.py:175-181 (scene
sharding keeps frames t-1 and t on the same rank, which that pass needs).  Backend: "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests."""
import torch.distributed as dist


def shard_scenes(scenes, rank, world):
    """Deterministic, balanced by frame count: scenes is a list of (scene_id, n_frames); longest-first greedy packing.
    Returns the scene ids of `rank` in their original order."""
    order = sorted(range(len(scenes)), key=lambda i: (-scenes[i][1], i))
    load = [0] * world
    owner = {}
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += scenes[i][1]
    return [scenes[i][0] for i in range(len(scenes)) if owner[i] == rank]


def gather_decoded(decoder, dst=0, group=None):
    """Merge the AffinityDecoder state of every rank on `dst` and finalize there; other ranks return None."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return decoder.finalize()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    payload = (decoder.results, decoder.dead_tracker)
    gathered = [None] * world if rank == dst else None
    dist.gather_object(payload, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    for r, (res, dead) in enumerate(gathered):
        if r == dst:
            continue
        for token, annos in res.items():
            if token in decoder.results:
                raise RuntimeError("token %s decoded by two ranks (scene sharding must keep a token on one rank)" % token)
            decoder.results[token] = annos
        for token, info in dead.items():
            mine = decoder.dead_tracker.setdefault(token, {"dead_idx": [], "keep_idx": []})
            mine["dead_idx"].extend(info["dead_idx"])
            if info["keep_idx"]:
                mine["keep_idx"] = info["keep_idx"]
    return decoder.finalize()
