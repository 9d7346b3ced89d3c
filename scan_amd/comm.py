"""Process-group helpers of the data-parallel path (one process per GPU, torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" in the CPU tests).  Mirrors what the reference's fcos_core/utils/comm.py and engine code do around
the hot path:

* ``reduce_loss_dict``        engine/trainer.py:76-98 -- loss scalars stacked in sorted-key order, ``dist.reduce`` to rank 0,
                              rank 0 divides by the world size (logging only),
* ``gather_detections``       utils/comm.py:48-88 + engine/inference.py:40-58 -- every rank's per-image detections end up
                              on rank 0 keyed by image id.  The reference pickles python objects into
                              padded ByteTensors; here the payload is fixed-layout float rows [image id, x1, y1, x2, y2,
                              score, label], padded to the largest rank and moved by ONE all_gather of a tensor -- no
                              pickle, no per-rank size exchange beyond one int,
* ``images_per_gpu``          data/build.py:181-188,254-261 -- the global IMS_PER_BATCH split.

The gradient all-reduce and the paradigm all-reduce (ii) live next to the data they move:
engine.Trainer._allreduce_async and modeling/condgraph.py update_prototype_nx1_rnn.
"""
import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    """reference utils/comm.py:33-45: barrier when there is more than one rank."""
    if get_world_size() > 1:
        dist.barrier()


def images_per_gpu(ims_per_batch, world_size=None):
    """reference data/build.py:181-188: SOLVER.IMS_PER_BATCH is GLOBAL and must divide by the number of GPUs; each of the
    source and the target loader hands every rank IMS_PER_BATCH / world images."""
    world_size = get_world_size() if world_size is None else world_size
    if ims_per_batch % world_size != 0:
        raise ValueError("SOLVER.IMS_PER_BATCH (%d) must be divisible by the number of GPUs (%d) used."
                         % (ims_per_batch, world_size))
    return ims_per_batch // world_size


def reduce_loss_dict(loss_dict):
    """Averaged loss dict on rank 0 (other ranks get their partial sums back, like the reference); one reduce of one
    small tensor.  World size 1: the dict itself."""
    world = get_world_size()
    if world < 2:
        return loss_dict
    with torch.no_grad():
        names = sorted(loss_dict.keys())
        vals = torch.stack([loss_dict[k].detach().float().reshape(()) for k in names], 0)
        dist.reduce(vals, dst=0)
        if get_rank() == 0:
            vals /= world
        return {k: v for k, v in zip(names, vals)}


def gather_detections(results, image_ids, device=None):
    """results: list of (boxes [k,4], scores [k], labels [k]) of this rank's images, image_ids their dataset indices.
    Returns on rank 0 a dict {image id: (boxes, scores, labels)} (CPU tensors) over ALL ranks, None elsewhere --
    what _accumulate_predictions_from_multiple_gpus hands to the evaluation (engine/inference.py:40-58)."""
    world = get_world_size()
    rows = []
    for iid, (b, s, l) in zip(image_ids, results):
        k = b.shape[0]
        r = torch.empty((k, 7), dtype=torch.float32, device=b.device)
        r[:, 0] = float(iid)
        r[:, 1:5] = b
        r[:, 5] = s
        r[:, 6] = l.to(torch.float32)
        rows.append(r)
    dev = device if device is not None else (results[0][0].device if results else torch.device("cpu"))
    mine = torch.cat(rows, 0) if rows else torch.empty((0, 7), dtype=torch.float32, device=dev)
    # images without a detection still have to show up in the result: their ids travel in a second small tensor
    ids = torch.tensor(list(image_ids), dtype=torch.int64, device=dev)
    if world == 1:
        packs, id_packs = [mine], [ids]
    else:
        cnt = torch.tensor([mine.shape[0], ids.shape[0]], dtype=torch.int64, device=dev)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        n_max = max(int(c[0]) for c in cnts)
        i_max = max(int(c[1]) for c in cnts)
        pad = torch.zeros((n_max, 7), dtype=torch.float32, device=dev)
        pad[:mine.shape[0]] = mine
        ipad = torch.full((i_max,), -1, dtype=torch.int64, device=dev)
        ipad[:ids.shape[0]] = ids
        out = [torch.empty_like(pad) for _ in range(world)]
        iout = [torch.empty_like(ipad) for _ in range(world)]
        dist.all_gather(out, pad)
        dist.all_gather(iout, ipad)
        packs = [o[:int(c[0])] for o, c in zip(out, cnts)]
        id_packs = [o[:int(c[1])] for o, c in zip(iout, cnts)]
    if not is_main_process():
        return None
    merged = {}
    for p, ip in zip(packs, id_packs):
        p, ip = p.cpu(), ip.cpu()
        for iid in ip.tolist():
            sel = p[:, 0] == float(iid)
            merged[int(iid)] = (p[sel, 1:5].contiguous(), p[sel, 5].contiguous(), p[sel, 6].to(torch.int64))
    return merged


# ----------------------------------------------------------------------------- rank placement on the host
# One process per GPU: each rank's Python thread enqueues ~850 launches per step and must not share cores with its seven
# neighbours or sit on the other socket's memory.  torch.distributed.run places nothing; the reference leaves it to the user
# (tools/train_net_da.py:421-515 starts under torch.distributed.launch with no binding).  The GPU -> NUMA node -> CPU list map
# is read from sysfs (kfd topology node -> drm render minor -> PCI device's numa_node / local_cpulist); nothing here opens the GPU.
def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_map(sysfs_root="/sys"):
    """[{gpu, render_minor, numa_node, cpus}] for every kfd GPU node, in kfd order (= HIP device order without
    HIP_VISIBLE_DEVICES); [] when sysfs has no kfd topology."""
    import os
    base = os.path.join(sysfs_root, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted(os.listdir(base), key=lambda n: int(n))
    except (OSError, ValueError):
        return []
    out = []
    for node in nodes:
        try:
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(l.split(None, 1) for l in f.read().splitlines() if " " in l)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) <= 0:
            continue  # a CPU node
        minor = int(props.get("drm_render_minor", "-1"))
        dev = os.path.join(sysfs_root, "class", "drm", "renderD%d" % minor, "device")
        numa, cpus = -1, []
        try:
            with open(os.path.join(dev, "numa_node")) as f:
                numa = int(f.read().strip())
            with open(os.path.join(dev, "local_cpulist")) as f:
                cpus = _parse_cpulist(f.read())
        except (OSError, ValueError):
            pass
        out.append({"gpu": len(out), "render_minor": minor, "numa_node": numa, "cpus": cpus})
    return out


def rank_cpu_set(local_rank, local_world, sysfs_root="/sys", allowed=None):
    """The CPUs rank ``local_rank`` of ``local_world`` should run on: the CPUs local to its GPU (restricted to ``allowed``, the
    process's current affinity mask), split evenly among the ranks whose GPUs share that NUMA node, in rank order.  None when
    the topology is unknown or the share would be empty (then nothing is bound)."""
    gpus = gpu_numa_map(sysfs_root)
    if local_rank >= len(gpus) or not gpus[local_rank]["cpus"]:
        return None
    me = gpus[local_rank]
    cpus = [c for c in me["cpus"] if allowed is None or c in allowed]
    peers = [g["gpu"] for g in gpus[:local_world] if g["numa_node"] == me["numa_node"] and g["cpus"] == me["cpus"]]
    if not cpus or me["gpu"] not in peers:
        return None
    k, n = peers.index(me["gpu"]), len(peers)
    share = cpus[k * len(cpus) // n:(k + 1) * len(cpus) // n]
    return {"numa_node": me["numa_node"], "cpus": share} if share else None


def bind_rank(local_rank, local_world, sysfs_root="/sys"):
    """sched_setaffinity of this process to rank_cpu_set(...) (memory then follows by first touch).  SCAN_RANK_BINDING=0
    switches it off.  Returns what was applied (for the bench line) or None."""
    import os
    if os.environ.get("SCAN_RANK_BINDING", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = os.sched_getaffinity(0)
        sel = rank_cpu_set(local_rank, local_world, sysfs_root, allowed)
        if sel is None:
            return None
        os.sched_setaffinity(0, sel["cpus"])
        return {"numa_node": sel["numa_node"], "n_cpus": len(sel["cpus"]), "first_cpu": sel["cpus"][0], "last_cpu": sel["cpus"][-1]}
    except OSError:
        return None
