"""CPU test of the N > 1 path: two processes, gloo.  Covers what the data-parallel step adds to the
single-GPU path: the flat-gradient all-reduce (average) and the paradigm all-reduce that keeps the
prototype buffer identical on every rank."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scan_amd import synth
    from scan_amd.modeling import condgraph
    torch.manual_seed(0)
    mh = condgraph.GRAPHModule(256, 9)
    mh.load_state_dict(synth.middle_head_state_dict(9))
    g = torch.Generator().manual_seed(100 + rank)  # different shard per rank
    pbs = []
    for it in range(4):
        pb = torch.randn(9, 256, generator=g)
        if rank == 0 and it == 1:
            pb[2] = 0  # class 2 only seen by rank 1 this iteration
        if it == 2:
            pb[5] = 0  # class 5 seen by nobody
        pbs.append(pb)
        mh.update_prototype_nx1_rnn(pb)
    # flat gradient average
    flat = torch.full((1000,), float(rank + 1))
    flat.div_(world)
    dist.all_reduce(flat)
    # numpy arrays are pickled by value; torch tensors would travel as shared-memory handles that die with this process
    q.put((rank, mh.prototype.clone().numpy(), flat.clone().numpy(), torch.stack(pbs).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    res = [(r, torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c)) for r, a, b, c in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, proto0, flat0, pb0), (_, proto1, flat1, pb1) = res
    assert torch.equal(proto0, proto1), "paradigm buffers diverged across ranks"
    assert torch.allclose(flat0, torch.full((1000,), 1.5)) and torch.equal(flat0, flat1)
    # single-process replay with the rank-averaged class means (mean over the ranks that saw the class)
    sys.path.insert(0, ROOT)
    from scan_amd import synth
    from scan_amd.modeling import condgraph
    mh = condgraph.GRAPHModule(256, 9)
    mh.load_state_dict(synth.middle_head_state_dict(9))
    for it in range(4):
        a, b = pb0[it], pb1[it]
        ea, eb = a.sum(-1).bool().float()[:, None], b.sum(-1).bool().float()[:, None]
        mh.update_prototype_nx1_rnn((a * ea + b * eb) / (ea + eb).clamp(min=1))
    assert torch.allclose(mh.prototype, proto0, rtol=1e-6, atol=1e-7)


def _bucket_worker(rank, world, port, q):
    """engine.Trainer's gradient buckets on host tensors: rank 0 marks them ready in backward order, rank 1 in a
    scrambled order with two marks missing (hooks that never fired).  The collective sequence must be the canonical
    one on both ranks and the arena the rank average."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scan_amd import engine
    model = engine.build_model(9, device="cpu")
    trainer = engine.Trainer(model, distributed=True)
    names = [b[0] for b in trainer._buckets()]
    trainer.grad_arena.copy_(torch.arange(trainer.grad_arena.numel(), dtype=torch.float32) % 97 + 100.0 * rank)
    trainer._begin_buckets()
    order = names if rank == 0 else [names[3], names[0], names[2], names[4], names[1]]  # two never reported
    for n in order:
        trainer._bucket_ready(n)
    trainer._flush_buckets()
    # loss-key invariance: a rank without sampled target nodes still reports consistency_loss_gt
    from scan_amd import comm
    ld = {"a_loss": torch.tensor(1.0 + rank), "consistency_loss_gt": torch.tensor(0.5 if rank == 0 else 0.0)}
    red = comm.reduce_loss_dict(ld)
    q.put((rank, names, list(trainer.collective_log), trainer.grad_arena[:4096].clone().numpy(),
           trainer.grad_arena.double().sum().item(), {k: float(v) for k, v in red.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_buckets_keep_one_collective_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, names, log0, head0, sum0, red0), (_, _, log1, head1, sum1, red1) = res
    # FCOS head | discriminators | middle head | backbone conv5 | conv4 + FPN | rest (conv3 + biases)
    assert names == ["fcos", "dis", "middle_head", "backbone:c4", "backbone:c3", "backbone:rest"]
    assert log0 == log1 and len(log0) >= 7, (log0, log1)
    # every element reduced exactly once: ranges are disjoint and cover the arena
    sys.path.insert(0, ROOT)
    from scan_amd import engine
    tr = engine.Trainer(engine.build_model(9, device="cpu"))
    cover = engine._merge_ranges(log0)
    assert cover == [(0, tr.grad_arena.numel())] and sum(b - a for a, b in log0) == tr.grad_arena.numel()
    n = tr.grad_arena.numel()
    expect = (torch.arange(n, dtype=torch.float32) % 97 + 50.0)
    assert torch.equal(torch.from_numpy(head0), expect[:4096]) and torch.equal(torch.from_numpy(head1), expect[:4096])
    assert abs(sum0 - expect.double().sum().item()) < 1e-3 * n and sum0 == sum1
    # conv5 weights are one contiguous 3 x 512 x 512 x 9 range; the c3 bucket is conv4 + the FPN weights (two ranges)
    bb = tr.arena_range["backbone"][0]
    g = tr.groups["backbone"]
    c5 = (bb + g.offset["body.features.24.weight"][0], bb + g.offset["body.features.28.weight"][0] + 512 * 512 * 9)
    assert c5 in log0 and log0.index(c5) == 3
    assert red0["a_loss"] == 1.5 and red0["consistency_loss_gt"] == 0.25


def _policy_worker(rank, world, port, q):
    """Trainer.dp_policy "coarse" and "tail" on host tensors: the hooks fire with the fine-grained names (rank 1 scrambled, one
    missing), the collective sequence is the policy's and the arena the rank average."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scan_amd import engine
    model = engine.build_model(9, device="cpu")
    out = {}
    for pol in ("coarse", "tail", "overlap"):
        trainer = engine.Trainer(model, distributed=True, dp_policy=pol)
        fine = [b[0] for b in trainer._fine_buckets()]
        n = trainer.grad_arena.numel()
        trainer.grad_arena.copy_(torch.arange(n, dtype=torch.float32) % 97 + 100.0 * rank)
        trainer._begin_buckets()
        order = fine if rank == 0 else [fine[3], fine[1], fine[0], fine[4], fine[2]]
        by_hooks = []
        for nm in order:
            trainer._bucket_ready(nm)
            by_hooks.append(len(trainer.collective_log))
        trainer._flush_buckets()
        dis_end = max(trainer.arena_range[k][1] for k in trainer.groups if k.startswith("dis_"))
        out[pol] = ([b[0] for b in trainer._buckets()], list(trainer.collective_log), by_hooks, dis_end, n,
                    trainer.grad_arena[:2048].clone().numpy(), trainer.grad_arena.double().sum().item())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_policies():
    """overlap / coarse / tail (engine.Trainer.dp_policy, SCAN_DP_POLICY): same reduced arena, different collective sequences,
    identical on both ranks whatever order the hooks fire in."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_policy_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for pol in ("coarse", "tail", "overlap"):
        names0, log0, hooks0, dis_end, n, head0, sum0 = res[0][pol]
        names1, log1, hooks1, _, _, head1, sum1 = res[1][pol]
        assert names0 == names1 and log0 == log1, (pol, log0, log1)
        expect = torch.arange(n, dtype=torch.float32) % 97 + 50.0
        assert torch.equal(torch.from_numpy(head0), expect[:2048]) and torch.equal(torch.from_numpy(head1), expect[:2048])
        assert sum0 == sum1 and abs(sum0 - expect.double().sum().item()) < 1e-3 * n
        if pol == "coarse":
            assert names0 == ["heads", "rest"] and log0 == [(0, dis_end), (dis_end, n)]
            # rank 0 (backward order fcos, dis, ..., backbone:rest): the heads range goes out with the second hook, the rest with
            # the last one (the conv3_1 node's hook = the end of the backward)
            assert hooks0 == [0, 1, 1, 1, 1, 2]
            # rank 1 (c4, dis, fcos, c3, middle head; backbone:rest never reported): 'dis' alone is not enough, 'fcos' completes
            # the heads; the rest waits for the flush
            assert hooks1 == [0, 0, 1, 1, 1]
        elif pol == "tail":
            # one range; rank 0 issues it from the last hook of the backward, rank 1 (a hook missing) at the flush
            assert names0 == ["all"] and log0 == [(0, n)] and hooks0 == [0, 0, 0, 0, 0, 1] and hooks1[-1] == 0
        else:
            assert len(log0) >= 7


# ----------------------------------------------------------------------------------------------------------------------
# Eight ranks (BASELINE.json configs[2] / [4]: one process per GPU of an 8-GPU node).  No 8-GPU box is available to the
# build, so everything of the N = 8 path that does not need a GPU runs here on gloo: the launcher, the global batch split,
# the gradient buckets with hooks firing in a different order (or not at all) on every rank, the paradigm all-reduce, the
# loss reduce and the detection gather (reference tools/train_net_da.py:421-515, utils/comm.py:48-117, data/build.py:181-188).
# ----------------------------------------------------------------------------------------------------------------------
def _eight_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import random
    from scan_amd import comm, engine, synth
    from scan_amd.modeling import condgraph
    out = {"rank": rank}
    # (a) SOLVER.IMS_PER_BATCH is global (data/build.py:181-188): 16 -> 2 per rank, 64 -> 8, 12 is refused
    out["split"] = (comm.images_per_gpu(16), comm.images_per_gpu(64))
    try:
        comm.images_per_gpu(12)
        out["split_err"] = None
    except ValueError as e:
        out["split_err"] = str(e)
    # (b) gradient buckets: every rank reports them in its own scrambled order, rank r leaves bucket r % 6 unreported
    model = engine.build_model(9, device="cpu")
    trainer = engine.Trainer(model, distributed=True)
    names = [b[0] for b in trainer._buckets()]
    n = trainer.grad_arena.numel()
    trainer.grad_arena.copy_(torch.arange(n, dtype=torch.float32) % 89 + 8.0 * rank)
    order = list(names)
    random.Random(1000 + rank).shuffle(order)
    if rank != 0:
        order.remove(names[rank % len(names)])
    for rep in range(2):  # two "iterations": the state machine resets cleanly
        if rep == 1:
            trainer.grad_arena.copy_(torch.arange(n, dtype=torch.float32) % 89 + 8.0 * rank)
            del trainer.collective_log[:]
        trainer._begin_buckets()
        for nm in order:
            trainer._bucket_ready(nm)
        issued_by_hooks = len(trainer.collective_log)
        trainer._flush_buckets()
    out.update(names=names, log=list(trainer.collective_log), issued_by_hooks=issued_by_hooks,
               arena_sum=trainer.grad_arena.double().sum().item(), arena_head=trainer.grad_arena[:2048].clone().numpy(),
               arena_tail=trainer.grad_arena[-2048:].clone().numpy())
    # (c) paradigm all-reduce: rank r sees class c only when (r + c) % 3 != 0; class 7 is seen by nobody
    mh = condgraph.GRAPHModule(256, 9)
    mh.load_state_dict(synth.middle_head_state_dict(9))
    g = torch.Generator().manual_seed(500 + rank)
    pbs = []
    for it in range(3):
        pb = torch.randn(9, 256, generator=g)
        for c in range(9):
            if (rank + c + it) % 3 == 0 or c == 7:
                pb[c] = 0
        pbs.append(pb)
        mh.update_prototype_nx1_rnn(pb)
    out.update(proto=mh.prototype.clone().numpy(), pbs=torch.stack(pbs).numpy())
    # (d) loss reduce (engine/trainer.py:76-98) and detection gather (utils/comm.py:48-88); rank 5 has no detection at all
    red = comm.reduce_loss_dict({"loss_cls_gs": torch.tensor(float(rank)), "zero_gt": torch.tensor(0.0)})
    out["red"] = {k: float(v) for k, v in red.items()}
    ids = [rank, rank + world]  # engine.validation: items rank, rank + world, ...
    res = []
    for iid in ids:
        k = 0 if rank == 5 else 1 + iid % 4
        res.append((torch.full((k, 4), float(iid)), torch.full((k,), 0.25 + 0.01 * iid), torch.full((k,), 1 + iid % 8, dtype=torch.int64)))
    merged = comm.gather_detections(res, ids)
    out["merged"] = None if merged is None else {i: (b.numpy(), s.numpy(), l.numpy()) for i, (b, s, l) in merged.items()}
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_gloo():
    import numpy as np
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0 = res[0]
    assert all(r["split"] == (2, 8) for r in res) and all("divisible" in r["split_err"] for r in res)
    # one canonical collective sequence on every rank, whatever order (or subset) of hooks fired there
    names = r0["names"]
    assert names == ["fcos", "dis", "middle_head", "backbone:c4", "backbone:c3", "backbone:rest"]
    assert all(r["log"] == r0["log"] for r in res), [r["log"] for r in res]
    assert r0["issued_by_hooks"] == len(r0["log"])  # rank 0 reported every bucket: nothing was left to the flush
    assert any(r["issued_by_hooks"] < len(r["log"]) for r in res[1:])  # ... the others needed it
    sys.path.insert(0, ROOT)
    from scan_amd import engine
    n = engine.Trainer(engine.build_model(9, device="cpu")).grad_arena.numel()
    assert engine._merge_ranges(r0["log"]) == [(0, n)] and sum(b - a for a, b in r0["log"]) == n  # each element exactly once
    expect = torch.arange(n, dtype=torch.float32) % 89 + 8.0 * 3.5  # the mean over ranks 0..7 of (x + 8 r)
    for r in res:
        assert np.allclose(r["arena_head"], expect[:2048].numpy(), rtol=1e-6) and np.allclose(r["arena_tail"], expect[-2048:].numpy(), rtol=1e-6)
        assert abs(r["arena_sum"] - expect.double().sum().item()) < 1e-4 * n
    # the paradigm buffer is the same on all ranks and equals a single process fed the class means over the ranks that saw the class
    for r in res[1:]:
        assert np.array_equal(r["proto"], r0["proto"]), "paradigm buffers diverged across ranks"
    from scan_amd import synth
    from scan_amd.modeling import condgraph
    mh = condgraph.GRAPHModule(256, 9)
    mh.load_state_dict(synth.middle_head_state_dict(9))
    for it in range(3):
        pbs = torch.stack([torch.from_numpy(r["pbs"][it]) for r in res])       # [8, 9, 256]
        seen = pbs.sum(-1).bool().float()[..., None]                            # [8, 9, 1]
        mh.update_prototype_nx1_rnn((pbs * seen).sum(0) / seen.sum(0).clamp(min=1))
    assert torch.allclose(mh.prototype, torch.from_numpy(r0["proto"]), rtol=1e-5, atol=1e-6)
    # loss scalars averaged on rank 0; detections of all 16 images on rank 0 only, empty images included
    assert r0["red"]["loss_cls_gs"] == 3.5 and r0["red"]["zero_gt"] == 0.0
    assert all(r["merged"] is None for r in res[1:]) and sorted(r0["merged"]) == list(range(16))
    for iid, (b, s, l) in r0["merged"].items():
        k = 0 if iid % 8 == 5 else 1 + iid % 4
        assert b.shape == (k, 4) and (b == iid).all() and (l == 1 + iid % 8).all() and np.allclose(s, 0.25 + 0.01 * iid)


def test_launcher_starts_eight_ranks():
    """`python bench.py --gpus 8` as the driver's SCALE run would start it, minus the GPUs: the launcher must bring up
    exactly eight ranks in one process group (tests/test_launcher.py covers two and the refusals)."""
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"launch_check": True, "n_gpus": 8, "ranks_in_process_group": 8}
