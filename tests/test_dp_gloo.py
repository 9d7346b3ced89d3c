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
