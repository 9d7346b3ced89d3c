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
