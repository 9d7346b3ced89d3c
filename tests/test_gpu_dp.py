"""Data-parallel Trainer on the GPU with two ranks sharing one device (gloo carries the collectives; RCCL refuses two
ranks on one GPU).  Exercises everything engine.Trainer adds for N > 1 -- gradient averaging through the per-sub-model
flat buffers on the side stream, the paradigm all-reduce inside the source forward, the waits before the fused SGD --
around the real HIP kernels:
  (a) different shards per rank -> parameters, momentum and the paradigm buffer are identical on both ranks;
  (b) the same shard on both ranks -> identical to the single-process run (averaging equal gradients is a no-op)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, H, W = 2, 128, 128


def _batch(shard, dev):
    from scan_amd import synth
    return (synth.synth_images(1, H, W, 11 + 10 * shard).to(dev), synth.synth_targets(1, H, W, 8, 6, 13 + 10 * shard),
            synth.synth_images(1, H, W, 12 + 10 * shard).to(dev))


def _run(rank, world, same_shard, dev_index=0, steps=STEPS, mixed_schedules=False):
    from scan_amd import engine, synth
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    model = engine.build_model(9, device=dev, attn_dropout=0.0)
    engine.load_procedural_weights(model)
    trainer = engine.Trainer(model, distributed=True if world > 1 else None)
    if mixed_schedules:  # rank 0: source + target frames as one pyramid; rank 1: the reference's three phases
        trainer.paired = rank == 0
    shard = 0 if same_shard else rank
    imgs_s = synth.synth_images(1, H, W, 11 + 10 * shard).to(dev)
    imgs_t = synth.synth_images(1, H, W, 12 + 10 * shard).to(dev)
    tg = synth.synth_targets(1, H, W, 8, 6, 13 + 10 * shard)
    for _ in range(steps):
        losses = trainer.step(imgs_s, tg, imgs_t)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v)) for v in losses.values())
    out = {k: g.flat_p.detach().cpu().clone() for k, g in trainer.groups.items()}
    if world > 1:
        out["collective_log"] = torch.tensor(trainer.collective_log, dtype=torch.int64)
        out["issued_before_flush"] = torch.tensor([trainer.issued_before_flush, len(trainer._buckets())])
    out["momentum_backbone"] = trainer.groups["backbone"].flat_m.detach().cpu().clone()
    out["prototype"] = model["middle_head"].prototype.detach().cpu().clone()
    return out


def _worker(rank, world, port, same_shard, outdir, backend="gloo", steps=STEPS, mixed_schedules=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":  # one GPU per rank, RCCL carries the collectives (side-stream all-reduce for real)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.save(_run(rank, world, same_shard, rank if backend == "nccl" else 0, steps, mixed_schedules),
                   os.path.join(outdir, "rank%d.pt" % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _spawn(same_shard, outdir, backend="gloo", steps=STEPS, mixed_schedules=False, world=2):
    ctx = mp.get_context("spawn")
    port = 29600 + os.getpid() % 2000 + (1 if same_shard else 0) + (2 if backend == "nccl" else 0) \
        + (4 if mixed_schedules else 0) + (8 if steps != STEPS else 0) + (16 if world != 2 else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, same_shard, str(outdir), backend, steps, mixed_schedules))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    return tuple(torch.load(os.path.join(str(outdir), "rank%d.pt" % r)) for r in range(world))


def test_two_ranks_different_shards_stay_identical(device, tmp_path):
    a, b = _spawn(False, tmp_path)
    for k in a:
        assert torch.equal(a[k], b[k]), "rank 0 and rank 1 diverged in %s" % k


def test_four_ranks_different_shards_stay_identical(device, tmp_path):
    """four ranks (gloo, all on this GPU) with four different shards: the 1 / world scale, the bucket ranges and the paradigm
    all-reduce at a world size other than two -- every rank ends with bit-identical parameters, momentum and paradigm buffer and
    has issued the same collective sequence, every bucket from the backward's own hooks."""
    outs = _spawn(False, tmp_path, world=4)
    for r in outs[1:]:
        for k in outs[0]:
            assert torch.equal(outs[0][k], r[k]), "ranks diverged in %s" % k
    fired, total = outs[0]["issued_before_flush"].tolist()
    assert fired == total == 6


def test_every_gradient_bucket_is_fired_by_the_backward_itself(device, tmp_path):
    """paired step, VGG backbone: all six buckets -- FCOS head, discriminators, middle head, conv5_x, conv4_x + FPN and (round 5)
    conv3_x + all biases, fired from conv3_1's autograd node -- are issued by hooks while backward() is still running; nothing
    is left to the flush behind it (until round 5 the last bucket, 42.6 MB, was).  Same collective sequence on both ranks."""
    a, b = _spawn(False, tmp_path)
    assert torch.equal(a["collective_log"], b["collective_log"])
    for r in (a, b):
        fired, total = r["issued_before_flush"].tolist()
        assert total == 6 and fired == total, (fired, total)


def test_two_ranks_same_shard_equal_single_process(device, tmp_path):
    a, b = _spawn(True, tmp_path)
    ref = _run(0, 1, True)
    for k in ref:
        assert torch.equal(a[k], b[k]), k
        # two runs of one schedule differ by ~7e-7 (float atomics; test_gpu_model.py::test_stream_overlap_is_race_free)
        assert torch.allclose(a[k], ref[k], rtol=1e-4, atol=5e-6), (k, (a[k] - ref[k]).abs().max().item())


@pytest.mark.parametrize("same_shard", [False, True])
def test_two_ranks_rccl(device, tmp_path, same_shard):
    """the same two checks with one GPU per rank and backend "nccl" (= RCCL): the gradient all-reduces really run on
    the side stream while the backbone back-propagates, and the wgrad kernels that bypass AccumulateGrad must be
    ordered before them.  Needs two visible GPUs; skipped on the 1-GPU boxes."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device)")
    a, b = _spawn(same_shard, tmp_path, backend="nccl")
    for k in a:
        assert torch.equal(a[k], b[k]), "rank 0 and rank 1 diverged in %s" % k
    if same_shard:
        ref = _run(0, 1, True)
        for k in ref:
            assert torch.allclose(a[k], ref[k], rtol=1e-4, atol=5e-6), (k, (a[k] - ref[k]).abs().max().item())


def test_two_ranks_equal_one_process_accumulating_both_shards(device, tmp_path):
    """SURVEY.md 8(e) parity clause (b): an N-rank run equals ONE process that accumulates the gradients of the same
    shards.  Two ranks (different shards, one optimizer step) against a single process that back-propagates shard 0 and
    shard 1 separately, averages the two gradient arenas and applies the same fused SGD step.  The paradigm buffer is
    part of the state: the single process feeds the middle head the rank-averaged class means the two ranks exchange
    (condgraph.update_prototype_nx1_rnn), taken from a recording pass -- the class means depend on the parameters and
    the shard only, not on the paradigm buffer."""
    from scan_amd import engine
    a, b = _spawn(False, tmp_path, steps=1)
    dev = torch.device("cuda", 0)

    def fresh():
        model = engine.build_model(9, device=dev, attn_dropout=0.0)
        engine.load_procedural_weights(model)
        return model, engine.Trainer(model)

    # pass 1: the per-shard class means
    pbs = []
    for shard in (0, 1):
        model, trainer = fresh()
        mh = model["middle_head"]
        orig = mh.update_prototype_nx1_rnn
        mh.update_prototype_nx1_rnn = lambda pb, _o=orig: (pbs.append(pb.detach().clone()), _o(pb))[1]
        trainer._optimizer_step = lambda: None
        trainer.step(*_batch(shard, dev))
        del model, trainer
    ex = [pb.sum(-1).bool().float()[:, None] for pb in pbs]
    pb_avg = (pbs[0] * ex[0] + pbs[1] * ex[1]) / (ex[0] + ex[1]).clamp(min=1)
    # pass 2: one model, both shards' gradients from the SAME initial state, then one optimizer step on their mean
    model, trainer = fresh()
    mh = model["middle_head"]
    orig = mh.update_prototype_nx1_rnn
    mh.update_prototype_nx1_rnn = lambda pb, _o=orig: _o(pb_avg)
    real_step = trainer._optimizer_step
    trainer._optimizer_step = lambda: None
    proto0, counter0 = mh.prototype.clone(), mh.counter_rnn.counter
    grads = []
    for shard in (0, 1):
        mh.prototype.copy_(proto0)
        mh.counter_rnn.counter = counter0
        trainer.step(*_batch(shard, dev))
        grads.append(trainer.grad_arena.clone())
    trainer.grad_arena.copy_((grads[0] + grads[1]) / 2)
    real_step()
    torch.cuda.synchronize()
    for k, g in trainer.groups.items():
        assert torch.equal(a[k], b[k]), k
        ref = g.flat_p.detach().cpu()
        # one lr-sized step: the parameters moved by ~1e-3 * grad; the bar is 1e-5 on the parameters themselves
        assert torch.allclose(a[k], ref, rtol=1e-5, atol=1e-6), (k, (a[k] - ref).abs().max().item())
    assert torch.allclose(a["prototype"], mh.prototype.detach().cpu(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(a["momentum_backbone"], trainer.groups["backbone"].flat_m.cpu(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_mixed_schedules_issue_the_same_collectives(device, tmp_path, backend):
    """(backend "nccl": one GPU per rank over RCCL -- gloo's CUDA path synchronises the host and would hide a missing stream
    dependency of the mid-backward bucket all-reduces; needs two GPUs, skipped on 1-GPU boxes)
    rank 0 takes Trainer.step_paired, rank 1 the three-phase schedule (what happens when the ranks' batches pad to
    different shapes): the gradient all-reduces must pair up -- same ranges, same order -- and, with the same shard on
    both ranks, the result equals the single-process run."""
    # ONE optimizer step: the two schedules sum in different orders, and at this frame size (levels down to 1x1 pixel)
    # a second step amplifies first-step rounding differences to ~1e-3 relative ("chaotic at rounding
    # level") -- after one step the comparison with the single-process run is a rounding-level one
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device)")
    a, b = _spawn(True, tmp_path, backend=backend, steps=1, mixed_schedules=True)
    assert torch.equal(a["collective_log"], b["collective_log"]) and a["collective_log"].shape[0] >= 7
    ref = _run(0, 1, True, steps=1)
    for k in ref:
        # both ranks hold the same averaged gradients: identical parameters whatever schedule produced their shares
        assert torch.equal(a[k], b[k]), (k, (a[k] - b[k]).abs().max().item())
        assert torch.allclose(a[k], ref[k], rtol=1e-4, atol=5e-6), (k, (a[k] - ref[k]).abs().max().item())


def _policy_worker(rank, world, port, outdir, policy):
    """one rank on RCCL: the data-parallel machinery (hooks, buckets, comm stream = side stream s2, ReduceOp.AVG launched on the
    calling stream) under one dp_policy; the mean over one rank is the identity"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from scan_amd import engine, ops, synth
        dev = torch.device("cuda", 0)
        model = engine.build_model(9, device=dev, attn_dropout=0.0)
        engine.load_procedural_weights(model)
        trainer = engine.Trainer(model, distributed=True, dp_policy=policy)
        assert trainer.comm_stream is ops.borrow_side_streams(3)[2] and trainer.out_stream is ops.borrow_side_streams(3)[1]
        imgs_s, tg, imgs_t = _batch(0, dev)
        for _ in range(STEPS):
            losses = trainer.step(imgs_s, tg, imgs_t)
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(v)) for v in losses.values())
        out = {k: g.flat_p.detach().cpu().clone() for k, g in trainer.groups.items()}
        out["collective_log"] = torch.tensor(trainer.collective_log, dtype=torch.int64)
        out["n_arena"] = torch.tensor([trainer.grad_arena.numel()])
        torch.save(out, os.path.join(outdir, "policy_%s.pt" % policy))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("policy", ["overlap", "coarse", "tail"])
def test_one_rank_rccl_every_policy_equals_plain_step(device, tmp_path, policy):
    """engine.Trainer.dp_policy on a real RCCL process group (one rank: what a 1-GPU box allows): overlap / coarse / tail issue
    6+ / 2 / 1 all-reduce ranges that cover the arena once, on side stream s2 (no fifth stream), and leave the parameters where
    the plain single-process step leaves them (run-to-run tolerance of the loss atomics)."""
    ctx = mp.get_context("spawn")
    port = 30100 + os.getpid() % 1500 + {"overlap": 0, "coarse": 1, "tail": 2}[policy]
    p = ctx.Process(target=_policy_worker, args=(0, 1, port, str(tmp_path), policy))
    p.start()
    p.join(timeout=900)
    assert p.exitcode == 0
    got = torch.load(os.path.join(str(tmp_path), "policy_%s.pt" % policy))
    log = [tuple(r) for r in got["collective_log"].tolist()]
    n = int(got["n_arena"][0])
    assert sum(b - a for a, b in log) == n and min(a for a, _ in log) == 0 and max(b for _, b in log) == n
    assert {"overlap": len(log) >= 6, "coarse": len(log) == 2, "tail": len(log) == 1}[policy], log
    ref = _run(0, 1, True)
    for k, v in ref.items():
        if k in got:
            assert torch.allclose(got[k], v, rtol=1e-4, atol=5e-6), (policy, k, (got[k] - v).abs().max().item())
