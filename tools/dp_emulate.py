#!/usr/bin/env python
"""What the gradient all-reduce of an N-rank run costs the step of ONE rank, measured on one GPU by emulation.

The data-parallel step (engine.Trainer(distributed=True)) runs here on a one-rank RCCL group -- hooks, buckets, stream
waits and the collective calls are the real ones, but a one-rank all-reduce moves nothing.  What an 8-rank ring all-reduce
adds on each GPU is a kernel of `channels` workgroups resident for 2 (N - 1) / N x bytes / (link rate), and THAT is put
behind every bucket's collective on the comm stream by Trainer.comm_hook: scan_comm_standin (scan_amd/csrc/commsim.hip),
`--wgs` workgroups of 256 threads read-modify-writing the bucket's range, paced to `--gbps` (the ring's send rate: one xGMI
link = 153 GB/s; 0 = unpaced, HBM rate).  Policies (Trainer.dp_policy): overlap = six ranges as they become final, beside
the backward; coarse = FCOS head + discriminators early, the rest at the end; tail = one all-reduce after the backward.

    python tools/dp_emulate.py [--ranks 8] [--wgs 8,16,32,64] [--gbps 153,459] [--steps 10] [--rounds 3]

Prints one line per configuration per round (ms/step) and a summary table (median over rounds, difference to the plain
step of the same round).  profiles/r06_dp_emulation.txt is this output.
"""
import argparse
import ctypes
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8, help="ring size the stand-in emulates (traffic factor 2 (N - 1) / N)")
    ap.add_argument("--wgs", default="8,16,32,64", help="workgroups of the stand-in kernel = RCCL channels")
    ap.add_argument("--gbps", default="153,459", help="aggregate ring send rates to pace the stand-in to; 0 = unpaced")
    ap.add_argument("--policies", default="overlap,coarse,tail")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from scan_amd import _lib, engine, ops, synth
    mcfg = engine.CONFIGS["c2f"]
    model = engine.build_model(device=dev, settings=mcfg)
    engine.load_procedural_weights(model, mcfg["num_classes"], mcfg["conv_body"])
    plain = engine.Trainer(model, settings=mcfg, distributed=False)
    dp = engine.Trainer(model, settings=mcfg, distributed=True)
    H, W, B = a.height, a.width, a.batch
    imgs_s = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 1234)], 32)
    imgs_t = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 2234)], 32)
    tg = synth.synth_targets(B, H, W, mcfg["num_classes"] - 1, 12, 4321)
    traffic = 2.0 * (a.ranks - 1) / a.ranks

    def activate(tr):  # the two trainers share the model: per-trainer stream roles live in model / ops attributes
        model["middle_head"].out_stream = tr.out_stream
        ops.WGRAD_STREAM = tr.wgrad_stream

    def hook_for(wgs, gbps):
        def hook(lo, hi):  # runs inside `with torch.cuda.stream(comm_stream)`, behind the collective of [lo, hi)
            g = dp.grad_arena[lo:hi]
            _lib.call("scan_comm_standin", ctypes.c_void_p(g.data_ptr()), hi - lo, traffic, wgs, float(gbps), ops._stream())
        return hook

    def run(tr, steps):
        activate(tr)
        for _ in range(2):
            tr.step(imgs_s, tg, imgs_t)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            losses = tr.step(imgs_s, tg, imgs_t)
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(v)) for v in losses.values())
        return (time.time() - t0) / steps * 1e3

    configs = [("plain", None, None, None)]
    for pol in a.policies.split(","):
        configs.append(("dp1 %s (no stand-in)" % pol, pol, None, None))
    for gbps in [float(x) for x in a.gbps.split(",")]:
        for wgs in [int(x) for x in a.wgs.split(",")]:
            for pol in a.policies.split(","):
                configs.append(("%s wgs=%d gbps=%g" % (pol, wgs, gbps), pol, wgs, gbps))
    nbytes = 4 * dp.grad_arena.numel()
    print("# tools/dp_emulate.py: ring of %d ranks emulated on one GPU, %d src + %d tgt frames at %dx%d, %d steps per cell, %d rounds"
          % (a.ranks, B, B, H, W, a.steps, a.rounds))
    print("# gradient arena %.1f MB; stand-in traffic factor %.3f -> %.1f MB read + written per step; paced duration per step: %s"
          % (nbytes / 1e6, traffic, traffic * nbytes / 1e6,
             ", ".join("%.2f ms at %g GB/s" % (traffic * nbytes / (float(g) * 1e9) * 1e3, float(g)) for g in a.gbps.split(",") if float(g) > 0)))
    dp.dp_policy = "overlap"
    dp._bucket_list = None
    print("# buckets (overlap): " + ", ".join("%s %.1f MB" % (n, 4 * sum(h - l for l, h in r) / 1e6) for n, r, _, _ in dp._buckets()))
    res = {c[0]: [] for c in configs}
    for rnd in range(a.rounds):
        for name, pol, wgs, gbps in configs:
            if pol is None:
                ms = run(plain, a.steps)
            else:
                dp.dp_policy, dp._bucket_list = pol, None
                dp.comm_hook = hook_for(wgs, gbps) if wgs is not None else None
                ms = run(dp, a.steps)
            res[name].append(ms)
            print("round %d  %-32s %7.2f ms/step" % (rnd, name, ms), flush=True)
    print("# summary: median ms/step over rounds, and the median of (cell - plain of the same round)")
    for name, _, _, _ in configs:
        d = [x - p for x, p in zip(res[name], res["plain"])]
        print("%-32s %7.2f   %+6.2f" % (name, statistics.median(res[name]), statistics.median(d)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
