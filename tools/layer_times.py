#!/usr/bin/env python
"""Where the conv time of one serial training step goes, by LAUNCH SHAPE: every MFMA conv launch of a step (bench workload:
2 source + 2 target frames at 1024x2048, paired schedule, streams serialised) timed with HIP events and grouped by
(kernel-timer name, rows, flops) -- the per-layer view the per-symbol table of bench.py averages away.

    python tools/layer_times.py [--mode bf16x6|bf16x3|fp32] [--steps 2]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from scan_amd import engine, ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf16x6")
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.CONV_MODE = a.mode
    mcfg = engine.CONFIGS["c2f"]
    model = engine.build_model(device=dev, settings=mcfg)
    engine.load_procedural_weights(model, mcfg["num_classes"], mcfg["conv_body"])
    tr = engine.Trainer(model, settings=mcfg)
    tr.overlap_target = False
    tr.dis_streams = {}
    model["middle_head"].out_stream = None
    H, W, B = 1024, 2048, 2
    imgs_s = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 1234)], 32)
    imgs_t = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 2234)], 32)
    tg = synth.synth_targets(B, H, W, 8, 12, 4321)
    for _ in range(2):
        tr.step(imgs_s, tg, imgs_t)
    torch.cuda.synchronize()
    # wrap the timer so that a record is keyed by its shape too
    kt = ops.kernel_timer
    orig = kt.begin

    def begin(name, flops):
        return orig("%s|%.4g" % (name, flops), flops)

    kt.begin = begin
    kt.enabled = True
    kt.reset()
    for _ in range(a.steps):
        tr.step(imgs_s, tg, imgs_t)
    torch.cuda.synchronize()
    kt.enabled = False
    rows = []
    for key, r in kt.summary().items():
        name, fl = key.split("|")
        rows.append((r["total_ms"] / a.steps, name, float(fl), r["launches"] // a.steps, r["avg_ms"], r["tflops"]))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("mode %s: %.1f ms of conv launches per step" % (a.mode, tot))
    print("%8s %6s  %-34s %10s %6s %9s %8s" % ("ms/step", "share", "record", "GFLOP", "n/step", "avg ms", "TFLOP/s"))
    for ms, name, fl, n, avg, tf in rows[:40]:
        print("%8.3f %5.1f%%  %-34s %10.1f %6d %9.4f %8.1f" % (ms, 100 * ms / tot, name, fl / 1e9, n, avg, tf))


if __name__ == "__main__":
    main()
