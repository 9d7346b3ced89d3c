#!/usr/bin/env python
"""What the drop-in operator surface costs beside the engine: the SAME DA iteration through scan_amd.surface (NCHW tensors, one
module call per level, scan_amd.layers on the C++ autograd operators -- the call shape of the reference's module files) and
through scan_amd.engine (one row matrix per pyramid, one launch per layer), ms/step and kernel launches/step, plus a per-operator
table: one layer's forward + backward through layers.Conv2d per level vs ops.conv2d on the pyramid, layout conversions counted.

    python tools/surface_bench.py [--steps 5]          (also: bench.py's `drop_in` leg calls measure())
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def count_launches(fn):
    """GPU kernel launches of one call of fn (torch profiler, device activities)"""
    import torch
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    n = 0
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA and not e.name.startswith(("Memcpy", "Memset")):
            n += 1
    return n


def op_table(dev, reps=5):
    """forward + backward of one layer: engine call (ops.conv2d on the pyramid rows) vs surface calls (layers.Conv2d per level on
    NCHW channels_last tensors, to-rows / to-NCHW views and the per-call weight split included)"""
    import torch
    from scan_amd import layers as L
    from scan_amd import ops
    cases = [("tower 3x3 256->256, P3..P7, 4 frames", 4, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)], 256, 256, 3),
             ("conv4_x 3x3 512->512 @128x256, 4 frames", 4, [(128, 256)], 512, 512, 3),
             ("FPN lateral 1x1 512->256 @128x256, 4 frames", 4, [(128, 256)], 512, 256, 1),
             ("head 3x3 256->8, P3..P7, 2 frames", 2, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)], 256, 8, 3)]
    out = []
    for name, n, sizes, cin, cout, k in cases:
        shape = ops.PyramidShape(n, sizes)
        g = torch.Generator(device=dev).manual_seed(1)
        conv = L.Conv2d(cin, cout, k, 1, k // 2).to(dev)
        conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
        rows = torch.randn((shape.rows, cin), device=dev, generator=g).requires_grad_(True)
        lv = [rows.detach()[shape.row_off[l]:shape.row_off[l + 1]].view(n, h, w, cin).permute(0, 3, 1, 2).requires_grad_(True)
              for l, (h, w) in enumerate(sizes)]

        def engine():
            ops.invalidate_weight_planes()
            ops.begin_weight_epoch()
            y = ops.conv2d(rows, conv.weight, conv.bias, shape, k, 1)
            y.backward(y.detach())

        def surface():
            ys = [conv(x) for x in lv]
            torch.autograd.backward(ys, [y.detach() for y in ys])

        rec = {"op": name, "calls_surface": len(sizes)}
        for tag, fn in (("engine", engine), ("surface", surface)):
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.time()
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            torch.cuda.synchronize()
            rec[tag + "_us"] = round(s.elapsed_time(e) * 1e3 / reps, 1)
            rec[tag + "_host_us"] = round((time.time() - t0) * 1e6 / reps, 1)
            rec[tag + "_launches"] = count_launches(fn)
        rec["ratio"] = round(rec["surface_us"] / rec["engine_us"], 3)
        out.append(rec)
    return out


def measure(trainer, imgs_s, tg, imgs_t, steps=5, with_ops=True):
    """trainer: an engine.Trainer whose model is then ADOPTED by the surface (scan_amd.surface.adopt re-classes its Conv2d /
    GroupNorm modules; the engine path ignores module classes, so the trainer keeps working).  imgs_*: ImageList or NCHW."""
    import torch
    from scan_amd import layers as L
    from scan_amd import surface
    xs = imgs_s.tensors if hasattr(imgs_s, "tensors") else imgs_s
    xt = imgs_t.tensors if hasattr(imgs_t, "tensors") else imgs_t
    st = surface.SurfaceTrainer(trainer)
    for _ in range(2):
        losses = st.step(xs, tg, xt)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        losses = st.step(xs, tg, xt)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    rec = {"ms_per_step": round(ms, 2), "steps": steps, "ops_backend": L.OPS_BACKEND,
           "losses_finite": all(bool(torch.isfinite(v)) for v in losses.values()),
           "launches_per_step": count_launches(lambda: st.step(xs, tg, xt)),
           "graph": "scan_amd.surface: NCHW, one module call per level, layers.Conv2d / GroupNorm / dynamic_conv_softmax on "
                    "scan_ops._ops, three-phase schedule, no side streams"}
    rec["engine_launches_per_step"] = count_launches(lambda: trainer.step(imgs_s, tg, imgs_t))
    if with_ops:
        rec["per_op"] = op_table(xs.device)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    a = ap.parse_args()
    import torch
    from scan_amd import engine, synth
    dev = torch.device("cuda", 0)
    mcfg = engine.CONFIGS["c2f"]
    model = engine.build_model(device=dev, settings=mcfg)
    engine.load_procedural_weights(model, mcfg["num_classes"], mcfg["conv_body"])
    trainer = engine.Trainer(model, settings=mcfg)
    H, W, B = a.height, a.width, 2
    imgs_s = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 1234)], 32)
    imgs_t = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 2234)], 32)
    tg = synth.synth_targets(B, H, W, mcfg["num_classes"] - 1, 12, 4321)
    for _ in range(3):
        trainer.step(imgs_s, tg, imgs_t)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        trainer.step(imgs_s, tg, imgs_t)
    torch.cuda.synchronize()
    eng = (time.time() - t0) / a.steps * 1e3
    rec = measure(trainer, imgs_s, tg, imgs_t, a.steps)
    rec["engine_ms_per_step"] = round(eng, 2)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
