#!/usr/bin/env python
"""A/B timing of the 3x3 conv kernels on the layer shapes of the bench workload (4 frames of 1024x2048 in the paired
step), with HIP events on the launch stream, variants selected through scan_tune.  Each variant's output is compared
with the first one's (they must agree bit for bit: the K order per output element is the same).

    python tools/conv_bench.py [--op fwd|dgrad|wgrad] [--mode bf16x6|bf16x3|fp32] [--reps 5] [--variants conv_bn256=0,conv_bn256=1]

--op dgrad: the data gradient (the forward kernel on dY with flipped planes) through autograd's backward with the weight
frozen.  TF = fp32-equivalent TFLOP/s; ceilings: bf16x6 416.7, bf16x3 833.3, fp32 157.3.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from scan_amd import _lib, ops  # noqa: E402

# (name, n_images, level sizes, Cin, Cout)
SHAPES = [
    ("towers 256->256, P3..P7, 4 frames", 4, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)], 256, 256),
    ("towers 256->256, P3..P7, 2 frames (FCOS)", 2, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)], 256, 256),
    ("conv3_x 256->256 @256x512, 4 frames", 4, [(256, 512)], 256, 256),
    ("conv4_x 512->512 @128x256, 4 frames", 4, [(128, 256)], 512, 512),
    ("conv5_x 512->512 @64x128, 4 frames", 4, [(64, 128)], 512, 512),
    ("conv2_2 128->128 @512x1024, 4 frames", 4, [(512, 1024)], 128, 128),
    ("conv1_2 64->64 @1024x2048, 4 frames", 4, [(1024, 2048)], 64, 64),
    ("conv2_1 64->128 @512x1024, 4 frames", 4, [(512, 1024)], 64, 128),
    ("dis P3 264->1024, 4 frames", 4, [(128, 256)], 264, 1024),
    ("head_out 268->256 pyramid, 4 frames", 4, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)], 268, 256),
]


# 1x1 convs of the ResNet-50 body at the K2C bench shape (4 frames of 1024x2048 in the paired step: C2 at 256x512 ... C5 at 32x64)
# and the FPN laterals; --ksize 1 selects these
SHAPES_1X1 = [
    ("layer1 conv1 256->64 @256x512, 4 frames", 4, [(256, 512)], 256, 64),
    ("layer1 conv3 64->256 @256x512, 4 frames", 4, [(256, 512)], 64, 256),
    ("layer2 conv1 512->128 @128x256, 4 frames", 4, [(128, 256)], 512, 128),
    ("layer2 conv3 128->512 @128x256, 4 frames", 4, [(128, 256)], 128, 512),
    ("layer3 conv1 1024->256 @64x128, 4 frames", 4, [(64, 128)], 1024, 256),
    ("layer3 conv3 256->1024 @64x128, 4 frames", 4, [(64, 128)], 256, 1024),
    ("layer4 conv1 2048->512 @32x64, 4 frames", 4, [(32, 64)], 2048, 512),
    ("layer4 conv3 512->2048 @32x64, 4 frames", 4, [(32, 64)], 512, 2048),
    ("FPN lateral 512->256 @128x256, 4 frames", 4, [(128, 256)], 512, 256),
    ("FPN lateral 2048->256 @32x64, 4 frames", 4, [(32, 64)], 2048, 256),
]
KS = 3  # --ksize


def run(shape_def, reps, dev):
    name, n, sizes, cin, cout = shape_def
    shape = ops.PyramidShape(n, sizes)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((shape.rows, ops.pad4(cin)), device=dev, generator=g)
    w = (torch.randn((cout, cin, KS, KS), device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn((cout,), device=dev, generator=g)
    flops = 2.0 * shape.rows * cout * KS * KS * cin

    def fwd():
        with torch.no_grad():
            return ops.conv2d(x, w, b, shape, KS, 1)

    fwd()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        y = fwd()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    return y, us, flops / us * 1e-6


def run_dgrad(shape_def, reps, dev):
    name, n, sizes, cin, cout = shape_def
    shape = ops.PyramidShape(n, sizes)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((shape.rows, ops.pad4(cin)), device=dev, generator=g).requires_grad_(True)
    w = (torch.randn((cout, cin, KS, KS), device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((shape.rows, cout), device=dev, generator=g)
    flops = 2.0 * shape.rows * cout * KS * KS * cin
    y = ops.conv2d(x, w, None, shape, KS, 1)

    def go():
        x.grad = None
        y.backward(dy, retain_graph=True)

    go()
    torch.cuda.synchronize()
    ops.kernel_timer.enabled = True
    ops.kernel_timer.reset()
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    ops.kernel_timer.enabled = False
    rec = [r for k, r in ops.kernel_timer.summary().items() if "dgrad" in k][0]
    us = rec["total_ms"] * 1e3 / rec["launches"]
    return x.grad.clone(), us, flops / us * 1e-6


def run_wgrad(shape_def, reps, dev):
    name, n, sizes, cin, cout = shape_def
    shape = ops.PyramidShape(n, sizes)
    cs = ops.pad4(cin)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((shape.rows, cs), device=dev, generator=g)
    dy = torch.randn((shape.rows, cout), device=dev, generator=g)
    dw = torch.empty((cout, 9, cs), device=dev)
    db = torch.empty((cout,), device=dev)
    sfx = ops.CONV_MODE
    if sfx == "fp32":
        ws = torch.empty((_lib.query("scan_conv2d_wgrad_ws_floats", shape.ref(), cs, cout, 3),), device=dev)
    else:
        ws = torch.empty((_lib.query("scan_conv3x3_wgrad_%s_ws_floats" % sfx, shape.ref(), cs, cout),), device=dev)
    flops = 2.0 * shape.rows * cout * 9 * cin

    def go():
        if sfx == "fp32":
            _lib.call("scan_conv2d_wgrad", ops._ptr(x), shape.ref(), cs, ops._ptr(dy), shape.ref(), cout, cout, 3, 1,
                      ops._ptr(dw), 0, ops._ptr(ws), ops._stream())
        else:
            _lib.call("scan_conv3x3_wgrad_" + sfx, ops._ptr(x), shape.ref(), cs, ops._ptr(dy), cout, cout, ops._ptr(dw),
                      ops._ptr(db), 0, ops._ptr(ws), ops._stream())

    go()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        go()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    return dw.clone(), us, flops / us * 1e-6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", choices=("fwd", "dgrad", "wgrad"), default="fwd")
    ap.add_argument("--mode", choices=("bf16x6", "bf16x3", "fp32"), default="bf16x6")
    ap.add_argument("--shapes", default="", help="comma-separated substrings selecting rows of SHAPES")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", default="conv_bn256=0,conv_bn256=1")
    ap.add_argument("--ksize", type=int, choices=(1, 3), default=3, help="1: the ResNet-50 / FPN 1x1 shapes (fwd, dgrad)")
    a = ap.parse_args()
    global KS
    KS = a.ksize
    if KS == 1 and a.op == "wgrad":
        raise SystemExit("--ksize 1: fwd and dgrad only")
    dev = torch.device("cuda:0")
    ops.CONV_MODE = a.mode
    print("op %s  mode %s" % (a.op, a.mode), flush=True)
    shapes = [sd for sd in (SHAPES if KS == 3 else SHAPES_1X1) if not a.shapes or any(t in sd[0] for t in a.shapes.split(","))]
    variants = [(v, "") for v in a.variants.split(",")]  # each variant: "key=value" or "key=value+key=value"
    for sd in shapes:
        ref = None
        line = "%-44s" % sd[0]
        best = {}
        for rnd in range(a.rounds):  # A B A B ...: clock / cache state drifts show up as round-to-round spread
            for key, val in variants:
                olds = []
                for kv in key.split("+"):
                    k, v = kv.split("=")
                    assert _lib.query("scan_tune_get", k.encode()) != _lib.TUNE_UNKNOWN, "unknown scan_tune key " + k
                    olds.append((k, _lib.query("scan_tune", k.encode(), int(v))))
                y, us, tf = {"fwd": run, "dgrad": run_dgrad, "wgrad": run_wgrad}[a.op](sd, a.reps, dev)
                for k, o in olds:
                    _lib.query("scan_tune", k.encode(), o)
                if ref is None:
                    ref = y
                b = best.setdefault((key, val), [])
                b.append(us)
                if rnd == 0 and not torch.equal(ref, y):  # different kernels agree to rounding, not bit for bit
                    line += " [%s differs: max %.2e of %.2e]" % (key, (ref - y).abs().max().item(), ref.abs().max().item())
        flops = 2.0 * ops.PyramidShape(sd[1], sd[2]).rows * sd[4] * KS * KS * sd[3]
        for (key, val), us in best.items():
            line += "  %s: %s us -> %6.1f TF" % (key, "/".join("%.0f" % u for u in us), flops / min(us) * 1e-6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
