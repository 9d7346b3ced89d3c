"""Class-branch output conv of the CKA discriminators: grouped kernels (ops.gconv3x3_to1) vs the dense conv over the
block-diagonal stacked weight, forward and backward, at the P3 / P4 / P6 sizes of the bench workload (4 frames).

    python tools/gconv_bench.py
"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan_amd import ops
dev = torch.device("cuda", 0)
G = 8
for (n, h, w) in [(4, 128, 256), (4, 64, 128), (4, 16, 32)]:
    shape = ops.PyramidShape(n, [(h, w)])
    x = torch.relu(torch.randn(shape.rows, G * 128, device=dev))
    ws = torch.zeros(G, G * 128, 3, 3, device=dev)
    for c in range(G):
        ws[c, c * 128:(c + 1) * 128] = torch.randn(128, 3, 3, device=dev) / 30
    ws = ws.contiguous(memory_format=torch.channels_last)
    b = torch.randn(G, device=dev)
    for name, fn in (("grouped", lambda xx, ww: ops.gconv3x3_to1(xx, ww, b, shape, G, mask_dx=True)),
                     ("dense", lambda xx, ww: ops.conv2d(xx, ww, b, shape, 3, 1, mask_dx=True))):
        xx = x.clone().requires_grad_(True)
        ww = ws.clone().requires_grad_(True)
        for _ in range(3):
            y = fn(xx, ww)
            y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        gy = torch.ones_like(y)
        for _ in range(10):
            e[0].record(); y = fn(xx, ww); e[1].record(); y.backward(gy); e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        print("%dx%dx%d %-8s fwd %.3f ms  bwd %.3f ms" % (n, h, w, name, tf / 10, tb / 10))
