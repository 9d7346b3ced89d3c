"""where the three-piece weight gradient's distance from fp64 comes from: split-K count sweep, signed bias."""
import numpy as np
import torch
from scan_amd import _lib, ops

dev = torch.device("cuda:0")
sizes, N, cin, cout = [(256, 512)], 4, 256, 256
shape = ops.PyramidShape(N, sizes)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((shape.rows, cin), device=dev, generator=g)
gy = torch.randn((shape.rows, cout), device=dev, generator=g)
w0 = (torch.randn((cout, cin, 3, 3), device=dev, generator=g) / 48).contiguous(memory_format=torch.channels_last)
rs = np.random.RandomState(0)
oi = torch.from_numpy(rs.choice(cout, 32, replace=False)).to(dev)
ci = torch.from_numpy(rs.choice(cin, 32, replace=False)).to(dev)
h, wd = sizes[0]
xs = x[:, ci].double().view(N, h, wd, 32)
gs = gy[:, oi].double().view(N, h, wd, 32)
xp = torch.zeros((N, h + 2, wd + 2, 32), dtype=torch.float64, device=dev)
xp[:, 1:-1, 1:-1] = xs
ref = torch.zeros((32, 32, 3, 3), dtype=torch.float64, device=dev)
for ky in range(3):
    for kx in range(3):
        ref[:, :, ky, kx] = torch.einsum("nyxo,nyxc->oc", gs, xp[:, ky:ky + h, kx:kx + wd])


def run(mode, **tune):
    ops.CONV_MODE = mode
    olds = {k: _lib.query("scan_tune", k.encode(), v) for k, v in tune.items()}
    try:
        w = w0.clone().requires_grad_(True)
        ops.conv2d(x, w, None, shape, 3, 1).backward(gy)
        d = (w.grad.double()[oi][:, ci] - ref)
        m = float(ref.abs().max())
        bias = float((d * ref.sign()).mean()) / m
        return float(d.abs().max()) / m, float((d ** 2).mean() ** 0.5) / m, bias
    finally:
        for k, v in olds.items():
            _lib.query("scan_tune", k.encode(), v)


print("fp32-MFMA", run("fp32"))
for mode in ("bf16x6", "bf16x3"):
    for wgs in (768, 1536, 3072, 6144, 12288):
        for v6 in (1, 0):
            print(mode, "wgs", wgs, "v6", v6, "max %.3e rms %.3e bias-toward-sign %.3e" % run(mode, wgrad_wgs=wgs, wgrad_v6=v6))
