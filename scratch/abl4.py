import os, sys, torch
sys.path.insert(0, '/root/repo')
from scan_amd import ops
dev = torch.device('cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
w = (torch.randn(256, 256, 3, 3, device=dev) / 48).contiguous(memory_format=torch.channels_last)
ops.CONV_MODE = "fp32"
for name, shape in [('tower N2', ops.PyramidShape(2, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)])), ('256x512x2', ops.PyramidShape(2, [(256, 512)]))]:
    x = torch.randn(shape.rows, 256, device=dev)
    fl = 2.0 * shape.rows * 256 * 2304
    ops.CONV_MODE = "fp32"; yref = ops.conv2d(x, w, None, shape); ops.CONV_MODE = "bf16x3"
    for rnd in range(2):
        for k in (0, 1):
            os.environ['SCAN_FWD_KERNEL_DYN'] = str(k)
            y = ops.conv2d(x, w, None, shape)
            err = (y - yref).abs().max().item() / yref.abs().max().item()
            ms = timeit(lambda: ops.conv2d(x, w, None, shape))
            print(name, 'kernel', k, '%.3f ms %.1f TF  relerr %.2e' % (ms, fl / ms / 1e9, err))
