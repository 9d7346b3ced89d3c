import os, sys, torch
sys.path.insert(0, '/root/repo')
from scan_amd import ops
from scan_amd._lib import call, query
from scan_amd.ops import _ptr, _stream
dev = torch.device('cuda')
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
w = (torch.randn(256, 256, 3, 3, device=dev) / 48).contiguous(memory_format=torch.channels_last)
dw = torch.empty((256, 9, 256), device=dev); db = torch.empty((256,), device=dev)
for name, shape in [('64x128x1', ops.PyramidShape(1, [(64, 128)])), ('128x256x1', ops.PyramidShape(1, [(128, 256)])),
                    ('tower N2', ops.PyramidShape(2, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)])),
                    ('256x512x2', ops.PyramidShape(2, [(256, 512)])), ('512x1024x2', ops.PyramidShape(2, [(512, 1024)]))]:
    x = torch.randn(shape.rows, 256, device=dev); dy = torch.randn(shape.rows, 256, device=dev)
    fl = 2.0 * shape.rows * 256 * 2304
    ws = x.new_empty((query("scan_conv3x3_wgrad_bf16x3_ws_floats", shape.ref(), 256, 256),))
    ms = timeit(lambda: call("scan_conv3x3_wgrad_bf16x3", _ptr(x), shape.ref(), 256, _ptr(dy), 256, 256, _ptr(dw), _ptr(db), 0, _ptr(ws), _stream()))
    msf = timeit(lambda: ops.conv2d(x, w, None, shape))
    print('%-12s rows %8d  wgrad %.3f ms %.1f TF (ws %.0f MB) | fwd %.3f ms %.1f TF' % (name, shape.rows, ms, fl / ms / 1e9, ws.numel() * 4 / 1e6, msf, fl / msf / 1e9))
