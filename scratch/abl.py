import os, sys, torch
sys.path.insert(0, '/root/repo')
from scan_amd import ops
dev = torch.device('cuda')
N = 2
shape = ops.PyramidShape(N, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)])
x = torch.randn(shape.rows, 256, device=dev)
w = (torch.randn(256, 256, 3, 3, device=dev) / 48).contiguous(memory_format=torch.channels_last)
flops = 2.0 * shape.rows * 256 * 2304
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for rnd in range(2):
    for var in [0, 1, 2]:
        os.environ['SCAN_FWD_KERNEL_DYN'] = str(var)
        ms = timeit(lambda: ops.conv2d(x, w, None, shape))
        print('fwd kernel', var, '%.3f ms' % ms, '%.1f TF' % (flops / ms / 1e9))
# wgrad alone
from scan_amd._lib import call, query
from scan_amd.ops import _ptr, _stream
dy = torch.randn(shape.rows, 256, device=dev)
ws = x.new_empty((query("scan_conv3x3_wgrad_bf16x3_ws_floats", shape.ref(), 256, 256),))
dw = x.new_empty((256, 9, 256)); db = x.new_empty((256,))
ms = timeit(lambda: call("scan_conv3x3_wgrad_bf16x3", _ptr(x), shape.ref(), 256, _ptr(dy), 256, 256, _ptr(dw), _ptr(db), 0, _ptr(ws), _stream()))
print('wgrad tower %.3f ms %.1f TF' % (ms, flops / ms / 1e9))
# VGG stage 3 like: 2 x 256x512, 256->256
shape2 = ops.PyramidShape(2, [(256, 512)])
x2 = torch.randn(shape2.rows, 256, device=dev); dy2 = torch.randn(shape2.rows, 256, device=dev)
fl2 = 2.0 * shape2.rows * 256 * 2304
for var in [0, 1, 2]:
    os.environ['SCAN_FWD_KERNEL_DYN'] = str(var)
    ms = timeit(lambda: ops.conv2d(x2, w, None, shape2), 10)
    print('vgg3 fwd kernel', var, '%.3f ms %.1f TF' % (ms, fl2 / ms / 1e9))
ws2 = x.new_empty((query("scan_conv3x3_wgrad_bf16x3_ws_floats", shape2.ref(), 256, 256),))
ms = timeit(lambda: call("scan_conv3x3_wgrad_bf16x3", _ptr(x2), shape2.ref(), 256, _ptr(dy2), 256, 256, _ptr(dw), _ptr(db), 0, _ptr(ws2), _stream()), 10)
print('wgrad vgg3 %.3f ms %.1f TF' % (ms, fl2 / ms / 1e9))
