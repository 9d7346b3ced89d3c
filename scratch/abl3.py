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
for name, shape in [('tower N2', ops.PyramidShape(2, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)])), ('256x512x2', ops.PyramidShape(2, [(256, 512)]))]:
    x = torch.randn(shape.rows, 256, device=dev)
    fl = 2.0 * shape.rows * 256 * 2304
    for rnd in range(3):
        for pre in (1, 0):
            if pre: os.environ['SCAN_FWD_PRE'] = '1'
            else: os.environ.pop('SCAN_FWD_PRE', None)
            ms = timeit(lambda: ops.conv2d(x, w, None, shape))
            print(name, 'pre ' if pre else 'base', '%.3f ms %.1f TF' % (ms, fl / ms / 1e9))
