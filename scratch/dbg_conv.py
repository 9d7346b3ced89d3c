import sys, torch, torch.nn.functional as F, numpy as np
sys.path.insert(0, '/root/repo')
from scan_amd import ops
dev = torch.device('cuda')
def run(sizes, N, cin, cout, relu):
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(N, cin, h, w, generator=g) for h, w in sizes]
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    xr = [x.clone().requires_grad_(True) for x in xs]
    wr = wgt.clone().requires_grad_(True)
    yr = [F.conv2d(x, wr, None, padding=1) for x in xr]
    if relu: yr = [F.relu(y) for y in yr]
    gys = [torch.randn(y.shape, generator=g) for y in yr]
    sum((y * gy).sum() for y, gy in zip(yr, gys)).backward()
    cs = ops.pad4(cin)
    rows = []; sz = []
    for x in xs:
        r, s = ops.nchw_to_rows(x.to(dev), cs); rows.append(r); sz.append(s.sizes[0])
    rows = torch.cat(rows, 0).contiguous().requires_grad_(True)
    shape = ops.PyramidShape(N, sz)
    wd = wgt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.conv2d(rows, wd, None, shape, 3, 1, relu=relu)
    gyr = torch.cat([ops.nchw_to_rows(gy.to(dev), y.shape[1])[0] for gy in gys], 0)
    y.backward(gyr)
    for l in range(len(sizes)):
        dx = ops.rows_to_nchw(rows.grad, shape, l, cin).cpu()
        err = (dx - xr[l].grad).abs()
        bad = err > 1e-3
        print(sizes, N, cin, cout, relu, 'level', l, 'max err %.3e' % err.max().item(), 'bad', int(bad.sum()),
              'bad channels', sorted(set(bad.nonzero()[:, 1].tolist()))[:12], 'bad ys', sorted(set(bad.nonzero()[:, 2].tolist())), 'bad xs', sorted(set(bad.nonzero()[:, 3].tolist())))
run([(8,16)], 1, 265, 256, False)
run([(8,16)], 1, 265, 256, True)
run([(8,16),(4,8)], 2, 265, 256, True)
run([(8,16),(4,8)], 2, 264, 256, True)
run([(8,16),(4,8)], 2, 256, 256, True)
run([(8,16),(4,8)], 2, 256, 256, False)
