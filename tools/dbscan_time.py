"""prepare (norms + pairwise-distance GEMM + core test) time of scan_dbscan_* for random 256-d points, bf16x3 vs fp32 GEMM,
and the membership of both against each other"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan_amd import _lib, ops
dev = torch.device('cuda')
for n, spread in ((32768, 0.12), (131072, 0.12), (131072, 0.2)):
    g = torch.Generator().manual_seed(n)
    centers = torch.randn(64, 256, generator=g) * 2
    pts = (centers[torch.randint(0, 64, (n,), generator=g)] + torch.randn(n, 256, generator=g) * spread).to(dev)
    res = {}
    for mode in (1, 0, 1, 0):
        old = _lib.query("scan_tune", b"dbscan_bf16x3", mode)
        try:
            torch.cuda.synchronize(); t0 = time.time()
            m = ops.dbscan_in_cluster0(pts, 3.0, 5)
            torch.cuda.synchronize(); dt = time.time() - t0
        finally:
            _lib.query("scan_tune", b"dbscan_bf16x3", old)
        print("n=%d spread %.2f gemm=%s: %.1f ms, in cluster 0: %d" % (n, spread, "bf16x3" if mode else "fp32", dt * 1e3, int(m.sum())))
        res[mode] = m
    assert torch.equal(res[0], res[1])
