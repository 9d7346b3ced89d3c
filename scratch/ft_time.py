import sys, time, torch
sys.path.insert(0, '/root/repo')
from scan_amd import engine, synth
from scan_amd.modeling import condgraph
dev = torch.device('cuda')
model = engine.build_model(9, device=dev, attn_dropout=0.0); engine.load_procedural_weights(model)
tr = engine.Trainer(model)
import os
H, W, N = (1024, 2048, 2) if os.environ.get('FULL') else (256, 512, 2)
s = synth.synth_images(N, H, W, 1234).to(dev); t = synth.synth_images(N, H, W, 2234).to(dev)
tg = synth.synth_targets(N, H, W, 8, 12, 4321)
from scan_amd import ops as _ops
_orig_db = _ops.dbscan_in_cluster0
def _db(pts, eps, ms=5):
    import time as _t
    n, d = pts.shape
    nbytes = _ops.query("scan_dbscan_ws_bytes", n)
    ws = torch.empty((nbytes // 8 + 1,), dtype=torch.float64, device=pts.device)
    info = torch.empty((2,), dtype=torch.int32, device=pts.device)
    st = _ops._stream()
    torch.cuda.synchronize(); t0 = _t.time()
    _ops.call("scan_dbscan_prepare", _ops._ptr(pts), n, d, float(eps), int(ms), _ops._ptr(ws), _ops._ptr(info), st)
    torch.cuda.synchronize(); t1 = _t.time()
    first = int(info[0].item())
    changed = torch.zeros((1,), dtype=torch.int32, device=pts.device)
    parity = 0; it = 0
    if first < n:
        while True:
            _ops.call("scan_dbscan_bfs_step", n, _ops._ptr(ws), parity, _ops._ptr(changed), st)
            it += 1
            if int(changed.item()) == 0: break
            parity ^= 1
    torch.cuda.synchronize(); t2 = _t.time()
    out = torch.empty((n,), dtype=torch.uint8, device=pts.device)
    _ops.call("scan_dbscan_finish", n, _ops._ptr(ws), _ops._ptr(out), st)
    torch.cuda.synchronize(); t3 = _t.time()
    print("      dbscan n=%d: prepare %.1f ms, bfs %d levels %.1f ms, finish %.1f ms, in0=%d" % (n, (t1-t0)*1e3, it, (t2-t1)*1e3, (t3-t2)*1e3, int(out.sum())))
    return out.bool() if first < n else torch.zeros((n,), dtype=torch.bool, device=pts.device)
_ops.dbscan_in_cluster0 = _db
orig = condgraph.dbscan_positive_rows
acc = {"t": 0.0, "n": 0, "pts": 0}
def timed(feat_l, act_l, n_images, eps, thr):
    torch.cuda.synchronize(); t0 = time.time()
    r = orig(feat_l, act_l, n_images, eps, thr)
    torch.cuda.synchronize(); acc["t"] += time.time() - t0; acc["n"] += 1
    npts = int((act_l[:, 1:] > thr).sum())
    acc["pts"] += npts
    print("   level with %d rows: %d points, %.1f ms" % (feat_l.shape[0], npts, (time.time() - t0) * 1e3))
    return r
condgraph.dbscan_positive_rows = timed
for ft in (False, True, True):
    acc.update(t=0.0, n=0, pts=0)
    torch.cuda.synchronize(); t0 = time.time()
    tr.step(s, tg, t, forward_target=ft)
    torch.cuda.synchronize()
    print("forward_target=%s: step %.1f ms; DBSCAN %.1f ms over %d levels, %d points" % (ft, (time.time() - t0) * 1e3, acc["t"] * 1e3, acc["n"], acc["pts"]))
