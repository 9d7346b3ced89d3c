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
