import sys, torch
sys.path.insert(0, '/root/repo')
from scan_amd import engine, synth
dev = torch.device('cuda')
def run(overlap):
    model = engine.build_model(9, device=dev, attn_dropout=0.0); engine.load_procedural_weights(model)
    tr = engine.Trainer(model)
    if not overlap:
        tr.dis_streams = {}; tr.overlap_target = False
    s = synth.synth_images(2, 256, 512, 11).to(dev); t = synth.synth_images(2, 256, 512, 12).to(dev)
    tg = synth.synth_targets(2, 256, 512, 8, 8, 13)
    for _ in range(2): tr.step(s, tg, t)
    torch.cuda.synchronize()
    return {k: g.flat_p.clone() for k, g in tr.groups.items()}
runs = [("serial", run(False)), ("serial", run(False)), ("overlap", run(True)), ("overlap", run(True)), ("serial", run(False))]
for i in range(len(runs)):
    for j in range(i + 1, len(runs)):
        d = {k: float((runs[i][1][k] - runs[j][1][k]).abs().max()) for k in runs[i][1]}
        w = max(d, key=d.get)
        print("%s#%d vs %s#%d: worst %s %.3e" % (runs[i][0], i, runs[j][0], j, w, d[w]))
