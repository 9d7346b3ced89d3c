"""inference-only loop for rocprofv3 --kernel-trace --stats: 2 frames of 1024x2048, precision mode, 10 timed batches"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan_amd import engine, synth
dev = torch.device('cuda')
cfg = engine.CONFIGS["c2f"]
model = engine.build_model(device=dev, settings=cfg)
engine.load_procedural_weights(model, cfg["num_classes"], cfg["conv_body"])
imgs = synth.synth_images(2, 1024, 2048, 5).to(dev)
for _ in range(3): res = engine.inference(model, imgs)
torch.cuda.synchronize(); t0 = time.time()
n = int(os.environ.get("N", 10))
for _ in range(n): res = engine.inference(model, imgs)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print("inference 2 frames: %.2f ms/batch, detections %s" % (dt * 1e3, [len(r[0]) for r in res]))
if os.environ.get("HOSTPROF"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): res = engine.inference(model, imgs)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
