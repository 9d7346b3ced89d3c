import sys, time, torch
sys.path.insert(0, '/root/repo')
from scan_amd import engine, synth, ops
dev = torch.device('cuda')
for mode in ("precision", "common"):
    model = engine.build_model(9, test_mode=mode, device=dev); engine.load_procedural_weights(model)
    imgs = synth.synth_images(2, 1024, 2048, 5).to(dev)
    for _ in range(2): res = engine.inference(model, imgs)
    torch.cuda.synchronize(); t0 = time.time()
    n = 5
    for _ in range(n): res = engine.inference(model, imgs)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(mode, "inference 2 frames 1024x2048: %.1f ms (%.1f frames/s), detections %s" % (dt * 1e3, 2 / dt, [len(r[0]) for r in res]))
# NMS kernel alone
for n in (1000, 4000, 8192):
    g = torch.Generator().manual_seed(n)
    xy = torch.rand(n, 2, generator=g) * 1000; wh = torch.rand(n, 2, generator=g) * 100 + 4
    boxes = torch.cat([xy, xy + wh], 1).to(dev); scores = torch.rand(n, generator=g).to(dev)
    for _ in range(3): k = ops.nms(boxes, scores, 0.6)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): k = ops.nms(boxes, scores, 0.6)
    torch.cuda.synchronize(); print("nms n=%d: %.1f us per call (incl. count readback), kept %d" % (n, (time.time() - t0) / 20 * 1e6, len(k)))
