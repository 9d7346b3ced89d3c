import sys, time, torch
sys.path.insert(0, '/root/repo')
from scan_amd import engine, synth
dev = torch.device('cuda')
model = engine.build_model(9, device=dev); engine.load_procedural_weights(model)
tr = engine.Trainer(model)
H, W, B = 1024, 2048, 2
pool = [(synth.synth_images(B, H, W, 100 + i).to(dev), [(b.to(dev), l.to(dev)) for b, l in synth.synth_targets(B, H, W, 8, 12, 300 + i)],
         synth.synth_images(B, H, W, 200 + i).to(dev)) for i in range(4)]
t0 = time.time()
for it in range(200):
    s, tg, t = pool[it % 4]
    l = tr.step(s, list(tg), t)   # a fresh list object each step: the target plan is rebuilt
    if it % 40 == 0 or it == 199:
        torch.cuda.synchronize()
        print(it, "loss %.4f  alloc %.2f GB  reserved %.2f GB  %.1f ms/step" % (float(sum(l.values())), torch.cuda.memory_allocated() / 2**30,
              torch.cuda.memory_reserved() / 2**30, (time.time() - t0) / (it + 1) * 1e3))
print("peak GB", torch.cuda.max_memory_allocated() / 2**30)
