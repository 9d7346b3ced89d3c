import sys, time, torch
sys.path.insert(0, '/root/repo')
from scan_amd import engine, synth
dev = torch.device('cuda')
model = engine.build_model(9, device=dev); engine.load_procedural_weights(model)
tr = engine.Trainer(model)
H, W, B = 1024, 2048, 2
s = synth.synth_images(B, H, W, 1).to(dev); t = synth.synth_images(B, H, W, 2).to(dev)
tg = [(b.to(dev), l.to(dev)) for b, l in synth.synth_targets(B, H, W, 8, 12, 3)]
for _ in range(2): tr.step(s, tg, t)
torch.cuda.synchronize()
# pure host time per step when GPU is not the limiter: time the python side without sync
t0 = time.time(); tr.step(s, tg, t); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
print('host-side step %.1f ms, +sync %.1f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    tr.step(s, tg, t)
torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=50))
