"""which autograd node (or python frame) issues the host synchronisations of one training step"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import ProfilerActivity, profile
from scan_amd import engine, synth
dev = torch.device("cuda", 0)
mcfg = engine.CONFIGS["c2f"]
model = engine.build_model(device=dev, settings=mcfg)
engine.load_procedural_weights(model, mcfg["num_classes"], mcfg["conv_body"])
trainer = engine.Trainer(model, settings=mcfg)
H, W, B = 1024, 2048, 2
imgs_s = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 1234)], 32)
imgs_t = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 2234)], 32)
tg = synth.synth_targets(B, H, W, mcfg["num_classes"] - 1, 12, 4321)
for _ in range(3):
    trainer.step(imgs_s, tg, imgs_t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.step(imgs_s, tg, imgs_t)
    torch.cuda.synchronize()
evs = prof.events()
names = ("aten::item", "aten::_local_scalar_dense", "aten::nonzero", "StreamSynchronize", "EventSynchronize", "DeviceSynchronize",
         "aten::is_nonzero", "hipMemcpy", "Memcpy")
hits = [e for e in evs if any(n.lower() in e.name.lower() for n in names)]
import collections
print(collections.Counter(e.name for e in evs if "hip" in e.name.lower() or "cuda" in e.name.lower()).most_common(12))
print(len(hits), "candidate events")
for e in hits:
    chain = []
    p = e.cpu_parent
    while p is not None and len(chain) < 6:
        chain.append(p.name)
        p = p.cpu_parent
    print("%-34s %8.1f us  thread %s  <- %s" % (e.name[:34], e.cpu_time_total, e.thread, " <- ".join(chain)))
