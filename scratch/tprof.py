import sys, torch
sys.path.insert(0, '/root/repo')
from scan_amd import engine, synth
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
import os
cfg = engine.CONFIGS[os.environ.get("MODEL", "c2f")]
body = cfg.get("conv_body", "VGG-16-FPN-RETINANET")
model = engine.build_model(cfg["num_classes"], cfg["test_mode"], device=dev, transfer_cfg=cfg["transfer_cfg"], conv_body=body)
engine.load_procedural_weights(model, cfg["num_classes"], body)
tr = engine.Trainer(model)
H, W, B = 1024, 2048, 2
s = synth.synth_images(B, H, W, 1).to(dev); t = synth.synth_images(B, H, W, 2).to(dev)
tg = [(b.to(dev), l.to(dev)) for b, l in synth.synth_targets(B, H, W, cfg['num_classes'] - 1, 12, 3)]
for _ in range(3): tr.step(s, tg, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(s, tg, t)
    torch.cuda.synchronize()
prof.export_chrome_trace('/root/repo/gpurun_out/trace.json')
