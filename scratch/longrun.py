import sys, time, torch
sys.path.insert(0, '/root/repo')
from scan_amd import engine, synth
dev = torch.device('cuda')
model = engine.build_model(9, device=dev); engine.load_procedural_weights(model)
tr = engine.Trainer(model)
H, W, B = 512, 1024, 2
for it in range(60):
    s = synth.synth_images(B, H, W, 1000 + 7 * it).to(dev); t = synth.synth_images(B, H, W, 5000 + 7 * it).to(dev)
    tg = synth.synth_targets(B, H, W, 8, 12, 9000 + it)
    l = tr.step(s, tg, t, forward_target=(it >= 40))
    if it % 5 == 0 or it >= 56:
        tot = float(sum(l.values()))
        print(it, "total %.4f" % tot, {k: round(float(v), 4) for k, v in l.items() if k in ("loss_cls_gs", "loss_reg_gs", "node_loss_gs", "act_loss_gs", "loss_adv_P3_CON_ds", "loss_adv_P3_CON_dt", "consistency_loss_gt")})
        assert tot == tot and abs(tot) < 1e4
print("ok; peak mem GB", torch.cuda.max_memory_allocated() / 2**30)
