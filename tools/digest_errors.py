"""Per-parameter gradient-digest errors of one DA iteration against a committed fixture (tests/golden/step_*.json), every
conv mode: where along the network the distance from the reference's gradients grows.  Diagnostic, GPU only.
    python tools/digest_errors.py [step_128x256|step_mid_512x1024] [modes]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from scan_amd import engine, ops, synth
    name = sys.argv[1] if len(sys.argv) > 1 else "step_128x256"
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp32", "bf16x6"]
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    H, W, N = gold["H"], gold["W"], gold["N"]
    dev = torch.device("cuda:0")
    rows = {}
    for mode in modes:
        ops.CONV_MODE = mode
        model = engine.build_model(9, device=dev, attn_dropout=0.0)
        engine.load_procedural_weights(model)
        trainer = engine.Trainer(model, base_lr=0.0)
        losses = trainer.step(synth.synth_images(N, H, W, gold["seeds"]["src"]).to(dev),
                              synth.synth_targets(N, H, W, 8, 12, gold["seeds"]["boxes"]),
                              synth.synth_images(N, H, W, gold["seeds"]["tgt"]).to(dev))
        torch.cuda.synchronize()
        print(mode, "losses rel err:", {k: "%.1e" % (abs(float(losses[k]) - v) / abs(v)) for k, v in gold["losses"].items() if v != 0})
        for mk, m in model.items():
            for pn, p in m.named_parameters():
                ref = gold["grad_digest"][mk].get(pn)
                if ref is None or p.grad is None or ref[1] / p.numel() < 1e-7:
                    continue
                g = p.grad.detach().double().reshape(-1)
                e_abs = abs(float(g.abs().sum()) - ref[1]) / max(ref[1], 1e-3)
                e_sum = abs(float(g.sum()) - ref[0]) / max(ref[1], 1e-3)
                rows.setdefault((mk, pn), {})[mode] = (e_abs, e_sum)
    print("%-60s " % "parameter" + "  ".join("%-22s" % (m + " |abs| / sum") for m in modes))
    for (mk, pn), r in rows.items():
        print("%-60s " % (mk + "/" + pn) + "  ".join("%.1e / %.1e      " % r[m] for m in modes))


if __name__ == "__main__":
    main()
