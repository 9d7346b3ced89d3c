"""List the host<->device synchronisation points of one training step (torch.cuda.set_sync_debug_mode), with the
scan_amd source line that caused each one.  A synchronisation drains the launch queue: the kernels after it are
issued at the host's pace, so each one is a candidate GPU-idle gap (tools/gpu_idle.py measures them).

    python tools/sync_points.py [--forward-target]
"""
import argparse
import collections
import os
import sys
import traceback
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--forward-target", action="store_true")
    a = ap.parse_args()
    import torch
    from scan_amd import engine, synth
    dev = torch.device("cuda", 0)
    mcfg = engine.CONFIGS["c2f"]
    model = engine.build_model(device=dev, settings=mcfg)
    engine.load_procedural_weights(model, mcfg["num_classes"], mcfg["conv_body"])
    trainer = engine.Trainer(model, settings=mcfg)
    H, W, B = 1024, 2048, 2
    imgs_s = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 1234)], 32)
    imgs_t = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 2234)], 32)
    tg = synth.synth_targets(B, H, W, mcfg["num_classes"] - 1, 12, 4321)  # host tensors, as the collator delivers them
    for _ in range(2):
        trainer.step(imgs_s, tg, imgs_t, forward_target=a.forward_target)
    torch.cuda.synchronize()
    seen = collections.OrderedDict()

    def show(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" not in str(message):
            return
        where = [f for f in traceback.extract_stack() if "scan_amd" in f.filename]
        key = " <- ".join("%s:%d(%s)" % (os.path.relpath(f.filename, ROOT), f.lineno, f.name) for f in where[::-1][:3])
        seen[key] = seen.get(key, 0) + 1

    warnings.showwarning = show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    trainer.step(imgs_s, tg, imgs_t, forward_target=a.forward_target)
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    print("synchronisation points in one step: %d" % sum(seen.values()))
    for k, v in seen.items():
        print("%3d  %s" % (v, k))


if __name__ == "__main__":
    main()
