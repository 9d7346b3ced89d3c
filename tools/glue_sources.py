"""Which source lines launch the small PyTorch kernels of a training step (copies, fills, adds, cats ...).

torch.profiler with python stacks over one steady-state step; every aten operator is attributed to the innermost
scan_amd frame on its stack.  Output: per (operator, source line) the calls per step and the device time, largest first.

    python tools/glue_sources.py > gpurun_out/glue_sources.txt
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
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
    tg = synth.synth_targets(B, H, W, mcfg["num_classes"] - 1, 12, 4321)  # host tensors, as the collator delivers them
    for _ in range(3):
        trainer.step(imgs_s, tg, imgs_t)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        trainer.step(imgs_s, tg, imgs_t)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, "self_device_time_total", None)
        if dt is None:
            dt = getattr(ev, "self_cuda_time_total", 0)
        if not dt or not ev.name.startswith("aten::"):
            continue
        where = "?"
        for fr in (ev.stack or []):
            if "scan_amd" in fr and "site-packages" not in fr:
                where = fr.split("scan_amd/")[-1]
                break
        else:
            for fr in (ev.stack or []):
                if "autograd" in fr or "backward" in fr:
                    where = "<autograd engine>"
                    break
        if where in ("?", "<autograd engine>") or (os.environ.get("GLUE_SHAPES") and dt > 20):
            shapes = getattr(ev, "input_shapes", None)
            where += " " + str(shapes)[:90]
        rec = agg[(ev.name, where)]
        rec[0] += 1
        rec[1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print("aten operators with device time in one step: %d calls, %.2f ms device time" % (sum(v[0] for _, v in rows), tot / 1e3))
    for (name, where), (n, t) in rows[:200]:
        print("%5d calls %8.1f us  %-28s %s" % (n, t, name, where))


if __name__ == "__main__":
    main()
