"""Host-side (Python) cost of one training step: cProfile over a few steady-state steps, top functions by own time.
The step is GPU-bound only while the host enqueues faster than the GPU executes (tools/dp_overhead.py prints both).

    python tools/host_profile.py [--steps 5]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
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
    for _ in range(3):
        trainer.step(imgs_s, tg, imgs_t)
    torch.cuda.synchronize()
    t0 = time.time()
    host = 0.0
    for _ in range(a.steps):
        h0 = time.time()
        trainer.step(imgs_s, tg, imgs_t)
        host += time.time() - h0
    torch.cuda.synchronize()
    print("unprofiled: %.2f ms/step wall, %.2f ms/step inside Trainer.step on the host" % (
        (time.time() - t0) / a.steps * 1e3, host / a.steps * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        trainer.step(imgs_s, tg, imgs_t)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats("tottime").print_stats(45)
    print(s.getvalue().replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", ""))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
    print(s.getvalue().replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", ""))


if __name__ == "__main__":
    main()
