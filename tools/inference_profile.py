"""The inference leg of bench.py alone (engine.inference_stream / engine.inference on a batch of frames, weights static), for rocprofv3:

    rocprofv3 --kernel-trace --stats -d out --output-format csv -- python3 tools/inference_profile.py [--batch 2] [--iters 10]
    python tools/gpu_idle.py out/*/*_kernel_trace.csv

Prints ms per batch and images/s (wall clock over the timed iterations)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--loop", choices=("stream", "batch"), default="stream",
                    help="stream: engine.inference_stream (dataset loop, one batch of look-ahead); batch: inference() per batch")
    a = ap.parse_args()
    import torch
    from scan_amd import engine, synth
    dev = torch.device("cuda", 0)
    mcfg = engine.CONFIGS["c2f"]
    model = engine.build_model(device=dev, settings=mcfg)
    engine.load_procedural_weights(model, mcfg["num_classes"], mcfg["conv_body"])
    frames = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(a.height, a.width)] * a.batch, 2234)], 32)
    with torch.no_grad():
        for _ in range(3):
            engine.inference(model, frames)
        torch.cuda.synchronize()
        t0 = time.time()
        if a.loop == "stream":
            for dets in engine.inference_stream(model, (frames for _ in range(a.iters)), static_weights=True):
                pass
        else:
            for _ in range(a.iters):
                dets = engine.inference(model, frames, static_weights=True)
        torch.cuda.synchronize()
    dt = (time.time() - t0) / a.iters
    print("inference (%s): %.2f ms per batch of %d = %.1f images/s; detections %s" % (
        a.loop, dt * 1e3, a.batch, a.batch / dt, [int(len(d[0])) for d in dets]))


if __name__ == "__main__":
    main()
