"""Where the data-parallel path spends its extra time: A/B of Trainer.distributed on/off inside ONE process with
a one-rank RCCL group (the collectives move nothing, so what is left is enqueue cost, stream waits and the
scale kernel), plus the host cost of enqueuing each of the three all-reduce ranges.

    RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29540 python tools/dp_overhead.py
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29540")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    from scan_amd import engine, synth
    mcfg = engine.CONFIGS["c2f"]
    model = engine.build_model(device=dev, settings=mcfg)
    engine.load_procedural_weights(model, mcfg["num_classes"], mcfg["conv_body"])
    trainer = engine.Trainer(model, settings=mcfg, distributed=True)
    H, W, B = 1024, 2048, 2
    imgs_s = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 1234)], 32)
    imgs_t = engine.to_image_list([t.to(dev) for t in synth.synth_image_list([(H, W)] * B, 2234)], 32)
    tg = synth.synth_targets(B, H, W, mcfg["num_classes"] - 1, 12, 4321)  # host tensors, as the collator delivers them

    def run(flag, steps):
        trainer.distributed = flag
        torch.cuda.synchronize()
        t0 = time.time()
        host = 0.0
        for _ in range(steps):
            h0 = time.time()
            trainer.step(imgs_s, tg, imgs_t)
            host += time.time() - h0
        torch.cuda.synchronize()
        return (time.time() - t0) / steps * 1e3, host / steps * 1e3

    run(True, 2)
    run(False, 2)
    out = {"dp_on_ms": [], "dp_off_ms": [], "dp_on_host_ms": [], "dp_off_host_ms": []}
    for _ in range(a.rounds):
        for flag, key in ((True, "dp_on"), (False, "dp_off")):
            ms, host = run(flag, a.steps)
            out[key + "_ms"].append(round(ms, 2))
            out[key + "_host_ms"].append(round(host, 2))
    # host cost of one enqueue per range (GPU idle, so this is pure enqueue + stream bookkeeping)
    enq = {}
    for name, keys in (("fcos", ["fcos"]), ("dis", [k for k in trainer.groups if k.startswith("dis_")]),
                       ("rest", [k for k in trainer.groups if k != "fcos" and not k.startswith("dis_")])):
        lo = min(trainer.arena_range[k][0] for k in keys)
        hi = max(trainer.arena_range[k][1] for k in keys)
        g = trainer.grad_arena[lo:hi]
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.time()
            w = dist.all_reduce(g, async_op=True)
            t1 = time.time()
            w.wait()
            torch.cuda.synchronize()
            ts.append([round((t1 - t0) * 1e3, 3), round((time.time() - t0) * 1e3, 3)])
        enq[name] = {"mbytes": round((hi - lo) * 4 / 1e6, 1), "enqueue_ms,total_ms": ts}
    out["allreduce_one_rank"] = enq
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
