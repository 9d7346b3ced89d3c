"""CPU tests of what N > 1 adds outside the kernels: bench.py's launcher (`--gpus N` must start N ranks or fail) and the
collectives of scan_amd/comm.py (loss-scalar reduce, detection gather, global batch split) on two gloo ranks."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True,
                          text=True, timeout=600)


def test_launcher_starts_the_ranks_it_was_asked_for():
    r = _bench("--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"launch_check": True, "n_gpus": 2, "ranks_in_process_group": 2}


def test_launcher_refuses_fewer_devices_than_ranks():
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "refusing to measure fewer ranks" in (r.stderr + r.stdout)


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, PYTHONPATH=ROOT, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scan_amd import comm
    losses = {"loss_cls_gs": torch.tensor(1.0 + rank), "act_loss_gs": torch.tensor(10.0 * (rank + 1)),
              "zero_gt": torch.tensor(0.0)}
    red = comm.reduce_loss_dict(losses)
    # rank r holds images 2r, 2r+1; image 2r+1 has no detection on rank 1
    res, ids = [], [2 * rank, 2 * rank + 1]
    for j, iid in enumerate(ids):
        k = 0 if (rank == 1 and j == 1) else 3 + iid
        res.append((torch.full((k, 4), float(iid)), torch.full((k,), 0.5 + 0.01 * iid), torch.full((k,), iid + 1, dtype=torch.int64)))
    merged = comm.gather_detections(res, ids)
    out = {"red": {k: float(v) for k, v in red.items()}, "per_gpu": comm.images_per_gpu(16)}
    if merged is not None:
        out["merged"] = {i: (b.tolist(), s.tolist(), l.tolist()) for i, (b, s, l) in merged.items()}
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_loss_reduce_and_detection_gather_two_ranks():
    """reference engine/trainer.py:76-98 (rank 0 gets the average) and utils/comm.py:48-88 + engine/inference.py:40-58
    (rank 0 gets every rank's predictions keyed by image id, the others None)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["red"] == {"act_loss_gs": 15.0, "loss_cls_gs": 1.5, "zero_gt": 0.0}
    assert res[0]["per_gpu"] == res[1]["per_gpu"] == 8
    assert "merged" not in res[1]
    m = res[0]["merged"]
    assert sorted(m) == [0, 1, 2, 3]
    assert len(m[0][0]) == 3 and len(m[1][0]) == 4 and len(m[2][0]) == 5 and len(m[3][0]) == 0
    assert m[2][0][0] == [2.0] * 4 and m[2][2] == [3] * 5 and abs(m[1][1][0] - 0.51) < 1e-6


def test_single_process_comm_is_passthrough():
    from scan_amd import comm
    d = {"a": torch.tensor(2.0)}
    assert comm.reduce_loss_dict(d) is d and comm.get_world_size() == 1 and comm.is_main_process()
    out = comm.gather_detections([(torch.zeros(2, 4), torch.ones(2), torch.ones(2, dtype=torch.int64))], [7])
    assert list(out) == [7] and out[7][0].shape == (2, 4)
    with pytest.raises(ValueError):
        comm.images_per_gpu(16, 3)


def test_bench_symbols_and_peaks():
    """bench.py maps its timer records to the kernel symbols rocprofv3 lists and prices each against the ceiling of ITS
    arithmetic: 2.5 PFLOP/s / 6 for a three-piece (bf16x6) kernel, / 3 for a two-piece one, 157.3 for the fp32-MFMA kernels."""
    import bench
    cases = {
        "conv3x3_bf16x6_fwd_bn256": ("conv_split_kernel<3,256,16,512,3,1,true>", 2500.0 / 6),
        "conv3x3_bf16x6_dgrad_bn128": ("conv_split_kernel<3,128,16,512,3,1,true>", 2500.0 / 6),
        "conv3x3_bf16x6_fwd_bn1064": ("conv_split_kernel<3,64,16,512,3,1,true>", 2500.0 / 6),
        "conv3x3_bf16x6_fwd_bn2064": ("conv_split_kernel<3,64,32,512,3,1,true>", 2500.0 / 6),
        "conv3x3_bf16x6_fwd_bn64": ("conv_split_kernel<3,64,8,256,3,1,true>", 2500.0 / 6),
        "conv1x1_bf16x6_fwd_bn128": ("conv_split_kernel<3,128,16,512,1>", 2500.0 / 6),
        "conv3x3_bf16x6_wgrad": ("conv_wgrad_v6_kernel<3,32,3>", 2500.0 / 6),
        "conv1x1_bf16x6_wgrad": ("conv_wgrad_v4_kernel<3,1,S>", 2500.0 / 6),
        "conv_smallcin_bf16x6": ("conv_smallcin_kernel<3>", 2500.0 / 6),
        "conv3x3_bf16x3_fwd_bn2256": ("conv_split_kernel<2,256,16,512,3,1,true>", 2500.0 / 3),
        "conv3x3_bf16x3_dgrad_bn1128": ("conv_split_kernel<2,128,16,1024,3>", 2500.0 / 3),
        "conv3x3_bf16x3_wgrad": ("conv_wgrad_v6_kernel<2,64,3>", 2500.0 / 3),
        "conv_igemm_fwd": ("conv_igemm_kernel<0,4>", 157.3),
        "conv_wgrad": ("conv_wgrad_kernel", 157.3),
    }
    for rec, (sym, peak) in cases.items():
        assert bench.symbol_of(rec) == sym, (rec, bench.symbol_of(rec))
        assert abs(bench.peak_for(sym) - peak) < 1e-6, (sym, bench.peak_for(sym))
    assert bench.HEADLINE_MODE == "bf16x6"
    from scan_amd import ops
    assert ops.CONV_MODE == bench.HEADLINE_MODE  # the shipped default is what the headline is measured in


def _fake_sysfs(root, gpus):
    """gpus: [(render_minor, numa_node, cpulist)]; plus one CPU node (simd_count 0) in front, like a real host"""
    nodes = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    os.makedirs(os.path.join(nodes, "0"))
    open(os.path.join(nodes, "0", "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
    for i, (minor, numa, cpus) in enumerate(gpus, 1):
        os.makedirs(os.path.join(nodes, str(i)))
        open(os.path.join(nodes, str(i), "properties"), "w").write("cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % minor)
        dev = os.path.join(root, "class", "drm", "renderD%d" % minor, "device")
        os.makedirs(dev)
        open(os.path.join(dev, "numa_node"), "w").write("%d\n" % numa)
        open(os.path.join(dev, "local_cpulist"), "w").write(cpus + "\n")


def test_rank_cpu_placement_from_sysfs(tmp_path):
    """8 GPUs on two sockets (4 + 4), SMT siblings listed as a second range: every rank gets a disjoint share of ITS socket's
    CPUs; an affinity mask narrower than the socket is respected; unknown topology binds nothing."""
    from scan_amd import comm
    root = str(tmp_path)
    _fake_sysfs(root, [(128 + i, 0 if i < 4 else 1, "0-15,32-47" if i < 4 else "16-31,48-63") for i in range(8)])
    m = comm.gpu_numa_map(root)
    assert [g["numa_node"] for g in m] == [0, 0, 0, 0, 1, 1, 1, 1] and m[5]["render_minor"] == 133
    sets = [comm.rank_cpu_set(r, 8, root) for r in range(8)]
    assert all(s is not None and len(s["cpus"]) == 8 for s in sets)
    assert sets[0]["cpus"] == list(range(0, 8)) and sets[3]["cpus"] == list(range(40, 48))
    assert sets[4]["numa_node"] == 1 and sets[4]["cpus"] == list(range(16, 24))
    flat = [c for s in sets for c in s["cpus"]]
    assert len(flat) == len(set(flat)) == 64
    # two ranks only: each takes half of socket 0
    assert comm.rank_cpu_set(1, 2, root)["cpus"] == list(range(32, 48))
    # a cpuset of 8 CPUs (this container): shares are cut from what the process may use
    assert comm.rank_cpu_set(1, 4, root, allowed=set(range(8)))["cpus"] == [2, 3]
    assert comm.rank_cpu_set(5, 8, root, allowed=set(range(8))) is None  # none of socket 1's CPUs allowed: nothing bound
    assert comm.rank_cpu_set(0, 1, str(tmp_path / "missing")) is None and comm.gpu_numa_map(str(tmp_path / "missing")) == []
    assert comm.rank_cpu_set(9, 8, root) is None
