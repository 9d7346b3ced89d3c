"""print the figures of a bench.py JSON line (stdin or file) in readable form"""
import json
import sys

d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
print("value %.3f %s   %.2f ms/step   n_gpus %d" % (d["value"], d["unit"].split(" (")[0], d["ms_per_step"], d["n_gpus"]))
r = d.get("roofline")
if r:
    print("dominant %s: %.1f TF/s of %.1f = %.3f, %d launches x %.4f ms; %s" % (
        r["kernel"], r["achieved"], r["peak"], r["frac"], r["launches"], r["avg_launch_ms"], r["measured"]))
    for k, v in sorted(r["all_conv_kernels"].items(), key=lambda kv: -kv[1]["share_of_serial_step"]):
        print("  %-34s %7.1f TF %8.4f ms x%4d  share %.3f" % (k, v["tflops"], v["avg_ms"], v["launches"], v["share_of_serial_step"]))
for k in ("strict_fp32", "bf16x3_two_piece", "three_phase_schedule", "inference", "dp1_nccl", "cpu_baseline"):
    if d.get(k):
        print(k, {a: b for a, b in d[k].items() if a not in ("dtype", "note", "sample", "thread_sweep_quarter_size_s")})
