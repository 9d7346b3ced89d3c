"""conditioned-kernel generator: fused launches (csrc/condrnn.hip) vs the torch loop, forward + backward, microseconds"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan_amd import ops
from scan_amd.modeling import condgraph
dev = torch.device("cuda")
m = condgraph.GRAPHModule(256, 9).to(dev)
dk = torch.randn(9, 256, device=dev)
for fused in (True, False, True, False):
    ops.COND_RNN_FUSED = fused
    for _ in range(5):
        m.get_conded_weight().backward(dk)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50):
        k = m.get_conded_weight()
    torch.cuda.synchronize(); t1 = time.time()
    for _ in range(50):
        m.get_conded_weight().backward(dk)
    torch.cuda.synchronize(); t2 = time.time()
    print("fused" if fused else "torch", "forward %.0f us, forward + backward %.0f us" % ((t1 - t0) / 50 * 1e6, (t2 - t1) / 50 * 1e6))
