import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan_amd import ops, _lib
dev = torch.device("cuda", 0)
G = 8
shape = ops.PyramidShape(4, [(128, 256)])
x = torch.relu(torch.randn(shape.rows, 1024, device=dev))
dy = torch.randn(shape.rows, 8, device=dev)
dw = torch.zeros(8, 9, 1024, device=dev)
ws = torch.empty((_lib.query("scan_gconv3x3_to1_ws_floats", shape.ref(), 8, 128),), device=dev)
P = ops._ptr
for dbg in (0, 100, 200):
    _lib.query("scan_tune", b"gconv_dbg", dbg)
    for blocks_note in (0,):
        for _ in range(3):
            _lib.call("scan_gconv3x3_to1_wgrad", P(x), P(dy), 8, shape.ref(), 8, 128, P(dw), 0, P(ws), ops._stream())
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            _lib.call("scan_gconv3x3_to1_wgrad", P(x), P(dy), 8, shape.ref(), 8, 128, P(dw), 0, P(ws), ops._stream())
        e.record(); torch.cuda.synchronize()
        print("dbg", dbg, "wgrad total %.1f us" % (s.elapsed_time(e) * 100))
