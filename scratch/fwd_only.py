import os, sys, torch
sys.path.insert(0, '/root/repo')
from scan_amd import ops
from scan_amd._lib import call, query
from scan_amd.ops import _ptr, _stream
dev = torch.device('cuda')
shape = ops.PyramidShape(2, [(128, 256), (64, 128), (32, 64), (16, 32), (8, 16)])
x = torch.randn(shape.rows, 256, device=dev); dy = torch.randn(shape.rows, 256, device=dev)
w = (torch.randn(256, 256, 3, 3, device=dev) / 48).contiguous(memory_format=torch.channels_last)
dw = torch.empty((256, 9, 256), device=dev); db = torch.empty((256,), device=dev)
ws = x.new_empty((query("scan_conv3x3_wgrad_bf16x3_ws_floats", shape.ref(), 256, 256),))
for _ in range(5):
    ops.conv2d(x, w, None, shape)
    call("scan_conv3x3_wgrad_bf16x3", _ptr(x), shape.ref(), 256, _ptr(dy), 256, 256, _ptr(dw), _ptr(db), 0, _ptr(ws), _stream())
torch.cuda.synchronize()
