"""What the vendor GEMM (hipBLASLt through torch.matmul) reaches on the implicit-GEMM shapes of the conv layers:
a practical bf16 MFMA ceiling on this device for comparison with the bf16x3 kernels (3 bf16 MFMAs per product)."""
import time, torch
dev = torch.device('cuda')
def bench(M, K, N, dtype, iters=20):
    a = torch.randn(M, K, device=dev, dtype=dtype); b = torch.randn(K, N, device=dev, dtype=dtype)
    for _ in range(3): c = a @ b
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): c = a @ b
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms, 2.0 * M * K * N / ms / 1e9
for name, (M, K, N) in {"tower 256->256 (M=87296, K=2304)": (87296, 2304, 256), "conv4 512->512 (M=65536, K=4608)": (65536, 4608, 512),
                         "P3 256->256 (M=262144, K=2304)": (262144, 2304, 256), "big square 8192^3": (8192, 8192, 8192)}.items():
    ms, tf = bench(M, K, N, torch.bfloat16)
    print("%-36s bf16 hipBLASLt: %8.1f us  %7.1f TFLOP/s  (= %6.1f TF fp32-equivalent if it were a bf16x3 product)" % (name, ms * 1e3, tf, tf / 3))
ms, tf = bench(87296, 2304, 256, torch.float32, 5)
print("tower shape fp32 GEMM (rocBLAS/hipBLASLt): %.1f us %.1f TFLOP/s" % (ms * 1e3, tf))
