"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on the GEMMs the 3x3 layers are equivalent to (no halo, no epilogue,
no im2col: an upper reference for an LDS-staged MFMA kernel on this chip, random fp16 operands).  GPU box only."""
import statistics
import torch

dev = torch.device("cuda:0")
for name, M, N, K in [("L10 512->1024 @20 bs32", 12800, 1024, 4608), ("L8 256->512 @40 bs32", 51200, 512, 2304), ("L6 128->256 @80 bs32", 204800, 256, 1152),
                      ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    b = torch.randn(N, K, device=dev, dtype=torch.float16)
    for form, fn in (("A @ B^T", lambda: torch.matmul(a, b.t())), ("A @ B", lambda bt=b.t().contiguous(): torch.matmul(a, bt))):
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) * 1e3 / 10)
        med = statistics.median(ts)
        print(f"{name:26s} {form:8s} M={M} N={N} K={K}: {med:8.1f} us  {2.0 * M * N * K / med / 1e6:8.1f} TFLOP/s", flush=True)
