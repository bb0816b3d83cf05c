#!/usr/bin/env python3
"""Library bf16 GEMM rate on this box (torch.matmul -> hipBLASLt/rocBLAS), random data, for calibrating the
2.5 PFLOP/s vendor peak the roofline divides by ("record the measured hipBLASLt bf16 GEMM peak").
Shapes: the head-tower GEMM the implicit-GEMM conv replaces (M = pixels, N = 256 cout, K = 2304) as an
EXPLICIT GEMM on an already-materialised im2col matrix (which the conv kernel never builds), and large
square GEMMs.  Prints one JSON line."""
import json
import torch

def rate(m, n, k, iters=20):
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(k, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        (a @ b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return {"m": m, "n": n, "k": k, "ms": round(ms, 4), "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}

if __name__ == "__main__":
    out = [rate(436480, 256, 2304), rate(1745920 // 4, 256, 2304), rate(8192, 8192, 8192), rate(16384, 16384, 4096),
           rate(4096, 4096, 4096)]
    print(json.dumps({"library_bf16_gemm": out, "torch": torch.__version__}))
