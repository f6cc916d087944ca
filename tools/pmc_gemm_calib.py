#!/usr/bin/env python3
"""A bf16 GEMM of known FLOP count, run under rocprofv3 --pmc to calibrate the MFMA-busy counter:
achieved FLOP/s from wall-clock vs SQ_VALU_MFMA_BUSY_CYCLES x 1024 FLOP (one 32x32x16 bf16 MFMA = 32768 FLOP in
32 busy cycles)."""
import time

import torch

n = 8192
a = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
b = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    c = a @ b
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    c = a @ b
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("GEMM %d^3 bf16: %.3f ms, %.1f TFLOP/s (wall, includes profiler serialisation)" % (n, dt * 1e3, 2 * n ** 3 / dt / 1e12))
