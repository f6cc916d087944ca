#!/usr/bin/env python3
"""Timing of the 5x5 max-pool kernels (csrc/td_maxpool.hip) on the four CRP-stage maps of the BASELINE config, through the C ABI,
with the bytes each launch has to move (in + out + 1-byte offsets forward; grad_out + offsets + grad_in backward).
   python tools/maxpool_bench.py [--json out.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tripled_amd  # noqa: F401,E402
from tripled_amd import native  # noqa: E402
from tripled_amd.ops import _raw  # noqa: E402

SHAPES = [(12, 256, 48, 160), (12, 256, 24, 80), (12, 256, 12, 40), (12, 256, 6, 20), (4, 256, 80, 256)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    lib = native.load()
    st = native.stream()
    out = []
    for (N, C, H, W) in SHAPES:
        x = torch.randn(N, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(x)
        idx = torch.empty(N, H, W, C, device="cuda", dtype=torch.uint8)
        g = torch.randn_like(x)
        gi = torch.empty_like(x)
        code = native.DTYPE_CODES[x.dtype]
        t_f = timeit(lambda: native.check(lib.td_maxpool5_fwd(_raw(x), code, N, H, W, C, _raw(y), _raw(idx), st), "f"), args.iters)
        t_b = timeit(lambda: native.check(lib.td_maxpool5_bwd(_raw(g), _raw(idx), code, N, H, W, C, _raw(gi), st), "b"), args.iters)
        nbytes = x.numel() * 5            # 2 + 2 + 1 bytes per element, either direction
        rec = dict(shape=[N, C, H, W], fwd_us=round(t_f, 2), bwd_us=round(t_b, 2), bytes=nbytes,
                   fwd_GBps=round(nbytes / t_f / 1e3, 1), bwd_GBps=round(nbytes / t_b / 1e3, 1))
        out.append(rec)
        print(json.dumps(rec))
    print("sum: fwd %.1f us, bwd %.1f us (x4 stages per step each)" % (sum(r["fwd_us"] for r in out[:4]), sum(r["bwd_us"] for r in out[:4])))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
