#!/usr/bin/env python3
"""Stand-alone timing of the hand-written loss kernels at the BASELINE size (B=12, 192x640),
through the C ABI.  Used under rocprofv3 (--kernel-trace / --pmc) and on its own.

  python tools/kernel_bench.py [--iters 20] [--B 12 --H 192 --W 640] [--only fwd,bwd]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import tripled_amd  # noqa: F401,E402
from tripled_amd import native  # noqa: E402
from mono.datasets.synthetic import synthetic_batch  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=12)
    ap.add_argument("--H", type=int, default=192)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--only", default="")
    ap.add_argument("--adversarial", action="store_true", help="i.i.d. random frames and large random poses")
    ap.add_argument("--flat-disp", action="store_true", help="constant disparity at every scale (separates the cost of the disparity's "
                                                             "detail -- gather coherence -- from the cost of its resolution)")
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else None
    lib = native.load()
    dev = torch.device("cuda", 0)
    B, H, W = args.B, args.H, args.W
    batch = synthetic_batch(B, H, W, seed=1000, device=dev)
    tgt = batch[("color", 0, 0)].contiguous()
    srcs = [batch[("color", -1, 0)].contiguous(), batch[("color", 1, 0)].contiguous()]
    if args.adversarial:
        tgt = torch.rand_like(tgt)
        srcs = [torch.rand_like(s) for s in srcs]
    n_src = 2
    from tripled_amd import ops
    frames = ops.pack_frames(tgt, srcs)                      # RGBX pixels: what the photometric kernels read
    tgt, srcs = frames.tgt, list(frames.srcs)
    tgt_p, sp_p = frames.tgt_planar, native.ptr_array(frames.srcs_planar)      # td_photo_bwd reads the NCHW frames
    invK = batch["inv_K"].contiguous()
    g = torch.Generator().manual_seed(5)
    Ts = []
    for _ in range(n_src):
        T = torch.eye(4).repeat(B, 1, 1)
        # realistic ego-motion: a few pixels of parallax at the depths below; adversarial: ~100 px
        T[:, :3, 3] = (0.5 if args.adversarial else 0.004) * torch.randn(B, 3, generator=g)
        Ts.append(T)
    P = torch.stack([torch.matmul(batch["K"].cpu(), T)[:, :3, :] for T in Ts], 0).contiguous().to(dev)
    idloss = torch.empty(B, H, W, n_src, device=dev)
    noise = torch.randn(n_src, B, H, W, device=dev)
    argmin = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
    coef = torch.empty(B, 9, H, W, device=dev)
    part = torch.empty(lib.td_photo_num_blocks(B, H, W), device=dev)
    d_up = torch.empty(n_src, B, H, W, device=dev)      # one plane per source frame (td_photo_bwd)
    dpp = torch.empty(lib.td_photo_bwd_num_blocks(B, H, W), n_src * 12, device=dev)
    dP = torch.empty(n_src, B, 3, 4, device=dev)
    gs = torch.ones(1, device=dev)
    loss = torch.empty(1, device=dev)
    st = native.stream()
    sp = native.ptr_array(srcs)
    px = B * H * W
    res = {}

    def want(k):
        return only is None or k in only

    if only is not None and "hbm" in only:
        # what a plain device-to-device copy reaches on this box, next to the 8 TB/s the rooflines are priced against
        # (SURVEY.md section 8d); 1 GiB each way, far past the Infinity Cache
        n = 1 << 30
        a8, b8 = torch.empty(n, device=dev, dtype=torch.uint8), torch.empty(n, device=dev, dtype=torch.uint8)
        us = timeit(lambda: b8.copy_(a8), args.iters)
        print(json.dumps({"d2d_copy_us": us, "bytes_moved": 2 * n, "GB_per_s": 2 * n / us / 1e3}))
        return
    if only is not None and "calib" in only:
        # calibration of the FETCH_SIZE / WRITE_SIZE counters for THIS access width: td_l1map_fwd on one-channel tensors
        # streams 2 x 4 B/pixel in and 4 B/pixel out, one dword per lane, far past the 256 MiB Infinity Cache
        n = 48 * 1024 * 1024
        pa, pb = torch.rand(1, 1, 4096, n // 4096, device=dev), torch.rand(1, 1, 4096, n // 4096, device=dev)
        po = torch.empty_like(pa)
        strides = native.strides_array(pa)
        res["calib_dword_stream"] = timeit(lambda: native.check(lib.td_l1map_fwd(
            native.ptr(pa), 0, strides, native.ptr(pb), 1, 1, 4096, n // 4096, 1.0, native.ptr(po), st), "calib"), args.iters)
        print(json.dumps({"us": res, "calib_read_bytes": 8 * n, "calib_write_bytes": 4 * n}))
        return
    if want("identity"):
        res["identity"] = timeit(lambda: native.check(lib.td_photo_identity(
            native.ptr(tgt_p), sp_p, n_src, B, H, W, native.ptr(idloss), native.ptr(tgt), sp, st), "id"), args.iters)
    else:
        native.check(lib.td_photo_identity(native.ptr(tgt_p), sp_p, n_src, B, H, W, native.ptr(idloss), native.ptr(tgt), sp, st), "id")
    for s in range(4):
        hs, ws = H >> (s + 1), W >> (s + 1)
        if args.adversarial:
            disp = (0.1 + 0.8 * torch.rand(B, 1, hs, ws, device=dev)).contiguous()
        else:   # spatially smooth disparity, like a decoder output
            low = torch.rand(B, 1, max(hs // 8, 2), max(ws // 8, 2), device=dev)
            disp = (0.3 + 0.4 * torch.nn.functional.interpolate(low, size=(hs, ws), mode="bilinear",
                                                                align_corners=False)).contiguous()
        if args.flat_disp:
            disp = torch.full_like(disp, 0.5)
        d_disp = torch.empty_like(disp)
        img = torch.rand(B, 3, hs, ws, device=dev)
        mean = torch.empty(B, device=dev)
        nsb = lib.td_smooth_num_blocks(B, hs, ws)
        sp6 = torch.empty(nsb, 6, device=dev)
        ghat = torch.empty(B, hs, ws, device=dev)
        dotp = torch.empty(nsb, device=dev)

        def fwd():
            native.check(lib.td_photo_fwd(native.ptr(tgt), sp, n_src, native.ptr(disp), native.ptr(P), native.ptr(invK),
                                          native.ptr(idloss), native.ptr(noise), B, H, W, hs, ws, 0.1, 100.0,
                                          native.ptr(argmin), None, None, native.ptr(part), native.ptr(coef), st), "fwd")

        def fwd_nocoef():      # inference / validation form: no SSIM-adjoint coefficient field
            native.check(lib.td_photo_fwd(native.ptr(tgt), sp, n_src, native.ptr(disp), native.ptr(P), native.ptr(invK),
                                          native.ptr(idloss), native.ptr(noise), B, H, W, hs, ws, 0.1, 100.0,
                                          native.ptr(argmin), None, None, native.ptr(part), None, st), "fwd")

        def fwd_nomask():      # no auto-mask terms (MODE 1), coefficients on
            native.check(lib.td_photo_fwd(native.ptr(tgt), sp, n_src, native.ptr(disp), native.ptr(P), native.ptr(invK),
                                          None, None, B, H, W, hs, ws, 0.1, 100.0,
                                          native.ptr(argmin), None, None, native.ptr(part), native.ptr(coef), st), "fwd")

        def bwd():
            native.check(lib.td_photo_bwd(native.ptr(tgt_p), sp_p, native.ptr(tgt), sp, n_src, native.ptr(disp), native.ptr(P), native.ptr(invK),
                                          native.ptr(argmin), native.ptr(coef), 1, native.ptr(gs), 1.0 / (px * 4), B, H, W, hs, ws,
                                          0.1, 100.0, native.ptr(d_up), native.ptr(dpp), st), "bwd")

        def adj():
            native.check(lib.td_upsample_adjoint_planes(native.ptr(d_up), n_src, B, H, W, hs, ws, native.ptr(d_disp), 0, st), "adj")

        def red():
            native.check(lib.td_reduce_dP(native.ptr(dpp), n_src, B, H, W, native.ptr(dP), st), "red")
            native.check(lib.td_sum_scaled(native.ptr(part), part.numel(), 1.0, native.ptr(loss), st), "sum")

        def smf():
            native.check(lib.td_smooth_fwd(native.ptr(disp), native.ptr(img), B, hs, ws, 1, native.ptr(mean),
                                           native.ptr(sp6), st), "smf")
            native.check(lib.td_smooth_finish(native.ptr(sp6), B, hs, ws, 1e-3, native.ptr(loss), st), "smfin")

        def smb():
            native.check(lib.td_smooth_bwd(native.ptr(disp), native.ptr(img), native.ptr(mean), B, hs, ws, 1,
                                           native.ptr(gs), 1e-3, native.ptr(ghat), native.ptr(dotp),
                                           native.ptr(d_disp), 0, st), "smb")

        fwd()
        for name, fn in (("fwd", fwd), ("fwd_nocoef", fwd_nocoef), ("fwd_nomask", fwd_nomask), ("bwd", bwd), ("adjoint", adj), ("reduce", red), ("smooth_fwd", smf),
                         ("smooth_bwd", smb)):
            if want(name):
                res["%s_s%d" % (name, s)] = timeit(fn, args.iters)
    f0 = (12 + 24 + 1 + 0.25) * px
    out = {"us": {k: round(v, 2) for k, v in res.items()}}
    if "fwd_s0" in res:
        out["fwd_s0_GBps"] = round(f0 / res["fwd_s0"] / 1e3, 1)
    if "bwd_s0" in res:
        out["bwd_s0_GBps"] = round((f0 + 0.25 * px) / res["bwd_s0"] / 1e3, 1)
    out["sum_us"] = round(sum(res.values()), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
