#!/usr/bin/env python3
"""Benchmark of the BASELINE metric: training images/s of cfg_kitti_tripleD (ResNet50 depth +
feature nets, ResNet18 pose net, 192x640, 12 images per GPU) on 1..8 MI355X.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = forward (depth/pose/auto-encoder nets under bf16 autocast, channels_last; the loss hot
path in the hand-written fp32 HIP kernels) + backward + gradient sync (bucketed RCCL all-reduce
overlapped with backward when N > 1) + grad-clip + Adam, on synthetic frame triplets already
resident in HBM.  Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the field
definitions (roofline = the dominant hand-written kernel timed live with HIP events;
cpu_baseline = the same training step with the CPU oracle loss path on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# Which stream the step graph is captured on.  "side" makes the graph one linear chain;
# ROCm 7.2's AQL-packet-capture fast path for such graphs replays this 4 000-node step wrongly from the second
# replay on (tools/diag_capture.py), so "side" is only used together with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0,
# which must be in the environment before the HIP runtime initialises.
CAPTURE_STREAM = os.environ.get("TD_CAPTURE_STREAM", "side")
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import tripled_amd  # noqa: F401,E402
from mmcv import Config  # noqa: E402
from mono.datasets.synthetic import synthetic_batch  # noqa: E402
from mono.model import MONO  # noqa: E402
from tripled_amd import dispatch  # noqa: E402
from tripled_amd.step import NonFiniteLossError, TrainStep, capture_step, ranks_agree, replicas_agree, warm_up  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable
# algorithmic bytes per full-resolution pixel of one fused photometric launch at scale s
# (SURVEY.md section 8d): fwd reads target 12 B + two sources 24 B + disp 4/4^(s+1) B and writes
# the 1-byte arg-min; bwd re-reads the same, reads the arg-min and writes d(disp) 4/4^(s+1) B.
def photo_fwd_bytes_per_px(s):
    return 12 + 24 + 4.0 / 4 ** (s + 1) + 1


def photo_bwd_bytes_per_px(s):
    return 12 + 24 + 4.0 / 4 ** (s + 1) + 1 + 4.0 / 4 ** (s + 1)


PKG = os.path.join(ROOT, "tripled-exploring-depth-estimation-with-self-supervised-representation-learning_amd")
# what a committed PMC figure depends on: it is only quoted in the line while these sources (and the workload) are the ones
# it was measured on (tools/traffic_from_pmc.py and tools/mfma_util.py write the same stamp into the file)
TRAFFIC_SOURCES = ["csrc/td_photo_fwd.hip", "csrc/td_photo_bwd.hip", "csrc/td_common.h"]
MFMA_SOURCES = ["csrc/td_conv1x1.hip", "csrc/td_bn.hip", "hostside/mono/model/networks.py", "ops.py"]


def source_stamp(files):
    import hashlib
    h = hashlib.sha256()
    for rel in files:
        with open(os.path.join(PKG, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def stamped(path, files, workload):
    """The JSON at ``path`` if its ``_stamp`` matches the current sources and workload, else (None, reason)."""
    if not os.path.exists(path):
        return None, "no %s" % os.path.basename(path)
    with open(path) as fh:
        blob = json.load(fh)
    st = blob.get("_stamp") or {}
    if st.get("sources_sha16") != source_stamp(files):
        return None, "%s was measured on other kernel sources (stamp %s)" % (os.path.basename(path), st.get("sources_sha16"))
    if st.get("workload") != workload:
        return None, "%s was measured on %s, this run is %s" % (os.path.basename(path), st.get("workload"), workload)
    return blob, None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--config", default=os.path.join(ROOT, "config", "cfg_kitti_tripleD.py"))
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--no-graph", action="store_true", help="do not capture the step in a HIP graph")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--prune-extractor-tail", action="store_true",
                   help="skip the ResNet tails whose output the reference discards (net.py:221); NOT the headline configuration")
    p.add_argument("--h2d", nargs="?", const="float32", default=None, choices=["float32", "uint8"],
                   help="include the host-to-device copy of every batch in the timed step: the reference's float32 wire format "
                        "(two float copies per frame) or this build's uint8 format (bytes + device-side ToTensor / colour jitter)")
    p.add_argument("--split-timing", action="store_true", help="two graphs (fwd+bwd | clip+Adam) and report each")
    p.add_argument("--no-flat", action="store_true",
                   help="autocast + per-parameter fused Adam instead of the flat bf16/fp32 parameter store (tripled_amd/flat_amp.py; "
                        "the store is the trainer's default with amp='bf16', mono/apis/trainer.py)")
    p.add_argument("--frames", default="coherent", choices=["coherent", "iid"],
                   help="coherent: the three frames are shifted crops of one canvas (like driving data; the headline input); "
                        "iid: independent uniform noise per frame (the adversarial case of SURVEY.md section 8d)")
    p.add_argument("--no-roofline", action="store_true", help="skip the isolated kernel timing (profiling runs)")
    p.add_argument("--fp8", action="store_true", help="forward GEMM of the eligible 1x1 convolutions on the fp8 MFMA path "
                                                      "(BASELINE config 5; off by default: measured slower at these sizes, DESIGN.md)")
    p.add_argument("--allow-fallbacks", action="store_true",
                   help="do not fail when a HIP-resident tensor takes an ATen composition instead of a hand-written kernel")
    p.add_argument("--cpu-batch", type=int, default=2)
    p.add_argument("--cpu-steps", type=int, default=3)
    p.add_argument("--syncbn", default="config", choices=["config", "on", "off"],
                   help="N > 1: batch statistics over all ranks (the hand-written BatchNorm passes with one small all-reduce per "
                        "layer and direction) or local statistics; default: what the config says (cfg_kitti_tripleD: syncbn=True)")
    p.add_argument("--entry", default="both", choices=["step", "runner", "both"],
                   help="step: tripled_amd.step.TrainStep replayed by this script (`value`); runner: the same iteration through "
                        "train_mono / the mmcv Runner, i.e. what train.py executes (`runner_entry`); both (default)")
    p.add_argument("--grad-sync", default="auto", choices=["auto", "overlap-graph", "two-graph", "eager"],
                   help="N > 1 gradient exchange (auto: overlap-graph, falling back to two-graph, then eager)")
    p.add_argument("--miopen-find", default="config", choices=["config", "on", "off"],
                   help="torch.backends.cudnn.benchmark = MIOpen find mode; default: the config's cudnn_benchmark")
    return p.parse_args()


def build_model(cfg, device, channels_last):
    model = MONO.module_dict[cfg.model["name"]](cfg.model)
    model = model.to(device)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    model.train()
    return model


def time_kernel(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters * 1e-3     # seconds per launch


def roofline_of_hot_kernels(cfg, batch):
    """Time the hand-written photometric kernels in isolation on the benchmark's own inputs
    (launched on torch's current stream, bracketed by HIP events on that stream)."""
    from tripled_amd import native
    lib = native.load()
    m = cfg.model
    B, H, W = m["imgs_per_gpu"], m["height"], m["width"]
    dev = batch["K"].device
    from tripled_amd import ops
    frames = ops.pack_frames(batch[("color", 0, 0)], [batch[("color", f, 0)] for f in m["frame_ids"][1:]])     # RGBX pixels
    tgt, srcs = frames.tgt, list(frames.srcs)
    tgt_p, sp_p = frames.tgt_planar, native.ptr_array(frames.srcs_planar)      # td_photo_bwd reads the NCHW frames
    n_src = len(srcs)
    invK = batch["inv_K"].contiguous()
    g = torch.Generator(device="cpu").manual_seed(5)
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, :3, 3] = 0.004 * torch.randn(B, 3, generator=g)      # a few pixels of parallax, like real ego-motion
    P = torch.stack([torch.matmul(batch["K"].cpu(), T)[:, :3, :]] * n_src, 0).contiguous().to(dev)
    idloss = torch.empty(B, H, W, n_src, device=dev)
    noise = torch.randn(n_src, B, H, W, device=dev)
    argmin = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
    coef = torch.empty(B, 9, H, W, device=dev)
    part = torch.empty(lib.td_photo_num_blocks(B, H, W), device=dev)
    d_up = torch.empty(n_src, B, H, W, device=dev)      # one plane per source frame (td_photo_bwd)
    dpp = torch.empty(lib.td_photo_bwd_num_blocks(B, H, W), n_src * 12, device=dev)
    gs = torch.ones(1, device=dev)
    st = native.stream()
    sp = native.ptr_array(srcs)
    native.check(lib.td_photo_identity(native.ptr(tgt_p), sp_p, n_src, B, H, W, native.ptr(idloss), native.ptr(tgt), sp, st), "identity")
    out = {}
    px = B * H * W
    for s in (0,):
        hs, ws = H >> (s + 1), W >> (s + 1)
        low = torch.rand(B, 1, hs // 8, ws // 8, device=dev)
        disp = (0.3 + 0.4 * torch.nn.functional.interpolate(low, size=(hs, ws), mode="bilinear",
                                                            align_corners=False)).contiguous()

        def fwd():
            native.check(lib.td_photo_fwd(native.ptr(tgt), sp, n_src, native.ptr(disp), native.ptr(P), native.ptr(invK),
                                          native.ptr(idloss), native.ptr(noise), B, H, W, hs, ws, 0.1, 100.0,
                                          native.ptr(argmin), None, None, native.ptr(part), native.ptr(coef), st), "fwd")

        def bwd():
            native.check(lib.td_photo_bwd(native.ptr(tgt_p), sp_p, native.ptr(tgt), sp, n_src, native.ptr(disp), native.ptr(P), native.ptr(invK),
                                          native.ptr(argmin), native.ptr(coef), 1, native.ptr(gs), 1.0 / (px * 4), B, H, W, hs, ws,
                                          0.1, 100.0, native.ptr(d_up), native.ptr(dpp), st), "bwd")

        t_f, t_b = time_kernel(fwd), time_kernel(bwd)
        out["photo_fwd_s%d" % s] = dict(seconds=t_f, bytes=photo_fwd_bytes_per_px(s) * px)
        out["photo_bwd_s%d" % s] = dict(seconds=t_b, bytes=photo_bwd_bytes_per_px(s) * px)
    return out


def loss_path_time(cfg, batch, iters=20):
    """SURVEY section 8d roofline 1: the whole hand-written loss path of one step -- identity term once, then per
    scale photometric forward + backward and smoothness forward + backward -- captured in a HIP graph (so that
    host launch gaps do not count) and timed with HIP events on the replay stream.  Returns seconds per step."""
    from tripled_amd import ops
    m = cfg.model
    B, H, W = m["imgs_per_gpu"], m["height"], m["width"]
    dev = batch["K"].device
    scales = list(m.get("scales", [0, 1, 2, 3]))
    tgt = batch[("color", 0, 0)].contiguous()
    srcs = [batch[("color", f, 0)].contiguous() for f in m["frame_ids"][1:]]
    invK = batch["inv_K"].contiguous()
    g = torch.Generator(device="cpu").manual_seed(5)
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, :3, 3] = 0.004 * torch.randn(B, 3, generator=g)
    P = torch.stack([torch.matmul(batch["K"].cpu(), T)[:, :3, :]] * len(srcs), 0).contiguous().to(dev).requires_grad_(True)
    noise = torch.randn(len(scales), len(srcs), B, H, W, device=dev)
    disps = []
    for s in scales:
        hs, ws = H >> (s + 1), W >> (s + 1)
        low = torch.rand(B, 1, max(hs // 8, 1), max(ws // 8, 1), device=dev)
        disps.append((0.3 + 0.4 * torch.nn.functional.interpolate(low, size=(hs, ws), mode="bilinear",
                                                                  align_corners=False)).contiguous().requires_grad_(True))

    def path():
        for d in disps:
            d.grad = None
        P.grad = None
        frames = ops.pack_frames(tgt, srcs, pack=False)      # once per step, as HipLossBackend.begin_step does (RGBX by the identity kernel)
        idloss = ops.photo_identity(frames)
        total = 0.0
        for i, s in enumerate(scales):
            loss, _, _ = ops.photometric_scale_loss(disps[i], P, frames, None, invK, idloss, noise[i], 0.1, 100.0, len(scales))
            img = ops.area_downsample(tgt, H >> (s + 1), W >> (s + 1))
            total = total + loss + ops.smooth_loss(disps[i], img, True, 1e-3 / (2 ** s) / len(scales))
        total.backward()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        path()
        path()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side if CAPTURE_STREAM == "side" else None):
        path()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


# algorithmic bytes of the loss path per full-resolution pixel per step (4 scales of
# photometric fwd+bwd = 300 B, smoothness + image pyramid = 28 B) and algorithmic conv FLOPs per image
LOSS_PATH_BYTES_PER_PX = 328.0
CONV_GFLOP_PER_IMG = {(192, 640): 377.2, (320, 1024): 911.9}
MFMA_BF16_PEAK_TFLOPS = 2500.0


def runner_entry(args, batch, world, dev):
    """The same workload through the reference's entry path: mono.apis.train_mono -> mmcv Runner.run -> hooks ->
    tripled_amd.step.RunnerIteration (what ``train.py`` executes per batch; reference: train.py:70-124,
    mono/apis/trainer.py:147-189), on the same HBM-resident batch.  Returns ms per iteration over ``args.steps`` iterations
    after the eager + capture + health-check iterations."""
    import shutil
    import tempfile
    from mono.apis import train_mono
    from mono.datasets import ResidentBatches
    cfg = Config.fromfile(args.config)
    warm = max(args.warmup, int(cfg.get("graph_warmup_iters", 3)) + 4)
    work = tempfile.mkdtemp(prefix="td_bench_runner_")
    cfg.work_dir, cfg.gpus, cfg.total_epochs, cfg.validate = work, [0], 1, False
    cfg.checkpoint_config = dict(interval=-1)             # no 1.3 GB file at the end of the epoch
    cfg.log_config = dict(interval=10 ** 9, hooks=[dict(type="TextLoggerHook")])
    cfg.log_level = "WARNING"
    cfg.strict_dispatch = not args.allow_fallbacks
    if args.dtype == "fp32":
        cfg.amp = "fp32"
    if args.no_graph:
        cfg.hip_graph = False
    if world > 1:
        cfg.syncbn = bool(cfg.get("syncbn", False)) if args.syncbn == "config" else args.syncbn == "on"
    torch.manual_seed(1024)
    model = MONO.module_dict[cfg.model["name"]](cfg.model)
    data = ResidentBatches(batch, warm + args.steps, timed_from=warm)
    try:
        train_mono(model, data, None, cfg, distributed=world > 1, validate=False)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    elapsed = data.elapsed
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final = float(torch.stack([p.detach().float().abs().max() for p in model.parameters()]).max())
    if final != final or final == float("inf"):
        raise NonFiniteLossError("runner entry: a parameter is not finite after %d iterations" % (warm + args.steps))
    return elapsed / data.timed_iters * 1e3


def cpu_baseline(cfg_path, batch_size, steps):
    """The same training step on the host: fp32 networks + the CPU oracle loss path (a port of
    the reference's unfused PyTorch ops, pinned against the reference in tests/)."""
    from oracle.backend import OracleLossBackend
    # the GPU box gives one-GPU jobs a 16-core share; more threads than that only oversubscribes
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, int(os.environ.get("TD_CPU_THREADS", "16")))))
    cfg = Config.fromfile(cfg_path)
    gpu_batch = cfg.model["imgs_per_gpu"]
    cfg.model["imgs_per_gpu"] = batch_size
    torch.manual_seed(1024)
    model = build_model(cfg, torch.device("cpu"), channels_last=False)
    model.set_loss_backend(OracleLossBackend())
    batch = synthetic_batch(batch_size, cfg.model["height"], cfg.model["width"], seed=1000,
                            frame_ids=tuple(cfg.model["frame_ids"]))
    step = TrainStep(model, cfg, batch, None)
    step()                      # warm-up
    t0 = time.time()
    for _ in range(steps):
        step()
    dt = time.time() - t0
    return dict(value=round(batch_size * steps / dt, 4), unit="imgs/s", cores=torch.get_num_threads(), kind="port",
                sample="%s fp32 on CPU, B=%d images per step (a sample of the GPU leg's B=%d batch: NOT the same batch size), %dx%d, "
                       "1 warm-up + %d timed steps (fwd+bwd+clip+Adam), oracle loss path" % (
                           cfg.model["name"], batch_size, gpu_batch, cfg.model["height"], cfg.model["width"], steps))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the loss hot path has no CPU implementation in the product)")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        # RCCL ("nccl") over xGMI; TD_DIST_BACKEND=gloo lets the N > 1 path be rehearsed on a single GPU
        backend = os.environ.get("TD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    cfg = Config.fromfile(args.config)
    torch.backends.cudnn.benchmark = {"config": bool(cfg.get("cudnn_benchmark", False)), "on": True,
                                      "off": False}[args.miopen_find]
    m = cfg.model
    if args.prune_extractor_tail:
        m["prune_extractor_tail"] = True
    B, H, W = m["imgs_per_gpu"], m["height"], m["width"]
    torch.manual_seed(1024)
    from mono.model.networks import set_fp8_conv1x1
    set_fp8_conv1x1(args.fp8)
    model = build_model(cfg, dev, channels_last=True)
    use_syncbn = bool(cfg.get("syncbn", False)) if args.syncbn == "config" else args.syncbn == "on"
    dtype = torch.bfloat16 if args.dtype == "bf16" else None
    dp = world > 1 or os.environ.get("TD_FORCE_DP") == "1"      # TD_FORCE_DP: rehearse the N > 1 code path in a one-rank group
    if dp and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    batch = synthetic_batch(B, H, W, seed=1000 + rank, device=dev, frame_ids=tuple(m["frame_ids"]),
                            coherent=args.frames == "coherent")
    # the same isolated kernel timing as the `roofline` block, taken BEFORE the training run (idle GPU): reported next to the
    # after-run figure, which stays the one `roofline.frac` is computed from
    kern_before = None
    if rank == 0 and not args.no_roofline:
        try:
            kern_before = roofline_of_hot_kernels(cfg, batch)
        except Exception as e:      # noqa: BLE001 -- auxiliary figure
            kern_before = {"error": "%s: %s" % (type(e).__name__, e)}
    side = torch.cuda.Stream()
    cap = side if CAPTURE_STREAM == "side" else None
    dispatch.reset()
    dispatch.set_strict(not args.allow_fallbacks)    # a layer that would route a HIP tensor to an ATen composition raises

    # Gradient-exchange modes for N > 1, tried in this order; every rank must agree on the outcome:
    #   overlap-graph  the bucket engine's all-reduces are issued from autograd hooks while backward is still running
    #                  (RCCL on the process group's side stream) and the WHOLE step -- collectives included -- is one HIP graph
    #   two-graph      forward+backward graph | eager bucketed all-reduce of the flat gradient buffer | clip+Adam graph
    #   eager          bucket engine with overlapped all-reduces, no graph
    nccl = dp and dist.get_backend() == "nccl"
    lowp_store = dtype is not None and not args.no_flat
    if not dp:
        modes = ["single-flat" if lowp_store and not args.split_timing else "single"]
    elif args.no_graph:
        modes = ["eager"]
    elif args.grad_sync != "auto":
        modes = [args.grad_sync]
    else:
        # both graph forms are brought up, validated and timed for a few steps; the faster one runs the benchmark
        # (collectives can only be captured on RCCL: a gloo rehearsal skips the first form)
        modes = (["overlap-graph"] if nccl else []) + ["two-graph"]
    import copy
    base_model = model

    def bring_up(mode):
        """Build the step for ``mode`` on its own copy of the model; returns a dict or raises."""
        net = copy.deepcopy(base_model)
        wrapped, flat_kind, split = net, False, False
        if use_syncbn and dp:
            from mono.model.networks import enable_sync_batchnorm
            enable_sync_batchnorm(net, force=world == 1)
        if mode in ("overlap-graph", "eager"):
            from mmcv.parallel import MMDistributedDataParallel
            wrapped = MMDistributedDataParallel(net, device_ids=[dev.index], broadcast_buffers=False,
                                                find_unused_parameters=cfg.get("find_unused_parameters", False),
                                                overlap=True, engine_at_world_1=world == 1)
            wrapped.train()
        elif mode == "two-graph":
            # no wrapper: rank 0's weights to everyone, gradients gathered into one flat fp32 buffer after backward
            for t in list(net.parameters()) + list(net.buffers()):
                dist.broadcast(t.data, 0)
            flat_kind, split = ("lowp" if lowp_store else "fp32"), True
        elif mode == "single-flat":
            flat_kind = "lowp"
        elif mode == "single" and args.split_timing:
            flat_kind, split = "fp32", True
        st = TrainStep(wrapped, cfg, batch, dtype, flat=flat_kind)
        dispatch.hip_calls.clear()
        warm_up(st, max(args.warmup, 1), side)
        calls = sum(dispatch.hip_calls.values()) // max(args.warmup, 1)
        after_warmup = st.check_finite("warm-up")
        gs = None
        if mode != "eager" and not args.no_graph:
            if dp:      # the warm-up's collectives have completed and the ranks line up before capturing;
                dist.barrier()      # "thread_local": the process group's watchdog thread may query events meanwhile
                torch.cuda.synchronize()
                # ... and its work list is given time to drain (it polls every 100 ms): a hipEventQuery on a warm-up work's
                # event from the watchdog thread while this thread is capturing aborted one one-rank RCCL rehearsal in round 4
                time.sleep(0.5)
            # captured but NOT replayed yet: under N > 1 a replay executes collectives, so it must not start before every
            # rank is known to have captured successfully (main loop below)
            gs = capture_step(st, stream=cap, split=split, capture_error_mode="thread_local" if dp else "global", validate=False)
        return dict(mode=mode, model=net, step=st, graphed=gs, calls=calls, loss_after_warmup=after_warmup)

    def trial_ms(cand, iters=5):
        run_ = cand["graphed"] if cand["graphed"] is not None else cand["step"]
        if dp:
            dist.barrier()
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for _ in range(iters):
            run_()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0_) / iters * 1e3], device=dev, dtype=torch.float64)
        if dp:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    candidates, trials = [], {}
    for mode in modes:
        ok, cand = True, None
        try:
            cand = bring_up(mode)
        except NonFiniteLossError:
            raise
        except Exception as e:      # noqa: BLE001 -- this form is not available here
            ok = False
            print("rank %d: step mode %r unavailable (%s: %s)" % (rank, mode, type(e).__name__, str(e)[:300]), file=sys.stderr)
        if dp:
            ok = ranks_agree(ok, dev)          # every rank runs the same step form (tests/test_dp_gloo.py)
        if ok:
            if cand["graphed"] is not None:
                for i in range(2):      # validation replays (all ranks together): a captured step that is not finite is an error
                    cand["graphed"]()
                    torch.cuda.synchronize()
                    cand["step"].check_finite("replay %d of the captured step (%s)" % (i, mode))
            if len(modes) > 1:
                trials[mode] = round(trial_ms(cand), 3)
            candidates.append(cand)
    if not candidates and dp and "eager" not in modes:
        candidates.append(bring_up("eager"))
    if not candidates:
        raise SystemExit("bench.py: no step mode could be brought up")
    best = min(candidates, key=lambda c: trials.get(c["mode"], 0.0))
    used_mode, model, step, graphed_step = best["mode"], best["model"], best["step"], best["graphed"]
    graphed = graphed_step is not None
    td_calls_per_step, loss_after_warmup = best["calls"], best["loss_after_warmup"]
    for c in candidates:
        if c is not best:
            c.clear()
    del candidates, base_model
    if dp and world > 1:
        # the replicas must hold identical parameters after the captured steps
        if not replicas_agree(model):
            raise SystemExit("bench.py: INVALID RUN: replicas diverged under step mode %r" % used_mode)
    split_graph = graphed and graphed_step.graph_b is not None
    use_flat = step.flat is not None
    graph = graphed_step.graph if graphed else None
    graph_b = graphed_step.graph_b if graphed else None

    if graphed and args.split_timing and rank == 0:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ta = tb = 0.0
        for _ in range(10):
            ev[0].record(); graph.replay(); ev[1].record(); step.sync(); graph_b.replay(); ev[2].record()
            torch.cuda.synchronize()
            ta += ev[0].elapsed_time(ev[1]) / 10
            tb += ev[1].elapsed_time(ev[2]) / 10
        print("split timing: forward+backward(+gather) graph %.3f ms, sync+clip+Adam graph %.3f ms" % (ta, tb),
              file=sys.stderr)
    run = graphed_step if graphed else step
    if args.h2d:
        # PCIe-inclusive variants (the headline `value` is measured without this flag): every step first
        # copies the batch from pinned host memory into the step's input buffers
        replay = run
        if args.h2d == "float32":
            host = {k: v.detach().cpu().pin_memory() for k, v in batch.items() if torch.is_tensor(v)}

            def run():
                for k, h in host.items():
                    batch[k].copy_(h, non_blocking=True)
                replay()
        else:
            from tripled_amd import ops as _ops
            frames = list(m["frame_ids"])
            u8 = torch.cat([(batch[("color", f, 0)] * 255).round().clamp(0, 255).to(torch.uint8) for f in frames], 0).cpu().pin_memory()
            aug_h = torch.zeros(len(frames) * B, 9).pin_memory()
            other = {k: v.detach().cpu().pin_memory() for k, v in batch.items()
                     if torch.is_tensor(v) and not (isinstance(k, tuple) and k[0] in ("color", "color_aug"))}
            u8_d, aug_d = torch.empty_like(u8, device=dev), torch.empty_like(aug_h, device=dev)

            def run():
                u8_d.copy_(u8, non_blocking=True)
                aug_d.copy_(aug_h, non_blocking=True)
                for k, h in other.items():
                    batch[k].copy_(h, non_blocking=True)
                color, color_aug = _ops.color_jitter_expand(u8_d, aug_d)
                for i, f in enumerate(frames):
                    batch[("color", f, 0)].copy_(color[i * B:(i + 1) * B])
                    batch[("color_aug", f, 0)].copy_(color_aug[i * B:(i + 1) * B])
                replay()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # a benchmark of a numerically broken step is not a benchmark: non-finite loss or parameters -> exit code 3
    try:
        final_loss = step.check_finite("after the %d timed steps" % args.steps)
    except NonFiniteLossError as e:
        print("bench.py: INVALID RUN: %s" % e, file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(3)

    # the same workload through train.py's path (train_mono -> Runner -> RunnerIteration).  N > 1: only on request -- a
    # failure there would involve collectives, and the headline line must not depend on it
    runner_ms, runner_err = None, None
    if args.entry == "runner" or (args.entry == "both" and world == 1):
        try:
            runner_ms = runner_entry(args, batch, world, dev)
        except NonFiniteLossError:
            raise
        except Exception as e:      # noqa: BLE001 -- reported in the line, never loses it
            runner_err = "%s: %s" % (type(e).__name__, str(e)[:300])

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": "train imgs/sec at KITTI %dx%d bs=%d/GPU" % (H, W, B),      # BASELINE.json's metric on the default config (192x640, 12)
            "value": round(world * B * args.steps / elapsed, 3),
            "unit": "imgs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 convs (MFMA) + f32 loss kernels" + (", fp8 forward GEMM of the 1x1 convs" if args.fp8 else ""))
            if dtype is not None else "f32",
            "data": "synthetic" if args.frames == "coherent" else "synthetic (i.i.d. noise frames)",
            "config": {"workload": "%s %dx%d bs=%d/GPU (%s), fwd+bwd+clip+Adam" % (
                m["name"], H, W, B, os.path.basename(args.config)), "global_batch": world * B,
                "parallelism": "dp%d" % world, "hip_graph": graphed,
                "entry": "tripled_amd.step.TrainStep replayed by bench.py (`value`); `runner_entry` = the same iteration "
                         "through mono.apis.train_mono / Runner.run, i.e. train.py's path",
                "syncbn": bool(use_syncbn and dp),
                "param_store": ("flat fp32 master + bf16 working copy" if step.flat is not None and lowp_store and used_mode != "overlap-graph"
                                else ("flat fp32" if step.flat is not None else "per-parameter")),
                "grad_sync": {"single": "none", "single-flat": "none",
                              "overlap-graph": "bucketed RCCL all-reduce overlapped with backward on the process group's side "
                                               "stream, captured with the whole step in one HIP graph",
                              "two-graph": "bucketed RCCL all-reduce of the flat gradient buffer between two HIP graphs",
                              "eager": "bucketed RCCL all-reduce overlapped with backward (eager step)"}[used_mode],
                "step_mode": used_mode, "step_mode_trials_ms": trials or None,
                "h2d_in_step": args.h2d or False, "extractor_tail_pruned": bool(m.get("prune_extractor_tail", False)),
                "capture_stream": CAPTURE_STREAM if graphed else None,
                # tripled_amd.streams: auto-encoder and pose network on side streams beside the depth chain (parallel branches
                # of the captured graph); off under the overlapped bucket engine, whose hooks assume one backward stream
                "branch_streams": bool(__import__("tripled_amd.streams", fromlist=["ENABLED"]).ENABLED and getattr(model, "branch_streams", True)
                                       and m["name"] in ("mono_fm_joint_inpaint_disentangle",
                                                         "mono_fm_joint_inpaint_disentangle_distill_sep_colorize")),
                "fallbacks": sum(dispatch.fallbacks.values()), "td_abi_calls_per_step": td_calls_per_step,
                "loss_after_warmup": round(loss_after_warmup, 6), "final_loss": round(final_loss, 6), "valid": True},
        }
        if runner_ms is not None:
            line["runner_entry"] = {"ms_per_step": round(runner_ms, 3), "value": round(world * B / (runner_ms * 1e-3), 3),
                                    "unit": "imgs/s", "vs_step_entry": round(ms / runner_ms, 4),
                                    "what": "train_mono -> Runner.run -> hooks -> RunnerIteration (HIP-graph replay), same batch "
                                            "resident in HBM, %d timed iterations" % args.steps}
        elif runner_err is not None:
            line["runner_entry"] = {"error": runner_err}
        if not args.no_roofline:
            kern = roofline_of_hot_kernels(cfg, batch)
            dom = max(kern, key=lambda k: kern[k]["seconds"])
            ach = kern[dom]["bytes"] / kern[dom]["seconds"] / 1e9
            workload = "B=%d %dx%d n_src=%d" % (B, H, W, len(m["frame_ids"]) - 1)
            tblob, why = stamped(os.path.join(ROOT, "profiles", "traffic.json"), TRAFFIC_SOURCES, workload)
            traffic = tblob.get(dom) if tblob else None
            line["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                                "traffic_source": (tblob["_stamp"] if tblob else why),
                                "launch_us": round(kern[dom]["seconds"] * 1e6, 2),
                                "before_the_run_us": ({k: round(v["seconds"] * 1e6, 2) for k, v in kern_before.items()}
                                                      if kern_before and "error" not in kern_before else kern_before),
                                "all": {k: {"us": round(v["seconds"] * 1e6, 2),
                                            "GBps": round(v["bytes"] / v["seconds"] / 1e9, 1)} for k, v in kern.items()}}
            try:
                t_loss = loss_path_time(cfg, batch)
                a = LOSS_PATH_BYTES_PER_PX * B * H * W / t_loss / 1e9
                line["roofline"]["loss_path"] = {"bound": "hbm", "what": "identity + 4 scales x (photometric, smoothness) "
                                                 "fwd+bwd, replayed as one HIP graph",
                                                 "ms_per_step": round(t_loss * 1e3, 3),
                                                 "bytes": LOSS_PATH_BYTES_PER_PX * B * H * W,
                                                 "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": round(a / HBM_PEAK_GBS, 4)}
            except Exception as e:      # noqa: BLE001 -- auxiliary figure; never lose the bench line over it
                line["roofline"]["loss_path"] = {"error": "%s: %s" % (type(e).__name__, e)}
        gflop = CONV_GFLOP_PER_IMG.get((H, W)) if m["name"] == "mono_fm_joint_inpaint_disentangle" else None
        if gflop is not None and dtype is not None:
            tf = gflop * B * world / (ms * 1e-3) / 1e3
            line["roofline_conv"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS * world,
                                     "unit": "TFLOP/s", "frac": round(tf / (MFMA_BF16_PEAK_TFLOPS * world), 4),
                                     "what": "reference-algorithmic conv FLOPs (SURVEY section 6) / whole step time"}
            mf, why = stamped(os.path.join(ROOT, "profiles", "mfma.json"), MFMA_SOURCES,
                              "%s B=%d %dx%d" % (os.path.basename(args.config), B, H, W))
            if mf is None:
                line["roofline_conv"]["pmc"] = None
                line["roofline_conv"]["pmc_note"] = why
            elif world == 1:
                # PMC evidence (separate rocprofv3 --pmc pass): MFMA-busy cycles of one step, summed over the 1024 SIMDs.
                # A busy cycle is 1024 bf16 FLOP only for the 32x32x16 / 16x16x32 instructions (the hand-written kernels); MIOpen's
                # igemm kernels issue the half-rate 32x32x8 form, so the count is a utilisation figure, not a FLOP count.
                busy = mf["mfma_busy_cycles_per_step"]
                line["roofline_conv"]["pmc"] = {
                    "mfma_busy_frac_of_step": round(busy / (ms * 1e-3 * mf["clock_ghz"] * 1e9 * mf["simds"]), 4),
                    "mfma_util_inside_conv_kernels": mf["mfma_util_conv_kernels"],
                    "mfma_util_inside_td_conv1x1": mf.get("mfma_util_td_conv1x1"), "source": mf["source"], "stamp": mf["_stamp"]}
        if world == 1 and not args.no_cpu_baseline:
            del model, step, graph, graph_b, graphed_step, run
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_batch, args.cpu_steps)
        print(json.dumps(line))
    if dist.is_initialized():
        # the result line is out; a peer that tears its sockets down first must not turn into a failed run
        try:
            dist.barrier()
            torch.cuda.synchronize()
            dist.destroy_process_group()
        except Exception as e:      # noqa: BLE001
            print("rank %d: teardown: %s" % (rank, e), file=sys.stderr)


if __name__ == "__main__":
    main()
