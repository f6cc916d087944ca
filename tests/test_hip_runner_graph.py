"""train.py's iteration = the benchmarked iteration: the Runner replays the whole training iteration from a HIP graph
(tripled_amd.step.RunnerIteration behind mono.apis.train_mono; reference loop: mono/apis/trainer.py:32-60,147-189,
mono/core/utils/dist_utils.py:54-60).

1. ``test_runner_graph_equals_runner_eager``: ``train_mono`` on a reduced tripleD config (ResNet18 x3, B=4, 96x320), MIOpen
   restricted to its deterministic solvers, LR warm-up schedule active (the learning rate changes EVERY iteration):
   (a) graph on (3 eager iterations, capture, 4 replays), (b) the same RunnerIteration that never captures
   (graph_warmup_iters huge), twice: the checkpoints of (a) and (b) -- fp32 master weights, BatchNorm buffers -- must agree as
   closely as two eager runs agree with each other (the f32-atomic scatter in td_featwarp_bwd is order-dependent, so not even
   those are bit-identical), the logged losses of every iteration to fp32 rounding.  (c) the hook path without RunnerIteration (batch_processor + FlatOptimizerHook, host-side float lr):
   max |delta parameter| <= 2.5 lr, mean <= 0.25 lr and the logged losses within 1e-5 + 1e-2 relative (the bounds of
   test_hip_graph_step.py; the update differs by the rounding of lr to fp32).  The checkpoint keeps the reference's layout
   and a fresh model resumes from it with the graph on.
2. ``test_runner_iteration_same_state_c2``: the full C2 shape (ResNet50, B=12, 192x640, benchmark solver set): replay i of
   the Runner's iteration against the eager iteration from the SAME state, with test_hip_graph_step.py's noise-floor
   bounds; the uint8 wire format (td_color_jitter inside the graph) on the same path.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ULP = 2.0 ** -8


def _cfg(tmp, **over):
    import tripled_amd  # noqa: F401
    from mmcv import Config
    cfg = Config.fromfile(os.path.join(ROOT, "config", "cfg_kitti_tripleD.py"))
    small = dict(depth_num_layers=18, pose_num_layers=18, extractor_num_layers=18, imgs_per_gpu=4, height=96, width=320)
    cfg.model.update(small)
    cfg.imgs_per_gpu = 4
    cfg.total_epochs = 1
    cfg.validate = False
    cfg.work_dir = str(tmp)
    cfg.gpus = [0]
    cfg.log_config = dict(interval=1, hooks=[dict(type="TextLoggerHook")])
    cfg.log_level = "WARNING"
    cfg.strict_dispatch = True
    cfg.syncbn = False
    for k, v in over.items():
        cfg[k] = v
    return cfg


def _train(cfg, n_iters, seed=7):
    from mono.apis import train_mono
    from mono.datasets import ResidentBatches, synthetic_batch
    from mono.model import MONO
    m = cfg.model
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    model = MONO.module_dict[m["name"]](m)
    batch = synthetic_batch(m["imgs_per_gpu"], m["height"], m["width"], seed=1000, device=torch.device("cuda", 0),
                            frame_ids=tuple(m["frame_ids"]))
    train_mono(model, ResidentBatches(batch, n_iters), None, cfg, distributed=False, validate=False)
    return model


def _logged(work):
    logs = sorted(f for f in os.listdir(work) if f.endswith(".log.json"))
    return [r for r in (json.loads(line) for line in open(os.path.join(work, logs[-1]))) if r.get("mode") == "train" and "loss" in r]


def test_runner_graph_equals_runner_eager(tmp_path, caplog):
    import logging
    from tripled_amd import dispatch
    caplog.set_level(logging.INFO)
    prev = torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    N = 7
    try:
        runs = {}
        for tag, over in (("graph", dict(hip_graph=True)), ("eager", dict(hip_graph=True, graph_warmup_iters=10 ** 9)),
                          ("eager2", dict(hip_graph=True, graph_warmup_iters=10 ** 9)), ("hooks", dict(hip_graph=False))):
            work = tmp_path / tag
            cfg = _cfg(work, cudnn_benchmark=False, **over)
            dispatch.reset()
            model = _train(cfg, N)
            dispatch.set_strict(False)
            assert sum(dispatch.fallbacks.values()) == 0, (tag, dict(dispatch.fallbacks))
            ckpt = torch.load(work / "epoch_1.pth", weights_only=True)
            assert ckpt["meta"]["iter"] == N
            runs[tag] = (ckpt, _logged(work), model)
    finally:
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = prev
    lr = 1e-4
    g, e, e2, h = (runs[t][0] for t in ("graph", "eager", "eager2", "hooks"))

    def spread(a, b):
        """max and mean |difference| over all floating-point parameters / buffers / Adam moments, in units of lr."""
        mx, tot, n = 0.0, 0.0, 0
        for k, v in a["state_dict"].items():
            if v.is_floating_point():
                d = (v - b["state_dict"][k]).abs()
                mx, tot, n = max(mx, float(d.max())), tot + float(d.sum()), n + d.numel()
        return mx / lr, tot / n / lr

    # The one order-dependent accumulation left under MIOpen's deterministic solvers is the f32-atomic scatter of the feature
    # warp's adjoint (td_featwarp_bwd), so two EAGER runs are not bit-identical either: the statement is
    # "the replayed run differs from the eager run by no more than the eager run differs from itself" (x3 margin, plus a floor
    # of 1e-3 lr), which is three orders below what a wrong replay, a stale learning rate or a skipped update would give.
    floor, got = spread(e, e2), spread(g, e)
    print("[runner] eager-vs-eager spread max %.2e lr / mean %.2e lr; graph-vs-eager max %.2e lr / mean %.2e lr" % (floor + got))
    assert got[0] <= max(3 * floor[0], 1e-3) and got[1] <= max(3 * floor[1], 1e-5), (floor, got)
    for i, st in g["optimizer"]["state"].items():
        assert float(st["step"]) == float(e["optimizer"]["state"][i]["step"]) == N
    # the logged loss entries of every iteration agree to fp32 rounding, and the schedule moved the device-side lr every iteration
    lg, le, lh = runs["graph"][1], runs["eager"][1], runs["hooks"][1]
    assert len(lg) == len(le) == len(lh) == N
    skip = ("time", "data_time", "memory")
    for a, b in zip(lg, le):
        assert a.keys() == b.keys()
        for k in a:
            if k not in skip:
                assert a[k] == b[k] or abs(a[k] - b[k]) <= 1e-5 * abs(b[k]) + 1e-9, (k, a[k], b[k])
    lrs = [r["lr"] for r in lg]
    assert lrs == [r["lr"] for r in lh] and len(set(lrs)) == N and lrs[0] < lrs[-1] < lr
    # the file: reference layout, plain numbers in the param group
    pg = g["optimizer"]["param_groups"][0]
    assert isinstance(pg["lr"], float) and not {"capturable", "fused", "foreach"} & set(pg)
    assert all(v.dtype == torch.float32 for v in g["state_dict"].values() if v.is_floating_point())
    # (c) the hook path (float lr on the host, FlatOptimizerHook): same trajectory up to the rounding of lr
    hk = spread(g, h)
    print("[runner] graph-vs-hook-path max %.2e lr / mean %.2e lr" % hk)
    assert hk[0] <= 2.5 and hk[1] <= 0.25, hk
    for a, b in zip(lg, lh):
        for k in a:
            if k not in skip + ("mode", "epoch", "iter", "lr"):
                assert abs(a[k] - b[k]) <= 1e-5 + 1e-2 * abs(b[k]), (k, a[k], b[k])
    # the graph run really captured (and the never-capturing run did not)
    assert sum("training iteration: one-graph" in r.getMessage() for r in caplog.records) == 1
    # resume from the file with the graph on: one more epoch
    work = tmp_path / "resume"
    cfg = _cfg(work, cudnn_benchmark=False, hip_graph=True, resume_from=str(tmp_path / "graph" / "epoch_1.pth"), total_epochs=2)
    _train(cfg, N, seed=11)
    ck2 = torch.load(work / "epoch_2.pth", weights_only=True)
    assert ck2["meta"]["iter"] == 2 * N and float(ck2["optimizer"]["state"][0]["step"]) == 2 * N
    assert not torch.equal(ck2["state_dict"]["DepthDecoder.disp1.0.conv.weight"], g["state_dict"]["DepthDecoder.disp1.0.conv.weight"])
    assert all(bool(torch.isfinite(v).all()) for v in ck2["state_dict"].values() if v.is_floating_point())


def _iteration(cfg, wire="float32"):
    """What mono.apis.trainer._non_dist_train builds, without the Runner around it."""
    from mmcv.parallel import MMDataParallel
    from mono.apis import trainer
    from mono.datasets import synthetic_batch
    from mono.model import MONO
    m = cfg.model
    dev = torch.device("cuda", 0)
    torch.manual_seed(1024)
    model = MONO.module_dict[m["name"]](m)
    model = MMDataParallel(trainer.configure_execution(model, cfg, dev), device_ids=[0])
    flat, _ = trainer._build_flat_store(model, cfg)
    iteration, optimizer, hook = trainer._graphed_iteration(model, cfg, dev, flat)
    batch = synthetic_batch(m["imgs_per_gpu"], m["height"], m["width"], seed=1000, device=dev, frame_ids=tuple(m["frame_ids"]),
                            wire=wire, augment=True)
    return model, iteration, batch


def _snapshot(model, step):
    import copy
    return (copy.deepcopy(model.state_dict()), copy.deepcopy(step.optimizer.state_dict()), torch.cuda.get_rng_state(),
            (step.flat.flat_w.clone(), step.flat.flat_lp.clone()))


def _restore(model, step, snap):
    sd, osd, rng, masters = snap
    with torch.no_grad():
        step.flat.flat_w.copy_(masters[0])
        step.flat.flat_lp.copy_(masters[1])
        cur = model.state_dict()
        for k, v in sd.items():
            cur[k].copy_(v)
        for state, saved in zip(step.optimizer.state.values(), osd["state"].values()):
            for k, v in saved.items():
                if torch.is_tensor(v):
                    state[k].copy_(v)
    torch.cuda.set_rng_state(rng)
    torch.cuda.synchronize()


def _module_slices(model, flat):
    """index tensors into the flat master buffer, one per top-level sub-network (DepthEncoder, PoseDecoder, ...)."""
    inner = model.module if hasattr(model, "module") else model
    by = {}
    for name, p in inner.named_parameters():
        off = flat._offset_of.get(id(p))
        if off is None:
            continue
        by.setdefault(name.split(".")[0], []).append(torch.arange(off, off + p.numel(), device=flat.flat_w.device))
    return {k: torch.cat(v) for k, v in by.items()}


@pytest.mark.parametrize("wire", ["float32", "uint8"])
def test_runner_iteration_same_state_c2(wire):
    import tripled_amd  # noqa: F401
    from mmcv import Config
    from tripled_amd import dispatch
    cfg = Config.fromfile(os.path.join(ROOT, "config", "cfg_kitti_tripleD.py"))
    cfg.strict_dispatch = True
    torch.backends.cudnn.benchmark = False
    dispatch.reset()
    model, it, batch = _iteration(cfg, wire)
    step = it.step
    lr = cfg.optimizer["lr"]
    try:
        for _ in range(3):                                   # the eager iterations of the Runner's first batches
            out = it(model, dict(batch), True)
        assert it.mode is None and it.eager_iterations == 3 and it.replays == 0
        groups = _module_slices(model, step.flat)
        for i in range(3):
            snap = _snapshot(model, step)
            w0 = step.flat.flat_w.clone()
            out = it(model, dict(batch), True)               # call 4 captures, then replays
            torch.cuda.synchronize()
            got = ({k: float(v) for k, v in out["log_vars"].items()}, step.flat.flat_w.clone())
            assert it.mode == "one-graph" and it.replays == i + 1 and out["num_samples"] == cfg.model["imgs_per_gpu"]
            # the reported scalars are the GRAPH's static tensors (round 3's one mismatch: after the eager iteration below
            # had rebound step.losses, the next call reported that eager iteration's values -- DESIGN.md section 6)
            assert all(step.losses[k].data_ptr() == it._graph_out[1][k].data_ptr() for k in step.losses)
            _restore(model, step, snap)
            step()                                           # the same iteration, eagerly, from the same state
            torch.cuda.synchronize()
            ref = {str(k): float(v) for k, v in step.losses.items()}
            ref["loss"] = float(step.loss)
            w_e1 = step.flat.flat_w.clone()
            _restore(model, step, snap)
            step()                                           # and once more: the eager path's own run-to-run spread
            torch.cuda.synchronize()
            w_e2 = step.flat.flat_w
            assert ref.keys() == got[0].keys()
            bad = [(k, ref[k], got[0][k]) for k in ref if not abs(ref[k] - got[0][k]) < 1e-5 + 1e-2 * abs(ref[k])]
            assert not bad, (wire, i, bad)                   # every entry is compared (not only the first that differs)
            # per sub-network: the replay's parameter update against the eager one, measured against eager-vs-eager.
            # A replay that ran a sub-network on stale inputs / stale gradients gives an update uncorrelated with the
            # eager one (cosine ~ 0, relative difference ~ 1.4), which the former global bound (max 2.5 lr) let through.
            worst = (1.0, "")
            for name, idx in groups.items():
                ug, u1, u2 = (got[1] - w0)[idx], (w_e1 - w0)[idx], (w_e2 - w0)[idx]
                n1 = float(u1.norm())
                assert n1 > 0, name
                rel_ge, rel_ee = float((ug - u1).norm()) / n1, float((u2 - u1).norm()) / n1
                cos_ge = float(torch.dot(ug, u1)) / (float(ug.norm()) * n1 + 1e-30)
                worst = min(worst, (cos_ge, name))
                # measured (round 4, tools of DESIGN.md section 6): two EAGER iterations from one state differ by 50-80 % in the
                # ResNet50 encoders' gradients at this early iteration whenever MIOpen runs its order-dependent solvers (the
                # per-pixel arg-min and ReLU masks turn 1e-3 of forward noise into different selections) -- the round-3 tree
                # gives the same spread -- and by 1e-3 when it runs deterministic ones (a fresh process right after a capture).
                # The bound follows the measured eager-vs-eager spread; a wrong replay (~1.41, cosine ~0) fails either way.
                assert rel_ge <= 1.25 * rel_ee + 0.02, (wire, i, name, rel_ge, rel_ee)
                assert cos_ge >= 0.9 or rel_ge <= 0.95, (wire, i, name, cos_ge, rel_ge, rel_ee)
            d = (w_e1 - got[1]).abs()
            print("[runner %s same-state %d] parameters: max %.2f lr, mean %.4f lr; worst sub-network cosine %.4f (%s)"
                  % (wire, i, float(d.max()) / lr, float(d.mean()) / lr, worst[0], worst[1]))
        # a ragged batch takes the eager iteration and leaves the graph usable
        half = {k: v[:6].clone() for k, v in batch.items()}
        out = it(model, half, True)
        assert out["num_samples"] == 6 and it.eager_iterations == 4
        it(model, dict(batch), True)
        torch.cuda.synchronize()
        step.check_finite("after a ragged batch")
    finally:
        dispatch.set_strict(False)
    assert sum(dispatch.fallbacks.values()) == 0, dict(dispatch.fallbacks)
    if wire == "uint8":
        assert dispatch.hip_calls["td_color_jitter"] > 0
