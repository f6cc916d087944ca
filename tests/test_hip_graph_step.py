"""The benchmarked configuration, tested: cfg_kitti_tripleD (ResNet50, B=12, 192x640) under bf16 autocast +
channels_last with the hand-written kernels, stepped K times (a) eagerly and (b) from the HIP graph exactly
as bench.py captures it (tripled_amd.step.capture_step), from identical weights, optimiser state, inputs and
RNG state.  Reference step semantics: mono/core/utils/dist_utils.py:54-60 (zero_grad, backward, clip, step)
around mono/model/mono_fm_joint_inpaint/net.py:477-518 (forward).

Two comparisons, tolerances stated:

1. SAME-STATE (tight; exact in the ``deterministic`` cases, which restrict MIOpen to solvers without
   order-dependent accumulation: every loss entry and every disparity of the replay is then BIT-IDENTICAL to the
   eager step's).  Before replay i the weights, BatchNorm buffers, Adam state and RNG offset are reset
   to what the eager run had before its step i.  The replay then executes the same kernels on the same data;
   what remains is order-dependent accumulation inside MIOpen: two EAGER forwards from the same state and RNG are
   bitwise equal up to DepthEncoder.encoder.layer2.0.conv2 (the first stride-2 3x3 convolution at 128 channels) and
   differ from there on (tools/diag_misc.py: per-module comparison; none of the hand-written kernels differs) --
   measured 3 % (scale 0) to 50 % (scale 3) of the bf16 disparity pixels off by one ulp, single pixels by 3, loss
   entries off by up to 3e-3 relative.  Every loss_dict entry must agree to 1e-5 + 1e-2 relative, the bf16
   disparities to max 8 ulp and mean 1 ulp (2-3x the spread measured between two eager runs, so that the test does
   not flake on the path's own nondeterminism).  The parameters after the update
   must agree with the eager run's next state to max |delta| <= 2.5 * lr (Adam's first updates are
   ~lr * sign(g): a noise-level gradient entry can flip a whole update; measured 1.3-1.6 lr) and mean
   |delta| <= 0.25 * lr (measured 0.03-0.06 lr).  The same bounds are asserted eager-vs-eager (noise floor:
   measured values are the same as graph-vs-eager, tools/diag_capture.py and gpurun logs of round 2).
2. FREE-RUNNING (loose).  K consecutive replays against K consecutive eager steps: the atomics noise is
   amplified by the sign-like Adam updates through ~100 bf16 layers, so the trajectories separate chaotically
   (two EAGER runs from the same state were measured 1 to 3 bf16 ulp apart in mean disparity after ONE update,
   run-dependent).  This part is therefore a sanity check on the loss trajectory only: every loss entry within
   2e-3 + 2 %, total within 2e-3, mean |disp difference| < 0.05 at every scale, asserted for graph-vs-eager and
   for eager-vs-eager alike.  The parity statement proper is comparison 1.

Plus: zero ATen fallbacks (strict mode), all parameters finite after 25 further replays.
"""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K_STEPS = 3          # (2 in the deterministic cases: MIOpen's deterministic solvers are slow)
ULP = 2.0 ** -8


def _build(cfg_name="cfg_kitti_tripleD.py", flat=False, **over):
    import tripled_amd  # noqa: F401
    from mmcv import Config
    from mono.datasets.synthetic import synthetic_batch
    from mono.model import MONO
    from tripled_amd.step import TrainStep
    cfg = Config.fromfile(os.path.join(ROOT, "config", cfg_name))
    for k, v in over.items():
        cfg.model[k] = v
    m = cfg.model
    torch.backends.cudnn.benchmark = False      # no MIOpen find: keeps the test to ~1 minute
    torch.manual_seed(1024)
    dev = torch.device("cuda", 0)
    model = MONO.module_dict[m["name"]](m).to(dev).to(memory_format=torch.channels_last)
    model.train()
    batch = synthetic_batch(m["imgs_per_gpu"], m["height"], m["width"], seed=1000, device=dev,
                            frame_ids=tuple(m["frame_ids"]))
    return cfg, model, TrainStep(model, cfg, batch, torch.bfloat16, flat="lowp" if flat else False)


def _snapshot(model, step):
    masters = (step.flat.flat_w.clone(), step.flat.flat_lp.clone()) if step.flat is not None else None
    return (copy.deepcopy(model.state_dict()), copy.deepcopy(step.optimizer.state_dict()),
            torch.cuda.get_rng_state(), masters)


def _restore(model, step, snap):
    sd, osd, rng, masters = snap
    with torch.no_grad():
        if masters is not None:                   # flat store: the fp32 masters and the bf16 working copy
            step.flat.flat_w.copy_(masters[0])
            step.flat.flat_lp.copy_(masters[1])
        cur = model.state_dict()
        for k, v in sd.items():
            cur[k].copy_(v)                       # in place: the graph holds these addresses
        for group_state, saved in zip(step.optimizer.state.values(), osd["state"].values()):
            for k, v in saved.items():
                if torch.is_tensor(v):
                    group_state[k].copy_(v)
    torch.cuda.set_rng_state(rng)
    torch.cuda.synchronize()


def _record(step, n_scales=4):
    torch.cuda.synchronize()
    return ({str(k): float(v) for k, v in step.losses.items()}, float(step.loss),
            [step.outputs[("disp", 0, s)].float().cpu().clone() for s in range(n_scales)])


def _compare_loose(tag, ref, got):
    for i, ((l_r, t_r, d_r), (l_g, t_g, d_g)) in enumerate(zip(ref, got)):
        assert l_r.keys() == l_g.keys()
        assert abs(t_r - t_g) < 2e-3, (tag, i, t_r, t_g)
        for k in l_r:
            assert abs(l_r[k] - l_g[k]) < 2e-3 + 2e-2 * abs(l_r[k]), (tag, i, k, l_r[k], l_g[k])
        for s, (a, b) in enumerate(zip(d_r, d_g)):
            assert float((a - b).abs().mean()) < 0.05, (tag, i, s, float((a - b).abs().mean()))


def _compare_tight(tag, i, ref, got, exact=False):
    (l_r, t_r, d_r), (l_g, t_g, d_g) = ref, got
    if exact:      # deterministic MIOpen solvers: the replayed forward must be bit-identical to the eager one
        for k in l_r:
            assert l_r[k] == l_g[k], (tag, i, k, l_r[k], l_g[k])
        for s, (a, b) in enumerate(zip(d_r, d_g)):
            assert torch.equal(a, b), (tag, i, s, float((a - b).abs().max()))
        return
    worst = max(abs(l_r[k] - l_g[k]) / (1e-5 + abs(l_r[k])) for k in l_r)
    frac = [float(((a - b).abs() > 0).float().mean()) for a, b in zip(d_r, d_g)]
    dmax = [float((a - b).abs().max()) / ULP for a, b in zip(d_r, d_g)]
    print("[%s same-state %d] worst loss-entry rel diff %.2e, disp pixels differing %s, max diff (ulp) %s" % (
        tag, i, worst, ["%.4f" % f for f in frac], dmax))
    if os.environ.get("TD_CALIBRATE"):
        return
    for k in l_r:
        assert abs(l_r[k] - l_g[k]) < 1e-5 + 1e-2 * abs(l_r[k]), (tag, i, k, l_r[k], l_g[k])
    for s, (a, b) in enumerate(zip(d_r, d_g)):
        d = (a - b).abs()
        assert float(d.max()) <= 8 * ULP + 1e-7, (tag, i, s, float(d.max()))
        assert float(d.mean()) < 1.0 * ULP, (tag, i, s, float(d.mean()) / ULP)


def _param_delta(model, snap, step):
    if step.flat is not None:                     # compare the fp32 masters (the module holds bf16 working copies)
        d = (step.flat.flat_w - snap[3][0]).abs()
        return float(d.max()), float(d.mean())
    sd = snap[0]
    cur = model.state_dict()
    mx, tot, n = 0.0, 0.0, 0
    for k, p in model.named_parameters():
        d = (cur[k].float() - sd[k].float()).abs()
        mx = max(mx, float(d.max()))
        tot += float(d.sum())
        n += d.numel()
    return mx, tot / n


# the deterministic cases run a reduced copy of the configuration (ResNet18 x3, B=4, 96x320: every kernel and the same
# capture path, but MIOpen's deterministic solvers are ~50x slower at the full C2 size); the C2 size itself runs with
# the benchmark's solver set against the measured noise floor
SMALL = dict(depth_num_layers=18, pose_num_layers=18, extractor_num_layers=18, imgs_per_gpu=4, height=96, width=320)


# last case = exactly what bench.py runs by default: C2 size, single-stream capture, flat mixed-precision parameter store
@pytest.mark.parametrize("capture,deterministic,flat", [("side", True, False), ("default", True, True), ("side", False, True)])
def test_graph_replay_matches_eager_c2(capture, deterministic, flat):
    from tripled_amd import dispatch
    from tripled_amd.step import capture_step, warm_up
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
    prev = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = deterministic      # MIOpen: only solvers without order-dependent accumulation
    try:
        _run(capture, deterministic, flat, dispatch, capture_step, warm_up)
    finally:
        torch.backends.cudnn.deterministic = prev


def _run(capture, deterministic, flat, dispatch, capture_step, warm_up):
    K_STEPS = 2 if deterministic else 3
    cfg, model, step = _build(flat=flat, **(SMALL if deterministic else {}))
    lr = cfg.optimizer["lr"]
    dispatch.reset()
    side = torch.cuda.Stream()
    with dispatch.strict():
        warm_up(step, 2, side)
    assert sum(dispatch.fallbacks.values()) == 0, dict(dispatch.fallbacks)
    bn_calls = sum(dispatch.hip_calls[k] for k in ("td_bn_fwd", "td_bn_fwd_from_partials", "td_conv1x1_fwd_bnrelu"))
    assert dispatch.hip_calls["td_photo_fwd"] == 2 * 4 and bn_calls > 100

    snaps, eager = [_snapshot(model, step)], []
    for _ in range(K_STEPS):
        warm_up(step, 1, side)
        eager.append(_record(step))
        snaps.append(_snapshot(model, step))
    step.check_finite("eager")

    if capture == "side" and not deterministic:    # noise floor of the free-running comparison
        _restore(model, step, snaps[0])
        eager2 = []
        for _ in range(K_STEPS):
            warm_up(step, 1, side)
            eager2.append(_record(step))
        _compare_loose("eager-vs-eager", eager, eager2)
        for i in range(K_STEPS):                   # ... and of the same-state comparison
            _restore(model, step, snaps[i])
            warm_up(step, 1, side)
            _compare_tight("eager-vs-eager", i, eager[i], _record(step))

    _restore(model, step, snaps[0])
    graphed = capture_step(step, stream=side if capture == "side" else None, validate=False)

    # 1. same-state replays
    for i in range(K_STEPS):
        _restore(model, step, snaps[i])
        graphed()
        _compare_tight(capture, i, eager[i], _record(step), exact=deterministic)
        mx, mean = _param_delta(model, snaps[i + 1], step)
        print("[%s same-state %d] parameters after the update vs eager: max %.2f lr, mean %.4f lr" % (capture, i, mx / lr, mean / lr))
        assert mx <= 2.5 * lr and mean <= 0.25 * lr, (capture, i, mx / lr, mean / lr)

    # 2. free-running replays
    _restore(model, step, snaps[0])
    replayed = []
    for _ in range(K_STEPS):
        graphed()
        replayed.append(_record(step))
    step.check_finite("graph replay")
    assert abs(replayed[0][1] - replayed[-1][1]) > 1e-4      # the trajectory moves (a frozen graph is also "finite")
    _compare_loose(capture, eager, replayed)
    # a longer replay stays finite (the ROCm packet-capture corruption appeared between replay 1 and 20)
    for _ in range(5 if deterministic else 25):
        graphed()
    step.check_finite("further replays")
