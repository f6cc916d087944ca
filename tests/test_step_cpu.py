"""tripled_amd.step on the host (oracle loss backend): the iteration bench.py / the tests share, and its health gate."""
import math

import pytest
import torch

import tripled_amd  # noqa: F401
from mmcv import ConfigDict
from mono.datasets import synthetic_batch
from mono.model import MONO
from oracle.backend import OracleLossBackend
from tripled_amd.step import NonFiniteLossError, TrainStep


def _step():
    B, H, W = 1, 64, 128
    opt = ConfigDict(name="mono_fm", depth_num_layers=18, pose_num_layers=18, extractor_num_layers=18, frame_ids=[0, -1, 1],
                     imgs_per_gpu=B, height=H, width=W, scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0,
                     depth_pretrained_path=None, pose_pretrained_path=None, extractor_pretrained_path=None, automask=True,
                     disp_norm=True, perception_weight=1e-3, smoothness_weight=1e-3)
    cfg = ConfigDict(model=opt, optimizer=dict(type="Adam", lr=1e-4, weight_decay=0),
                     optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)))
    torch.manual_seed(0)
    model = MONO.module_dict["mono_fm"](opt)
    model.set_loss_backend(OracleLossBackend())
    model.train()
    return model, TrainStep(model, cfg, synthetic_batch(B, H, W, seed=1, with_mask=False), None)


def test_train_step_runs_and_reports():
    model, step = _step()
    before = model.DepthDecoder.disp1[0].conv.weight.detach().clone()
    loss = step()
    assert math.isfinite(float(loss)) and set(map(str, step.losses)) >= {"('min_reconstruct_loss', 0)", "('smooth_loss', 3)"}
    assert abs(float(sum(step.losses.values())) - float(loss)) < 1e-6
    assert ("disp", 0, 0) in step.outputs and float(step.grad_norm) > 0
    assert not torch.equal(model.DepthDecoder.disp1[0].conv.weight, before)
    assert step.check_finite() == float(loss)


def test_health_gate_names_the_broken_piece():
    model, step = _step()
    step()
    with torch.no_grad():
        model.PoseDecoder.conv3.weight[0, 0, 0, 0] = float("nan")
    with pytest.raises(NonFiniteLossError, match="PoseDecoder.conv3.weight"):
        step.check_finite("poisoned")
    step.loss = torch.tensor(float("nan"))
    step.losses[("smooth_loss", 0)] = torch.tensor(float("inf"))
    with pytest.raises(NonFiniteLossError, match="smooth_loss"):
        step.check_finite("a non-finite loss")


def test_only_adam_configs_are_accepted():
    model, step = _step()
    cfg = ConfigDict(model=model.opt, optimizer=dict(type="SGD", lr=1e-2), optimizer_config=dict(grad_clip=None))
    with pytest.raises(ValueError):
        TrainStep(model, cfg, step.batch, None)


def test_list_valued_losses_reduce_like_the_reference():
    """reference: mono/apis/trainer.py:39-47 -- tensor -> mean, list -> sum of means, anything else -> TypeError."""
    from tripled_amd.step import reduce_loss
    a, b = torch.tensor([1.0, 3.0]), torch.tensor([[2.0], [6.0]])
    assert float(reduce_loss("t", a)) == 2.0
    assert float(reduce_loss("l", [a, b])) == 6.0
    with pytest.raises(TypeError, match="bad"):
        reduce_loss("bad", 3.0)


def test_runner_iteration_reports_the_graphs_outputs_after_an_eager_iteration():
    """Root cause of the one graph-vs-eager mismatch of round 3: an eager ``step()`` between two
    replays (a ragged batch; the same-state parity test) rebinds ``step.loss`` / ``step.losses`` to the eager iteration's
    tensors; a replay refreshes only the graph's STATIC tensors, so ``RunnerIteration`` must report those and not whatever
    ``step.losses`` points at.  Runs on the host with a stand-in for the captured graph."""
    from collections import OrderedDict
    from tripled_amd.step import RunnerIteration

    class FakeStep:
        device = torch.device("cpu")

        def __init__(self):
            self.loss, self.losses, self.batch = None, {}, None
            self.calls = 0

        def __call__(self):                         # an eager iteration: NEW tensors, like TrainStep.forward_backward
            self.calls += 1
            self.losses = OrderedDict(a=torch.tensor(100.0 + self.calls))
            self.loss = torch.tensor(100.0 + self.calls)

        def check_finite(self, what=""):
            return float(self.loss)

    step = FakeStep()
    it = RunnerIteration(step, stage=lambda d: d, eager_processor=None, warmup_iters=1)
    static_losses, static_loss = OrderedDict(a=torch.tensor(0.0)), torch.tensor(0.0)

    def replay():                                   # a graph replay: writes into the static tensors, rebinds nothing
        static_losses["a"].add_(1.0)
        static_loss.add_(1.0)

    it.mode, it.graphed, it._graph_out = "one-graph", replay, (static_loss, static_losses)
    it.static = {"K": torch.zeros(2, 4, 4)}
    it.signature = it._sig(it.static)
    it.seen = 1

    class Model:
        def train(self):
            pass

    data = {"K": torch.zeros(2, 4, 4)}
    out = it(Model(), dict(data), True)
    assert float(out["log_vars"]["a"]) == 1.0
    step()                                          # eager iteration in between (rebinds step.losses)
    assert float(step.losses["a"]) == 101.0
    out = it(Model(), dict(data), True)
    assert float(out["log_vars"]["a"]) == 2.0 and float(out["loss"]) == 2.0     # the replay's value, not 101
    ragged = {"K": torch.zeros(1, 4, 4)}
    out = it(Model(), ragged, True)                 # another shape: eager on that batch
    assert float(out["log_vars"]["a"]) == 102.0 and it.eager_iterations == 1
    out = it(Model(), dict(data), True)
    assert float(out["log_vars"]["a"]) == 3.0


def test_the_overlapped_bucket_engine_keeps_backward_on_one_stream():
    """The overlapped engine launches its collectives from per-parameter hooks on the stream it was built on, so a model that
    trains under it must not fork its sub-networks onto side streams (tripled_amd.streams); every other gradient exchange
    (none, or the flat buffer's all-reduce after backward) leaves the forks as they are."""
    model, step = _step()
    cfg = ConfigDict(model=model.opt, optimizer=dict(type="Adam", lr=1e-4, weight_decay=0),
                     optimizer_config=dict(grad_clip=None))
    assert getattr(model, "branch_streams", True) is True

    class _Wrapped(torch.nn.Module):
        def __init__(self, module, overlap):
            super().__init__()
            self.module = module
            self.reducer = type("Reducer", (), {"overlap": overlap})()

        def forward(self, x):
            return self.module(x)

    TrainStep(_Wrapped(model, overlap=False), cfg, step.batch, None)
    assert getattr(model, "branch_streams", True) is True
    TrainStep(_Wrapped(model, overlap=True), cfg, step.batch, None)
    assert model.branch_streams is False
