"""One training iteration as a replayable unit: ``TrainStep`` (what the Runner's batch_processor +
DistOptimizerHook do per iteration, reference: mono/apis/trainer.py:32-60,
mono/core/utils/dist_utils.py:54-60) and ``capture_step`` (the same iteration recorded once into a
HIP graph and replayed).  ``bench.py``, the training shim and the parity tests all go through this
module, so the configuration that is benchmarked is the configuration that is tested.
"""
import math

import torch


class NonFiniteLossError(RuntimeError):
    """The training loss (or a parameter) stopped being finite."""


class TrainStep:
    """zero-grad -> forward -> sum of loss means -> backward -> [grad sync] -> clip -> Adam.

    After a call ``loss`` (total), ``losses`` (the model's loss_dict reduced to scalars, detached) and
    ``outputs`` (the model's outputs dict) refer to the tensors of the last executed iteration; under
    graph replay they are the graph's static tensors and are refreshed by every replay."""

    def __init__(self, model, cfg, batch, autocast_dtype, flat=False):
        self.model, self.batch, self.dtype = model, batch, autocast_dtype
        inner = model.module if hasattr(model, "module") else model
        self.params = [p for p in inner.parameters() if p.requires_grad]
        ocfg = dict(cfg.optimizer)
        if ocfg.pop("type") != "Adam":
            raise ValueError("TrainStep implements the configs' Adam optimiser only")
        clip = cfg.optimizer_config.get("grad_clip", None)
        self.max_norm = clip["max_norm"] if clip else None
        self.reducer = getattr(model, "reducer", None)
        self.flat = None
        on_gpu = batch["K"].is_cuda
        if flat:
            from .flat_amp import FlatMixedPrecision
            self.flat = FlatMixedPrecision(inner, max_norm=self.max_norm, lowp=flat == "lowp", **ocfg)
            self.optimizer = self.flat.optimizer
        else:
            # fused multi-tensor Adam on the GPU (same update rule as torch.optim.Adam(lr, weight_decay=0))
            self.optimizer = torch.optim.Adam(self.params, capturable=on_gpu, fused=on_gpu, **ocfg)
        self.loss = None
        self.losses = {}
        self.outputs = {}
        self.grad_norm = None

    def forward_backward(self):
        if self.flat is not None:
            self.flat.zero_grad()
        elif self.reducer is None:
            self.optimizer.zero_grad(set_to_none=True)   # with the DP engine, forward() re-zeroes the flat buffer
        with torch.autocast("cuda" if self.batch["K"].is_cuda else "cpu", dtype=self.dtype,
                            enabled=self.dtype is not None):
            outputs, losses = self.model(dict(self.batch))
        means = {k: v.float().mean() for k, v in losses.items()}
        total = sum(means.values())
        total.backward()
        if self.flat is not None:
            self.flat.collect()
        self.loss = total.detach()
        self.losses = {k: v.detach() for k, v in means.items()}
        self.outputs = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in outputs.items()}

    def sync(self):
        if self.flat is not None:
            self.flat.allreduce()
        elif self.reducer is not None and not self.reducer.overlap:
            self.reducer.allreduce_all()

    def update(self):
        if self.flat is not None:
            self.grad_norm = self.flat.step()
            return
        if self.max_norm is not None:
            self.grad_norm = torch.nn.utils.clip_grad_norm_(self.params, self.max_norm, norm_type=2, foreach=True)
        self.optimizer.step()

    def __call__(self):
        self.forward_backward()
        self.sync()
        self.update()
        return self.loss

    # ---- health checks (host syncs: call them outside timed regions) ----------------------------
    def loss_value(self):
        return float(self.loss)

    def check_finite(self, what="training step"):
        """Raise NonFiniteLossError if the last loss, any loss entry or any parameter is not finite."""
        v = self.loss_value()
        if not math.isfinite(v):
            bad = [str(k) for k, t in self.losses.items() if not bool(torch.isfinite(t).all())]
            raise NonFiniteLossError("%s: loss = %r (non-finite entries: %s)" % (what, v, ", ".join(bad) or "none"))
        flags = torch.stack([torch.isfinite(p.detach()).all() for p in self.params])
        if not bool(flags.all()):
            inner = self.model.module if hasattr(self.model, "module") else self.model
            names = [n for n, p in inner.named_parameters() if p.requires_grad]
            first = names[int((~flags).nonzero()[0])]
            raise NonFiniteLossError("%s: parameter %s is not finite" % (what, first))
        return v


class GraphedStep:
    """A TrainStep recorded into one HIP graph (or two, around an eager gradient exchange)."""

    def __init__(self, step, graph, graph_b=None):
        self.step, self.graph, self.graph_b = step, graph, graph_b

    def __call__(self):
        self.graph.replay()
        if self.graph_b is not None:
            self.step.sync()
            self.graph_b.replay()
        return self.step.loss


def warm_up(step, iters, stream):
    """Run ``iters`` eager iterations on ``stream`` (a side stream: whole-step capture needs autograd's
    AccumulateGrad nodes bound to a non-default stream) and join it back into the current stream."""
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for _ in range(iters):
            step()
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()


def capture_step(step, stream=None, split=False, capture_error_mode="global", validate=True):
    """Record ``step`` into a HIP graph.  ``stream`` = the capture stream (None: PyTorch's own capture
    stream).  ``split`` records forward+backward and clip+Adam as two graphs so that an eager collective
    can run between them.  Two replays are executed and checked before the graph is handed out: a captured
    step whose loss or parameters are not finite raises (ROCm 7.2 replayed a single-stream capture of this
    step wrongly from the SECOND replay on; tests/test_hip_graph_step.py compares the
    replayed trajectory with the eager one)."""
    graph = torch.cuda.CUDAGraph()
    graph_b = None
    if not split:
        with torch.cuda.graph(graph, stream=stream, capture_error_mode=capture_error_mode):
            step()
    else:
        with torch.cuda.graph(graph, stream=stream, capture_error_mode=capture_error_mode):
            step.forward_backward()
        graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph_b, stream=stream, capture_error_mode=capture_error_mode):
            step.update()
    g = GraphedStep(step, graph, graph_b)
    for i in range(2 if validate else 0):
        g()
        torch.cuda.synchronize()
        step.check_finite("replay %d of the captured step" % i)
    return g
