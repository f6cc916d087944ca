"""One training iteration as a replayable unit: ``TrainStep`` (what the Runner's batch_processor +
DistOptimizerHook do per iteration, reference: mono/apis/trainer.py:32-60,
mono/core/utils/dist_utils.py:54-60), ``capture_step`` (the same iteration recorded once into a
HIP graph and replayed) and ``RunnerIteration`` (the captured iteration behind the Runner's
batch_processor slot: what ``train.py`` executes per batch, reference: mono/apis/trainer.py:147-189).
``bench.py``, the training shim and the parity tests all go through this module, so the
configuration that is benchmarked is the configuration that is tested and the one ``train.py`` runs.
"""
import logging
import math
import os
from collections import OrderedDict

import torch
import torch.distributed as dist


def single_stream_capture_ok():
    """A capture on ONE stream makes the step a linear graph, which ROCm 7.2 replays through its AQL-packet-capture
    fast path -- wrongly for this step from the second replay on -- unless
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was in the environment when the HIP runtime initialised.  The package records on
    import whether that was the case (``tripled_amd.PACKET_CAPTURE_OFF_AT_HIP_INIT``)."""
    import tripled_amd
    return bool(getattr(tripled_amd, "PACKET_CAPTURE_OFF_AT_HIP_INIT", False))


class NonFiniteLossError(RuntimeError):
    """The training loss (or a parameter) stopped being finite."""


def reduce_loss(name, value):
    """One entry of the model's loss dict as a scalar, like the reference's batch_processor (mono/apis/trainer.py:40-49):
    a tensor contributes its mean, a list of tensors the sum of their means, anything else is a TypeError."""
    if torch.is_tensor(value):
        v = value.float()
        return v.reshape(()) if v.numel() == 1 else v.mean()     # a scalar entry is its own mean: no reduction launch
    if isinstance(value, list):
        return sum(reduce_loss(name, v) for v in value)
    raise TypeError("%s is not a tensor or list of tensors" % (name,))


class TrainStep:
    """zero-grad -> forward -> sum of loss means -> backward -> [grad sync] -> clip -> Adam.

    After a call ``loss`` (total), ``losses`` (the model's loss_dict reduced to scalars, detached) and
    ``outputs`` (the model's outputs dict) refer to the tensors of the last executed iteration; under
    graph replay they are the graph's static tensors and are refreshed by every replay."""

    def __init__(self, model, cfg, batch, autocast_dtype, flat=False, prepare=None, device=None, lr_tensor=False,
                 keep_outputs=True):
        """``flat``: False (per-parameter fused Adam), "lowp" / "fp32" (a new flat store of that kind) or an existing
        ``FlatMixedPrecision`` (the trainer's, which the checkpoint shim also knows).  ``prepare``: applied to a shallow
        copy of the batch dict before the forward (the device-side expansion of the uint8 wire format).  ``lr_tensor``:
        keep the learning rate in a device tensor, so that a schedule can change it under graph replay."""
        self.model, self.batch, self.dtype = model, batch, autocast_dtype
        self.prepare, self.keep_outputs = prepare, keep_outputs
        inner = model.module if hasattr(model, "module") else model
        self.params = [p for p in inner.parameters() if p.requires_grad]
        ocfg = dict(cfg.optimizer)
        if ocfg.pop("type") != "Adam":
            raise ValueError("TrainStep implements the configs' Adam optimiser only")
        clip = cfg.optimizer_config.get("grad_clip", None)
        self.max_norm = clip["max_norm"] if clip else None
        self.reducer = getattr(model, "reducer", None)
        if self.reducer is not None and getattr(self.reducer, "overlap", False):
            # the overlapped bucket engine launches its collectives from per-parameter hooks on the stream it was built on:
            # keep every backward node on that stream (tripled_amd.streams forks the sub-networks otherwise)
            inner.branch_streams = False
        self.flat = None
        self.device = torch.device(device) if device is not None else batch["K"].device
        on_gpu = self.device.type == "cuda"
        if flat is not False and flat is not None and not isinstance(flat, str):
            self.flat = flat
            self.flat.max_norm = self.max_norm
            self.optimizer = self.flat.optimizer
        elif flat:
            from .flat_amp import FlatMixedPrecision
            self.flat = FlatMixedPrecision(inner, max_norm=self.max_norm, lowp=flat == "lowp", **ocfg)
            self.optimizer = self.flat.optimizer
        else:
            # fused multi-tensor Adam on the GPU (same update rule as torch.optim.Adam(lr, weight_decay=0))
            self.optimizer = torch.optim.Adam(self.params, capturable=on_gpu, fused=on_gpu, **ocfg)
        if lr_tensor and on_gpu:
            for group in self.optimizer.param_groups:
                if not torch.is_tensor(group["lr"]):
                    exact = float(group["lr"])
                    group["lr"] = torch.tensor(exact, dtype=torch.float32, device=self.device)
                    group["lr"]._host_value = exact      # what schedules and log lines read (mmcv.runner.hooks.lr_value)
        self.loss = None
        self.losses = {}
        self.outputs = {}
        self.grad_norm = None

    def forward_backward(self):
        if self.flat is not None:
            self.flat.zero_grad()
        elif self.reducer is None:
            self.optimizer.zero_grad(set_to_none=True)   # with the DP engine, forward() re-zeroes the flat buffer
        data = dict(self.batch)
        if self.prepare is not None:
            data = self.prepare(data)
        with torch.autocast(self.device.type, dtype=self.dtype, enabled=self.dtype is not None):
            outputs, losses = self.model(data)
        means = OrderedDict((k, reduce_loss(k, v)) for k, v in losses.items())
        # one stack + one sum instead of a chain of ~25 scalar adds (the entries' gradients are 1 either way; the total differs
        # from the reference's left-to-right Python sum by fp32 rounding of the summation order only)
        vals = list(means.values())
        total = vals[0] if len(vals) == 1 else torch.stack(vals).sum()
        if self.flat is not None and self.reducer is None and self.device.type == "cuda":
            # TD_WGRAD_GROUP=step: nothing reads a weight gradient before collect(), so the 1x1 weight gradients of the whole
            # step may be enqueued and launched together (tripled_amd.ops.deferred_wgrads; a no-op under the default "node"
            # scope, where every fused backward node groups its own -- the faster of the two, see ops.wgrad_group)
            from . import ops
            with ops.deferred_wgrads():
                total.backward()
        else:
            total.backward()
        if self.flat is not None:
            self.flat.collect()
        self.loss = total.detach()
        self.losses = OrderedDict((k, v.detach()) for k, v in means.items())
        self.outputs = ({k: (v.detach() if torch.is_tensor(v) else v) for k, v in outputs.items()}
                        if self.keep_outputs else {})

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
    if stream is not None and not single_stream_capture_ok():
        raise RuntimeError("single-stream capture needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment BEFORE the HIP "
                           "runtime initialises (import tripled_amd before the first torch.cuda call, or export it): with the "
                           "runtime's packet-capture fast path on, replays of this step are wrong from the second one on")
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


def ranks_agree(ok, device):
    """True on every rank iff ``ok`` is True on every rank (all_reduce(MIN)); a one-process job returns ``ok``.  Every rank
    must call it the same number of times: it is how the ranks settle which step form they all run."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(ok)
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() > 0)


def replicas_agree(model, rel=1e-6):
    """Data-parallel replicas must hold identical parameters after a synchronised update: compares a checksum (sum of all
    parameter sums, in float64) over the ranks.  True on every rank iff the minimum and the maximum coincide."""
    params = [p for p in model.parameters()]
    chk = torch.stack([p.detach().double().sum() for p in params]).sum().reshape(1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return True
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(hi - lo) <= rel * max(1.0, abs(float(hi)))


class RunnerIteration:
    """The training iteration of ``train.py`` as a HIP-graph replay, in the Runner's ``batch_processor`` slot.

    ``runner.train`` (mmcv 0.4.4: before_train_iter hooks -> batch_processor -> after_train_iter hooks, reference:
    mono/apis/trainer.py:147-189) calls this object once per batch.  A call

      1. moves the batch to the device (``stage``, normally a no-op behind DevicePrefetcher) and copies it into STATIC input
         buffers (first batch: allocated from it).  The uint8 wire format stays bytes there; ``td_color_jitter`` runs inside
         the step;
      2. executes the whole iteration -- zero-grad, forward, sum of loss means, backward, gradient exchange, clip, Adam
         (``TrainStep``) -- eagerly for the first ``warmup_iters`` batches (MIOpen picks its solvers, the allocator sizes its
         pools; these are real training iterations on real batches), then captures it ONCE and from then on replays it;
      3. returns ``dict(loss, log_vars, num_samples)`` like ``batch_processor``: ``log_vars`` are views of ONE snapshot of the
         graph's static loss scalars (a single small copy per iteration; read back only when a log line is due).

    The learning rate lives in a device tensor (``lr_tensor``), so ``LrUpdaterHook`` changes it between replays; the update
    has already happened when the optimiser hook's slot comes, so the trainer registers ``IterationDoneHook`` there.  A
    batch whose shapes differ from the static buffers (a ragged tail without drop_last) takes the eager iteration.  The
    evaluation hooks run outside the graph.  N > 1: on RCCL the collectives (flat gradient all-reduce, SyncBatchNorm
    statistics) are captured with the step; on a backend whose collectives cannot be captured the step is two graphs around
    an eager all-reduce (no SyncBatchNorm) or eager.  Every rank takes the same decision (all_reduce(MIN) of the outcome).
    """

    def __init__(self, step, stage, eager_processor, warmup_iters=3, logger=None, syncbn=False):
        self.step, self.stage, self.eager_processor = step, stage, eager_processor
        self.warmup_iters = max(1, int(warmup_iters))
        self.logger = logger or logging.getLogger(__name__)
        self.syncbn = syncbn
        self.static = None
        self.signature = None
        self.graphed = None
        self.mode = None              # "one-graph" | "two-graph" | "eager" once decided
        self.seen = 0
        self.replays = 0
        self.eager_iterations = 0
        self.side = None
        self._graph_out = None

    # ---- inputs --------------------------------------------------------------------------------------------
    @staticmethod
    def _sig(data):
        return tuple((str(k), tuple(v.shape), v.dtype) for k, v in data.items() if torch.is_tensor(v))

    def _load_static(self, data):
        if self.static is None:
            self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
            self.signature = self._sig(self.static)
            self.step.batch = self.static
            return True
        if self._sig(data) != self.signature:
            return False
        for k, v in data.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        return True

    # ---- execution modes -----------------------------------------------------------------------------------
    def _world(self):
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _candidate_modes(self):
        if self._world() == 1:
            return ["one-graph", "eager"]
        if dist.get_backend() == "nccl":
            return ["one-graph"] + ([] if self.syncbn else ["two-graph"]) + ["eager"]
        return ([] if self.syncbn else ["two-graph"]) + ["eager"]

    def _agree(self, ok):
        return ranks_agree(ok, self.step.device)

    def _capture(self):
        stream = self.side if single_stream_capture_ok() else None
        dp = self._world() > 1
        for mode in self._candidate_modes():
            if mode == "eager":
                self.mode, self.graphed = "eager", None
                break
            ok, g = True, None
            try:
                if dp:
                    dist.barrier()
                torch.cuda.synchronize()
                if dp:      # let the process group's watchdog (100 ms poll) retire the warm-up's works before the capture starts
                    import time
                    time.sleep(0.5)
                g = capture_step(self.step, stream=stream, split=mode == "two-graph",
                                 capture_error_mode="thread_local" if dp else "global", validate=False)
            except Exception as e:      # noqa: BLE001 -- this form is not available here; the next one is tried
                ok = False
                self.logger.warning("HIP-graph capture of the training iteration (%s) failed: %s: %s", mode, type(e).__name__,
                                    str(e)[:300])
            if self._agree(ok):
                self.mode, self.graphed = mode, g
                # the graph's own output tensors: an eager iteration in between (a ragged batch) rebinds step.loss / step.losses
                self._graph_out = (self.step.loss, self.step.losses)
                break
        # every rank logs its own decision: a divergence between ranks (which ranks_agree is there to prevent) would be visible.
        # The N > 1 forms (collectives captured in one graph on RCCL / two graphs around an eager all-reduce) have run on one
        # GPU only (one-rank RCCL, two gloo ranks: tests/test_hip_runner_dp.py); no multi-GPU node was available to this build.
        self.logger.info("[rank %d/%d] training iteration: %s%s", dist.get_rank() if self._world() > 1 else 0, self._world(), self.mode,
                         "" if self.graphed is None else " (captured on %s)" % ("one stream" if stream is not None
                                                                              else "the default capture stream"))

    def _run_eager(self):
        if self.side is None:
            self.side = torch.cuda.Stream()
        warm_up(self.step, 1, self.side) if self.mode is None else self.step()
        self.eager_iterations += 1

    # ---- the batch_processor call --------------------------------------------------------------------------
    def __call__(self, model, data, train_mode, **kwargs):
        if not train_mode:
            return self.eager_processor(model, data, train_mode, **kwargs)
        model.train()
        data = self.stage(data)
        n = int(data["K"].shape[0])
        if not self._load_static(data):
            # a batch of another shape: the same iteration, eagerly, on that batch
            self.step.batch = data
            try:
                self.step()
            finally:
                self.step.batch = self.static
            self.eager_iterations += 1
            return self._result(n)
        self.seen += 1
        if self.mode is None and self.seen <= self.warmup_iters:
            self._run_eager()
            return self._result(n)
        if self.mode is None:
            self._capture()
        if self.graphed is None:
            self._run_eager()
            return self._result(n)
        self.graphed()
        self.step.loss, self.step.losses = self._graph_out
        self.replays += 1
        if self.replays <= 2:       # the health gate of capture_step, on the first real replays (host sync, twice)
            if self.step.device.type == "cuda":
                torch.cuda.synchronize()
            self.step.check_finite("replay %d of the captured training iteration" % self.replays)
        return self._result(n)

    def _result(self, n):
        loss, losses = self.step.loss, self.step.losses
        keys = [str(k) for k in losses] + ["loss"]
        snap = torch.stack([v.reshape(()) for v in losses.values()] + [loss.reshape(())])     # one small copy per iteration
        log_vars = OrderedDict((k, snap[i]) for i, k in enumerate(keys))
        return dict(loss=loss, log_vars=log_vars, num_samples=n)
