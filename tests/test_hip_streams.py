"""tripled_amd.streams: the three independent chains of the TripleD step on concurrent HIP streams give the step the serial
order gives -- the forward losses to fp32 rounding of MIOpen's run-to-run summation order (the chains share no buffer), the
gradients within the run-to-run spread MIOpen's order-dependent weight-gradient solvers have anyway."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _step(fork):
    import tripled_amd  # noqa: F401
    from tripled_amd import streams
    from mono.datasets import synthetic_batch
    from mono.model import MONO
    from tests.test_hip_model_step import _opt
    name = "mono_fm_joint_inpaint_disentangle"
    B, H, W = 2, 96, 160
    torch.manual_seed(11)
    model = MONO.module_dict[name](_opt(name, B, H, W)).cuda().train()
    model.DepthDecoder.do.eval()
    model.set_noise_source(lambda shape, device: torch.zeros(shape, device=device))
    batch = {k: v.cuda() for k, v in synthetic_batch(B, H, W, seed=3).items()}
    old = streams.ENABLED
    streams.ENABLED = fork
    try:
        _, losses = model(batch)
        sum(v.mean() for v in losses.values()).backward()
        torch.cuda.synchronize()
    finally:
        streams.ENABLED = old
    grads = torch.cat([p.grad.float().flatten() for p in model.parameters() if p.grad is not None])
    return {str(k): v.detach().float().mean().item() for k, v in losses.items()}, grads


def test_forked_step_equals_serial_step():
    l_fork, g_fork = _step(True)
    l_ser, g_ser = _step(False)
    l_ser2, g_ser2 = _step(False)
    assert l_fork.keys() == l_ser.keys()
    for k in l_ser:
        # forward: the chains share no buffer, so the values agree to the last bits -- not always ALL bits: MIOpen's fp32
        # forward solvers on these small maps sum in an order that varies from run to run (1e-7 relative, serial runs too)
        spread = abs(l_ser2[k] - l_ser[k])
        assert abs(l_fork[k] - l_ser[k]) <= 4 * spread + 2e-6 * abs(l_ser[k]), (k, l_fork[k], l_ser[k], l_ser2[k])
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    noise = rel(g_ser2, g_ser)                   # serial against serial: MIOpen's own run-to-run spread
    assert rel(g_fork, g_ser) <= 1.25 * noise + 0.02, (rel(g_fork, g_ser), noise)
    assert torch.isfinite(g_fork).all()


def test_branch_orders_the_streams_and_marks_the_tensors():
    import tripled_amd  # noqa: F401
    from tripled_amd import streams
    dev = torch.device("cuda", 0)
    a = torch.full((1 << 22,), 2.0, device=dev)
    with streams.Branch(dev, 0) as b:
        assert torch.cuda.current_stream() == b.stream and b.stream != torch.cuda.default_stream()
        out = {"x": [a * 3.0, (a + 1.0,)]}
    b.join(out)
    assert torch.cuda.current_stream() != b.stream
    assert float((out["x"][0] + out["x"][1][0]).sum()) == (6.0 + 3.0) * (1 << 22)
    assert streams.side_stream(dev, 0) is b.stream and streams.side_stream(dev, 1) is not b.stream
