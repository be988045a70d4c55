"""Host-side pieces of the differentiable training forward (eamm_amd/train_graph.py) that need no GPU: the anti-aliasing
down-sampler as two banded matrix products against the reference's definition (zero-pad 6, depth-wise 13x13 Gaussian, keep
every 4th row / column -- reference modules/util.py:1005-1052), and the NHWC helpers."""
import pytest
import torch
import torch.nn.functional as F

from eamm_amd import train_graph
from eamm_amd.weights import antialias_kernel


def reference_antialias(x, weight, step):
    ka = weight.shape[-1] // 2
    return F.conv2d(F.pad(x, (ka, ka, ka, ka)), weight, groups=x.shape[1])[:, :, ::step, ::step]


@pytest.mark.parametrize("shape", [(2, 3, 64, 64), (1, 3, 72, 56), (1, 3, 256, 256), (2, 3, 30, 18)])
def test_banded_antialias_matches_the_depthwise_convolution(shape):
    w = antialias_kernel(shape[1]).double()
    x = torch.randn(*shape, dtype=torch.float64, generator=torch.Generator().manual_seed(3)).requires_grad_()
    want = reference_antialias(x, w, 4)
    got = train_graph._antialias_down(x, w, 0.25)
    assert got.shape == want.shape
    # the stored buffer is the outer product ROUNDED to fp32 entry by entry: its row / column sums reproduce it to ~1e-8
    assert float((got - want).detach().abs().max()) < 1e-7
    gw, = torch.autograd.grad(want.square().sum(), x)
    gg, = torch.autograd.grad(got.square().sum(), x)
    assert float((gw - gg).abs().max()) < 2e-7


def test_antialias_falls_back_to_the_convolution_for_a_buffer_that_is_not_an_outer_product():
    w = antialias_kernel(3).clone()
    w[:, :, 2, 5] += 0.01          # no longer rank one
    x = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(4))
    assert float((train_graph._antialias_down(x, w, 0.25) - reference_antialias(x, w, 4)).abs().max()) == 0.0


def test_nhwc_helpers():
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)            # NHWC [B,H,W,C]
    up = train_graph._upsample2(x)
    want = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2).permute(0, 2, 3, 1)     # util.py:896, nearest
    assert torch.equal(up, want)
    assert train_graph._pad_last(x, 8).shape == (2, 3, 4, 8) and torch.equal(train_graph._pad_last(x, 8)[..., :5], x)
    assert train_graph._pad_last(x, 5) is x
    g = train_graph._grid(4, 6, x)
    assert g.shape == (4, 6, 2) and float(g[0, 0, 0]) == -1.0 and float(g[-1, -1, 1]) == 1.0 and float(g[0, -1, 0]) == 1.0


def test_differentiable_operators_have_no_cpu_fallback_and_validate_shapes():
    """The op-level wrappers (eamm_amd/autograd_ops.py) refuse CPU tensors (the product path never falls back) and reject
    shapes the kernels do not cover before touching the library."""
    from eamm_amd import autograd_ops as ops
    x = torch.zeros(1, 8, 8, 32)
    w3 = torch.zeros(32, 32, 3, 3)
    for call in (lambda: ops.conv2d_same_nhwc(x, w3), lambda: ops.conv2d_same(x.permute(0, 3, 1, 2), w3),
                 lambda: ops.warp_nhwc(x, torch.zeros(1, 8, 8, 2)), lambda: ops.warp(x.permute(0, 3, 1, 2), torch.zeros(1, 8, 8, 2)),
                 lambda: ops.first_conv7(torch.zeros(1, 8, 8, 4), torch.zeros(32, 3, 7, 7), torch.zeros(32)),
                 lambda: ops.final_conv7_sigmoid(x, torch.zeros(3, 32, 7, 7), torch.zeros(3))):
        with pytest.raises(RuntimeError, match="ROCm GPU"):
            call()


def test_generator_picks_the_graph_path_only_when_something_needs_a_gradient():
    from eamm_amd import OcclusionAwareGenerator, tiny_config
    gen = OcclusionAwareGenerator(**tiny_config()).train()
    src = torch.zeros(2, 3, 64, 64)
    kp = {"value": torch.zeros(2, 10, 2), "jacobian": torch.eye(2).expand(2, 10, 2, 2).clone()}
    assert not gen._wants_graph(src, kp, kp)                         # inference-style module: parameters frozen by default
    kpg = {k: v.clone().requires_grad_() for k, v in kp.items()}
    assert gen._wants_graph(src, kpg, kp)                            # the audio-to-key-point stage trains through the generator
    with torch.no_grad():
        assert not gen._wants_graph(src, kpg, kp)
    gen.requires_grad_(True)
    assert gen._wants_graph(src, kp, kp)                             # fine-tuning opts in
    with pytest.raises(RuntimeError, match="ROCm GPU"):             # ... and still has no CPU fallback
        gen(src, kp_driving=kp, kp_source=kp)
