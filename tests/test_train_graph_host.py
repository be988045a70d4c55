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


def test_generator_picks_the_graph_path_exactly_when_the_reference_would_build_a_graph():
    """Reference semantics (modules/generator.py:59-97 are plain differentiable torch ops; train.py:136 builds
    ``Adam(generator.parameters())`` straight after construction): parameters require grad by default, so any forward with
    gradients enabled is differentiable -- in .train() AND in .eval(); torch.no_grad() (demo.py:195) is the graph-free engine."""
    from eamm_amd import OcclusionAwareGenerator, tiny_config
    gen = OcclusionAwareGenerator(**tiny_config())
    assert all(p.requires_grad for p in gen.parameters())            # PyTorch's default, as the reference module
    assert len(torch.optim.Adam(gen.parameters()).param_groups[0]["params"]) == len(list(gen.parameters()))
    src = torch.zeros(2, 3, 64, 64)
    kp = {"value": torch.zeros(2, 10, 2), "jacobian": torch.eye(2).expand(2, 10, 2, 2).clone()}
    kpg = {k: v.clone().requires_grad_() for k, v in kp.items()}
    for mode in (gen.train(), gen.eval()):
        assert mode._wants_graph(src, kp, kp)
        with torch.no_grad():
            assert not mode._wants_graph(src, kpg, kp)
    gen.requires_grad_(False)                                        # a frozen generator: only an input can ask for a graph
    assert not gen._wants_graph(src, kp, kp)
    assert gen._wants_graph(src, kpg, kp)                            # the audio-to-key-point stage trains through the generator
    assert gen._wants_graph(src.clone().requires_grad_(), kp, kp)
    gen.requires_grad_(True)
    with pytest.raises(RuntimeError, match="ROCm GPU"):             # ... and still has no CPU fallback
        gen(src, kp_driving=kp, kp_source=kp)


def test_tensor_slots_follow_structural_changes():
    """the cached slot list must see replaced sub-modules, late-assigned tensors and copies."""
    import copy
    from eamm_amd import OcclusionAwareGenerator, tiny_config
    gen = OcclusionAwareGenerator(**tiny_config())
    v0 = gen._weights_version()
    assert gen._weights_version() == v0
    gen.bottleneck[0] = type(gen.bottleneck[0])(128)                 # Sequential item assignment
    v1 = gen._weights_version()
    assert v1 != v0
    gen.first = type(gen.first)(3, 32, 7)                            # attribute assignment of a sub-module
    v2 = gen._weights_version()
    assert v2 != v1
    gen.final.bias = None                                            # a slot that holds None ...
    v3 = gen._weights_version()
    assert len(v3) == len(v2) - 2
    gen.final.bias = torch.nn.Parameter(torch.zeros(3))              # ... and is assigned later
    assert len(gen._weights_version()) == len(v2)
    twin = copy.deepcopy(gen)
    ids = set(twin._weights_version()[:len(v2) // 2])
    assert ids <= {id(t) for t in list(twin.parameters()) + list(twin.buffers())}
    assert not ids & set(gen._weights_version()[:len(v2) // 2])      # the copy reads ITS tensors, not the original's


def test_epoch_sees_container_mutators_and_submodule_apply():
    """(1) ``ModuleList.insert`` / ``Sequential.insert`` / ``pop`` / ``append`` write ``_modules`` directly -- no registration
    call -- and (2) ``_apply`` on a SUB-module (``gen.first.double()``) changes neither a tensor's identity nor its version counter: both
    must move the generator's structure epoch, or ``_weights_unchanged()`` keeps an engine with stale weights."""
    from eamm_amd import OcclusionAwareGenerator, tiny_config
    gen = OcclusionAwareGenerator(**tiny_config())
    gen._tensor_slots()
    gen._remember_weights()
    assert gen._weights_unchanged()
    n_slots = len(gen._tensor_slots())
    blk = type(gen.down_blocks[0])(32, 32, 3)
    gen.down_blocks.insert(0, blk)                                   # (structurally nonsense; the point is that it is SEEN)
    assert not gen._weights_unchanged()
    assert len(gen._tensor_slots()) > n_slots
    gen._remember_weights()
    assert gen._weights_unchanged()
    gen.down_blocks.pop(0)
    assert not gen._weights_unchanged() and len(gen._tensor_slots()) == n_slots
    gen._remember_weights()
    from eamm_amd.generator import _Sequential
    res = type(gen.bottleneck[0])
    gen.extra = _Sequential(res(128), res(128))                      # (the bottleneck itself has NAMED entries r0..r5, which
    gen._remember_weights()                                          #  torch's own Sequential.insert cannot renumber)
    assert gen._weights_unchanged()
    gen.extra.insert(0, res(128))                                    # nn.Sequential.insert
    assert not gen._weights_unchanged()
    gen._remember_weights()
    del gen.extra[0]
    assert not gen._weights_unchanged()
    del gen.extra
    gen._remember_weights()
    assert gen._weights_unchanged()
    before = gen.first.conv.weight
    gen.first.double()                                               # _apply on a sub-module: same Parameter object, same version
    assert gen.first.conv.weight is before and before.dtype == torch.float64
    assert not gen._weights_unchanged()
    gen._remember_weights()
    gen.dense_motion_network.hourglass.float()
    assert not gen._weights_unchanged()


def test_clip_interface_refuses_inputs_that_require_grad():
    """encode_source / forward_frames are the graph-free inference path; an input that asks for a gradient is refused
    (forward() is the differentiable entry) instead of being detached silently."""
    from eamm_amd import OcclusionAwareGenerator, tiny_config
    gen = OcclusionAwareGenerator(**tiny_config()).eval()
    with pytest.raises(RuntimeError, match="graph-free inference path"):
        gen.encode_source(torch.zeros(1, 3, 64, 64, requires_grad=True))
    kp = {"value": torch.zeros(1, 10, 2)}
    with pytest.raises(RuntimeError, match="graph-free inference path"):
        gen.forward_frames({"value": torch.zeros(1, 10, 2, requires_grad=True)}, kp)
    with torch.no_grad(), pytest.raises(RuntimeError, match="encode_source"):     # under no_grad the flag is moot: the normal checks apply
        gen.forward_frames({"value": torch.zeros(1, 10, 2, requires_grad=True)}, kp)
