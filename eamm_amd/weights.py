"""Checkpoint layout of the generator and seeded synthetic weights / inputs.

``state_dict_spec(cfg)`` derives, from the constructor kwargs alone, the exact key set and
shapes that ``generator.load_state_dict(checkpoint['generator'])`` expects in the reference
(reference demo.py:91; key set listed in SURVEY.md section 8b): conv weights are OIHW, every
``norm`` carries ``weight, bias, running_mean, running_var, num_batches_tracked`` and the
dense-motion network owns the ``down.weight`` anti-alias buffer.

No pretrained checkpoint ships with the reference (README.md:21 is a download link), so parity is
pinned with synthetic weights produced by a frozen ``numpy.random.RandomState`` stream: He-scaled
convolutions and non-trivial BatchNorm statistics (PyTorch's default init makes the prediction
almost constant, which would let a broken kernel pass).  The same generator is used in the
container (to drive the imported reference) and on the GPU box, so 182 MB of weights never have to
be committed.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch


# ----------------------------------------------------------------------------------------------
# layout
# ----------------------------------------------------------------------------------------------
def _block(prefix, cin, cout, k, spec, gain):
    spec.append((prefix + ".conv.weight", (cout, cin, k, k), "conv_w", gain))
    spec.append((prefix + ".conv.bias", (cout,), "conv_b", None))
    _norm(prefix + ".norm", cout, spec)


def _norm(prefix, c, spec):
    spec.append((prefix + ".weight", (c,), "bn_w", None))
    spec.append((prefix + ".bias", (c,), "bn_b", None))
    spec.append((prefix + ".running_mean", (c,), "bn_mean", None))
    spec.append((prefix + ".running_var", (c,), "bn_var", None))
    spec.append((prefix + ".num_batches_tracked", (), "nbt", None))


def hourglass_channels(block_expansion, in_features, num_blocks, max_features):
    """(encoder [(cin, cout)], decoder [(cin, cout)], out_filters) -- reference util.py:941-987."""
    enc = []
    for i in range(num_blocks):
        cin = in_features if i == 0 else min(max_features, block_expansion * (2 ** i))
        cout = min(max_features, block_expansion * (2 ** (i + 1)))
        enc.append((cin, cout))
    dec = []
    for i in reversed(range(num_blocks)):
        mult = 1 if i == num_blocks - 1 else 2
        cin = mult * min(max_features, block_expansion * (2 ** (i + 1)))
        cout = min(max_features, block_expansion * (2 ** i))
        dec.append((cin, cout))
    return enc, dec, block_expansion + in_features


def generator_channels(cfg):
    """Encoder / decoder channel plan of the generator (reference generator.py:25-46)."""
    be, mf, nd = cfg["block_expansion"], cfg["max_features"], cfg["num_down_blocks"]
    down = [(min(mf, be * 2 ** i), min(mf, be * 2 ** (i + 1))) for i in range(nd)]
    up = [(min(mf, be * 2 ** (nd - i)), min(mf, be * 2 ** (nd - i - 1))) for i in range(nd)]
    return down, up, min(mf, be * 2 ** nd)


def state_dict_spec(cfg):
    """Ordered [(key, shape, kind, gain)] in the reference's registration order."""
    spec = []
    nc, nk = cfg["num_channels"], cfg["num_kp"]
    dm = cfg.get("dense_motion_params")
    if dm is not None:
        enc, dec, out_filters = hourglass_channels(
            dm["block_expansion"], (nk + 1) * (nc + 1), dm["num_blocks"], dm["max_features"])
        p = "dense_motion_network."
        for i, (ci, co) in enumerate(enc):
            _block(f"{p}hourglass.encoder.down_blocks.{i}", ci, co, 3, spec, 2.0)
        for i, (ci, co) in enumerate(dec):
            _block(f"{p}hourglass.decoder.up_blocks.{i}", ci, co, 3, spec, 2.0)
        spec.append((p + "mask.weight", (nk + 1, out_filters, 7, 7), "conv_w", 4.0))
        spec.append((p + "mask.bias", (nk + 1,), "conv_b", None))
        if cfg.get("estimate_occlusion_map", False):
            spec.append((p + "occlusion.weight", (1, out_filters, 7, 7), "conv_w", 4.0))
            spec.append((p + "occlusion.bias", (1,), "conv_b", None))
        if dm.get("scale_factor", 1) != 1:
            spec.append((p + "down.weight", (nc, 1, 13, 13), "aa", None))
    down, up, bott = generator_channels(cfg)
    _block("first", nc, cfg["block_expansion"], 7, spec, 2.0)
    for i, (ci, co) in enumerate(down):
        _block(f"down_blocks.{i}", ci, co, 3, spec, 2.0)
    for i, (ci, co) in enumerate(up):
        _block(f"up_blocks.{i}", ci, co, 3, spec, 2.0)
    for i in range(cfg["num_bottleneck_blocks"]):
        r = f"bottleneck.r{i}"
        spec.append((r + ".conv1.weight", (bott, bott, 3, 3), "conv_w", 1.0))
        spec.append((r + ".conv1.bias", (bott,), "conv_b", None))
        spec.append((r + ".conv2.weight", (bott, bott, 3, 3), "conv_w", 1.0))
        spec.append((r + ".conv2.bias", (bott,), "conv_b", None))
        _norm(r + ".norm1", bott, spec)
        _norm(r + ".norm2", bott, spec)
    spec.append(("final.weight", (nc, cfg["block_expansion"], 7, 7), "conv_w", 1.0))
    spec.append(("final.bias", (nc,), "conv_b", None))
    return spec


def kp_state_dict_spec(cfg):
    """Checkpoint layout of KPDetector / KPDetector_a (reference keypoint_detector.py:12-37, 115-141): the
    `predictor` hourglass (unused by KPDetector_a.forward but present in its checkpoint), the 7x7 `kp` and
    `jacobian` heads and the anti-alias buffer."""
    spec = []
    in_features = cfg.get("num_channels_a", cfg["num_channels"])
    enc, dec, out_filters = hourglass_channels(cfg["block_expansion"], in_features, cfg["num_blocks"], cfg["max_features"])
    for i, (ci, co) in enumerate(enc):
        _block(f"predictor.encoder.down_blocks.{i}", ci, co, 3, spec, 2.0)
    for i, (ci, co) in enumerate(dec):
        _block(f"predictor.decoder.up_blocks.{i}", ci, co, 3, spec, 2.0)
    k = cfg["num_kp"]
    spec.append(("kp.weight", (k, out_filters, 7, 7), "conv_w", 1.0))
    spec.append(("kp.bias", (k,), "conv_b", None))
    if cfg.get("estimate_jacobian", False):
        njm = 1 if cfg.get("single_jacobian_map", False) else k
        spec.append(("jacobian.weight", (4 * njm, out_filters, 7, 7), "conv_w", 1.0))
        spec.append(("jacobian.bias", (4 * njm,), "conv_b", None))
    if cfg.get("scale_factor", 1) != 1:
        spec.append(("down.weight", (cfg["num_channels"], 1, 13, 13), "aa", None))
    return spec


DECONV_CHANNELS = (256, 256, 128, 128, 128, 35)   # AT_net2.decon, reference util.py:559-574


def deconv_state_dict_spec(channels=DECONV_CHANNELS):
    """Checkpoint layout of the nn.Sequential `AT_net2.decon` (reference util.py:559-576): ConvTranspose2d at index
    3i ([Cin,Cout,k,k], k = 6 for the first layer, 4 after), BatchNorm2d at 3i+1 for every layer but the last."""
    spec = []
    n = len(channels) - 1
    for i in range(n):
        k = 6 if i == 0 else 4
        # "conv_w" draws with fan_in = shape[1]*k*k; a stride-2 transposed conv sums k*k/4 taps of shape[0] inputs
        spec.append((f"{3 * i}.weight", (channels[i], channels[i + 1], k, k), "convT_w", 2.0))
        spec.append((f"{3 * i}.bias", (channels[i + 1],), "conv_b", None))
        if i + 1 < n:
            _norm(f"{3 * i + 1}", channels[i + 1], spec)
    return spec


def antialias_kernel(channels: int, sigma: float = 1.5) -> torch.Tensor:
    """The fixed 13x13 Gaussian buffer of the anti-alias down-sampler.

    Reference util.py:1009-1035: sigma is hard-coded to 1.5 regardless of the scale, the window is
    2*round(4*sigma)+1 = 13 taps, normalised to sum 1, replicated per channel as a depthwise weight.
    """
    size = 2 * round(sigma * 4) + 1
    ax = torch.arange(size, dtype=torch.float32)
    mean = (size - 1) / 2
    g = torch.exp(-(ax - mean) ** 2 / (2 * sigma ** 2))
    k2 = g[:, None] * g[None, :]
    k2 = k2 / torch.sum(k2)
    return k2.view(1, 1, size, size).repeat(channels, 1, 1, 1).contiguous()


# ----------------------------------------------------------------------------------------------
# seeded synthetic weights and inputs (frozen RandomState stream)
# ----------------------------------------------------------------------------------------------
def synthetic_state_dict(cfg, seed: int = 1234, spec=None) -> "OrderedDict[str, torch.Tensor]":
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for key, shape, kind, gain in (state_dict_spec(cfg) if spec is None else spec):
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            v = rs.standard_normal(shape) * math.sqrt(gain / fan_in)
        elif kind == "convT_w":
            fan_in = shape[0] * shape[2] * shape[3] / 4.0
            v = rs.standard_normal(shape) * math.sqrt(gain / fan_in)
        elif kind == "conv_b":
            v = 0.1 * rs.standard_normal(shape)
        elif kind == "bn_w":
            v = rs.uniform(0.75, 1.25, shape)
        elif kind in ("bn_b", "bn_mean"):
            v = 0.1 * rs.standard_normal(shape)
        elif kind == "bn_var":
            v = rs.uniform(0.75, 1.25, shape)
        elif kind == "nbt":
            sd[key] = torch.tensor(1000, dtype=torch.int64)
            continue
        elif kind == "aa":
            sd[key] = antialias_kernel(shape[0])
            continue
        else:  # pragma: no cover
            raise ValueError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def adversarial_state_dict(cfg, seed: int = 1234, bn_stats=None) -> "OrderedDict[str, torch.Tensor]":
    """Seeded synthetic generator weights with the statistics of a badly conditioned TRAINED checkpoint (VERDICT r04 item 6;
    the benign `synthetic_state_dict` keeps BatchNorm variance and gain in [0.75, 1.25]):
      * every convolution in front of a BatchNorm has per-output-channel scales 10^U(-2, 0.5) (pre-normalisation standard
        deviations 0.01 .. 3.2, variances 1e-4 .. 10) and twice the He gain in the weights (a gain of 4 in variance);
      * 3 % of those channels are DEAD: weights x 1e-3 with the bias kept, i.e. a constant plus a fluctuation a thousand times
        smaller, which BatchNorm then amplifies by gamma / sqrt(var + eps) ~ 100 .. 300 (reference batchnorm.py:48-53);
      * BatchNorm gains 10^U(-0.6, 0.6) = 0.25 .. 4, one in ten negative; shifts 0.3 N(0, 1).
    The running statistics cannot be drawn at random -- activations would explode or vanish through 30 layers; a trained
    network's statistics are those of its own activations -- so they come from ``bn_stats`` ({norm prefix: (mean, var)}):
    the fixture generator (make_golden.py) calibrates them with one batch-statistics pass of the reference-equivalent forward on the fixture's
    inputs, perturbs the variances by 10^U(-0.3, 0.3) and stores them IN the fixture (a few hundred kB), so that the GPU box
    rebuilds exactly these weights from the seed + the fixture.  Without ``bn_stats``: mean 0, variance 1 (calibration input)."""
    sd = synthetic_state_dict(cfg, seed=seed)
    rs = np.random.RandomState(seed + 4242)
    pairs = []   # (convolution prefix, norm prefix)
    for key in sd:
        if key.endswith(".conv.weight"):
            pairs.append((key[:-len(".weight")], key[:-len(".conv.weight")] + ".norm"))
        elif key.endswith(".conv1.weight"):
            pairs.append((key[:-len(".weight")], key[:-len(".conv1.weight")] + ".norm2"))
    for conv, norm in pairs:
        w, b = sd[conv + ".weight"], sd[conv + ".bias"]
        co = w.shape[0]
        scale = 10.0 ** rs.uniform(-2.0, 0.5, co)
        dead = rs.uniform(size=co) < 0.03
        wscale = np.where(dead, 1e-3, 2.0 * scale).astype(np.float32)
        sd[conv + ".weight"] = w * torch.from_numpy(wscale)[:, None, None, None]
        sd[conv + ".bias"] = b * torch.from_numpy(np.where(dead, 1.0, scale).astype(np.float32))
    for key in list(sd):
        if key.endswith(".running_var"):
            norm = key[:-len(".running_var")]
            c = sd[key].numel()
            gamma = 10.0 ** rs.uniform(-0.6, 0.6, c) * np.where(rs.uniform(size=c) < 0.1, -1.0, 1.0)
            sd[norm + ".weight"] = torch.from_numpy(gamma.astype(np.float32))
            sd[norm + ".bias"] = torch.from_numpy((0.3 * rs.standard_normal(c)).astype(np.float32))
            if bn_stats is None:
                sd[norm + ".running_mean"] = torch.zeros(c)
                sd[norm + ".running_var"] = torch.ones(c)
            else:
                mean, var = bn_stats[norm]
                sd[norm + ".running_mean"] = torch.as_tensor(mean, dtype=torch.float32).clone()
                sd[norm + ".running_var"] = torch.as_tensor(var, dtype=torch.float32).clone()
    return sd


def adversarial_inputs(size: int, frames: int, num_kp: int = 10, channels: int = 3):
    """Inputs for the adversarial fixtures: a SATURATED source (every pixel exactly 0 or 1) and key points on or just inside the
    frame border (|coordinate| in [0.9, 1.0]; a few at exactly +-1), jacobians I + 0.3 N(0,1)."""
    rs = np.random.RandomState(77)
    src = torch.from_numpy((rs.uniform(size=(1, channels, size, size)) > 0.5).astype(np.float32))

    def kps(n, seed):
        r = np.random.RandomState(seed)
        v = r.uniform(0.9, 1.0, (n, num_kp, 2)) * np.where(r.uniform(size=(n, num_kp, 2)) < 0.5, -1.0, 1.0)
        v[:, 0] = np.sign(v[:, 0])                       # key point 0 sits exactly in a corner
        v[:, 1:4] *= r.uniform(0.0, 1.0, (n, 3, 2))      # a few stay inside, or nothing of the source would be sampled
        j = np.eye(2)[None, None] + 0.3 * r.standard_normal((n, num_kp, 2, 2))
        return {"value": torch.from_numpy(v.astype(np.float32)), "jacobian": torch.from_numpy(j.astype(np.float32))}
    return src, kps(1, 300), kps(frames, 301)


def trained_like_kp_state_dict(cfg, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic key-point-detector weights with the jacobian head near its trained shape: the reference initialises that head to
    zero weights and an identity bias (modules/keypoint_detector.py:27-28) and training keeps the jacobians near I; He-scaled
    random weights instead give jacobians with |det| ~ 0.01, whose inverse (normalize_kp, demo.py:128; dense_motion.py:56)
    amplifies every rounding difference a hundredfold -- fine for a detector's own parity test, useless for a chain through the
    generator.  Here: the seeded random head scaled by 0.05 plus bias = identity + 0.05 N(0,1)."""
    sd = synthetic_state_dict(cfg, seed=seed, spec=kp_state_dict_spec(cfg))
    if "jacobian.weight" in sd:
        rs = np.random.RandomState(seed + 1000)
        n = sd["jacobian.bias"].numel() // 4
        sd["jacobian.weight"] = sd["jacobian.weight"] * 0.05
        sd["jacobian.bias"] = torch.from_numpy((np.tile([1.0, 0.0, 0.0, 1.0], n) + 0.05 * rs.standard_normal(4 * n)).astype(np.float32))
    return sd


def synthetic_lstm_features(frames: int, channels: int = 256, seed: int = 5) -> torch.Tensor:
    """[T,C] stand-in for the audio network's LSTM output (AT_net2, util.py:600-607): a random pose plus a slow random walk,
    so that consecutive frames' key points move like a talking head's (the One-Euro filter then has something to smooth)."""
    rs = np.random.RandomState(seed)
    base = 0.3 * rs.standard_normal((1, channels))
    walk = 0.05 * np.cumsum(rs.standard_normal((frames, channels)), axis=0)
    return torch.from_numpy((base + walk).astype(np.float32))


def synthetic_source(size: int, seed: int = 1, batch: int = 1, channels: int = 3) -> torch.Tensor:
    """uniform[0,1) RGB source, float32 [batch,3,size,size]; ``channels``: grey-scale / two-channel variants."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.uniform(0.0, 1.0, (batch, channels, size, size)).astype(np.float32))


def synthetic_keypoints(n: int, num_kp: int = 10, seed: int = 0, jacobian: bool = True) -> dict:
    """n keypoint sets: value ~ U[-0.8, 0.8], jacobian = I + 0.1 N(0,1) (well conditioned).

    Frame t of a clip uses seed 2+t, the source keypoints seed 0; each
    frame draws from its own stream so that a shard of a clip is reproducible in isolation.
    """
    vals, jacs = [], []
    for i in range(n):
        rs = np.random.RandomState(seed + i)
        vals.append(rs.uniform(-0.8, 0.8, (num_kp, 2)))
        jacs.append(np.eye(2)[None] + 0.1 * rs.standard_normal((num_kp, 2, 2)))
    kp = {"value": torch.from_numpy(np.stack(vals).astype(np.float32))}
    if jacobian:
        kp["jacobian"] = torch.from_numpy(np.stack(jacs).astype(np.float32))
    return kp
