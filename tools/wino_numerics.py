"""CPU experiment: error of fp32 Winograd variants for the bottleneck 3x3 convs, measured at the prediction.

Runs the oracle's generator at 256^2 with the bottleneck convs replaced by an fp32 Winograd emulation
(F(2x2), F(2x4), F(4x4); transforms in fp32, weights transformed in fp64 then rounded) and reports max |diff|
against the fp64 oracle.  Test tooling only (imports oracle/)."""
import sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import eamm_oracle as O
from eamm_amd.config import hot_path_config
from eamm_amd.weights import synthetic_state_dict, synthetic_source, synthetic_keypoints

MATS = {
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)),
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                  [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], float),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)),
}


def wino_conv(x, w, b, my, mx):
    """x [B,C,H,W] fp32, 3x3 pad 1; output tile my x mx."""
    BTy, Gy, ATy = MATS[my]; BTx, Gx, ATx = MATS[mx]
    U = np.einsum("ia,ocab,jb->ijoc", Gy, w.double().numpy(), Gx)            # fp64 weight transform
    U = torch.from_numpy(U).float()
    B_, C, H, W = x.shape
    ty, tx = my + 2, mx + 2
    xp = F.pad(x, (1, 1 + (-W) % mx, 1, 1 + (-H) % my))
    tiles = xp.unfold(2, ty, my).unfold(3, tx, mx)                            # [B,C,nty,ntx,ty,tx]
    BTy32, BTx32 = torch.from_numpy(BTy).float(), torch.from_numpy(BTx).float()
    V = torch.einsum("ia,bcyxad,jd->ijbcyx", BTy32, tiles, BTx32)             # fp32
    M = torch.einsum("ijoc,ijbcyx->ijboyx", U, V)
    Y = torch.einsum("pi,ijboyx,qj->boypxq", torch.from_numpy(ATy).float(), M, torch.from_numpy(ATx).float())
    nty, ntx = tiles.shape[2], tiles.shape[3]
    Y = Y.reshape(B_, w.shape[0], nty * my, ntx * mx)[:, :, :H, :W]
    return Y + b.view(1, -1, 1, 1)


def run(mode):
    cfg = hot_path_config()
    sd = synthetic_state_dict(cfg)
    src = synthetic_source(256); kd = synthetic_keypoints(2); ks = synthetic_keypoints(1, seed=100)
    ks = {k: v.expand(2, *v.shape[1:]) for k, v in ks.items()}
    orig = O._conv
    if mode is not None:
        def conv(x, sd_, prefix, pad):
            if prefix.startswith("bottleneck"):
                return wino_conv(x, sd_[prefix + ".weight"], sd_[prefix + ".bias"], *mode)
            return orig(x, sd_, prefix, pad)
        O._conv = conv
    try:
        out = O.generator_forward(sd, cfg, src.expand(2, -1, -1, -1), kd, ks)
    finally:
        O._conv = orig
    return out["prediction"]


if __name__ == "__main__":
    torch.set_num_threads(32)
    cfg = hot_path_config()
    sd64 = {k: v.double() for k, v in synthetic_state_dict(cfg).items()}
    src = synthetic_source(256); kd = synthetic_keypoints(2); ks = synthetic_keypoints(1, seed=100)
    ks = {k: v.expand(2, *v.shape[1:]) for k, v in ks.items()}
    ref64 = O.generator_forward(sd64, cfg, src.expand(2, -1, -1, -1).double(), {k: v.double() for k, v in kd.items()},
                                {k: v.double() for k, v in ks.items()})["prediction"]
    for name, mode in (("direct fp32", None), ("F(2x2)", (2, 2)), ("F(2x4)", (2, 4)), ("F(4x4)", (4, 4))):
        p = run(mode)
        print(f"{name:12s} max|pred - fp64| = {(p.double() - ref64).abs().max().item():.3e}", flush=True)
