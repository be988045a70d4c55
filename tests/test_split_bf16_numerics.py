"""Numerics of VERDICT r02's exploratory item 9 -- a 3-term bf16 split of the bottleneck GEMM's operands (V = B^T d B and
U = G g G^T of the F(4x4,3x3) form, each as hi + mid + lo in bfloat16), the six products hi*hi, hi*mid, mid*hi, hi*lo, mid*mid,
lo*hi accumulated in fp32 -- measured on the CPU in exact arithmetic of the formats (a bf16 x bf16 product is exact in fp32),
on one 3x3 256 -> 256 convolution at 64 x 64 with the bottleneck's value ranges (reference modules/util.py:872-880).

No kernel is built for it (the default path must stay exact fp32 and the item may never be the headline); this pins the one
fact that decides whether it is worth building: the split's error against float64 is within ~1.5x of the fp32 Winograd form's
own -- i.e. inside the reference's fp32-vs-fp64 floor -- because the three dropped products are of order 2^-24 of a term."""
import numpy as np
import torch

BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                   [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                  [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def split3(t32):
    """fp32 -> three bfloat16 terms (as fp32 tensors holding bf16 values): t = hi + mid + lo up to ~2^-24 |t|."""
    hi = t32.to(torch.bfloat16).to(torch.float32)
    r = t32 - hi
    mid = r.to(torch.bfloat16).to(torch.float32)
    lo = (r - mid).to(torch.bfloat16).to(torch.float32)
    return hi, mid, lo


def winograd(x, w, product):
    """F(4x4,3x3) of x [C,H,W] with w [Co,C,3,3] (pad 1); `product(V, U)` does the 36 GEMMs V [36,T,C] x U [36,C,Co]."""
    C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    d = xp.unfold(1, 6, 4).unfold(2, 6, 4)                                   # [C, H/4, W/4, 6, 6]
    V = torch.einsum("ia,cyxab,jb->ijyxc", BT.to(x.dtype), d, BT.to(x.dtype)).reshape(36, -1, C)      # fp32 transform, as the kernel
    U = torch.einsum("ia,ocab,jb->ijco", G, w.double(), G).reshape(36, C, -1).to(x.dtype)           # double on the host, rounded once
    M = product(V, U).reshape(6, 6, H // 4, W // 4, -1)
    Y = torch.einsum("pi,ijyxo,qj->oypxq", AT.to(M.dtype), M, AT.to(M.dtype))
    return Y.reshape(-1, H, W)


def test_three_term_bf16_split_stays_within_the_fp32_winograd_error():
    g = torch.Generator().manual_seed(0)
    C = Co = 256
    H = W = 64
    x = torch.relu(torch.randn(C, H, W, generator=g))                        # pre-activated features: relu(norm(.))
    w = torch.randn(Co, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    truth = torch.nn.functional.conv2d(x.double()[None], w.double(), padding=1)[0]
    scale = float(truth.abs().max())

    fp32 = winograd(x, w, lambda V, U: torch.bmm(V, U))

    def split_product(V, U):
        v, u = split3(V), split3(U)
        acc = torch.zeros(V.shape[0], V.shape[1], U.shape[2])
        for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):         # small terms first
            acc = acc + torch.bmm(v[i], u[j])                                 # bf16 x bf16 products are exact in fp32
        return acc

    split = winograd(x, w, split_product)
    e32 = float((fp32.double() - truth).abs().max()) / scale
    esp = float((split.double() - truth).abs().max()) / scale
    direct = float((torch.nn.functional.conv2d(x[None], w, padding=1)[0].double() - truth).abs().max()) / scale
    print(f"relative max error vs float64: direct fp32 {direct:.2e}, F(4x4) fp32 {e32:.2e}, F(4x4) 3-term bf16 split {esp:.2e}")
    # measured: direct 2.3e-7, F(4x4) fp32 1.1e-5 (the transform-domain sums cancel: the op tests' bar is 6e-5), split 4.8e-6
    assert e32 < 3e-5 and esp < 1.5 * e32
    # two terms (three products) would NOT do: ~2^-16 per term
    def split2(V, U):
        v, u = split3(V), split3(U)
        return torch.bmm(v[1], u[0]) + torch.bmm(v[0], u[1]) + torch.bmm(v[0], u[0])
    e2 = float((winograd(x, w, split2).double() - truth).abs().max()) / scale
    print(f"two-term split (three products): {e2:.2e}")
    assert e2 > 2 * e32
