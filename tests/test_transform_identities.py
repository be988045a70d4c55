"""The algebraic identities the transform-domain kernels rest on, checked in float64 on the CPU (no GPU, no library):
* polyphase minimal filtering of UpBlock2d's "nearest x2 -> 3x3 convolution" (csrc/conv_mfma_patch_poly.hip):
  nine products per low-resolution pixel and channel give its four output phases;
* Winograd F(4x4,3x3) with the interpolation points {0, +-1, +-2, inf} (csrc/conv_winograd4.hip), incl. the
  2x2 average of DownBlock2d taken after the output transform (the hourglass encoder levels).
Reference semantics: modules/util.py:883-900 (UpBlock2d), :903-921 (DownBlock2d), :872-880 (ResBlock2d convolutions)."""
import numpy as np
import pytest

# polyphase: V = T e T^T, U = G w G^T, out(py, px) = sum_ij A[py][i] A[px][j] U_ij V_ij
T = np.array([[0, 1, 0], [1, -1, 0], [0, -1, 1]], dtype=np.float64)
G = np.array([[1, 1, 1], [1, 0, 0], [0, 0, 1]], dtype=np.float64)
A = np.array([[1, 1, 0], [1, 0, 1]], dtype=np.float64)

# Winograd F(4x4,3x3)
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], dtype=np.float64)
GW = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def conv3x3_same(x, w):
    """x [H,W,C], w [Co,C,3,3] -> [H,W,Co], zero padding 1 (cross-correlation, as torch.nn.Conv2d)."""
    H, W, _ = x.shape
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    out = np.zeros((H, W, w.shape[0]))
    for dy in range(3):
        for dx in range(3):
            out += xp[dy:dy + H, dx:dx + W, :] @ w[:, :, dy, dx].T
    return out


@pytest.mark.parametrize("seed", [0, 1])
def test_polyphase_nine_products_equal_nearest_upsample_then_conv(seed):
    rng = np.random.default_rng(seed)
    H, W, C, Co = 5, 6, 3, 4
    e = rng.standard_normal((H, W, C))
    w = rng.standard_normal((Co, C, 3, 3))
    want = conv3x3_same(np.repeat(np.repeat(e, 2, axis=0), 2, axis=1), w)          # [2H,2W,Co]
    ep = np.pad(e, ((1, 1), (1, 1), (0, 0)))                                       # zero padding of the UP-SAMPLED map == of e
    U = np.einsum("ia,ocab,jb->ijoc", G, w, G)                                     # [3,3,Co,C]
    got = np.zeros_like(want)
    for y in range(H):
        for x in range(W):
            nb = ep[y:y + 3, x:x + 3, :]                                           # 3x3 low-resolution neighbourhood
            V = np.einsum("ia,abc,jb->ijc", T, nb, T)                              # differences with the centre
            M = np.einsum("ijoc,ijc->ijo", U, V)                                   # the nine products, summed over channels
            for py in range(2):
                for px in range(2):
                    got[2 * y + py, 2 * x + px] = np.einsum("i,j,ijo->o", A[py], A[px], M)
    assert np.abs(got - want).max() < 1e-12
    # the output-side coefficients are 0 / 1 only: every product enters a phase at most once, none with a minus sign
    assert set(np.unique(A)) == {0.0, 1.0} and set(np.unique(np.abs(T))) == {0.0, 1.0}


def test_polyphase_padding_of_the_upsampled_map_is_padding_of_the_low_resolution_map():
    """At the image border the 3x3 window of the up-sampled map reaches one up-sampled pixel outside = half a low-resolution
    pixel: the polyphase form reads a whole zero low-resolution pixel there, which is the same thing because every tap that
    falls outside multiplies zero either way (checked above on the border pixels; here on a 1x1 map, all border)."""
    e = np.array([[[2.0]]])
    w = np.arange(9, dtype=np.float64).reshape(1, 1, 3, 3) + 1
    want = conv3x3_same(np.repeat(np.repeat(e, 2, 0), 2, 1), w)[..., 0]
    U = np.einsum("ia,ocab,jb->ijoc", G, w, G)[..., 0, 0]
    V = np.einsum("ia,ab,jb->ij", T, np.pad(e[..., 0], 1), T)
    got = np.array([[np.einsum("i,j,ij->", A[py], A[px], U * V) for px in range(2)] for py in range(2)])
    assert np.abs(got - want).max() < 1e-12


@pytest.mark.parametrize("pool", [False, True])
def test_winograd_f4x4_equals_conv_and_pools_after_the_output_transform(pool):
    rng = np.random.default_rng(3)
    H, W, C, Co = 8, 12, 2, 3
    x = rng.standard_normal((H, W, C))
    w = rng.standard_normal((Co, C, 3, 3))
    b = rng.standard_normal(Co)
    want = np.maximum(conv3x3_same(x, w) + b, 0.0)
    if pool:
        want = want.reshape(H // 2, 2, W // 2, 2, Co).mean(axis=(1, 3))
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    Uw = np.einsum("ia,ocab,jb->ijoc", GW, w, GW)                                  # [6,6,Co,C]
    got = np.zeros((H, W, Co))
    for ty in range(H // 4):
        for tx in range(W // 4):
            d = xp[4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6, :]
            V = np.einsum("ia,abc,jb->ijc", BT, d, BT)
            M = np.einsum("ijoc,ijc->ijo", Uw, V)
            Z = np.einsum("qj,ijo->iqo", AT, M)                                    # x fold (in the GEMM kernel's registers)
            Y = np.einsum("pi,iqo->pqo", AT, Z)                                    # y fold (GEMM epilogue / output-transform kernel)
            got[4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = Y
    got = np.maximum(got + b, 0.0)
    if pool:
        got = got.reshape(H // 2, 2, W // 2, 2, Co).mean(axis=(1, 3))               # the four pixels of a window sit in ONE 4x4 tile
    assert np.abs(got - want).max() < 1e-10


def test_first_block_reduction_order_covers_every_tap_and_colour_once():
    """csrc/conv_first.hip enumerates K as k = 4 * tap + c (c = 3: a zero row); K step s of the MFMA covers k = 2 s + half
    with tap = s >> 1 and c = 2 (s & 1) + half."""
    seen = set()
    for s in range(98):
        for half in range(2):
            tap, c = s >> 1, 2 * (s & 1) + half
            assert 4 * tap + c == 2 * s + half
            seen.add((tap, c))
    assert seen == {(t, c) for t in range(49) for c in range(4)}


# Winograd F(3x3,4x4) for the WEIGHT gradient (csrc/backward.hip): the filter gradient of a 4x4 output tile is the same
# correlation with filter and output exchanged; same points, so B^T d B is the forward's transformed input
GW4 = np.array([[1 / 4, 0, 0, 0], [-1 / 6, -1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6, 1 / 6], [1 / 24, 1 / 12, 1 / 6, 1 / 3],
                [1 / 24, -1 / 12, 1 / 6, -1 / 3], [0, 0, 0, 1]], dtype=np.float64)
AT3 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 1]], dtype=np.float64)


def test_winograd_f3x3_4x4_gives_the_weight_gradient_of_a_same_convolution():
    """dW[co][ci] = sum over 4x4 tiles of A'^T [(G' dY_tile G'^T) (.) (B^T patch B)] A' equals the gradient autograd derives for
    F.conv2d(x, w, padding=1) (reference modules/util.py:858-938), zero padding included (patches clipped at the borders)."""
    rng = np.random.default_rng(5)
    H, W, C, Co = 8, 12, 3, 2
    x = rng.standard_normal((H, W, C))
    dy = rng.standard_normal((H, W, Co))
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    want = np.zeros((Co, C, 3, 3))
    for k in range(3):
        for l in range(3):
            want[:, :, k, l] = np.einsum("hwo,hwc->oc", dy, xp[k:k + H, l:l + W, :])
    got = np.zeros((Co, C, 3, 3))
    m = np.zeros((6, 6, Co, C))
    for qy in range(H // 4):
        for qx in range(W // 4):
            d = xp[4 * qy:4 * qy + 6, 4 * qx:4 * qx + 6, :]                      # patch origin (4qy - 1, 4qx - 1) of the unpadded map
            g = dy[4 * qy:4 * qy + 4, 4 * qx:4 * qx + 4, :]
            V = np.einsum("ia,abc,jb->ijc", BT, d, BT)
            Yh = np.einsum("ia,abo,jb->ijo", GW4, g, GW4)
            m += np.einsum("ijo,ijc->ijoc", Yh, V)                                # the 36 GEMMs' K step
    got = np.einsum("ki,ijoc,lj->ockl", AT3, m, AT3)
    assert np.abs(got - want).max() < 1e-11
