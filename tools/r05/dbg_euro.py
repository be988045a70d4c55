import sys, torch
sys.path.insert(0, '.')
from eamm_amd import one_euro_smooth
g = torch.Generator().manual_seed(0)
T = 37
seq = (torch.eye(2)[None, None] + 0.02 * torch.cumsum(torch.randn(T, 10, 2, 2, generator=g), 0)).cuda()
kw = dict(mincutoff=0.05, beta=8.0, dcutoff=1.0, freq=100.0, scale=10.0)
whole = one_euro_smooth(seq, **kw)
for cuts in ((0, 8, 16, 24, 32, 37), (0, 1, 20, 37), (0, 1, 37), (0, 20, 37), (0, 17, 37), (0, 16, 37), (0, 2, 37)):
    state = torch.zeros(3, 40, device="cuda")
    parts = [one_euro_smooth(seq[a:b], state=state, resume=a > 0, **kw) for a, b in zip(cuts[:-1], cuts[1:])]
    got = torch.cat(parts)
    d = (got - whole).abs().flatten(1).max(1).values
    print(cuts, "first differing frame:", (d > 0).nonzero().flatten().tolist()[:5], "max", float(d.max()))
