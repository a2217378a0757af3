import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.fused_check import mk, dev, status
from outlier_suppression_amd import ops

def run(shape, lengths):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(*shape, device=dev, generator=g)
    x[..., 5] *= 20
    out = {}
    for fused in (1, 2, 0):
        ops.set_tuning("fused_step", 1 if fused else 0)
        q = mk()
        with torch.no_grad():
            out[fused] = q(x, lengths, 1).clone()
        torch.cuda.synchronize()
        s, z = q.scale.item(), q.zero_point.item()
    a, b, a2 = out[1].flatten(0, 1), out[0].flatten(0, 1), out[2].flatten(0, 1)
    xx = x.flatten(0, 1)
    print(shape, "fused run1 vs run2 equal:", torch.equal(a, a2), " s,z=", s, z)
    bad = (a != b)
    rows = bad.any(dim=1).nonzero().flatten()
    print(" bad rows", rows.numel(), " bad elems", int(bad.sum()))
    lanes = torch.zeros(64, 4, dtype=torch.long)
    r_, c_ = bad.nonzero(as_tuple=True)
    cc = c_.cpu()
    for comp in range(4):
        sel = cc[(cc % 4) == comp]
        lanes[:, comp] = torch.bincount((sel // 4) % 64, minlength=64)
    print(" per-lane bad counts (x,y,z,w):")
    for l in range(64):
        if lanes[l].sum():
            print("   lane", l, lanes[l].tolist())
    u = torch.bincount(cc // 256, minlength=shape[2] // 256)
    print(" per-u bad counts:", u.tolist())
    for k in range(min(6, r_.numel())):
        r, c = int(r_[k]), int(c_[k])
        print("   row", r, "col", c, "x=", xx[r, c].item(), "fused=", a[r, c].item(), "ref=", b[r, c].item(), " fused/s=", a[r, c].item() / s, " x/s=", xx[r, c].item() / s)

g = torch.Generator().manual_seed(0)
run((256, 128, 768), torch.randint(8, 129, (256,), generator=g).to(dev))
print("status", status())
