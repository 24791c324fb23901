import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bsms_gnn_amd as eng
torch.manual_seed(21)
D = 128
# MLP input gradients: small launch vs the same rows inside a large one
for in_dim, out_dim, ln in ((D, D, True), (D, 3, False)):
    mlp = eng.MLP(in_dim, D, out_dim, 3, ln).cuda()
    x = torch.randn(30000, in_dim, device="cuda")
    r = torch.randn(30000, out_dim, device="cuda")
    xs = x[:5000].clone().requires_grad_(True); xb = x.clone().requires_grad_(True)
    (mlp(xs) * r[:5000]).sum().backward(); (mlp(xb) * r).sum().backward()
    d = (xs.grad - xb.grad[:5000]).abs().max().item()
    print(f"MLP {in_dim}->{out_dim} ln={ln}: input-gradient max|small - large| = {d:.3e}  (equal: {torch.equal(xs.grad, xb.grad[:5000])})")
# GMP pieces
rng = np.random.default_rng(5)
n, e = 5000, 30000
g = torch.tensor(np.stack([rng.integers(0, n, e), rng.integers(0, n - n // 8, e)]), dtype=torch.int64).cuda()
gmp = eng.GMP(D, 3, 2).cuda()
x1, p1 = torch.randn(1, n, D, device="cuda"), torch.rand(1, n, 2, device="cuda")
def step(xx, pp):
    gmp.zero_grad(set_to_none=True)
    xx = xx.clone().requires_grad_(True)
    y = gmp(xx, g, pp)
    (y * y).sum().backward()
    return y.detach(), xx.grad
ya, ga = step(x1, p1)
yb, gb = step(x1.repeat(6, 1, 1).contiguous(), p1.repeat(6, 1, 1).contiguous())
print("GMP fwd equal:", torch.equal(ya, yb[4:5]), " grad equal:", torch.equal(ga, gb[4:5]), " max diff", (ga - gb[4:5]).abs().max().item(), "scale", ga.abs().max().item())
yc, gc = step(x1.repeat(2, 1, 1).contiguous(), p1.repeat(2, 1, 1).contiguous())
print("B=2 vs B=1 grad equal:", torch.equal(ga, gc[1:2]), " B=2 vs B=6:", torch.equal(gc[1:2], gb[4:5]))
