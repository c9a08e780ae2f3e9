import torch, time
dev = torch.device("cuda:0")
B, C = 64, 3
def crit(inter, sets):
    dice = (2 * inter / (sets + 1e-6)).mean(dim=0)
    return 1 - dice.mean()
leaves = [torch.rand(B if i < 4 else 32, C, device=dev, requires_grad=True) for i in range(8)]
cs = [torch.rand(3, device=dev, requires_grad=True) for _ in range(4)]
ps = [torch.rand((), device=dev, requires_grad=True) for _ in range(4)]
def build():
    cps = crit(leaves[0], leaves[1]) + crit(leaves[2], leaves[3])
    s1, s2 = crit(leaves[4], leaves[5]), crit(leaves[6], leaves[7])
    com = (cs[0] + cs[1] + cs[2] + cs[3]) * 0.25
    pro = (ps[0] + ps[1] + ps[2] + ps[3]) * 0.01
    return s1 + s2 + 1.5 * cps + com.sum() + pro.float()
for _ in range(5):
    build().backward()
torch.cuda.synchronize()
for name in ("fwd", "bwd"):
    ts = []
    for _ in range(20):
        if name == "bwd":
            l = build(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "fwd": l = build()
        else: l.backward()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(name, f"{1e3 * sorted(ts)[len(ts) // 2]:.3f} ms wall (incl. sync)")
