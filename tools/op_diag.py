import torch, torch.nn.functional as F
torch.manual_seed(0)
dev = torch.device("cuda:0")
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
def leaf(t, d): return t.detach().to(d).clone().requires_grad_(True)
for (c, n, hw) in [(64, 2, 32), (32, 2, 32), (2048, 2, 2), (3, 2, 32)]:
    x = torch.randn(n, c, hw, hw) * 3 + 1; g = torch.randn(n, c, hw, hw); wt = torch.rand(c) + 0.5; b = torch.randn(c)
    for cl in (False, True):
        def bn(d):
            xx = leaf(x, d); ww = leaf(wt, d); bb = leaf(b, d)
            xin = xx.contiguous(memory_format=torch.channels_last) if (cl and d != "cpu") else xx
            y = F.relu(F.batch_norm(xin, None, None, ww, bb, True, 0.1, 1e-5)); y.backward(g.to(d)); return y, xx.grad, ww.grad, bb.grad
        yc, gxc, gwc, gbc = bn("cpu"); yg, gxg, gwg, gbg = bn(dev)
        print(f"bn+relu c{c} n{n} hw{hw} cl={cl}: fwd {rel(yg,yc):.2e} dx {rel(gxg,gxc):.2e} dw {rel(gwg,gwc):.2e} db {rel(gbg,gbc):.2e}")
for ac in (True, False):
    x = torch.randn(2, 3, 32, 32); g = torch.randn(2, 3, 64, 64)
    def up(d):
        xx = leaf(x, d); y = F.interpolate(xx, scale_factor=2, mode="bilinear", align_corners=ac); y.backward(g.to(d)); return y, xx.grad
    yc, gc = up("cpu"); yg, gg = up(dev)
    print(f"upsample ac={ac}: fwd {rel(yg,yc):.2e} dx {rel(gg,gc):.2e}")
x = torch.randn(2, 64, 32, 32); g = torch.randn(2, 64, 16, 16)
def mp(d):
    xx = leaf(x, d); y = F.max_pool2d(xx, 3, 2, 1); y.backward(g.to(d)); return y, xx.grad
yc, gc = mp("cpu"); yg, gg = mp(dev); print(f"maxpool: fwd {rel(yg,yc):.2e} dx {rel(gg,gc):.2e}")
x = torch.randn(2, 32, 32, 32); w = torch.randn(3, 32, 1, 1); g = torch.randn(2, 3, 32, 32)
def head(d):
    xx = leaf(x, d); ww = leaf(w, d); y = F.conv2d(xx, ww); y.backward(g.to(d)); return y, xx.grad, ww.grad
yc, gc, wc = head("cpu"); yg, gg, wg = head(dev); print(f"head1x1: fwd {rel(yg,yc):.2e} dx {rel(gg,gc):.2e} dw {rel(wg,wc):.2e}")
x = torch.randn(2, 64, 33, 33); w = torch.randn(64, 64, 3, 3) / 24; g = torch.randn(2, 64, 33, 33)
def refl(d):
    xx = leaf(x, d); ww = leaf(w, d); y = F.conv2d(F.pad(xx, (1, 1, 1, 1), mode="reflect"), ww); y.backward(g.to(d)); return y, xx.grad, ww.grad
yc, gc, wc = refl("cpu"); yg, gg, wg = refl(dev); print(f"reflect conv: fwd {rel(yg,yc):.2e} dx {rel(gg,gc):.2e} dw {rel(wg,wc):.2e}")
