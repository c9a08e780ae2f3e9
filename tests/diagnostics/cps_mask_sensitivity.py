"""How many pseudo-label mask pixels of iteration 1 move when the CPU oracle's inputs of iteration 0 are perturbed by 1e-6?
(Adam's first step is lr * sign(g): a gradient within rounding error of zero flips a weight by 2 * lr, and the entropy percentile
of iteration 1 is an order statistic -- so iteration-1 masks are NOT reproducible to the pixel across devices; this prints the
CPU-vs-CPU figure that tests/test_compat_gpu.py's slack for iteration 1 is set from.)   python -m tests.diagnostics.cps_mask_sensitivity"""
import torch

from oracle.cps_ref import CPSReference
from tests import cps_loop
from tests.test_oracle_golden import _prepared_oracle_params


def run(eps):
    data = cps_loop.batches(2)
    cfg = cps_loop.model_cfg(1)["params"]
    sds = [_prepared_oracle_params(1, seed, data[0][0], data[0][1], cfg["margin"], cfg["scale"]) for seed in cps_loop.SEEDS]
    ref = CPSReference(sds, version=1)
    outs = []
    for i, (l_in, l_tg, ul_in) in enumerate(data):
        if i == 0 and eps:
            l_in = l_in + eps * torch.sign(torch.sin(torch.arange(l_in.numel()).reshape(l_in.shape) * 1.7))
        outs.append(ref.step(l_in, l_tg, ul_in))
    return outs


if __name__ == "__main__":
    torch.set_num_threads(8)
    base = run(0.0)
    for eps in (1e-7, 1e-6):
        pert = run(eps)
        for i in range(2):
            d1 = int((base[i]["mask_1"] != pert[i]["mask_1"]).sum())
            d2 = int((base[i]["mask_2"] != pert[i]["mask_2"]).sum())
            print(f"eps {eps:g} it{i}: mask_1 {d1} mask_2 {d2} of {base[i]['mask_1'].numel()} pixels; loss {base[i]['loss']:.7f} vs {pert[i]['loss']:.7f}")
