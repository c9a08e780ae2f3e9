"""Diagnostic: is tests/cases.py::codebook_from_labels machine independent?  Prints checksums of the pieces per vq_big case."""
import numpy as np
import torch

from tests import cases, golden_io, synth

fx = golden_io.load("vq_big")
print("numpy", np.__version__, "torch", torch.__version__, "threads", torch.get_num_threads())
for case in fx.meta["cases"]:
    name = case["name"]
    rows = cases.vq_big_rows(case)
    lab = fx[f"{name}/labels"].long()
    k = case["k"]
    W1 = cases.codebook_from_labels(rows, lab.numpy(), k)
    s2 = torch.zeros(k, rows.shape[1], dtype=torch.float64).index_add_(0, lab, rows.double())
    cnt = torch.bincount(lab, minlength=k).double()
    W2 = (s2 / cnt[:, None]).float()
    order = torch.argsort(lab, stable=True)
    W3 = torch.stack([rows[order[(lab[order] == j)]].double().sum(0) / cnt[j] for j in range(k)]).float() if k * rows.shape[0] < 2 ** 24 else W2
    print(name, "rows ok", synth.bits_checksum(rows) == case["rows_bits"], "| W add.at", synth.bits_checksum(W1) == case["w_bits"], "index_add", synth.bits_checksum(W2) == case["w_bits"],
          "loop", synth.bits_checksum(W3) == case["w_bits"], "| add.at == index_add", torch.equal(W1, W2), "max diff", (W1 - W2).abs().max().item(),
          "| sum", repr(synth.checksum(W1)),  "np sum", repr(float(W1.numpy().astype(np.float64).sum())))
