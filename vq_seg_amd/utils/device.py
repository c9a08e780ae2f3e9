import torch


def device_setting(device):
    """Reference: utils/device.py:4-10.  'cuda:N' is PyTorch-ROCm's native device spelling."""
    if device in ("-1", -1, "cpu"):
        return torch.device("cpu")
    if device == "cuda":
        return torch.device("cuda")
    return torch.device("cuda:" + str(device))
