import os
import random

import numpy as np
import torch


def seed_everything(seed: int = 42):
    """Reference: utils/seed.py:6-13 (seed 42 everywhere)."""
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
