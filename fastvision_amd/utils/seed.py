"""set_random_seeds (reference utils/seed.py:7-19): seed Python, numpy and torch (CPU + device generators)."""
import os
import random

import numpy as np
import torch

__all__ = ['set_random_seeds']


def set_random_seeds(seed):
    if seed is not None:
        os.environ['PYTHONHASHSEED'] = str(seed)
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)            # also seeds every device generator torch has created or will create
    print(f'RANDOM SEED : {seed}')
