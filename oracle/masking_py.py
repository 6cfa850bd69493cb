"""Pure-Python restatement of the reference mask generators.  TEST INFRASTRUCTURE.

Follows /root/reference/mem/masking_generator.py:18-81 (block-wise, BEiT style)
and :85-116 (random location).  Randomness comes from CPython's ``random``
module exactly like the reference, so "the k-th mask after random.seed(s)" is
the parity unit.  Pinned by oracle/gen_golden.py -> tests/golden/masks_*.npz.
"""
import math
import random

import numpy as np


class BlockMaskOracle:
    def __init__(self, input_size, num_masking_patches, min_num_patches=4, max_num_patches=None,
                 min_aspect=0.3, max_aspect=None):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.H, self.W = input_size
        self.total = num_masking_patches
        self.lo = min_num_patches
        self.hi = num_masking_patches if max_num_patches is None else max_num_patches
        max_aspect = max_aspect or 1 / min_aspect
        self.log_ar = (math.log(min_aspect), math.log(max_aspect))

    def _try_block(self, mask, budget):
        """masking_generator.py:44-66: up to 10 rejection-sampled rectangles;
        every attempt consumes draws even when rejected."""
        for _ in range(10):
            area = random.uniform(self.lo, budget)
            ar = math.exp(random.uniform(*self.log_ar))
            h = int(round(math.sqrt(area * ar)))
            w = int(round(math.sqrt(area / ar)))
            if w < self.W and h < self.H:
                top = random.randint(0, self.H - h)
                left = random.randint(0, self.W - w)
                window = mask[top:top + h, left:left + w]
                fresh = h * w - int(window.sum())
                if 0 < fresh <= budget:
                    window[...] = 1
                    return fresh
        return 0

    def __call__(self):
        """masking_generator.py:68-81."""
        mask = np.zeros((self.H, self.W), dtype=np.int64)
        done = 0
        while done < self.total:
            budget = min(self.total - done, self.hi)
            got = self._try_block(mask, budget)
            if got == 0:
                break
            done += got
        return mask


class RandomLocationMaskOracle:
    """masking_generator.py:85-116: random.sample over arange(H*W-1) -- the last
    patch can never be masked (reference off-by-one, kept)."""

    def __init__(self, input_size, num_masking_patches):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.H, self.W = input_size
        self.k = num_masking_patches
        assert self.k < self.H * self.W

    def __call__(self):
        mask = np.zeros(self.H * self.W, dtype=np.int64)
        pool = list(range(self.H * self.W - 1))
        mask[random.sample(pool, self.k)] = 1
        return mask.reshape(self.H, self.W)
