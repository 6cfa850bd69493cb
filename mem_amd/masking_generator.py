"""Mask generators -- drop-in for /root/reference/mem/masking_generator.py.

Same constructor arguments, ``__call__() -> ndarray[h, w] int64``, ``get_shape()``
and ``__repr__`` as the reference (:18-42, :85-104).  The sampling itself runs
in libmemhip.so (host C++, CPython-`random`-exact MT19937; csrc/mask.cpp).

Stream semantics: by default the generators draw from Python's global
``random`` state exactly like the reference (state is pulled with
``random.getstate()``, advanced natively, pushed back), so
``random.seed(s); gen()`` reproduces the reference bit for bit even when other
``random`` users are interleaved.  ``seed=`` gives the generator a private
stream instead, and ``batch(n)`` produces n masks in one native call.
"""
import math
import random

import numpy as np

from ._lib import check, declare, f64, i32, lib, np_ptr, vp

declare({
    "memhip_mt_seed": (i32, [vp, vp, i32]),
    "memhip_mt_random": (f64, [vp]),
    "memhip_mask_blockwise": (i32, [vp, i32, i32, i32, i32, i32, f64, f64, i32, vp]),
    "memhip_mask_random_location": (i32, [vp, i32, i32, i32, i32, vp]),
})


class _Stream:
    """MT19937 state: 624 words + position (== random.getstate()[1])."""

    def __init__(self, seed=None):
        self.private = seed is not None
        self.state = np.zeros(625, dtype=np.uint32)
        if self.private:
            a = abs(int(seed))
            key = []
            while True:
                key.append(a & 0xFFFFFFFF)
                a >>= 32
                if not a:
                    break
            k = np.asarray(key, dtype=np.uint32)
            check(lib.memhip_mt_seed(np_ptr(self.state), np_ptr(k), len(key)), "mt_seed")

    def __enter__(self):
        if not self.private:
            ver, st, gauss = random.getstate()
            self._ver, self._gauss = ver, gauss
            self.state[:] = np.asarray(st, dtype=np.uint32)
        return np_ptr(self.state)

    def __exit__(self, *exc):
        if not self.private:
            random.setstate((self._ver, tuple(self.state.tolist()), self._gauss))
        return False

    def random(self):
        with self as p:
            return lib.memhip_mt_random(p)


class MaskingGenerator:
    def __init__(self, input_size, num_masking_patches, min_num_patches=4, max_num_patches=None,
                 min_aspect=0.3, max_aspect=None, seed=None):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.height, self.width = input_size
        self.num_patches = self.height * self.width
        self.num_masking_patches = num_masking_patches
        self.min_num_patches = min_num_patches
        self.max_num_patches = num_masking_patches if max_num_patches is None else max_num_patches
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))
        self._stream = _Stream(seed)

    def __repr__(self):
        return "Generator(%d, %d -> [%d ~ %d], max = %d, %.3f ~ %.3f)" % (
            self.height, self.width, self.min_num_patches, self.max_num_patches,
            self.num_masking_patches, self.log_aspect_ratio[0], self.log_aspect_ratio[1])

    def get_shape(self):
        return self.height, self.width

    def batch_u8(self, n):
        """n masks as uint8 [n, h, w] (one native call)."""
        out = np.empty((n, self.height, self.width), dtype=np.uint8)
        with self._stream as st:
            check(lib.memhip_mask_blockwise(st, self.height, self.width, self.num_masking_patches,
                                            self.min_num_patches, self.max_num_patches,
                                            self.log_aspect_ratio[0], self.log_aspect_ratio[1], n,
                                            np_ptr(out)), "mask_blockwise")
        return out

    def batch(self, n):
        return self.batch_u8(n).astype(np.int64)

    def __call__(self):
        return self.batch(1)[0]


class MaskingGeneratorRandomLocation:
    def __init__(self, input_size, num_masking_patches, seed=None):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.height, self.width = input_size
        self.num_patches = self.height * self.width
        self.num_masking_patches = num_masking_patches
        print(f"Masking Ration for RandomLocation-Masker is = {self.num_masking_patches/self.num_patches}")
        assert self.num_masking_patches < self.num_patches
        self._stream = _Stream(seed)

    def __repr__(self):
        return "Generator(patchesY: %d, patchesX %d, numMaskingPatches: %d" % (
            self.height, self.width, self.num_masking_patches)

    def get_shape(self):
        return self.height, self.width

    def batch_u8(self, n):
        out = np.empty((n, self.height, self.width), dtype=np.uint8)
        with self._stream as st:
            check(lib.memhip_mask_random_location(st, self.height, self.width, self.num_masking_patches,
                                                  n, np_ptr(out)), "mask_random_location")
        return out

    def batch(self, n):
        return self.batch_u8(n).astype(np.int64)

    def __call__(self):
        return self.batch(1)[0]
