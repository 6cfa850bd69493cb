"""mem_amd -- MI355X-native (gfx950) implementation of the MEM pretraining hot path.

Mirrors the Python surface of tum-vision/mem's pretraining path
(run_mem_pretraining / modeling_pretrain / engine_for_pretraining / datasets /
transforms / masking_generator) on top of hand-written HIP kernels in
``libmemhip.so`` (C ABI: include/memhip.h).  No CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)

__version__ = "0.1.0"
