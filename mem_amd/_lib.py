"""ctypes binding of libmemhip.so (include/memhip.h).

The HIP library is the product: there is NO fallback.  If the shared object is
missing or a symbol cannot be resolved, importing this module raises.
"""
import ctypes as C
import os

# torch first: its bundled HIP runtime (libamdhip64) must be the one already loaded when
# libmemhip.so resolves its dependency, so that streams / device pointers are shared.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MEMHIP_LIB points at an alternative build of the same library (kernel experiments)
LIB_PATH = os.environ.get("MEMHIP_LIB") or os.path.join(_HERE, "libmemhip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C mem_amd/csrc` (hipcc --offload-arch=gfx950) "
        "or `python -c 'import __graft_entry__ as g; g.build()'`. mem_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH)

ABI_VERSION = 5

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t


class MemhipError(RuntimeError):
    pass


def declare(sigs):
    """name -> (restype, argtypes).  A missing symbol raises AttributeError: loud by design."""
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


declare({
    "memhip_abi_version": (i32, []),
    "memhip_last_error": (C.c_char_p, []),
    "memhip_arch": (C.c_char_p, []),
    "memhip_build_flags": (C.c_char_p, []),
    "memhip_set_option": (i32, [C.c_char_p, i32]),
    "memhip_get_option": (i32, [C.c_char_p, C.POINTER(i32)]),
})

if lib.memhip_abi_version() != ABI_VERSION:
    raise ImportError(f"libmemhip.so ABI {lib.memhip_abi_version()} != expected {ABI_VERSION}; rebuild")


BUILD_FLAGS = lib.memhip_build_flags().decode()      # "" = the shipped build
IS_SHIPPED_LIB = os.environ.get("MEMHIP_LIB") in (None, "") and BUILD_FLAGS == ""


def check(rc, what=""):
    if rc != 0:
        raise MemhipError(f"{what} failed ({rc}): {lib.memhip_last_error().decode()}")


def set_option(name, value):
    """Kernel-selection switch of the library (include/memhip.h: memhip_set_option) -- A/B tools only."""
    check(lib.memhip_set_option(name.encode(), int(value)), "set_option")


def get_option(name):
    v = i32(0)
    check(lib.memhip_get_option(name.encode(), C.byref(v)), "get_option")
    return v.value


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise MemhipError("mem_amd needs an MI355X (gfx950) device: torch.cuda.is_available() is False "
                          "and there is no CPU fallback")
