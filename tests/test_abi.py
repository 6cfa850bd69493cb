"""-m "not gpu": the C-ABI library loads and exports every symbol include/memhip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "memhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(memhip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "mem_amd", "libmemhip.so"))
    names = declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_arch():
    from mem_amd import _lib
    assert _lib.lib.memhip_abi_version() == _lib.ABI_VERSION
    assert _lib.lib.memhip_arch() == b"gfx950"


def test_error_reporting_without_gpu():
    """Argument validation happens before any launch: usable (and loud) on a CPU box."""
    from mem_amd import _lib
    from mem_amd import datasets  # noqa: F401  (declares the signatures)
    rc = _lib.lib.memhip_rasterize_f64(None, None, 1, 0, 0, 0, None, None, None, 0, None)
    assert rc == -1
    assert b"bad shape" in _lib.lib.memhip_last_error()


def test_no_oracle_in_product():
    """The product package must never import the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "mem_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_options_table_and_no_environment_reads():
    """Kernel-selection switches exist only behind memhip_set_option; the shared object imports no getenv."""
    import subprocess
    from mem_amd import _lib
    assert _lib.get_option("gemm_p8") == 1 and _lib.get_option("gemm_p8_min_n") == 512
    _lib.set_option("gemm_p8", 0)
    assert _lib.get_option("gemm_p8") == 0
    _lib.set_option("gemm_p8", 1)
    try:
        _lib.set_option("no_such_option", 1)
        raise AssertionError("unknown option accepted")
    except _lib.MemhipError as e:
        assert "unknown option" in str(e)
    syms = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(ROOT, "mem_amd", "libmemhip.so")],
                          capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_attn16_counted_waits_match_the_instruction_stream(tmp_path):
    """attn16.hip waits for the LDS-DMA of the next sample with a COUNTED s_waitcnt vmcnt(N): the N youngest vector-memory
    operations (stores of the previous sample) may stay in flight.  That is only sound while a wave issues at least N stores
    per sample behind the LDS-DMA (in execution order: DMA at the top of the iteration, stores at its end) and nothing else
    (a register spill would add vector-memory operations the count does not know) -- checked here on the compiler's output."""
    import re, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "mem_amd", "csrc", "attn16.hip")
    out = str(tmp_path / "attn16.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-S", "--cuda-device-only",
                    "-o", out, src], check=True, capture_output=True)
    lines = open(out).read().split("\n")
    checked = 0
    for kname, n_wait in (("attn16_fwd_kernel", 4), ("attn16_bwd_kernelILb1ELb0", 10), ("attn16_bwd_kernelILb0ELb0", 10),
                          ("attn16_bwd_kernelILb1ELb1", 10), ("attn16_bwd_kernelILb0ELb1", 10)):
        i0 = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and kname in l and l.rstrip().endswith(tuple(": ;"))
                  or (l.startswith("_ZN") and kname in l and ": ;" in l))
        j = i0
        while "s_endpgm" not in lines[j]:
            j += 1
        body = lines[i0:j]
        glds = [k for k, l in enumerate(body) if "global_load_lds" in l]
        stores = [k for k, l in enumerate(body) if re.search(r"\bglobal_store_dwordx4\b", l)]
        waits = [l for l in body if re.search(r"s_waitcnt vmcnt\(%d\)" % n_wait, l)]
        assert glds and waits, (kname, len(glds), len(waits))
        assert len(stores) >= n_wait, (kname, len(stores), n_wait)    # all of them sit in the sample loop, behind its DMA
        assert not any("scratch_" in l for l in body), kname        # a spill would add uncounted vector-memory operations
        if kname.endswith("ELb1"):
            # fused delta: the forward-output rows of the next sample are plain (asm) loads issued in front of the sample's stores; at
            # least n_wait stores must follow the LAST of them in the sample loop, so that vmcnt(n_wait) covers them
            loads = [k for k, l in enumerate(body) if re.search(r"\bglobal_load_dwordx4\b", l)]
            assert len(loads) >= 8, (kname, len(loads))           # 4 in the prologue (first sample) + 4 in the loop
            assert sum(1 for k in stores if k > loads[-1]) >= n_wait, (kname, loads[-1], stores)
        if kname == "attn16_fwd_kernel":
            # round 4: the staging wave waits with vmcnt(28) = "all but the newest IMAGE": an image must be exactly 28 pieces
            # issued back to back through the scalar-base form (nothing else of that wave in between), two images per sample
            assert any(re.search(r"s_waitcnt vmcnt\(28\)", l) for l in body), kname
            sb = [k for k, l in enumerate(body) if re.search(r"global_load_lds_dwordx4 v\d+, s\[", l)]
            assert len(sb) == 56, (kname, len(sb))
            for img in (sb[:28], sb[28:]):
                between = body[img[0]:img[-1] + 1]
                assert not any(re.search(r"\b(global_load|global_store|buffer_|scratch_)", l) and "global_load_lds_dwordx4" not in l
                               for l in between), kname
        checked += 1
    assert checked == 5


def test_gemm_p8_residual_epilogue_counted_waits(tmp_path):
    """The residual epilogue of the 256-row gemm_p8 kernel loads its rows with inline asm and waits with HAND-COUNTED
    s_waitcnt vmcnt(N): N = the vector-memory operations issued behind a row's three loads (later rows' loads, earlier rows'
    stores).  If a toolchain change made hipcc emit FEWER stores per row than the count assumes, a row would be consumed before
    it has landed -- silently wrong.  Replay the compiler's instruction stream: at the k-th counted wait of the epilogue, the
    three loads of row k must be older than the N youngest vector-memory operations issued so far."""
    import re, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "mem_amd", "csrc", "gemm_p8.hip")
    out = str(tmp_path / "gemm_p8.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-S", "--cuda-device-only",
                    "-o", out, src], check=True, capture_output=True)
    lines = open(out).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and "gemm_p8_kernelILi2ELi256ELb0ELb0E" in l and ":" in l)
    j = i0
    while "s_endpgm" not in lines[j]:
        j += 1
    body = lines[i0:j]
    assert not any("scratch_" in l for l in body)               # a spill would add uncounted vector-memory operations
    rm = [k for k, l in enumerate(body) if re.search(r"\bglobal_load_dword v\d+, v\[", l)]      # the row-mask load closes a row's 3 loads
    assert len(rm) == 16, len(rm)
    # the epilogue's operation stream from the first row's loads on: L = asm row loads, S = stores, D = LDS-DMA, W = counted waits
    start = min(k for k in range(rm[0] - 4, rm[0] + 1) if re.search(r"\bglobal_load_dwordx4\b", body[k]) or k == rm[0])
    ops, row_last_load, waits = 0, [], []
    for k in range(start, len(body)):
        l = body[k]
        if re.search(r"\bglobal_load_dword(x4)? v", l) and "lds" not in l:
            ops += 1
            if k in rm:
                row_last_load.append(ops)                      # ordinal of the last load of this row
        elif re.search(r"\bglobal_store_dword", l) or "global_load_lds" in l or re.search(r"\bbuffer_(load|store)", l):
            ops += 1
        else:
            m = re.search(r"s_waitcnt vmcnt\((\d+)\)", l)
            if m and len(waits) < 16 and len(row_last_load) > len(waits):
                waits.append((ops, int(m.group(1))))
        if len(waits) == 16:
            break
    assert len(waits) == 16 and len(row_last_load) == 16, (len(waits), len(row_last_load))
    for r, (issued, n) in enumerate(waits):
        assert row_last_load[r] <= issued - n, (r, row_last_load[r], issued, n)     # row r's loads are not among the n youngest
        assert n <= 10


def test_asm_lds_dma_owns_its_m0(tmp_path):
    """The kernels issue LDS-DMA (`global_load_lds_dwordx4`) from inline asm that writes the destination into M0 itself
    (`s_mov_b32 m0, <sgpr>; s_nop 0; global_load_lds ...`, "m0" on the clobber list) next to compiler-issued LDS-DMA through the
    builtin, for which hipcc manages M0 on its own.  The two must never be interleaved: on the compiler's output, every LDS-DMA
    instruction must be preceded -- within a few instructions and with no other M0 write, no other LDS-DMA and no branch target
    in between -- by exactly one write of M0."""
    import re, shutil, subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    files = ["gemm_p8.hip", "gemm_tn_p8.hip", "attn16.hip", "attn.hip", "attn_stream.hip", "attn_win.hip"]

    def build(f):
        out = str(tmp_path / (f + ".s"))
        subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-w", "-S",
                        "--cuda-device-only", "-o", out, os.path.join(ROOT, "mem_amd", "csrc", f)], check=True, capture_output=True)
        return out
    with ThreadPoolExecutor(max_workers=6) as ex:
        outs = list(ex.map(build, files))
    total = 0
    for f, out in zip(files, outs):
        lines = [l for l in open(out).read().split("\n")]
        code = [(i, l.strip()) for i, l in enumerate(lines) if l.startswith("\t") and not l.strip().startswith((";", "."))
                or re.match(r"^\.LBB\d+_\d+:", l)]
        for k, (i, l) in enumerate(code):
            if not re.match(r"(global_load_lds_|buffer_load_.*\blds\b)", l):
                continue
            total += 1
            # walk back to the M0 write that feeds this instruction
            j, found = k - 1, None
            while j >= 0 and k - j <= 8:
                prev = code[j][1]
                assert not re.match(r"^\.LBB", prev), (f, i, "a branch target between the m0 write and its LDS-DMA")
                assert not re.match(r"(global_load_lds_|buffer_load_.*\blds\b)", prev), (f, i, "two LDS-DMA behind one m0 write")
                if re.match(r"s_\w+\s+m0\b", prev):
                    found = j
                    break
                j -= 1
            assert found is not None, (f, i, l, "no m0 write within 8 instructions in front of the LDS-DMA")
    assert total >= 100, total


def test_shipped_kernels_carry_no_measurement_switches():
    """Round 6: the laboratory stays out of the product.  tools/strip_lab.py (the small `unifdef` that resolved the
    ~60 measurement switches of round 5 at their shipped values) finds nothing left to resolve in mem_amd/csrc, and the sources
    hold at most 20 `#if` lines (the stamp builds of gemm_p8 / gemm_tn_p8 / attn_win)."""
    import glob
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "mem_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "mem_amd", "csrc", "*.hpp")))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "strip_lab.py"), "--check"] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    n_if = sum(sum("#if" in line for line in open(f)) for f in files)
    assert n_if <= 20, n_if
    assert not os.path.exists(os.path.join(root, "mem_amd", "exp")) and not glob.glob(os.path.join(root, "mem_amd", "csrc", "_build_*"))
