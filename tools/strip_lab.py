#!/usr/bin/env python
"""tools/strip_lab.py: resolve the measurement-build switches of mem_amd/csrc (a small `unifdef`).

Round 5 left ~60 `-D` switches (88 `#if` lines) inside the shipped kernels: timing experiments with wrong results, variants
that were measured and dropped, tuning constants.  This script fixes every such macro at its shipped value: conditionals on
them are resolved (the dead branch is deleted), `#ifndef X / #define X v / #endif` becomes a plain `#define X v` where the
value is still used in code, and macros that are never defined (the experiments) vanish with their branches.  Conditionals
on anything else (the stamp builds, include guards) are left alone.  The result compiles to the same object code
(checked by the caller with cmp on the .o files); the removed branches live on as tools/exp/r05_lab_switches.patch.

usage: strip_lab.py [--check] file..."""
import re
import sys

VALUES = {  # macro -> shipped value
    "ATTN16_FD_POS": 0, "ATTN16_SKEW": 1, "ATTN16_W3STAGE": 0, "ATTN16_EXP": 0,
    "WIN_Q_CKF_EARLY": 0, "WIN_DMA_LATE": 0, "WIN_PRIO": 0,
    "P8_EPI_AHEAD": 2, "P8_REALIGN": 1, "P8_RESID_NT": 1, "P8_EPI_RESID_LATE": 2, "P8_STAGE_MID": 0, "P8_BAR_MID": 0,
    "P8_PRIO_MODE": 0, "P8_SADDR": 1, "P8_RESID_ROWS_AHEAD": 2,
    "RASTER_NT": 0, "RASTER_EXP": 0,
}
UNDEFINED = {"WIN_EXP", "P8_EXP_NODMA", "P8_EXP_L2HOT", "P8_EXP_MFMA32", "P8_EXP_NOREAD", "P8_EXP_HALFN", "P8_RESID_AHEAD",
             "P8_FORCE_HALF", "TN_EXP_NOSTORE", "MEMHIP_EXP_NOGELU", "MEMHIP_EXP_PLAINSTORE", "MEMHIP_EXP_NOSTORE",
             "MEMHIP_EXP_NOLOAD", "ATTN16_TIMING"}
DROP_DEFINE = {"ATTN16_EXP", "RASTER_EXP"}     # used in #if only: the constant itself goes too
KNOWN = set(VALUES) | UNDEFINED
IDENT = re.compile(r"[A-Za-z_][A-Za-z_0-9]*")


def strip_comment(s):
    s = re.sub(r"//.*$", "", s)
    return re.sub(r"/\*.*?\*/", "", s).strip()


def evaluate(expr):
    """value of a preprocessor expression over KNOWN macros, or None when it mentions anything else"""
    e = strip_comment(expr)
    names = set(IDENT.findall(re.sub(r"defined\s*\(\s*\w+\s*\)|defined\s+\w+", "", e)))
    dnames = set(re.findall(r"defined\s*\(?\s*(\w+)", e))
    if not (names | dnames) <= KNOWN or not (names | dnames):
        return None
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)", lambda m: "1" if (m.group(1) or m.group(2)) in VALUES else "0", e)
    e = IDENT.sub(lambda m: str(VALUES.get(m.group(0), 0)), e)       # an undefined macro is 0 in #if
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return bool(eval(e, {"__builtins__": {}}))


def process(lines):
    out = []
    # frame: [kind, emitting_before, taken, live_now]; kind 'res' = resolved (directives dropped), 'keep' = left alone
    stack = []
    i = 0
    live = lambda: all(f[3] for f in stack)
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$", ln.rstrip("\n"))
        if not m:
            if live():
                d = re.match(r"\s*#\s*define\s+(\w+)\b", ln)
                if not (d and d.group(1) in DROP_DEFINE):
                    out.append(ln)
            i += 1
            continue
        kw, rest = m.group(1), m.group(2)
        if kw in ("ifdef", "ifndef", "if"):
            if kw == "if":
                v = evaluate(rest)
            else:
                name = strip_comment(rest).split()[0]
                if name in VALUES:
                    # `#ifndef X` guards the default definition of X: that branch is the shipped one
                    v = (kw == "ifndef")
                elif name in UNDEFINED:
                    v = (kw == "ifndef")
                else:
                    v = None
            if v is None:
                if live():
                    out.append(ln)
                stack.append(["keep", None, None, True])
            else:
                stack.append(["res", None, v, v])
        elif kw == "elif":
            f = stack[-1]
            if f[0] == "keep":
                if live():
                    out.append(ln)
            else:
                if f[2]:
                    f[3] = False
                else:
                    v = evaluate(rest)
                    assert v is not None, f"mixed known / unknown #elif: {ln}"
                    f[2] = f[3] = v
        elif kw == "else":
            f = stack[-1]
            if f[0] == "keep":
                if live():
                    out.append(ln)
            else:
                f[3] = not f[2]
                f[2] = True
        else:  # endif
            f = stack.pop()
            if f[0] == "keep" and live():
                out.append(ln)
        i += 1
    assert not stack
    return out


def main():
    check = "--check" in sys.argv
    rc = 0
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        src = open(path).readlines()
        dst = process(src)
        n0 = sum("#if" in l for l in src); n1 = sum("#if" in l for l in dst)
        print(f"{path}: {len(src)} -> {len(dst)} lines, #if {n0} -> {n1}")
        if check:
            rc |= int(src != dst)
        elif src != dst:
            open(path, "w").writelines(dst)
    sys.exit(rc)


if __name__ == "__main__":
    main()
