#!/usr/bin/env python
"""Build check for the kernels that receive LDS data from hand-written ``ds_read_b128`` asm with a SEPARATE, counted
``s_waitcnt lgkmcnt(N)`` (csrc/gemm_bf3p.hip: gemm_bf3p_pipe_kernel, gemm_bf3q_pipe_kernel, gemm_bf3s_kernel; csrc/embed.hip).

The compiler does not know that the destination registers of such a read are undefined until the wait: its own waitcnt insertion
ignores inline asm, so nothing stops it from copying (``v_mov``), spilling or otherwise touching a destination register between the
read and the wait that covers it (round-4 advisor finding: "a hazard rather than an observed failure").  This script makes it an
observed non-failure: it compiles the file for gfx950 to assembly (device only, ~seconds) and walks every kernel, tracking the
outstanding LDS reads in issue order (LDS returns in order; ``lgkmcnt(N)`` = all but the N youngest LGKM operations have returned).
It fails if any instruction reads or writes a VGPR of a read that is still outstanding.

    python tools/check_lds_asm.py [file.hip ...]        (default: gemm_bf3p.hip embed.hip)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bbdm_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _vregs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def assembly(path):
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", path, "-o", "-"],
                       capture_output=True, text=True, cwd=os.path.dirname(path))
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    return r.stdout


def check_text(asm):
    """-> (kernels checked, LDS reads seen, [violations])."""
    kernels, reads, bad = 0, 0, []
    name, pending = None, []                    # pending: [(dest regs, line no)] of the outstanding LDS operations, oldest first
    for no, raw in enumerate(asm.splitlines(), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            name, pending = line[:-1], []
            kernels += 1
            continue
        if name is None or line.startswith("."):
            continue
        op = line.split()[0]
        if op == "s_endpgm":
            name = None
            continue
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            continue        # scalar loads share lgkmcnt and return out of order, but an outstanding one only makes a counted wait MORE
                            # conservative for the LDS reads: of the (outstanding - n) operations lgkmcnt(n) guarantees, at most the
                            # scalar ones are not LDS, and they were counted as outstanding too -- the LDS-only formula below is the worst case
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))              # in-order LDS returns: everything but the n youngest has landed
                pending = pending[len(pending) - n:] if n < len(pending) else ([] if n == 0 else pending)
            continue
        regs = _vregs(line.split(None, 1)[1]) if " " in line else set()
        for dest, at in pending:
            hit = regs & dest
            if hit:
                bad.append(f"{name}: line {no}: `{line}` touches v{sorted(hit)} of the ds_read at line {at} before its s_waitcnt")
        if op.startswith("ds_read") or op.startswith("ds_load"):
            operands = line.split(None, 1)[1]
            pending.append((_vregs(operands.split(",")[0]), no))
            reads += 1
        elif op.startswith("ds_") or op.startswith("s_sendmsg"):
            pending.append((set(), no))          # other LGKM operations occupy a counter slot too
    return kernels, reads, bad


def main(argv):
    files = argv or ["gemm_bf3p.hip", "embed.hip"]
    rc = 0
    for f in files:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        kernels, reads, bad = check_text(assembly(path))
        print(f"{os.path.basename(path)}: {kernels} functions, {reads} LDS reads, {len(bad)} violations")
        for b in bad[:20]:
            print("  " + b)
        rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
