"""Build check (no GPU): the kernels that read LDS through hand-written ``ds_read_b128`` asm with a separate, counted
``s_waitcnt lgkmcnt(N)`` are compiled to gfx950 assembly and walked by tools/check_lds_asm.py -- no instruction may touch the
destination registers of such a read before the wait that covers it (the compiler's own waitcnt insertion does not see inline asm;
round-4 advisor finding)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_lds_asm as C  # noqa: E402

GOOD = """
k:
\tds_read_b128 v[0:3], v9
\tds_read_b128 v[4:7], v9 offset:1024
\ts_load_dword s4, s[0:1], 0x0
\tv_add_u32_e32 v9, 1, v9
\ts_waitcnt lgkmcnt(1)
\tv_mov_b32_e32 v10, v0
\ts_waitcnt lgkmcnt(0)
\tv_mov_b32_e32 v11, v4
\ts_endpgm
"""
BAD = GOOD.replace("\ts_waitcnt lgkmcnt(1)\n", "")


def test_checker_sees_a_planted_hazard():
    assert C.check_text(GOOD)[2] == []
    k, reads, bad = C.check_text(BAD)
    assert reads == 2 and len(bad) == 1 and "v[0]" in bad[0]


@pytest.mark.parametrize("src", ["gemm_bf3p.hip", "embed.hip"])
def test_no_instruction_touches_an_outstanding_lds_read(src):
    if not os.path.exists(C.HIPCC):
        pytest.skip("hipcc not available")
    kernels, reads, bad = C.check_text(C.assembly(os.path.join(C.CSRC, src)))
    assert reads > 100, (src, reads)              # the hand-written reads are there
    assert not bad, bad[:5]
