"""The built library must not contain v_pk_{fma,mul,add}_f32 with the high register of a pair selected for src1
(op_sel:[x,1,..]): on gfx950 that form returns wrong values in lanes 48..63 while a wave of another kernel issues
MFMAs on the same SIMD -- which is what the two-stream backward pass arranges on purpose (activezero_amd/overlap.py).
Found through the 32->1 weight-gradient kernel; reproduced stand-alone by tools/probes/pkfma_corun.hip; numbers in
profiles/r03_pkfma_corun.md.  hipcc picks the form by itself when a broadcast scalar lands in an odd register as the
SECOND factor of a packed multiply, so the check runs on the ISA, not on the source."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import isa_lint  # noqa: E402


def test_no_packed_fp32_with_high_src1_selection():
    from activezero_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    if not os.path.exists(isa_lint.OBJDUMP):
        pytest.fail(f"{isa_lint.OBJDUMP} is missing although the library is built: the packed-fp32 lint cannot be skipped "
                    "(a kernel with the form silently corrupts gradients beside MFMA waves, profiles/r03_pkfma_corun.md)")
    bad = isa_lint.risky_packed_ops(_lib.LIB_PATH)
    assert not bad, "\n".join(f"{k}: {i}" for k, i in bad[:20])


def test_lint_recognises_the_form():
    assert isa_lint.OPSEL.search("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0]").group(1).split(",")[1] == "1"
    assert isa_lint.PACKED.search("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]")
    assert not isa_lint.PACKED.search("v_pk_fma_f16 v0, v1, v2, v3 op_sel:[0,1,0]")
