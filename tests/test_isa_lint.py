"""The built library must not contain the packed-fp32 form that gfx950 miscomputes while f16/bf16 MFMAs are in flight on the SIMD - in ANY
kernel, since the MFMA can belong to another kernel's wave (tools/isa_hazard_lint.py, profiles/r05_pk_opsel_hazard.md,
profiles/r06_lanes_48_63.md).  Needs the built .so and the ROCm binary utilities, no GPU."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import isa_hazard_lint  # noqa: E402

LIB = os.path.join(REPO, "diffab-pytorch_amd", "lib", "libdiffab_hip.so")


def test_lint_recognises_the_hazard_form():
    text = """
0000000000001000 <kernel_a>:
\tv_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3] // 0
\tv_pk_mul_f32 v[92:93], v[110:111], v[92:93] op_sel:[0,1]   // 1
\tv_pk_mul_f32 v[92:93], v[110:111], v[92:93] op_sel_hi:[1,0] // 2
\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,0]  // 3
0000000000002000 <kernel_b>:
\tv_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]  // fp32 MFMAs do not trigger it
\tv_pk_mul_f32 v[92:93], v[110:111], v[92:93] op_sel:[0,1]
"""
    found = isa_hazard_lint.lint_text(text)
    # kernel_b has no f16/bf16 MFMA of its own (dist -1): still a finding, another kernel's wave on the same SIMD can supply the MFMA
    assert [(k, d) for k, _, d in found] == [("kernel_a", 1), ("kernel_a", 3), ("kernel_b", -1)]


@pytest.mark.skipif(not os.path.exists(LIB) or not os.path.exists(os.path.join(isa_hazard_lint.LLVM, "llvm-objdump")),
                    reason="needs the built library and the ROCm llvm binary utilities")
def test_built_library_has_no_op_sel_01_packed_ops_in_any_kernel():
    texts = list(isa_hazard_lint.code_objects(LIB))
    assert texts, "no gfx950 code object found in the library"
    assert sum(t.count("v_mfma") for t in texts) > 1000  # the disassembly is the real one
    found = isa_hazard_lint.lint_paths([LIB])
    assert not found, found


def test_no_inline_assembly_vector_op_outside_the_reviewed_list():
    """The compiler guards neither waits nor matrix-core hazards it cannot see: it does not insert the MFMA -> VALU wait states for the
    operands of an inline-assembly statement (round 6: the projection tile's `asm("v_mul_f32 ...")` read an accumulator three
    instructions behind its MFMA in one instantiation and stored a stale register; proj_frames_h3_tile.h mul1).  Every assembly
    statement that names a vector instruction is therefore listed here with the reason its operands can never be MFMA results."""
    import glob
    import re

    reviewed = {
        # packed subtract of point coordinates: both operands come from LDS / global loads (phase 1 of the attention tile)
        ("ipa_attn_tile.h", "v_pk_add_f32"),
        ("attention_split.hip", "v_pk_add_f32"),
    }
    seen = set()
    for path in glob.glob(os.path.join(REPO, "diffab-pytorch_amd", "csrc", "*")):
        if not path.endswith((".h", ".hip")):
            continue
        for m in re.finditer(r'asm\s*(?:volatile)?\s*\(\s*"\s*(v_\w+|ds_\w+|global_\w+|buffer_\w+)', open(path).read()):
            seen.add((os.path.basename(path), m.group(1)))
    assert seen <= reviewed, sorted(seen - reviewed)
