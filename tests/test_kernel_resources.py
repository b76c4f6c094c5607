"""No shipped kernel may spill registers or use scratch memory: compile the product sources for gfx950 with -save-temps
(hipcc cross-compiles without a GPU) and read .vgpr_spill_count / .private_segment_fixed_size of every kernel from the
code-object metadata.  The -save-temps compile has a directory of its own (build.py BUILD_TEMPS), outside
what git and gpurun ship."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402


def test_every_shipped_kernel_is_free_of_spills_and_scratch():
    from quantumattention_amd import build as hip_build

    d = hip_build.build(save_temps=True)   # no-op when the objects and their .s files are current
    assert d == hip_build.BUILD_TEMPS
    units = [name for _, _, name in hip_build.UNITS]
    files = [name + "-hip-amdgcn-amd-amdhsa-gfx950.s" for name in units]
    assert all(os.path.exists(os.path.join(d, f)) for f in files), "build(save_temps=True) must leave one .s per translation unit"
    total, bad = 0, []
    for f in sorted(files):
        if False:
            continue
        for r in kernel_resources.parse(os.path.join(d, f)):
            total += 1
            # (SGPR "spills" go to spare VGPR lanes with v_writelane / v_readlane, never to memory; the 106-SGPR attention kernels
            # have some around the rare rescue / redo paths -- none inside the hot loop, tools/kernel_resources.py --loops shows it)
            if r.get("vgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0):
                bad.append((r["name"], r.get("vgpr_count"), r.get("vgpr_spill_count"), r.get("private_segment_fixed_size")))
            assert r["vgpr_count"] <= 256, r   # two waves per SIMD for the 512-thread attention kernels
    assert total >= 100, total   # quant + pack + attention (fp8, 16-bit) instantiations
    assert not bad, bad


def test_no_inline_asm_reads_an_mfma_result_too_early():
    """The compiler pads the MFMA -> VALU wait states only in front of its own instructions; an asm statement that reads a fresh
    MFMA accumulator sees the old register value (tools/asm_hazards.py; found on the GPU as wrong row maxima at D = 64)."""
    import asm_hazards
    from quantumattention_amd import build as hip_build

    d = hip_build.build(save_temps=True)
    findings = []
    for _, _, name in hip_build.UNITS:
        findings += asm_hazards.check(os.path.join(d, name + "-hip-amdgcn-amd-amdhsa-gfx950.s"))
    assert not findings, findings[:5]


def test_the_hazard_checker_sees_a_planted_hazard(tmp_path):
    import asm_hazards

    src = """
_Z4kernv:
	v_mfma_f32_32x32x64_f8f6f4 v[34:49], v[74:81], v[34:41], 0
	ds_read_b128 v[82:85], v117 offset:12288
	s_nop 3
	;;#ASMSTART
	v_max3_f32 v90, v50, v51, v34
	;;#ASMEND
	s_nop 15
	;;#ASMSTART
	v_max3_f32 v91, v35, v36, v37
	;;#ASMEND
.Lfunc_end0:
"""
    p = tmp_path / "k.s"
    p.write_text(src)
    found = asm_hazards.check(str(p))
    assert len(found) == 1 and found[0][3] == 34, found
