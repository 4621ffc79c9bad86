"""Build-time guard of the MFMA-result hazard (csrc/split_mfma.h `mfma_results_fence`): every non-MFMA
read / overwrite of an MFMA destination register in the built gfx950 code objects must sit >= 19 wait states behind the
MFMA (tools/isa_lint.py explains the number).  CPU only: hipcc cross-compiles, llvm-objdump disassembles."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import isa_lint  # noqa: E402


def test_scanner_counts_wait_states():
    ok = ["kern:", "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]", "s_nop 15", "s_nop 2", "v_mul_f32_e32 v40, v0, v41"]
    assert isa_lint.lint_listing(ok) == []
    bad = ["kern:", "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]", "s_nop 10", "v_pk_mul_f32 v[32:33], v[0:1], v[56:57]"]
    v = isa_lint.lint_listing(bad)
    assert len(v) == 1 and v[0][4] == 11
    # a dependent MFMA (srcC) is interlocked; an MFMA result used as the A / B operand of the next MFMA is not
    chain = ["kern:", "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]",
             "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[28:31], v[0:15]"]
    assert isa_lint.lint_listing(chain) == []
    feed = ["kern:", "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]",
            "v_mfma_f32_32x32x16_bf16 v[32:47], v[0:3], v[28:31], v[32:47]"]
    assert len(isa_lint.lint_listing(feed)) == 1
    # overwriting a destination (WAW) counts; an unconditional branch ends the straight-line window
    waw = ["kern:", "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]", "v_mov_b32_e32 v3, 0"]
    assert len(isa_lint.lint_listing(waw)) == 1
    br = ["kern:", "v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]", "s_branch 20", "v_mov_b32_e32 v3, 0"]
    assert isa_lint.lint_listing(br) == []


def test_built_code_objects_keep_the_distance():
    csrc = os.path.join(REPO, "hcatgnet_amd", "csrc")
    if not os.path.isfile(isa_lint.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    subprocess.run(["make", "-C", csrc, "-j", "3"] + list(isa_lint.MFMA_FILES), check=True, stdout=subprocess.DEVNULL)
    rep = isa_lint.lint_objects([os.path.join(csrc, f) for f in isa_lint.MFMA_FILES])
    for path, viol in rep.items():
        assert viol == [], (path, viol[:3])


def test_flat_rule_on_a_listing():
    lines = ["kern:", "global_load_dword v1, v[2:3], off", "flat_load_dword v4, v[5:6]", "ds_read_b32 v7, v8"]
    hits = isa_lint.lint_flat(lines)
    assert len(hits) == 1 and hits[0][0] == "kern" and hits[0][2].startswith("flat_load_dword")


def test_built_code_objects_have_no_flat_memory_instructions():
    """Every pointer is a global kernel argument or carved from LDS: a flat access = the compiler lost the address space
    (and a flat LDS read's s_waitcnt vmcnt(0) serialises the prefetch pipelines; tools/isa_lint.py, second rule)."""
    csrc = os.path.join(REPO, "hcatgnet_amd", "csrc")
    if not os.path.isfile(isa_lint.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    subprocess.run(["make", "-C", csrc, "-j", "3"], check=True, stdout=subprocess.DEVNULL)
    rep = isa_lint.lint_objects_flat(isa_lint.all_kernel_objects())
    for path, hits in rep.items():
        assert hits == [], (path, hits[:3])
