"""The device side of libcurdle_g1.so is built in explicit stages so that one LLVM pass (Reassociate) can be left out
(curdleproofs_pie_amd/build.py).  No GPU needed: the checks read the code object inside the .so."""
import os
import re
import subprocess

import pytest

from curdleproofs_pie_amd import build as B

LLVM = B.LLVM_BIN


def _disassemble(lib, symbol_re, tmp_path):
    fat = tmp_path / "fat.bin"
    co = tmp_path / "dev.hsaco"
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, str(tmp_path / "discard.so")])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
    syms = subprocess.run([f"{LLVM}/llvm-objdump", "-t", str(co)], check=True, capture_output=True, text=True).stdout
    names = [ln.split()[-1] for ln in syms.splitlines() if re.search(symbol_re, ln) and " F .text" in ln]
    assert len(names) == 1, names
    out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={names[0]}", str(co)], check=True, capture_output=True, text=True).stdout
    ops = {}
    for ln in out.splitlines():
        m = re.match(r"\s+([sv]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+)", ln)
        if m:
            ops[m.group(1)] = ops.get(m.group(1), 0) + 1
    return ops


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-objdump"), reason="ROCm LLVM tools not installed")
def test_staged_build_drops_the_reassociate_adds(tmp_path):
    B.build(verbose=False)
    info = B.build_info()
    assert info.get("pipeline") == "staged", info
    assert info.get("dropped_passes") == ["reassociate"]
    ops = _disassemble(B.LIB, r"k_accumulate", tmp_path)
    mads = ops.get("v_mad_u64_u32", 0)
    assert mads > 6000                                   # mixed addition + affine pair addition + the cold doubling paths
    # Reassociate costs one v_lshl_add_u64 per Montgomery column (449 in this kernel with plain hipcc)
    assert ops.get("v_lshl_add_u64", 0) <= 16, ops.get("v_lshl_add_u64")      # address arithmetic only
    valu = sum(n for op, n in ops.items() if op.startswith("v_"))
    assert valu / mads < 1.35, (valu, mads)              # 1.33 staged, 1.39 plain


def test_pipeline_choice_is_part_of_the_source_hash(monkeypatch):
    a = B._source_hash()
    monkeypatch.setenv("CURDLE_G1_PIPELINE", "plain")
    assert B._source_hash() != a
    assert B.default_pipeline() == "plain"
    monkeypatch.setenv("CURDLE_G1_PIPELINE", "bogus")
    with pytest.raises(ValueError):
        B.default_pipeline()
