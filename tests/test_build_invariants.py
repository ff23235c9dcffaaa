"""Build-time invariants of libquanto_hip.so that no run-time test would notice except by luck.

``profiles/r05_packed_fp32_op_sel_next_to_mfma.md``: on gfx950 a ``v_pk_{add,mul}_f32`` whose ``op_sel`` takes the HIGH half of a 64-bit source
returns a wrong value in lanes 48-63 once in 10^3..10^7 executions while MFMAs are in flight on the SIMD (stand-alone reproducer:
``scripts/probes/pk_f32_opsel_probe.hip``).  hipcc's SLP vectorizer produces exactly that form from adjacent scalar fp32 math, so the kernels that do
fp32 conversion math between MFMAs are compiled with ``-fno-slp-vectorize`` (csrc/Makefile) and written on scalars.  This test fails when a
toolchain or a source edit brings packed fp32 instructions with ``op_sel`` back into those translation units."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optimum_quanto_amd", "csrc")
HAVE_HIPCC = shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not HAVE_HIPCC, reason="needs hipcc")
@pytest.mark.parametrize("unit", ["qbits_mfma_large", "qconv_mfma", "qmm_mfma"])
def test_no_packed_fp32_with_op_sel_next_to_mfmas(unit):
    listing = os.path.join(CSRC, "build", unit + ".s")
    proc = subprocess.run(["make", "-C", CSRC, f"build/{unit}.s"], capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    text = open(listing).read()
    assert "v_mfma_f32_16x16x32" in text  # the listing is the device code of an MFMA kernel
    packed = [ln.strip() for ln in text.splitlines() if re.search(r"\bv_pk_(add|mul|fma)_f32\b", ln)]
    risky = [ln for ln in packed if "op_sel" in ln]
    assert not risky, f"{unit}: {len(risky)} packed fp32 instructions with op_sel, e.g. {risky[:3]}"
    assert not packed, f"{unit}: hipcc re-packed scalar fp32 math ({len(packed)} v_pk_*_f32): is -fno-slp-vectorize still applied?"


MFMA_UNITS = ["qmm_mfma_large", "qbits_skinny", "qbytes_skinny", "qmm_native8", "qbits_mmv", "qbits_mfma_fused", "qbits_mfma_large", "qconv_mfma", "qmm_mfma"]


@pytest.mark.skipif(not HAVE_HIPCC, reason="needs hipcc")
def test_no_mfma_kernel_takes_the_high_half_of_src1_into_a_low_result():
    """Every translation unit that issues MFMAs: packed fp32 is allowed (the split-K folds and epilogues use it on aligned pairs and with
    ``op_sel_hi:[1,0]`` - the low half of src1 broadcast to both results, a form that never failed in 6.7 x 10^11 probe results), the failing
    form - ``op_sel:[x,1]``: the LOW result reads the HIGH half of src1 - is not."""
    proc = subprocess.run(["make", "-C", CSRC, f"-j{os.cpu_count() or 4}"] + [f"build/{u}.s" for u in MFMA_UNITS], capture_output=True, text=True, timeout=1800)
    assert proc.returncode == 0, proc.stderr[-2000:]
    for unit in MFMA_UNITS:
        text = open(os.path.join(CSRC, "build", unit + ".s")).read()
        assert "v_mfma" in text, unit
        bad = [ln.strip() for ln in text.splitlines() if re.search(r"\bv_pk_(add|mul|fma)_f32\b", ln) and re.search(r"op_sel:\[[01],1", ln)]
        assert not bad, f"{unit}: {len(bad)} packed fp32 instructions whose low result reads the high half of src1, e.g. {bad[:3]}"


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs llvm-objdump")
def test_the_built_library_itself_carries_no_high_half_src1_packed_fp32():
    """The two tests above rebuild the listings with the Makefile's flags; this one disassembles the code objects of the library that SHIPS
    (optimum_quanto_amd/lib/libquanto_hip.so).  r6: a probe build with CXXFLAGS given on the make command line dropped the per-file
    -fno-slp-vectorize, left six objects of that build behind and produced a library whose int4 large-tile kernel returned wrong lanes on the
    GPU while both listing tests stayed green."""
    import importlib.util
    import tempfile

    lib = os.path.join(ROOT, "optimum_quanto_amd", "lib", "libquanto_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    spec = importlib.util.spec_from_file_location("so_kernel_report", os.path.join(ROOT, "scripts", "so_kernel_report.py"))
    rep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rep)
    bad, mfma_objects = [], 0
    for blob in rep.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            text = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, timeout=900).stdout
        if "v_mfma" not in text:
            continue
        mfma_objects += 1
        bad += [ln.strip() for ln in text.splitlines() if re.search(r"\bv_pk_(add|mul|fma)_f32\b", ln) and re.search(r"op_sel:\[[01],1", ln)]
    assert mfma_objects >= 8, mfma_objects
    assert not bad, f"{len(bad)} packed fp32 instructions whose low result reads the high half of src1 in the shipped library, e.g. {bad[:3]}: rebuild with `make clean && make`"


def _regs_in(line):
    """VGPR numbers a line of gfx950 assembly mentions (single registers and v[a:b] ranges)."""
    regs = set(int(m) for m in re.findall(r"\bv(\d+)\b", line))
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", line):
        regs.update(range(int(a), int(b) + 1))
    return regs


@pytest.mark.skipif(not HAVE_HIPCC, reason="needs hipcc")
@pytest.mark.parametrize("unit", ["qmm_mfma_large", "qmm_native8", "qbytes_skinny"])
def test_asm_prefetch_registers_are_untouched_until_the_counted_wait(unit):
    """r6: the tile's scale / bias values are requested by inline-asm ``global_load_ushort`` / ``global_load_dword`` ahead of the operand DMA (hipcc would
    drain the DMA queue at a load it can see) and used after a hand-counted ``s_waitcnt vmcnt``.  To hipcc the destination is defined the moment the asm
    statement ends: a register copy or a spill between the load and the wait would read it before the data has landed.  This test reads the listing and
    fails if any instruction between such a load and the next vmcnt wait mentions its destination register."""
    listing = os.path.join(CSRC, "build", unit + ".s")
    proc = subprocess.run(["make", "-C", CSRC, f"build/{unit}.s"], capture_output=True, text=True, timeout=1800)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = open(listing).read().splitlines()
    loads = 0
    for i, ln in enumerate(lines):
        m = re.match(r"\s+global_load_(?:ushort|dword) v(\d+), v\[\d+:\d+\], off\s*$", ln)
        if not m:
            continue
        loads += 1
        dst = int(m.group(1))
        for j in range(i + 1, min(i + 4000, len(lines))):
            nxt = lines[j]
            if "s_waitcnt" in nxt and "vmcnt(" in nxt:
                break
            if re.match(r"\s+global_load_(?:ushort|dword) v\d+, v\[\d+:\d+\], off\s*$", nxt):
                continue  # the neighbouring prefetch loads (their own destinations are checked in turn; addresses are register pairs)
            assert dst not in _regs_in(nxt.split(";")[0]), f"{unit}.s:{j + 1}: v{dst} (asm prefetch at line {i + 1}) is touched before the wait: {nxt.strip()}"
        else:
            raise AssertionError(f"{unit}.s:{i + 1}: no vmcnt wait after the asm prefetch")
    assert loads >= 2, f"{unit}: the prefetch loads were not found ({loads})"
