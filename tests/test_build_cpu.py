"""Audit of the compiled four-wave GEMM kernels (gemm256w_kernel, devias_amd/csrc/gemm.hip).  Their 256 accumulator registers are AGPRs named
LITERALLY in inline asm: the compiler does not know they are live, so the kernels are correct only if the compiler itself touches no AGPR and
spills nothing (a spill would go through the AGPR half or scratch).  This test compiles the device code with the production flags and checks
exactly that in the ISA, for every instantiation.  CPU only: hipcc cross-compiles gfx950 without a GPU."""
import os
import re
import subprocess
import tempfile

import pytest

from devias_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_four_wave_gemm_kernels_own_their_accumulators():
    src = os.path.join(ROOT, "devias_amd", "csrc", "gemm.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gemm.s")
        cmd = [build.HIPCC] + list(build.FLAGS) + ["--cuda-device-only", "-S", src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = open(out).read().split("\n")
    found = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S*gemm256w_kernel\S*):", l)
        if not m:
            continue
        name = m.group(1)
        found += 1
        j = i
        while "s_endpgm" not in lines[j]:
            j += 1
        inasm, compiler_agpr, scratch, mfma = False, [], 0, 0
        for b in lines[i:j + 1]:
            code = b.split(";")[0]
            if "ASMSTART" in b:
                inasm = True
            elif "ASMEND" in b:
                inasm = False
            elif not inasm and re.search(r"v_accvgpr|\ba\[\d+:\d+\]|\ba\d+\b", code):
                compiler_agpr.append(code.strip())
            scratch += "scratch_" in code
            mfma += "v_mfma" in code
        assert not compiler_agpr, (name, compiler_agpr[:5])
        assert scratch == 0, name
        assert mfma == 128, (name, mfma)                                     # one K-tile of 128 x 128 per wave, nothing duplicated by the compiler
        meta = "\n".join(x for x in lines if name in x and (".num_agpr" in x or ".private_seg_size" in x))
        assert re.search(r"\.num_agpr, 256", meta) and re.search(r"\.private_seg_size, 0\b", meta), meta
    assert found == 4          # <B k-contiguous | k-strided> x <rows the epilogue reads>


@pytest.mark.timeout(600)
def test_layernorm_kernels_do_not_spill():
    """the backward LayerNorm kernel's waves per workgroup are chosen per row width and element type so that the row fits the register budget of that
    occupancy (devias_amd/csrc/layernorm.hip: ln_bwd_dispatch): no instantiation may use scratch"""
    src = os.path.join(ROOT, "devias_amd", "csrc", "layernorm.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ln.s")
        cmd = [build.HIPCC] + list(build.FLAGS) + ["--cuda-device-only", "-S", src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read()
    sizes = re.findall(r"\.set (\S*ln_(?:fwd|bwd)_kernel\S*)\.private_seg_size, (\d+)", text)
    assert len(sizes) >= 16, len(sizes)
    assert all(int(v) == 0 for _, v in sizes), [(n[-40:], v) for n, v in sizes if int(v)]


@pytest.mark.timeout(900)
def test_inline_asm_vmem_never_reads_an_sgpr_the_valu_just_wrote():
    """gfx9 hazard: a vector-memory instruction that reads an SGPR written by the VALU (v_readlane / v_readfirstlane -- which is how the compiler
    reloads a spilled SGPR from a lane of its spill VGPR) needs 5 wait states.  The compiler's hazard recogniser inserts them for its own
    instructions but not inside inline asm, and this library issues stores and atomics from inline asm with an SGPR-pair base (the deferred epilogue
    stores, the stream-K partials, the dynamic tile queue's dequeues).  Round 4 found the failure mode on the GPU: a dequeue whose slot pointer had just
    come out of a spill lane went to an address with a stale high half (memory fault).  This audit walks the ISA of gemm.hip -- release and
    -DDEVIAS_GEMM_DEBUG builds -- and requires, for every inline-asm vector-memory instruction with an SGPR base, that none of the 5 issue slots before it
    (s_nop N counts N + 1) holds a VALU write of that SGPR."""
    src = os.path.join(ROOT, "devias_amd", "csrc", "gemm.hip")
    vmem = re.compile(r"^\s*(global_(?:store|load|atomic)\w*|buffer_\w+)\s+(.*)$")
    valu_sgpr_write = re.compile(r"^\s*(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)\b")
    for extra in ([], ["-DDEVIAS_GEMM_DEBUG"]):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "gemm.s")
            r = subprocess.run([build.HIPCC] + list(build.FLAGS) + extra + ["--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            lines = [l.split(";")[0].rstrip() if not l.lstrip().startswith(";;#") else l.strip() for l in open(out).read().split("\n")]
        inasm, checked, bad = False, 0, []
        code = []                                      # (text, in_asm) of real instructions, in order
        for l in lines:
            if "ASMSTART" in l:
                inasm = True
            elif "ASMEND" in l:
                inasm = False
            elif l.strip() and not l.strip().endswith(":") and not l.strip().startswith("."):
                code.append((l.strip(), inasm))
        for i, (text, ia) in enumerate(code):
            m = vmem.match(text)
            if not (ia and m):
                continue
            base = re.search(r"\bs\[(\d+):(\d+)\]", m.group(2))
            if not base:
                continue
            checked += 1
            regs = set(range(int(base.group(1)), int(base.group(2)) + 1))
            slots, j = 0, i - 1
            while j >= 0 and slots < 5:
                t = code[j][0]
                w = valu_sgpr_write.match(t)
                if w and int(w.group(2)) in regs:
                    bad.append((extra, t, text))
                    break
                nop = re.match(r"s_nop\s+(\d+)", t)
                slots += int(nop.group(1)) + 1 if nop else 1
                j -= 1
        assert checked > 500, (extra, checked)          # the epilogue stores alone are thousands
        assert not bad, bad[:5]
