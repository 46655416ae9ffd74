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
_ISA = {}


def _isa(stem, extra=()):
    """the gfx950 ISA of one translation unit under the production flags (+ extra), compiled once per session: gemm.hip takes over a minute"""
    key = (stem, tuple(extra))
    if key not in _ISA:
        src = os.path.join(ROOT, "devias_amd", "csrc", stem + ".hip")
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, stem + ".s")
            r = subprocess.run([build.HIPCC] + list(build._flags(stem + ".hip")) + list(extra) + ["--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            _ISA[key] = open(out).read()
    return _ISA[key]


@pytest.mark.timeout(900)
def test_four_wave_gemm_kernels_own_their_accumulators():
    lines = _isa("gemm").split("\n")
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


@pytest.mark.timeout(900)
def test_eight_wave_persistent_gemm_kernels_do_not_spill():
    """gemm256p_kernel sits at 185-245 of the 256 registers two waves per SIMD allow, and a scratch reload costs more than its instructions there: its s_waitcnt vmcnt(0)
    drains the tile's whole store burst, which the kernel otherwise leaves in flight under the next tile's K loop.  Round 6 found how close that is: the first
    specialised epilogues with column sums (EPI_CS) spilled 52-132 bytes -- with no branch left between the sixteen pieces the compiler sank all 128 column-sum adds
    behind the last piece and kept every piece alive for them.  Every instantiation -- <B layout, side rows, static / dynamic, generic / specialised epilogue> --
    must therefore have no scratch, exactly the MFMAs of its K-tile bodies, and every specialisation the host dispatches to must exist."""
    text = _isa("gemm")
    sizes = dict(re.findall(r"\.set (\S*gemm256p_kernel\S*)\.private_seg_size, (\d+)", text))
    assert len(sizes) == 22, sorted(sizes)       # 2 (dynamic) x [ (F,0): generic, bias, bias+GELU+aux, none, colsum | (F,1): generic, bias | (F,2): generic, dGELU+colsum | (T,0): generic | (T,2): generic ]
    assert all(int(v) == 0 for v in sizes.values()), {k[-34:]: v for k, v in sizes.items() if int(v)}
    lines = text.split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S*gemm256p_kernel\S*):", l)
        if not m:
            continue
        j = i
        while not lines[j].startswith(".Lfunc_end"):        # (not the first s_endpgm: a workgroup that finds every queue empty leaves early)
            j += 1
        body = [b.split(";")[0] for b in lines[i:j + 1]]
        assert sum("scratch_" in b for b in body) == 0, m.group(1)
        # one K-tile of MFMAs in the persistent loop; the static-list, B-k-contiguous, no-column-sum instantiations also carry the six row ranges of a tail tile's thirds /
        # quarters ([0,5) [5,8) [0,3) [3,8) [0,4) [4,8): 24 row tiles x 4 column tiles x 2 k-steps)
        t = re.search(r"gemm256p_kernelILb(\d)ELi(\d)ELb(\d)ELi(n?\d+)E", m.group(1))
        tb, dyn, epi = t.group(1) == "1", t.group(3) == "1", int(t.group(4).replace("n", "-"))
        pieces = not tb and not dyn and not (epi >= 0 and epi & 64)
        assert sum("v_mfma" in b for b in body) == 64 + (192 if pieces else 0), m.group(1)


@pytest.mark.timeout(600)
def test_layernorm_kernels_do_not_spill():
    """the backward LayerNorm kernel's waves per workgroup are chosen per row width and element type so that the row fits the register budget of that
    occupancy (devias_amd/csrc/layernorm.hip: ln_bwd_dispatch): no instantiation may use scratch"""
    text = _isa("layernorm")
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
    vmem = re.compile(r"^\s*(global_(?:store|load|atomic)\w*|buffer_\w+)\s+(.*)$")
    valu_sgpr_write = re.compile(r"^\s*(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)\b")
    for extra in ([], ["-DDEVIAS_GEMM_DEBUG"]):
        lines = [l.split(";")[0].rstrip() if not l.lstrip().startswith(";;#") else l.strip() for l in _isa("gemm", extra).split("\n")]
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


def _reg_set(tok: str):
    """VGPR numbers named by an operand token: v12, v[4:7]; anything else -> empty"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("stem,kernel,mfmas,agprs", [("attn_bwd1w", "mhsa_bwd_dkdv1w_kernel", 32, 192)])
def test_one_wave_attention_backward_kernels_own_their_agprs_and_keep_their_hazard_windows(stem, kernel, mfmas, agprs):
    """Audit of the one-wave-per-SIMD attention backward kernels (mhsa_bwd_dkdv1w_kernel, devias_amd/csrc/attn_bwd1w.hip; parametrised: a dQ kernel of the same build lives under tools/exp).
    (1) their accumulators and resident operand fragments (a[0:191] / a[0:127]) are named literally in inline asm: the compiler must touch none of THOSE AGPRs itself and spill nothing.  (2) Its MFMAs are inline asm, so the
    compiler's hazard recognizer does not see them; the source keeps the windows by construction and this test checks the result in the ISA of every
    instantiation: behind an MFMA that writes VGPRs no other instruction reads or writes those VGPRs before two further MFMAs have issued (XDL write -> VALU /
    LDS access needs 11 wait states at 8 passes), and no vector-ALU instruction overwrites the VGPRs of its C operand before one further MFMA has issued (the
    MFMA reads C while it runs: 7 wait states; LDS loads into those registers return much later and are fine).  (3) the kernel holds exactly 8 (prologue) + 32 per ring stage (the slice loop is unrolled over the stages) + 8 (drain) MFMAs."""
    src = os.path.join(ROOT, "devias_amd", "csrc", stem + ".hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, stem + ".s")
        cmd = [build.HIPCC] + list(build._flags(stem + ".hip")) + ["--cuda-device-only", "-S", src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = open(out).read().split("\n")
    found = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S*" + kernel + r"\S*):", l)
        if not m:
            continue
        name = m.group(1)
        found += 1
        j = i
        while not lines[j].startswith(".Lfunc_end"):        # (not the first s_endpgm: a wave without keys leaves early)
            j += 1
        inasm, compiler_agpr, scratch, in_loop = False, [], 0, False
        body = []                                            # (opcode, operand tokens, in_asm) of every instruction
        loops = []                                           # MFMA count per innermost loop
        for b in lines[i:j + 1]:
            code = b.split(";")[0].strip()
            if "ASMSTART" in b:
                inasm = True; continue
            if "ASMEND" in b:
                inasm = False; continue
            if "Inner Loop Header" in b:
                loops.append(0); in_loop = b.split(":")[0].strip()             # the header's label; the loop ends at the branch back to it
            if not code or code.endswith(":") or code.startswith("."):
                continue
            if not inasm and stem == "attn_fwd1w":              # (the retired forward kernel, tools/exp/attn_fwd1w.hip.txt: its score tiles live in v[224:255], which only its inline asm may name)
                vs = [int(x) for x in re.findall(r"\bv(\d+)\b", code)] + [int(x) for y in re.findall(r"\bv\[(\d+):(\d+)\]", code) for x in y]
                assert all(v < 224 for v in vs), (name, "the compiler uses a VGPR that belongs to the asm-owned score tiles", code)
            if not inasm:                                      # the compiler may use AGPRs of its own ABOVE the claimed ones (the persistent form parks the next block's K / V rows there)
                used = [int(x) for x in re.findall(r"\ba(\d+)\b", code)] + [int(x) for y in re.findall(r"\ba\[(\d+):(\d+)\]", code) for x in y]
                if any(u < agprs for u in used):
                    compiler_agpr.append(code)
            scratch += "scratch_" in code
            parts = code.replace(",", " ").split()
            body.append((parts[0], parts[1:], inasm))
            if parts[0].startswith("v_mfma") and in_loop:
                loops[-1] += 1
            if (parts[0].startswith("s_cbranch") or parts[0] == "s_branch") and in_loop and parts[1] == in_loop:
                in_loop = False
        assert not compiler_agpr, (name, compiler_agpr[:5])
        assert scratch == 0, name
        total_mfma = sum(1 for op, _, _ in body if op.startswith("v_mfma"))
        # dK / dV: prologue 8 + drain 8 + one slice step per ring stage (4 or 8); forward: prologue 8 + one step per stage (8) + the last slice's step (8): nothing duplicated
        assert total_mfma in ((16 + 4 * mfmas, 16 + 8 * mfmas) if stem == "attn_bwd1w" else (8 + 8 * mfmas + 8,)), (name, total_mfma, loops)
        for k, (op, args, _) in enumerate(body):
            if not op.startswith("v_mfma") or not args[0].startswith("v"):
                continue                                     # (MFMAs that write AGPRs: their registers are nobody else's)
            dst, srcc = _reg_set(args[0]), _reg_set(args[3])
            seen_mfma, nop_states = 0, 0
            for op2, args2, _ in body[k + 1:k + 40]:
                if op2.startswith("v_mfma"):
                    seen_mfma += 1
                    if seen_mfma >= 2:
                        break
                    continue
                if op2 == "s_nop":
                    nop_states += int(args2[0]) + 1
                    if nop_states >= 32:                 # an explicit pad of more than the longest hazard window (18 wait states) is as good as two MFMAs
                        break
                    continue
                if op2.startswith("s_") or op2.startswith("buffer_load"):
                    continue
                touched = set().union(*[_reg_set(a) for a in args2]) if args2 else set()
                assert not (touched & dst), (name, "an instruction touches the VGPRs an asm MFMA is still writing", op, args, op2, args2)
                if seen_mfma == 0 and op2.startswith("v_") and srcc != dst:
                    assert not (_reg_set(args2[0]) & srcc), (name, "a vector instruction overwrites the C operand of a running asm MFMA", op, args, op2, args2)
        meta = "\n".join(x for x in lines if name in x and (".num_agpr" in x or ".private_seg_size" in x))
        m2 = re.search(r"\.num_agpr, (\d+)", meta)
        assert m2 and int(m2.group(1)) >= agprs and re.search(r"\.private_seg_size, 0\b", meta), meta
    assert found == (6 if stem == "attn_bwd1w" else 1)          # dK / dV: <4 waves, 8 stages> for whole 256-key blocks (persistent / one block per workgroup); for the ragged rest of a head <1, 4> (1, 2 or 4 waves share its queries) and <2, 4>
