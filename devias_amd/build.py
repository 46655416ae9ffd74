"""Build libdevias_amd.so in-tree with hipcc for gfx950 (no torch / pybind dependency: the ABI is plain C).

    python -m devias_amd.build          # build if sources are newer than the library
    python -m devias_amd.build --force

The library is linked against libamdhip64.so.7 only; inside a PyTorch process the already-loaded HIP runtime
bundled with torch (same soname) satisfies it, so no second runtime is pulled in.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdevias_amd.so")
SOURCES = ["api.hip", "gemm.hip", "elementwise.hip", "layernorm.hip", "attention.hip", "slot_attn.hip", "loss.hip", "fame.hip", "regions.hip", "probe.hip", "attn_bwd1w.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HEADERS = ["common.h", "roctx_shim.h", "attn1w.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Wno-unused-result",
         "-fno-gpu-rdc", "-mllvm", "-amdgpu-early-inline-all=true",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]   # keep MFMA accumulators in VGPRs: no v_accvgpr_* shuffles around the VALU epilogues


# per-file additions.  attention.hip: no SLP vectorisation -- the compiler packs adjacent scalar fp32 adds / multiplies of the softmax arithmetic into v_pk_* instructions,
# and a packed fp32 instruction beside MFMAs costs more issue time than the two it replaces (MI355X_MICROARCH.md, per-instruction constants): forward attention -4 %
# (profiles/r4_packed_fp32.txt)
FILE_FLAGS = {"attention.hip": ["-fno-slp-vectorize"], "attn_bwd1w.hip": ["-fno-slp-vectorize"]}


def _flags(src: str):
    return FLAGS + FILE_FLAGS.get(src, [])


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources, the C ABI header and the compiler flags: identifies the code a library / a profile was built from
    without needing .git (the GPU box has none).  bench.py quotes PMC traffic only from a profile whose hash matches."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES) + HEADERS:
        h.update(name.encode()); h.update(open(os.path.join(CSRC, name), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "devias_amd.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()[:16]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, "..", "include", "devias_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str) -> str:
    obj = os.path.join(CSRC, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, "..", "include", "devias_amd.h")]
    if os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps):
        return obj
    cmd = [HIPCC] + _flags(src) + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build libdevias_amd.so")
    if force:
        for s in SOURCES:
            o = os.path.join(CSRC, s.replace(".hip", ".o"))
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB)")
    return LIB


def build_asan(out_path: str) -> str:
    """CPU-side sanitizer build (the reference has no native code to sanitize; this library's host side does): the two host-only
    translation units -- api.hip (error strings, options, counters, RCCL / roctx resolution) and regions.hip (argument structs, arena layouts,
    workspace arithmetic of the fused regions) -- compiled with AddressSanitizer + UBSan on the HOST side only, linked with the regular objects
    of the kernel files.  Never run on the GPU box (GPU sanitizers are unavailable there); tests/test_sanitizer_cpu.py drives the
    argument-validation paths of every entry point through it without launching anything (the workspace-size queries do ask the HIP runtime for the
    current device's CU count; without a device they fall back to 256)."""
    build(force=False, verbose=False)                  # regular objects of the kernel translation units
    san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-fno-sanitize-recover=undefined", "-g"]
    objs = []
    for src in SOURCES:
        if src in ("api.hip", "regions.hip"):
            obj = os.path.join(os.path.dirname(out_path), src.replace(".hip", ".asan.o"))
            cmd = [HIPCC] + [f for f in _flags(src) if f != "-O3"] + ["-O1"] + san + ["-c", os.path.join(CSRC, src), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc (sanitizer build) failed for {src}:\n{r.stdout}\n{r.stderr}")
            objs.append(obj)
        else:
            objs.append(os.path.join(CSRC, src.replace(".hip", ".o")))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", out_path] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link (sanitizer build) failed:\n{r.stdout}\n{r.stderr}")
    return out_path


if __name__ == "__main__":
    build(force="--force" in sys.argv)
