"""AddressSanitizer + UBSan build of the library's host-only translation units (api.hip, regions.hip) driven through the argument-validation
paths of the C ABI -- no kernel is launched (validation precedes every launch; the size queries ask the HIP runtime for the current device's CU count, which answers "no device" here and falls back to 256) and nothing here runs on the GPU box.  SURVEY.md §5 lists the
sanitizer build among the aux subsystems this stack adds (the reference is pure Python and has none)."""
import glob
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent(r'''
    import ctypes, os, sys
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("lib_defs", os.path.join(ROOT, "devias_amd", "_lib_defs_only.py"))
    from ctypes import byref, c_void_p, c_int64
    lib = ctypes.CDLL(os.environ["DEVIAS_ASAN_LIB"])
    lib.devias_last_error.restype = ctypes.c_char_p
    for n in ("devias_encoder_block_save_bytes", "devias_encoder_block_scratch_bytes", "devias_encoder_block_workspace_bytes", "devias_head_workspace_bytes",
              "devias_head_save_bytes", "devias_agg_block_save_bytes", "devias_agg_block_scratch_bytes", "devias_agg_block_workspace_bytes",
              "devias_gemm_workspace_bytes", "devias_counter"):
        getattr(lib, n).restype = c_int64
    assert lib.devias_version() >= 150
    # options, counters, error strings
    assert lib.devias_set_option(b"no_such_option", 1) == -1 and b"no_such_option" in lib.devias_last_error()
    assert lib.devias_set_option(None, 1) == -1
    assert lib.devias_set_option(b"x" * 4000, 1) == -1 and len(lib.devias_last_error()) < 512          # the error buffer is bounded
    lib.devias_counters_reset(); assert lib.devias_counter(0) == 0 and lib.devias_counter(-5) == -1 and lib.devias_counter(10 ** 6) == -1
    assert lib.devias_allreduce_bucket(None, None, 0, 0, None) == -1
    lib.devias_range_push(b"unit"); lib.devias_range_pop(); lib.devias_range_push(None); lib.devias_range_pop()
    # fused regions: null / zeroed / inconsistent argument structs are refused before anything is enqueued
    Block = type("B", (ctypes.Structure,), {"_fields_": [("raw", ctypes.c_char * 184)]})
    Grads = type("G", (ctypes.Structure,), {"_fields_": [("raw", ctypes.c_char * 112)]})
    a, g = Block(), Grads()
    assert lib.devias_encoder_block_fwd(None, None, None, None) == -1
    assert lib.devias_encoder_block_fwd(byref(a), None, None, None) == -1 and b"devias_encoder_block_fwd" in lib.devias_last_error()
    assert lib.devias_encoder_block_bwd(byref(a), None, None, None, byref(g), None, c_int64(0), None) == -1
    Agg = type("A", (ctypes.Structure,), {"_fields_": [("raw", ctypes.c_char * 2032)]})
    AggG = type("AG", (ctypes.Structure,), {"_fields_": [("raw", ctypes.c_char * 1968)]})
    ag, agg = Agg(), AggG()
    assert lib.devias_agg_block_fwd(byref(ag), None, None, None, None) == -1
    assert lib.devias_agg_block_bwd(byref(ag), None, None, None, None, byref(agg), None, c_int64(0), None) == -1
    assert lib.devias_agg_block_save_bytes(None) == 0 and lib.devias_agg_block_workspace_bytes(None) == 0
    # arena / workspace arithmetic at the measured and at degenerate sizes (int64 throughout: no overflow at ViT-L, 6400 tokens, B = 64)
    for (B, N, D, H) in ((32, 1568, 768, 12), (64, 6400, 1024, 16), (1, 1, 64, 1), (2, 784, 384, 6)):
        for dt in (0, 1):
            s = lib.devias_encoder_block_save_bytes(B, N, D, H, 4 * D, dt)
            assert s > B * N * D * (2 if dt else 4) * 10, (B, N, D, s)
            assert lib.devias_encoder_block_scratch_bytes(B, N, D, H, 4 * D, dt) > 0 and lib.devias_encoder_block_workspace_bytes(B, N, D, H, 4 * D, dt) > 0
    assert lib.devias_encoder_block_save_bytes(64, 6400, 1024, 16, 4096, 1) > 2 ** 31
    assert lib.devias_head_workspace_bytes(64, 768, 466, 512, 256, 196, 1) > 0 and lib.devias_head_save_bytes(64, 768, 512, 256, 1) > 0
    Head = type("H", (ctypes.Structure,), {"_fields_": [("raw", ctypes.c_char * 120)]})
    assert lib.devias_head_fwd(byref(Head()), None, None, None, None, None) == -1
    # the split policies never divide by zero / return < 1
    for M in (1, 64, 256, 257, 50176):
        for N in (1, 196, 765, 768, 3072):
            for K in (1, 63, 64, 512, 50176):
                assert lib.devias_policy_small_m_split(M, N, K, 0) >= 1 and lib.devias_policy_wgrad_split(M, N, K, 1) >= 1 and lib.devias_policy_wgrad_split(M, N, K, 0) >= 1
    lib.devias_shutdown()
    print("ASAN-DRIVER-OK")
''')


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    sys.path.insert(0, ROOT)
    from devias_amd import build as b
    lib = b.build_asan(str(tmp_path / "libdevias_amd_asan.so"))
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    assert rt, "clang AddressSanitizer runtime not found"
    import torch
    env = dict(os.environ)
    env.update({"LD_PRELOAD": rt[0], "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=66", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
                "DEVIAS_ASAN_LIB": lib, "DEVIAS_ROCTX": "0",
                "LD_LIBRARY_PATH": os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")})
    script = tmp_path / "driver.py"
    script.write_text("ROOT = %r\n" % ROOT + DRIVER)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ASAN-DRIVER-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
