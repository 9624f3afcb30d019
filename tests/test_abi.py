"""The C-ABI library loads without a GPU and exports every symbol that
include/gtgvae.h declares; the Python binding table covers the same set."""
import ctypes
import os
import re

from gtg_vae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gtgvae.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gtg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    l = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/gtgvae.h but not exported"


def test_binding_table_matches_header():
    assert sorted(_lib.EXPORTS) == _declared()


def test_version_and_limits_callable_without_gpu():
    l = _lib.lib()
    assert l.gtg_version() == 1
    assert l.gtg_max_nodes() == 64
    # pure host-side size queries
    assert l.gtg_encoder_fwd_workspace_bytes(512, 64, 64, 512, 0) >= 512 * 64 * 512 * 4
    assert l.gtg_gemm_splitk_workspace_bytes(512, 1024, 32768) > 0
    # a shape that does not split along K still gets the scratch of the two-plane fp16 mode (operand maxima)
    assert l.gtg_gemm_splitk_workspace_bytes(32768, 512, 1024) == 8192


def test_bad_arguments_are_reported_not_crashed():
    l = _lib.lib()
    rc = l.gtg_mst_batched_triu(None, 4, 65, None, None, None)
    assert rc == -3 and b"64" in l.gtg_last_error()
    rc = l.gtg_adj_build(None, 0, 2, 0, None, None, None, None, None)
    assert rc == -1


def test_host_code_under_address_and_ub_sanitizers():
    """The library's HOST code built with AddressSanitizer + UndefinedBehaviorSanitizer (make hostcheck) and
    driven by tests/host/abi_host_check.cpp: every *_workspace_bytes function over a grid of sizes up to the
    largest accepted, every compute entry point with arguments its validation must refuse before any launch.
    No GPU involved.  The binary is built by __graft_entry__.build(); it is built here if missing."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "gtg-vae_amd", "csrc")
    exe = os.path.join(csrc, "san", "abi_host_check")
    subprocess.run(["make", "-C", csrc, "-j8", "hostcheck"], check=True, capture_output=True, timeout=900)
    # (verify_asan_link_order=0: the process may run with a preloaded library in front of the sanitizer runtime)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "host check ok" in res.stdout and "runtime error" not in res.stderr
