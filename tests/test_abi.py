"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/modcr_hip.h declares, and the ctypes table mirrors the header (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "modcr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(modcr_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("void", "") else args.count(",") + 1
    return out


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    import modcr_hip
    return modcr_hip


def test_header_declares_the_survey_minimum_set():
    fns = header_functions()
    for name in ("modcr_qkv_attn_fwd", "modcr_chunk_mean_q_fwd", "modcr_proj_residual_ln_fwd",
                 "modcr_ffn_up_gelu_fwd", "modcr_ffn_down_residual_ln_fwd", "modcr_embed_ln_fwd",
                 "modcr_align_attn_fwd", "modcr_align_attn_bwd", "modcr_mc_ce_fwd_bwd", "modcr_version"):
        assert name in fns, name


def test_library_exports_every_declared_symbol(built):
    for path in (built.LIB_PATH, built.TUNING_LIB_PATH):
        lib = ctypes.CDLL(path)
        fns = header_functions()
        assert len(fns) >= 20
        for name in fns:
            assert hasattr(lib, name), "%s does not export %s" % (os.path.basename(path), name)


def test_product_library_reads_no_environment_variable(built):
    """timing-only / A-B knobs must not ship in the product .so.  The product library does not even
    import getenv; the tuning build (tools/, forced-path tests) does."""
    import subprocess
    def imports_getenv(path):
        out = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        return any(l.split()[-1].split("@")[0] == "getenv" for l in out.splitlines() if l.strip())
    assert not imports_getenv(built.LIB_PATH)
    assert imports_getenv(built.TUNING_LIB_PATH)
    strings = subprocess.run(["strings", built.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "MODCR_ATTN_DEBUG" not in strings and "MODCR_GEMM_ORDER" not in strings


def test_ctypes_table_matches_header(built):
    fns = header_functions()
    assert set(fns) == set(built.SIGNATURES), set(fns) ^ set(built.SIGNATURES)
    for name, nargs in fns.items():
        assert len(built.SIGNATURES[name][1]) == nargs, name


def test_version_and_error_string(built):
    lib = built.lib()
    assert lib.modcr_version() == 100
    # an invalid call must return an error code and a message, not crash (no GPU needed: argument
    # validation happens before any launch)
    rc = lib.modcr_linear_fwd(None, 0, None, 0, None, None, 0, 0, None, 0, 0, 0, 0, 0, 0, 0, None)
    assert rc == -1
    assert b"linear_fwd" in lib.modcr_last_error()


def test_missing_library_fails_loudly(built, monkeypatch):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libmodcr_hip.so")
    with pytest.raises(built.ModcrHipError):
        built.lib()
