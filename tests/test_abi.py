"""The C-ABI library loads without a GPU and exports every symbol include/gprhip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gprhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gprhip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("gprhip_problem_create", "gprhip_eval", "gprhip_eval_pass1", "gprhip_eval_pass2",
                 "gprhip_eval_finish", "gprhip_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from gpr_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), "libgprhip.so does not export %s" % name


def test_binding_table_matches_header():
    from gpr_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_no_device_fails_loudly(gpu_available):
    """There is no CPU fallback: creating a problem without a GPU raises."""
    import gpr_amd
    if gpu_available:
        pytest.skip("a GPU is present")
    with pytest.raises(gpr_amd.GprHipError):
        gpr_amd.Problem(gpr_amd.COV_SE_ISO, 10, 2, 2, 3)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under gpr_amd/ may import or execute it."""
    pkg = os.path.join(ROOT, "gpr_amd")
    pat = re.compile(r"(^|\n)\s*(from|import)\s+oracle\b|fitc_oracle|staged_double")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), "%s references the oracle" % os.path.join(dirpath, f)


def test_cpp_mirror_program_is_built_and_links():
    """include/gprhip.hpp (C++ mirror of Fitc_gp.Make_deriv) compiles against the C ABI: the Makefile builds
    tests/cpp/mirror_check.cpp with g++; without arguments it only prints its usage (no device call)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "gpr_amd", "_build", "mirror_check")
    assert os.path.exists(exe), "run `make -C gpr_amd/csrc` (or __graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "usage" in out.stderr


def test_header_is_plain_c(tmp_path):
    """include/gprhip.h is the boundary an OCaml stub (C) includes: it must compile as C99 without C++."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('#include "gprhip.h"\nint main(void) { gprhip_hypers h; gprhip_result r; (void)h; (void)r; '
                   'return sizeof(h) + sizeof(r) == 0; }\n')
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                          "-c", str(src), "-o", str(tmp_path / "use.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_ocaml_stub_sources_cover_the_abi():
    """bindings/gpr_hip_stubs.c (shipped uncompiled: no OCaml toolchain in the image) names only functions the header
    declares, binds every entry point a host needs, and fills every field of gprhip_hypers."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stubs = open(os.path.join(root, "bindings", "gpr_hip_stubs.c")).read()
    header = open(os.path.join(root, "include", "gprhip.h")).read()
    declared = set(re.findall(r"\b(gprhip_[a-z0-9_]+)\s*\(", header))
    used = set(re.findall(r"\b(gprhip_(?!ml_)[a-z0-9_]+)\s*\(", stubs))
    assert used <= declared, used - declared
    needed = {"gprhip_problem_create_ex", "gprhip_problem_destroy", "gprhip_set_inputs", "gprhip_set_targets",
              "gprhip_eval", "gprhip_n_hypers", "gprhip_predict", "gprhip_train_stats", "gprhip_covariances",
              "gprhip_cov_samples", "gprhip_co_variance_coeffs", "gprhip_load_predictor", "gprhip_eval_pass1",
              "gprhip_eval_pass2", "gprhip_eval_finish", "gprhip_ar1_len", "gprhip_ar2_len", "gprhip_last_error"}
    assert needed <= used, needed - used
    fields = re.search(r"typedef struct \{(.*?)\} gprhip_hypers;", header, re.S).group(1)
    names = re.findall(r"(\w+);", fields)
    assert len(names) == 11
    for f in names:
        assert "out->%s" % f in stubs, f
    ml = open(os.path.join(root, "bindings", "gpr_hip.ml")).read()
    for ext in re.findall(r'= "(gprhip_ml_[a-z0-9_]+)"', ml) + re.findall(r'"(gprhip_ml_[a-z0-9_]+)"\s+"', ml):
        assert "value %s(" % ext in stubs, ext
