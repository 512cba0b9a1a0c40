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


def _load_order_probe(first):
    """A fresh interpreter that loads libgprhip.so and torch in the given order, then calls gprhip_problem_create and
    gprhip_ctx_create; prints status + message of each and the distinct libamdhip64 files in its link map."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "def lib():\n"
        "    from gpr_amd import _lib\n"
        "    return _lib.load()\n"
        + ("L = lib(); import torch\n" if first == "library" else "import torch; L = lib()\n") +
        "h = C.c_void_p()\n"
        "st = L.gprhip_problem_create(0, 0, 100, 2, 2, 5, 0, C.byref(h)); print('P', st, L.gprhip_last_error().decode())\n"
        "dev = (C.c_int * 1)(0)\n"
        "st = L.gprhip_ctx_create(dev, 1, C.byref(h)); print('C', st, L.gprhip_last_error().decode())\n"
        "print('N', len(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)))\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return {ln[0]: ln[2:] for ln in out.stdout.splitlines() if ln[:2] in ("P ", "C ", "N ")}


def test_two_hip_runtimes_are_refused_by_name(gpu_available):
    """libgprhip.so is linked against the system HIP runtime, torch bundles its own.  Library first, torch second: two
    runtimes are mapped and only one can own the devices -- the creation entry points say so, naming both files, instead
    of leaving the host with "No HIP GPUs are available" from whichever came second.  Torch first: one copy serves both
    (the supported order; INTEGRATION.md, "hosts that also load torch")."""
    pytest.importorskip("torch")
    bad = _load_order_probe("library")
    if bad["N"] == "2":
        for key, who in (("P", "gprhip_problem_create"), ("C", "gprhip_ctx_create")):
            st, msg = bad[key].split(" ", 1)
            assert st == "3" and msg.startswith(who + ": two HIP runtimes are mapped"), bad[key]
            assert "torch/lib/libamdhip64" in msg and "/opt/rocm" in msg and "import torch" in msg
    else:  # an image whose torch uses the system runtime: nothing to refuse
        assert bad["N"] == "1"
    good = _load_order_probe("torch")
    assert good["N"] == "1"
    assert "two HIP runtimes" not in good["P"] and "two HIP runtimes" not in good["C"]
    if gpu_available:
        assert good["P"].split(" ", 1)[0] == "0" and good["C"].split(" ", 1)[0] == "0"


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


def test_plain_c_host_program_compiles_and_links(tmp_path):
    """tests/cpp/eval_latency.c -- a complete host in C99 (create, set inputs / targets, evaluate, destroy) -- compiles
    with -pedantic -Werror against the header and links against the library (no device call here)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "cpp", "eval_latency.c")
    exe = tmp_path / "eval_latency"
    out = subprocess.run(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=199309L", "-Wall", "-Werror", "-pedantic", "-I",
                          os.path.join(ROOT, "include"), src, "-L", os.path.join(ROOT, "gpr_amd"), "-lgprhip", "-lm",
                          "-Wl,-rpath," + os.path.join(ROOT, "gpr_amd"), "-o", str(exe)], capture_output=True, text=True)
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
    needed = {"gprhip_ctx_create", "gprhip_ctx_destroy", "gprhip_sharded_create", "gprhip_sharded_destroy",
              "gprhip_sharded_set_inputs", "gprhip_sharded_set_targets", "gprhip_sharded_eval", "gprhip_sharded_problem",
              "gprhip_sharded_shard", "gprhip_shard_rows", "gprhip_sharded_comm_stats", "gprhip_condition",
              "gprhip_problem_create_ex", "gprhip_problem_destroy", "gprhip_set_inputs", "gprhip_set_targets",
              "gprhip_eval", "gprhip_n_hypers", "gprhip_predict", "gprhip_train_stats", "gprhip_covariances",
              "gprhip_cov_samples", "gprhip_co_variance_coeffs", "gprhip_load_predictor", "gprhip_eval_pass1",
              "gprhip_eval_pass2", "gprhip_eval_finish", "gprhip_ar1_len", "gprhip_ar2_len", "gprhip_last_error"}
    assert needed <= used, needed - used
    fields = re.search(r"typedef struct \{([^}]*)\} gprhip_hypers;", header).group(1)
    names = re.findall(r"(\w+);", fields)
    assert len(names) == 11
    for f in names:
        assert "out->%s" % f in stubs, f
    ml = open(os.path.join(root, "bindings", "gpr_hip.ml")).read()
    for ext in re.findall(r'= "(gprhip_ml_[a-z0-9_]+)"', ml) + re.findall(r'"(gprhip_ml_[a-z0-9_]+)"\s+"', ml):
        assert "value %s(" % ext in stubs, ext


def _sig_vals(text, start_marker, end_marker):
    """{module path: [val names]} of a module type in lib/interfaces.ml between two markers (comments stripped)."""
    a = text.index(start_marker, text.index("module Sigs = struct"))
    body = text[a:text.index(end_marker, a)]
    # strip OCaml comments (they nest)
    res, depth, i = [], 0, 0
    while i < len(body):
        if body.startswith("(*", i):
            depth += 1
            i += 2
        elif body.startswith("*)", i) and depth > 0:
            depth -= 1
            i += 2
        else:
            if depth == 0:
                res.append(body[i])
            i += 1
    body = "".join(res)
    out, stack = {}, []
    for line in body.splitlines():
        m = re.match(r"\s*module (\w+) : sig", line)
        if m:
            stack.append(m.group(1))
            continue
        if re.match(r"\s*end\b", line) and stack:
            stack.pop()
            continue
        m = re.match(r"\s*val (\w+)", line)
        if m and stack:
            out.setdefault(".".join(stack), []).append(m.group(1))
    return out


@pytest.mark.skipif(not os.path.exists("/root/reference/lib/interfaces.ml"),
                    reason="the reference tree exists in the build container only")
def test_ocaml_module_defines_every_value_of_the_reference_signature():
    """bindings/gpr_hip.ml (uncompiled here: no OCaml toolchain) is ascribed to Gpr.Interfaces.Sigs.Deriv: every `val`
    that signature lists under Eval.* and Deriv.* (lib/interfaces.ml:373-1154) must be defined in the module of the
    same name inside Make_variant -- a textual check, the strongest one available without a compiler."""
    ref = open("/root/reference/lib/interfaces.ml").read()
    ml = open(os.path.join(ROOT, "bindings", "gpr_hip.ml")).read()
    ml = ml[ml.index("module Make_variant"):ml.index("module Make_deriv (S : Device_spec)")]
    ev = _sig_vals(ref, "  module type Eval = sig\n    module Spec : Specs.Eval", "  module type Deriv = sig\n    module Eval : Eval")
    dv = _sig_vals(ref, "  module type Deriv = sig\n    module Eval : Eval", "  module type Optimizer = sig")
    assert {"Inducing", "Inputs", "Model", "Trained", "Stats", "Means", "Variances", "Covariances", "Cov_sampler"} <= set(ev)
    assert {"Deriv.Inducing", "Deriv.Inputs", "Deriv.Model", "Deriv.Trained", "Deriv.Test", "Deriv.Optim.Gsl",
            "Deriv.Optim.SGD", "Deriv.Optim.SMD"} <= set(dv)

    def module_body(path, text):
        for name in path:
            m = re.search(r"module %s\b[^=\n]*= struct" % name, text)
            assert m, "gpr_hip.ml has no module %s" % ".".join(path)
            text = text[m.end():]
        return text

    eval_body = module_body(["Eval"], ml)
    deriv_body = ml[ml.index("  module Deriv = struct"):]
    missing = []
    for mod, vals in ev.items():
        body = module_body(mod.split("."), eval_body)
        for v in vals:
            if not re.search(r"\blet (rec )?%s\b" % v, body):
                missing.append("Eval.%s.%s" % (mod, v))
    for mod, vals in dv.items():
        body = module_body(mod.split(".")[1:], deriv_body)
        for v in vals:
            if not re.search(r"\blet (rec )?%s\b" % v, body):
                missing.append("%s.%s" % (mod, v))
    assert not missing, missing


def test_ocaml_stub_file_type_checks_against_the_header():
    """bindings/gpr_hip_stubs.c has never met OCaml's own <caml/*.h> (no OCaml in the image), but everything a C
    compiler can check without them IS checked: `gcc -std=c99 -Wall -Wextra -Werror -fsyntax-only` over the stub file
    against include/gprhip.h (every call of a gprhip_* entry point: argument count and types) and against
    tests/caml_standin/caml/*.h -- a declarations-only stand-in for the part of the documented OCaml C interface the
    stubs use (value, CAMLparam*/CAMLlocal*/CAMLreturn, Caml_ba_array_val, caml_alloc_custom, ...), clearly labelled as
    such, never linked or run."""
    import subprocess
    standin = os.path.join(ROOT, "tests", "caml_standin")
    for h in os.listdir(os.path.join(standin, "caml")):
        assert "STAND-IN, NOT OCaml's header" in open(os.path.join(standin, "caml", h)).read(), h
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", standin,
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "bindings", "gpr_hip_stubs.c")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-3000:]
    # the check has teeth: a stub calling an entry point with one argument too few must be rejected
    src = open(os.path.join(ROOT, "bindings", "gpr_hip_stubs.c")).read()
    assert "gprhip_set_targets(Problem_val(prob), " in src
    broken = src.replace("gprhip_set_targets(Problem_val(prob), ", "gprhip_set_targets(", 1)
    out = subprocess.run(cmd[:-1] + ["-x", "c", "-"], input=broken, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0


@pytest.mark.skipif(not os.path.exists("/root/reference/lib/fitc_gp.ml"),
                    reason="the reference tree exists in the build container only")
def test_ocaml_smd_step_is_the_references_update():
    """bindings/gpr_hip.ml's Optim.SMD (uncompiled): the pieces of the reference's step that a drop-in must not change
    silently (lib/fitc_gp.ml:1952-1995) are there -- a central difference (gradients at +eps and -eps, scaled by
    lambda / (2 eps)), the nu update with the OLD gains, and the reference's defaults (lambda 0.1, mu 1e-3, eta0 = nu0 =
    1e-3)."""
    ref = open("/root/reference/lib/fitc_gp.ml").read()
    assert "scal (lambda /. (2. *. eps)) res" in ref and "Vec.mul old_eta (Vec.add old_gradient lambda_hessian_nu)" in ref
    ml = open(os.path.join(ROOT, "bindings", "gpr_hip.ml")).read()
    smd = ml[ml.index("module SMD = struct"):]
    smd = smd[:smd.index("let test ")]
    assert "grad_at t.eps" in smd and "grad_at (-.t.eps)" in smd
    assert "t.lambda /. (2. *. t.eps)" in smd
    assert "old_eta.{i} *. (old_gradient.{i} +. lambda_hessian_nu.{i})" in smd and "t.lambda *. old_nu.{i}" in smd
    assert "| None -> 0.1" in smd and "| None -> 1e-3" in smd and smd.count("Vec.make n 1e-3") == 2
