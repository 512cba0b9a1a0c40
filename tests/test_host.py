"""Host-side logic that needs no GPU: hyper-parameter enumeration / get / set of the two covariance
specs, row sharding, and the world_size-2 all-reduce path of gpr_amd.dist over gloo with the CPU
staged double standing in for the device."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpr_amd import cov_se_fat, cov_se_iso
from gpr_amd.dist import ShardedProblem, shard_rows
from oracle import fitc_oracle as O
from tests.staged_double import StagedDouble
from tests.util import relinf, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_iso_hyper_enumeration_matches_reference_order():
    k = cov_se_iso.Kernel.create(cov_se_iso.Params(0.3, -0.2))
    Z = np.asfortranarray(np.arange(6.0).reshape(2, 3))
    hs = cov_se_iso.HyperModule.get_all(k, Z)
    assert hs[:2] == ["Log_ell", "Log_sf2"]
    assert hs[2:] == [cov_se_iso.Inducing_hyper(i, d) for i in (1, 2, 3) for d in (1, 2)]
    assert [cov_se_iso.HyperModule.index_of(k, Z, h) for h in hs] == list(range(len(hs)))
    assert cov_se_iso.HyperModule.get_value(k, Z, None, hs[0]) == 0.3
    assert cov_se_iso.HyperModule.get_value(k, Z, None, cov_se_iso.Inducing_hyper(3, 2)) == Z[1, 2]
    oracle_order = O.se_iso_hypers(2, 3)
    assert [(h if isinstance(h, str) else ("inducing", h.ind, h.dim)) for h in hs][2:] == oracle_order[2:]


def test_iso_set_values_copies_inducing_lazily():
    k = cov_se_iso.Kernel.create(cov_se_iso.Params(0.0, 0.0))
    Z = np.asfortranarray(np.zeros((2, 3)))
    k2, Z2, _ = cov_se_iso.HyperModule.set_values(k, Z, None, ["Log_sf2"], [1.5])
    assert Z2 is Z and k2.params.log_sf2 == 1.5 and abs(k2.sf2 - np.exp(1.5)) < 1e-15
    k3, Z3, _ = cov_se_iso.HyperModule.set_values(k, Z, None, [cov_se_iso.Inducing_hyper(2, 1)], [7.0])
    assert Z3 is not Z and Z3[0, 1] == 7.0 and Z[0, 1] == 0.0
    assert abs(k.inv_ell2_05 + 0.5) < 1e-15


def test_fat_hypers_and_param_checks():
    with pytest.raises(ValueError, match="disagrees with target dimension"):
        cov_se_fat.Params.create(3, 0.0, tproj=np.ones((5, 2)))
    with pytest.raises(ValueError, match="log_multiscales_m05"):
        cov_se_fat.Params.create(3, 0.0, log_multiscales_m05=np.zeros((2, 4)))
    km = cov_se_fat.Kernel.create(cov_se_fat.Params.create(2, 0.0, np.ones((3, 2)), np.zeros(2), np.zeros((2, 2))))
    hm = cov_se_fat.HyperModule.get_all(km, np.zeros((2, 2), order="F"))
    assert hm[-4:] == [cov_se_fat.Log_multiscale_m05(i, d_) for i in (1, 2) for d_ in (1, 2)]   # lib/cov_se_fat.ml:334-341
    assert [cov_se_fat.HyperModule.index_of(km, np.zeros((2, 2)), h) for h in hm] == list(range(len(hm)))
    kh = cov_se_fat.Kernel.create(cov_se_fat.Params.create(2, 0.0, np.ones((3, 2)), np.zeros(2)))
    hh = cov_se_fat.HyperModule.get_all(kh, np.zeros((2, 2), order="F"))
    assert hh[-2:] == [cov_se_fat.Log_hetero_skedasticity(1), cov_se_fat.Log_hetero_skedasticity(2)]
    assert [cov_se_fat.HyperModule.index_of(kh, np.zeros((2, 2)), h) for h in hh] == list(range(len(hh)))
    P = np.ones((3, 2))
    k = cov_se_fat.Kernel.create(cov_se_fat.Params.create(2, 0.1, P))
    Z = np.zeros((2, 2), order="F")
    hs = cov_se_fat.HyperModule.get_all(k, Z)
    assert hs[0] == "Log_sf2" and len(hs) == 1 + 4 + 6
    assert hs[5:] == [cov_se_fat.Proj_hyper(b, s) for b in (1, 2, 3) for s in (1, 2)]
    assert [cov_se_fat.HyperModule.index_of(k, Z, h) for h in hs] == list(range(len(hs)))
    k2, _, _ = cov_se_fat.HyperModule.set_values(k, Z, None, [cov_se_fat.Proj_hyper(3, 2)], [9.0])
    assert k2.params.tproj[2, 1] == 9.0 and k.params.tproj[2, 1] == 1.0


def test_gradient_family_slices_follow_hyper_get_all():
    """tests/margins.families (the per-family gradient checks of the parity suite) against the mirrors' Hyper.get_all
    (lib/cov_se_iso.ml:188-202, lib/cov_se_fat.ml:290-342): every hyper falls into the slice of its own family, and a
    family-wise error hides nothing a whole-vector max-norm would hide."""
    from tests import margins as M
    d, m, D = 2, 3, 4
    Z = np.zeros((d, m), order="F")
    kind_of = lambda h: {"Log_ell": "log_ell", "Log_sf2": "log_sf2"}.get(h) if isinstance(h, str) else {
        "Inducing_hyper": "inducing", "Proj_hyper": "proj", "Log_hetero_skedasticity": "hetero",
        "Log_multiscale_m05": "multiscale"}[type(h).__name__]
    hs = cov_se_iso.HyperModule.get_all(cov_se_iso.Kernel.create(cov_se_iso.Params(0.0, 0.0)), Z)
    fams = M.families("iso", d, m)
    assert fams[-1][1].stop == len(hs)
    for name, sl in fams:
        assert all(kind_of(h) == name for h in hs[sl]), name
    for proj, het, ms in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1), (1, 0, 1)]:
        k = cov_se_fat.Kernel.create(cov_se_fat.Params.create(
            d, 0.0, np.ones((D, d)) if proj else None, np.zeros(m) if het else None, np.zeros((d, m)) if ms else None))
        hs = cov_se_fat.HyperModule.get_all(k, Z)
        fams = M.families("fat", d, m, D=D, proj=bool(proj), het=bool(het), ms=bool(ms))
        assert fams[-1][1].stop == len(hs), (proj, het, ms)
        for name, sl in fams:
            assert all(kind_of(h) == name for h in hs[sl]), (name, proj, het, ms)
    # a percent-level error in the inducing block next to a huge Log_ell entry: invisible to the whole-vector norm
    ref = np.concatenate([[1e6, -3e5], np.linspace(1.0, 50.0, d * m)])
    got = ref.copy()
    got[4] *= 1.01
    assert relinf(got, ref) < 1e-6
    errs = M.family_errors(got, ref, M.families("iso", d, m))
    assert errs["log_ell"] == 0.0 and errs["inducing"] > 1e-3
    with pytest.raises(AssertionError, match="inducing"):
        M.check_grad(got, ref, M.families("iso", d, m), 1e-7)


def test_memory_plan_of_the_baseline_configs():
    """gprhip_memory_plan (device-free): BASELINE.json configs[3] (n = 8M, m = 4096, d = 16, fp64) per shard of its 1 / 2 / 4 /
    8-way row partition against the 288 GB of an MI355X -- one device cannot hold it, two can (187.5 GB each), the 8-way
    shard takes 64.6 GB; the headline problem 27 GB.  The parts add up and the V store is rows x padded columns."""
    import gpr_amd
    HBM = 288e9
    tot = {}
    for nd in (1, 2, 4, 8):
        lo, hi = shard_rows(8_000_000, nd - 1, nd)
        plan = gpr_amd.memory_plan(gpr_amd.COV_SE_ISO, hi - lo, 16, 16, 4096)
        parts = sum(plan[k] for k in ("v_store", "chunk_buffers", "slices", "inputs", "row_vectors", "mxm", "rest"))
        assert parts == plan["total"]
        chunks = -(-(hi - lo) // plan["chunk_rows"])
        assert plan["v_store"] == chunks * plan["chunk_rows"] * 4096 * 8
        tot[nd] = plan["total"]
    assert tot[1] > HBM and tot[2] < HBM and tot[8] < 70e9 and tot[8] < tot[4] < tot[2] < tot[1]
    c2 = gpr_amd.memory_plan(gpr_amd.COV_SE_ISO, 1_000_000, 8, 8, 2048)
    assert 25e9 < c2["total"] < 30e9
    c3 = gpr_amd.memory_plan(gpr_amd.COV_SE_FAT, 1_000_000, 32, 32, 4096, precision=gpr_amd.F32_BULK)
    assert c3["k_store_optional"] == c3["v_store"] == 1_048_576 * 4096 * 4 and c3["total"] + c3["k_store_optional"] < HBM
    # mid-size shards reserve the split-K factor they will use, not the cap (n = 100 000, m = 1024 used to take 128 slices)
    mid = gpr_amd.memory_plan(gpr_amd.COV_SE_ISO, 100_000, 8, 8, 1024)
    assert 8 <= mid["kslices"] < 128 and mid["slices"] < mid["v_store"]
    with pytest.raises(gpr_amd.GprHipError):
        gpr_amd.memory_plan(gpr_amd.COV_SE_ISO, 0, 8, 8, 1024)


def test_shard_rows_partitions_exactly():
    for n, w in ((1_000_000, 8), (10, 3), (7, 7), (1001, 4)):
        spans = [shard_rows(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_library_row_partition_is_the_one_the_process_per_gpu_path_uses():
    """gprhip_shard_rows -- the partition of the single-process context (gprhip_sharded_create) -- is pure arithmetic
    and runs without a device: identical to gpr_amd.dist.shard_rows, ragged counts and argument checks included."""
    import ctypes as C
    from gpr_amd import _lib
    lib = _lib.load()
    for n, w in ((1_000_000, 8), (8_000_000, 8), (160003, 8), (10, 3), (7, 7), (1001, 4), (5, 1)):
        for r in range(w):
            lo, hi = C.c_int64(), C.c_int64()
            assert lib.gprhip_shard_rows(n, w, r, C.byref(lo), C.byref(hi)) == _lib.OK
            assert (lo.value, hi.value) == shard_rows(n, r, w)
    lo, hi = C.c_int64(), C.c_int64()
    for bad in ((0, 1, 0), (10, 0, 0), (10, 2, 2), (10, 2, -1)):
        assert lib.gprhip_shard_rows(*bad, C.byref(lo), C.byref(hi)) == _lib.EBADARG
    assert b"gprhip_shard_rows" in lib.gprhip_last_error()


def test_context_without_a_device_fails_loudly(gpu_available):
    import gpr_amd
    if gpu_available:
        pytest.skip("a GPU is present")
    with pytest.raises(gpr_amd.GprHipError):
        gpr_amd.Context([0])


def _factory(log_ell, log_sf2, tproj):
    return O.SeIsoKernel(log_ell, log_sf2)


@pytest.mark.parametrize("variational", [False, True])
def test_staged_double_matches_reference_sequence_oracle(variational):
    """The build's two-pass formulation (numpy restatement) against the reference-sequence oracle."""
    X, y, Z = synth(6, 400, 20, 3)
    le = 0.5 * np.log(3)
    ref = O.evaluate_fast(O.SeIsoKernel(le, 0.1), Z, X, y, 0.1, variational=variational)
    sd = StagedDouble(_factory, 400, 3, 3, 20)
    sp = ShardedProblem(0, 400, 3, 3, 20, rank=0, world=1, backend=sd)
    sp.set_inputs(X)
    sp.set_targets(y)
    ev = sp.eval(log_ell=le, log_sf2=0.1, sigma2=0.1, inducing=Z, variational=variational)
    assert abs(ev.l - ref["l"]) < 1e-9 * abs(ref["l"])
    assert abs(ev.dl_dsigma2 - ref["dl_dsigma2"]) < 1e-7 * abs(ref["dl_dsigma2"])
    assert relinf(ev.grad, ref["grad"]) < 1e-6
    assert relinf(ev.coeffs, ref["coeffs"]) < 1e-6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q, shape=(501, 16, 3)):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, m, d = shape
        X, y, Z = synth(8, n, m, d)
        lo, hi = shard_rows(n, rank, world)
        # the row partition of the single-process context (gprhip_shard_rows, device-free) is the same one
        import ctypes as C
        from gpr_amd import _lib
        a, b = C.c_int64(), C.c_int64()
        _lib.check(_lib.load().gprhip_shard_rows(n, world, rank, C.byref(a), C.byref(b)))
        assert (a.value, b.value) == (lo, hi)
        sd = StagedDouble(_factory, hi - lo, d, d, m)
        sp = ShardedProblem(0, n, d, d, m, backend=sd)
        # the buffers this test all-reduces have the length and packed layout of the device path's
        mp, nt = (m + 127) // 128 * 128, (m + 127) // 128
        packed = nt * (nt + 1) // 2 * 128 * 128  # upper 128 x 128 tiles only
        assert sp.ar1.numel() == sd.ar1_len() == packed + mp + 4 and sp.ar2.numel() == packed + (d + 1) * mp + 8
        sp.set_inputs(X[:, lo:hi])
        sp.set_targets(y[lo:hi])
        ev = sp.eval(log_ell=0.4, log_sf2=0.0, sigma2=0.1, inducing=Z)
        c_grad = sp.collectives
        ev0 = sp.eval(log_ell=0.4, log_sf2=0.0, sigma2=0.1, inducing=Z, want_grad=False)
        # two collectives per gradient evaluation, one per evidence-only evaluation
        assert c_grad == 2 and sp.collectives - c_grad == 1, (c_grad, sp.collectives)
        q.put((rank, ev.l, ev.dl_dsigma2, ev.grad, ev0.l))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", [(501, 16, 3), (703, 150, 3)])  # one tile of inducing points; two tile rows (three packed tiles)
def test_world_size_2_sharded_eval_over_gloo(shape):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, shape)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, m, d = shape
    X, y, Z = synth(8, n, m, d)
    ref = O.evaluate_fast(O.SeIsoKernel(0.4, 0.0), Z, X, y, 0.1)
    for rank, l, dls2, grad, l0 in results:
        assert abs(l - ref["l"]) < 1e-9 * abs(ref["l"])
        assert abs(l0 - ref["l"]) < 1e-9 * abs(ref["l"])
        assert abs(dls2 - ref["dl_dsigma2"]) < 1e-7 * abs(ref["dl_dsigma2"])
        assert relinf(grad, ref["grad"]) < 1e-6
    # every rank returns the same numbers (replicated m x m work on identical reduced buffers)
    assert results[0][1] == results[1][1]
    assert np.array_equal(results[0][3], results[1][3])


def test_model_file_text_standardisation_and_roundtrip(tmp_path):
    """Host-side data formats of the reference's tool (bin/ocaml_gpr.ml:148-201, :254-265, :203-232): sample
    text, the standardisation statistics exactly as the reference computes them, the model archive."""
    from types import SimpleNamespace

    from gpr_amd import cov_se_fat, model_file
    s = model_file.read_samples("1,2,3\n4,5,6\n\n7,8,10\n")
    assert s.shape == (3, 3)
    with pytest.raises(ValueError, match="incompatible dimension"):
        model_file.read_samples("1,2\n3\n")
    with pytest.raises(ValueError, match="converting sample"):
        model_file.read_samples("1,x\n")
    with pytest.raises(ValueError, match="no data"):
        model_file.read_samples("\n\n")
    x, y = model_file.read_training_samples("1,2,3\n4,5,6\n7,8,10\n")
    assert x.shape == (2, 3) and np.array_equal(y, [3.0, 6.0, 10.0])
    xs, mean, sd = model_file.standardize_inputs(x)
    assert np.allclose(mean, [4.0, 5.0]) and np.allclose(sd, np.sqrt(18.0))     # sqrt of the SUM of squares
    assert np.allclose(model_file.apply_standardization(x, mean, sd), xs)
    rng = np.random.default_rng(0)
    params = model_file.default_params(5, 7, amplitude=2.0, dim_red=3, log_het_sked=-5.0, multiscale=True, rng=rng)
    assert params.d == 3 and params.tproj.shape == (5, 3) and np.max(np.abs(params.tproj)) <= 0.2
    assert abs(params.log_sf2 - 2 * np.log(2.0)) < 1e-15 and np.all(params.log_hetero_skedasticity == -5.0)
    assert params.log_multiscales_m05.shape == (3, 7) and not params.log_multiscales_m05.any()
    u = np.triu(rng.normal(size=(7, 7)))
    model = SimpleNamespace(sigma2=0.3, target_mean=1.5, input_means=mean, input_stddevs=sd,
                            kernel=cov_se_fat.Kernel.create(params),
                            inducing_points=np.asfortranarray(rng.normal(size=(3, 7))),
                            coeffs=rng.normal(size=7), co_variance_coeffs=(u, 2 * u))
    path = tmp_path / "m.npz"
    model_file.save_model(path, model)
    back = model_file.load_model(path)
    assert back.sigma2 == 0.3 and back.target_mean == 1.5
    assert np.array_equal(back.inducing_points, model.inducing_points) and np.array_equal(back.coeffs, model.coeffs)
    assert np.array_equal(back.kernel.params.tproj, params.tproj)
    assert np.array_equal(back.co_variance_coeffs[1], 2 * u)
    assert model_file.format_predictions([1.0, 2.5]) == "1.000000\n2.500000\n"
    assert model_file.format_predictions([1.0], [0.25]) == "1.000000,0.250000\n"


def test_exp_fast_within_one_ulp_of_libm(tmp_path):
    """gpr_amd/csrc/exp_fast.h (the exponential of the covariance and gradient kernels) compiled for the host: the
    same IEEE fma sequence the device executes, against libm over the argument range the kernels produce."""
    import subprocess
    exe = tmp_path / "exp_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "cpp", "exp_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "special cases bad=0" in out.stdout


def test_inducing_choice_and_default_kernels_follow_the_reference():
    """Inducing.choose_n_first_inputs / choose_n_random_inputs (lib/fitc_gp.ml:45-92: partial Fisher-Yates over the
    input indexes, check_n_inducing), Inputs.create_default_kernel (:128-130) with the two specs' default parameters
    (lib/cov_se_iso.ml:122-123, lib/cov_se_fat.ml:191-213) -- host logic only, no device."""
    from gpr_amd import fitc_gp
    rng = np.random.default_rng(0)
    X = np.asfortranarray(rng.normal(size=(4, 50)) + 3.0)
    GP = fitc_gp.Make_deriv(cov_se_iso)
    E = GP.FITC.Eval
    k = E.Inputs.create_default_kernel(X, n_inducing=7)
    assert k.get_params().log_ell == 0.0 and k.get_params().log_sf2 == 0.0
    pts = E.Inducing.choose_n_first_inputs(k, X, n_inducing=7)  # Spec.Inducing.t (the points), as in the reference
    assert np.array_equal(pts, X[:, :7])
    assert E.Inducing.get_points(E.Inducing.calc(k, pts)) is pts
    # the same partial shuffle, restated: step i swaps position i with a draw below n - i
    g = np.random.default_rng(11)
    idx = list(range(50))
    for i in range(9):
        j = int(g.integers(50 - i))
        idx[j], idx[i] = idx[i], idx[j]
    pts = E.Inducing.choose_n_random_inputs(k, X, n_inducing=9, rnd_state=11)
    assert np.array_equal(pts, X[:, idx[:9]])
    assert len(set(idx[:9])) == 9
    with pytest.raises(ValueError, match=r"check_n_inducing: violating 1 <= n_inducing \(51\) <= n_inputs \(50\)"):
        E.Inducing.choose_n_random_inputs(k, X, n_inducing=51)
    with pytest.raises(ValueError, match="check_n_inducing"):
        E.Inducing.choose_n_first_inputs(k, X[:, :0], n_inducing=0)
    # Cov_se_fat defaults: d = min(big_dim, 10), tproj rows scaled by n / (big_dim * row sum), hetero -5, multiscales 0
    GPf = fitc_gp.Make_deriv(cov_se_fat)
    Xb = np.asfortranarray(rng.uniform(0.5, 1.5, size=(12, 40)))
    kf = GPf.FITC.Eval.Inputs.create_default_kernel(Xb, n_inducing=6, rng=4)
    pf = kf.get_params()
    assert pf.d == 10 and pf.tproj.shape == (12, 10) and -1.0 <= pf.log_sf2 < 1.0
    bound = (40.0 / 12.0) / Xb.sum(1)
    assert np.all(np.abs(pf.tproj) <= bound[:, None] * (1 + 1e-15))
    assert np.array_equal(pf.log_hetero_skedasticity, np.full(6, -5.0)) and pf.log_multiscales_m05.shape == (10, 6)
    assert not np.any(pf.log_multiscales_m05)
    ptsf = GPf.FITC.Eval.Inducing.choose_n_first_inputs(kf, Xb, n_inducing=6)
    assert np.allclose(ptsf, pf.tproj.T @ Xb[:, :6], rtol=0, atol=1e-14)
