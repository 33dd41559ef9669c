"""GPU tests of the drop-in boundary: the user callback `fun` evaluated on the host, the Fortran
modules with the reference's names, and -- where oracle/_ref holds them -- the reference's OWN drivers
(test_crs_*.f90 compiled unchanged in the build container) linked against the engine."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from conftest import ROOT, fortran_exe
from golden_util import GOLDEN, parse_log
from ttcross_amd import drivers as D
from ttcross_amd import engine as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def userfun():
    """tests/userfun.c -> shared object; returns (CDLL, address of ttx_test_userfun)."""
    bdir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libuserfun.so")
    src = os.path.join(ROOT, "tests", "userfun.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", src, "-o", so, "-lm"], check=True)
    lib = ctypes.CDLL(so)
    return lib, ctypes.cast(lib.ttx_test_userfun, ctypes.c_void_p).value


def _user_setup(d, n):
    x, w = D.lgwt(n)
    par = np.concatenate([0.5 * (x + 1.0), 0.5 * w])          # nodes and weights on [0,1] (test_crs_box.inc with a=0, b=1)
    return dict(n=[n] * d, par=par, quad=[par[n:].copy()] * d, acc=500 * D.EPS)


@pytest.mark.parametrize("d,n,r,piv,nproc", [(5, 17, 10, 2, 1), (6, 13, 8, 1, 3), (4, 9, 6, 0, 1), (8, 11, 7, 3, 2),
                                             # full pivoting with the user's fun (lib/dmrgg.f90:341-408): one superblock column per launch pair
                                             (4, 7, 5, -1, 1), (5, 5, 4, -1, 2)])
def test_host_callback_bit_exact_vs_oracle(userfun, d, n, r, piv, nproc):
    """An integrand that is not built in: the engine asks the host for every fiber (ttx_set_integrand_host), the oracle
    calls the same C function -- tapes, evaluation counts, per-sweep values, cores and integral must be identical."""
    _, addr = userfun
    s = _user_setup(d, n)
    tt = E.TTCross(s["n"], E.TTX_FUN_HOST, [], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], nproc=nproc)
    tt.set_integrand_host(addr, s["par"]).run()
    oo = O.dmrgg(s["n"], 4, s["par"], r, piv=piv, accuracy=s["acc"], quad=s["quad"], nproc=nproc, user=addr)
    assert np.array_equal(tt.tapes()[:, 1:tt.d], oo["tapes"][:, 1:tt.d])
    assert [a["neval"] for a in tt.sweeps()] == [b["neval"] for b in oo["sweeps"]]
    assert [a["val"] for a in tt.sweeps()] == [b["val"] for b in oo["sweeps"]]
    assert [a["amax"] for a in tt.sweeps()] == [b["amax"] for b in oo["sweeps"]]
    assert tt.neval == oo["neval"] and tt.host_calls >= tt.neval      # groups evaluate the initial samples redundantly
    assert np.array_equal(tt.ranks(), oo["r"])
    assert all(np.array_equal(tt.core(k), oo["cores"][k - 1]) for k in range(1, d + 1))
    assert tt.quad(s["quad"]) == oo["value"]


@pytest.mark.parametrize("name", ["ttx_test_userfun_nan", "ttx_test_userfun_partnan"])
@pytest.mark.parametrize("piv,nproc", [(2, 1), (1, 3), (-1, 1), (0, 2)])
def test_nan_integrand_neither_faults_nor_hangs(userfun, name, piv, nproc):
    """An integrand that returns NaN (everywhere / on part of the domain): no arg-max comparison succeeds, and an index left at its
    start value would address memory far outside the tables (found with an mvn normalisation that underflowed to 0: a GPU fault).
    The searches now fall back to the first position, as the reference's idamax does; the run must end with finite or NaN numbers,
    every pivot inside its ranges."""
    import ctypes
    lib, _ = userfun
    addr = ctypes.cast(getattr(lib, name), ctypes.c_void_p).value
    d, n, r = 5, 5, 4
    x, w = D.lgwt(n)
    par = np.concatenate([0.5 * (x + 1.0), 0.5 * w])
    quad = [par[n:].copy()] * d
    tt = E.TTCross([n] * d, E.TTX_FUN_HOST, [], r, pivoting=piv, accuracy=500 * D.EPS, quad=quad, nproc=nproc)
    tt.set_integrand_host(addr, par).run()
    tp = tt.tapes()
    assert tp.shape[0] >= 1
    act = tp[:, 1:tt.d, :]
    assert ((act == -1) | ((act >= 1) & (act <= max(n, r + 1)))).all()
    assert all(1 <= rk <= r for rk in tt.ranks())
    tt.close()


NAN_CASES = [("c_node", 2, 1, {}), ("c_node", 2, 3, {}), ("c_node", 1, 2, {"TTX_SWEEP": "fused"}), ("c_node", 2, 2, {"TTX_SWEEP": "chain"}),
             ("c_node", 0, 1, {}), ("d_node", 2, 1, {}), ("d_node", 3, 2, {}), ("e_node", 1, 1, {"TTX_DE_TEAM_UNITS": "1000000"}),
             ("d_weight", 2, 2, {"TTX_DE_CUT": "0"}), ("d_weight", 3, 1, {"TTX_DE_LANE": "1"}),
             ("d_weight", 2, 1, {}), ("d_weight", 2, 2, {"TTX_DE_CUT": "0", "TTX_DE_TEAM_UNITS": "1000000"}),
             ("d_weight", 1, 1, {"TTX_DE_CUT": "0", "TTX_DE_TEAM_UNITS": "0", "TTX_DE_TEAM6_UNITS": "1000000"}),
             ("d_weight", 2, 1, {"TTX_DE_CUT": "0", "TTX_DE_TEAM": "0"}), ("d_weight_fast", 2, 2, {}), ("d_weight_fast", 0, 1, {}), ("d_node", -1, 1, {}),
             ("mvn_inf", 2, 1, {}), ("mvn_inf", 1, 2, {}), ("mvn_inf_fast", 2, 2, {}), ("mvn_inf", -1, 1, {})]


@pytest.mark.parametrize("case,piv,nproc,env", NAN_CASES, ids=[f"{c[0]}_p{c[1]}_g{c[2]}" + "".join("_" + v for v in c[3].values()) for c in NAN_CASES])
def test_nan_through_builtin_device_integrands(case, piv, nproc, env):
    """NaN / Inf through the BUILT-IN integrands (the host-callback test above drives only the generic chain kernels): a NaN node
    or weight of the Ising integrands on the cluster, fused and chain paths and through the D/E division kernels (wave per unit,
    14- and 6-wave teams, lane per element), an infinite entry of the mvn inverse covariance, the table evaluators of
    TTX_ARITH=fast.  Every arg-max falls back to the first position when nothing compares (idamax), so every pivot must stay in
    range and the run must end.  One process per case: a GPU memory fault (the round-2 finding) fails the case, not the runner."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "nan_worker.py"), case, str(piv), str(nproc)], capture_output=True, text=True, env=e, timeout=300)
    blob = p.stdout + p.stderr
    assert "Memory access fault" not in blob and "core dump" not in blob.lower(), blob[-3000:]
    assert p.returncode == 0 and " OK" in p.stdout, blob[-3000:]


def test_mvn_normalisation_that_underflows_is_refused():
    """det of the covariance underflows to 0 at d = 300 (0.08^300): sqrt((2 pi)^d det) = 0 would make every value inf/NaN."""
    import oracle_lib as O
    s = D.box_setup("mvn", 300, 2)
    s["aux"] = O.mvn_init(300)
    assert s["aux"][300 + 300 * 300] == 0.0
    with pytest.raises(E.TTXError, match="normalisation"):
        E.TTCross(s["n"], s["fun_id"], s["par"], 2, pivoting=1, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"])


def test_host_callback_accchk(userfun):
    _, addr = userfun
    s = _user_setup(5, 17)
    tt = E.TTCross(s["n"], E.TTX_FUN_HOST, [], 10, pivoting=2, accuracy=s["acc"], quad=s["quad"])
    tt.set_integrand_host(addr, s["par"]).run()
    oo = O.dmrgg(s["n"], 4, s["par"], 10, piv=2, accuracy=s["acc"], quad=s["quad"], user=addr, accchk=3000)
    got = tt.accchk(3000)
    for k in ("einf", "efro", "ainf", "afro"):
        assert got[k] == oo["accchk"][k], k
    assert np.array_equal(got["pivot"], oo["accchk"]["pivot"])


def test_host_callbacks_of_two_engines_in_two_threads(userfun):
    """Two engines driven from two host threads share the process-wide worker pool of the host integrand: their batches take
    turns on it (ctypes releases the GIL inside ttx_run, so the two runs really overlap).  Both must equal the oracle."""
    import threading
    _, addr = userfun
    cases = [(5, 17, 10, 2), (6, 13, 8, 1)]
    out = [None, None]

    def job(i):
        d, n, r, piv = cases[i]
        s = _user_setup(d, n)
        tt = E.TTCross(s["n"], E.TTX_FUN_HOST, [], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"])
        tt.set_integrand_host(addr, s["par"]).run()
        out[i] = (tt.neval, tt.quad(s["quad"]), [tt.core(k) for k in range(1, d + 1)])
        tt.close()

    th = [threading.Thread(target=job, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i, (d, n, r, piv) in enumerate(cases):
        s = _user_setup(d, n)
        oo = O.dmrgg(s["n"], 4, s["par"], r, piv=piv, accuracy=s["acc"], quad=s["quad"], user=addr)
        assert out[i] is not None and out[i][0] == oo["neval"] and out[i][1] == oo["value"]
        assert all(np.array_equal(a, b) for a, b in zip(out[i][2], oo["cores"]))


def test_host_callback_needs_the_function():
    s = _user_setup(4, 9)
    tt = E.TTCross(s["n"], E.TTX_FUN_HOST, [], 6, pivoting=1, accuracy=s["acc"])
    with pytest.raises(E.TTXError, match="ttx_set_integrand_host"):
        tt.run()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_fortran_driver_with_user_callback(userfun, mode, monkeypatch):
    """ttcross_amd/fortran/test_crs_user.f90: `call dtt_dmrgg(tt, fun, ...)` with a callback of the driver's own --
    mode 1 without `par` (the callback reads module data, as calc_coefficient of test_crs_coscoeff.f90:186 does),
    mode 2 also without `maxrank` (lib/dmrgg.f90:19-21: both are optional).  Per-sweep lines against the oracle
    running the C twin of the callback."""
    _, addr = userfun
    d, n, r, piv = 5, 17, 9, 2
    exe = fortran_exe("test_crs_user")
    if mode == 2:
        monkeypatch.setenv("TTX_MAXRANK_DEFAULT", "12")
        r = 12
    p = subprocess.run([exe, str(d), str(n), str(r), str(piv), str(mode)], capture_output=True, text=True, timeout=300, env=dict(os.environ))
    assert p.returncode == 0, p.stdout + p.stderr
    rows, val, nev = parse_log(p.stdout)
    s = _user_setup(d, n)
    oo = O.dmrgg(s["n"], 4, s["par"], r, piv=piv, accuracy=s["acc"], quad=s["quad"], user=addr)
    assert len(rows) == len(oo["sweeps"]) and nev == oo["neval"]
    for a, b in zip(rows, oo["sweeps"]):
        assert a["neval"] == b["neval"] and abs(a["val"] - b["val"]) <= 1e-13 * abs(b["val"])
    assert abs(val - oo["value"]) <= 1e-15 * abs(val)


def _h5(tool, *args):
    exe = os.path.join("/opt/conda/bin", tool)
    if not os.path.exists(exe):
        pytest.skip(f"{tool} not on this box")
    return subprocess.run([exe] + list(args), capture_output=True, text=True, check=True).stdout


def test_hdf5_layout_of_the_reference(tmp_path):
    """N3: save_dtt_to_hdf5 (lib/utils.f90:8-57) -- group TT, datasets modes / ranks (native int) and core_k with the Fortran
    shape (r(k-1), n(k), r(k)), i.e. the C dataspace reversed.  Checked with the HDF5 command-line tools and by reading
    the file back into an engine."""
    s = D.ising_setup("c", 6, 17)
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], 7, pivoting=2, accuracy=s["acc"], quad=s["quad"]).run()
    f = str(tmp_path / "tt.h5")
    try:
        tt.write_hdf5(f)
    except E.TTXError as e:
        if "libhdf5" in str(e):
            pytest.skip("no libhdf5 on this box")
        raise
    r, n = tt.ranks(), s["n"]
    ls = _h5("h5ls", "-r", f)
    assert "/TT" in ls and "/TT/modes" in ls and "/TT/ranks" in ls
    for k in range(tt.d):
        line = [ln for ln in ls.splitlines() if ln.split()[0] == f"/TT/core_{k}"][0]
        assert "{%d, %d, %d}" % (r[k + 1], n[k], r[k]) in line, line
    dump = _h5("h5dump", "-d", "/TT/ranks", f)
    assert "H5T_STD_I32LE" in dump
    vals = [int(v) for v in dump.split("DATA {")[1].split("}")[0].replace("(0):", "").replace(",", " ").split()]
    assert vals == list(r)
    core2 = _h5("h5dump", "-d", "/TT/core_2", "-y", "-w", "1", "-m", "%.17g", f).split("DATA {")[1].split("}")[0]
    got = np.array([float(v.strip(", ")) for v in core2.split() if v.strip(", ")])
    assert np.array_equal(got, tt.core(3).ravel(order="F"))              # file bytes = Fortran column-major core
    t2 = E.TTCross.read_hdf5(f)
    assert np.array_equal(t2.ranks(), r)
    assert all(np.array_equal(t2.core(k), tt.core(k)) for k in range(1, tt.d + 1))
    assert t2.quad(s["quad"]) == tt.quad(s["quad"])


def test_cos_approx_pdf_from_the_characteristic_function():
    """N2 tail: the COS-method density (lib/cos_approx.f90) from the 32 complex quadratures of the chf pipeline
    (test_crs_pdf.f90:153-190).  The density of the basket average must be non-negative up to the truncation ripple,
    integrate to ~1 over [0, 300] and agree with the direct cosine sum."""
    tt, xs, pdf = D.run_pdf(["4", "17", "8", "2"], verbose=False)
    vals = tt.zquad(D.chf_weights(D.box_setup("mvn", 4, 17)["par"], 17, 4))
    k = np.arange(32)
    w = k * np.pi / 300.0
    c = 2.0 / 300.0 * vals.real
    c[0] /= 2
    assert np.allclose(pdf, np.cos(np.outer(xs, w)) @ c, rtol=0, atol=1e-15)
    assert abs(np.trapezoid(pdf, xs) - vals[0].real) < 5e-3 * abs(vals[0].real)


def test_fortran_tt_generics(tmp_path):
    """Host-side generics of the drop-in tt_lib and mat_lib; `b = a` deep copy with disjoint ownership."""
    h5 = str(tmp_path / "gen.h5")
    p = subprocess.run([fortran_exe("test_tt_generics"), h5], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    out = {ln.split()[0]: ln.split()[1:] for ln in p.stdout.splitlines() if ln.strip()}
    n = [2, 3, 4, 5]
    tot = np.prod([0.5 * (k + 1) * n[k] for k in range(4)])          # every core k is the constant k/2: rank-1 train
    elem = np.prod([0.5 * (k + 1) for k in range(4)])
    assert [float(v) for v in out["copy_independent"]] == [1.0, 7.0]
    assert float(out["numel"][0]) == 120.0 and int(out["memory"][0]) == 14
    assert abs(float(out["sumall"][0]) - tot) < 1e-9
    assert [int(v) for v in out["plus_ranks"]] == [1, 2, 2, 2, 1]
    assert abs(float(out["plus_sumall"][0]) - 2 * tot) < 1e-9 and abs(float(out["mul_sumall"][0]) - 3 * tot) < 1e-9
    assert abs(float(out["tijk"][0]) - elem) < 1e-12 and abs(float(out["elem"][0]) - elem) < 1e-12
    assert abs(float(out["value0"][0]) - elem) < 1e-12
    nrm = elem * np.sqrt(120.0)
    assert abs(float(out["norm"][0]) - nrm) < 1e-5 and abs(float(out["lognrm"][0]) - np.log10(nrm)) < 1e-5
    assert abs(float(out["dot"][0]) - nrm ** 2) < 1e-4
    assert [int(v) for v in out["svd_ranks"]] == [1, 1, 1, 1, 1] and abs(float(out["svd_norm"][0]) - 2 * nrm) < 1e-5
    assert float(out["zeros_sumall"][0]) == 0.0 and abs(float(out["copy_sumall"][0]) - tot) < 1e-9
    assert float(out["matinv_err"][0]) < 1e-14 and float(out["svd_err"][0]) < 1e-13 and int(out["chop"][0]) == 2
    phis = np.array([1.0, 0.5 - 0.25j, -0.125 + 0.0625j, 0.03 + 0.01j])
    want = D.cos_approximate([0.5, 1.25, 2.75], phis, 0.25, 3.0, 4)
    assert np.allclose([float(v) for v in out["cos_array"]], want, rtol=1e-14, atol=1e-16)
    assert abs(float(out["cos_point"][0]) - want[1]) <= 1e-14 * abs(want[1])
    if "hdf5_written" in out and os.path.exists("/opt/conda/bin/h5ls"):
        ls = _h5("h5ls", "-r", h5)
        assert "/TT/core_3" in ls and "{1, 5, 1}" in [ln for ln in ls.splitlines() if ln.startswith("/TT/core_3")][0]
    assert "dealloc_ok" in out


REF_DROPIN = [("ising", "ising_C_6_33_20_2"), ("ising", "ising_D_6_33_12_2"), ("ising", "ising_C_16_33_24_0"),
              ("stdnorm", "stdnorm_4_33_10_2"), ("mvn", "mvn_6_33_12_2")]


@pytest.mark.parametrize("drv,name", REF_DROPIN, ids=[c[1] for c in REF_DROPIN])
def test_reference_drivers_unchanged_on_the_engine(drv, name):
    """The reference's OWN test_crs_{ising,stdnorm,mvn}.f90, compiled UNCHANGED in the build container against the drop-in
    modules and linked with libttx.so (oracle/Makefile target `dropin`, binaries in oracle/_ref/ -- the sources stay
    behind), run on the GPU against the golden logs of the genuine reference."""
    exe = os.path.join(ROOT, "oracle", "_ref", f"dropin_test_crs_{drv}")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin_test_crs_* not built (needs /root/reference + amdflang: make -C oracle dropin)")
    argv = name.split("_")[1:]
    p = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    g_rows, g_val, g_nev = parse_log(open(os.path.join(GOLDEN, name + ".txt")).read())
    o_rows, o_val, o_nev = parse_log(p.stdout)
    assert len(g_rows) == len(o_rows)
    need = {"ising": len(g_rows), "stdnorm": 2, "mvn": 8}[drv]        # as the oracle against the same logs (test_oracle_golden.py)
    k = 0
    for a, b in zip(g_rows, o_rows):
        if a["erank"] == b["erank"] and a["neval"] == b["neval"] and abs(a["val"] - b["val"]) <= 2e-13 * abs(a["val"]):
            k += 1
        else:
            break
    assert k >= need, f"only {k} leading sweeps match the reference (need {need})"
    assert abs(g_val - o_val) <= {"ising": 1e-14, "stdnorm": 1e-13, "mvn": 1e-3}[drv] * abs(g_val)
