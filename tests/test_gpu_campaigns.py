"""Randomised and repeated-run campaigns on the GPU (the scripts behind the figures DESIGN.md section 2 quotes).

Inside `pytest -m gpu` they run a SMALL sample (seconds); a campaign is the same code with more cases:

    TTX_FUZZ_CASES=2000 TTX_FUZZ_SEED=7 python -m pytest tests/test_gpu_campaigns.py -m gpu -q -k fuzz
    TTX_SOAK_RUNS=500 python -m pytest tests/test_gpu_campaigns.py -m gpu -q -k soak
    TTX_MPFUZZ_CASES=40 python -m pytest tests/test_gpu_campaigns.py -m gpu -q -k multi_process

Every case is compared with the CPU oracle bit for bit (tapes, per-sweep records, integral); a failing case prints the
arguments that reproduce it."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

import oracle_lib as O
from ttcross_amd import drivers as D
from ttcross_amd import engine as E

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_case(rng):
    kind = rng.choice(["c", "c", "c", "d", "e", "stdnorm", "mvn"])
    long_de = kind in ("d", "e") and rng.random() < 0.4     # long pair chains: several 16-column chunks per row, 8-wide division batches
    if kind in ("c", "d", "e"):
        m = int(rng.integers(3, 40 if kind == "c" else 90 if long_de else 14))
        d = m - 1
    else:
        m = d = int(rng.integers(2, 14))
    n = int(rng.choice([2, 3, 5, 9, 17] if long_de else [2, 3, 5, 9, 17, 25, 33, 41]))
    r = int(rng.integers(2, 40 if kind == "c" else 9 if long_de else 14))
    piv = int(rng.choice([-1, 0, 1, 2, 3, 4])) if r * n <= 160 else int(rng.choice([0, 1, 2, 3]))
    ng = int(rng.integers(1, min(5, d - 1) + 1)) if d > 2 else 1
    own = None
    if ng > 1 and rng.random() < 0.3:        # caller-supplied uneven bond groups instead of share()
        cuts = sorted(rng.choice(np.arange(2, d), size=ng - 1, replace=False).tolist())
        own = [1] + [int(c) for c in cuts] + [d]
    acc = None
    if rng.random() < 0.25:                  # the stopping rule at other thresholds (-1: no accuracy rule, maxrank sweeps)
        acc = float(rng.choice([-1.0, 1e-3, 1e-6, 1e-10]))
    ragged = None
    if kind in ("c", "d", "e") and rng.random() < 0.2:      # ragged mode sizes, the first the largest
        ragged = [n] + [int(rng.integers(1, n + 1)) for _ in range(d - 1)]
    return kind, m, n, r, piv, ng, own, acc, ragged


def _setup(kind, m, n):
    if kind in ("c", "d", "e"):
        return D.ising_setup(kind, m, n)
    s = D.box_setup(kind, m, n)
    if kind == "mvn":
        s["aux"] = O.mvn_init(m)
    return s


def _ragged_ising(kind, n_list):
    """Ising-type problem with ragged mode sizes (n(1) the largest: the integrand addresses the weights at par(n(1)+ind))."""
    nmax = n_list[0]
    x, w = D.lgwt(nmax)
    par = np.zeros(2 * nmax + 1)
    par[:nmax] = (x + 1.0) / 2
    par[nmax:2 * nmax] = 0.5 * w * float(max(nmax // 2, 1))
    par[2 * nmax] = {"c": 1.0, "d": 2.0, "e": 3.0}[kind]
    quad = [np.full(nk, 1.0 / float(max(nmax // 2, 1))) for nk in n_list]
    return dict(n=list(n_list), par=par, quad=quad, fun_id=E.TTX_FUN_ISING, aux=None, acc=500 * D.EPS, tru=None)


def _compare(kind, m, n, r, piv, ng, own=None, acc=None, ragged=None):
    s = _ragged_ising(kind, ragged) if ragged else _setup(kind, m, n)
    if acc is not None:
        s["acc"] = acc
    tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=ng, mybonds=own).run()
    oo = O.dmrgg(s["n"], s["fun_id"], s["par"], r, piv=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=ng, mybonds=own)
    ok = (np.array_equal(tt.tapes()[:, 1:tt.d], oo["tapes"][:, 1:tt.d]) and
          [a["val"] for a in tt.sweeps()] == [b["val"] for b in oo["sweeps"]] and
          [a["neval"] for a in tt.sweeps()] == [b["neval"] for b in oo["sweeps"]] and
          [a["amax"] for a in tt.sweeps()] == [b["amax"] for b in oo["sweeps"]] and
          tt.quad(s["quad"]) == oo["value"])
    val = tt.quad(s["quad"])
    tt.close()
    return ok, val


def test_fuzz_random_shapes_vs_oracle():
    """Random integrand, dimension, mode size, rank, pivoting mode and number of bond groups."""
    ncases = int(os.environ.get("TTX_FUZZ_CASES", "12"))
    rng = np.random.default_rng(int(os.environ.get("TTX_FUZZ_SEED", "20261004")))
    bad = []
    for _ in range(ncases):
        case = _random_case(rng)
        ok, _ = _compare(*case)
        if not ok:
            bad.append(case)
    assert not bad, f"cases that differ from the oracle (kind, m, n, r, piv, groups): {bad}"


def test_fuzz_host_callback_vs_oracle():
    """Random shapes with the integrand evaluated on the HOST (ttx_set_integrand_host) against the oracle with the same C function."""
    import ctypes
    ncases = int(os.environ.get("TTX_FUZZ_CASES", "12")) // 2 + 1
    rng = np.random.default_rng(int(os.environ.get("TTX_FUZZ_SEED", "20261004")) + 2)
    bdir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libuserfun.so")
    if not os.path.exists(so):
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", os.path.join(ROOT, "tests", "userfun.c"), "-o", so, "-lm"], check=True)
    lib = ctypes.CDLL(so)
    addr = ctypes.cast(lib.ttx_test_userfun, ctypes.c_void_p).value
    bad = []
    for _ in range(ncases):
        d = int(rng.integers(2, 12)); n = int(rng.choice([3, 5, 9, 17, 25])); r = int(rng.integers(2, 16)); piv = int(rng.choice([0, 1, 2, 3]))
        ng = int(rng.integers(1, min(4, d - 1) + 1)) if d > 2 else 1
        x, w = D.lgwt(n)
        par = np.concatenate([0.5 * (x + 1.0), 0.5 * w])
        quad = [par[n:].copy()] * d
        tt = E.TTCross([n] * d, E.TTX_FUN_HOST, [], r, pivoting=piv, accuracy=500 * D.EPS, quad=quad, nproc=ng)
        tt.set_integrand_host(addr, par).run()
        oo = O.dmrgg([n] * d, 4, par, r, piv=piv, accuracy=500 * D.EPS, quad=quad, nproc=ng, user=addr)
        ok = (np.array_equal(tt.tapes()[:, 1:tt.d], oo["tapes"][:, 1:tt.d]) and [a["val"] for a in tt.sweeps()] == [b["val"] for b in oo["sweeps"]] and
              tt.neval == oo["neval"] and tt.quad(quad) == oo["value"])
        tt.close()
        if not ok:
            bad.append((d, n, r, piv, ng))
    assert not bad, f"host-callback cases that differ from the oracle (d, n, r, piv, groups): {bad}"


def test_soak_repeated_runs_are_identical():
    """The same engine run again and again (cluster kernel: its record tags and barrier counters carry over between
    launches) must return the identical integral and evaluation count every time."""
    runs = int(os.environ.get("TTX_SOAK_RUNS", "25"))
    for kind, m, n, r, piv, ng in [("c", 64, 51, 32, 2, 8), ("c", 20, 17, 24, 3, 3), ("d", 8, 17, 8, 2, 2), ("d", 60, 9, 6, 2, 3)]:
        s = _setup(kind, m, n)
        tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], nproc=ng)
        ref = None
        for _ in range(runs):
            tt.run()
            cur = (tt.quad(s["quad"]), tt.neval, len(tt.sweeps()))
            ref = ref or cur
            assert cur == ref, (kind, m, n, r, piv, ng, cur, ref)
        assert tt.cluster_fallbacks == 0
        tt.close()


def test_fuzz_multi_process_jobs():
    """Random jobs as 2-4 engine processes over the shared-memory transport (tests/mp_worker.py checks each against the oracle)."""
    ncases = int(os.environ.get("TTX_MPFUZZ_CASES", "3"))
    rng = np.random.default_rng(int(os.environ.get("TTX_FUZZ_SEED", "20261004")) + 1)
    for _ in range(ncases):
        world = int(rng.integers(2, 5))
        kind = str(rng.choice(["c", "c", "d", "e"]))
        m = int(rng.integers(world + 3, 24 if kind == "c" else 11))
        n = int(rng.choice([5, 9, 17, 33]))
        r = int(rng.integers(3, 20 if kind == "c" else 9))
        piv = int(rng.choice([0, 1, 2, 3]))
        ng = int(rng.integers(world, min(2 * world, m - 2) + 1))
        name = "ttx_" + uuid.uuid4().hex[:12]
        procs = []
        for rk in range(world):
            env = dict(os.environ, RANK=str(rk), WORLD_SIZE=str(world), TTX_SHM_NAME=name)
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_worker.py"), kind, str(m), str(n), str(r), str(piv), str(ng), "shm"],
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
        for p in procs:
            o, e = p.communicate(timeout=600)
            assert p.returncode == 0 and " OK" in o, f"world={world} {kind} {m} {n} {r} {piv} groups={ng}\n" + o[-1500:] + e[-1500:]


def test_fuzz_reference_driver_on_the_engine_vs_genuine_reference():
    """Random command lines for the reference's OWN test_crs_{ising,stdnorm,mvn}.f90: compiled unchanged against the drop-in modules
    and run on the GPU (oracle/_ref/dropin_test_crs_*) against the GENUINE reference run on the host at the same moment
    (oracle/_ref/test_crs_*, test infrastructure; the binaries are built in the build container and travel).  The genuine
    reference sums with MKL (and inverts the mvn covariance with LAPACK), so a near-tie may turn a later pivot: the first three
    sweeps (stdnorm: two, mvn: the first) must agree in (erank, n_evals) and to 2e-13 in the value; the number of cases that agree in EVERY sweep
    is reported."""
    from golden_util import parse_log
    exe = lambda pre, drv: os.path.join(ROOT, "oracle", "_ref", f"{pre}test_crs_{drv}")
    if not all(os.path.exists(exe(pre, drv)) for pre in ("", "dropin_") for drv in ("ising", "stdnorm", "mvn")):
        pytest.skip("oracle/_ref binaries not built (needs /root/reference + amdflang: make -C oracle ref dropin)")
    ncases = int(os.environ.get("TTX_REFFUZZ_CASES", "3"))
    rng = np.random.default_rng(int(os.environ.get("TTX_FUZZ_SEED", "20261004")) + 3)
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="4")
    full, bad = 0, []
    for _ in range(ncases):
        drv = str(rng.choice(["ising", "ising", "ising", "stdnorm", "mvn"]))
        n = int(rng.choice([9, 17, 25, 33])); r = int(rng.integers(3, 17)); piv = int(rng.integers(0, 4))
        if drv == "ising":
            kind = str(rng.choice(["C", "C", "D", "E"]))
            argv = [kind, str(int(rng.integers(3, 15 if kind == "C" else 10))), str(n), str(r), str(piv)]
        else:                                   # the exp-based integrands: the leading sweeps only (the noise floor decides near-ties)
            argv = [str(int(rng.integers(2, 9))), str(n), str(r), str(piv)]
        # Ising also as P bond groups: the reference under `mpiexec -np P` (the build with the right-going exchange re-inserted,
        # oracle/_ref/test_crs_ising_mpi) beside the drop-in with TTX_NGROUPS=P on one GPU
        groups = 1
        mpi_exe, mpiexec = exe("", "ising") + "_mpi", "/opt/conda/bin/mpiexec"
        if drv == "ising" and os.path.exists(mpi_exe) and os.path.exists(mpiexec) and rng.random() < 0.4:
            groups = int(rng.integers(2, min(5, int(argv[1]) - 2) + 1)) if int(argv[1]) - 2 >= 2 else 1
        need = {"ising": 3, "stdnorm": 2, "mvn": 1}[drv]      # mvn: equal correlations make exact ties, settled by the last bits of the inverse covariance
        argv_t = [drv] + argv + ([f"groups={groups}"] if groups > 1 else [])
        if groups > 1:
            a = subprocess.run([mpiexec, "-np", str(groups), mpi_exe] + argv, capture_output=True, text=True, env=dict(env, OMP_NUM_THREADS="1"), timeout=600)
            b = subprocess.run([exe("dropin_", drv)] + argv, capture_output=True, text=True, env=dict(env, TTX_NGROUPS=str(groups)), timeout=600)
        else:
            a = subprocess.run([exe("", drv)] + argv, capture_output=True, text=True, env=env, timeout=600)
            b = subprocess.run([exe("dropin_", drv)] + argv, capture_output=True, text=True, env=env, timeout=600)
        if a.returncode != 0 or b.returncode != 0:
            bad.append((argv_t, "exit codes", a.returncode, b.returncode)); continue
        ra, va, na = parse_log(a.stdout)
        rb, vb, nb = parse_log(b.stdout)
        k = 0
        for x, y in zip(ra, rb):
            if x["erank"] == y["erank"] and x["neval"] == y["neval"] and abs(x["val"] - y["val"]) <= 2e-13 * abs(x["val"]):
                k += 1
            else:
                break
        if k == len(ra) == len(rb) and na == nb:
            full += 1
        if k < min(need, len(ra)):
            bad.append((argv_t, f"only {k} leading sweeps agree of {len(ra)} / {len(rb)}"))
    print(f"reference-driver fuzz: {full} of {ncases} command lines agree in every sweep")
    assert not bad, f"command lines on which the drop-in departs from the genuine reference early: {bad}"


def test_fuzz_tt_lib_vs_genuine_reference():
    """dtt_ort / dtt_svd / dtt_norm / dot_product / tijk on the device against the GENUINE reference's tt_lib run live on the host
    (oracle/_ref/ref_ttops: the fixture driver tests/golden/ref_ttops.f90 linked with the reference's own modules) on random
    Ising-C trains: identical ranks after ort and after every svd, norms / dots / elements to 1e-11 of the train's scale."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_ttops")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_ttops not built (needs /root/reference + amdflang: make -C oracle ref)")
    ncases = int(os.environ.get("TTX_TTOPSFUZZ_CASES", "2"))
    rng = np.random.default_rng(int(os.environ.get("TTX_FUZZ_SEED", "20261004")) + 4)
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="4")
    bad, diverged = [], 0
    for _ in range(ncases):
        m = int(rng.integers(4, 13)); n = int(rng.choice([9, 17, 25, 33])); r = int(rng.integers(3, 25)); piv = int(rng.integers(0, 4))
        p = subprocess.run([exe, str(m), str(n), str(r), str(piv)], capture_output=True, text=True, env=env, timeout=600)
        if p.returncode != 0:
            bad.append(((m, n, r, piv), "reference exit code", p.returncode)); continue
        fx = {}
        for line in p.stdout.splitlines():
            t = line.split()
            if t:
                fx.setdefault(t[0], []).append(t[1:])
        s = D.ising_setup("c", m, n)
        mk = lambda: E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"]).run()      # no quad, as the fixture driver
        tt = mk()
        nrm0 = float(fx["norm0"][0][0])
        why = []
        if list(tt.ranks()) != [int(x) for x in fx["ranks0"][0]]:
            # the cross itself took another pivot somewhere (the reference sums with MKL: a near-tie, see the reference-driver fuzz):
            # not a statement about tt_lib -- counted, not compared
            diverged += 1
            tt.close()
            continue
        if abs(tt.norm() - nrm0) > 1e-12 * nrm0 or abs(tt.dot(tt) - float(fx["dot00"][0][0])) > 1e-12 * nrm0 ** 2:
            why.append("norm0/dot00")
        t1 = mk().ort()
        if list(t1.ranks()) != [int(x) for x in fx["ranks_ort"][0]] or abs(t1.norm() - float(fx["norm_ort"][0][0])) > 1e-11 * nrm0:
            why.append("ort")
        for case, (tol, rmax) in enumerate([(1e-4, 0), (1e-8, 0), (1e-12, 5)], start=1):
            t2 = mk().svd(tol, rmax)
            if list(t2.ranks()) != [int(x) for x in fx["ranks_svd"][case - 1][1:]]:
                why.append(f"ranks_svd{case}")
            elif abs(t2.norm() - float(fx["norm_svd"][case - 1][1])) > 1e-11 * nrm0 or abs(tt.dot(t2) - float(fx["dot_svd"][case - 1][1])) > 1e-11 * nrm0 ** 2:
                why.append(f"norm/dot_svd{case}")
            else:
                scale = nrm0 / np.sqrt(float(n) ** (m - 1))
                for k in range(1, 5):
                    ind = [(5 * k + 3 * i + i * i * k) % n + 1 for i in range(1, m)]
                    row = fx["elem"][4 * (case - 1) + k - 1]
                    # the reference multiplies the cores with MKL's dgemv: same numbers to rounding, relative to the size of the element
                    # plus a share of the train's scale (an element can be a difference of much larger terms)
                    if abs(tt.tijk(ind) - float(row[2])) > 1e-10 * abs(float(row[2])) + 1e-12 * scale or abs(t2.tijk(ind) - float(row[3])) > 1e-9 * scale + 1e-9 * abs(float(row[3])):
                        why.append(f"elem{case}.{k}")
            t2.close()
        tt.close(); t1.close()
        if why:
            bad.append(((m, n, r, piv), why))
    print(f"tt_lib fuzz: {ncases - diverged} of {ncases} trains compared ({diverged} crosses took another pivot path than the reference's)")
    assert not bad, f"cases (m, n, r, piv) that depart from the genuine reference's tt_lib: {bad}"
    assert diverged <= max(1, ncases // 5)


def test_fuzz_accchk_and_zquad_vs_genuine_reference():
    """dtt_accchk and ztt_quad against the GENUINE reference run live (oracle/_ref/ref_accchk, ref_zquad) on random Ising-C jobs.
    accchk draws its samples from the same random stream as the reference, so the worst sample and the sup norms must be THE
    SAME numbers (the value norms exactly; the error norms to 1e-6: the reference evaluates the train with MKL), provided the
    cross took the same pivots -- the cases where it did not are counted (see the reference-driver fuzz)."""
    exa, exz = (os.path.join(ROOT, "oracle", "_ref", x) for x in ("ref_accchk", "ref_zquad"))
    if not (os.path.exists(exa) and os.path.exists(exz)):
        pytest.skip("oracle/_ref/ref_accchk / ref_zquad not built (needs /root/reference + amdflang: make -C oracle ref)")
    ncases = int(os.environ.get("TTX_ACCFUZZ_CASES", "2"))
    rng = np.random.default_rng(int(os.environ.get("TTX_FUZZ_SEED", "20261004")) + 5)
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", OMP_NUM_THREADS="4")
    bad, diverged = [], 0
    for _ in range(ncases):
        m = int(rng.integers(4, 12)); n = int(rng.choice([9, 17, 25, 33])); r = int(rng.integers(3, 20)); piv = int(rng.integers(0, 4)); nlot = int(rng.integers(50, 4000))
        s = D.ising_setup("c", m, n)
        why = []
        # ---- accchk (the fixture driver passes the quadrature weights to dtt_dmrgg) ----
        p = subprocess.run([exa, str(m), str(n), str(r), str(piv), str(nlot)], capture_output=True, text=True, env=env, timeout=600)
        ref = [float(x) for l in p.stdout.splitlines() if l.startswith("accchk") for x in l.split()[1:]]
        refpiv = [int(x) for l in p.stdout.splitlines() if l.startswith("pivot") for x in l.split()[1:]]
        tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"]).run()
        g = tt.accchk(nlot)
        tt.close()
        if g["ainf"] != ref[2]:
            diverged += 1                       # another sample set would not give the identical sup norm: the crosses differ
        else:
            # the worst sample is a statement about the interpolation error only where that error is above the rounding noise
            # (a train that reproduces a small tensor exactly has its "worst" sample wherever the last bits fall)
            if (ref[0] > 1e-9 * ref[2] and list(g["pivot"]) != refpiv) or abs(g["afro"] - ref[3]) > 1e-13 * ref[3]:
                why.append("accchk values")
            if abs(g["einf"] - ref[0]) > 1e-5 * ref[0] + 1e-13 * ref[2] or abs(g["efro"] - ref[1]) > 1e-5 * ref[1] + 1e-13 * ref[3]:
                why.append("accchk errors")
        # ---- zquad (no quadrature in dtt_dmrgg, as the fork's driver) ----
        p = subprocess.run([exz, str(m), str(n), str(r), str(piv)], capture_output=True, text=True, env=env, timeout=600)
        refz = [l.split() for l in p.stdout.splitlines() if l.startswith("zquad")]
        tz = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"]).run()
        sc = float(n // 2); x = s["par"][:n]
        W = np.array([np.tile((1.0 / sc) * np.exp(1j * (k * np.pi / 300.0) * np.exp(x) / (m - 1)), m - 1) for k in range(len(refz))])
        got = tz.zquad(W)
        tz.close()
        for k in range(len(refz)):
            want = complex(float(refz[k][2]), float(refz[k][3]))
            if abs(got[k] - want) > 1e-9 * abs(want):
                why.append(f"zquad {k}"); break
        if why:
            bad.append(((m, n, r, piv, nlot), why))
    print(f"accchk / zquad fuzz: {ncases} jobs, {diverged} with another pivot path than the reference's")
    assert not bad, f"cases (m, n, r, piv, nlot) that depart from the genuine reference: {bad}"
    assert diverged <= max(1, ncases // 5)
