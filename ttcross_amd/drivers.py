"""Host-side mirror of the reference's command-line drivers (test_crs_ising.f90, test_crs_stdnorm.f90,
test_crs_mvn.f90): same positional arguments, same parameter set-up, same report lines; the sweep itself
runs on the GPU through ttcross_amd.engine.  Usage:

    python -m ttcross_amd.drivers ising KIND INDEX N RANK PIV [NGROUPS]
    python -m ttcross_amd.drivers stdnorm D N RANK PIV [NGROUPS]
    python -m ttcross_amd.drivers mvn D N RANK PIV [NGROUPS]
"""
import math
import sys

import numpy as np

from .engine import TTX_FUN_ISING, TTX_FUN_MVN, TTX_FUN_STDNORM, TTCross

EPS = 2.220446049250313e-16
TPI = 6.283185307179586476925286766559

# Ising integrals C_m, D_m, E_m (Bailey, Borwein & Crandall 2006) -- the table of test_crs_ising.f90:71-100
ISING_TRU = {
    ("c", 2): 1.0, ("c", 3): 0.78130241289648629687, ("c", 4): 0.70119986017642999982,
    ("c", 5): 0.66575980019993742832, ("c", 6): 0.64863420903100707526, ("c", 8): 0.63548402675916322614,
    ("c", 16): 0.63050394617323726351, ("c", 32): 0.63047350420733980638, ("c", 64): 0.63047350337438679649,
    ("c", 128): 0.63047350337438679612, ("c", 256): 0.63047350337438679612, ("c", 512): 0.63047350337438679612,
    ("c", 1024): 0.63047350337438679612,
    ("d", 2): 1.0 / 3, ("d", 5): 0.0024846057623403154800, ("d", 6): 0.00048914170018803477510,
    ("e", 5): 0.0034936537117295217407, ("e", 6): 0.00068783287182640943700,
}


def lgwt(n):
    """Gauss-Legendre nodes/weights on [-1,1] (lib/quad.f90:97-131)."""
    x = np.zeros(n)
    w = np.zeros(n)
    small = 5 * EPS
    for i in range(1, (n + 1) // 2 + 1):
        z = math.cos((TPI * (4 * i - 1)) / (8 * n + 4))
        while True:
            p1, p2 = 1.0, 0.0
            for j in range(1, n + 1):
                p3, p2 = p2, p1
                p1 = ((2 * j - 1) * z * p2 - (j - 1) * p3) / j
            pp = n * (z * p1 - p2) / (z * z - 1)
            z1 = z
            z = z1 - p1 / pp
            if abs(z - z1) <= small:
                break
        x[i - 1], x[n - i] = -z, z
        w[i - 1] = w[n - i] = 2.0 / ((1 - z * z) * pp * pp)
    return x, w


def ising_setup(kind, m, n):
    """test_crs_ising.f90:40,60-69,102-144 -> dict(n, par, quad, tru, acc, rescale)."""
    kind = kind.lower()
    if n % 2 == 0:
        n += 1
    x, w = lgwt(n)
    par = np.zeros(2 * n + 1)
    par[2 * n] = {"c": 1.0, "d": 2.0, "e": 3.0}[kind]
    par[n:2 * n] = 0.5 * w
    par[:n] = (x + 1.0) / 2
    rescale = kind in "de" and m >= 10
    val = float(n // 2)
    par[n:2 * n] = (5.0 * val if rescale else val) * par[n:2 * n]
    d = m - 1
    return dict(n=[n] * d, par=par, quad=[np.full(n, 1.0 / val)] * d, tru=ISING_TRU.get((kind, m)), acc=500 * EPS,
                rescale=rescale, fun_id=TTX_FUN_ISING, aux=None)


def mvn_init(d, r=0.0, T=1.0):
    """lib/mvn_pdf.f90:21-60: mean, inverse covariance and determinant of the driver's test distribution.
    Sigma = s2*((1-c) I + c 11') has the closed-form inverse/determinant used here (host set-up only)."""
    sigma, corr = 0.4, 0.5
    mu = np.full(d, math.log(100.0) + (r - 0.5 * sigma ** 2) * T)
    cov = (np.full((d, d), sigma * corr * sigma) + np.diag(np.full(d, sigma * sigma - sigma * corr * sigma))) * T
    inv = np.linalg.inv(cov)
    det = np.linalg.det(cov)
    return np.concatenate([mu, inv.ravel(order="F"), [det]])


def box_setup(kind, d, n):
    """test_crs_stdnorm.f90:70-112 / test_crs_mvn.f90:72-118."""
    if n % 2 == 0:
        n += 1
    x, w = lgwt(n)
    if kind == "stdnorm":
        a, b, acc, tru = -10.0, 10.0, 5 * EPS, math.sqrt(3.141592653589793238) ** d
    else:
        a, b, acc, tru = float(np.float32(0.525170)), float(np.float32(8.525170)), 500 * EPS, 1.0
    par = np.zeros(2 * n)
    par[:n] = 0.5 * ((b - a) * x + (a + b))
    par[n:] = (0.5 * (b - a)) * w
    return dict(n=[n] * d, par=par, quad=[par[n:].copy()] * d, tru=tru, acc=acc, rescale=False,
                fun_id=TTX_FUN_STDNORM if kind == "stdnorm" else TTX_FUN_MVN, aux=mvn_init(d) if kind == "mvn" else None)


def run_driver(argv, device=0, verbose=True):
    drv = argv[0]
    if drv == "ising":
        kind, m, n, r, piv = argv[1], int(argv[2]), int(argv[3]), int(argv[4]), int(argv[5])
        ng = int(argv[6]) if len(argv) > 6 else 1
        s = ising_setup(kind, m, n)
    else:
        m, n, r, piv = int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4])
        ng = int(argv[5]) if len(argv) > 5 else 1
        s = box_setup(drv, m, n)
    tt = TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], quad=s["quad"], tru=s["tru"],
                 aux=s["aux"], nproc=ng, device=device, verbose=verbose)
    tt.run()
    val = tt.quad(s["quad"])
    if verbose:
        print("...with%12d evaluations completed in %12.4E sec." % (tt.neval, tt.seconds))
        print("computed value: %.16e%s" % (val, "  / 5**(m-1)" if s["rescale"] else ""))
        if s["tru"]:
            print("analytic value: %.16e" % s["tru"])
            print("correct digits:%7.2f" % (-math.log10(abs(1.0 - val / s["tru"]))))
    return tt, val, s


def chf_weights(par, n, d, nfreq=32, upper=300.0):
    """The complex rank-1 quadrature weights of test_crs_chf.f90:153-168: for frequency k the weight of node p in every
    mode is w(p) * exp(i * omega_k * exp(x(p)) / d), omega_k = k*pi/(upper - 0).  Returns an (nfreq, d*n) array."""
    x, w = par[:n], par[n:2 * n]
    return np.array([np.tile(w * np.exp(1j * (k * math.pi / upper) * np.exp(x) / d), d) for k in range(nfreq)])


def cos_approximate(xs, phis, lower_bound, upper_bound, n_terms=None):
    """lib/cos_approx.f90: COS-method density at the points xs from characteristic-function values phis[k] = phi(k pi/(b-a)):
    f(x) ~ sum' 2/(b-a) Re(phi_k exp(-i w_k a)) cos(w_k (x - a)), first term halved (test_crs_pdf.f90:190)."""
    phis = np.asarray(phis, dtype=np.complex128)
    n = len(phis) if n_terms is None else n_terms
    if n > len(phis):
        raise ValueError("n_terms exceeds the size of phis")
    w = np.arange(n) * (math.pi / (upper_bound - lower_bound))
    c = 2.0 / (upper_bound - lower_bound) * np.real(phis[:n] * np.exp(-1j * w * lower_bound))
    c[0] /= 2.0
    return np.cos(np.outer(np.asarray(xs, dtype=np.float64) - lower_bound, w)) @ c


def run_pdf(argv, device=0, verbose=True, n_pts=200, upper=300.0):
    """test_crs_pdf.f90: the chf pipeline followed by the COS-method density of the basket average on linspace(0, upper)."""
    tt, vals, s = run_chf(argv, device=device, verbose=False)
    xs = np.linspace(0.0, upper, n_pts)
    pdf = cos_approximate(xs, vals, 0.0, upper, 32)
    if verbose:
        for x, f in zip(xs, pdf):
            print("%.16e %.16e" % (x, f))
    return tt, xs, pdf


def run_chf(argv, device=0, verbose=True):
    """test_crs_chf.f90:104-168: TT-cross of the multivariate-normal density WITHOUT a quadrature argument, then the
    characteristic function of the basket average at 32 frequencies as complex rank-1 quadratures of the resident
    train (ztt_quad, lib/dmrgg.f90:1418), all 32 in ONE batched device call."""
    d, n, r, piv = int(argv[0]), int(argv[1]), int(argv[2]), int(argv[3])
    ng = int(argv[4]) if len(argv) > 4 else 1
    s = box_setup("mvn", d, n)
    n = s["n"][0]
    tt = TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=piv, accuracy=s["acc"], aux=s["aux"], nproc=ng, device=device, verbose=verbose)
    tt.run()
    vals = tt.zquad(chf_weights(s["par"], n, d))
    if verbose:
        print("...with%12d evaluations completed in %12.4E sec." % (tt.neval, tt.seconds))
        for k, v in enumerate(vals):
            print("computed value: %3d %.16e %.16e" % (k, v.real, v.imag))
    return tt, vals, s


if __name__ == "__main__":
    if sys.argv[1] == "chf":
        run_chf(sys.argv[2:])
    elif sys.argv[1] == "pdf":
        run_pdf(sys.argv[2:])
    else:
        run_driver(sys.argv[1:])
