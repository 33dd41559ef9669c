"""ctypes wrapper of the CPU oracle (oracle/libttx_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int32, c_int64, c_uint64

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Rec(ctypes.Structure):
    _fields_ = [("it", c_int32), ("dir", c_int32), ("erank", c_double), ("neval", c_int64), ("val", c_double),
                ("amax", c_double), ("pivotmax", c_double), ("pivotmin", c_double)]


class _Problem(ctypes.Structure):
    _fields_ = [("d", c_int32), ("n", POINTER(c_int32)), ("fun_id", c_int32), ("par", POINTER(c_double)), ("npar", c_int32),
                ("aux", POINTER(c_double)), ("naux", c_int32), ("quadw", POINTER(c_double)), ("accuracy", c_double),
                ("maxrank", c_int32), ("piv", c_int32), ("tru", c_double), ("has_tru", c_int32), ("nproc", c_int32),
                ("mybonds", POINTER(c_int32)), ("verbose", c_int32), ("draws", POINTER(c_double)), ("ndraws", c_int64),
                ("user", ctypes.c_void_p)]


class _Result(ctypes.Structure):
    _fields_ = [("d", c_int32), ("nsweeps", c_int32), ("sweeps", POINTER(_Rec)), ("tapes", POINTER(c_int32)),
                ("r", POINTER(c_int32)), ("cores", POINTER(POINTER(c_double))), ("neval", c_int64), ("value", c_double),
                ("seconds", c_double), ("rngpos", c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "oracle", "libttx_oracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
        _lib = ctypes.CDLL(so)
        _lib.ttxo_dmrgg.argtypes = [POINTER(_Problem), POINTER(_Result)]
        _lib.ttxo_free_result.argtypes = [POINTER(_Result)]
        _lib.ttxo_fun.restype = c_double
        _lib.ttxo_fun.argtypes = [c_int32, c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_double), POINTER(c_double)]
        _lib.ttxo_flang_draw.restype = c_double
        _lib.ttxo_flang_draw.argtypes = [c_uint64]
        _lib.ttxo_lottery2.argtypes = [c_int32, c_int32, c_int32, POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_int32)]
        _lib.ttxo_mvn_init.argtypes = [c_int32, c_double, c_double, POINTER(c_double)]
    return _lib


def _dp(a):
    return a.ctypes.data_as(POINTER(c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(POINTER(c_int32)) if a is not None else None


def dmrgg(n, fun_id, par, maxrank, piv=3, accuracy=None, quad=None, tru=None, aux=None, nproc=1, mybonds=None, accchk=0, user=None):
    """user: address of a C function double f(const int *m, const int *ind, const int *n, const double *par) for fun_id 4"""
    L = lib()
    n = np.ascontiguousarray(n, dtype=np.int32)
    par = np.ascontiguousarray(par, dtype=np.float64)
    aux_ = None if aux is None else np.ascontiguousarray(aux, dtype=np.float64)
    qw = None if quad is None else np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.float64).ravel() for q in quad]))
    mb = None if mybonds is None else np.ascontiguousarray(mybonds, dtype=np.int32)
    pb = _Problem()
    pb.d, pb.n, pb.fun_id, pb.par, pb.npar = n.size, _ip(n), fun_id, _dp(par), par.size
    pb.aux, pb.naux, pb.quadw = _dp(aux_), (0 if aux_ is None else aux_.size), _dp(qw)
    pb.accuracy = -1.0 if accuracy is None else accuracy
    pb.maxrank, pb.piv = maxrank, piv
    pb.tru, pb.has_tru = (0.0 if tru is None else tru), (0 if tru is None else 1)
    pb.nproc, pb.mybonds, pb.verbose, pb.draws, pb.ndraws = nproc, _ip(mb), 0, None, 0
    pb.user = user
    res = _Result()
    rc = L.ttxo_dmrgg(ctypes.byref(pb), ctypes.byref(res))
    if rc:
        raise RuntimeError("oracle failed")
    d = n.size
    sweeps = [dict(it=res.sweeps[i].it, dir=res.sweeps[i].dir, erank=res.sweeps[i].erank, neval=res.sweeps[i].neval,
                   val=res.sweeps[i].val, amax=res.sweeps[i].amax, pivotmax=res.sweeps[i].pivotmax,
                   pivotmin=res.sweeps[i].pivotmin) for i in range(res.nsweeps)]
    if res.nsweeps > 1:
        tapes = np.ctypeslib.as_array(res.tapes, shape=((res.nsweeps - 1) * (d + 1) * 4,)).copy().reshape(res.nsweeps - 1, d + 1, 4)
    else:
        tapes = np.zeros((0, d + 1, 4), dtype=np.int32)
    r = np.ctypeslib.as_array(res.r, shape=(d + 1,)).copy()
    cores = []
    for k in range(d):
        sz = int(r[k]) * int(n[k]) * int(r[k + 1])
        cores.append(np.ctypeslib.as_array(res.cores[k], shape=(sz,)).copy().reshape((r[k], n[k], r[k + 1]), order="F"))
    out = dict(sweeps=sweeps, tapes=tapes, r=r, cores=cores, neval=int(res.neval), value=float(res.value), seconds=float(res.seconds),
               rngpos=int(res.rngpos))
    if accchk:
        L.ttxo_accchk.argtypes = [POINTER(_Problem), POINTER(_Result), ctypes.c_int, POINTER(c_double), POINTER(c_double),
                                  POINTER(c_double), POINTER(c_double), POINTER(c_int32)]
        L.ttxo_accchk.restype = None
        e1, e2, a1, a2 = c_double(), c_double(), c_double(), c_double()
        piv_ = np.zeros(d, dtype=np.int32)
        L.ttxo_accchk(ctypes.byref(pb), ctypes.byref(res), accchk, ctypes.byref(e1), ctypes.byref(e2), ctypes.byref(a1), ctypes.byref(a2), _ip(piv_))
        out["accchk"] = dict(einf=e1.value, efro=e2.value, ainf=a1.value, afro=a2.value, pivot=piv_)
    L.ttxo_free_result(ctypes.byref(res))
    return out


def fun(fun_id, n, par, ind, aux=None):
    L = lib()
    n = np.ascontiguousarray(n, dtype=np.int32)
    par = np.ascontiguousarray(par, dtype=np.float64)
    ind = np.ascontiguousarray(ind, dtype=np.int32)
    aux_ = None if aux is None else np.ascontiguousarray(aux, dtype=np.float64)
    return np.array([L.ttxo_fun(fun_id, n.size, _ip(ind[t]), _ip(n), _dp(par), _dp(aux_)) for t in range(ind.shape[0])])


def lottery2(npnt, wcol, wrow, draws):
    L = lib()
    wcol = np.ascontiguousarray(wcol, dtype=np.float64)
    wrow = np.ascontiguousarray(wrow, dtype=np.float64)
    d = np.ascontiguousarray(draws, dtype=np.float64)
    pts = np.zeros(2 * npnt, dtype=np.int32)
    L.ttxo_lottery2(npnt, wcol.size, wrow.size, _dp(wcol), _dp(wrow), _dp(d), _ip(pts))
    return pts.reshape(2, npnt)


def flang_draws(start, count):
    L = lib()
    return np.array([L.ttxo_flang_draw(start + k) for k in range(count)])


def mvn_init(d):
    aux = np.zeros(d + d * d + 1)
    lib().ttxo_mvn_init(d, 0.0, 1.0, _dp(aux))
    return aux


# ---- tt_lib utilities (oracle/ttx_oracle_tt.c) -------------------------------------------------------------
class _TT(ctypes.Structure):
    _fields_ = [("d", c_int32), ("n", POINTER(c_int32)), ("r", POINTER(c_int32)), ("cores", POINTER(POINTER(c_double)))]


class OracleTT:
    """A TT on the C heap of the oracle; cores in/out as Fortran-ordered (r0, n, r1) numpy arrays."""

    def __init__(self, cores):
        L = lib()
        L.ttxo_tt_new.restype = POINTER(_TT)
        L.ttxo_tt_new.argtypes = [c_int32, POINTER(c_int32), POINTER(c_int32)]
        L.ttxo_tt_free.argtypes = [POINTER(_TT)]
        L.ttxo_tt_ort.argtypes = [POINTER(_TT)]
        L.ttxo_tt_svd.argtypes = [POINTER(_TT), c_double, ctypes.c_int]
        L.ttxo_tt_norm.argtypes = [POINTER(_TT), c_double]
        L.ttxo_tt_norm.restype = c_double
        L.ttxo_tt_dot.argtypes = [POINTER(_TT), POINTER(_TT)]
        L.ttxo_tt_dot.restype = c_double
        L.ttxo_tt_ijk.argtypes = [POINTER(_TT), POINTER(c_int32)]
        L.ttxo_tt_ijk.restype = c_double
        d = len(cores)
        n = np.array([c.shape[1] for c in cores], dtype=np.int32)
        r = np.array([cores[0].shape[0]] + [c.shape[2] for c in cores], dtype=np.int32)
        self.p = L.ttxo_tt_new(d, _ip(n), _ip(r))
        for k, c in enumerate(cores):
            flat = np.asfortranarray(c, dtype=np.float64).ravel(order="F")
            ctypes.memmove(self.p.contents.cores[k], flat.ctypes.data, flat.nbytes)

    def __del__(self):
        if getattr(self, "p", None):
            lib().ttxo_tt_free(self.p)
            self.p = None

    @property
    def ranks(self):
        d = self.p.contents.d
        return np.array([self.p.contents.r[i] for i in range(d + 1)], dtype=np.int32)

    def cores(self):
        t, out = self.p.contents, []
        for k in range(t.d):
            shp = (t.r[k], t.n[k], t.r[k + 1])
            out.append(np.ctypeslib.as_array(t.cores[k], shape=(shp[0] * shp[1] * shp[2],)).copy().reshape(shp, order="F"))
        return out

    def ort(self):
        lib().ttxo_tt_ort(self.p)

    def svd(self, tol, rmax=0):
        lib().ttxo_tt_svd(self.p, tol, rmax)

    def norm(self, tol=None):
        return lib().ttxo_tt_norm(self.p, -1.0 if tol is None else tol)

    def dot(self, other):
        return lib().ttxo_tt_dot(self.p, other.p)

    def zquad(self, w):
        """w: complex array of length sum(n) (rank-1 weights per mode, concatenated)"""
        L = lib()
        L.ttxo_tt_zquad.argtypes = [POINTER(_TT), POINTER(c_double), POINTER(c_double)]
        ww = np.ascontiguousarray(np.asarray(w, dtype=np.complex128)).view(np.float64)
        out = np.zeros(2)
        L.ttxo_tt_zquad(self.p, _dp(ww), _dp(out))
        return complex(out[0], out[1])

    def ijk(self, ind):
        a = np.ascontiguousarray(ind, dtype=np.int32)
        return lib().ttxo_tt_ijk(self.p, _ip(a))
