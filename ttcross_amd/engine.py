"""ctypes binding of libttx.so (include/ttx.h) and a host-side mirror of the reference's dmrgg_lib API.

Reference interface mirrored (lib/dmrgg.f90:11-26):
    subroutine dtt_dmrgg(arg, fun, par, accuracy, maxrank, mybonds, pivoting, neval, quad, tru)
    double precision function dtt_quad(arg, quad, mybonds)
`fun` is replaced by the id of a built-in device integrand (TTX_FUN_*), everything else keeps its name,
meaning and error behaviour (the reference prints and stops; here TTXError carries the same message).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int32, c_int64, c_uint8, c_uint64, c_void_p

import numpy as np

TTX_FUN_ISING, TTX_FUN_STDNORM, TTX_FUN_MVN, TTX_FUN_HOST = 1, 2, 3, 4
K_NAMES = ("lottery", "halfstep", "accept", "exchange", "quad", "other")

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    """libttx.so of this tree; TTX_LIB names another build of the same sources (e.g. the -DTTX_STAMPS phase-timing build)."""
    return os.environ.get("TTX_LIB") or os.path.join(_HERE, "lib", "libttx.so")


class TTXError(RuntimeError):
    pass


class _Config(ctypes.Structure):
    _fields_ = [("d", c_int32), ("n", POINTER(c_int32)), ("fun_id", c_int32), ("par", POINTER(c_double)),
                ("npar", c_int32), ("aux", POINTER(c_double)), ("naux", c_int32), ("quadw", POINTER(c_double)),
                ("accuracy", c_double), ("maxrank", c_int32), ("pivoting", c_int32), ("tru", c_double),
                ("has_tru", c_int32), ("nproc", c_int32), ("mybonds", POINTER(c_int32)), ("device", c_int32),
                ("world_rank", c_int32), ("world_size", c_int32), ("verbose", c_int32), ("arith", c_int32)]


_SENDRECV = ctypes.CFUNCTYPE(ctypes.c_int, c_void_p, ctypes.c_int, c_void_p, c_int64, ctypes.c_int, c_void_p, c_int64)
_ALLREDUCE = ctypes.CFUNCTYPE(ctypes.c_int, c_void_p, POINTER(c_double), c_int64, ctypes.c_int)


class _Transport(ctypes.Structure):
    _fields_ = [("ctx", c_void_p), ("sendrecv", _SENDRECV), ("allreduce", _ALLREDUCE)]


class SweepRec(ctypes.Structure):
    _fields_ = [("it", c_int32), ("dir", c_int32), ("erank", c_double), ("neval", c_int64), ("val", c_double),
                ("amax", c_double), ("pivotmax", c_double), ("pivotmin", c_double), ("seconds", c_double)]


_lib = None


def load_library():
    """Load libttx.so; fails loudly if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise TTXError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the engine has no CPU path)")
    L = ctypes.CDLL(p)
    L.ttx_last_error.restype = c_char_p
    L.ttx_version.restype = ctypes.c_int
    L.ttx_create.argtypes = [POINTER(c_void_p), POINTER(_Config)]
    L.ttx_destroy.argtypes = [c_void_p]
    L.ttx_destroy.restype = None
    L.ttx_comm_unique_id.argtypes = [POINTER(c_uint8)]
    L.ttx_comm_init.argtypes = [c_void_p, POINTER(c_uint8)]
    L.ttx_set_transport.argtypes = [c_void_p, POINTER(_Transport)]
    L.ttx_run.argtypes = [c_void_p]
    L.ttx_num_sweeps.argtypes = [c_void_p]
    L.ttx_get_sweeps.argtypes = [c_void_p, POINTER(SweepRec), ctypes.c_int]
    L.ttx_get_tapes.argtypes = [c_void_p, POINTER(c_int32), c_int64]
    L.ttx_neval.argtypes = [c_void_p]
    L.ttx_neval.restype = c_int64
    L.ttx_seconds.argtypes = [c_void_p]
    L.ttx_seconds.restype = c_double
    L.ttx_get_ranks.argtypes = [c_void_p, POINTER(c_int32)]
    L.ttx_core_size.argtypes = [c_void_p, ctypes.c_int]
    L.ttx_core_size.restype = c_int64
    L.ttx_get_core.argtypes = [c_void_p, ctypes.c_int, POINTER(c_double)]
    L.ttx_quad.argtypes = [c_void_p, POINTER(c_double), POINTER(c_double)]
    L.ttx_set_profile.argtypes = [c_void_p, ctypes.c_int]
    L.ttx_ort.argtypes = [c_void_p]
    L.ttx_svd.argtypes = [c_void_p, c_double, c_int32]
    L.ttx_norm.argtypes = [c_void_p, c_double, POINTER(c_double)]
    L.ttx_dot.argtypes = [c_void_p, c_void_p, POINTER(c_double)]
    L.ttx_ijk.argtypes = [c_void_p, POINTER(c_int32), POINTER(c_double)]
    L.ttx_zquad.argtypes = [c_void_p, c_int32, POINTER(c_double), POINTER(c_double)]
    L.ttx_accchk.argtypes = [c_void_p, c_int32, POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_int32)]
    L.ttx_from_tt.argtypes = [POINTER(c_void_p), c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_double), c_int32]
    L.ttx_write.argtypes = [c_void_p, ctypes.c_char_p]
    L.ttx_read.argtypes = [POINTER(c_void_p), ctypes.c_char_p, c_int32]
    L.ttx_get_modes.argtypes = [c_void_p, POINTER(c_int32), POINTER(c_int32)]
    L.ttx_kernel_stats.argtypes = [c_void_p, POINTER(c_int64), POINTER(c_double), POINTER(c_double)]
    L.ttx_k_residual_argmax.argtypes = [c_int32, c_int32, c_int32, POINTER(c_double), POINTER(c_double),
                                        POINTER(c_double), POINTER(c_double), POINTER(c_int32), POINTER(c_double)]
    L.ttx_k_residual_bench.argtypes = [c_int32, c_int64, c_int32, c_int32, POINTER(c_double), POINTER(c_double)]
    L.ttx_k_eval.argtypes = [c_int32, c_int32, c_int32, POINTER(c_int32), POINTER(c_double), c_int32,
                             POINTER(c_double), c_int32, c_int64, POINTER(c_int32), POINTER(c_double)]
    L.ttx_k_lottery.argtypes = [c_int32, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32),
                                c_uint64, POINTER(c_int32)]
    L.ttx_set_integrand_host.argtypes = [c_void_p, c_void_p, POINTER(c_double)]
    L.ttx_host_calls.argtypes = [c_void_p]
    L.ttx_host_calls.restype = c_int64
    L.ttx_k_exp.argtypes = [c_int32, c_int64, POINTER(c_double), POINTER(c_double)]
    L.ttx_exp_host.argtypes = [c_int64, POINTER(c_double), POINTER(c_double)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise TTXError(load_library().ttx_last_error().decode())


def _dp(a):
    return a.ctypes.data_as(POINTER(c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(POINTER(c_int32)) if a is not None else None


def split_groups(nproc, world_rank, world_size):
    """Bond groups [g0, g0+G) held by one process: contiguous deal, as ttx_create does (include/ttx.h, nproc/world_size)."""
    g0 = nproc * world_rank // world_size
    return g0, nproc * (world_rank + 1) // world_size - g0


def neighbour_ranks(nproc, world_rank, world_size):
    """(left, right) process ranks the boundary groups exchange with; -1 where the chain ends."""
    g0, G = split_groups(nproc, world_rank, world_size)
    return (world_rank - 1 if g0 > 0 else -1), (world_rank + 1 if g0 + G < nproc else -1)


def make_dist_transport(dist, group=None):
    """ctypes thunks (sendrecv, allreduce) of include/ttx.h's ttx_transport over torch.distributed CPU tensors."""
    import torch

    def _t(ptr, nbytes):
        return torch.frombuffer((ctypes.c_char * nbytes).from_address(ptr), dtype=torch.uint8)

    def sendrecv(ctx, to, sbuf, ns, frm, rbuf, nr):
        try:
            ops = []
            if to >= 0:
                ops.append(dist.P2POp(dist.isend, _t(sbuf, ns), to, group))
            if frm >= 0:
                ops.append(dist.P2POp(dist.irecv, _t(rbuf, nr), frm, group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return 0
        except Exception as e:  # noqa: BLE001
            print("transport sendrecv failed:", e, flush=True)
            return 1

    def allreduce(ctx, buf, count, op):
        try:
            t = torch.frombuffer((ctypes.c_double * count).from_address(ctypes.addressof(buf.contents)), dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:  # noqa: BLE001
            print("transport allreduce failed:", e, flush=True)
            return 1

    return _SENDRECV(sendrecv), _ALLREDUCE(allreduce)


class TTCross:
    """One dtt_dmrgg problem resident on one MI355X (the `type(dtt) :: arg` of the reference plus the
    sweep state).  n: mode sizes arg%n(1:d); quad: list/array of per-mode weight vectors (rank-1 TT)."""

    def __init__(self, n, fun_id, par, maxrank, pivoting=3, accuracy=None, quad=None, tru=None, aux=None,
                 nproc=1, mybonds=None, device=0, verbose=False, arith=None, world_rank=0, world_size=1):
        L = load_library()
        self._n = np.ascontiguousarray(n, dtype=np.int32)
        self.d = int(self._n.size)
        self._par = np.ascontiguousarray(par, dtype=np.float64)
        self._aux = None if aux is None else np.ascontiguousarray(aux, dtype=np.float64)
        self._quad = None if quad is None else np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.float64).ravel() for q in quad]))
        self._mybonds = None if mybonds is None else np.ascontiguousarray(mybonds, dtype=np.int32)
        c = _Config()
        c.d = self.d
        c.n = _ip(self._n)
        c.fun_id = fun_id
        c.par = _dp(self._par)
        c.npar = self._par.size
        c.aux = _dp(self._aux)
        c.naux = 0 if self._aux is None else self._aux.size
        c.quadw = _dp(self._quad)
        c.accuracy = -1.0 if accuracy is None else float(accuracy)
        c.maxrank = int(maxrank)
        c.pivoting = int(pivoting)
        c.tru = 0.0 if tru is None else float(tru)
        c.has_tru = 0 if tru is None else 1
        c.nproc = int(nproc)
        c.mybonds = _ip(self._mybonds)
        c.device = int(device)
        c.world_rank, c.world_size = int(world_rank), int(world_size)
        self.world_rank, self.world_size = int(world_rank), int(world_size)
        c.verbose = 1 if verbose else 0
        c.arith = 1 if arith in (1, "fast") else 0      # None / "exact": exact unless TTX_ARITH=fast is set
        self._h = c_void_p()
        _check(L.ttx_create(ctypes.byref(self._h), ctypes.byref(c)))

    # ---- trains that do not come from a sweep (lib/ttio.f90) -----------------------------------
    @classmethod
    def _adopt(cls, handle):
        L = load_library()
        self = cls.__new__(cls)
        self._h = handle
        d = c_int32()
        _check(L.ttx_get_modes(handle, ctypes.byref(d), None))
        self.d = d.value
        self._n = np.zeros(self.d, dtype=np.int32)
        _check(L.ttx_get_modes(handle, ctypes.byref(d), _ip(self._n)))
        self.world_rank, self.world_size = 0, 1
        return self

    @classmethod
    def from_cores(cls, cores, device=0):
        """Upload a train given as (r(k-1), n(k), r(k)) arrays; it becomes the resident train of a new engine."""
        cores = [np.asarray(c, dtype=np.float64) for c in cores]
        n = np.array([c.shape[1] for c in cores], dtype=np.int32)
        r = np.array([cores[0].shape[0]] + [c.shape[2] for c in cores], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([c.ravel(order="F") for c in cores]))
        h = c_void_p()
        _check(load_library().ttx_from_tt(ctypes.byref(h), len(cores), _ip(n), _ip(r), _dp(flat), int(device)))
        return cls._adopt(h)

    @classmethod
    def read(cls, path, device=0):
        """dtt_read (lib/ttio.f90:196-297): load the reference's stream file onto the device."""
        h = c_void_p()
        _check(load_library().ttx_read(ctypes.byref(h), os.fsencode(path), int(device)))
        return cls._adopt(h)

    def write_hdf5(self, path):
        """save_dtt_to_hdf5 (lib/utils.f90:8-57): group TT with modes, ranks and core_k."""
        L = load_library()
        L.ttx_write_hdf5.argtypes = [c_void_p, ctypes.c_char_p]
        _check(L.ttx_write_hdf5(self._h, os.fsencode(path)))

    @classmethod
    def read_hdf5(cls, path, device=0):
        L = load_library()
        L.ttx_read_hdf5.argtypes = [POINTER(c_void_p), ctypes.c_char_p, c_int32]
        h = c_void_p()
        _check(L.ttx_read_hdf5(ctypes.byref(h), os.fsencode(path), int(device)))
        return cls._adopt(h)

    def replicate(self):
        """The train of a multi-process job gathered onto this process as a new single-process engine (collective; include/ttx.h)."""
        L = load_library()
        L.ttx_replicate.argtypes = [c_void_p, POINTER(c_void_p)]
        h = c_void_p()
        _check(L.ttx_replicate(self._h, ctypes.byref(h)))
        return TTCross._adopt(h)

    def write(self, path):
        """dtt_write (lib/ttio.f90:29-108): the resident train in the reference's stream format."""
        _check(load_library().ttx_write(self._h, os.fsencode(path)))

    def close(self):
        if getattr(self, "_h", None):
            load_library().ttx_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- multi-GPU bootstrap (replaces mpi_init of the drivers) ---------------------------------------
    def comm_init(self, dist):
        """RCCL transport: rank 0 creates the unique id, torch.distributed broadcasts it, every rank joins."""
        import torch
        L = load_library()
        buf = (c_uint8 * 128)()
        if self.world_rank == 0:
            _check(L.ttx_comm_unique_id(buf))
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor(list(buf), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        ids = t.cpu().tolist()
        for i in range(128):
            buf[i] = ids[i]
        _check(L.ttx_comm_init(self._h, buf))

    def comm_init_shm(self, name):
        """The engine's built-in node-local transport over POSIX shared memory (include/ttx.h: ttx_comm_init_shm): every
        rank of the job passes the same name.  No MPI, no RCCL, no torch.distributed."""
        L = load_library()
        L.ttx_comm_init_shm.argtypes = [c_void_p, ctypes.c_char_p]
        _check(L.ttx_comm_init_shm(self._h, name.encode()))

    def set_dist_transport(self, dist, group=None):
        """Host-callback transport over torch.distributed CPU tensors (gloo): used where RCCL cannot run
        (several ranks on one GPU) -- same engine code path, messages staged through pinned host memory."""
        self._cb = make_dist_transport(dist, group)                   # keep the thunks alive
        tr = _Transport(None, self._cb[0], self._cb[1])
        _check(load_library().ttx_set_transport(self._h, ctypes.byref(tr)))

    def set_integrand_host(self, fun_addr, par=None):
        """The reference's user callback `fun(m, ind, n, par)` (lib/dmrgg.f90:18) for an engine created with
        fun_id = TTX_FUN_HOST: fun_addr is the address of a C / Fortran function with that interface (everything by
        reference), par the array handed to it untouched.  Fibers are then evaluated on the host, the sweep on the GPU."""
        self._hpar = None if par is None else np.ascontiguousarray(par, dtype=np.float64)
        _check(load_library().ttx_set_integrand_host(self._h, c_void_p(fun_addr), _dp(self._hpar)))
        return self

    @property
    def host_calls(self):
        return int(load_library().ttx_host_calls(self._h))

    @property
    def arith(self):
        """'exact' or 'fast': the arithmetic the integrand is evaluated with (include/ttx.h: ttx_arith)."""
        L = load_library()
        L.ttx_arith.argtypes = [c_void_p]
        return ("exact", "fast")[L.ttx_arith(self._h)]

    def sweep_path(self):
        """'chain', 'fused' or 'cluster': the sweep implementation chosen at creation (TTX_SWEEP)."""
        L = load_library()
        L.ttx_sweep_path.argtypes = [c_void_p]
        return ("chain", "fused", "cluster")[L.ttx_sweep_path(self._h)]

    @property
    def resid_halfsteps(self):
        L = load_library()
        L.ttx_resid_halfsteps.argtypes = [c_void_p]
        L.ttx_resid_halfsteps.restype = c_int64
        return int(L.ttx_resid_halfsteps(self._h))

    @property
    def cluster_fallbacks(self):
        """Runs replayed on the chain path after a wait inside the cluster sweep kernel timed out (include/ttx.h)."""
        L = load_library()
        L.ttx_cluster_fallbacks.argtypes = [c_void_p]
        return int(L.ttx_cluster_fallbacks(self._h))

    @property
    def det_fallbacks(self):
        """Runs repeated without the wave teams / relay of the Ising D/E half-step after a reported fault (include/ttx.h)."""
        L = load_library()
        L.ttx_det_fallbacks.argtypes = [c_void_p]
        return int(L.ttx_det_fallbacks(self._h))

    def set_profile(self, on=True):
        _check(load_library().ttx_set_profile(self._h, 1 if on else 0))

    def run(self):
        _check(load_library().ttx_run(self._h))
        return self

    # ---- results -----------------------------------------------------------------------------
    @property
    def neval(self):
        return int(load_library().ttx_neval(self._h))

    @property
    def seconds(self):
        return float(load_library().ttx_seconds(self._h))

    def sweeps(self):
        L = load_library()
        k = L.ttx_num_sweeps(self._h)
        buf = (SweepRec * k)()
        _check(L.ttx_get_sweeps(self._h, buf, k))
        return [dict(it=b.it, dir=b.dir, erank=b.erank, neval=b.neval, val=b.val, amax=b.amax,
                     pivotmax=b.pivotmax, pivotmin=b.pivotmin, seconds=b.seconds) for b in buf]

    def tapes(self):
        L = load_library()
        k = L.ttx_num_sweeps(self._h) - 1
        out = np.zeros((max(k, 0), self.d + 1, 4), dtype=np.int32)
        if k > 0:
            _check(L.ttx_get_tapes(self._h, _ip(out), out.size))
        return out

    def ranks(self):
        r = np.zeros(self.d + 1, dtype=np.int32)
        _check(load_library().ttx_get_ranks(self._h, _ip(r)))
        return r

    def core(self, k):
        """arg%u(k)%p as a Fortran-ordered (r(k-1), n(k), r(k)) array, k = 1..d."""
        L = load_library()
        r = self.ranks()
        sz = L.ttx_core_size(self._h, k)
        buf = np.zeros(sz, dtype=np.float64)
        _check(L.ttx_get_core(self._h, k, _dp(buf)))
        return buf.reshape((r[k - 1], self._n[k - 1], r[k]), order="F")

    def quad(self, w=None):
        """dtt_quad(arg, quad) (lib/dmrgg.f90:1261); w = list of per-mode weight vectors or None."""
        v = c_double()
        if w is not None:     # the C entry point takes the concatenated vectors on trust (as the reference takes its rank-1 train)
            if len(w) != self.d or any(np.size(q) != int(nk) for q, nk in zip(w, self._n)):
                raise ValueError(f"quad: {self.d} weight vectors of lengths {list(map(int, self._n))} expected")
        wa = None if w is None else np.ascontiguousarray(np.concatenate([np.asarray(q, dtype=np.float64).ravel() for q in w]))
        _check(load_library().ttx_quad(self._h, _dp(wa), ctypes.byref(v)))
        return v.value

    # ---- tt_lib utilities on the resident TT (lib/tt.f90: ort, svd, norm, dot_product, tijk) --------------
    def ort(self):
        _check(load_library().ttx_ort(self._h))
        return self

    def svd(self, tol, rmax=0):
        _check(load_library().ttx_svd(self._h, float(tol), int(rmax)))
        return self

    def norm(self, tol=None):
        v = c_double()
        _check(load_library().ttx_norm(self._h, -1.0 if tol is None else float(tol), ctypes.byref(v)))
        return v.value

    def dot(self, other):
        v = c_double()
        _check(load_library().ttx_dot(self._h, other._h, ctypes.byref(v)))
        return v.value

    def zquad(self, w):
        """ztt_quad (lib/dmrgg.f90:1418) batched: w complex array (nf, sum(n)) of rank-1 weights; returns nf complex values."""
        ww = np.ascontiguousarray(np.atleast_2d(np.asarray(w, dtype=np.complex128)))
        if ww.shape[1] != int(self._n.sum()):
            raise ValueError(f"zquad: weight rows of length sum(n) = {int(self._n.sum())} expected (got {ww.shape[1]})")
        out = np.zeros(2 * ww.shape[0])
        _check(load_library().ttx_zquad(self._h, ww.shape[0], _dp(ww.view(np.float64)), _dp(out)))
        return out.view(np.complex128).copy()

    def tijk(self, ind):
        a = np.ascontiguousarray(ind, dtype=np.int32)
        if a.size != self.d:
            raise ValueError(f"tijk: a multi-index of {self.d} entries expected")
        v = c_double()
        _check(load_library().ttx_ijk(self._h, _ip(a), ctypes.byref(v)))
        return v.value

    def accchk(self, nlot):
        """dtt_accchk (lib/dmrgg.f90:1081): dict(einf, efro, ainf, afro, pivot) from nlot random samples."""
        e1, e2, a1, a2 = c_double(), c_double(), c_double(), c_double()
        pv = np.zeros(self.d, dtype=np.int32)
        _check(load_library().ttx_accchk(self._h, nlot, ctypes.byref(e1), ctypes.byref(e2), ctypes.byref(a1), ctypes.byref(a2), _ip(pv)))
        return dict(einf=e1.value, efro=e2.value, ainf=a1.value, afro=a2.value, pivot=pv)

    def kernel_stats(self):
        n = (c_int64 * 6)()
        ms = (c_double * 6)()
        by = (c_double * 6)()
        _check(load_library().ttx_kernel_stats(self._h, n, ms, by))
        return {K_NAMES[i]: dict(launches=int(n[i]), ms=float(ms[i]), bytes=float(by[i])) for i in range(6)}


def dtt_dmrgg(n, fun_id, par, accuracy=None, maxrank=None, mybonds=None, pivoting=3, quad=None, tru=None, aux=None,
              nproc=1, device=0, verbose=False):
    """Mirror of `call dtt_dmrgg(arg, fun, par, accuracy, maxrank, mybonds, pivoting, neval, quad, tru)`;
    returns the engine (holding the finalised cores, ranks, neval and per-sweep records)."""
    if maxrank is None:
        raise TTXError("dtt_dmrgg: maxrank is required by the device engine (it sizes HBM storage)")
    return TTCross(n, fun_id, par, maxrank, pivoting=pivoting, accuracy=accuracy, quad=quad, tru=tru, aux=aux,
                   nproc=nproc, mybonds=mybonds, device=device, verbose=verbose).run()


# ---- kernel-level entry points (parity tests) --------------------------------------------------------
def k_residual_argmax(a, F, x, device=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    F = np.asfortranarray(F, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    m, r = F.shape
    b = np.zeros(m)
    im = c_int32()
    bm = c_double()
    _check(load_library().ttx_k_residual_argmax(device, m, r, _dp(a), F.ctypes.data_as(POINTER(c_double)), _dp(x), _dp(b),
                                                ctypes.byref(im), ctypes.byref(bm)))
    return b, im.value, bm.value


def k_residual_bench(m, r, iters=20, device=0):
    """(avg kernel ms, algorithmic bytes) of the K2 residual + arg-max kernel on an m x r factor in HBM."""
    ms, by = c_double(), c_double()
    _check(load_library().ttx_k_residual_bench(device, m, r, iters, ctypes.byref(ms), ctypes.byref(by)))
    return ms.value, by.value


def k_eval(fun_id, n, par, ind, aux=None, device=0, arith=None):
    n = np.ascontiguousarray(n, dtype=np.int32)
    par = np.ascontiguousarray(par, dtype=np.float64)
    ind = np.ascontiguousarray(ind, dtype=np.int32)
    aux_ = None if aux is None else np.ascontiguousarray(aux, dtype=np.float64)
    out = np.zeros(ind.shape[0])
    L = load_library()
    L.ttx_k_eval_arith.argtypes = L.ttx_k_eval.argtypes + [c_int32]
    _check(L.ttx_k_eval_arith(device, fun_id, n.size, _ip(n), _dp(par), par.size, _dp(aux_), 0 if aux_ is None else aux_.size,
                              ind.shape[0], _ip(ind), _dp(out), 1 if arith in (1, "fast") else 0))
    return out


def k_lottery(npnt, m, n, zcol, zrow, rngpos=0, device=0):
    zc = np.ascontiguousarray(zcol, dtype=np.int32)
    zr = np.ascontiguousarray(zrow, dtype=np.int32)
    pts = np.zeros(2 * npnt, dtype=np.int32)
    _check(load_library().ttx_k_lottery(device, npnt, m, n, zc.size, _ip(zc), _ip(zr), rngpos, _ip(pts)))
    return pts.reshape(2, npnt)


def k_exp(x, device=0):
    """The integrands' exp (ttx_exp.h) evaluated on the device."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    _check(load_library().ttx_k_exp(device, x.size, _dp(x), _dp(out)))
    return out


def exp_host(x):
    """The same source instantiated on the host (needs no GPU): pins ttx_exp.h against the run-time libm."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    _check(load_library().ttx_exp_host(x.size, _dp(x), _dp(out)))
    return out


def k_latency_probe(device=0):
    """dict of unit latencies in ns measured on one wave (include/ttx.h: ttx_k_latency_probe)."""
    L = load_library()
    L.ttx_k_latency_probe.argtypes = [c_int32, POINTER(c_double)]
    o = (c_double * 5)()
    _check(L.ttx_k_latency_probe(device, o))
    return dict(fp64_mul_ns=o[0], fp64_mul_add_ns=o[1], l2_roundtrip_ns=o[2], lds_read_ns=o[3], fp64_div_ns=o[4])
