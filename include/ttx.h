/*
 * ttx.h -- C-ABI of libttx.so, the MI355X-native engine for the reference's dtt_dmrgg greedy-cross sweep.
 *
 * The reference (aukeschaap/ttcross) has no FFI layer: its boundary for this path is the Fortran module
 * interface of lib/dmrgg.f90 / lib/tt.f90 that the test_crs_* drivers `use`.  Each entry point below names
 * the reference interface it replaces; INTEGRATION.md shows the ISO_C_BINDING stubs with which the
 * reference's own modules (or the drop-in modules in ttcross_amd/fortran/) bind to them.
 *
 * Plain C types only (no torch / HIP types); every function returns 0 on success, a TTX_E* code
 * otherwise, and ttx_last_error() gives the message (the Fortran shim turns it into the reference's
 * `write(*,*) ...; stop`).  One host thread per engine handle.  The engine NEVER falls back to the CPU:
 * without a gfx950 device ttx_create fails with TTX_ENODEV.
 */
#ifndef TTX_H
#define TTX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTX_OK 0
#define TTX_EINVAL 1   /* bad argument (reference: print + stop, e.g. lib/dmrgg.f90:88-91,114-117,590-592) */
#define TTX_ENODEV 2   /* no usable HIP device                                                           */
#define TTX_EHIP 3     /* HIP / RCCL runtime error                                                       */
#define TTX_ESTATE 4   /* call out of order                                                              */

/* built-in integrands (device code).  A user `fun` of the reference (lib/dmrgg.f90:18) cannot run on the
 * GPU; the drivers' integrands are provided natively and selected by id. */
#define TTX_FUN_ISING 1    /* dfunc_ising_discr, test_crs_ising.f90:176-218 (par(2n+1) = 1/2/3 -> C/D/E) */
#define TTX_FUN_STDNORM 2  /* integrand, test_crs_stdnorm.f90:154-170                                     */
#define TTX_FUN_MVN 3      /* integrand -> mvn_pdf, test_crs_mvn.f90:156-172, lib/mvn_pdf.f90:63-83       */
#define TTX_FUN_HOST 4     /* any user `fun` (lib/dmrgg.f90:18), evaluated on the HOST: ttx_set_integrand_host */

#define TTX_ARITH_EXACT 0
#define TTX_ARITH_FAST 1

typedef struct ttx_engine ttx_engine;

/* Arguments of dtt_dmrgg(arg, fun, par, accuracy, maxrank, mybonds, pivoting, neval, quad, tru),
 * lib/dmrgg.f90:11-26.  arg%l is always 1 (as in every driver); arg%m = d; arg%n = n[]. */
typedef struct ttx_config {
    int32_t d;              /* arg%m : number of TT cores                                            */
    const int32_t *n;       /* arg%n(1:d) : mode sizes                                               */
    int32_t fun_id;         /* TTX_FUN_* : replaces the `fun` callback                               */
    const double *par;      /* par(*) as the driver builds it (nodes, weights, id)                   */
    int32_t npar;
    const double *aux;      /* TTX_FUN_MVN: mu[d], inv_cov[d*d] column-major, det (mvn_data)         */
    int32_t naux;
    const double *quadw;    /* quad : rank-1 TT of per-mode weights, d blocks of n[k]; NULL = absent */
    double accuracy;        /* < 0 : absent                                                          */
    int32_t maxrank;        /* required (>= 1): also sizes device storage                            */
    int32_t pivoting;       /* -1 full, 0 lottery only, p >= 1 rook (reference default 3)            */
    double tru;             /* tru (only used for the `err` column)                                  */
    int32_t has_tru;
    int32_t nproc;          /* total number of bond groups ("MPI ranks" of the reference), >= 1, < d */
    const int32_t *mybonds; /* own(0:nproc), 1-based bonds; NULL -> share(1, d-1) lib/default.f90:78 */
    int32_t device;         /* HIP device ordinal                                                    */
    int32_t world_rank;     /* this process within the multi-GPU job (0 when single process)         */
    int32_t world_size;     /* number of processes (GPUs); groups are split contiguously over them   */
    int32_t verbose;        /* 1: print the reference's per-sweep log lines (lib/dmrgg.f90:971-1008) */
    int32_t arith;          /* TTX_ARITH_EXACT (0, default): every fp64 operation of the integrand in the reference's order,
                             * results bit-identical to the reference restatement; TTX_ARITH_FAST (1): products / sums of the
                             * O(d^2) integrands (Ising D/E, mvn) re-associated, O(d) per fiber element, results equal to
                             * rounding (tolerance-checked).  The environment variable TTX_ARITH=fast|exact overrides 0.     */
} ttx_config;

/* one line of the reference's per-sweep report (lib/dmrgg.f90:971-1008) */
typedef struct ttx_sweep_rec {
    int32_t it;
    int32_t dir;            /* 0 '::', 1 '>>', 2 '<<' */
    double erank;
    int64_t neval;
    double val;
    double amax, pivotmax, pivotmin;
    double seconds;         /* since the start of ttx_run */
} ttx_sweep_rec;

const char *ttx_last_error(void);
int ttx_version(void);

/* allocate device state for one dtt_dmrgg problem (replaces the implicit set-up of lib/dmrgg.f90:58-148) */
int ttx_create(ttx_engine **out, const ttx_config *cfg);
void ttx_destroy(ttx_engine *h);

/* RCCL bootstrap for world_size > 1 (replaces mpi_init / MPI_COMM_WORLD, test_crs_ising.f90:31-36):
 * rank 0 calls ttx_comm_unique_id, the host layer broadcasts the 128 bytes, every rank calls ttx_comm_init. */
int ttx_comm_unique_id(uint8_t id[128]);
int ttx_comm_init(ttx_engine *h, const uint8_t id[128]);

/* Alternative transport for world_size > 1 when RCCL cannot be used (and for multi-process tests on one GPU):
 * the host layer supplies MPI-like primitives on HOST buffers; the engine stages messages through pinned memory.
 * sendrecv: send ns bytes to rank `to` (skip if to < 0) and receive nr bytes from rank `from` (skip if < 0);
 * allreduce: in place on `count` doubles, op 0 = sum, 1 = max.  Both return 0 on success. */
typedef struct ttx_transport {
    void *ctx;
    int (*sendrecv)(void *ctx, int to, const void *sbuf, int64_t ns, int from, void *rbuf, int64_t nr);
    int (*allreduce)(void *ctx, double *buf, int64_t count, int op);
} ttx_transport;
int ttx_set_transport(ttx_engine *h, const ttx_transport *t);
/* Built-in host transport for processes of ONE node, over a POSIX shared-memory segment `name` (every rank passes the same
 * name; rank 0 creates it): the same two primitives, no MPI and no RCCL needed.  Used by the Fortran drop-in layer
 * (TTX_TRANSPORT=shm) and by tests that run several engine processes on one GPU, where RCCL cannot. */
int ttx_comm_init_shm(ttx_engine *h, const char *name);

/* The reference's integrand callback, `double precision,external :: fun` called as fun(m, ind, n, par) (lib/dmrgg.f90:18,
 * walked by dmrgg_fun :1053-1078): Fortran calling convention, everything by reference; ind and n are default integers.
 * An engine created with fun_id = TTX_FUN_HOST evaluates every fiber through this function on the host -- each
 * evaluating kernel first hands the multi-indices it needs to the host, the host calls `fun` from a pool of threads
 * (TTX_HOST_THREADS, else OMP_NUM_THREADS, else the hardware; `fun` must be thread-safe, as under the reference's
 * OpenMP loops), and the kernel continues with the values.  par is passed through untouched (may be NULL: the
 * reference's `par` is optional).  The sweep logic, pivot search and factor updates stay on the device; results are
 * those of the reference for the same `fun`.  With pivoting = -1 the superblock goes to the host one column (k,q) at a time. */
typedef double (*ttx_host_fun)(const int32_t *m, const int32_t *ind, const int32_t *n, const double *par);
/* LIFETIME: the engine keeps the two pointers, it does not copy par.  They must stay valid for every later call that evaluates
 * (ttx_run, ttx_accchk); a caller whose par may move or die calls ttx_set_integrand_host again before such a call (the
 * Fortran dtt_accchk does, with the fun / par it was given: lib/dmrgg.f90:1081 checks against ITS arguments). */
int ttx_set_integrand_host(ttx_engine *h, ttx_host_fun fun, const double *par);
int64_t ttx_host_calls(const ttx_engine *h);                      /* calls of `fun` made by the last ttx_run */

/* dtt_dmrgg itself: initial cross, sweeps until maxrank / 3 strikes, finalisation dtt_lua (lib/dmrgg.f90:151-1049) */
int ttx_run(ttx_engine *h);

/* results */
int ttx_num_sweeps(const ttx_engine *h);                          /* records available (sweep 0 included) */
int ttx_get_sweeps(const ttx_engine *h, ttx_sweep_rec *out, int cap);
int ttx_get_tapes(const ttx_engine *h, int32_t *out /* [nsweeps-1][d+1][4] */, int64_t cap);
int64_t ttx_neval(const ttx_engine *h);                           /* `neval` of dtt_dmrgg                 */
double ttx_seconds(const ttx_engine *h);                          /* wall time of the last ttx_run        */
int ttx_get_ranks(const ttx_engine *h, int32_t *r /* [d+1] */);   /* arg%r(0:d)                           */
int64_t ttx_core_size(const ttx_engine *h, int k);                /* r(k-1)*n(k)*r(k) or 0 if not owned   */
int ttx_get_core(const ttx_engine *h, int k, double *buf);        /* arg%u(k)%p, column-major, k = 1..d   */

/* dtt_quad(arg, quad) on the finalised cores (lib/dmrgg.f90:1261-1415); w = d blocks of n[k] weights, NULL = sum */
int ttx_quad(ttx_engine *h, const double *w, double *val);

/* dtt_accchk(nlot, arg, einf, efro, ainf, afro, fun, par, pivot), lib/dmrgg.f90:1081-1166: nlot random samples of
 * |fun - TT| drawn from the run-time RNG stream where dtt_dmrgg left it (irnd, lib/rnd.f90:83-88); element
 * evaluation as dtt_ijk (lib/tt.f90:630-652).  As in the reference every rank needs all cores: on a multi-process engine the
 * call is COLLECTIVE and works on a replica of the job's train (ttx_replicate); every process gets the same numbers.
 * pivot: d ints (worst sample) or NULL. */
int ttx_accchk(ttx_engine *h, int32_t nlot, double *einf, double *efro, double *ainf, double *afro, int32_t *pivot);

/* tt_lib utilities on the tensor train resident on the device.  ttx_norm / ttx_dot / ttx_zquad / ttx_write on a
 * multi-process engine are COLLECTIVE (ztt_quad folds per-process partial products, lib/dmrgg.f90:1418-1523; the others work on a
 * replica, ttx_replicate); ttx_ort / ttx_svd / ttx_ijk change or address single cores and take single-process engines (or a replica).
 * ttx_ort  : dtt_ort  (lib/tt.f90:130-198)  left-to-right Householder QR, in place
 * ttx_svd  : dtt_svd  (lib/tt.f90:307-368)  rounding: ort, then truncated SVD right-to-left; tol relative
 *            (lib/mat.f90:433-458 chop), rmax <= 0: absent
 * ttx_norm : dtt_norm (lib/tt.f90:1074-1092) Frobenius norm, tol < 0: absent (the TT itself is left unchanged)
 * ttx_dot  : dtt_dot  (lib/tt.f90:1155-1175) scalar product of two resident TTs with equal mode sizes
 * ttx_ijk  : dtt_ijk  (lib/tt.f90:630-652)  one element, ind = d 1-based indices */
int ttx_ort(ttx_engine *h);
int ttx_svd(ttx_engine *h, double tol, int32_t rmax);
int ttx_norm(ttx_engine *h, double tol, double *val);
int ttx_dot(ttx_engine *hx, ttx_engine *hy, double *val);
int ttx_ijk(ttx_engine *h, const int32_t *ind, double *val);
/* ztt_quad (lib/dmrgg.f90:1418-1523) of the (real) resident TT with COMPLEX rank-1 weights, batched over nf weight
 * sets (the 32 frequencies of test_crs_chf.f90:153-168 in one call): w = nf blocks of sum(n) interleaved (re, im)
 * doubles, out = nf (re, im) pairs.  Multi-process engines: collective, the value on every process. */
int ttx_zquad(ttx_engine *h, int32_t nf, const double *w, double *out);

/* The finalised train of a MULTI-PROCESS job gathered onto EVERY process as a new single-process engine (same integrand, ranks,
 * RNG position; *out is owned by the caller: ttx_destroy).  Collective over the job's transport (each process contributes the
 * cores it holds; the others arrive by a SUM all-reduce into zero-filled slots, which is exact).  The reference's dtt_accchk,
 * norm, dot_product, ort, svd and dtt_write expect a type(dtt) with all cores (lib/tt.f90; lib/dmrgg.f90:1081-1166). */
int ttx_replicate(ttx_engine *h, ttx_engine **out);

/* Tensor trains that do not come from a sweep.
 * ttx_from_tt : upload a train given as compact column-major cores (d blocks r(k-1)*n(k)*r(k), concatenated) and make
 *               it the resident train of a new single-process engine (the reference: plain assignment into arg%u(k)%p);
 *               all of ttx_get_*, ttx_quad, ttx_zquad and the tt_lib utilities work on it; ttx_run / ttx_accchk do not
 *               (no integrand).  ranks must be <= 128.
 * ttx_write   : dtt_write (lib/ttio.f90:29-108), the reference's raw stream file: 128-byte header `tthead`
 *               (lib/ttio.f90:10-17), l, m, n(l:m), r(l-1:m) as int32, all cores as float64; l = 1.
 * ttx_read    : dtt_read (lib/ttio.f90:196-297) of such a file into a new engine (checks 'TT' and version 1).
 * ttx_get_modes: arg%m and arg%n(1:m) of an engine (n may be NULL). */
int ttx_from_tt(ttx_engine **out, int32_t d, const int32_t *n, const int32_t *r, const double *cores, int32_t device);
int ttx_write(const ttx_engine *h, const char *path);
int ttx_read(ttx_engine **out, const char *path, int32_t device);
int ttx_get_modes(const ttx_engine *h, int32_t *d, int32_t *n);
/* save_dtt_to_hdf5 (lib/utils.f90:8-57): group "TT", datasets "modes", "ranks" (native int) and "core_k", k = 0..m-1, with
 * the Fortran shape (r(k-1), n(k), r(k)); ttx_read_hdf5 loads such a file into a new engine (the reference has no
 * reader; this one exists for round trips and checkpoints).  libhdf5.so is resolved at run time (dlopen): without it both
 * return TTX_EINVAL with a message. */
int ttx_write_hdf5(const ttx_engine *h, const char *path);
int ttx_read_hdf5(ttx_engine **out, const char *path, int32_t device);

/* profiling: with on != 0 the next ttx_run brackets every kernel launch with HIP events on the engine's
 * stream; ttx_kernel_stats then reports, per kernel kind, launches and total milliseconds. */
#define TTX_K_LOTTERY 0
#define TTX_K_HALFSTEP 1   /* fiber evaluation + residual + argmax (the north star's "maxvol" kernel) */
#define TTX_K_ACCEPT 2
#define TTX_K_EXCHANGE 3
#define TTX_K_QUAD 4
#define TTX_K_OTHER 5
#define TTX_K_NKINDS 6
int ttx_set_profile(ttx_engine *h, int on);
/* which implementation of the sweep this engine uses (TTX_SWEEP=auto|chain|fused|cluster chooses at ttx_create):
 * 0 multi-kernel chain (k_lottery / k_halfstep / k_accept per bond), 1 one workgroup per bond group for the whole
 * sweep (k_sweep_fused), 2 a cluster of workgroups per bond group for the whole sweep (k_sweep_cluster) */
int ttx_sweep_path(const ttx_engine *h);
/* the arithmetic this engine evaluates its integrand with (TTX_ARITH_*): FAST only where ttx_config.arith / TTX_ARITH asked for
 * it AND the integrand has a re-associated evaluator (Ising D/E with all nodes in [0,1], mvn); everything else runs exact */
int ttx_arith(const ttx_engine *h);
/* runs of this engine that were replayed on the multi-kernel chain because a wait inside the cluster kernel timed out
 * (its workgroups must all be resident at once; the launch is cooperative and gated by the occupancy calculator, so this
 * is a safety net: after a fallback the engine stays on the chain path; results are identical on every path) */
int ttx_cluster_fallbacks(const ttx_engine *h);
/* runs of this engine that were repeated without the wave teams / the wave relay of the Ising D/E half-step because a launch
 * reported a fault (more units than the grid the host sized, a broken hand-over); results are identical on every path */
int ttx_det_fallbacks(const ttx_engine *h);
int ttx_fun_id(const ttx_engine *h);                /* TTX_FUN_* the engine was created with (0: a loaded train without integrand) */
int64_t ttx_resid_halfsteps(const ttx_engine *h);   /* rook half-steps of the last run that computed a residual + arg-max (all groups) */
int ttx_kernel_stats(const ttx_engine *h, int64_t launches[TTX_K_NKINDS], double ms[TTX_K_NKINDS], double bytes[TTX_K_NKINDS]);

/* ---- kernel-level entry points used by the parity tests (host buffers in, host buffers out) ---- */

/* K2: residual + first-argmax of one rook half-step (lib/dmrgg.f90:537-546 / 570-579):
 * b = a - F*x in netlib dgemv order, F = m x r column-major (ld m); returns 0-based argmax and b[argmax]. */
int ttx_k_residual_argmax(int32_t device, int32_t m, int32_t r, const double *a, const double *F, const double *x,
                          double *b_out, int32_t *imax, double *bmax);
/* K2 streaming benchmark: the same residual + arg-max kernel on a synthetic m x r factor resident in HBM
 * (m*r*8 bytes, generated on the device); returns the average kernel time over `iters` launches measured with
 * HIP events, and the algorithmic bytes 8*(m*r + r + 2*m) one launch moves. */
int ttx_k_residual_bench(int32_t device, int64_t m, int32_t r, int32_t iters, double *avg_ms, double *bytes);
/* K1: batch integrand evaluation, ind = npts x d (row-major, 1-based indices) */
int ttx_k_eval(int32_t device, int32_t fun_id, int32_t d, const int32_t *n, const double *par, int32_t npar,
               const double *aux, int32_t naux, int64_t npts, const int32_t *ind, double *out);
/* the same with the arithmetic named (TTX_ARITH_FAST: the re-associated one-thread evaluator of ttx_fast.h; Ising D/E, nodes in [0,1]) */
int ttx_k_eval_arith(int32_t device, int32_t fun_id, int32_t d, const int32_t *n, const double *par, int32_t npar,
                     const double *aux, int32_t naux, int64_t npts, const int32_t *ind, double *out, int32_t arith);
/* lottery2 (lib/rnd.f90:105-126) with unit weights except zero at the listed 1-based positions */
int ttx_k_lottery(int32_t device, int32_t npnt, int32_t m, int32_t n, int32_t nz, const int32_t *zcol,
                  const int32_t *zrow, uint64_t rngpos, int32_t *points /* [2*npnt] */);

/* exp() of the integrands (test_crs_stdnorm.f90:168, lib/mvn_pdf.f90:82): the device code restates the run-time
 * library's algorithm operation for operation (ttx_exp.h).  ttx_k_exp evaluates it on the device, ttx_exp_host the same
 * source on the host (needs no GPU) -- both exist so that tests can pin it against libm bit for bit. */
int ttx_k_exp(int32_t device, int64_t n, const double *x, double *out);
int ttx_exp_host(int64_t n, const double *x, double *out);

/* latency probe (bench.py's latency model of the sweep kernels): ns per dependent fp64 multiply [0], per dependent
 * multiply+add pair [1], per dependent L2 round trip (L1-bypassing pointer chase) [2], per dependent LDS read [3] and
 * per fp64 IEEE division in a dependent chain [4], each measured on one wave */
int ttx_k_latency_probe(int32_t device, double out[5]);

/* RCCL loop-back self-test on one device (pools without a second GPU): a one-rank communicator, a grouped ncclSend / ncclRecv of
 * a msg_bytes message to itself, the MAX all-reduce of 4 doubles and a SUM all-reduce of nsum doubles, all on a non-blocking
 * stream as in the multi-GPU data path (ttx_comm_init, lib/dmrgg.f90:763-958 replaced); TTX_OK when every byte came back */
int ttx_k_rccl_selftest(int32_t device, int64_t msg_bytes, int32_t nsum);

/* placement probe: the XCD (XCC_ID hardware register) on which each of `nblocks` workgroups of a plain 1-D launch
 * ran; the cluster sweep kernel relies on workgroups being dealt round-robin to the 8 XCDs */
int ttx_k_xcc_map(int32_t device, int32_t nblocks, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
