/*
 * ttx_oracle.h -- CPU restatement (test oracle) of the reference's dtt_dmrgg greedy-cross sweep.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (libttx.so, the HIP engine) never links,
 * loads or calls anything in oracle/.
 *
 * Parity status: PINNED -- checked against golden logs produced by the genuine reference built in this
 * container from /root/reference (oracle/Makefile target `ref`, outputs in oracle/_ref/, fixtures in
 * tests/golden/ made by tests/golden/make_golden.sh) and against the analytic Ising values embedded in
 * the reference driver (test_crs_ising.f90:71-100).
 *
 * Every function cites the reference file:line it restates.  BLAS is an external dependency of the
 * reference (any conforming implementation; the golden logs were made with MKL 'sequential'); its
 * level-1/2/3 routines are restated here in the summation order of the published netlib reference BLAS
 * (idamax, ddot, dgemv 'n'/'t', dgemm 'nn', dscal, dasum).
 */
#ifndef TTX_ORACLE_H
#define TTX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* integrand ids shared with include/ttx.h */
enum {
    TTXO_FUN_ISING = 1,   /* dfunc_ising_discr, test_crs_ising.f90:176-218; par(2n+1) selects C/D/E   */
    TTXO_FUN_STDNORM = 2, /* integrand, test_crs_stdnorm.f90:154-170                                   */
    TTXO_FUN_MVN = 3,     /* integrand -> mvn_pdf, test_crs_mvn.f90:156-172, lib/mvn_pdf.f90:63-83     */
    TTXO_FUN_USER = 4     /* the caller's own `fun` (lib/dmrgg.f90:18): ttxo_problem.user                */
};
/* the reference's callback interface fun(m, ind, n, par), Fortran convention (everything by reference) */
typedef double (*ttxo_user_fun)(const int32_t *m, const int32_t *ind, const int32_t *n, const double *par);

typedef struct ttxo_sweep_rec {
    int32_t it;        /* sweep number, 0 = initial cross                     */
    int32_t dir;       /* 0 '::', 1 '>>', 2 '<<'                              */
    double erank;      /* erank(arg) as seen by rank 0, lib/tt.f90:1228-1245  */
    int64_t neval;     /* summed over ranks, lib/dmrgg.f90:273,963            */
    double val;        /* per-sweep quadrature value (0 when no quad)         */
    double amax;       /* after the per-sweep MAX allreduce                   */
    double pivotmax;   /* idem (-1 when nothing was accepted)                 */
    double pivotmin;
} ttxo_sweep_rec;

typedef struct ttxo_problem {
    int32_t d;               /* number of cores (reference: l=1, m=d)                                */
    const int32_t *n;        /* mode sizes n[0..d-1]                                                 */
    int32_t fun_id;          /* TTXO_FUN_*                                                           */
    const double *par;       /* integrand parameters (nodes, weights, id), as the reference's par(*) */
    int32_t npar;
    const double *aux;       /* MVN: mu[d], inv_cov[d*d] column-major, det; else NULL                */
    int32_t naux;
    const double *quadw;     /* rank-1 quadrature TT: d blocks of n[k] weights, or NULL              */
    double accuracy;         /* <0: absent                                                           */
    int32_t maxrank;         /* <=0: absent                                                          */
    int32_t piv;             /* -1 full, 0 lottery only, >0 rook half-steps (default 3)              */
    double tru;              /* analytic value                                                       */
    int32_t has_tru;
    int32_t nproc;           /* virtual MPI ranks (>=1, < d)                                         */
    const int32_t *mybonds;  /* own(0:nproc) 1-based bonds, or NULL -> share()                       */
    int32_t verbose;         /* 1: print the reference's per-sweep lines to stdout                   */
    const double *draws;     /* optional replay tape of uniform draws (NULL -> flang-compatible LCG)  */
    int64_t ndraws;
    ttxo_user_fun user;      /* TTXO_FUN_USER: the integrand; par is handed to it untouched           */
} ttxo_problem;

typedef struct ttxo_result {
    int32_t d;
    int32_t nsweeps;           /* number of records (initial cross included)                        */
    ttxo_sweep_rec *sweeps;    /* [nsweeps]                                                          */
    int32_t *tapes;            /* [nsweeps-1][d+1][4] owner tapes per sweep (bond p at [p], 1-based) */
    int32_t *r;                /* final ranks r[0..d] (rank 0's view == global after last exchange) */
    double **cores;            /* finalised cores k=0..d-1, column-major r[k] x n[k] x r[k+1]        */
    int64_t neval;
    double value;              /* dtt_quad(tt, qq) after finalisation (0 when no quad)               */
    double seconds;            /* wall time of the dmrgg call                                        */
    uint64_t rngpos;           /* uniform draws consumed by rank 0 (position of the run-time RNG stream) */
} ttxo_result;

/* lib/dmrgg.f90:11-1050 (+ lib/dmrggmp.f90:572-629 for the right-going boundary exchange). */
int ttxo_dmrgg(const ttxo_problem *prob, ttxo_result *res);
void ttxo_free_result(ttxo_result *res);

/* lib/dmrgg.f90:1081-1166 dtt_accchk on the finalised cores of `res` (1 rank): nlot random samples drawn from
 * the same run-time RNG stream (irnd, lib/rnd.f90:83-88), element evaluation dtt_ijk (lib/tt.f90:630-652) */
void ttxo_accchk(const ttxo_problem *prob, const ttxo_result *res, int nlot, double *einf, double *efro, double *ainf,
                 double *afro, int32_t *pivot);
/* tt_lib utilities on a TT with compact column-major cores (oracle/ttx_oracle_tt.c): dtt_ort lib/tt.f90:130,
 * dtt_svd :307 (tol, rmax <= 0: absent), dtt_norm :1074 (tol < 0: absent), dtt_dot :1155, dtt_ijk :630 */
typedef struct ttxo_tt { int32_t d; int32_t *n; int32_t *r; double **cores; } ttxo_tt;   /* n[d], r[d+1], cores[d] (malloc'd) */
void ttxo_tt_ort(ttxo_tt *t);
void ttxo_tt_svd(ttxo_tt *t, double tol, int rmax);
double ttxo_tt_norm(const ttxo_tt *t, double tol);
double ttxo_tt_dot(const ttxo_tt *x, const ttxo_tt *y);
double ttxo_tt_ijk(const ttxo_tt *t, const int32_t *ind);
void ttxo_tt_zquad(const ttxo_tt *t, const double *w /* interleaved re,im */, double *out /* re, im */);   /* lib/dmrgg.f90:1418 */
ttxo_tt *ttxo_tt_new(int d, const int32_t *n, const int32_t *r);
void ttxo_tt_free(ttxo_tt *b);
/* lib/quad.f90:97-131 */
void ttxo_lgwt(int n, double *x, double *w);
/* lib/default.f90:78-97 */
void ttxo_share(int first, int last, int nproc, int32_t *own);
/* lib/mvn_pdf.f90:21-60,85-111 : mu, inv_cov (column-major d*d), det for the driver's r=0,T=1 */
void ttxo_mvn_init(int d, double r, double T, double *aux /* d + d*d + 1 */);
/* integrand evaluation (1-based ind), test_crs_*.f90 */
double ttxo_fun(int fun_id, int m, const int32_t *ind, const int32_t *n, const double *par, const double *aux);
/* bit-neutral shortcut of the Ising D/E product for nodes in [0,1] (see ttx_oracle.c); off by default */
void ttxo_set_unit_skip(int on);
/* flang runtime random_number stream (unseeded): draw #k (0-based) */
double ttxo_flang_draw(uint64_t k);
/* lib/lr.f90:124-154 */
void ttxo_lual(int m, int r, const double *g, double *col, int from);
void ttxo_luar(int n, int r, const double *g, double *row, int from);
/* lib/rnd.f90:105-144 */
void ttxo_lottery2(int npnt, int m, int n, const double *wcol, const double *wrow, const double *d, int32_t *points);
/* driver parameter set-up: test_crs_ising.f90:102-144, test_crs_mvn.f90:76-118, test_crs_stdnorm.f90:72-112.
 * kind: 'c','d','e' (ising; m = integral index, d = m-1), 's' stdnorm, 'm' mvn (m = dimension).
 * par must hold 2n+1 doubles, quadw d*n doubles; returns d and *tru (0 when the driver has none). */
int ttxo_driver_setup(char kind, int m, int n, double *par, double *quadw, double *tru, double *acc, int *rescale);

#ifdef __cplusplus
}
#endif
#endif
