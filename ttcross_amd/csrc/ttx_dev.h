// ttx_dev.h -- device-side data model of the MI355X TT-cross engine (shared by kernels and host code).
//
// HBM layout (per bond group g = "MPI rank" of the reference; all arrays are allocated once at max rank RM,
// so growing a rank is an in-place append instead of the reference's realloc+copy, lib/dmrgg.f90:650-749):
//
//   arg / col cores  (raw fibers / U^-1-scaled factors, lib/dmrgg.f90:46,243):  A[(i + RM*j) + SS*s]
//        i < r(p-1) left rank (fastest, padded to RM), j < n(p) mode, s < r(p) right rank (slowest,
//        SS = RM*NM).  A column half-step reads r(p) slabs of r(p-1)*n(p) contiguous-in-(i,j) doubles:
//        consecutive lanes -> consecutive addresses.  A new pivot appends slab s = r(p).
//   row cores  (L^-1-scaled factors, lib/dmrgg.f90:244):  W[(k + NM*q) + SW*s],  SW = NM*RM
//        stored TRANSPOSED w.r.t. the reference's (s,k,q) so that the row half-step is the same coalesced
//        slab sweep; s < r(p) is the slowest index, appended in place.
//   inv   packed incremental LU per bond (lib/lr.f90:124-154 layout), RM*RM doubles
//   vip   (i,j,k,q) of each pivot per own bond (lib/dmrgg.f90:134), 4*RM int32
//   L / R flattened index tables: L[bond s][dim-1][t] = mode index of dim <= s for left pivot t of bond s,
//        R[bond s][x][t] = mode index of dim s+1+x.  They replace the O(d) dependent loads of the
//        reference's nested walk dmrgg_fun (lib/dmrgg.f90:1062-1075) by independent, coalesced int16 loads.
#pragma once
#include <stdint.h>
#include "ttx_cdf.h"
#define TTX_TABSEG 64

#define TTX_MAXH 24        // half-step state slots (2*piv+2 <= TTX_MAXH)
#define TTX_MAXPART 512    // partial arg-max records per half-step (blocks per fiber)
#define TTX_BLK 256
#define TTX_FS 8          // rows of the per-pivot scalar tables of the fast evaluators (ttx_fast.h)
#define XH 8             // ints in a message header
// layout of the per-sweep summary (doubles): scalars, then initval per global group, then (r, tape[4]) per bond
#define SUM_NEVAL 0
#define SUM_AMAX 1
#define SUM_PMAX 2
#define SUM_PMIN 3
#define SUM_BYTES 4
#define SUM_NRESID 5
#define SUM_VAL 6
#define SUM_HDR 8

struct StepState {
    int active;            // this group has a bond at this step of the sweep
    int p, r0, r1, r2, n1, n2;   // snapshot taken by the lottery kernel
    int ii, jj, kk, qq;    // current pivot candidate (1-based, reference names)
    int done, havecol, haverow, crs;
    int pending;           // 0 none, 1 column residual partials pending, 2 row residual partials pending
    int npart;             // number of partial arg-max records behind `pending`
    double pivot;
};

struct Partial {
    double absmax;         // < 0 : empty
    double val;
    int idx;
    int pad;
};

struct GroupState {
    int first, last;       // own bonds (1-based); own cores first..last+1
    int gglobal;           // global group index ("me")
    int pad0;
    double amax, pivotmax, pivotmin, pivotmax_prev;
    long long neval;
    unsigned long long rngpos;
    double val;            // per-sweep quadrature value (group 0)
    double initval;        // initial-cross value factor of this group
    double bytes_half;     // algorithmic bytes moved by the half-step kernel, for the roofline
    long long n_resid;     // half-steps that computed a residual
#ifdef TTX_STAMPS
    long long stamp[2][16]; long long nstamp[2];   // debug build: accumulated wall_clock64 deltas per phase
#endif
    StepState S[TTX_MAXH];
    Partial Pt[2][TTX_MAXPART];
};

// partial result of one workgroup of the lottery kernel (candidates [b*CH, (b+1)*CH) of a group): the block's first
// arg-max of |residual| with its candidate (i,j,k,q), and the block's max |f|; the last block to finish combines them
struct LotPart { double ab, bv, ma; int il, i, j, k, q, pad; };

#define TTX_CLMAX 16            // most workgroups one bond group's cluster may have
#define TTX_CLREC (4 * TTX_CLMAX) // arg-max records per group and half-step parity: one per WAVE of the cluster (4 waves per workgroup)
struct ClPart { double ab, bb, mx; int ix, pad; };

struct DevProb {
    int d, RM, NM, G, NC;      // cores, max rank, max mode size, local groups, max cores per group
    int fun_id, piv, npar, ising_id;
    // TTX_ARITH=fast (ttx_fast.h): the integrand may be evaluated in any association of its products and sums (results agree
    // with the exact mode to rounding, not bit for bit).  Effective for Ising D/E with all nodes in [0,1] and for mvn; 0 elsewhere.
    int arith;
    // per-bond-step tables of the fast evaluators (k_fast_tables), per group; side 0 = left pivots of bond p-1, side 1 = right
    // pivots of bond p+1.  fNear [G][FD][RM]: Ising: decay vector of a pivot (product of the t dims nearest to the bond, t = row);
    // mvn: Y = S d (row = dimension).  fDv [G][FD][RM] (mvn): the pivot's difference vector.  fPiv [G][FS][RM]: scalars per pivot.
    double *fNear[2], *fDv[2], *fPiv[2];
    int FD;
    // fpersist (Ising D/E): the tables are kept PER BOND ([G][NC] slots each side) and maintained incrementally -- a new pivot's entry is
    // derived from its parent's in O(cut length) when it is accepted (k_accept), entries of rank-1 starts and of the neighbours'
    // boundary pivots are made from scratch; 0 (mvn): one table per group, rebuilt per bond step by k_fast_tables
    int fpersist;
    const double *auxS;        // mvn, fast: (inv_cov + inv_cov') / 2, d x d
    int nprocs;                // global number of groups
    int has_quad;
    double mvn_norm;           // sqrt((2 pi)^d det)
    double small_element, small_pivot;
    size_t SS, SW, CS;         // slab strides and core size
    const int *n;              // [d+2], 1-based
    const double *par;
    const double *aux;
    double *deTL, *deUL, *deTR;  // Ising D/E: per-bond pair-factor tables [G][RM][de_npair], [G][RM][d+1], [G][RM][de_npair] (k_de_tables)
    int de_npair;
    int de_unit;                 // all nodes of par lie in [0,1]: the exact short division fdiv_unit applies
    int de_cut;                  // ... and the tables are the COMPACT ones of k_de_ctables: every row of the pair triangle ends at the unit cut
    int *deCL, *deCR;            // [G][RM][d+1]: entries per row of a pivot's compact table; [..][d] = their total
    const double *auxT;        // mvn: inv_cov transposed, auxT[j + d*i] = inv_cov(i,j) (same values, row walk contiguous)
    const double *quadw;       // [d+1][NM] padded, 1-based core index
    double *arg, *col, *row;   // [G][NC][CS]
    double *inv;               // [G][NC][RM*RM]   local bond bL = s - (first-1)
    int *vip;                  // [G][NC][4*RM]
    short *L, *R;              // [G][NC][d*RM]    L: bL = s-(first-1) ; R: bR = s-first
    int *r, *rr, *upd, *tape;  // [G][d+2] (tape x4)
    double *acol, *arow;       // [G][RM*NM]
    double *Tq;                // [G][NC][RM*RM]
    int *ind0;                 // [d+2] initial cross index
    // per-sweep neighbour exchange (lib/dmrgg.f90:763-958 + lib/dmrggmp.f90:572-629).  Each group packs one
    // contiguous message for its right and one for its left neighbour:
    //   [XH int header: upd, new rank][d+2 int: flattened multi-index of the new boundary pivot]
    //   [XD double: boundary fiber (RM*NM) + packed inv (RM*RM, right-going only)]
    // inL/inR[g] point at the message group g consumes: the neighbour's send buffer when it lives on this GPU,
    // the RCCL receive buffer otherwise (nullptr: no neighbour).
    char *msgR, *msgL;             // [G][MSZ]
    char **inL, **inR;             // [G]
    size_t MSZ, IOFF, XD;          // message bytes, byte offset of the double payload, doubles in the payload
    double *red;                   // [G][4] amax, pivotmax, -pivotmin per group (MAX all-reduce, :852-870)
    double *redsend, *redrecv;     // [4] this GPU's entry / the job-wide result
    double *qsend, *qall;          // [nprocs*RM*RM + 2*nprocs] partial matrices + dims of ALL groups (SUM all-reduce)
    double *qwork;                 // [(nprocs+2)*RM*RM] scratch of the quadrature tree
    double *qscr;                  // [G][2*RM*RM] chain matrices of k_quad_chain when they do not fit the LDS (maxrank >= 98); else nullptr
    double *sumsend, *sumrecv;     // per-sweep job summary (SUM all-reduce): see SUM_* offsets
    int g0;                        // global index of local group 0
    Partial *pfull;                // [G][NM*RM*nfb] partial arg-max records of the full-superblock search (piv = -1)
    // piv = -1 as ONE dense step (TTX_FULLPIV=mfma): the superblock is evaluated once into sb [G][(RM*NM)^2], its residual
    // against col(p) x row(p+1) is taken by an fp64 MFMA GEMM fused with the arg-max (k_full_gemm_argmax), one partial
    // record per 64x64 tile in pfull2 [G][fp_tiles]
    double *sb; Partial *pfull2; int fp_mfma, fp_tiles;
    int nfb;                       // fiber blocks per half-step launch
    const ttx_cdfseg *cdf_tab;     // [cdf_kmax+1][TTX_TABSEG] lottery CDF segments for every K (nullptr: build in-kernel)
    const int *cdf_ns;             // [cdf_kmax+1]
    int cdf_kmax;
    GroupState *gs;            // [G]
    // device-side stopping rule (lib/dmrgg.f90:1011-1019) so that the host can enqueue the next sweep before it has
    // read the summary of the current one: ctl[0] stop, ctl[1] strike counter, ctl[2] stop as seen when the
    // quadrature of the sweep was forked to its own stream
    int *ctl;
    int *rq;                       // [G][d+2] ranks at the end of the sweep, for the forked quadrature
    double accuracy; int maxrank;
    // cluster sweep kernel (ttx_cluster.h): per-group barrier counters, abort flag, per-block partial arg-max records
    // lottery spread over several workgroups per group (heavy integrands): [G][lot_nb] partials + arrival counters [G]
    struct LotPart *lotp;
    unsigned *lot_ctr;
    int lot_nb;
    // Ising D/E lottery in three launches (pick / one candidate per wave / fold): candidates [G][lot_max][4], values [G][lot_max]
    int *lotc; double *lotf; int lot_max;
    int bnd_wave;                  // boundary corner entries of Ising D/E by one wave each (k_exch_boundary has the LDS for it)
    // host-evaluated integrand (TTX_FUN_HOST, the reference's user callback `fun`, lib/dmrgg.f90:18): every kernel that
    // evaluates runs twice.  Pass 1 (hostpass = 1) writes the multi-index of each point it needs to hidx[slot][d], raises
    // hreq[slot] and stops before any side effect; the host calls `fun`; pass 2 (hostpass = 2) reads hval[slot].
    // All three arrays live in pinned host memory; a group's slots are [g*HS, (g+1)*HS).
    int hostpass, HS;
    int zbase;                     // full pivoting with a host integrand: first superblock column (k,q) of this launch (one column per launch)
    short *hidx;
    double *hval;
    unsigned char *hreq;
    unsigned *cl_ctr;              // [G]
    int *cl_abort;                 // [1]
    int cl_test_abort;             // test hook: block (group 0, block 0) raises the abort flag in launch number cl_test_abort (0: never)
    struct ClPart *cl_part;        // [2][G][TTX_CLREC]
    long long *dbg;                // -DTTX_STAMPS builds: per-wave cycle stamps of one bond step of the cluster kernel (group 0)
};
