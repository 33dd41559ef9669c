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

#define TTX_MAXH 24        // half-step state slots (2*piv+2 <= TTX_MAXH)
#define TTX_MAXPART 512    // partial arg-max records per half-step (blocks per fiber)
#define TTX_BLK 256

struct StepState {
    int active;            // this group has a bond at this step of the sweep
    int p, r0, r1, r2, n1, n2;   // snapshot taken by the lottery kernel
    int ii, jj, kk, qq;    // current pivot candidate (1-based, reference names)
    int done, havecol, haverow, crs;
    int pending;           // 0 none, 1 column residual partials pending, 2 row residual partials pending
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
    StepState S[TTX_MAXH];
    Partial Pt[2][TTX_MAXPART];
};

struct DevProb {
    int d, RM, NM, G, NC;      // cores, max rank, max mode size, local groups, max cores per group
    int fun_id, piv, npar, ising_id;
    int nprocs;                // global number of groups
    int has_quad;
    double mvn_norm;           // sqrt((2 pi)^d det)
    double small_element, small_pivot;
    size_t SS, SW, CS;         // slab strides and core size
    const int *n;              // [d+2], 1-based
    const double *par;
    const double *aux;
    const double *quadw;       // [d+1][NM] padded, 1-based core index
    double *arg, *col, *row;   // [G][NC][CS]
    double *inv;               // [G][NC][RM*RM]   local bond bL = s - (first-1)
    int *vip;                  // [G][NC][4*RM]
    short *L, *R;              // [G][NC][d*RM]    L: bL = s-(first-1) ; R: bR = s-first
    int *r, *rr, *upd, *tape;  // [G][d+2] (tape x4)
    double *acol, *arow;       // [G][RM*NM]
    double *Tq;                // [G][NC][RM*RM]
    double *qpart;             // [G][RM*RM]
    int *ind0;                 // [d+2] initial cross index
    // per-sweep neighbour exchange (lib/dmrgg.f90:763-958 + lib/dmrggmp.f90:572-629).  Each group packs one
    // message for its right and one for its left neighbour; in* point at the message to consume: the
    // neighbour's send buffer when it lives on this GPU, an RCCL receive buffer otherwise.
    int *sendR_h, *sendL_h;        // [G][XH]      header: upd, tape(4), new rank
    int *sendR_i, *sendL_i;        // [G][d+2]     full multi-index of the new boundary pivot
    double *sendR_d, *sendL_d;     // [G][XD]      boundary fiber (+ inv for the right-going message)
    int **inL_h, **inR_h, **inL_i, **inR_i;   // [G] message from the left / right neighbour (nullptr: none)
    double **inL_d, **inR_d;
    double *red;                   // [G][4] amax, pivotmax, -pivotmin (for the MAX all-reduce, :852-870)
    size_t XD;
    int *qdims;                    // [G][2] (mym, myn) of each group's partial quadrature matrix
    double *qwork;                 // [(G+1)*RM*RM] scratch of the quadrature tree
    GroupState *gs;            // [G]
};
