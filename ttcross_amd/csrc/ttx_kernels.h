// ttx_kernels.h -- CDNA4 (gfx950) kernels of the dtt_dmrgg sweep.  Included by ttx_engine.hip only.
// Compiled with -ffp-contract=off: every product and sum below is an individual IEEE fp64 rounding, in the
// order of the netlib reference BLAS the oracle restates, so pivot paths are reproducible bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include "ttx_cdf.h"
#include "ttx_dev.h"
#include "ttx_exp.h"

#ifdef TTX_STAMPS
#define STAMP_DECL const bool t_me = (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0); long long t_prev = wall_clock64(); int t_slot = 0
#define STAMP(gs, k) do { __syncthreads(); if (t_me) { long long t_now = wall_clock64(); (gs).stamp[k][t_slot++] += t_now - t_prev; t_prev = t_now; } } while (0)
#define STAMP_END(gs, k) do { if (t_me) (gs).nstamp[k]++; } while (0)
#else
#define STAMP_DECL
#define STAMP(gs, k)
#define STAMP_END(gs, k)
#endif
#define FUN_ISING 1
#define FUN_STDNORM 2
#define FUN_MVN 3
#define FUN_HOST 4     // values come from the host (the user callback of lib/dmrgg.f90:18), see DevProb::hostpass

// ------------------------------------------------------------------------------------------------
// address helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double *core_ptr(const DevProb &P, double *base, int g, int p, int first)
{ return base + ((size_t)g * P.NC + (p - first)) * P.CS; }
__device__ __forceinline__ double *inv_ptr(const DevProb &P, int g, int s, int first)   // bond s in first-1..last
{ return P.inv + ((size_t)g * P.NC + (s - first + 1)) * (size_t)P.RM * P.RM; }
__device__ __forceinline__ int *vip_ptr(const DevProb &P, int g, int s, int first)
{ return P.vip + ((size_t)g * P.NC + (s - first + 1)) * (size_t)4 * P.RM; }
__device__ __forceinline__ short *L_ptr(const DevProb &P, int g, int s, int first)      // bond s in first-1..last
{ return P.L + ((size_t)g * P.NC + (s - first + 1)) * (size_t)P.d * P.RM; }
__device__ __forceinline__ short *R_ptr(const DevProb &P, int g, int s, int first)      // bond s in first..last+1
{ return P.R + ((size_t)g * P.NC + (s - first)) * (size_t)P.d * P.RM; }

#include "ttx_fast.h"     // TTX_ARITH=fast: re-associated evaluation of the heavy integrands (tolerance-checked mode)

// ------------------------------------------------------------------------------------------------
// integrands (reference drivers' callbacks).  idx(s), s = 1..m, returns the 1-based mode index of dim s.
// ------------------------------------------------------------------------------------------------
template <class IDX>
__device__ __forceinline__ double f_ising(int id, int m, int n1, const double *par, IDX idx, bool unit = false)
{
    // test_crs_ising.f90:176-218.  unit (all nodes in [0,1]): a row of the pair triangle ends where the running product has reached
    // 2^-54 -- from there on every factor is EXACTLY 1 in fp64 (see f_ising_de), so leaving them out changes no bit
    const double *nodes = par - 1, *weights = par + n1 - 1;
    double a = 1.0, b = 0.0, f;
    if (id == 2 || id == 3) {
        for (int i = 0; i <= m; i++) {
            double uij = 1.0;
            for (int j = i + 1; j <= m; j++) {
                uij = uij * nodes[idx(j)];
                if (unit && uij <= 0x1p-54) break;
                double t = (uij - 1.0) / (uij + 1.0);
                a = a * (t * t);
            }
        }
    }
    if (id == 1 || id == 2) {
        double v = 1.0, w = 1.0, vk = 1.0, wk = 1.0;
        for (int i = 1; i <= m; i++) {
            vk = vk * nodes[idx(m - i + 1)];
            wk = wk * nodes[idx(i)];
            v = v + vk;
            w = w + wk;
        }
        b = 1.0 / (v * w);
    }
    f = (id == 1) ? 2 * b : (id == 2) ? 2 * a * b : 2 * a;
    for (int i = 1; i <= m; i++) f = f * weights[idx(i)];
    return f;
}

// TTX_ARITH=fast, any multi-index (initial cross, boundary corners, dtt_accchk): rho by de_fast_rho, then the b-part and
// the weights as above; f = (2 rho b w_1 ... w_m) rho keeps rho^2 out of the intermediate results (more range than a itself)
template <class IDX>
__device__ __forceinline__ double f_ising_fast(int id, int m, int n1, const double *par, IDX idx)
{
    const double *nodes = par - 1, *weights = par + n1 - 1;
    const double rho = de_fast_rho(m, nodes, idx);
    double b = 1.0;
    if (id == 2) {
        double v = 1.0, w = 1.0, vk = 1.0, wk = 1.0;
        for (int i = 1; i <= m; i++) {
            vk = vk * nodes[idx(m - i + 1)];
            wk = wk * nodes[idx(i)];
            v = v + vk;
            w = w + wk;
        }
        b = 1.0 / (v * w);
    }
    double f = 2 * rho * b;
    for (int i = 1; i <= m; i++) f = f * weights[idx(i)];
    return f * rho;
}

template <class IDX>
__device__ __forceinline__ double f_stdnorm(int m, const double *par, IDX idx)
{
    // test_crs_stdnorm.f90:154-170
    double s = 0.0;
    for (int i = 1; i <= m; i++) { double x = par[idx(i) - 1]; s = s + x * x; }
    return ttx_exp(-s);
}

template <class IDX>
__device__ __forceinline__ double f_mvn(int m, const double *par, const double *aux, double norm, IDX idx)
{
    // test_crs_mvn.f90:156-172 + lib/mvn_pdf.f90:63-83
    const double *mu = aux, *ic = aux + m;
    double ex = 0.0;
    for (int i = 1; i <= m; i++) {
        double di = par[idx(i) - 1] - mu[i - 1];
        for (int j = 1; j <= m; j++) {
            double dj = par[idx(j) - 1] - mu[j - 1];
            ex = ex + di * ic[(i - 1) + (size_t)m * (j - 1)] * dj;
        }
    }
    return ttx_exp(-0.5 * ex) / norm;
}

// The same quadratic form from rows of DIFFERENCES x - mu staged in LDS: dims 1..A from da, dim A+1 = d1, (NS == 2:
// dim A+2 = d2,) the remaining nb dims from db.  Term order and association are those of f_mvn / mvn_pdf
// (lib/mvn_pdf.f90:74-80): ex = ex + (diff(i) * inv_cov(i,j)) * diff(j), i outer, j inner -- only the O(d) index
// decoding per term is gone (the inverse covariance is read with wave-uniform addresses).
template <int NS, class FN>
__device__ __forceinline__ void mvn_dims(const double *da, int A, double d1, double d2, const double *db, int nb, FN fn)
{
#pragma unroll 8
    for (int t = 0; t < A; t++) fn(da[t]);
    fn(d1);
    if (NS == 2) fn(d2);
#pragma unroll 8
    for (int t = 0; t < nb; t++) fn(db[t]);
}
template <int NS>
__device__ __forceinline__ double f_mvn_rows(int m, const double *icT, double norm, const double *da, int A, double d1, double d2,
                                             const double *db)
{
    const int nb = m - A - NS;
    double ex = 0.0;
    int i = 0;
    mvn_dims<NS>(da, A, d1, d2, db, nb, [&](double di) {
        const double *row = icT + (size_t)m * i;       // icT[j + m*i] = inv_cov(i,j): the j loop walks contiguous memory
        int j = 0;
        mvn_dims<NS>(da, A, d1, d2, db, nb, [&](double dj) { ex = ex + di * row[j] * dj; j++; });
        i++;
    });
    return ttx_exp(-0.5 * ex) / norm;
}

// host integrand: pass 1 records the multi-index of the point in slot `slot`, pass 2 returns the host's value
template <class IDX>
__device__ __forceinline__ double f_host(const DevProb &P, IDX idx, long slot)
{
    if (P.hostpass == 1) {
        short *row = P.hidx + (size_t)slot * P.d;
        for (int s = 1; s <= P.d; s++) row[s - 1] = (short)idx(s);
        P.hreq[slot] = 1;
        return 0.0;
    }
    return P.hval[slot];
}
#define HOST_PASS1(FUN, P) ((FUN) == FUN_HOST && (P).hostpass == 1)

template <int FUN, class IDX>
__device__ __forceinline__ double eval_fun(const DevProb &P, const double *par, IDX idx, long slot = -1)
{
    if (FUN == FUN_HOST) return f_host(P, idx, slot);
    if (FUN == FUN_ISING) return (P.arith && P.ising_id != 1) ? f_ising_fast(P.ising_id, P.d, P.n[1], par, idx) : f_ising(P.ising_id, P.d, P.n[1], par, idx, P.de_unit != 0);
    if (FUN == FUN_STDNORM) return f_stdnorm(P.d, par, idx);
    return f_mvn(P.d, par, P.aux, P.mvn_norm, idx);
}

// Index source of one superblock entry: dims 1..A from pa[0..A-1], dim A+1 = self, dims A+2..m from pb[0..].
// The flattened tables are staged in LDS with one CONTIGUOUS row per thread; this replaces the reference's
// nested vip walk (dmrgg_fun, lib/dmrgg.f90:1062-1075).
struct Src3 {
    const short *pa; int A, self; const short *pb;
    __device__ __forceinline__ int operator()(int s) const
    { return (s <= A) ? (int)pa[s - 1] : (s == A + 1) ? self : (int)pb[s - A - 2]; }
};
// the same with two explicit dims (A+1 = s1, A+2 = s2): a lottery candidate (i,j,k,q) read straight from the
// transposed pivot tables, pa = row of left pivot i, pb = row of right pivot q
struct Src4 {
    const short *pa; int A, s1, s2; const short *pb;
    __device__ __forceinline__ int operator()(int s) const
    { return (s <= A) ? (int)pa[s - 1] : (s == A + 1) ? s1 : (s == A + 2) ? s2 : (int)pb[s - A - 3]; }
};

// 8 indices with one 128-bit LDS read (rows are 16-byte aligned and padded with the valid index 1)
struct __align__(16) Short8 { short v[8]; };
__device__ __forceinline__ Short8 ld8(const short *p) { return *reinterpret_cast<const Short8 *>(__builtin_assume_aligned(p, 16)); }

// One dependent recurrence over n table entries addressed by a 16-byte-aligned, padded index row: per chunk of
// 8 dims one 128-bit index read and 8 independent table reads precede the 8 dependent fp64 steps; only the
// chunk that holds the row's end is predicated.
template <bool DESC, class STEP>
__device__ __forceinline__ void chain8(const short *p, int n, const double *tab, STEP step)
{
    if (n <= 0) return;
    const int nfull = n >> 3, rem = n & 7;
    if (DESC && rem) {
        Short8 ix = ld8(p + 8 * nfull); double x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = tab[ix.v[k]];
#pragma unroll
        for (int k = 7; k >= 0; k--) if (k < rem) step(x[k]);
    }
#pragma unroll 2
    for (int ch = 0; ch < nfull; ch++) {
        const int c = 8 * (DESC ? nfull - 1 - ch : ch);
        Short8 ix = ld8(p + c); double x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = tab[ix.v[k]];
#pragma unroll
        for (int k = 0; k < 8; k++) step(x[DESC ? 7 - k : k]);
    }
    if (!DESC && rem) {
        Short8 ix = ld8(p + 8 * nfull); double x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = tab[ix.v[k]];
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < rem) step(x[k]);
    }
}

// ---- Ising D / E (id 2, 3) over aligned index rows ----------------------------------------------------------
// The O(d^2) product a = prod_{i<j} ((u_ij-1)/(u_ij+1))^2 (test_crs_ising.f90:186-195) is one IEEE division per
// step on a sequential chain; the generic accessor exposed two dependent LDS latencies (index, node value) plus
// scalar control flow in every step.  Here the dims are walked as branch-free ranges of the staged rows with
// 8-wide chunk loads (one 128-bit index read + 8 independent table reads ahead of the 8 dependent steps).
// Layout: dims 1..A from pa, then NS explicit dims (s1[, s2]), then dims A+NS+1..m from pb.
template <class STEP>
__device__ __forceinline__ void seg_range(const short *p, int lo, int hi, const double *tab, STEP step)
{   // elements [lo, hi) of an aligned, padded row, ascending
#pragma unroll 2
    for (int c = lo & ~7; c < hi; c += 8) {
        Short8 ix = ld8(p + c); double x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = tab[ix.v[k]];
        if (c >= lo && c + 8 <= hi) {
#pragma unroll
            for (int k = 0; k < 8; k++) step(x[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) if (c + k >= lo && c + k < hi) step(x[k]);
        }
    }
}
template <int NS, class STEP>
__device__ __forceinline__ void dims_asc(const short *pa, int A, int s1, int s2, const short *pb, int m, const double *tab,
                                         int jlo, int jhi, STEP step)
{   // dims jlo..jhi (1-based, inclusive), ascending
    if (jlo > jhi) return;
    const int a1 = jhi < A ? jhi : A;
    if (jlo <= a1) seg_range(pa, jlo - 1, a1, tab, step);
    if (jlo <= A + 1 && A + 1 <= jhi) step(tab[s1]);
    if (NS == 2 && jlo <= A + 2 && A + 2 <= jhi) step(tab[s2]);
    const int b0 = jlo > A + NS + 1 ? jlo : A + NS + 1;
    if (b0 <= jhi) seg_range(pb, b0 - A - NS - 1, jhi - A - NS, tab, step);
}
// b-part (id 2) and the weights of the D / E integrand once the pair product a is known (:197-218)
template <int NS>
__device__ __forceinline__ double de_finish(int id, double a, int m, int n1, const double *par, const short *pa, int A, int s1, int s2, const short *pb);

// IEEE-exact fp64 division for operands in the "unit" range of the Ising integrands: with all nodes in [0,1] every
// running product u lies in [0,1], so d = u+1 is in [1,2] and n = u-1 in [-1,0].  The compiler's a/b is
// v_div_scale x2, v_rcp, 4 fma, mul, fma, v_div_fmas, v_div_fixup; for such operands neither v_div_scale rescales nor
// does v_div_fixup change anything but the sign it would set anyway, so the Newton core alone -- the same v_rcp_f64 and
// the same fused operations in the same order -- returns the identical bits with four instructions fewer.
// (P.de_unit is set by the host only after checking the nodes; any other node set takes the plain division.)
__device__ __forceinline__ double fdiv_unit(double n, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = n * r;
    const double rem = __builtin_fma(-d, q, n);
    return __builtin_fma(rem, r, q);
}
template <bool FAST>
__device__ __forceinline__ double de_t2(double u)
{   // ((u-1)/(u+1))^2, test_crs_ising.f90:190-191
    const double n = u - 1.0, d = u + 1.0;
    const double t = FAST ? fdiv_unit(n, d) : n / d;
    return t * t;
}
// W independent pair factors at once, written STAGE BY STAGE.  One division is a chain of nine dependent fp64
// instructions, most of them FMAs with two or three VGPR operands, which a lone wave issues only every ~8.5 cycles
// (profiles/r02_probe_dpp.txt: 3.5 ns per instruction against 1.9 ns with constant operands).  Measured per pair on one
// wave: 85 ns with one chain, 57 with two, 42 with four (what the compiler makes of four de_t2 calls), 35 with eight
// chains interleaved -- the SIMD's own limit with several resident waves is 29 ns.  Same operations per element as
// de_t2 / fdiv_unit, so the same bits.
template <bool FAST, int W>
__device__ __forceinline__ void de_t2xw(const double (&u)[W], double (&t)[W])
{
    if (!FAST) {
#pragma unroll
        for (int k = 0; k < W; k++) t[k] = de_t2<false>(u[k]);
        return;
    }
    double n[W], d[W], r[W], e[W], q[W];
#pragma unroll
    for (int k = 0; k < W; k++) { n[k] = u[k] - 1.0; d[k] = u[k] + 1.0; }
#pragma unroll
    for (int k = 0; k < W; k++) r[k] = __builtin_amdgcn_rcp(d[k]);
#pragma unroll
    for (int k = 0; k < W; k++) e[k] = __builtin_fma(-d[k], r[k], 1.0);
#pragma unroll
    for (int k = 0; k < W; k++) r[k] = __builtin_fma(r[k], e[k], r[k]);
#pragma unroll
    for (int k = 0; k < W; k++) e[k] = __builtin_fma(-d[k], r[k], 1.0);
#pragma unroll
    for (int k = 0; k < W; k++) r[k] = __builtin_fma(r[k], e[k], r[k]);
#pragma unroll
    for (int k = 0; k < W; k++) q[k] = n[k] * r[k];
#pragma unroll
    for (int k = 0; k < W; k++) e[k] = __builtin_fma(-d[k], q[k], n[k]);
#pragma unroll
    for (int k = 0; k < W; k++) q[k] = __builtin_fma(e[k], r[k], q[k]);
#pragma unroll
    for (int k = 0; k < W; k++) t[k] = q[k] * q[k];
}
template <bool FAST>
__device__ __forceinline__ void de_t2x4(double u1, double u2, double u3, double u4, double &t1, double &t2, double &t3, double &t4)
{
    const double u[4] = {u1, u2, u3, u4}; double t[4];
    de_t2xw<FAST, 4>(u, t);
    t1 = t[0]; t2 = t[1]; t3 = t[2]; t4 = t[3];
}

// Pair product of an element (left pivot row il | s1 | s2 | right pivot row q) from the per-bond tables built by
// k_de_tables: the factor ((u_ij-1)/(u_ij+1))^2 of a pair depends only on the dims i+1..j, so every pair that lies
// entirely in the left pivot's dims (TL) or entirely in the right pivot's dims (TR) is shared by all elements through
// that pivot and is only MULTIPLIED here -- in the reference's order, so the product is bit-identical; the IEEE
// divisions are left for the pairs that span the bond (about a third of all pairs on average).
// TLc / ULc / TRc point at the pivot's own CONTIGUOUS row of the [pivot][pair] tables (this thread streams it);
// rix[ro .. ro+B): the right pivot's index entries (rix itself 16-byte aligned and padded).
template <bool FAST>
__device__ __forceinline__ double de_pairs_tab(int m, const double *nodes, int A, const double *TLc, const double *ULc,
                                               int s1, int s2, const short *rix, int ro, const double *TRc)
{
    const int B = m - A - 2;
    const double x1 = nodes[s1], x2 = nodes[s2];
    double a = 1.0;
    size_t pr = 0;
    auto run = [&](double u) {                          // the bond-spanning tail of a row: ... s2, right dims
        auto step = [&](double xv) { u = u * xv; a = a * de_t2<FAST>(u); };
        step(x2);
        seg_range(rix, ro, ro + B, nodes, step);
    };
    for (int i = 0; i <= A; i++) {
        const int cnt = A - i;
#pragma unroll 8
        for (int t = 0; t < cnt; t++) a = a * TLc[pr + t];
        pr += cnt;
        double u = ULc[i];
        u = u * x1; a = a * de_t2<FAST>(u);
        run(u);
    }
    run(1.0);                                           // i = A+1: starts after s1
    const size_t nr = (size_t)B * (B + 1) / 2;          // i >= A+2: pairs inside the right pivot's dims
#pragma unroll 8
    for (size_t t = 0; t < nr; t++) a = a * TRc[t];
    return a;
}

// ... the same walk that STOPS a row once the running product has reached the unit cut (`done()` is asked after every chunk
// of 8 dims and after the explicit dims).  Exact mode with all nodes in [0,1] (P.de_unit): the running product u_ij never grows,
// and at u_ij <= 2^-54 the factor is EXACTLY 1 in fp64 -- u-1 rounds to -1, u+1 to 1, (-1/1)^2 = 1, a*1 = a -- so the rest of
// the row changes no bit of the product (the factors between the cut and the end of its chunk are multiplied: they are 1 too).
// At the drivers' Gauss-Legendre nodes a row ends after ~16 dims: 12 % of the 32 640 pairs of D_256 are left
// (oracle: ttxo_set_unit_skip, tests/test_oracle_golden.py::test_unit_skip_changes_no_bit).
template <class STEP, class DONE>
__device__ __forceinline__ bool seg_range_cut(const short *p, int lo, int hi, const double *tab, STEP step, DONE done)
{
    for (int c = lo & ~7; c < hi; c += 8) {
        Short8 ix = ld8(p + c); double x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = tab[ix.v[k]];
        if (c >= lo && c + 8 <= hi) {
#pragma unroll
            for (int k = 0; k < 8; k++) step(x[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) if (c + k >= lo && c + k < hi) step(x[k]);
        }
        if (done()) return true;
    }
    return false;
}
template <int NS, class STEP, class DONE>
__device__ __forceinline__ void dims_asc_cut(const short *pa, int A, int s1, int s2, const short *pb, int m, const double *tab,
                                             int jlo, int jhi, STEP step, DONE done)
{
    if (jlo > jhi) return;
    const int a1 = jhi < A ? jhi : A;
    if (jlo <= a1 && seg_range_cut(pa, jlo - 1, a1, tab, step, done)) return;
    if (jlo <= A + 1 && A + 1 <= jhi) step(tab[s1]);
    if (NS == 2 && jlo <= A + 2 && A + 2 <= jhi) step(tab[s2]);
    if (done()) return;
    const int b0 = jlo > A + NS + 1 ? jlo : A + NS + 1;
    if (b0 <= jhi) seg_range_cut(pb, b0 - A - NS - 1, jhi - A - NS, tab, step, done);
}
// ... and from the COMPACT tables of k_de_ctables (P.de_cut: nodes in [0,1]): a row of the pair triangle is tabulated only up to the
// unit cut (CLc[i] factors for start i; CLc[m] / CRc[m] = totals), ULc[i] = running product of the start's row through the last left
// dim, or 0 when the row has ended inside the pivot's dims; the bond-spanning tail of a row ends at the cut as well.  Every factor
// left out is exactly 1: the product is bit-identical to de_pairs_tab's and to the reference's.
__device__ __forceinline__ double de_pairs_ctab(int m, const double *nodes, int A, const double *TLc, const int *CLc, const double *ULc,
                                                int s1, int s2, const short *rix, int ro, const double *TRc, const int *CRc)
{
    const int B = m - A - 2;
    const double x1 = nodes[s1], x2 = nodes[s2];
    double a = 1.0, u = 1.0;
    size_t pr = 0;
    auto step = [&](double xv) { u = u * xv; a = a * de_t2<true>(u); };
    auto done = [&]() { return u <= 0x1p-54; };
    auto run = [&]() {                                  // the bond-spanning tail of a row from s2 on (u holds the product so far)
        u = u * x2;
        if (u <= 0x1p-54) return;
        a = a * de_t2<true>(u);
        seg_range_cut(rix, ro, ro + B, nodes, step, done);
    };
    // one run of tabulated factors in order, 16 loads in flight ahead of the 16 dependent multiplies
    auto stream = [&](const double *T, int n) {
        int t = 0;
        for (; t + 16 <= n; t += 16) {
            double f[16];
#pragma unroll
            for (int k = 0; k < 16; k++) f[k] = T[t + k];
#pragma unroll
            for (int k = 0; k < 16; k++) a = a * f[k];
        }
        for (; t < n; t++) a = a * T[t];
    };
    const int i0 = (int)ULc[m], pre = CLc[m - 1];       // rows before i0 end inside the left pivot's dims
    stream(TLc, pre);
    pr = pre;
    for (int i = i0; i <= A; i++) {
        const int cnt = (i < A) ? CLc[i] : 0;
        stream(TLc + pr, cnt);
        pr += cnt;
        u = ULc[i] * x1;
        if (u > 0x1p-54) { a = a * de_t2<true>(u); run(); }
    }
    u = 1.0; run();                                     // i = A+1: starts after s1
    stream(TRc, CRc[m]);                                // i >= A+2: pairs inside the right pivot's dims
    return a;
}

template <int NS>
__device__ __forceinline__ double f_ising_de(int id, int m, int n1, const double *par, const short *pa, int A, int s1, int s2, const short *pb, bool unit = false)
{
    const double *nodes = par - 1;
    double a = 1.0;
    if (unit) {
        for (int i = 0; i <= m; i++) {
            double uij = 1.0;
            const int jlo = i + 1, jhi = m;
            if (jlo > jhi) continue;
            dims_asc_cut<NS>(pa, A, s1, s2, pb, m, nodes, jlo, jhi, [&](double xv) {
                uij = uij * xv;
                a = a * de_t2<true>(uij);
            }, [&]() { return uij <= 0x1p-54; });
        }
        return de_finish<NS>(id, a, m, n1, par, pa, A, s1, s2, pb);
    }
    for (int i = 0; i <= m; i++) {                                   // :186-195
        double uij = 1.0;
        const int jlo = i + 1, jhi = m;
        if (jlo > jhi) continue;
        dims_asc<NS>(pa, A, s1, s2, pb, m, nodes, jlo, jhi, [&](double xv) {
            uij = uij * xv;
            const double t = (uij - 1.0) / (uij + 1.0);
            a = a * (t * t);
        });
    }
    return de_finish<NS>(id, a, m, n1, par, pa, A, s1, s2, pb);
}
template <int NS>
__device__ __forceinline__ double de_finish(int id, double a, int m, int n1, const double *par, const short *pa, int A, int s1, int s2, const short *pb)
{
    const double *nodes = par - 1, *weights = par + n1 - 1;
    const int nb = m - A - NS;
    double b = 0.0;
    if (id == 2) {                                                   // :197-205
        double v = 1.0, w = 1.0, vk = 1.0, wk = 1.0;
        auto vstep = [&](double xv) { vk = vk * xv; v = v + vk; };
        auto wstep = [&](double xv) { wk = wk * xv; w = w + wk; };
        chain8<true>(pb, nb, nodes, vstep);
        if (NS == 2) vstep(nodes[s2]);
        vstep(nodes[s1]);
        chain8<true>(pa, A, nodes, vstep);
        chain8<false>(pa, A, nodes, wstep);
        wstep(nodes[s1]);
        if (NS == 2) wstep(nodes[s2]);
        chain8<false>(pb, nb, nodes, wstep);
        b = 1.0 / (v * w);
    }
    double f = (id == 2) ? 2 * a * b : 2 * a;
    auto fstep = [&](double xv) { f = f * xv; };
    chain8<false>(pa, A, weights, fstep);
    fstep(weights[s1]);
    if (NS == 2) fstep(weights[s2]);
    chain8<false>(pb, nb, weights, fstep);
    return f;
}

// Ising C (id 1) over aligned rows.  Same arithmetic as f_ising: the v- and w-recurrences of
// test_crs_ising.f90:199-204 are independent, so they run as separate loops.
__device__ __forceinline__ double f_ising_c3(int m, int n1, const double *par, const Src3 &S)
{
    const double *nodes = par - 1, *weights = par + n1 - 1;
    const int A = S.A, nb = m - A - 1;
    double v = 1.0, w = 1.0, vk = 1.0, wk = 1.0;
    auto vstep = [&](double xv) { vk = vk * xv; v = v + vk; };
    auto wstep = [&](double xv) { wk = wk * xv; w = w + wk; };
    chain8<true>(S.pb, nb, nodes, vstep);
    vstep(nodes[S.self]);
    chain8<true>(S.pa, A, nodes, vstep);
    chain8<false>(S.pa, A, nodes, wstep);
    wstep(nodes[S.self]);
    chain8<false>(S.pb, nb, nodes, wstep);
    double b = 1.0 / (v * w);
    double f = 2 * b;
    auto fstep = [&](double xv) { f = f * xv; };
    chain8<false>(S.pa, A, weights, fstep);
    fstep(weights[S.self]);
    chain8<false>(S.pb, nb, weights, fstep);
    return f;
}
__device__ __forceinline__ double f_ising_c4(int m, int n1, const double *par, const Src4 &S)
{
    const double *nodes = par - 1, *weights = par + n1 - 1;
    const int A = S.A, nb = m - A - 2;
    double v = 1.0, w = 1.0, vk = 1.0, wk = 1.0;
    auto vstep = [&](double xv) { vk = vk * xv; v = v + vk; };
    auto wstep = [&](double xv) { wk = wk * xv; w = w + wk; };
    chain8<true>(S.pb, nb, nodes, vstep);
    vstep(nodes[S.s2]); vstep(nodes[S.s1]);
    chain8<true>(S.pa, A, nodes, vstep);
    chain8<false>(S.pa, A, nodes, wstep);
    wstep(nodes[S.s1]); wstep(nodes[S.s2]);
    chain8<false>(S.pb, nb, nodes, wstep);
    double b = 1.0 / (v * w);
    double f = 2 * b;
    auto fstep = [&](double xv) { f = f * xv; };
    chain8<false>(S.pa, A, weights, fstep);
    fstep(weights[S.s1]); fstep(weights[S.s2]);
    chain8<false>(S.pb, nb, weights, fstep);
    return f;
}
template <int FUN>
__device__ __forceinline__ double eval_src4(const DevProb &P, const double *par, const Src4 &S, long slot = -1)
{
    if (FUN == FUN_ISING && P.ising_id == 1) return f_ising_c4(P.d, P.n[1], par, S);
    if (FUN == FUN_ISING && P.arith) return f_ising_fast(P.ising_id, P.d, P.n[1], par, S);
    if (FUN == FUN_ISING) return f_ising_de<2>(P.ising_id, P.d, P.n[1], par, S.pa, S.A, S.s1, S.s2, S.pb, P.de_unit != 0);
    return eval_fun<FUN>(P, par, S, slot);
}

// Same recurrences over rows of VALUES (node / weight doubles staged in LDS, 16-byte aligned, padded): a step
// is one LDS read (two doubles per ds_read_b128) + the dependent multiply(+add).  A lone wave issues one VALU
// instruction per 4 cycles, so the instruction count per step -- not the fp64 latency -- bounds these chains.
struct __align__(16) Double2 { double a, b; };
template <bool DESC, class STEP>
__device__ __forceinline__ void chain8v(const double *p, int n, STEP step)
{
    if (n <= 0) return;
    const int nfull = n >> 3, rem = n & 7;
    if (DESC && rem) {
        double x[8];
#pragma unroll
        for (int k = 0; k < 4; k++) { Double2 t = *reinterpret_cast<const Double2 *>(p + 8 * nfull + 2 * k); x[2 * k] = t.a; x[2 * k + 1] = t.b; }
#pragma unroll
        for (int k = 7; k >= 0; k--) if (k < rem) step(x[k]);
    }
#pragma unroll 4
    for (int ch = 0; ch < nfull; ch++) {
        const int c = 8 * (DESC ? nfull - 1 - ch : ch);
        double x[8];
#pragma unroll
        for (int k = 0; k < 4; k++) { Double2 t = *reinterpret_cast<const Double2 *>(p + c + 2 * k); x[2 * k] = t.a; x[2 * k + 1] = t.b; }
#pragma unroll
        for (int k = 0; k < 8; k++) step(x[DESC ? 7 - k : k]);
    }
    if (!DESC && rem) {
        double x[8];
#pragma unroll
        for (int k = 0; k < 4; k++) { Double2 t = *reinterpret_cast<const Double2 *>(p + 8 * nfull + 2 * k); x[2 * k] = t.a; x[2 * k + 1] = t.b; }
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < rem) step(x[k]);
    }
}
// chain8v with the LDS reads of the NEXT chunk issued before the dependent steps of the current one (two register sets, the
// loop unrolled by two so that they swap roles without copies).  Same values in the same order.
__device__ __forceinline__ void ld8d(const double *p, double (&x)[8])
{
#pragma unroll
    for (int k = 0; k < 4; k++) { const Double2 t = *reinterpret_cast<const Double2 *>(p + 2 * k); x[2 * k] = t.a; x[2 * k + 1] = t.b; }
}
template <bool DESC, class STEP>
__device__ __forceinline__ void chain8p(const double *p, int n, STEP step)
{
    if (n <= 0) return;
    const int nfull = n >> 3, rem = n & 7;
    double x[8], y[8];
    auto base = [&](int ch) { return 8 * (DESC ? nfull - 1 - ch : ch); };
    if (DESC && rem) {
        ld8d(p + 8 * nfull, x);
        if (nfull) ld8d(p + base(0), y);
#pragma unroll
        for (int k = 7; k >= 0; k--) if (k < rem) step(x[k]);
    } else if (nfull) ld8d(p + base(0), y);
    int ch = 0;
    for (; ch + 1 < nfull; ch += 2) {
        ld8d(p + base(ch + 1), x);
#pragma unroll
        for (int k = 0; k < 8; k++) step(y[DESC ? 7 - k : k]);
        if (ch + 2 < nfull) ld8d(p + base(ch + 2), y); else if (!DESC && rem) ld8d(p + 8 * nfull, y);
#pragma unroll
        for (int k = 0; k < 8; k++) step(x[DESC ? 7 - k : k]);
    }
    if (ch < nfull) {                      // one full chunk left (in y); the ascending remainder is requested under it
        if (!DESC && rem) ld8d(p + 8 * nfull, x);
#pragma unroll
        for (int k = 0; k < 8; k++) step(y[DESC ? 7 - k : k]);
        if (!DESC && rem) {
#pragma unroll
            for (int k = 0; k < 8; k++) if (k < rem) step(x[k]);
        }
    } else if (!DESC && rem) {             // the remainder sits in y (requested above) -- or nothing was requested yet (nfull == 0)
        if (nfull == 0) ld8d(p, y);
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < rem) step(y[k]);
    }
}
// dims 1..A: (an, aw); dim A+1: (sn, sw); dims A+2..m: (bn, bw)   [n = node values, w = weight values]
__device__ __forceinline__ double f_ising_c3v(int m, int A, const double *an, const double *aw, double sn, double sw,
                                              const double *bn, const double *bw)
{
    const int nb = m - A - 1;
    double v = 1.0, w = 1.0, vk = 1.0, wk = 1.0;
    auto vstep = [&](double xv) { vk = vk * xv; v = v + vk; };
    auto wstep = [&](double xv) { wk = wk * xv; w = w + wk; };
    chain8v<true>(bn, nb, vstep);
    vstep(sn);
    chain8v<true>(an, A, vstep);
    chain8v<false>(an, A, wstep);
    wstep(sn);
    chain8v<false>(bn, nb, wstep);
    double b = 1.0 / (v * w);
    double f = 2 * b;
    auto fstep = [&](double xv) { f = f * xv; };
    chain8v<false>(aw, A, fstep);
    fstep(sw);
    chain8v<false>(bw, nb, fstep);
    return f;
}

// aligned = rows are 16-byte aligned and padded (half-step / lottery staging); else the generic accessor
template <int FUN, bool ALIGNED>
__device__ __forceinline__ double eval_src3(const DevProb &P, const double *par, const Src3 &S, long slot = -1)
{
    if (ALIGNED && FUN == FUN_ISING && P.ising_id == 1) return f_ising_c3(P.d, P.n[1], par, S);
    if (ALIGNED && FUN == FUN_ISING && P.arith) return f_ising_fast(P.ising_id, P.d, P.n[1], par, S);
    if (ALIGNED && FUN == FUN_ISING) return f_ising_de<1>(P.ising_id, P.d, P.n[1], par, S.pa, S.A, S.self, 0, S.pb, P.de_unit != 0);
    return eval_fun<FUN>(P, par, S, slot);
}

// ------------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)
// ------------------------------------------------------------------------------------------------
// Wave-wide reductions on the DPP data path (cross-lane operands of the VALU itself) instead of ds_bpermute round trips
// through the LDS crossbar: butterfly inside each row of 16 lanes (quad_perm xor 1, xor 2, row_ror 4, row_ror 8), then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3; lane 63 holds the result, broadcast by readlane.
// Both operations are idempotent, so lanes that a step does not address combine with their own value.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, ROWMASK, 0xf, false); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_d(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = dpp_i<CTRL, ROWMASK>((int)b), hi = dpp_i<CTRL, ROWMASK>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane63(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_d<0xb1, 0xf>(v));
    v = fmax(v, dpp_d<0x4e, 0xf>(v));
    v = fmax(v, dpp_d<0x124, 0xf>(v));
    v = fmax(v, dpp_d<0x128, 0xf>(v));
    v = fmax(v, dpp_d<0x142, 0xa>(v));
    v = fmax(v, dpp_d<0x143, 0xc>(v));
    return readlane63(v);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void argmax_step(double &a, double &v, int &idx)
{
    // first-max rule of idamax: larger |.| wins, ties go to the lower index
    const double a2 = dpp_d<CTRL, ROWMASK>(a), v2 = dpp_d<CTRL, ROWMASK>(v);
    const int i2 = dpp_i<CTRL, ROWMASK>(idx);
    if (a2 > a || (a2 == a && i2 < idx)) { a = a2; v = v2; idx = i2; }
}
__device__ __forceinline__ void wave_argmax(double &a, double &v, int &idx)
{
    argmax_step<0xb1, 0xf>(a, v, idx);
    argmax_step<0x4e, 0xf>(a, v, idx);
    argmax_step<0x124, 0xf>(a, v, idx);
    argmax_step<0x128, 0xf>(a, v, idx);
    argmax_step<0x142, 0xa>(a, v, idx);
    argmax_step<0x143, 0xc>(a, v, idx);
    a = readlane63(a); v = readlane63(v); idx = __builtin_amdgcn_readlane(idx, 63);
}
// returns block max to every thread; sh must hold blockDim/64 doubles
__device__ __forceinline__ double block_max(double v, double *sh)
{
    v = wave_max(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = sh[0];
    for (int x = 1; x < nw; x++) r = fmax(r, sh[x]);
    return r;
}
// block arg-max; result valid in thread 0; sha/shv/shi hold blockDim/64 entries
__device__ __forceinline__ void block_argmax(double &a, double &v, int &idx, double *sha, double *shv, int *shi)
{
    wave_argmax(a, v, idx);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sha[w] = a; shv[w] = v; shi[w] = idx; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int x = 1; x < nw; x++)
            if (sha[x] > a || (sha[x] == a && shi[x] < idx)) { a = sha[x]; v = shv[x]; idx = shi[x]; }
}

__device__ __forceinline__ void atomic_max_pos(double *addr, double v)
{   // v >= 0: the IEEE bit patterns of non-negative doubles order like unsigned integers
    atomicMax((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}

// resolve pending partial arg-max records of the previous half-step into the step state
// (lib/dmrgg.f90:540-546 and :573-579)
__device__ inline void resolve_state(StepState &c, const Partial *pt)
{
    if (!c.active || !c.pending) return;
    const int nb = c.npart;                      // records the half-step kernel wrote (its number of fiber workgroups)
    double ba = -1.0, bv = 0.0; int bi = INT_MAX;
    for (int b = 0; b < nb; b++) {
        double a = pt[b].absmax; int ix = pt[b].idx;
        if (a > ba || (a == ba && ix < bi)) { ba = a; bv = pt[b].val; bi = ix; }
    }
    if (bi == INT_MAX) bi = 0;      // nothing compared (every residual a NaN): the first position, as idamax returns (no wild index)
    if (c.pending == 1) {
        int i = bi % c.r0 + 1, j = bi / c.r0 + 1;
        c.done = c.havecol && c.haverow && (i == c.ii && j == c.jj);
        c.ii = i; c.jj = j;
    } else {
        int k = bi % c.n2 + 1, q = bi / c.n2 + 1;
        c.done = c.havecol && c.haverow && (k == c.kk && q == c.qq);
        c.kk = k; c.qq = q;
    }
    c.pivot = bv;
    c.pending = 0;
}

// the same by the first WAVE of a workgroup (all 64 lanes call it; lane 0 writes c): the partial records are loaded by the lanes
// side by side and folded on the DPP path instead of one thread walking them (2 us of every half-step at 26 records)
__device__ inline void resolve_state_wave(StepState &c, const StepState &src, const Partial *pt, int lane)
{
    double ba = -1.0, bv = 0.0; int bi = INT_MAX;
    const bool pend = src.active && src.pending;
    if (pend)
        for (int b = lane; b < src.npart; b += 64) {
            const double a = pt[b].absmax; const int ix = pt[b].idx;
            if (a > ba || (a == ba && ix < bi)) { ba = a; bv = pt[b].val; bi = ix; }
        }
    wave_argmax(ba, bv, bi);
    if (lane != 0) return;
    c = src;
    if (!pend) return;
    if (bi == INT_MAX) bi = 0;
    if (c.pending == 1) {
        int i = bi % c.r0 + 1, j = bi / c.r0 + 1;
        c.done = c.havecol && c.haverow && (i == c.ii && j == c.jj);
        c.ii = i; c.jj = j;
    } else {
        int k = bi % c.n2 + 1, q = bi / c.n2 + 1;
        c.done = c.havecol && c.haverow && (k == c.kk && q == c.qq);
        c.kk = k; c.qq = q;
    }
    c.pivot = bv;
    c.pending = 0;
}

// ------------------------------------------------------------------------------------------------
// K0: initial cross (lib/dmrgg.f90:151-248)
// ------------------------------------------------------------------------------------------------
struct DiagIdx { const int *n; int k, s; __device__ __forceinline__ int operator()(int p) const { return (k - 1 + s * (p - 1)) % n[p] + 1; } };
struct FixIdx { const int *ind; int self, selfval; __device__ __forceinline__ int operator()(int s) const { return s == self ? selfval : ind[s]; } };

// every group evaluates all nn*snum shifted-diagonal samples (a few hundred) so that the MAXLOC of :196
// needs no exchange; each group is charged only its own share of evaluations (:160,181)
template <int FUN>
__global__ __launch_bounds__(256) void k_init_samples(DevProb P, int snum, int nn, int ldsrows, int shift_hi)
{
    extern __shared__ __align__(16) double dyn[];
    double *par = dyn;
    __shared__ double sha[4], shv[4]; __shared__ int shi[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    GroupState &gs = P.gs[g];
    for (int x = tid; x < P.npar; x += blockDim.x) par[x] = P.par[x];
    __syncthreads();
    double ba = -1.0, bv = 0.0; int bi = INT_MAX;
    // each thread writes the multi-index of its sample once into an aligned, padded LDS row (dims 1..m-1 | dim m) and
    // evaluates through the chunked evaluators instead of recomputing the shifted-diagonal index in every pass
    const int m = P.d, VSr = ((m + 7) & ~7) + 8;
    short *rows = (short *)(((size_t)(dyn + P.npar) + 15) & ~(size_t)15);
    short *row = rows + (size_t)tid * VSr;
    if (FUN == FUN_ISING && P.arith && P.ising_id != 1 && ldsrows) {
        // TTX_ARITH=fast, Ising D/E: one WAVE per sample (de_fast_point_wave: the lanes share the range ends).  A shifted diagonal with
        // shift 0 repeats one node in every dimension; for a node close to 1 no range reaches the cut and one thread would walk the
        // whole pair triangle (7 ms of the 220 ms of D_256 went into this kernel)
        const int lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6, n1m = P.n[1];
        double *xv = (double *)rows + (size_t)wv * 2 * (m + 64), *wvv = xv + (m + 64);
        int *nl = (int *)((double *)rows + (size_t)nw * 2 * (m + 64));      // mode sizes (one global round trip instead of one per sample)
        for (int x = tid; x < m; x += blockDim.x) nl[x] = P.n[x + 1];
        __syncthreads();
        for (int il = wv; il < nn * snum; il += nw) {
            const int k = il % nn + 1, sft = il / nn;
            __builtin_amdgcn_wave_barrier();
            for (int x = lane; x < m; x += 64) { const int ix = (k - 1 + sft * x) % nl[x]; xv[x] = par[ix]; wvv[x] = par[n1m + ix]; }
            __builtin_amdgcn_wave_barrier();
            const double f = de_fast_point_wave(P.ising_id, m, xv, wvv, lane);
            const double a = fabs(f);
            if (lane == 0 && (a > ba || (a == ba && il < bi))) { ba = a; bv = f; bi = il; }
        }
    } else
    if (!ldsrows)                                             // rows do not fit the LDS: generic accessor
        for (int il = tid; il < nn * snum; il += blockDim.x) {
            DiagIdx ix{P.n, il % nn + 1, il / nn};
            double f = eval_fun<FUN>(P, par, ix, (long)g * P.HS + il);
            double a = fabs(f);
            if (a > ba || (a == ba && il < bi)) { ba = a; bv = f; bi = il; }
        }
    else
    for (int il = tid; il < nn * snum; il += blockDim.x) {
        const int k = il % nn + 1, sft = il / nn;
        for (int p = 1; p <= m; p++) row[p - 1] = (short)((k - 1 + sft * (p - 1)) % P.n[p] + 1);
        for (int x = m; x < VSr; x++) row[x] = 1;
        row[m - 1 + 0] = row[m - 1];                          // (dim m stays readable as `self`)
        const int self = row[m - 1];
        row[m - 1] = 1;                                       // pad of the dims 1..m-1 part
        Src3 sx{row, m - 1, self, row + m};                   // empty right part: nb = 0, never read
        double f = eval_src3<FUN, true>(P, par, sx, (long)g * P.HS + il);
        double a = fabs(f);
        if (a > ba || (a == ba && il < bi)) { ba = a; bv = f; bi = il; }
    }
    if (HOST_PASS1(FUN, P)) return;
    block_argmax(ba, bv, bi, sha, shv, shi);
    if (tid == 0) {
        gs.amax = ba;                                             // :180,201
        // :153-156 shifts(p) = int(dble(snum)*dble(p)/nproc); own share of the samples (:160,181)
        int gg = gs.gglobal;
        int lo = (int)((double)snum * (double)gg / P.nprocs);
        int hi = (gg + 1 == P.nprocs) ? snum : (int)((double)snum * (double)(gg + 1) / P.nprocs);
        gs.neval = (long long)nn * (hi - lo);
        if (g == 0) {
            if (bi == INT_MAX) bi = 0;                            // every sample a NaN: the first one, as idamax
            int s = bi / nn, k = bi % nn + 1;
            for (int p = 1; p <= P.d; p++) P.ind0[p] = (k - 1 + s * (p - 1)) % P.n[p] + 1;   // :205-209
            P.ind0[P.d + 1] = 1;
        }
    }
}

// fibers through the initial index for own cores (:221-232)
template <int FUN>
__global__ __launch_bounds__(256) void k_init_fibers(DevProb P)
{
    extern __shared__ __align__(16) double dyn[];
    double *par = dyn;
    __shared__ double shm[4];
    const int g = blockIdx.y, tid = threadIdx.x;
    GroupState &gs = P.gs[g];
    const int p = gs.first + blockIdx.x;
    if (p > gs.last + 1) return;
    for (int x = tid; x < P.npar; x += blockDim.x) par[x] = P.par[x];
    __syncthreads();
    double *A = core_ptr(P, P.arg, g, p, gs.first);
    double mx = 0.0;
    for (int j = tid; j < P.n[p]; j += blockDim.x) {
        FixIdx ix{P.ind0, p, j + 1};
        double f = eval_fun<FUN>(P, par, ix, (long)g * P.HS + (long)blockIdx.x * P.NM + j);
        A[(size_t)P.RM * j] = f;
        mx = fmax(mx, fabs(f));
    }
    if (HOST_PASS1(FUN, P)) return;
    mx = block_max(mx, shm);
    if (tid == 0) { atomic_max_pos(&gs.amax, mx); atomicAdd((unsigned long long *)&gs.neval, (unsigned long long)P.n[p]); }
}

// inv, col, row, index tables of the rank-1 start (:213-217, :235-248)
__global__ __launch_bounds__(256) void k_init_factors(DevProb P)
{
    const int g = blockIdx.y, tid = threadIdx.x;
    GroupState &gs = P.gs[g];
    const int first = gs.first, p = first + blockIdx.x, m = P.d;
    if (p > gs.last + 1) return;
    const double *A = core_ptr(P, P.arg, g, p, first);
    if (p <= gs.last) {
        double piv = A[(size_t)P.RM * (P.ind0[p] - 1)];
        double rp = 1.0 / piv;
        double *C = core_ptr(P, P.col, g, p, first);
        for (int j = tid; j < P.n[p]; j += blockDim.x) C[(size_t)P.RM * j] = rp * A[(size_t)P.RM * j];  // d2_lual, r=1
        if (tid == 0) {
            inv_ptr(P, g, p, first)[0] = piv;
            int *v = vip_ptr(P, g, p, first);
            v[0] = 1; v[1] = P.ind0[p]; v[2] = P.ind0[p + 1]; v[3] = 1;
        }
    }
    if (p >= first + 1) {
        double *W = core_ptr(P, P.row, g, p, first);
        for (int k = tid; k < P.n[p]; k += blockDim.x) W[k] = A[(size_t)P.RM * k];                      // d2_luar, r=1: identity
    }
    // tables: L for bonds first-1..last (block b handles bond first-1+b), R for bonds first..last+1
    {
        int s = first - 1 + blockIdx.x;                    // L bond
        short *Lt = L_ptr(P, g, s, first);
        for (int x = tid; x < s; x += blockDim.x) Lt[(size_t)x * P.RM] = (short)P.ind0[x + 1];
        int s2 = first + blockIdx.x;                       // R bond
        short *Rt = R_ptr(P, g, s2, first);
        for (int x = tid; x < m - s2; x += blockDim.x) Rt[(size_t)x * P.RM] = (short)P.ind0[s2 + 1 + x];
        if (tid == 0 && blockIdx.x == 0) inv_ptr(P, g, first - 1, first)[0] = 1.0;   // :147
        if (P.arith && P.fpersist) {
            // TTX_ARITH=fast, Ising D/E: the table entries of the rank-1 start (pivot 0 of every bond) from scratch: wave 0 the left
            // multi-index ind0[1..s] of bond s, wave 1 the right multi-index ind0[s2+1..m] of bond s2
            extern __shared__ __align__(16) double dyn_i[];
            const int wv_ = tid >> 6, lane_ = tid & 63;
            if (wv_ < 2) {
                double *xs = dyn_i + (size_t)wv_ * 2 * (m + 8), *ws = xs + m + 8;
                const int len = wv_ == 0 ? s : m - s2, o0 = wv_ == 0 ? 1 : s2 + 1, bnd = wv_ == 0 ? s : s2;
                if (P.fun_id == FUN_MVN) {                     // mvn: dv = x - mu, Y = S dv, Q (entry k of the right side is dim s2 + k, 0-based)
                    for (int k = lane_; k < len; k += 64) xs[k] = P.par[P.ind0[o0 + k] - 1] - P.aux[o0 - 1 + k];
                    __builtin_amdgcn_wave_barrier();
                    mvn_entry_scratch(P, xs, len, o0 - 1, fast_dv(P, wv_, g, bnd, first), fast_near(P, wv_, g, bnd, first), fast_piv(P, wv_, g, bnd, first), lane_);
                } else {
                for (int k = lane_; k < len; k += 64) { const int ix = P.ind0[o0 + k]; xs[k] = P.par[ix - 1]; ws[k] = P.par[P.n[1] + ix - 1]; }
                __builtin_amdgcn_wave_barrier();
                fast_entry_scratch(xs, ws, len, wv_, fast_near(P, wv_, g, bnd, first), fast_piv(P, wv_, g, bnd, first), P.RM, lane_);
                }
            }
        }
    }
}

// pivotmax_prev (:234) and the group's factor of the initial quadrature value (:250-258); one block per group,
// one thread per core for the ddot, then the ordered product
__global__ __launch_bounds__(256) void k_init_final(DevProb P)
{
    __shared__ double dots[256];
    const int g = blockIdx.x, tid = threadIdx.x;
    GroupState &gs = P.gs[g];
    const bool lastgroup = (gs.last + 1 == P.d && gs.gglobal == P.nprocs - 1);
    const int ncore = gs.last - gs.first + 1 + (lastgroup ? 1 : 0);
    double val = 1.0;
    if (P.has_quad) {
        for (int c0 = 0; c0 < ncore; c0 += 256) {
            const int c = c0 + tid;
            if (c < ncore) {
                const int p = gs.first + c;
                const double *A = core_ptr(P, P.arg, g, p, gs.first), *w = P.quadw + (size_t)p * P.NM;
                double t = 0.0;
                for (int j = 0; j < P.n[p]; j++) t = t + A[(size_t)P.RM * j] * w[j];
                dots[tid] = t;
            }
            __syncthreads();
            if (tid == 0)
                for (int c2 = c0; c2 < min(ncore, c0 + 256); c2++) {
                    const int p = gs.first + c2;
                    if (p <= gs.last) val = val * dots[c2 - c0] / inv_ptr(P, g, p, gs.first)[0];
                    else val = val * dots[c2 - c0];
                }
            __syncthreads();
        }
    }
    if (tid == 0) { gs.pivotmax_prev = gs.amax; gs.pivotmax = -1.0; gs.pivotmin = -1.0; gs.initval = val; }
}

// start of a run (lib/dmrgg.f90:96-100, 141-148, 279-288): every per-run quantity back to its initial value in ONE launch
// (the bond ranges first/last/gglobal of the groups never change and are set when the engine is created)
__global__ __launch_bounds__(256) void k_reset(DevProb P, size_t SB, size_t QB)
{
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    const size_t nr = (size_t)P.G * (P.d + 2);
    for (size_t x = t0; x < nr; x += nt) { P.r[x] = 1; P.rr[x] = 1; P.upd[x] = 0; }
    for (size_t x = t0; x < 4 * nr; x += nt) P.tape[x] = -1;
    for (size_t x = t0; x < SB; x += nt) P.sumsend[x] = 0.0;
    for (size_t x = t0; x < QB; x += nt) P.qsend[x] = 0.0;
    if (t0 < 4) P.ctl[t0] = 0;
    if (P.lot_ctr) for (size_t x = t0; x < (size_t)P.G; x += nt) P.lot_ctr[x] = 0u;
    if (P.cl_ctr) {
        for (size_t x = t0; x < (size_t)P.G; x += nt) P.cl_ctr[x] = 0u;
        ClPart z; z.ab = 0.0; z.bb = 0.0; z.mx = 0.0; z.ix = 0; z.pad = 0;
        for (size_t x = t0; x < (size_t)2 * P.G * TTX_CLREC; x += nt) P.cl_part[x] = z;
    }
    for (size_t g = t0; g < (size_t)P.G; g += nt) {
        GroupState &gs = P.gs[g];
        gs.amax = 0.0; gs.pivotmax = -1.0; gs.pivotmin = -1.0; gs.pivotmax_prev = 0.0;
        gs.neval = 0; gs.rngpos = 0; gs.val = 0.0; gs.initval = 0.0; gs.bytes_half = 0.0; gs.n_resid = 0;
#ifdef TTX_STAMPS
        for (int a = 0; a < 2; a++) { gs.nstamp[a] = 0; for (int b = 0; b < 16; b++) gs.stamp[a][b] = 0; }
#endif
    }
}

// snapshot of the bond this group works on at step pp of the sweep (:325-335); thread 0 only
__device__ inline void bond_state(const DevProb &P, int g, int dir, int pp, StepState &st)
{
    GroupState &gs = P.gs[g];
    const int m = P.d;
    int *r = P.r + (size_t)g * (m + 2);
    if (pp == 1) {                                        // sweep start, :325-327
        gs.pivotmax = -1.0; gs.pivotmin = -1.0;
        int *rr = P.rr + (size_t)g * (m + 2);
        for (int s = 0; s <= m; s++) rr[s] = r[s];
    }
    int nb = gs.last - gs.first + 1;
    st.active = (pp <= nb);
    st.done = 0; st.havecol = 0; st.haverow = 0; st.crs = 0; st.pending = 0; st.npart = 0; st.pivot = 0.0;
    st.ii = st.jj = st.kk = st.qq = 0;
    if (st.active) {
        int p = (dir == 1) ? gs.first + pp - 1 : gs.last + 1 - pp;   // :330-331
        st.p = p; st.r0 = r[p - 1]; st.r1 = r[p]; st.r2 = r[p + 1]; st.n1 = P.n[p]; st.n2 = P.n[p + 1];
    } else { st.p = 0; st.r0 = st.r1 = st.r2 = st.n1 = st.n2 = 0; st.done = 1; }
}

// full pivoting (piv = -1, :341-408): the bond state, then one column half-step per (k,q) (k_halfstep mode 3),
// then the global first arg-max over the partial records
__global__ void k_bond_begin(DevProb P, int dir, int pp)
{
    if (threadIdx.x != 0) return;
    StepState st;
    bond_state(P, blockIdx.x, dir, pp, st);
    P.gs[blockIdx.x].S[0] = st;
}
__global__ __launch_bounds__(256) void k_full_resolve(DevProb P)
{
    __shared__ double sha[4], shv[4]; __shared__ int shi[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    GroupState &gs = P.gs[g];
    StepState st = gs.S[0];
    if (!st.active) return;
    const int nf = st.r0 * st.n1, nb = (nf + TTX_BLK - 1) / TTX_BLK, ncol = st.n2 * st.r2;
    double ba = -1.0, bv = 0.0; int bi = INT_MAX;
    for (int x = tid; x < ncol * nb; x += blockDim.x) {
        const Partial pr = P.pfull[((size_t)g * P.NM * P.RM + x / nb) * P.nfb + x % nb];
        if (pr.absmax > ba || (pr.absmax == ba && pr.idx < bi)) { ba = pr.absmax; bv = pr.val; bi = pr.idx; }
    }
    block_argmax(ba, bv, bi, sha, shv, shi);
    if (tid == 0) {                                       // :388-396
        int x = (bi == INT_MAX) ? 0 : bi;                  // every residual a NaN: the first position, as idamax
        st.qq = x / (nf * st.n2) + 1; x %= nf * st.n2;
        st.kk = x / nf + 1; x %= nf;
        st.jj = x / st.r0 + 1; st.ii = x % st.r0 + 1;
        st.pivot = bv;
        gs.S[0] = st;
    }
}

// piv = -1 as one dense step (lib/dmrgg.f90:384-396): B = A - col(p) * row(p+1) over the whole superblock by fp64 MFMA
// (v_mfma_f64_16x16x4_f64; operand lane maps as k_gemm_mfma in ttx_ttops.h) fused with the arg-max of |B|.
// A = sb [(i,j) fastest, nf = r0*n1 rows][(k,q), nc = n2*r2 columns]; col(p)[(i + RM j) + SS s]; row(p+1)[(k + NM q) + SW s].
// grid = (ceil(nf/64), ceil(nc/64), groups), 256 threads: wave w owns rows 16w..16w+15 of the 64x64 tile, the row-factor
// tile (K x 64) is staged in LDS once per block.  The MFMA accumulates in its own order, so the residuals differ from the
// reference's dgemm in the last bits: this path is checked by tolerance, the column-by-column path (mode 3) stays the
// bit-exact checker.  One Partial per tile, first-max rule on the superblock's linear index t + nf * column.
typedef double dbl4_ __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_full_gemm_argmax(DevProb P)
{
    __shared__ double Bt[64 * 65];                     // row factor tile: Bt[k * 65 + c], K <= 64
    __shared__ double sha[4], shv[4]; __shared__ int shi[4];
    const int g = blockIdx.z, tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const GroupState &gs = P.gs[g];
    const StepState &st = gs.S[0];
    if (!st.active) return;
    const int p = st.p, r0 = st.r0, r1 = st.r1, r2 = st.r2, n1 = st.n1, n2 = st.n2, first = gs.first;
    const int nf = r0 * n1, nc = n2 * r2;
    const int tm = blockIdx.x * 64, tn = blockIdx.y * 64;
    if (tm >= nf || tn >= nc) return;
    const double *Cp = core_ptr(P, P.col, g, p, first), *Wq = core_ptr(P, P.row, g, p + 1, first);
    const double *sb = P.sb + (size_t)g * ((size_t)P.RM * P.NM) * ((size_t)P.RM * P.NM);
    for (int x = tid; x < r1 * 64; x += 256) {
        const int k = x >> 6, c = x & 63, col = tn + c;
        double v = 0.0;
        if (col < nc) { const int kk = col % n2, qq = col / n2; v = Wq[kk + (size_t)P.NM * qq + P.SW * k]; }
        Bt[k * 65 + c] = v;
    }
    __syncthreads();
    const int ar = tm + 16 * wave + (l & 15), kq = l >> 4;
    const bool arok = ar < nf;
    const size_t aoff = arok ? (size_t)(ar % r0) + (size_t)P.RM * (ar / r0) : 0;
    dbl4_ acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc[nt] = dbl4_{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < r1; k0 += 4) {
        const int k = k0 + kq;
        const double a = (arok && k < r1) ? Cp[aoff + P.SS * k] : 0.0;
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const double b = (k < r1) ? Bt[k * 65 + 16 * nt + (l & 15)] : 0.0;
            acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[nt], 0, 0, 0);
        }
    }
    double ab = -1.0, bv = 0.0; int bi = INT_MAX;
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int row = tm + 16 * wave + (l >> 4) + 4 * reg, col = tn + 16 * nt + (l & 15);
            if (row < nf && col < nc) {
                const size_t ix = (size_t)row + (size_t)nf * col;
                const double b = sb[ix] - acc[nt][reg];
                const double a_ = fabs(b);
                if (a_ > ab || (a_ == ab && (int)ix < bi)) { ab = a_; bv = b; bi = (int)ix; }
            }
        }
    block_argmax(ab, bv, bi, sha, shv, shi);
    if (tid == 0) { Partial pr; pr.absmax = ab; pr.val = bv; pr.idx = bi; pr.pad = 0; P.pfull2[(size_t)g * P.fp_tiles + blockIdx.y * gridDim.x + blockIdx.x] = pr; }
}
__global__ __launch_bounds__(256) void k_full_resolve2(DevProb P, int gx, int gy)
{
    __shared__ double sha[4], shv[4]; __shared__ int shi[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    GroupState &gs = P.gs[g];
    StepState st = gs.S[0];
    if (!st.active) return;
    const int nf = st.r0 * st.n1, nc = st.n2 * st.r2;
    const int ux = (nf + 63) / 64, uy = (nc + 63) / 64;          // tiles that hold data
    double ba = -1.0, bv = 0.0; int bi = INT_MAX;
    for (int x = tid; x < ux * uy; x += blockDim.x) {
        const Partial pr = P.pfull2[(size_t)g * P.fp_tiles + (x / ux) * gx + (x % ux)];
        if (pr.absmax > ba || (pr.absmax == ba && pr.idx < bi)) { ba = pr.absmax; bv = pr.val; bi = pr.idx; }
    }
    block_argmax(ba, bv, bi, sha, shv, shi);
    if (tid == 0) {                                       // :388-396
        int x = (bi == INT_MAX) ? 0 : bi;                  // every residual a NaN: the first position, as idamax
        st.qq = x / (nf * st.n2) + 1; x %= nf * st.n2;
        st.kk = x / nf + 1; x %= nf;
        st.jj = x / st.r0 + 1; st.ii = x % st.r0 + 1;
        st.pivot = bv;
        gs.S[0] = st;
    }
}

// Ising D / E: per-bond tables of the pair factors that do not span the bond (see de_pairs_tab).  For every left
// pivot c of bond p-1 (dims 1..A, A = p-1) and every start i: TL[c*NP + off(i) + j-i-1] = ((u-1)/(u+1))^2 with
// u = x_{i+1}*...*x_j accumulated left to right (test_crs_ising.f90:188-192), UL[c*(m+1) + i] = u after j = A; the
// same for every right pivot of bond p+1 over its own dims (TR).  One thread per (start, pivot).  A pivot's factors are
// CONTIGUOUS in the order in which an element through that pivot multiplies them, so that a wave whose elements share
// the pivot (k_halfstep_de) streams them with coalesced 512-byte loads and a lone lane (lottery) walks its own row.
__global__ __launch_bounds__(256) void k_de_tables(DevProb P, int dir, int pp)
{
    const int g = blockIdx.y, m = P.d, RM = P.RM;
    const GroupState &gs = P.gs[g];
    const int first = gs.first, last = gs.last;
    if (pp > last - first + 1) return;
    const int p = (dir == 1) ? first + pp - 1 : last + 1 - pp;
    const int *r = P.r + (size_t)g * (m + 2);
    const int r0 = r[p - 1], r2 = r[p + 1], A = p - 1, B = m - p - 1;
    const double *nodes = P.par - 1;
    const size_t NP = (size_t)P.de_npair, tsz = NP * RM;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = x % RM, i = (x / RM) % (m + 1), side = x / (RM * (m + 1));
    if (side == 0) {
        if (c >= r0 || i > A) return;
        const short *Lt = L_ptr(P, g, p - 1, first);
        double *TL = P.deTL + (size_t)g * tsz + (size_t)c * NP, *UL = P.deUL + ((size_t)g * RM + c) * (m + 1);
        const size_t off = (size_t)i * A - (size_t)i * (i - 1) / 2;
        double u = 1.0;
        for (int j = i + 1; j <= A; j++) {
            u = u * nodes[Lt[(size_t)(j - 1) * RM + c]];
            const double t = (u - 1.0) / (u + 1.0);
            TL[off + (j - i - 1)] = t * t;
        }
        UL[i] = u;
    } else if (side == 1) {
        if (c >= r2 || i >= B) return;
        const short *Rt = R_ptr(P, g, p + 1, first);
        double *TR = P.deTR + (size_t)g * tsz + (size_t)c * NP;
        const size_t off = (size_t)i * B - (size_t)i * (i - 1) / 2;
        double u = 1.0;
        for (int j = i + 1; j <= B; j++) {
            u = u * nodes[Rt[(size_t)(j - 1) * RM + c]];
            const double t = (u - 1.0) / (u + 1.0);
            TR[off + (j - i - 1)] = t * t;
        }
    }
}

// Ising D / E with all nodes in [0,1] (P.de_cut): the COMPACT tables.  One workgroup per pivot of the two sets of the bond step
// (left pivots of bond p-1, right pivots of bond p+1).  Row i of a pivot (start at its dim i+1): the factors ((u-1)/(u+1))^2 of
// u = x_{i+1} ... x_j, j = i+1, i+2, ..., while u > 2^-54 -- at u <= 2^-54 the factor is exactly 1 and, u being non-increasing, so is
// every later one of the row.  A pivot's rows are CONTIGUOUS in the order in which an element through that pivot multiplies
// them; CL / CR[i] = entries of row i, [m] = their total; UL[i] = the running product of row i through the pivot's last dim when
// the row gets that far, else 0 (UL[A] = 1: the row that starts behind the pivot); UL[m] = first such row i0 (all later ones get
// that far too), CL[m-1] = entries of the rows before it.  ~12 % of the full triangle at D_256.
// grid = (2 RM, groups), 256 threads, dynamic LDS d doubles.
__global__ __launch_bounds__(256) void k_de_ctables(DevProb P, int dir, int pp)
{
    extern __shared__ __align__(16) double xs[];
    __shared__ int s_w[4], s_carry;
    const int g = blockIdx.y, m = P.d, RM = P.RM, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const GroupState &gs = P.gs[g];
    const int first = gs.first, last = gs.last;
    if (pp > last - first + 1) return;
    const int p = (dir == 1) ? first + pp - 1 : last + 1 - pp;
    const int *r = P.r + (size_t)g * (m + 2);
    const int side = (int)blockIdx.x / RM, c = (int)blockIdx.x % RM;
    if (c >= (side == 0 ? r[p - 1] : r[p + 1])) return;
    const int len = side == 0 ? p - 1 : m - p - 1;
    const short *tab = side == 0 ? L_ptr(P, g, p - 1, first) : R_ptr(P, g, p + 1, first);
    const size_t NP = (size_t)P.de_npair, tsz = NP * RM;
    double *T = (side == 0 ? P.deTL : P.deTR) + (size_t)g * tsz + (size_t)c * NP;
    int *C = (side == 0 ? P.deCL : P.deCR) + ((size_t)g * RM + c) * (m + 1);
    double *UL = P.deUL + ((size_t)g * RM + c) * (m + 1);
    __shared__ int s_i0, s_pre;
    for (int k = tid; k < len; k += 256) xs[k] = P.par[tab[(size_t)k * RM + c] - 1];
    if (tid == 0) { s_carry = 0; s_i0 = len; s_pre = -1; }
    __syncthreads();
    for (int base = 0; base < len; base += 256) {
        const int i = base + tid;
        int cnt = 0; double u = 1.0;
        if (i < len) for (int j = i; j < len; j++) { u = u * xs[j]; if (u <= 0x1p-54) break; cnt++; }
        // exclusive scan of cnt over the workgroup
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
        if (lane == 63) s_w[wv] = inc;
        if (i < len && cnt == len - i) atomicMin(&s_i0, i);          // the row gets through the pivot's last dim: it and all later ones span the bond
        __syncthreads();
        int off = s_carry + inc - cnt;
        for (int w = 0; w < wv; w++) off += s_w[w];
        if (i == s_i0 && i < len) s_pre = off;
        if (i < len) {
            C[i] = cnt;
            if (side == 0) UL[i] = (cnt == len - i) ? u : 0.0;
            double uu = 1.0;
            for (int t = 0; t < cnt; t++) { uu = uu * xs[i + t]; T[off + t] = de_t2<true>(uu); }
        }
        __syncthreads();
        if (tid == 0) s_carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    // rows 0 .. i0-1 end inside the pivot's dims: their C[m-1] factors form one run without a bond-spanning tail in between
    if (tid == 0) { C[m] = s_carry; C[m - 1] = (s_pre >= 0) ? s_pre : s_carry; if (side == 0) { UL[len] = 1.0; UL[m] = (double)s_i0; } }
}

// ------------------------------------------------------------------------------------------------
// K_lottery: lottery2 candidates, their values and residuals, start pivot (lib/dmrgg.f90:410-484)
// one block per group
// ------------------------------------------------------------------------------------------------
template <int FUN>
__global__ __launch_bounds__(512) void k_lottery(DevProb P, int dir, int pp, int vals, int phase)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ StepState st;
    __shared__ int zc[128], zr[128], zcs[128], zrs[128], keepc[128], keepr[128];
    __shared__ int nzc, nzr, nsc, nsr;
    __shared__ ttx_cdfseg segc[TTX_MAXSEG], segr[TTX_MAXSEG];
    __shared__ double sha[8], shv[8]; __shared__ int shi[8];
    __shared__ int s_last;
    const int g = blockIdx.y, tid = threadIdx.x, m = P.d;
    const int nbl = (int)gridDim.x, blk = (int)blockIdx.x;
    GroupState &gs = P.gs[g];
    STAMP_DECL;
    // phase 0: everything in this launch.  Ising D/E (one candidate per wave, ttx_de.h): phase 1 draws the candidates into
    // P.lotc and leaves the state in gs.S[0]; k_lottery_eval_de fills P.lotf; phase 2 takes residuals, arg-max and state.
    if (tid == 0) { if (phase == 2) st = gs.S[0]; else bond_state(P, g, dir, pp, st); }
    __syncthreads();
    if (!st.active) { if (tid == 0 && phase != 2) gs.S[0] = st; return; }
    STAMP(gs, 0);   // 0: state
    const int p = st.p, r0 = st.r0, r1 = st.r1, r2 = st.r2, n1 = st.n1, n2 = st.n2, first = gs.first;
    const int nlot = r0 + n1 + n2 + r2;
    // LDS: par | lot[4*nlot] (int) | LT[r0][VS] | RT[r2][VS] (short): the pivot tables of bonds p-1 and p+1,
    // transposed so that the multi-index of a left / right pivot is one contiguous 16-byte-aligned row
    const int VS = ((m + 7) & ~7) + 8;
    double *par = dyn;
    int *lot = (int *)(dyn + ((P.npar + 1) & ~1));
    short *LT = (short *)(lot + ((4 * nlot + 3) & ~3));
    short *RT = LT + (size_t)r0 * VS;
    const double *Cp = core_ptr(P, P.col, g, p, first), *Wq = core_ptr(P, P.row, g, p + 1, first);
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = tid; x < P.npar; x += blockDim.x) par[x] = P.par[x];
    // mvn: rows of differences x - mu (doubles) take the place of the index rows (host sized the LDS for them: vals)
    const bool fastp = fast_path(P, FUN);           // TTX_ARITH=fast: candidates from the tables of k_fast_tables (vals = LDS rows per side)
    const bool usem = (FUN == FUN_MVN) && (vals != 0) && !fastp;
    double *DLv = (double *)(((size_t)LT + 15) & ~(size_t)15), *DRv = DLv + (size_t)r0 * VS;
    double *sNL = DLv, *sNR = DLv + (size_t)vals * P.RM;
    __shared__ int s_mc[2];
    int fcap = 0;
    if (phase != 0) {
        // no evaluation in this launch: the pivot rows are not needed
    } else if (fastp) {
        if (FUN == FUN_ISING && vals > 0) {
            // the leading rows of the two decay tables (those above the cut for at least one pivot, at most vals) into LDS
            if (tid < 2) s_mc[tid] = 0;
            __syncthreads();
            for (int x = tid; x < r0; x += blockDim.x) atomicMax(&s_mc[0], (int)fast_piv(P, 0, g, p - 1, first)[FP_N * P.RM + x]);
            for (int x = tid; x < r2; x += blockDim.x) atomicMax(&s_mc[1], (int)fast_piv(P, 1, g, p + 1, first)[FP_N * P.RM + x]);
            __syncthreads();
            fcap = min(vals, max(s_mc[0], s_mc[1]));
            const double *nL = fast_near(P, 0, g, p - 1, first), *nR = fast_near(P, 1, g, p + 1, first);
            for (int x = tid; x < fcap * P.RM; x += blockDim.x) { const int c = x % P.RM; sNL[x] = (c < r0) ? nL[x] : 0.0; sNR[x] = (c < r2) ? nR[x] : 0.0; }
        }
    } else if (usem) {
        __syncthreads();
        const double *mu = P.aux;
        for (int x = tid; x < r0 * VS; x += blockDim.x) { const int c = x / VS, o = x - c * VS; DLv[x] = (o < p - 1) ? par[Lt[(size_t)o * P.RM + c] - 1] - mu[o] : 0.0; }
        for (int x = tid; x < r2 * VS; x += blockDim.x) { const int c = x / VS, o = x - c * VS; DRv[x] = (o < m - p - 1) ? par[Rt[(size_t)o * P.RM + c] - 1] - mu[p + 1 + o] : 0.0; }
    } else {
    for (int x = tid; x < r0 * VS; x += blockDim.x) { const int c = x / VS, o = x - c * VS; LT[x] = (o < p - 1) ? Lt[(size_t)o * P.RM + c] : (short)1; }
    for (int x = tid; x < r2 * VS; x += blockDim.x) { const int c = x / VS, o = x - c * VS; RT[x] = (o < m - p - 1) ? Rt[(size_t)o * P.RM + c] : (short)1; }
    }
    // rnd.f90:120: d(nlot,2) column-major from the (never seeded) run-time generator.  Draw #k comes from the
    // generator word 48271^(2k+1): split as [48271^(2 pos+1)] * [48271^2]^il so that the long jump-ahead is done
    // once per block (threads 32/33) while every thread raises the short power
    __shared__ unsigned long long sA[2];
    int Kc = 0, Kr = 0;
    const int CH = (nbl == 1) ? nlot : 64;                     // candidates per block
    const int il_first = blk * CH + tid, il_end = min(nlot, (blk + 1) * CH);
    unsigned long long bil = 0;
    if (phase != 2) {
    if (tid == 32) sA[0] = ttx_minstd_pow(2 * gs.rngpos + 1);
    if (tid == 33) sA[1] = ttx_minstd_pow(2 * (gs.rngpos + nlot) + 1);
    bil = ttx_minstd_pow(2ull * (unsigned long long)il_first);
    // zero-weight positions (existing pivots), :432-439
    const int *vp = vip_ptr(P, g, p, first);
    if (tid < r1) {
        zc[tid] = (vp[4 * tid + 0] - 1) + r0 * (vp[4 * tid + 1] - 1) + 1;
        zr[tid] = (vp[4 * tid + 2] - 1) + n2 * (vp[4 * tid + 3] - 1) + 1;
    }
    __syncthreads();
    if (tid < r1) {          // rank sort (total order with index tie-break)
        int a = zc[tid], b = zr[tid], ra = 0, rb = 0;
        for (int u = 0; u < r1; u++) {
            ra += (zc[u] < a) || (zc[u] == a && u < tid);
            rb += (zr[u] < b) || (zr[u] == b && u < tid);
        }
        zcs[ra] = a; zrs[rb] = b;
    }
    __syncthreads();
    if (tid < r1) { keepc[tid] = (tid == 0) || (zcs[tid] != zcs[tid - 1]); keepr[tid] = (tid == 0) || (zrs[tid] != zrs[tid - 1]); }
    __syncthreads();
    if (tid < r1) {          // compaction of distinct values into zc / zr
        int pc = 0, pr = 0;
        for (int u = 0; u < tid; u++) { pc += keepc[u]; pr += keepr[u]; }
        if (keepc[tid]) zc[pc] = zcs[tid];
        if (keepr[tid]) zr[pr] = zrs[tid];
        if (tid == r1 - 1) { nzc = pc + keepc[tid]; nzr = pr + keepr[tid]; }
    }
    __syncthreads();
    STAMP(gs, 0);   // 1: tables + powers + zero lists
    Kc = r0 * n1 - nzc; Kr = n2 * r2 - nzr;
    if (P.cdf_tab && Kc <= P.cdf_kmax && Kr <= P.cdf_kmax) {
        // segment tables depend on K only: precomputed at ttx_create, copied here
        if (tid < 64) { if (tid < P.cdf_ns[Kc]) segc[tid] = P.cdf_tab[(size_t)Kc * TTX_TABSEG + tid]; if (tid == 0) nsc = P.cdf_ns[Kc]; }
        else if (tid < 128) { const int t2 = tid - 64; if (t2 < P.cdf_ns[Kr]) segr[t2] = P.cdf_tab[(size_t)Kr * TTX_TABSEG + t2]; if (t2 == 0) nsr = P.cdf_ns[Kr]; }
    } else {
        if (tid == 0) nsc = ttx_cdf_build(Kc, segc);
        if (tid == 64) nsr = ttx_cdf_build(Kr, segr);
    }
    }
    __syncthreads();
    STAMP(gs, 0);   // 2: cdf
    double ma = 0.0;
    double ba = -1.0, bv = 0.0; int bi = INT_MAX;
    for (int il = il_first; il < il_end; il += blockDim.x) {
        int i, j, k, q;
        if (phase != 2) {
            double d1, d2;
            if (il == il_first) { d1 = ttx_flang_from_word(ttx_mulmod31(sA[0], bil)); d2 = ttx_flang_from_word(ttx_mulmod31(sA[1], bil)); }
            else { d1 = ttx_flang_draw(gs.rngpos + il); d2 = ttx_flang_draw(gs.rngpos + nlot + il); }
            const int x = ttx_lottery_index(segc, nsc, Kc, r0 * n1, zc, nzc, d1);      // rnd.f90:122-123
            const int y = ttx_lottery_index(segr, nsr, Kr, n2 * r2, zr, nzr, d2);
            i = (x - 1) % r0 + 1; j = (x - 1) / r0 + 1; k = (y - 1) % n2 + 1; q = (y - 1) / n2 + 1;   // :447-452
            if (phase == 1) { int *c_ = P.lotc + ((size_t)g * P.lot_max + il) * 4; c_[0] = i; c_[1] = j; c_[2] = k; c_[3] = q; continue; }
        } else {
            const int *c_ = P.lotc + ((size_t)g * P.lot_max + il) * 4;
            i = c_[0]; j = c_[1]; k = c_[2]; q = c_[3];
        }
        lot[4 * il] = i; lot[4 * il + 1] = j; lot[4 * il + 2] = k; lot[4 * il + 3] = q;
        double f;
        if (phase == 2) f = P.lotf[(size_t)g * P.lot_max + il];
        else if (fastp) f = (FUN == FUN_MVN) ? mvn_fast_value(P, g, p, first, i - 1, j - 1, k - 1, q - 1, mvn_fast_cross(P, g, p, first, i - 1, q - 1))
                                             : de_fast_elem4(P, g, p, first, i - 1, j - 1, k - 1, q - 1, sNL, sNR, fcap);
        else if (usem) f = f_mvn_rows<2>(m, P.auxT, P.mvn_norm, DLv + (size_t)(i - 1) * VS, p - 1, par[j - 1] - P.aux[p - 1], par[k - 1] - P.aux[p],
                                    DRv + (size_t)(q - 1) * VS);
        else if (FUN == FUN_ISING && P.ising_id != 1 && P.deTL) {
            const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
            const double *TL = P.deTL + (size_t)g * tsz + (size_t)(i - 1) * NP, *UL = P.deUL + ((size_t)g * P.RM + (i - 1)) * (m + 1);
            const double *TR = P.deTR + (size_t)g * tsz + (size_t)(q - 1) * NP;
            const short *pa_ = LT + (size_t)(i - 1) * VS, *pb_ = RT + (size_t)(q - 1) * VS;
            const double ap = P.de_cut ? de_pairs_ctab(m, par - 1, p - 1, TL, P.deCL + ((size_t)g * P.RM + (i - 1)) * (m + 1), UL, j, k, pb_, 0, TR,
                                                       P.deCR + ((size_t)g * P.RM + (q - 1)) * (m + 1))
                            : P.de_unit ? de_pairs_tab<true>(m, par - 1, p - 1, TL, UL, j, k, pb_, 0, TR)
                                        : de_pairs_tab<false>(m, par - 1, p - 1, TL, UL, j, k, pb_, 0, TR);
            f = de_finish<2>(P.ising_id, ap, m, P.n[1], par, pa_, p - 1, j, k, pb_);
        } else {
            Src4 sx{LT + (size_t)(i - 1) * VS, p - 1, j, k, RT + (size_t)(q - 1) * VS};
            f = eval_src4<FUN>(P, par, sx, (long)g * P.HS + il);                   // :455-463
        }
        ma = fmax(ma, fabs(f));
        const double *c = Cp + (i - 1) + (size_t)P.RM * (j - 1), *w = Wq + (k - 1) + (size_t)P.NM * (q - 1);
        double t = 0.0;                                                            // ddot, :474
#pragma unroll 8
        for (int s = 0; s < r1; s++) t = t + c[P.SS * s] * w[P.SW * s];
        const double b = f - t;
        const double a = fabs(b);
        if (a > ba || (a == ba && il < bi)) { ba = a; bv = b; bi = il; }
    }
    STAMP(gs, 0);   // 3: selection + eval + ddot
    if (phase == 1) { if (tid == 0) gs.S[0] = st; return; }
    if (HOST_PASS1(FUN, P)) return;
    ma = block_max(ma, sha);
    block_argmax(ba, bv, bi, sha, shv, shi);
    if (nbl > 1) {
        STAMP(gs, 0);   // (several blocks: block reductions; the fold of the partials is not stamped)
        STAMP_END(gs, 0);
        // this block's partial, then the arrival counter; the last block folds all partials (blocks sit on different XCDs:
        // the fence pair makes the records visible across their L2s)
        LotPart *lp = P.lotp + (size_t)g * P.lot_nb;
        if (tid == 0) {
            LotPart r; r.ab = ba; r.bv = bv; r.ma = ma; r.il = bi; r.pad = 0;
            if (bi != INT_MAX) { r.i = lot[4 * bi]; r.j = lot[4 * bi + 1]; r.k = lot[4 * bi + 2]; r.q = lot[4 * bi + 3]; } else if (il_first < nlot) { r.i = lot[4 * il_first]; r.j = lot[4 * il_first + 1]; r.k = lot[4 * il_first + 2]; r.q = lot[4 * il_first + 3]; }   // no comparable candidate: the block's first one
            else { r.i = r.j = r.k = r.q = 1; }
            lp[blk] = r;
            __threadfence();
            s_last = (atomicAdd(&P.lot_ctr[g], 1u) == (unsigned)(nbl - 1));
        }
        __syncthreads();
        if (!s_last) return;
        if (tid == 0) {
            __threadfence();
            P.lot_ctr[g] = 0u;
            int wi = 0, wj = 0, wk = 0, wq = 0;
            ba = -1.0; bv = 0.0; bi = INT_MAX; ma = 0.0;
            for (int b = 0; b < nbl; b++) {
                const LotPart r = lp[b];
                ma = fmax(ma, r.ma);
                if (r.ab > ba || (r.ab == ba && r.il < bi)) { ba = r.ab; bv = r.bv; bi = r.il; wi = r.i; wj = r.j; wk = r.k; wq = r.q; }
            }
            if (bi == INT_MAX) { const LotPart r0 = lp[0]; wi = r0.i; wj = r0.j; wk = r0.k; wq = r0.q; }   // every residual a NaN: the first candidate, as idamax
            gs.amax = fmax(gs.amax, ma);                         // :467
            gs.neval += nlot;                                    // :465
            gs.rngpos += 2ull * nlot;
            st.ii = wi; st.jj = wj; st.kk = wk; st.qq = wq;      // :479-484
            st.pivot = bv;
            gs.S[0] = st;
        }
        return;
    }
    if (tid == 0) {
        gs.amax = fmax(gs.amax, ma);                             // :467
        gs.neval += nlot;                                        // :465
        gs.rngpos += 2ull * nlot;
        if (bi == INT_MAX) bi = 0;                               // every residual a NaN: the first candidate, as idamax
        st.ii = lot[4 * bi]; st.jj = lot[4 * bi + 1]; st.kk = lot[4 * bi + 2]; st.qq = lot[4 * bi + 3];   // :479-484
        st.pivot = bv;
        gs.S[0] = st;
    }
    STAMP(gs, 0);   // 7: reductions + state
    STAMP_END(gs, 0);
}

// ------------------------------------------------------------------------------------------------
// K_halfstep: one rook half-step = fiber evaluation (K1) + residual against the cross factor (K2, the
// "maxvol" kernel of the north star: a bandwidth-bound slab sweep) + block arg-max.
// lib/dmrgg.f90:518-582 (rook), :492-513 (piv=0).  grid = (fiber blocks, groups).
// mode 0: rook (type alternates, residual unless crs reached 2*piv); mode 1: piv=0 (h=0 column, h=1 row,
// no residual).
// ------------------------------------------------------------------------------------------------
template <int FUN>
__global__ __launch_bounds__(TTX_BLK) void k_halfstep(DevProb P, int h, int dir, int mode, int vals)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ StepState cur;
    __shared__ double sha[4], shv[4]; __shared__ int shi[4];
    const int g = blockIdx.y, tid = threadIdx.x, m = P.d;
    GroupState &gs = P.gs[g];
    STAMP_DECL;
    if (tid < 64) {
        resolve_state_wave(cur, gs.S[h], gs.Pt[(h + 1) & 1], tid);
        if (tid == 0 && (mode == 3 || mode == 4)) { cur.kk = ((int)blockIdx.z + P.zbase) % cur.n2 + 1; cur.qq = ((int)blockIdx.z + P.zbase) / cur.n2 + 1; }   // :356-369 column (k,q)
    }
    __syncthreads();
    const int zcol = (int)blockIdx.z + P.zbase;
    if (mode == 3 || mode == 4) { if (!cur.active || zcol >= cur.n2 * cur.r2) return; }
    else if (!cur.active || cur.done) { if (blockIdx.x == 0 && tid == 0) gs.S[h + 1] = cur; return; }
    STAMP(gs, 1);   // 0: resolve
    const bool iscol = (mode == 3 || mode == 4) ? true : (mode == 1 || mode == 2) ? (h == 0) : (((h + (dir == 2 ? 1 : 0)) & 1) == 0);    // :517,550
    const int p = cur.p, r0 = cur.r0, r1 = cur.r1, r2 = cur.r2, n1 = cur.n1, n2 = cur.n2, first = gs.first;
    const int nf = iscol ? r0 * n1 : n2 * r2;
    if ((int)(blockIdx.x * TTX_BLK) >= nf) return;
    // LDS: par | xs[RM] | then either VALUE rows (Ising C when they fit: node and weight doubles, one 16-byte
    // aligned row per varying index + one fixed row) or INDEX rows (short) for the generic integrands
    const int VS = ((m + 7) & ~7) + 8;
    double *par = dyn, *xs = dyn + ((P.npar + 1) & ~1);
    double *vbase = xs + ((P.RM + 1) & ~1);
    const double *Cp = core_ptr(P, P.col, g, p, first), *Wq = core_ptr(P, P.row, g, p + 1, first);
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = tid; x < P.npar; x += TTX_BLK) par[x] = P.par[x];
    const int vrows = iscol ? p - 1 : m - p - 1, vcols = iscol ? r0 : r2;
    const bool usev = (FUN == FUN_ISING) && (P.ising_id == 1) && (vals != 0);
    const int n1m = P.n[1];
    // TTX_ARITH=fast (ttx_fast.h): the fixed side of the fiber as a decay vector far[] (Ising D/E) or the cross terms of the
    // varying pivots with the fixed pivot (mvn), then O(small) work per element from the tables of k_fast_tables
    const bool fastp = fast_path(P, FUN);
    double *far = vbase, *Xv = vbase + ((max(P.FD, TTX_FNR) + 3) & ~1);
    __shared__ int s_nfar; __shared__ double s_rfix;
    const int fside = iscol ? 0 : 1;                                         // the varying side
    const int vfix = iscol ? cur.qq - 1 : cur.ii - 1, xfix = iscol ? cur.kk - 1 : cur.jj - 1;   // fixed pivot (other side), fixed mode index next to the free dim
    if (fastp && FUN == FUN_ISING) {
        const int bO = fside == 0 ? p + 1 : p - 1;             // the bond whose pivot set holds the fixed pivot
        const double *nearO = fast_near(P, 1 - fside, g, bO, first) + vfix, *pO = fast_piv(P, 1 - fside, g, bO, first) + vfix;
        const int cntO = (int)pO[FP_N * P.RM];
        const double xf = P.par[xfix];
        if (tid == 0) far[0] = 1.0;
        for (int b = tid; b < cntO; b += TTX_BLK) far[1 + b] = xf * nearO[(size_t)b * P.RM];
        __syncthreads();
        if (tid < 64) {          // ranges of the fixed side that begin / end at its mode index next to the free dim; entries above the cut
            double N = 1.0, D = 1.0; int cnt = 0;
            for (int b0 = 1; b0 <= cntO; b0 += 64) {
                const int b = b0 + tid;
                const double c = (b <= cntO) ? far[b] : 0.0;
                const bool on = c > TTX_FCUT;
                if (on) { N = N * (1.0 - c); D = D * (1.0 + c); }
                cnt += __popcll(__ballot(on));
            }
            N = wave_prod(N); D = wave_prod(D);
            if (tid == 0) { s_nfar = 1 + cnt; s_rfix = pO[FP_T * P.RM] * (N / D); }
        }
    }
    if (fastp && FUN == FUN_MVN) {
        const int nv = iscol ? r0 : r2;
        for (int v_ = tid; v_ < nv; v_ += TTX_BLK) Xv[v_] = iscol ? mvn_fast_cross(P, g, p, first, v_, vfix) : mvn_fast_cross(P, g, p, first, vfix, v_);
    }
    __syncthreads();      // par is read below when staging values
    if (usev) {
        // rows: [0 .. vcols) varying index, row vcols = fixed side; each row: VS node values then VS weight values
        double *fn = vbase + (size_t)vcols * 2 * VS, *fw = fn + VS;
        for (int x = tid; x < vcols * VS; x += TTX_BLK) {
            const int c = x / VS, o = x % VS;
            int ix = 1;
            if (o < vrows) ix = iscol ? Lt[(size_t)o * P.RM + c] : Rt[(size_t)o * P.RM + c];
            vbase[(size_t)c * 2 * VS + o] = par[ix - 1];
            vbase[(size_t)c * 2 * VS + VS + o] = par[n1m + ix - 1];
        }
        for (int x = tid; x < VS; x += TTX_BLK) {
            int ix = 1;
            if (iscol) { if (x == 0) ix = cur.kk; else if (x < m - p) ix = Rt[(size_t)(x - 1) * P.RM + (cur.qq - 1)]; }
            else       { if (x < p - 1) ix = Lt[(size_t)x * P.RM + (cur.ii - 1)]; else if (x == p - 1) ix = cur.jj; }
            fn[x] = par[ix - 1]; fw[x] = par[n1m + ix - 1];
        }
    }
    const bool usem = (FUN == FUN_MVN) && (vals != 0);      // rows of differences x - mu (doubles): [vcols][VS] + fixed row
    if (usem) {
        const double *mu = P.aux;
        double *df = vbase + (size_t)vcols * VS;
        if (iscol) {
            for (int x = tid; x < vcols * VS; x += TTX_BLK) { const int c = x / VS, o = x % VS; vbase[x] = (o < vrows) ? par[Lt[(size_t)o * P.RM + c] - 1] - mu[o] : 0.0; }
            for (int x = tid; x < VS; x += TTX_BLK)
                df[x] = (x == 0) ? par[cur.kk - 1] - mu[p] : (x < m - p) ? par[Rt[(size_t)(x - 1) * P.RM + (cur.qq - 1)] - 1] - mu[p + x] : 0.0;
        } else {
            for (int x = tid; x < vcols * VS; x += TTX_BLK) { const int c = x / VS, o = x % VS; vbase[x] = (o < vrows) ? par[Rt[(size_t)o * P.RM + c] - 1] - mu[p + 1 + o] : 0.0; }
            for (int x = tid; x < VS; x += TTX_BLK)
                df[x] = (x < p - 1) ? par[Lt[(size_t)x * P.RM + (cur.ii - 1)] - 1] - mu[x] : (x == p - 1) ? par[cur.jj - 1] - mu[p - 1] : 0.0;
        }
    }
    short *fxs = (short *)vbase;
    short *vt = fxs + VS;
    if (!usev && !usem && !fastp) {
        if (iscol) {   // varying: left pivot i (dims 1..p-1) and j; fixed: kk and the right multi-index of qq (dims p+1..m)
            for (int x = tid; x < vcols * VS; x += TTX_BLK) { const int c = x / VS, o = x % VS; vt[x] = (o < vrows) ? Lt[(size_t)o * P.RM + c] : (short)1; }
            for (int x = tid; x < VS; x += TTX_BLK) fxs[x] = (x == 0) ? (short)cur.kk : (x < m - p) ? Rt[(size_t)(x - 1) * P.RM + (cur.qq - 1)] : (short)1;
        } else {       // varying: k and right pivot q (dims p+2..m); fixed: left multi-index of ii and jj (dims 1..p)
            for (int x = tid; x < vcols * VS; x += TTX_BLK) { const int c = x / VS, o = x % VS; vt[x] = (o < vrows) ? Rt[(size_t)o * P.RM + c] : (short)1; }
            for (int x = tid; x < VS; x += TTX_BLK) fxs[x] = (x < p - 1) ? Lt[(size_t)x * P.RM + (cur.ii - 1)] : (x == p - 1) ? (short)cur.jj : (short)1;
        }
    }
    if (iscol) for (int s = tid; s < r1; s += TTX_BLK) xs[s] = Wq[(cur.kk - 1) + (size_t)P.NM * (cur.qq - 1) + P.SW * s];
    else       for (int s = tid; s < r1; s += TTX_BLK) xs[s] = Cp[(cur.ii - 1) + (size_t)P.RM * (cur.jj - 1) + P.SS * s];
    __syncthreads();
    STAMP(gs, 1);   // 1: staging
    const int t = blockIdx.x * TTX_BLK + tid;
    const bool live = t < nf;
    double a = 0.0;
    int u = 0, v = 0;                         // col: (i,j) 0-based ; row: (k,q) 0-based
    if (live) {
        if (iscol) { u = t % r0; v = t / r0; } else { u = t % n2; v = t / n2; }
        if (fastp && FUN == FUN_ISING) {
            const int pv = iscol ? u : v, nd = iscol ? v : u;                // varying pivot, free mode index (0-based)
            const int bV = fside == 0 ? p - 1 : p + 1;
            const double *nearV = fast_near(P, fside, g, bV, first) + pv, *pV = fast_piv(P, fside, g, bV, first) + pv;
            const double *pO = fast_piv(P, 1 - fside, g, fside == 0 ? p + 1 : p - 1, first) + vfix;
            const double xn = par[nd], xf = par[xfix];
            double N = 1.0, D = 1.0;
            if (s_nfar <= TTX_FNR) {
                FarReg C;
                far_load(C, far, s_nfar);
                de_fast_span_reg(nearV, (size_t)P.RM, (int)pV[FP_N * P.RM], xn, C, N, D);
            } else
                de_fast_span(nearV, (size_t)P.RM, (int)pV[FP_N * P.RM], xn, far, s_nfar, N, D);
            const double rho = pV[FP_T * P.RM] * (N / D) * s_rfix;
            a = iscol ? de_fast_value(P.ising_id, P.RM, rho, pV, xn, par[n1m + nd], xf, par[n1m + xfix], pO)
                      : de_fast_value(P.ising_id, P.RM, rho, pO, xf, par[n1m + xfix], xn, par[n1m + nd], pV);
        } else if (fastp && FUN == FUN_MVN) {
            a = iscol ? mvn_fast_value(P, g, p, first, u, v, xfix, vfix, Xv[u]) : mvn_fast_value(P, g, p, first, vfix, xfix, u, v, Xv[v]);
        } else if (usev) {
            const double *fn = vbase + (size_t)vcols * 2 * VS, *fw = fn + VS;
            if (iscol) { const double *rn = vbase + (size_t)u * 2 * VS; a = f_ising_c3v(m, p - 1, rn, rn + VS, par[v], par[n1m + v], fn, fw); }
            else       { const double *rn = vbase + (size_t)v * 2 * VS; a = f_ising_c3v(m, p, fn, fw, par[u], par[n1m + u], rn, rn + VS); }
        } else if (FUN == FUN_ISING && P.ising_id != 1 && P.deTL) {
            const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
            const double *TL = P.deTL + (size_t)g * tsz, *UL = P.deUL + (size_t)g * P.RM * (m + 1), *TR = P.deTR + (size_t)g * tsz;
            double pa_;
            const int *CL = P.deCL + (size_t)g * P.RM * (m + 1), *CR = P.deCR + (size_t)g * P.RM * (m + 1);
            if (P.de_cut) {
                if (iscol) pa_ = de_pairs_ctab(m, par - 1, p - 1, TL + (size_t)u * NP, CL + (size_t)u * (m + 1), UL + (size_t)u * (m + 1), v + 1, cur.kk, fxs, 1,
                                               TR + (size_t)(cur.qq - 1) * NP, CR + (size_t)(cur.qq - 1) * (m + 1));
                else pa_ = de_pairs_ctab(m, par - 1, p - 1, TL + (size_t)(cur.ii - 1) * NP, CL + (size_t)(cur.ii - 1) * (m + 1), UL + (size_t)(cur.ii - 1) * (m + 1), cur.jj, u + 1,
                                         vt + (size_t)v * VS, 0, TR + (size_t)v * NP, CR + (size_t)v * (m + 1));
                a = iscol ? de_finish<1>(P.ising_id, pa_, m, n1m, par, vt + (size_t)u * VS, p - 1, v + 1, 0, fxs) : de_finish<1>(P.ising_id, pa_, m, n1m, par, fxs, p, u + 1, 0, vt + (size_t)v * VS);
            } else
            if (iscol) {     // left pivot u varies, s1 = v+1, s2 = kk, right pivot qq fixed (fxs = [kk, right dims])
                pa_ = de_pairs_tab<false>(m, par - 1, p - 1, TL + (size_t)u * NP, UL + (size_t)u * (m + 1), v + 1, cur.kk, fxs, 1, TR + (size_t)(cur.qq - 1) * NP);
                a = de_finish<1>(P.ising_id, pa_, m, n1m, par, vt + (size_t)u * VS, p - 1, v + 1, 0, fxs);
            } else {         // left pivot ii fixed (fxs = [left dims, jj]), s1 = jj, s2 = u+1, right pivot v varies
                pa_ = de_pairs_tab<false>(m, par - 1, p - 1, TL + (size_t)(cur.ii - 1) * NP, UL + (size_t)(cur.ii - 1) * (m + 1), cur.jj, u + 1, vt + (size_t)v * VS, 0, TR + (size_t)v * NP);
                a = de_finish<1>(P.ising_id, pa_, m, n1m, par, fxs, p, u + 1, 0, vt + (size_t)v * VS);
            }
        } else if (usem) {
            const double *mu = P.aux, *df = vbase + (size_t)vcols * VS;
            if (iscol) a = f_mvn_rows<1>(m, P.auxT, P.mvn_norm, vbase + (size_t)u * VS, p - 1, par[v] - mu[p - 1], 0.0, df);
            else       a = f_mvn_rows<1>(m, P.auxT, P.mvn_norm, df, p, par[u] - mu[p], 0.0, vbase + (size_t)v * VS);
        } else {
            Src3 sx;
            if (iscol) { sx.pa = vt + (size_t)u * VS; sx.A = p - 1; sx.self = v + 1; sx.pb = fxs; }
            else       { sx.pa = fxs; sx.A = p; sx.self = u + 1; sx.pb = vt + (size_t)v * VS; }
            a = eval_src3<FUN, true>(P, par, sx, (long)g * P.HS + t);         // :520-526 / :553-559
        }
        if (mode == 4) P.sb[(size_t)g * ((size_t)P.RM * P.NM) * ((size_t)P.RM * P.NM) + (size_t)nf * zcol + t] = a;   // superblock column (k,q)
        else if (mode != 3) (iscol ? P.acol : P.arow)[(size_t)g * P.RM * P.NM + t] = a;
    }
    STAMP(gs, 1);   // 2: eval
    if (HOST_PASS1(FUN, P)) return;
    double mx = block_max(live ? fabs(a) : 0.0, sha);
    if (tid == 0 && mode != 1) atomic_max_pos(&gs.amax, mx);                  // :531 / :564 (the piv = 0 branch :492-513 does not touch amax)
    const int crs = cur.crs + 1;
    const int havecol = cur.havecol | (iscol ? 1 : 0), haverow = cur.haverow | (iscol ? 0 : 1);
    const int done = (mode == 1 || mode == 2) ? (h == 1) : (havecol && haverow && (crs >= 2 * P.piv));   // :534 / :567
    const bool resid = (mode == 3) || ((mode == 0) && !done);
    if (resid) {
        double b = a, ab = -1.0; int bi = INT_MAX;
        if (live) {
            if (iscol) {   // dgemv 'n', alpha=-1 (:538): b += (-x_s) * col(:, s)
                const double *c = Cp + u + (size_t)P.RM * v;
#pragma unroll 8
                for (int s = 0; s < r1; s++) b = b + (-xs[s]) * c[P.SS * s];
            } else {       // dgemv 't', alpha=-1 (:571): b += -1 * sum_s row(s, kq) * x_s
                const double *w = Wq + u + (size_t)P.NM * v;
                double tt = 0.0;
#pragma unroll 8
                for (int s = 0; s < r1; s++) tt = tt + w[P.SW * s] * xs[s];
                b = b + (-1.0) * tt;
            }
            ab = fabs(b); bi = t;
        }
        STAMP(gs, 1);   // 3: residual
        block_argmax(ab, b, bi, sha, shv, shi);
        if (tid == 0) {
            Partial pr; pr.absmax = ab; pr.val = b; pr.idx = bi; pr.pad = 0;
            if (mode == 3) {   // superblock linear index ijkq = t + r0*n1*(column), :388-394
                if (bi != INT_MAX) pr.idx = bi + r0 * n1 * zcol;
                P.pfull[((size_t)g * P.NM * P.RM + zcol) * P.nfb + blockIdx.x] = pr;
            } else gs.Pt[h & 1][blockIdx.x] = pr;
        }
    }
    if (mode == 3 || mode == 4) {
        if (blockIdx.x == 0 && zcol == 0 && tid == 0 && P.hostpass != 1) gs.neval += (long long)r0 * n1 * n2 * r2;   // :372
        return;
    }
    if (blockIdx.x == 0 && tid == 0) {
        StepState nx = cur;
        nx.crs = crs; nx.havecol = havecol; nx.haverow = haverow; nx.done = done;
        nx.pending = resid ? (iscol ? 1 : 2) : 0;
        nx.npart = (nf + TTX_BLK - 1) / TTX_BLK;
        gs.S[h + 1] = nx;
        if (mode != 2) gs.neval += nf;                                        // :527 / :560 / :509
        // algorithmic traffic: factor slabs + vector + fiber in/out when a residual is taken, else the fiber
        gs.bytes_half += resid ? 8.0 * ((double)nf * r1 + r1 + 2.0 * nf) : 8.0 * nf;
        gs.n_resid += resid ? 1 : 0;
    }
    STAMP(gs, 1);   // 4: argmax + state
    STAMP_END(gs, 1);
}

// ------------------------------------------------------------------------------------------------
// K_accept: threshold test and in-place append of the new cross (lib/dmrgg.f90:598-758).
// block roles along grid.x: [0,nA) column factor + raw column; [nA,2nA) row factor + raw row;
// [2nA, 2nA+NM) left-neighbour row fix-up (one column j each); [.., +NM) right-neighbour column fix-up
// (one row k each); last block: scalars, packed LU, pivot sets and index tables.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TTX_BLK) void k_accept(DevProb P, int H, int nA)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ StepState cur;
    __shared__ int s_upd;
    __shared__ double s_bc;
    const int g = blockIdx.y, tid = threadIdx.x, m = P.d;
    GroupState &gs = P.gs[g];
    if (tid < 64) resolve_state_wave(cur, gs.S[H], gs.Pt[(H + 1) & 1], tid);
    if (tid == 0) {
        s_upd = cur.active && (fabs(cur.pivot) > P.small_element * gs.amax) && (fabs(cur.pivot) > P.small_pivot * gs.pivotmax_prev);  // :599-600
    }
    __syncthreads();
    if (!cur.active) return;
    const int p = cur.p, r0 = cur.r0, r1 = cur.r1, r2 = cur.r2, n1 = cur.n1, n2 = cur.n2, first = gs.first;
    const int bx = blockIdx.x, nblk = gridDim.x;
    int *tape = P.tape + ((size_t)g * (m + 2) + p) * 4;
    if (!s_upd) {
        if (bx == nblk - 1 && tid == 0) { tape[0] = tape[1] = tape[2] = tape[3] = -1; P.upd[(size_t)g * (m + 2) + p] = 0; }
        return;
    }
    const int ii = cur.ii - 1, jj = cur.jj - 1, kk = cur.kk - 1, qq = cur.qq - 1;   // 0-based
    double *Ap = core_ptr(P, P.arg, g, p, first), *Aq = core_ptr(P, P.arg, g, p + 1, first);
    double *Cp = core_ptr(P, P.col, g, p, first), *Wq = core_ptr(P, P.row, g, p + 1, first);
    const double *acol = P.acol + (size_t)g * P.RM * P.NM, *arow = P.arow + (size_t)g * P.RM * P.NM;
    double *xs = dyn;   // RM doubles

    if (bx < nA) {
        // role A: arg(p)(:,:,r1+1) = acol1 (:662-668); col(p)(:,:,r1+1) = (acol1 - col*Ucol) / pivot (:701, d2_lual from=r1+1)
        for (int s = tid; s < r1; s += TTX_BLK) xs[s] = Wq[kk + (size_t)P.NM * qq + P.SW * s];
        __syncthreads();
        int t = bx * TTX_BLK + tid;
        if (t < r0 * n1) {
            int i = t % r0, j = t / r0;
            size_t o = i + (size_t)P.RM * j;
            double a = acol[t];
            Ap[o + P.SS * r1] = a;
            double y = a;
            for (int s = 0; s < r1; s++) y = y + (-xs[s]) * Cp[o + P.SS * s];
            y = (1.0 / cur.pivot) * y;
            Cp[o + P.SS * r1] = y;
        }
    } else if (bx < 2 * nA) {
        // role B: arg(p+1)(r1+1,:,:) = arow1 (:669-674); row(p+1)(r1+1,:,:) = arow1 - Lrow'*row (:702, d2_luar from=r1+1)
        for (int s = tid; s < r1; s += TTX_BLK) xs[s] = Cp[ii + (size_t)P.RM * jj + P.SS * s];
        __syncthreads();
        int t = (bx - nA) * TTX_BLK + tid;
        if (t < n2 * r2) {
            int k = t % n2, q = t / n2;
            double a = arow[t];
            Aq[r1 + (size_t)P.RM * k + P.SS * q] = a;
            size_t o = k + (size_t)P.NM * q;
            double tt = 0.0;
            for (int s = 0; s < r1; s++) tt = tt + Wq[o + P.SW * s] * xs[s];
            Wq[o + P.SW * r1] = a + (-1.0) * tt;
        }
    } else if (bx < 2 * nA + P.NM) {
        // role C: row(p)(:, j, r1+1) = L(p-1)^-1 * acol1(:, j)  (:715-728, d2_luar full) -- wavefront over i
        int j = bx - 2 * nA;
        if (p <= first || j >= n1) return;
        const double *gI = inv_ptr(P, g, p - 1, first);
        double *Wp = core_ptr(P, P.row, g, p, first);
        double a = (tid < r0) ? acol[tid + r0 * j] : 0.0, tmp = 0.0, xf = 0.0;
        if (r0 <= 64) {                               // one wave: the solve runs as a shuffle wavefront over the LDS-staged LU (as k_exch_boundary)
            double *lu = dyn;
            for (int x = tid; x < r0 * r0; x += TTX_BLK) lu[x] = gI[x];
            __syncthreads();
            if (tid < 64)
                for (int s = 0; s < r0; s++) {
                    const double cand = (s == 0) ? a : a + (-1.0) * tmp;
                    const double xsv = __shfl(cand, s, 64);
                    if (tid == s) xf = xsv;
                    if (tid > s && tid < r0) tmp = tmp + xsv * lu[tid * tid + s];
                }
        } else
        for (int s = 0; s < r0; s++) {
            if (tid == s) { xf = a + (-1.0) * tmp; if (s == 0) xf = a; s_bc = xf; }
            __syncthreads();
            if (tid > s && tid < r0) tmp = tmp + s_bc * gI[tid * tid + s];
            __syncthreads();
        }
        if (tid < r0) Wp[j + (size_t)P.NM * r1 + P.SW * tid] = xf;
    } else if (bx < 2 * nA + 2 * P.NM) {
        // role D: col(p+1)(r1+1, k, :) = arow1(k, :) * U(p+1)^-1  (:730-749, d2_lual full) -- wavefront over q
        int k = bx - 2 * nA - P.NM;
        if (p >= gs.last || k >= n2) return;
        const double *gI = inv_ptr(P, g, p + 1, first);
        double *Cq = core_ptr(P, P.col, g, p + 1, first);
        double y = (tid < r2) ? arow[k + n2 * tid] : 0.0;
        if (r2 <= 64) {
            double *lu = dyn;
            for (int x = tid; x < r2 * r2; x += TTX_BLK) lu[x] = gI[x];
            __syncthreads();
            if (tid < 64) {
                const double rdg = (tid < r2) ? 1.0 / lu[(tid + 1) * (tid + 1) - 1] : 0.0;
                for (int s = 0; s < r2; s++) {
                    const double cand = rdg * y;          // only lane s's product is used
                    const double ys = __shfl(cand, s, 64);
                    if (tid == s) y = ys;
                    if (tid > s && tid < r2) y = y + (-lu[tid * tid + tid + s]) * ys;
                }
            }
        } else
        for (int s = 0; s < r2; s++) {
            if (tid == s) { y = (1.0 / gI[(s + 1) * (s + 1) - 1]) * y; s_bc = y; }
            __syncthreads();
            if (tid > s && tid < r2) y = y + (-gI[tid * tid + tid + s]) * s_bc;
            __syncthreads();
        }
        if (tid < r2) Cq[r1 + (size_t)P.RM * k + P.SS * tid] = y;
    } else {
        // role E: scalars (:604-635), packed LU (:649-660), index tables (replaces the vip walk of :1062-1075)
        double *gI = inv_ptr(P, g, p, first);
        for (int s = tid; s < r1; s += TTX_BLK) {
            gI[r1 * r1 + s] = Cp[ii + (size_t)P.RM * jj + P.SS * s];
            gI[r1 * r1 + r1 + s] = Wq[kk + (size_t)P.NM * qq + P.SW * s];
        }
        short *Ln = L_ptr(P, g, p, first), *Rn = R_ptr(P, g, p, first);
        const short *Lo = L_ptr(P, g, p - 1, first), *Ro = R_ptr(P, g, p + 1, first);
        for (int x = tid; x < p; x += TTX_BLK) Ln[(size_t)x * P.RM + r1] = (x < p - 1) ? Lo[(size_t)x * P.RM + ii] : (short)(jj + 1);
        for (int x = tid; x < m - p; x += TTX_BLK) Rn[(size_t)x * P.RM + r1] = (x == 0) ? (short)(kk + 1) : Ro[(size_t)(x - 1) * P.RM + qq];
        if (P.arith && P.fpersist && tid < 128) {
            // TTX_ARITH=fast, Ising D/E: the table entries of the new pivot as a LEFT multi-index of bond p (parent: left pivot ii of bond
            // p-1, extended by node jj) and as a RIGHT multi-index of bond p (parent: right pivot qq of bond p+1, extended by node kk)
            const int sd = tid >> 6, lane_ = tid & 63;
            const double *pn = fast_near(P, sd, g, sd == 0 ? p - 1 : p + 1, first) + (sd == 0 ? ii : qq);
            const double *pp_ = fast_piv(P, sd, g, sd == 0 ? p - 1 : p + 1, first) + (sd == 0 ? ii : qq);
            const int nd = sd == 0 ? jj : kk;
            if (P.fun_id == FUN_MVN) {       // mvn: the new dimension is p-1 (0-based) behind the left parent's p-1 dims, p in front of the right parent's m-p-1
                const int dnew = sd == 0 ? p - 1 : p, plen = sd == 0 ? p - 1 : m - p - 1;
                mvn_entry_child(P, sd, fast_dv(P, sd, g, sd == 0 ? p - 1 : p + 1, first) + (sd == 0 ? ii : qq), pn, pp_, plen, dnew, P.par[nd] - P.aux[dnew],
                                fast_dv(P, sd, g, p, first) + r1, fast_near(P, sd, g, p, first) + r1, fast_piv(P, sd, g, p, first) + r1, lane_);
            } else
            fast_entry_child(pn, pp_, P.par[nd], P.par[P.n[1] + nd], fast_near(P, sd, g, p, first) + r1, fast_piv(P, sd, g, p, first) + r1, P.RM, lane_);
        }
        if (tid == 0) {
            gI[(r1 + 1) * (r1 + 1) - 1] = cur.pivot;
            int *vp = vip_ptr(P, g, p, first) + 4 * r1;
            vp[0] = tape[0] = ii + 1; vp[1] = tape[1] = jj + 1; vp[2] = tape[2] = kk + 1; vp[3] = tape[3] = qq + 1;
            double ap = fabs(cur.pivot);
            gs.pivotmax = (gs.pivotmax < 0.0) ? ap : fmax(gs.pivotmax, ap);
            gs.pivotmin = (gs.pivotmin < 0.0) ? ap : fmin(gs.pivotmin, ap);
            P.upd[(size_t)g * (m + 2) + p] = 1;
            P.r[(size_t)g * (m + 2) + p] = r1 + 1;                                      // :752
        }
    }
}

// ------------------------------------------------------------------------------------------------
// quadrature (lib/dmrgg.f90:975-993 per sweep, :1261-1415 dtt_quad) and finalisation dtt_lua (:1169-1258)
// ------------------------------------------------------------------------------------------------
// mode 0: T = sum_j w_j arg(:,j,:) on raw fibers, then L(p-1)^-1 T U(p)^-1 (ttqq + dtt_lua);
// mode 1: T = sum_j w_j arg(:,j,:) on finalised cores (w == NULL: plain sum)
__global__ __launch_bounds__(256) void k_quad_build(DevProb P, int mode, const double *wq)
{
    if (P.ctl[2]) return;
    extern __shared__ double T[];   // RM*RM
    const int g = blockIdx.y, tid = threadIdx.x, RM = P.RM;
    GroupState &gs = P.gs[g];
    const int first = gs.first, p = first + blockIdx.x;
    const bool lastgroup = (gs.gglobal == P.nprocs - 1);
    if (p > gs.last && !(lastgroup && p == P.d)) return;
    const int *r = P.r + (size_t)g * (P.d + 2);
    const int r0 = r[p - 1], r1 = r[p], n = P.n[p];
    const double *A = core_ptr(P, P.arg, g, p, first);
    const double *w = wq ? wq + (size_t)p * P.NM : nullptr;
    for (int x = tid; x < r0 * r1; x += blockDim.x) {
        int i = x % r0, k = x / r0;
        const double *a = A + i + P.SS * k;
        double y = 0.0;
        if (w) {
#pragma unroll 8
            for (int j = 0; j < n; j++) y = y + w[j] * a[(size_t)RM * j];        // dgemv 'n', :988 / :1327
        } else {
#pragma unroll 8
            for (int j = 0; j < n; j++) y = y + a[(size_t)RM * j];               // :1331
        }
        T[i + RM * k] = y;
    }
    __syncthreads();
    if (mode == 0) {
        const double *gL = inv_ptr(P, g, p - 1, first);
        if (tid < r1) {                               // d2_luar(n*r1 -> r1 columns, r0, inv(p-1)) :1250
            double *c = T + RM * tid;
            for (int t = 1; t < r0; t++) {
                double tmp = 0.0;
                for (int s = 0; s < t; s++) tmp = tmp + c[s] * gL[t * t + s];
                c[t] = c[t] + (-1.0) * tmp;
            }
        }
        __syncthreads();
        if (p <= gs.last && tid < r0) {               // d2_lual(r0 rows, r1, inv(p)) :1251
            const double *gU = inv_ptr(P, g, p, first);
            for (int t = 0; t < r1; t++) {
                double y = T[tid + RM * t];
                for (int s = 0; s < t; s++) y = y + (-gU[t * t + t + s]) * T[tid + RM * s];
                T[tid + RM * t] = (1.0 / gU[(t + 1) * (t + 1) - 1]) * y;
            }
        }
        __syncthreads();
    }
    double *out = P.Tq + ((size_t)g * P.NC + (p - first)) * (size_t)RM * RM;
    for (int x = tid; x < r0 * r1; x += blockDim.x) out[(x % r0) + RM * (x / r0)] = T[(x % r0) + RM * (x / r0)];
}

// chain product of the group's T matrices (dgemm 'n','n' order, :1340); one block per group
__global__ __launch_bounds__(256) void k_quad_chain(DevProb P)
{
    if (P.ctl[2]) return;
    extern __shared__ double sh[];  // 2*RM*RM
    const int g = blockIdx.x, tid = threadIdx.x, RM = P.RM;
    GroupState &gs = P.gs[g];
    const int first = gs.first;
    const bool lastgroup = (gs.gglobal == P.nprocs - 1);
    const int lastc = lastgroup ? P.d : gs.last;
    const int *r = P.r + (size_t)g * (P.d + 2);
    const int mym = r[first - 1];
    // the two chain matrices: LDS, or the group's global scratch when 2 RM^2 doubles exceed it (same block: barriers order the accesses)
    double *prev = P.qscr ? P.qscr + (size_t)g * 2 * RM * RM : sh, *next = prev + RM * RM;
    const double *T0 = P.Tq + ((size_t)g * P.NC) * (size_t)RM * RM;
    for (int x = tid; x < mym * r[first]; x += blockDim.x) prev[(x % mym) + RM * (x / mym)] = T0[(x % mym) + RM * (x / mym)];
    __syncthreads();
    for (int p = first + 1; p <= lastc; p++) {
        const double *Tc = P.Tq + ((size_t)g * P.NC + (p - first)) * (size_t)RM * RM;
        const int r0 = r[p - 1], r1 = r[p];
        for (int x = tid; x < mym * r1; x += blockDim.x) {
            int i = x % mym, j = x / mym;
            double c = 0.0;
#pragma unroll 8
            for (int l = 0; l < r0; l++) c = c + Tc[l + RM * j] * prev[i + RM * l];
            next[i + RM * j] = c;
        }
        __syncthreads();
        double *t = prev; prev = next; next = t;
    }
    const int myn = r[lastc];
    const int gg = gs.gglobal;
    double *out = P.qsend + (size_t)gg * RM * RM;
    for (int x = tid; x < mym * myn; x += blockDim.x) out[(x % mym) + RM * (x / mym)] = prev[(x % mym) + RM * (x / mym)];
    if (tid == 0) {
        double *dm = P.qsend + (size_t)P.nprocs * RM * RM;
        dm[2 * gg] = (double)mym; dm[2 * gg + 1] = (double)myn;
        if (P.nprocs == 1) gs.val = prev[0];
    }
}

// dtt_lua on the raw fibers: luar pass then lual pass (two launches: the second needs the whole core)
// lds != 0: the block's 256 columns (rows) and the packed LU live in LDS while the substitution runs -- same operations
// in the same order, but the O(r^2) walk no longer goes through global memory (90 -> ~25 us per launch at r = 20..32)
__global__ __launch_bounds__(256) void k_fin_luar(DevProb P, int lds)
{
    extern __shared__ double fsh[];
    const int g = blockIdx.y, RM = P.RM;
    GroupState &gs = P.gs[g];
    const int first = gs.first, p = first + blockIdx.x;
    const bool lastgroup = (gs.gglobal == P.nprocs - 1);
    if (p > gs.last && !(lastgroup && p == P.d)) return;
    const int *r = P.r + (size_t)g * (P.d + 2);
    const int r0 = r[p - 1], r1 = r[p], n = P.n[p];
    double *A = core_ptr(P, P.arg, g, p, first);
    const double *gL = inv_ptr(P, g, p - 1, first);
    if (lds) {
        double *sg = fsh, *sc = fsh + (size_t)RM * RM;               // LU | columns as [t][thread]
        const int tid = threadIdx.x;
        // the columns of a core are spread over the workgroups along grid.z (T = blockDim.x columns each): every column is independent
        const int T = blockDim.x;
        if ((int)(blockIdx.z * T) >= n * r1) return;
        for (int x = tid; x < r0 * r0; x += T) sg[x] = gL[x];
        for (int x0 = blockIdx.z * T; x0 < n * r1; x0 += gridDim.z * T) {
            const int x = x0 + tid;
            const bool on = x < n * r1;
            double *c = on ? A + (size_t)RM * (x % n) + P.SS * (x / n) : A;
            __syncthreads();
            if (on) for (int t = 0; t < r0; t++) sc[t * T + tid] = c[t];
            __syncthreads();                                         // (also: sg complete before the first use)
            if (on) {
                for (int t = 1; t < r0; t++) {
                    double tmp = 0.0;
#pragma unroll 8
                    for (int s = 0; s < t; s++) tmp = tmp + sc[s * T + tid] * sg[t * t + s];
                    sc[t * T + tid] = sc[t * T + tid] + (-1.0) * tmp;
                }
                for (int t = 1; t < r0; t++) c[t] = sc[t * T + tid];
            }
        }
        return;
    }
    for (int x = blockIdx.z * blockDim.x + threadIdx.x; x < n * r1; x += gridDim.z * blockDim.x) {       // one column (j,k) per thread, :1250
        double *c = A + (size_t)RM * (x % n) + P.SS * (x / n);
        for (int t = 1; t < r0; t++) {
            double tmp = 0.0;
            for (int s = 0; s < t; s++) tmp = tmp + c[s] * gL[t * t + s];
            c[t] = c[t] + (-1.0) * tmp;
        }
    }
}
__global__ __launch_bounds__(256) void k_fin_lual(DevProb P, int lds)
{
    extern __shared__ double fsh[];
    const int g = blockIdx.y, RM = P.RM;
    GroupState &gs = P.gs[g];
    const int first = gs.first, p = first + blockIdx.x;
    if (p > gs.last) return;
    const int *r = P.r + (size_t)g * (P.d + 2);
    const int r0 = r[p - 1], r1 = r[p], n = P.n[p];
    double *A = core_ptr(P, P.arg, g, p, first);
    const double *gU = inv_ptr(P, g, p, first);
    if (lds) {
        double *sg = fsh, *sc = fsh + (size_t)RM * RM;
        const int tid = threadIdx.x;
        const int T = blockDim.x;
        if ((int)(blockIdx.z * T) >= r0 * n) return;
        for (int x = tid; x < r1 * r1; x += T) sg[x] = gU[x];
        for (int x0 = blockIdx.z * T; x0 < r0 * n; x0 += gridDim.z * T) {
            const int x = x0 + tid;
            const bool on = x < r0 * n;
            double *c = on ? A + (x % r0) + (size_t)RM * (x / r0) : A;
            __syncthreads();
            if (on) for (int t = 0; t < r1; t++) sc[t * T + tid] = c[P.SS * t];
            __syncthreads();
            if (on) {
                for (int t = 0; t < r1; t++) {
                    double y = sc[t * T + tid];
#pragma unroll 8
                    for (int s = 0; s < t; s++) y = y + (-sg[t * t + t + s]) * sc[s * T + tid];
                    sc[t * T + tid] = (1.0 / sg[(t + 1) * (t + 1) - 1]) * y;
                }
                for (int t = 0; t < r1; t++) c[P.SS * t] = sc[t * T + tid];
            }
        }
        return;
    }
    for (int x = blockIdx.z * blockDim.x + threadIdx.x; x < r0 * n; x += gridDim.z * blockDim.x) {       // one row (i,j) per thread, :1251
        double *c = A + (x % r0) + (size_t)RM * (x / r0);
        for (int t = 0; t < r1; t++) {
            double y = c[P.SS * t];
            for (int s = 0; s < t; s++) y = y + (-gU[t * t + t + s]) * c[P.SS * s];
            c[P.SS * t] = (1.0 / gU[(t + 1) * (t + 1) - 1]) * y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stand-alone kernels behind the ttx_k_* test entry points (same device functions as the sweep)
// ------------------------------------------------------------------------------------------------
struct ListIdx { const int *ind; __device__ __forceinline__ int operator()(int s) const { return ind[s - 1]; } };
template <int FUN>
__global__ void k_eval_list(DevProb P, long long npts, const int *ind, double *out)
{
    extern __shared__ __align__(16) double dyn[];
    for (int x = threadIdx.x; x < P.npar; x += blockDim.x) dyn[x] = P.par[x];
    __syncthreads();
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npts) return;
    ListIdx ix{ind + t * P.d};
    out[t] = eval_fun<FUN>(P, dyn, ix);
}

// b = a - F x (dgemv 'n' order) + per-block first arg-max; F is m x r with leading dimension ld
__global__ __launch_bounds__(TTX_BLK) void k_resid_argmax(int m, int r, size_t ld, const double *a, const double *F, const double *x,
                                                          double *b_out, Partial *parts)
{
    __shared__ double xs[256];
    __shared__ double sha[4], shv[4]; __shared__ int shi[4];
    for (int s = threadIdx.x; s < r; s += TTX_BLK) xs[s] = x[s];
    __syncthreads();
    int t = blockIdx.x * TTX_BLK + threadIdx.x;
    double b = 0.0, ab = -1.0; int bi = INT_MAX;
    if (t < m) {
        b = a[t];
#pragma unroll 8
        for (int s = 0; s < r; s++) b = b + (-xs[s]) * F[t + ld * s];
        b_out[t] = b; ab = fabs(b); bi = t;
    }
    block_argmax(ab, b, bi, sha, shv, shi);
    if (threadIdx.x == 0) { Partial pr; pr.absmax = ab; pr.val = b; pr.idx = bi; pr.pad = 0; parts[blockIdx.x] = pr; }
}

__global__ void k_lottery_only(int npnt, int m, int n, int nz, const int *zcol, const int *zrow, unsigned long long rngpos, int *points)
{
    __shared__ ttx_cdfseg segc[TTX_MAXSEG], segr[TTX_MAXSEG];
    __shared__ int nsc, nsr;
    if (threadIdx.x == 0) nsc = ttx_cdf_build(m - nz, segc);
    if (threadIdx.x == 64) nsr = ttx_cdf_build(n - nz, segr);
    __syncthreads();
    for (int il = threadIdx.x; il < npnt; il += blockDim.x) {
        double d1 = ttx_flang_draw(rngpos + il), d2 = ttx_flang_draw(rngpos + npnt + il);
        points[il] = ttx_lottery_index(segc, nsc, m - nz, m, zcol, nz, d1);
        points[npnt + il] = ttx_lottery_index(segr, nsr, n - nz, n, zrow, nz, d2);
    }
}

// ------------------------------------------------------------------------------------------------
// per-sweep neighbour exchange between bond groups (lib/dmrgg.f90:763-958; the right-going block lost in
// the fp64 source is restated from lib/dmrggmp.f90:572-629).  Instead of propagating the 4-int pivot tape
// through every rank (which makes far bonds lag by one sweep per hop), the owner ships the FLATTENED
// multi-index of its new boundary pivot to its direct neighbour -- the only rank that ever evaluates with it.
// ------------------------------------------------------------------------------------------------
// pack both outgoing messages of group g and its entry of the MAX all-reduce (called by every thread of a block)
__device__ __forceinline__ void exch_pack_group(const DevProb &P, int g)
{
    const int tid = threadIdx.x, m = P.d;
    GroupState &gs = P.gs[g];
    const int first = gs.first, last = gs.last;
    const int *r = P.r + (size_t)g * (m + 2), *rr = P.rr + (size_t)g * (m + 2), *upd = P.upd + (size_t)g * (m + 2);
    if (tid == 0) {
        double *red = P.red + (size_t)g * 4;
        red[0] = gs.amax; red[1] = gs.pivotmax; red[2] = (gs.pivotmin > 0.0) ? -gs.pivotmin : -999e9;   // :853-859
    }
    {   // to the right neighbour: own last bond q = last, boundary core q+1
        const int q = last, u = upd[q];
        char *base = P.msgR + (size_t)g * P.MSZ;
        int *hh = (int *)base; int *ix = hh + XH; double *dd = (double *)(base + P.IOFF);
        if (tid == 0) { hh[0] = u; hh[1] = r[q]; }
        const int rq = r[q];
        const double *gI = inv_ptr(P, g, q, first);
        for (int x = tid; x < rq * rq; x += blockDim.x) dd[(size_t)P.RM * P.NM + x] = gI[x];           // inv(q) for dtt_lua, :1225
        if (u) {
            const short *Lt = L_ptr(P, g, q, first);
            for (int x = tid; x < q; x += blockDim.x) ix[x] = Lt[(size_t)x * P.RM + (rq - 1)];
            // newest row of core q+1: arg(q+1)(r(q), :, 1:rr(q+1))  (dmrggmp.f90:580)
            const double *A = core_ptr(P, P.arg, g, q + 1, first);
            const int n2 = P.n[q + 1], rrq1 = rr[q + 1];
            for (int x = tid; x < n2 * rrq1; x += blockDim.x) dd[x] = A[(rq - 1) + (size_t)P.RM * (x % n2) + P.SS * (x / n2)];
        }
    }
    {   // to the left neighbour: own first bond q = first, boundary core q
        const int q = first, u = upd[q];
        char *base = P.msgL + (size_t)g * P.MSZ;
        int *hh = (int *)base; int *ix = hh + XH; double *dd = (double *)(base + P.IOFF);
        if (tid == 0) { hh[0] = u; hh[1] = r[q]; }
        if (u) {
            const int rq = r[q];
            const short *Rt = R_ptr(P, g, q, first);
            for (int x = tid; x < m - q; x += blockDim.x) ix[x] = Rt[(size_t)x * P.RM + (rq - 1)];
            // newest column of core q: arg(q)(1:rr(q-1), :, r(q))  (dmrgg.f90:889)
            const double *A = core_ptr(P, P.arg, g, q, first);
            const int n1 = P.n[q], rr0 = rr[q - 1];
            for (int x = tid; x < rr0 * n1; x += blockDim.x) dd[x] = A[(x % rr0) + (size_t)P.RM * (x / rr0) + P.SS * (rq - 1)];
        }
    }
}

// device functions shared by the multi-GPU kernels (one block per group) and the fused single-GPU kernel
__device__ __forceinline__ void exch_pack_group(const DevProb &P, int g);
__device__ __forceinline__ void exch_apply_group(const DevProb &P, int g);

// MAX all-reduce (:861), stage 1: combine the groups of this GPU into redsend[0..2]
__global__ void k_exch_localmax(DevProb P)
{
    if (P.ctl[0]) return;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double a = -1e300, b = -1e300, c = -1e300;
    for (int g = 0; g < P.G; g++) { a = fmax(a, P.red[4 * g]); b = fmax(b, P.red[4 * g + 1]); c = fmax(c, P.red[4 * g + 2]); }
    P.redsend[0] = a; P.redsend[1] = b; P.redsend[2] = c; P.redsend[3] = 0.0;
}
// apply the neighbours' pivots: ranks, index tables, inv of the left boundary bond (:822-850, :1209-1246)
__device__ __forceinline__ void exch_apply_group(const DevProb &P, int g)
{
    const int tid = threadIdx.x, m = P.d;
    GroupState &gs = P.gs[g];
    const int first = gs.first, last = gs.last;
    int *r = P.r + (size_t)g * (m + 2), *upd = P.upd + (size_t)g * (m + 2);
    if (P.inL[g]) {
        const int bl = first - 1;
        const int *hh = (const int *)P.inL[g], *ix = hh + XH; const double *dd = (const double *)(P.inL[g] + P.IOFF);
        const int u = hh[0], rnew = hh[1];
        if (u) { short *Lt = L_ptr(P, g, bl, first); for (int x = tid; x < bl; x += blockDim.x) Lt[(size_t)x * P.RM + (rnew - 1)] = (short)ix[x]; }
        double *gI = inv_ptr(P, g, bl, first);
        for (int x = tid; x < rnew * rnew; x += blockDim.x) gI[x] = dd[(size_t)P.RM * P.NM + x];
        if (tid == 0) { upd[bl] = u; r[bl] = rnew; }
        if (u && P.arith && P.fpersist && tid < 64) {        // fast tables: the neighbour's new pivot as a left multi-index of bond bl, from scratch
            extern __shared__ __align__(16) double dyn_x[];
            double *xs = dyn_x, *ws = xs + m + 8;
            if (P.fun_id == FUN_MVN) {
                for (int k = tid; k < bl; k += 64) xs[k] = P.par[ix[k] - 1] - P.aux[k];
                __builtin_amdgcn_wave_barrier();
                mvn_entry_scratch(P, xs, bl, 0, fast_dv(P, 0, g, bl, first) + (rnew - 1), fast_near(P, 0, g, bl, first) + (rnew - 1), fast_piv(P, 0, g, bl, first) + (rnew - 1), tid);
            } else {
            for (int k = tid; k < bl; k += 64) { xs[k] = P.par[ix[k] - 1]; ws[k] = P.par[P.n[1] + ix[k] - 1]; }
            __builtin_amdgcn_wave_barrier();
            fast_entry_scratch(xs, ws, bl, 0, fast_near(P, 0, g, bl, first) + (rnew - 1), fast_piv(P, 0, g, bl, first) + (rnew - 1), P.RM, tid);
            }
        }
    }
    if (P.inR[g]) {
        const int br = last + 1;
        const int *hh = (const int *)P.inR[g], *ix = hh + XH;
        const int u = hh[0], rnew = hh[1];
        if (u) { short *Rt = R_ptr(P, g, br, first); for (int x = tid; x < m - br; x += blockDim.x) Rt[(size_t)x * P.RM + (rnew - 1)] = (short)ix[x]; }
        if (tid == 0) { upd[br] = u; r[br] = rnew; }
        if (u && P.arith && P.fpersist && tid >= 64 && tid < 128) {   // ... and as a right multi-index of bond br
            extern __shared__ __align__(16) double dyn_x[];
            double *xs = dyn_x + 2 * (m + 8), *ws = xs + m + 8;
            const int lane_ = tid - 64;
            if (P.fun_id == FUN_MVN) {
                for (int k = lane_; k < m - br; k += 64) xs[k] = P.par[ix[k] - 1] - P.aux[br + k];
                __builtin_amdgcn_wave_barrier();
                mvn_entry_scratch(P, xs, m - br, br, fast_dv(P, 1, g, br, first) + (rnew - 1), fast_near(P, 1, g, br, first) + (rnew - 1), fast_piv(P, 1, g, br, first) + (rnew - 1), lane_);
            } else {
            for (int k = lane_; k < m - br; k += 64) { xs[k] = P.par[ix[k] - 1]; ws[k] = P.par[P.n[1] + ix[k] - 1]; }
            __builtin_amdgcn_wave_barrier();
            fast_entry_scratch(xs, ws, m - br, 1, fast_near(P, 1, g, br, first) + (rnew - 1), fast_piv(P, 1, g, br, first) + (rnew - 1), P.RM, lane_);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_exch_pack(DevProb P) { if (P.ctl[0]) return; exch_pack_group(P, blockIdx.x); }
__global__ __launch_bounds__(256) void k_exch_apply(DevProb P) { exch_apply_group(P, blockIdx.x); }

// MAX all-reduce result (:867-870, :961) folded into the apply launch: every block reduces the same inputs and
// writes only its own group.  from_recv: P.redrecv already holds the job-wide maxima (after the RCCL all-reduce)
__global__ __launch_bounds__(256) void k_exch_max_apply(DevProb P, int from_recv, int do_apply)
{
    if (P.ctl[0]) return;
    const int g = blockIdx.x;
    if (threadIdx.x == 0) {
        double a = -1e300, b = -1e300, c = -1e300;
        if (from_recv) { a = P.redrecv[0]; b = P.redrecv[1]; c = P.redrecv[2]; }
        else for (int x = 0; x < P.G; x++) { a = fmax(a, P.red[4 * x]); b = fmax(b, P.red[4 * x + 1]); c = fmax(c, P.red[4 * x + 2]); }
        GroupState &gs = P.gs[g];
        gs.amax = a; gs.pivotmax = b; gs.pivotmin = (-c == 999e9) ? -1.0 : -c;
        gs.pivotmax_prev = b;
    }
    if (do_apply) exch_apply_group(P, g);
}

// one element per wave (ttx_de.h)
template <bool FAST> __device__ __forceinline__ double rows_span(double a, double u0, const double *xs, int L, bool neutral0, int n);
__host__ __device__ inline int de_rows_stride(int m);
__device__ __forceinline__ double de_finish_vals(int id, double a, int m, const double *xv, const double *wv);
// Pair product a = prod ((u_ij-1)/(u_ij+1))^2 of ONE multi-index by one wave, exact, rows ended at the unit cut (nodes in [0,1]); xv = its m
// node values in LDS, sf = 64 x 32 doubles of LDS.  64 rows of the pair triangle per step: lane l walks row base+l left to right (its
// own running product, the reference's; one division per pair above the cut) and writes the factors, compacted by a wave prefix sum
// of the row lengths, to LDS; then the factors of the 64 rows go into `a` in order, sixteen LDS broadcasts ahead of sixteen dependent
// multiplies.  A row that is still above the cut after 32 pairs (nodes close to 1) sends the whole step down a plain serial walk.
// The factors left out are exactly 1: the product has the reference's bits.  (Boundary corners, lottery candidates.)
__device__ __forceinline__ double de_pairs_point_wave_cut(int m, const double *xv, double *sf, int lane)
{
        double a = 1.0;
        for (int base = 0; base < m; base += 64) {
            const int row = base + lane;
            double u = 1.0, fr[32];
            int L = 0; bool on = row < m;
#pragma unroll
            for (int c = 0; c < 32; c++) {
                fr[c] = 1.0;
                if (on && row + c < m) { u = u * xv[row + c]; if (u > 0x1p-54) { fr[c] = de_t2<true>(u); L = c + 1; } else on = false; }
                else on = false;
            }
            const bool lng = (L == 32) && (row + 32 < m) && (u > 0x1p-54);
            if (__builtin_amdgcn_ballot_w64(lng) != 0ull) {                 // rare: serial walk of these rows by every lane
                const int rend = (base + 64 < m) ? base + 64 : m;
                for (int r = base; r < rend; r++) {
                    double uu = 1.0;
                    for (int j = r; j < m; j++) { uu = uu * xv[j]; if (uu <= 0x1p-54) break; a = a * de_t2<true>(uu); }
                }
                continue;
            }
            int inc = L;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o, 64); if (lane >= o) inc += y; }
            const int off = inc - L, total = __shfl(inc, 63, 64);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 32; c++) if (c < L) sf[off + c] = fr[c];
            __builtin_amdgcn_wave_barrier();
            int t = 0;
            for (; t + 16 <= total; t += 16) {                              // two factors per ds_read_b128 (sf is 16-byte aligned)
                double f16[16];
#pragma unroll
                for (int k = 0; k < 8; k++) { const Double2 q2 = *reinterpret_cast<const Double2 *>(sf + t + 2 * k); f16[2 * k] = q2.a; f16[2 * k + 1] = q2.b; }
#pragma unroll
                for (int k = 0; k < 16; k++) a = a * f16[k];
            }
            for (; t < total; t++) a = a * sf[t];
        }
        return a;
}
// corner entry of the Ising D / E integrands by the first wave of the block: multi-index = rowA (dims 1..p-1) | self | rowB.
// Every pair by division with the row-wise evaluator of the lottery (ttx_de.h): the four DPP rows of the wave work on the same
// element (lane n of a row = column n of a 16-column chunk of the pair triangle), the result is identical in all lanes.
__device__ __forceinline__ double de_corner_wave(const DevProb &P, const double *par, const short *rowA, int p, int self, const short *rowB,
                                                 double *scratch, int lane)
{
    const int m = P.d, RSW = de_rows_stride(m), n1m = P.n[1], n = lane & 15;
    double *xv = scratch, *wv = xv + RSW;
    for (int x = lane; x < m; x += 64) {
        const int ix = ((x < p - 1) ? (int)rowA[x] : (x == p - 1) ? self : (int)rowB[x - p]) - 1;
        xv[x] = par[ix]; wv[x] = par[n1m + ix];
    }
    if (lane < 56) xv[m + lane] = 1.0;                      // the evaluator's running products read up to 47 columns past a row's end
    __builtin_amdgcn_wave_barrier();
    if (P.arith) return de_fast_point_wave(P.ising_id, m, xv, wv, lane);      // TTX_ARITH=fast (ttx_fast.h)
    if (P.de_cut) return de_finish_vals(P.ising_id, de_pairs_point_wave_cut(m, xv, (double *)(((size_t)(wv + RSW) + 15) & ~(size_t)15), lane), m, xv, wv);   // (the host sized the scratch for the 64 x 32 factors)
    double a = 1.0;
    if (P.de_unit) for (int i = 0; i < m; i++) a = rows_span<true>(a, 1.0, xv + i, m - i, false, n);
    else for (int i = 0; i < m; i++) a = rows_span<false>(a, 1.0, xv + i, m - i, false, n);
    return de_finish_vals(P.ising_id, a, m, xv, wv);
}

// corner entry of the mvn integrand by the first wave of the block (ttx_mvn.h)
__device__ __forceinline__ double mvn_quadform_wave(int m, const double *dv, int L, double dL, const double *icT, double *tb, int lane);
__device__ __forceinline__ double mvn_corner_wave(const DevProb &P, const double *par, const short *rowA, int p, int self, const short *rowB,
                                                  double *scratch, int lane)
{
    const int m = P.d;
    double *dv = scratch, *tb = scratch + ((m + 1) & ~1);
    for (int x = lane; x < m; x += 64) {
        const int ix = ((x < p - 1) ? (int)rowA[x] : (x == p - 1) ? self : (int)rowB[x - p]) - 1;
        dv[x] = par[ix] - P.aux[x];
    }
    __builtin_amdgcn_wave_barrier();
    if (P.arith) return mvn_fast_point_wave(P, dv, lane);                     // TTX_ARITH=fast (ttx_fast.h)
    const double ex = mvn_quadform_wave(m, dv, -1, 0.0, P.auxT, tb, lane);
    return ttx_exp(-0.5 * ex) / P.mvn_norm;
}

// grow the boundary cores with the neighbours' fibers, evaluate the corner entries, LU-apply
// blocks [0,NM): "share blocks to the LEFT" receive side (:912-952), one mode index k each;
// blocks [NM,2NM): "share blocks to the RIGHT" receive side (dmrggmp.f90:598-627), one mode index j each
template <int FUN>
__global__ __launch_bounds__(TTX_BLK) void k_exch_boundary(DevProb P)
{
    if (P.ctl[0]) return;
    extern __shared__ __align__(16) double dyn[];
    __shared__ double s_bc;
    const int g = blockIdx.y, tid = threadIdx.x, m = P.d;
    GroupState &gs = P.gs[g];
    const int first = gs.first, last = gs.last;
    const int *r = P.r + (size_t)g * (m + 2), *rr = P.rr + (size_t)g * (m + 2), *upd = P.upd + (size_t)g * (m + 2);
    double *par = dyn;
    // corner evaluation: the multi-index as two 16-byte aligned, padded rows (dims 1..p-1 | dims p+1..m) around dim p,
    // the layout the fast chunked evaluators read
    const int VSr = ((m + 7) & ~7) + 8;
    short *rowA = (short *)(((size_t)(dyn + P.npar) + 15) & ~(size_t)15), *rowB = rowA + VSr;
    double *lu = (double *)(rowB + VSr);                  // packed LU of the boundary bond (ranks <= 64)
    double *wscr = lu + 64 * 64 + 4;                      // scratch of the one-element-per-wave evaluator (Ising D/E, P.bnd_wave)
    __shared__ double s_corner;
    const bool dewave = (FUN == FUN_ISING) && P.ising_id != 1 && P.bnd_wave;
    const bool mvwave = (FUN == FUN_MVN) && P.bnd_wave;
    if ((int)blockIdx.x < P.NM) {
        const int k = blockIdx.x, p = last, br = last + 1;
        if (!P.inR[g] || !upd[br] || k >= P.n[br]) return;
        const int rp = r[p], rrp = rr[p], snew = r[br] - 1, n2 = P.n[br];
        const double *msg = (const double *)(P.inR[g] + P.IOFF);      // (rr(p), n(p+1))
        double *A = core_ptr(P, P.arg, g, br, first), *W = core_ptr(P, P.row, g, br, first);
        double a = 0.0;
        if (tid < rrp) a = msg[tid + (size_t)rrp * k];
        if (upd[p]) {                                                // corner arg(p+1)(r(p), k, r(p+1)), :925-937
            for (int x = tid; x < P.npar; x += TTX_BLK) par[x] = P.par[x];
            const int *vp = vip_ptr(P, g, p, first) + 4 * (rp - 1);
            const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, br, first);
            auto dimv = [&](int s) -> short { return (s < p) ? Lt[(size_t)(s - 1) * P.RM + (vp[0] - 1)] : (s == p) ? (short)vp[1] : (s == p + 1) ? (short)(k + 1) : Rt[(size_t)(s - p - 2) * P.RM + snew]; };
            for (int x = tid; x < VSr; x += TTX_BLK) { rowA[x] = (x < p - 1) ? dimv(x + 1) : (short)1; rowB[x] = (x < m - p) ? dimv(p + 1 + x) : (short)1; }
            __syncthreads();
            if (dewave || mvwave) {
                if (tid < 64) {
                    const double c_ = dewave ? de_corner_wave(P, par, rowA, p, (int)vp[1], rowB, wscr, tid) : mvn_corner_wave(P, par, rowA, p, (int)vp[1], rowB, wscr, tid);
                    if (tid == 0) s_corner = c_;
                }
                __syncthreads();
            }
            if (tid == rrp) {
                Src3 sx{rowA, p - 1, (int)vp[1], rowB};
                a = (dewave || mvwave) ? s_corner : eval_src3<FUN, true>(P, par, sx, (long)g * P.HS + k);
                if (!HOST_PASS1(FUN, P)) {
                    atomic_max_pos(&gs.amax, fabs(a));
                    if (k == 0) atomicAdd((unsigned long long *)&gs.neval, (unsigned long long)n2);   // :936
                }
            }
        }
        if (HOST_PASS1(FUN, P)) return;
        if (tid < rp) A[tid + (size_t)P.RM * k + P.SS * snew] = a;
        // row(p+1)(:, k, new) = L(p)^-1 * column : d2_luar(n(p+1), r(p), inv(p), .) :940-951
        const double *gI = inv_ptr(P, g, p, first);
        double tmp = 0.0, xf = 0.0;
        if (rp <= 64) {                               // one wave: the solve runs as a shuffle wavefront over LDS-staged LU
            __syncthreads();
            for (int x = tid; x < rp * rp; x += TTX_BLK) lu[x] = gI[x];
            __syncthreads();
            if (tid < 64)
                for (int s = 0; s < rp; s++) {
                    const double cand = (s == 0) ? a : a + (-1.0) * tmp;
                    const double xsv = __shfl(cand, s, 64);
                    if (tid == s) xf = xsv;
                    if (tid > s && tid < rp) tmp = tmp + xsv * lu[tid * tid + s];
                }
        } else
        for (int s = 0; s < rp; s++) {
            if (tid == s) { xf = (s == 0) ? a : a + (-1.0) * tmp; s_bc = xf; }
            __syncthreads();
            if (tid > s && tid < rp) tmp = tmp + s_bc * gI[tid * tid + s];
            __syncthreads();
        }
        if (tid < rp) W[k + (size_t)P.NM * snew + P.SW * tid] = xf;
    } else {
        const int j = blockIdx.x - P.NM, p = first, bl = first - 1;
        if (!P.inL[g] || !upd[bl] || j >= P.n[p]) return;
        const int rp = r[p], rrp = rr[p], inew = r[bl] - 1, n1 = P.n[p];
        const double *msg = (const double *)(P.inL[g] + P.IOFF);      // (n(p), rr(p))
        double *A = core_ptr(P, P.arg, g, p, first), *C = core_ptr(P, P.col, g, p, first);
        double y = 0.0;
        if (tid < rrp) y = msg[j + (size_t)n1 * tid];
        if (upd[p]) {                                                // corner arg(p)(r(p-1), j, r(p)), dmrggmp.f90:608-616
            for (int x = tid; x < P.npar; x += TTX_BLK) par[x] = P.par[x];
            const int *vp = vip_ptr(P, g, p, first) + 4 * (rp - 1);
            const short *Lt = L_ptr(P, g, bl, first), *Rt = R_ptr(P, g, p + 1, first);
            auto dimv = [&](int s) -> short { return (s < p) ? Lt[(size_t)(s - 1) * P.RM + inew] : (s == p) ? (short)(j + 1) : (s == p + 1) ? (short)vp[2] : Rt[(size_t)(s - p - 2) * P.RM + (vp[3] - 1)]; };
            for (int x = tid; x < VSr; x += TTX_BLK) { rowA[x] = (x < p - 1) ? dimv(x + 1) : (short)1; rowB[x] = (x < m - p) ? dimv(p + 1 + x) : (short)1; }
            __syncthreads();
            if (dewave || mvwave) {
                if (tid < 64) {
                    const double c_ = dewave ? de_corner_wave(P, par, rowA, p, j + 1, rowB, wscr, tid) : mvn_corner_wave(P, par, rowA, p, j + 1, rowB, wscr, tid);
                    if (tid == 0) s_corner = c_;
                }
                __syncthreads();
            }
            if (tid == rrp) {
                Src3 sx{rowA, p - 1, j + 1, rowB};
                y = (dewave || mvwave) ? s_corner : eval_src3<FUN, true>(P, par, sx, (long)g * P.HS + P.NM + j);
                if (!HOST_PASS1(FUN, P)) {
                    atomic_max_pos(&gs.amax, fabs(y));
                    if (j == 0) atomicAdd((unsigned long long *)&gs.neval, (unsigned long long)n1);
                }
            }
        }
        if (HOST_PASS1(FUN, P)) return;
        if (tid < rp) A[inew + (size_t)P.RM * j + P.SS * tid] = y;
        // col(p)(new, j, :) = row * U(p)^-1 : d2_lual(n(p), r(p), inv(p), .) dmrggmp.f90:622
        const double *gI = inv_ptr(P, g, p, first);
        if (rp <= 64) {
            __syncthreads();
            for (int x = tid; x < rp * rp; x += TTX_BLK) lu[x] = gI[x];
            __syncthreads();
            if (tid < 64) {
                const double rdg = (tid < rp) ? 1.0 / lu[(tid + 1) * (tid + 1) - 1] : 0.0;
                for (int s = 0; s < rp; s++) {
                    const double cand = rdg * y;          // only lane s's product is used
                    const double ys = __shfl(cand, s, 64);
                    if (tid == s) y = ys;
                    if (tid > s && tid < rp) y = y + (-lu[tid * tid + tid + s]) * ys;
                }
            }
        } else
        for (int s = 0; s < rp; s++) {
            if (tid == s) { y = (1.0 / gI[(s + 1) * (s + 1) - 1]) * y; s_bc = y; }
            __syncthreads();
            if (tid > s && tid < rp) y = y + (-gI[tid * tid + tid + s]) * s_bc;
            __syncthreads();
        }
        if (tid < rp) C[inew + (size_t)P.RM * j + P.SS * tid] = y;
    }
}

// binary-tree product of the partial quadrature matrices of ALL groups (lib/dmrgg.f90:1355-1405); single
// block, run redundantly by every GPU on the all-reduced P.qall so that every rank holds the same value
__global__ __launch_bounds__(256) void k_quad_tree(DevProb P)
{
    if (P.ctl[2]) return;
    const int tid = threadIdx.x, RM = P.RM, ng = P.nprocs;
    __shared__ int mym[256], myn[256];
    const double *part = P.qall, *dims = P.qall + (size_t)ng * RM * RM;
    double *work = P.qwork;
    for (int g = tid; g < ng; g += blockDim.x) { mym[g] = (int)dims[2 * g]; myn[g] = (int)dims[2 * g + 1]; }
    __syncthreads();
    for (int g = 0; g < ng; g++)
        for (int x = tid; x < mym[g] * myn[g]; x += blockDim.x) work[(size_t)g * RM * RM + (x % mym[g]) + RM * (x / mym[g])] = part[(size_t)g * RM * RM + (x % mym[g]) + RM * (x / mym[g])];
    __syncthreads();
    double *tmp = work + (size_t)ng * RM * RM;
    for (int q = 1; q < ng; q *= 2) {
        for (int me = 0; me + q < ng; me += 2 * q) {
            const int her = me + q;
            const double *Aa = work + (size_t)me * RM * RM, *Bb = work + (size_t)her * RM * RM;
            const int mm = mym[me], kk = myn[me], nn = myn[her];
            for (int x = tid; x < mm * nn; x += blockDim.x) {
                int i = x % mm, j = x / mm;
                double c = 0.0;
                for (int l = 0; l < kk; l++) c = c + Bb[l + RM * j] * Aa[i + RM * l];     // dgemm 'n','n', :1381
                tmp[i + RM * j] = c;
            }
            __syncthreads();
            double *dst = work + (size_t)me * RM * RM;
            for (int x = tid; x < mm * nn; x += blockDim.x) dst[(x % mm) + RM * (x / mm)] = tmp[(x % mm) + RM * (x / mm)];
            __syncthreads();
            if (tid == 0) myn[me] = nn;
            __syncthreads();
        }
    }
    if (tid == 0) for (int g = 0; g < P.G; g++) P.gs[g].val = work[0];
}

// per-sweep summary of this GPU in the job-wide layout (slots of other GPUs stay zero; SUM all-reduce)
__device__ __forceinline__ void collect_summary(const DevProb &P);
__global__ __launch_bounds__(256) void k_collect(DevProb P)
{
    if (blockIdx.x != 0 || P.ctl[0]) return;
    collect_summary(P);
}
// single-GPU end of sweep in one launch: snapshot for the forked quadrature, summary (written straight into the
// pinned host slot `out`), stopping rule
__global__ __launch_bounds__(256) void k_sweep_end(DevProb P, int it, double *out)
{
    const int n = P.G * (P.d + 2), m = P.d, tid = threadIdx.x;
    for (int x = tid; x < n; x += blockDim.x) P.rq[x] = P.r[x];
    if (tid == 0) P.ctl[2] = P.ctl[0];
    if (P.ctl[0]) return;
    // multi-GPU job: `out` is this GPU's contribution to a SUM all-reduce -- only the bonds of its own groups, and the
    // job-wide scalars from the GPU that holds global group 0 (after k_exch_max_apply every GPU has the same maxima)
    const int p_lo = P.gs[0].first, p_hi = P.gs[P.G - 1].last;
    for (int p = p_lo + tid; p <= p_hi; p += blockDim.x) {   // owner of bond p, its rank and tape entry
        int g = 0;
        while (g + 1 < P.G && P.gs[g + 1].first <= p) g++;
        const int *r = P.r + (size_t)g * (m + 2), *tp = P.tape + (size_t)g * (m + 2) * 4;
        double *e = out + SUM_HDR + P.nprocs + 5 * p;
        e[0] = (double)r[p]; e[1] = (double)tp[4 * p]; e[2] = (double)tp[4 * p + 1]; e[3] = (double)tp[4 * p + 2]; e[4] = (double)tp[4 * p + 3];
    }
    if (tid < P.G) out[SUM_HDR + P.gs[tid].gglobal] = P.gs[tid].initval;
    if (tid != 0) return;
    double nev = 0.0, by = 0.0, nr = 0.0;
    for (int g = 0; g < P.G; g++) { const GroupState &gs = P.gs[g]; nev += (double)gs.neval; by += gs.bytes_half; nr += (double)gs.n_resid; }
    const double amax = P.gs[0].amax, pmax = P.gs[0].pivotmax;    // job-wide maxima (k_exch_max_apply), the same on every GPU
    out[SUM_NEVAL] = nev; out[SUM_BYTES] = by; out[SUM_NRESID] = nr;
    if (P.g0 == 0) { out[SUM_AMAX] = amax; out[SUM_PMAX] = pmax; out[SUM_PMIN] = P.gs[0].pivotmin; }
    out[SUM_VAL] = 0.0;
    int ready = (it + 1 >= P.maxrank);
    if (P.accuracy >= 0.0) {
        if (pmax <= P.accuracy * amax) P.ctl[1]++; else P.ctl[1] = 0;
        ready = ready || (P.ctl[1] >= 3);
    }
    P.ctl[0] = ready;
}
__device__ __forceinline__ void collect_summary(const DevProb &P)
{
    const int m = P.d, tid = threadIdx.x;
    double *o = P.sumsend;
    for (int g = 0; g < P.G; g++) {                                   // one thread per own bond of the group
        const GroupState &gs = P.gs[g];
        const int *r = P.r + (size_t)g * (m + 2), *tp = P.tape + (size_t)g * (m + 2) * 4;
        for (int p = gs.first + tid; p <= gs.last; p += blockDim.x) {
            double *e = o + SUM_HDR + P.nprocs + 5 * p;
            e[0] = (double)r[p]; e[1] = (double)tp[4 * p]; e[2] = (double)tp[4 * p + 1]; e[3] = (double)tp[4 * p + 2]; e[4] = (double)tp[4 * p + 3];
        }
    }
    if (tid != 0) return;
    double nev = 0.0, by = 0.0, nr = 0.0;
    for (int g = 0; g < P.G; g++) {
        const GroupState &gs = P.gs[g];
        nev += (double)gs.neval; by += gs.bytes_half; nr += (double)gs.n_resid;
        o[SUM_HDR + gs.gglobal] = gs.initval;
        if (gs.gglobal == 0) { o[SUM_AMAX] = gs.amax; o[SUM_PMAX] = gs.pivotmax; o[SUM_PMIN] = gs.pivotmin; }
    }
    o[SUM_NEVAL] = nev; o[SUM_BYTES] = by; o[SUM_NRESID] = nr;
    if (P.g0 == 0) o[SUM_VAL] = P.gs[0].val;     // identical on every GPU; contributed once
}

__global__ void k_fill_synth(double *p, size_t n, unsigned long long seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}
// grid-stride variant of k_resid_argmax for factors far larger than one wave of blocks (same arithmetic per row)
__global__ __launch_bounds__(TTX_BLK) void k_resid_argmax_stream(long long m, int r, size_t ld, const double *a, const double *F, const double *x,
                                                                 double *b_out, Partial *parts)
{
    __shared__ double xs[256];
    __shared__ double sha[4], shv[4]; __shared__ int shi[4];
    for (int s = threadIdx.x; s < r; s += TTX_BLK) xs[s] = x[s];
    __syncthreads();
    double ab = -1.0, bv = 0.0; int bi = INT_MAX;
    for (long long t = (long long)blockIdx.x * TTX_BLK + threadIdx.x; t < m; t += (long long)gridDim.x * TTX_BLK) {
        double b = a[t];
#pragma unroll 8
        for (int s = 0; s < r; s++) b = b + (-xs[s]) * F[t + ld * s];
        b_out[t] = b;
        double aa = fabs(b);
        if (aa > ab || (aa == ab && (int)t < bi)) { ab = aa; bv = b; bi = (int)t; }
    }
    block_argmax(ab, bv, bi, sha, shv, shi);
    if (threadIdx.x == 0) { Partial pr; pr.absmax = ab; pr.val = bv; pr.idx = bi; pr.pad = 0; parts[blockIdx.x] = pr; }
}

// ------------------------------------------------------------------------------------------------
// dtt_accchk (lib/dmrgg.f90:1081-1166): one wave per random sample.  out[4*il..] = (|aval-bval|, (aval-bval)^2,
// aval, aval^2); ind_out[il*d..] = the sample's multi-index
// ------------------------------------------------------------------------------------------------
template <int FUN>
__global__ __launch_bounds__(64) void k_accchk(DevProb P, unsigned long long rngpos, int nlot, const int *owner, double *out, int *ind_out, int il0)
{
    extern __shared__ __align__(16) double dyn[];
    const int il = il0 + blockIdx.x, tid = threadIdx.x, m = P.d;
    double *par = dyn, *x = dyn + ((P.npar + 1) & ~1), *z = x + P.RM;
    int *ind = (int *)(z + P.RM);
    for (int s = tid; s < P.npar; s += 64) par[s] = P.par[s];
    for (int i = tid; i < m; i += 64) {                      // irnd: int(d*maxi)+1, draws in sample-major order (:1119-1122)
        double d = ttx_flang_draw(rngpos + (unsigned long long)il * m + i);
        ind[i] = (int)(d * P.n[i + 1]) + 1;
        ind_out[(size_t)il * m + i] = ind[i];
    }
    __syncthreads();
    double aval = 0.0;
    if (tid == 0) { ListIdx ix{ind}; aval = eval_fun<FUN>(P, par, ix, (long)blockIdx.x); }
    if (HOST_PASS1(FUN, P)) return;
    // dtt_ijk (lib/tt.f90:630-652): x = U_m(:, ind_m, 1); for i = m-1..1: x = U_i(:, ind_i, :) x
    {
        const int g = owner[m], first = P.gs[g].first;
        const int *r = P.r + (size_t)g * (m + 2);
        const double *A = core_ptr(P, P.arg, g, m, first);
        if (tid < r[m - 1]) x[tid] = A[tid + (size_t)P.RM * (ind[m - 1] - 1)];
    }
    __syncthreads();
    for (int i = m - 1; i >= 1; i--) {
        const int g = owner[i], first = P.gs[g].first;
        const int *r = P.r + (size_t)g * (m + 2);
        const int q0 = r[i - 1], q1 = r[i];
        const double *A = core_ptr(P, P.arg, g, i, first) + (size_t)P.RM * (ind[i - 1] - 1);
        for (int t = tid; t < q0; t += 64) {
            double s = 0.0;
            for (int k = 0; k < q1; k++) s = s + A[t + P.SS * k] * x[k];
            z[t] = s;
        }
        __syncthreads();
        for (int t = tid; t < q0; t += 64) x[t] = z[t];
        __syncthreads();
    }
    if (tid == 0) {
        const double bval = x[0], e = aval - bval;
        out[4 * (size_t)il] = fabs(e); out[4 * (size_t)il + 1] = e * e; out[4 * (size_t)il + 2] = aval; out[4 * (size_t)il + 3] = aval * aval;
    }
}
