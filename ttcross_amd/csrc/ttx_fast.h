// ttx_fast.h -- TTX_ARITH=fast: the heavy integrands evaluated in O(d) per fiber element instead of O(d^2).
// Included by ttx_kernels.h (after its address helpers and wave reductions); no other includes.
//
// The exact mode (default, the checker) performs every product and sum of the reference's integrand in the reference's
// order: test_crs_ising.f90:186-195 is one dependent chain of d(d+1)/2 divisions and multiplies per evaluation (32 640 at
// D_256), lib/mvn_pdf.f90:74-80 one chain of d^2 additions.  In fast mode only the VALUE of the integrand is kept (to
// rounding); products and sums are re-associated so that everything a fiber's elements share is computed once.
//
// Ising D / E.  a = prod over all contiguous dimension ranges [s,e] of g(u)^2, g(u) = (1-u)/(1+u), u = x_s ... x_e.
//  * rho = prod g = prod(1-u) / prod(1+u): numerator and denominator are accumulated separately (no division per pair) and
//    divided once; a = rho^2.  With nodes in [0,1] both products are monotone, prod(1-u) <= rho and prod(1+u) >= 1, so an
//    under/overflow of either implies that a itself underflows (ln rho <= -2 ln prod(1+u)): no range is lost.
//  * u <= 2^-54 gives 1-u = 1+u = 1 EXACTLY in fp64 (also in the reference's arithmetic), and u is non-increasing when a
//    range is extended (nodes <= 1): every scan over ranges stops at the first such u.  At the Gauss-Legendre nodes of the
//    drivers a range of more than ~16 dimensions is below the cut, so ~12 % of the pairs of D_256 are left.
//  * A fiber element is (pivot v of the varying side | free node x | fixed side).  Ranges inside the pivot's dims: one number
//    T_v per pivot (k_fast_tables).  Ranges inside the fixed side: one number per fiber.  Ranges through the free dimension:
//    u = near_v[a] * x * far[b] with the DECAY VECTORS near_v[a] = product of the a dims of the pivot nearest to the bond and
//    far[b] likewise on the fixed side -- a small rectangle (~18 x 18, triangular after the cut) per element.
//  * b = 1/(v w) (:197-205) and the weights are affine in the free node given four sums / products per pivot.
// mvn.  Q = d' S d with d = (dL_i | dj | dk | dR_q):  Q = QL_i + QR_q + 2 dL_i' S_LR dR_q + (terms linear and quadratic in
// dj, dk with coefficients (S dL_i)_p, (S dR_q)_p, ...).  Per pivot: Y = S d (k_fast_tables); per element O(1) plus one
// dot product of length p per (i, q) pair.  S is symmetrised on the host ((S + S')/2: the same quadratic form).
//
// Parity of this mode is by tolerance (tests/test_gpu_fast.py): same sweeps, leading sweeps with identical pivots, values to
// 2e-13, integrals to 1e-12 of the exact mode / the reference.
#pragma once
#define TTX_FCUT 5.551115123125783e-17      // 2^-54
// rows of fPiv
#define FP_T 0      // Ising: rho of the ranges inside the pivot's dims        mvn: Q = d' S d of the pivot's dims
#define FP_W 1      // Ising: product of the weights of the pivot's dims
#define FP_S 2      // Ising: sum_{t>=1} near[t]
#define FP_P 3      // Ising: left: sum of prefix products from dim 1; right: sum of suffix products from dim d
#define FP_F 4      // Ising: product of all nodes of the pivot
#define FP_N 5      // Ising: number of leading entries of near (t = 0 included) above the cut

__device__ __forceinline__ double wave_prod(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v * __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o, 64);
    return v;
}

// rho of one arbitrary multi-index by ONE thread: for every end e the ranges [s,e], s = e, e-1, ... until the cut.
// nodes: 1-based (par - 1); idx(s) = mode index of dim s.  O(d * L) with L ~ 16 instead of d(d+1)/2 divisions.
template <class IDX>
__device__ __forceinline__ double de_fast_rho(int m, const double *nodes, IDX idx)
{
    double N = 1.0, D = 1.0;
    for (int e = 1; e <= m; e++) {
        double u = 1.0;
        for (int s = e; s >= 1; s--) {
            u = u * nodes[idx(s)];
            if (u <= TTX_FCUT) break;
            N = N * (1.0 - u); D = D * (1.0 + u);
        }
    }
    return N / D;
}

// b = 1/(v w) of test_crs_ising.f90:197-205 for the element (left pivot | x_j | x_k | right pivot) from the pivots' sums:
// PL = sum of prefix products of the left pivot, FL = its full product, SL = sum_{t>=1} of its decay vector; SR, VR
// (suffix-product sum from dim d), FR likewise on the right.
__device__ __forceinline__ double de_fast_b(double PL, double FL, double SL, double xj, double xk, double SR, double VR, double FR)
{
    const double w = 1.0 + PL + FL * xj * (1.0 + xk * (1.0 + SR));
    const double v = 1.0 + VR + FR * xk * (1.0 + xj * (1.0 + SL));
    return 1.0 / (v * w);
}

// ranges through the free dimension: prod over a < na, b < nb of (1 -/+ near[a*ns] * x * far[b]), cut as soon as the
// argument is at or below 2^-54 (both vectors are non-increasing)
__device__ __forceinline__ void de_fast_span(const double *near, size_t ns, int na, double x, const double *far, int nb, double &N, double &D)
{
    for (int a = 0; a < na; a++) {
        const double A = near[a * ns] * x;
        if (A <= TTX_FCUT) break;
        for (int b = 0; b < nb; b++) {
            const double c = far[b];
            if (A * c <= TTX_FCUT) break;
            N = N * __builtin_fma(-A, c, 1.0);
            D = D * __builtin_fma(A, c, 1.0);
        }
    }
}

// The same with the far vector in REGISTERS: three blocks of eight (entries past the vector's end are 0).  A factor whose
// argument is at or below the cut is EXACTLY 1 (fma(-/+A, c, 1) rounds to 1), so inside a block nothing is tested; a block is
// skipped when its first argument is below the cut (the vector is non-increasing).  Two accumulator pairs per lane; the next
// near entry is requested one outer iteration ahead.  With one wave per SIMD a per-pair LDS read is not hidden by anything.
#define TTX_FNR 24
struct FarReg { double c0[8], c1[8], c2[8]; };
__device__ __forceinline__ void far_load(FarReg &C, const double *far, int nb)
{
#pragma unroll
    for (int b = 0; b < 8; b++) { C.c0[b] = (b < nb) ? far[b] : 0.0; C.c1[b] = (8 + b < nb) ? far[8 + b] : 0.0; C.c2[b] = (16 + b < nb) ? far[16 + b] : 0.0; }
}
__device__ __forceinline__ void far_block(double A, const double (&c)[8], double &N0, double &D0, double &N1, double &D1)
{
#pragma unroll
    for (int b = 0; b < 8; b += 2) {
        N0 = N0 * __builtin_fma(-A, c[b], 1.0);     D0 = D0 * __builtin_fma(A, c[b], 1.0);
        N1 = N1 * __builtin_fma(-A, c[b + 1], 1.0); D1 = D1 * __builtin_fma(A, c[b + 1], 1.0);
    }
}
__device__ __forceinline__ void de_fast_span_reg(const double *near, size_t ns, int na, double x, const FarReg &C, double &N, double &D)
{
    double nx = near[0], N1 = 1.0, D1 = 1.0;
    for (int a = 0; a < na; a++) {
        const double A = nx * x;
        nx = near[(size_t)(a + 1 < na ? a + 1 : a) * ns];
        if (A * C.c0[0] <= TTX_FCUT) break;
        far_block(A, C.c0, N, D, N1, D1);
        if (A * C.c1[0] > TTX_FCUT) {
            far_block(A, C.c1, N, D, N1, D1);
            if (A * C.c2[0] > TTX_FCUT) far_block(A, C.c2, N, D, N1, D1);
        }
    }
    N = N * N1; D = D * D1;
}

// value of the Ising D/E integrand at ONE multi-index by one wave (boundary corners): xv / wv[0..m) = node and weight values
// of the point in LDS.  Lane = end of the range; every scan stops at the cut.
__device__ __forceinline__ double de_fast_point_wave(int id, int m, const double *xv, const double *wv, int lane)
{
    double N = 1.0, D = 1.0, W = 1.0;
    for (int e = lane; e < m; e += 64) {
        double u = 1.0;
        for (int s = e; s >= 0; s--) {
            u = u * xv[s];
            if (u <= TTX_FCUT) break;
            N = N * (1.0 - u); D = D * (1.0 + u);
        }
        W = W * wv[e];
    }
    N = wave_prod(N); D = wave_prod(D); W = wave_prod(W);
    const double rho = N / D;
    double b = 1.0;
    if (id == 2) {
        double v = 1.0, w = 1.0, vk = 1.0, wk = 1.0;
        for (int i = m - 1; i >= 0; i--) { vk = vk * xv[i]; if (vk <= TTX_FCUT) break; v = v + vk; }
        for (int i = 0; i < m; i++) { wk = wk * xv[i]; if (wk <= TTX_FCUT) break; w = w + wk; }
        b = 1.0 / (v * w);
    }
    return 2 * b * (rho * W) * rho;
}

__device__ __forceinline__ bool fast_path(const DevProb &P, int FUN)
{ return P.arith && ((FUN == FUN_ISING && P.ising_id != 1) || FUN == FUN_MVN); }

// base of the table rows a bond step at bond p works with: side 0 = left pivots of bond p-1, side 1 = right pivots of bond p+1
// (slots as L_ptr / R_ptr: left tables of bonds first-1..last, right tables of bonds first..last+1)
__device__ __forceinline__ size_t fast_slot(const DevProb &P, int side, int g, int bond, int first)
{ return P.fpersist ? (size_t)g * P.NC + (size_t)(side == 0 ? bond - first + 1 : bond - first) : (size_t)g; }
__device__ __forceinline__ double *fast_near(const DevProb &P, int side, int g, int bond, int first)
{ return P.fNear[side] + fast_slot(P, side, g, bond, first) * (size_t)P.FD * P.RM; }
__device__ __forceinline__ double *fast_piv(const DevProb &P, int side, int g, int bond, int first)
{ return P.fPiv[side] + fast_slot(P, side, g, bond, first) * (size_t)TTX_FS * P.RM; }
__device__ __forceinline__ double *fast_dv(const DevProb &P, int side, int g, int bond, int first)
{ return P.fDv[side] + fast_slot(P, side, g, bond, first) * (size_t)P.FD * P.RM; }

// Table entry of ONE pivot FROM SCRATCH by one wave (all 64 lanes call it).  xs / ws: node and weight values of the pivot's own
// dims in natural order (len of them, LDS, visible to the wave); side 0: the bond lies behind the last dim, side 1: before the
// first.  near / piv: the pivot's column of the tables (stride RM).
__device__ __forceinline__ void fast_entry_scratch(const double *xs, const double *ws, int len, int side, double *near, double *piv, int RM, int lane)
{
    // uniform scans (every lane the same): the decay vector towards the bond with its sum, and the sum of the products that start
    // at the chain's end.  Both stop at the cut: the products are non-increasing, a later term is below 2^-54 of a sum that is
    // only ever added to 1.  The two full products (all nodes, all weights) are taken by the wave.
    double dc = 1.0, S1 = 0.0, pq = 1.0, S2 = 0.0;
    int cnt = 1;
    if (lane == 0) near[0] = 1.0;
    for (int t = 1; t <= len; t++) {
        dc = dc * (side == 0 ? xs[len - t] : xs[t - 1]);          // the t dims nearest to the bond
        if (dc <= TTX_FCUT) break;
        S1 = S1 + dc;
        cnt = t + 1;
        if (lane == (t & 63)) near[(size_t)t * RM] = dc;
    }
    for (int t = 1; t <= len; t++) {
        pq = pq * (side == 0 ? xs[t - 1] : xs[len - t]);          // left: prefix products from dim 1; right: suffix products from dim d
        if (pq <= TTX_FCUT) break;
        S2 = S2 + pq;
    }
    double W = 1.0, F = 1.0;
    for (int k = lane; k < len; k += 64) { W = W * ws[k]; F = F * xs[k]; }
    W = wave_prod(W); F = wave_prod(F);
    // ranges inside the pivot's dims: lane = end position, walk the start towards smaller positions until the cut
    double N = 1.0, D = 1.0;
    for (int e = lane; e < len; e += 64) {
        double u = 1.0;
        for (int s = e; s >= 0; s--) {
            u = u * xs[s];
            if (u <= TTX_FCUT) break;
            N = N * (1.0 - u); D = D * (1.0 + u);
        }
    }
    N = wave_prod(N); D = wave_prod(D);
    if (lane == 0) {
        piv[FP_T * RM] = N / D; piv[FP_W * RM] = W; piv[FP_S * RM] = S1; piv[FP_P * RM] = S2;
        piv[FP_F * RM] = F; piv[FP_N * RM] = (double)cnt;
    }
}
// Table entry of a NEW pivot from its PARENT's by one wave: the new pivot is the parent's multi-index with one more dimension (node
// x, weight w) on the bond's side.  Decay vector: near_c[t] = x near_p[t-1]; the new ranges are the ones that contain the new dim and
// end at the bond (arguments near_c[t], t >= 1); sums and products are extended by one term.  O(cut length) instead of O(d L).
__device__ __forceinline__ void fast_entry_child(const double *pnear, const double *ppiv, double x, double w, double *cnear, double *cpiv, int RM, int lane)
{
    const int cp = (int)ppiv[FP_N * RM];
    double N = 1.0, D = 1.0, S1 = 0.0;
    int cnt = 1;
    if (lane == 0) cnear[0] = 1.0;
    for (int t0 = 1; t0 <= cp; t0 += 64) {                        // child entries t = 1 .. cp come from parent entries t-1 = 0 .. cp-1
        const int t = t0 + lane;
        const double v = (t <= cp) ? x * pnear[(size_t)(t - 1) * RM] : 0.0;
        const bool on = v > TTX_FCUT;
        if (on) { cnear[(size_t)t * RM] = v; N = N * (1.0 - v); D = D * (1.0 + v); S1 = S1 + v; }
        cnt += __popcll(__ballot(on));
    }
    N = wave_prod(N); D = wave_prod(D); S1 = wave_sum(S1);
    if (lane == 0) {
        const double F = ppiv[FP_F * RM] * x;
        cpiv[FP_T * RM] = ppiv[FP_T * RM] * (N / D); cpiv[FP_W * RM] = ppiv[FP_W * RM] * w; cpiv[FP_S * RM] = S1;
        cpiv[FP_P * RM] = ppiv[FP_P * RM] + (F > TTX_FCUT ? F : 0.0);
        cpiv[FP_F * RM] = F; cpiv[FP_N * RM] = (double)cnt;
    }
}

// value of the Ising D/E integrand from rho of the ranges through the free dimension(s) and the two pivots' table entries
// (pL / pR point at the pivot's column of fPiv, stride RM): rho -> 2 rho^2 [b] w_1 ... w_m
__device__ __forceinline__ double de_fast_value(int id, int RM, double rho, const double *pL, double xj, double wj, double xk, double wk, const double *pR)
{
    const double t = rho * ((pL[FP_W * RM] * wj) * (wk * pR[FP_W * RM]));
    if (id != 2) return 2 * t * rho;
    const double b = de_fast_b(pL[FP_P * RM], pL[FP_F * RM], pL[FP_S * RM], xj, xk, pR[FP_S * RM], pR[FP_P * RM], pR[FP_F * RM]);
    return 2 * b * t * rho;
}

// one lottery candidate (left pivot i | j | k | right pivot q), 0-based, by one thread.  sL / sR: LDS copies of the first `cap`
// rows of the two near tables ([t][RM]); rows beyond come from global memory.
__device__ __forceinline__ double de_fast_elem4(const DevProb &P, int g, int p, int first, int i, int j, int k, int q, const double *sL, const double *sR, int cap)
{
    const int RM = P.RM;
    const double *nodes = P.par, *weights = P.par + P.n[1];
    const double *nL = fast_near(P, 0, g, p - 1, first) + i, *nR = fast_near(P, 1, g, p + 1, first) + q;
    const double *pL = fast_piv(P, 0, g, p - 1, first) + i, *pR = fast_piv(P, 1, g, p + 1, first) + q;
    const int cl = (int)pL[FP_N * RM], cr = (int)pR[FP_N * RM];
    const double xj = nodes[j], xk = nodes[k];
    double N = 1.0, D = 1.0;
    if (cr + 1 <= TTX_FNR && cr <= cap && cl <= cap) {
        // far vector of the candidate in registers: C[0] = 1 (ranges that end at dim p), C[1+t] = x_k * dr[t]
        FarReg C;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            C.c0[b] = (b == 0) ? 1.0 : (b - 1 < cr) ? xk * sR[(size_t)(b - 1) * RM + q] : 0.0;
            C.c1[b] = (7 + b < cr) ? xk * sR[(size_t)(7 + b) * RM + q] : 0.0;
            C.c2[b] = (15 + b < cr) ? xk * sR[(size_t)(15 + b) * RM + q] : 0.0;
        }
        // ranges that start at dim p+1: arguments C[1..]; an argument at or below the cut gives the factor 1 exactly
#pragma unroll
        for (int b = 0; b < 8; b++) {
            if (b > 0) { N = N * (1.0 - C.c0[b]); D = D * (1.0 + C.c0[b]); }
            N = N * (1.0 - C.c1[b]); D = D * (1.0 + C.c1[b]);
            N = N * (1.0 - C.c2[b]); D = D * (1.0 + C.c2[b]);
        }
        de_fast_span_reg(sL + i, (size_t)RM, cl, xj, C, N, D);
        const double rho_ = pL[FP_T * RM] * pR[FP_T * RM] * (N / D);
        return de_fast_value(P.ising_id, RM, rho_, pL, xj, weights[j], xk, weights[k], pR);
    }
    for (int t = 0; t < cr; t++) {                               // ranges that start at dim p+1
        const double c = xk * (t < cap ? sR[(size_t)t * RM + q] : nR[(size_t)t * RM]);
        if (c <= TTX_FCUT) break;
        N = N * (1.0 - c); D = D * (1.0 + c);
    }
    for (int a = 0; a < cl; a++) {                               // ranges through dim p
        const double A = xj * (a < cap ? sL[(size_t)a * RM + i] : nL[(size_t)a * RM]);
        if (A <= TTX_FCUT) break;
        N = N * (1.0 - A); D = D * (1.0 + A);
        for (int t = 0; t < cr; t++) {
            const double c = xk * (t < cap ? sR[(size_t)t * RM + q] : nR[(size_t)t * RM]);
            if (A * c <= TTX_FCUT) break;
            N = N * __builtin_fma(-A, c, 1.0);
            D = D * __builtin_fma(A, c, 1.0);
        }
    }
    const double rho = pL[FP_T * RM] * pR[FP_T * RM] * (N / D);
    return de_fast_value(P.ising_id, RM, rho, pL, xj, weights[j], xk, weights[k], pR);
}

// mvn: cross term dL_i' S_LR dR_q = sum over the left dims a of dvL_i[a] * (S dR_q)[a]
__device__ __forceinline__ double mvn_fast_cross(const DevProb &P, int g, int p, int first, int i, int q)
{
    const int RM = P.RM;
    const double *dv = fast_dv(P, 0, g, p - 1, first) + i, *Y = fast_near(P, 1, g, p + 1, first) + q;
    double x = 0.0;
#pragma unroll 8
    for (int a = 0; a < p - 1; a++) x = __builtin_fma(dv[(size_t)a * RM], Y[(size_t)a * RM], x);
    return x;
}
// mvn value of the element (left pivot i | j | k | right pivot q), 0-based, given the cross term X of (i, q)
__device__ __forceinline__ double mvn_fast_value(const DevProb &P, int g, int p, int first, int i, int j, int k, int q, double X)
{
    const int RM = P.RM, m = P.d;
    const double *YL = fast_near(P, 0, g, p - 1, first) + i, *YR = fast_near(P, 1, g, p + 1, first) + q;
    const double QL = fast_piv(P, 0, g, p - 1, first)[FP_T * RM + i], QR = fast_piv(P, 1, g, p + 1, first)[FP_T * RM + q];
    const double *S = P.auxS, *mu = P.aux;
    const double dj = P.par[j] - mu[p - 1], dk = P.par[k] - mu[p];
    const double yj = YL[(size_t)(p - 1) * RM] + YR[(size_t)(p - 1) * RM], yk = YL[(size_t)p * RM] + YR[(size_t)p * RM];
    const double sjj = S[(size_t)(p - 1) * (m + 1)], skk = S[(size_t)p * (m + 1)], sjk = S[(size_t)(p - 1) + (size_t)m * p];
    const double Q = (QL + QR + 2 * X) + 2 * (dj * yj + dk * yk) + (sjj * dj * dj + skk * dk * dk + 2 * sjk * dj * dk);
    return ttx_exp(-0.5 * Q) / P.mvn_norm;
}
// mvn table entry of ONE pivot FROM SCRATCH by one wave: xs = dv = x - mu over the pivot's len dims (LDS, visible to the wave),
// d0 = 0-based dimension of entry 0; columns (stride RM) dvo[len], Y[m] = S dv, piv[FP_T] = dv' S dv
__device__ __forceinline__ void mvn_entry_scratch(const DevProb &P, const double *xs, int len, int d0, double *dvo, double *Y, double *piv, int lane)
{
    const int m = P.d, RM = P.RM;
    const double *S = P.auxS;
    for (int k = lane; k < len; k += 64) dvo[(size_t)k * RM] = xs[k];
    double q = 0.0;
    for (int row = lane; row < m; row += 64) {
        double y = 0.0;
#pragma unroll 4
        for (int k = 0; k < len; k++) y = __builtin_fma(S[(size_t)row + (size_t)m * (d0 + k)], xs[k], y);
        Y[(size_t)row * RM] = y;
        if (row >= d0 && row < d0 + len) q = __builtin_fma(xs[row - d0], y, q);
    }
    q = wave_sum(q);
    if (lane == 0) piv[FP_T * RM] = q;
}
// ... and of a pivot that is its PARENT extended by one dimension (0-based dnew, deviation dval) next to the bond: side 0 appends
// the dimension behind the parent's plen dims, side 1 puts it in front.  O(m) instead of O(m len).
__device__ __forceinline__ void mvn_entry_child(const DevProb &P, int side, const double *pdv, const double *pY, const double *ppiv, int plen, int dnew,
                                                double dval, double *cdv, double *cY, double *cpiv, int lane)
{
    const int m = P.d, RM = P.RM;
    const double *S = P.auxS;
    if (side == 0) { for (int k = lane; k < plen; k += 64) cdv[(size_t)k * RM] = pdv[(size_t)k * RM]; if (lane == 0) cdv[(size_t)plen * RM] = dval; }
    else { for (int k = lane; k < plen; k += 64) cdv[(size_t)(k + 1) * RM] = pdv[(size_t)k * RM]; if (lane == 0) cdv[0] = dval; }
    for (int row = lane; row < m; row += 64) cY[(size_t)row * RM] = __builtin_fma(S[(size_t)row + (size_t)m * dnew], dval, pY[(size_t)row * RM]);
    if (lane == 0) cpiv[FP_T * RM] = ppiv[FP_T * RM] + 2.0 * dval * pY[(size_t)dnew * RM] + S[(size_t)dnew * (m + 1)] * dval * dval;
}
// mvn value of one arbitrary point by ONE wave (boundary corners): dv[m] in LDS; Q = dv' S dv with one row of S per lane
__device__ __forceinline__ double mvn_fast_point_wave(const DevProb &P, const double *dv, int lane)
{
    const int m = P.d;
    const double *S = P.auxS;
    double q = 0.0;
    for (int row = lane; row < m; row += 64) {
        double y = 0.0;
#pragma unroll 4
        for (int k = 0; k < m; k++) y = __builtin_fma(S[(size_t)row + (size_t)m * k], dv[k], y);
        q = __builtin_fma(dv[row], y, q);
    }
    q = wave_sum(q);
    return ttx_exp(-0.5 * q) / P.mvn_norm;
}

// ------------------------------------------------------------------------------------------------
// k_fast_tables: per bond step, one WAVE per pivot of the two sets a bond step evaluates with (left pivots of bond p-1,
// right pivots of bond p+1).  grid = (2*RM, groups), 64 threads, dynamic LDS 2*(d+8) doubles.
// ------------------------------------------------------------------------------------------------
template <int FUN>
__global__ __launch_bounds__(64) void k_fast_tables(DevProb P, int dir, int pp)
{
    extern __shared__ __align__(16) double dyn[];
    const int g = blockIdx.y, lane = threadIdx.x, m = P.d, RM = P.RM;
    const GroupState &gs = P.gs[g];
    const int first = gs.first, last = gs.last;
    if (pp > last - first + 1) return;
    const int p = (dir == 1) ? first + pp - 1 : last + 1 - pp;
    const int *r = P.r + (size_t)g * (m + 2);
    const int side = (int)blockIdx.x / RM, c = (int)blockIdx.x % RM;
    if (c >= (side == 0 ? r[p - 1] : r[p + 1])) return;
    const int A = p - 1, B = m - p - 1, len = side == 0 ? A : B;
    const short *tab = side == 0 ? L_ptr(P, g, p - 1, first) : R_ptr(P, g, p + 1, first);
    double *near = fast_near(P, side, g, side == 0 ? p - 1 : p + 1, first), *piv = fast_piv(P, side, g, side == 0 ? p - 1 : p + 1, first);
    double *xs = dyn, *ws = dyn + m + 8;
    if (FUN == FUN_MVN) {
        // dv = x - mu over the pivot's dims (0-based dim of entry k: k on the left, p+1+k on the right); Y = S dv; Q = dv' Y
        const int d0 = side == 0 ? 0 : p + 1;
        for (int k = lane; k < len; k += 64) xs[k] = P.par[tab[(size_t)k * RM + c] - 1] - P.aux[d0 + k];
        __syncthreads();
        mvn_entry_scratch(P, xs, len, d0, fast_dv(P, side, g, side == 0 ? p - 1 : p + 1, first) + c, near + c, piv + c, lane);
        return;
    }
    const double *nodes = P.par - 1, *weights = P.par + P.n[1] - 1;
    for (int k = lane; k < len; k += 64) { const int ix = tab[(size_t)k * RM + c]; xs[k] = nodes[ix]; ws[k] = weights[ix]; }
    __syncthreads();
    fast_entry_scratch(xs, ws, len, side, near + c, piv + c, RM, lane);
}
