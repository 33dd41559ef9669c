/*
 * ttx_oracle_tt.c -- CPU restatement (TEST ORACLE) of the reference's tt_lib utilities on the sweep's result:
 * dtt_ort (lib/tt.f90:130-198), dtt_svd (:307-368) with d_svd/chop (lib/mat.f90:340-385, 433-458),
 * dtt_norm (:1074-1092), dtt_dot (:1155-1175), dtt_ijk (:630-652).  Test infrastructure only (see ttx_oracle.h).
 *
 * LAPACK is an external dependency of the reference.  dgeqrf/dorgqr are restated as the unblocked Householder
 * algorithm of the published LAPACK (dgeqr2 / dlarfg / dlarf / dorg2r; at these sizes, n <= 128 < NX, dgeqrf itself
 * runs unblocked).  dgesvd is restated as a one-sided Jacobi (Hestenes) SVD -- a different algorithm for the same
 * decomposition, so parity with the reference is to rounding-level tolerance (tests/golden/ttops_*.txt).
 */
#include "ttx_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double nrm2(size_t n, const double *x) { double s = 0.0; for (size_t i = 0; i < n; i++) s += x[i] * x[i]; return sqrt(s); }

/* dgeqr2: A (m x n, ld m) -> R in the upper triangle, reflectors below; tau[min(m,n)] */
static void geqr2(int m, int n, double *a, double *tau)
{
    int k = m < n ? m : n;
    for (int i = 0; i < k; i++) {
        double *x = a + i + (size_t)m * i;
        double alpha = x[0], xn = nrm2((size_t)(m - i - 1), x + 1);
        if (xn == 0.0) { tau[i] = 0.0; }
        else {
            double beta = -copysign(hypot(alpha, xn), alpha);
            tau[i] = (beta - alpha) / beta;
            double sc = 1.0 / (alpha - beta);
            for (int r = 1; r < m - i; r++) x[r] *= sc;
            x[0] = beta;
        }
        if (i < n - 1) {
            double aii = x[0]; x[0] = 1.0;
            for (int c = i + 1; c < n; c++) {
                double *cc = a + i + (size_t)m * c, w = 0.0;
                for (int r = 0; r < m - i; r++) w += x[r] * cc[r];
                w *= tau[i];
                for (int r = 0; r < m - i; r++) cc[r] -= x[r] * w;
            }
            x[0] = aii;
        }
    }
}
/* dorg2r: first k columns of Q from the reflectors in a (m x k used), in place */
static void org2r(int m, int k, double *a, const double *tau)
{
    for (int i = k - 1; i >= 0; i--) {
        double *x = a + i + (size_t)m * i;
        if (i < k - 1) {
            x[0] = 1.0;
            for (int c = i + 1; c < k; c++) {
                double *cc = a + i + (size_t)m * c, w = 0.0;
                for (int r = 0; r < m - i; r++) w += x[r] * cc[r];
                w *= tau[i];
                for (int r = 0; r < m - i; r++) cc[r] -= x[r] * w;
            }
        }
        for (int r = 1; r < m - i; r++) x[r] *= -tau[i];
        x[0] = 1.0 - tau[i];
        for (int r = 0; r < i; r++) a[r + (size_t)m * i] = 0.0;
    }
}

/* one-sided Jacobi SVD of X (p x q, p >= q, ld p): X -> U (in place, p x q), s[q] descending, V (q x q) */
static void jacobi_svd(int p, int q, double *x, double *s, double *v)
{
    for (int i = 0; i < q * q; i++) v[i] = 0.0;
    for (int i = 0; i < q; i++) v[i + q * i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        int rot = 0;
        for (int a = 0; a < q - 1; a++)
            for (int b = a + 1; b < q; b++) {
                double *xa = x + (size_t)p * a, *xb = x + (size_t)p * b, al = 0, be = 0, ga = 0;
                for (int i = 0; i < p; i++) { al += xa[i] * xa[i]; be += xb[i] * xb[i]; ga += xa[i] * xb[i]; }
                if (fabs(ga) <= 1e-16 * sqrt(al * be) || ga == 0.0) continue;
                rot++;
                double zeta = (be - al) / (2.0 * ga);
                double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < p; i++) { double u = xa[i], w = xb[i]; xa[i] = c * u - sn * w; xb[i] = sn * u + c * w; }
                double *va = v + (size_t)q * a, *vb = v + (size_t)q * b;
                for (int i = 0; i < q; i++) { double u = va[i], w = vb[i]; va[i] = c * u - sn * w; vb[i] = sn * u + c * w; }
            }
        if (!rot) break;
    }
    for (int j = 0; j < q; j++) { s[j] = nrm2((size_t)p, x + (size_t)p * j); if (s[j] > 0) for (int i = 0; i < p; i++) x[i + (size_t)p * j] /= s[j]; }
    for (int j = 0; j < q - 1; j++) {           /* selection sort, descending */
        int mx = j;
        for (int k = j + 1; k < q; k++) if (s[k] > s[mx]) mx = k;
        if (mx != j) {
            double t = s[j]; s[j] = s[mx]; s[mx] = t;
            for (int i = 0; i < p; i++) { t = x[i + (size_t)p * j]; x[i + (size_t)p * j] = x[i + (size_t)p * mx]; x[i + (size_t)p * mx] = t; }
            for (int i = 0; i < q; i++) { t = v[i + (size_t)q * j]; v[i + (size_t)q * j] = v[i + (size_t)q * mx]; v[i + (size_t)q * mx] = t; }
        }
    }
}

/* lib/mat.f90:433-458 */
static int chop(int n, const double *s, int has_tol, double tol, int rmax, double *err)
{
    int r = n; double er2 = 0.0;
    if (rmax > 0 && rmax < r) { for (int i = rmax; i < r; i++) er2 += s[i] * s[i]; r = rmax; }
    if (has_tol) {
        double nrm = nrm2((size_t)n, s), bound = tol * tol * nrm * nrm, er = er2 + s[r - 1] * s[r - 1];
        while (er < bound) { er2 = er; r--; er = er + s[r - 1] * s[r - 1]; }
    }
    if (err) *err = sqrt(er2);
    return r;
}

static size_t csz(const ttxo_tt *t, int k) { return (size_t)t->r[k] * t->n[k] * t->r[k + 1]; }   /* k = 0..d-1 */

/* lib/tt.f90:130-198 */
void ttxo_tt_ort(ttxo_tt *t)
{
    const int d = t->d;
    double lognrm = 0.0;
    for (int k = 0; k < d - 1; k++) {
        int mm = t->r[k] * t->n[k], nn = t->r[k + 1], mn = mm < nn ? mm : nn, kk = t->n[k + 1] * t->r[k + 2];
        double *u = (double *)malloc(sizeof(double) * (size_t)mm * nn), *tau = (double *)calloc((size_t)mn + 1, sizeof(double));
        double *mat = (double *)calloc((size_t)mn * nn + 1, sizeof(double));
        memcpy(u, t->cores[k], sizeof(double) * (size_t)mm * nn);
        geqr2(mm, nn, u, tau);
        for (int j = 0; j < nn; j++) for (int i = 0; i < mn; i++) mat[i + (size_t)mn * j] = (i <= j && i < mm) ? u[i + (size_t)mm * j] : 0.0;
        double nrm = nrm2((size_t)mn * nn, mat);
        if (nrm != 0.0) { for (size_t x = 0; x < (size_t)mn * nn; x++) mat[x] *= 1.0 / nrm; lognrm += log(nrm); }
        org2r(mm, mn, u, tau);
        double *nxt = (double *)calloc((size_t)mn * kk + 1, sizeof(double));
        for (int c = 0; c < kk; c++) for (int l = 0; l < nn; l++) { double b = t->cores[k + 1][l + (size_t)nn * c]; for (int i = 0; i < mn; i++) nxt[i + (size_t)mn * c] += mat[i + (size_t)mn * l] * b; }
        free(t->cores[k]); free(t->cores[k + 1]);
        t->cores[k] = (double *)malloc(sizeof(double) * ((size_t)mm * mn + 1));
        memcpy(t->cores[k], u, sizeof(double) * (size_t)mm * mn);
        t->cores[k + 1] = nxt;
        t->r[k + 1] = mn;
        free(u); free(tau); free(mat);
    }
    double nrm = nrm2(csz(t, d - 1), t->cores[d - 1]);
    if (nrm != 0.0) { for (size_t x = 0; x < csz(t, d - 1); x++) t->cores[d - 1][x] *= 1.0 / nrm; lognrm += log(nrm); }
    lognrm /= d;
    nrm = exp(lognrm);
    for (int k = 0; k < d; k++) for (size_t x = 0; x < csz(t, k); x++) t->cores[k][x] *= nrm;
}

/* lib/tt.f90:307-368; has_tol / rmax as the optional arguments of svd() */
void ttxo_tt_svd(ttxo_tt *t, double tol, int rmax)
{
    const int d = t->d;
    if (d <= 1) return;
    ttxo_tt_ort(t);
    double lognrm = 0.0;
    for (int k = d - 1; k >= 1; k--) {
        int mm = t->r[k], nn = t->n[k] * t->r[k + 1], mn = mm < nn ? mm : nn, kk = t->r[k - 1] * t->n[k - 1];
        /* SVD of A (mm x nn) through its transpose: At (nn x mm) = U' S V'^T  =>  A = V' S U'^T */
        double *at, *s = (double *)calloc((size_t)mn + 1, sizeof(double)), *vv;
        double *u, *v;      /* u: mm x mn, v: mn x nn */
        if (mm <= nn) {
            at = (double *)malloc(sizeof(double) * (size_t)nn * mm);
            for (int i = 0; i < mm; i++) for (int j = 0; j < nn; j++) at[j + (size_t)nn * i] = t->cores[k][i + (size_t)mm * j];
            vv = (double *)malloc(sizeof(double) * (size_t)mm * mm);
            jacobi_svd(nn, mm, at, s, vv);                      /* at -> U' (nn x mm), vv = V' (mm x mm) */
            u = vv;                                             /* A's left vectors = V' */
            v = (double *)malloc(sizeof(double) * (size_t)mn * nn);
            for (int c = 0; c < mn; c++) for (int j = 0; j < nn; j++) v[c + (size_t)mn * j] = at[j + (size_t)nn * c];
            free(at);
        } else {
            at = (double *)malloc(sizeof(double) * (size_t)mm * nn);
            memcpy(at, t->cores[k], sizeof(double) * (size_t)mm * nn);
            vv = (double *)malloc(sizeof(double) * (size_t)nn * nn);
            jacobi_svd(mm, nn, at, s, vv);                      /* at -> U (mm x nn), vv = V (nn x nn) */
            u = at;
            v = (double *)malloc(sizeof(double) * (size_t)mn * nn);
            for (int c = 0; c < mn; c++) for (int j = 0; j < nn; j++) v[c + (size_t)mn * j] = vv[j + (size_t)nn * c];
            free(vv);
        }
        int rr = chop(mn, s, 1, tol, rmax, NULL);
        double nrm = nrm2((size_t)rr, s);
        if (nrm != 0.0) { for (int j = 0; j < rr; j++) s[j] *= 1.0 / nrm; lognrm += log(nrm); }
        for (int j = 0; j < rr; j++) for (int i = 0; i < mm; i++) u[i + (size_t)mm * j] *= s[j];
        double *prv = (double *)calloc((size_t)kk * rr + 1, sizeof(double));
        for (int c = 0; c < rr; c++) for (int l = 0; l < mm; l++) { double b = u[l + (size_t)mm * c]; for (int i = 0; i < kk; i++) prv[i + (size_t)kk * c] += t->cores[k - 1][i + (size_t)kk * l] * b; }
        double *cur = (double *)malloc(sizeof(double) * ((size_t)rr * nn + 1));
        for (int j = 0; j < nn; j++) for (int c = 0; c < rr; c++) cur[c + (size_t)rr * j] = v[c + (size_t)mn * j];
        free(t->cores[k - 1]); free(t->cores[k]);
        t->cores[k - 1] = prv; t->cores[k] = cur; t->r[k] = rr;
        free(u); free(v); free(s);
    }
    double nrm = nrm2(csz(t, 0), t->cores[0]);
    if (nrm != 0.0) { for (size_t x = 0; x < csz(t, 0); x++) t->cores[0][x] *= 1.0 / nrm; lognrm += log(nrm); }
    lognrm /= d;
    nrm = exp(lognrm);
    for (int k = 0; k < d; k++) for (size_t x = 0; x < csz(t, k); x++) t->cores[k][x] *= nrm;
}

static ttxo_tt *tt_copy(const ttxo_tt *a)
{
    ttxo_tt *b = (ttxo_tt *)calloc(1, sizeof(ttxo_tt));
    b->d = a->d;
    b->n = (int32_t *)malloc(sizeof(int32_t) * (size_t)a->d); memcpy(b->n, a->n, sizeof(int32_t) * (size_t)a->d);
    b->r = (int32_t *)malloc(sizeof(int32_t) * ((size_t)a->d + 1)); memcpy(b->r, a->r, sizeof(int32_t) * ((size_t)a->d + 1));
    b->cores = (double **)calloc((size_t)a->d, sizeof(double *));
    for (int k = 0; k < a->d; k++) { b->cores[k] = (double *)malloc(sizeof(double) * (csz(a, k) + 1)); memcpy(b->cores[k], a->cores[k], sizeof(double) * csz(a, k)); }
    return b;
}
static void tt_free(ttxo_tt *b) { for (int k = 0; k < b->d; k++) free(b->cores[k]); free(b->cores); free(b->n); free(b->r); free(b); }

/* lib/tt.f90:1074-1092; tol < 0: absent */
double ttxo_tt_norm(const ttxo_tt *t, double tol)
{
    ttxo_tt *c = tt_copy(t);
    double nrm;
    if (tol >= 0.0) { ttxo_tt_svd(c, tol, 0); nrm = nrm2(csz(c, 0), c->cores[0]); }
    else { ttxo_tt_ort(c); nrm = nrm2(csz(c, c->d - 1), c->cores[c->d - 1]); }
    nrm = pow(nrm, c->d);
    tt_free(c);
    return nrm;
}

/* lib/tt.f90:1155-1175 */
double ttxo_tt_dot(const ttxo_tt *x, const ttxo_tt *y)
{
    const int d = x->d;
    double *phi = (double *)calloc(1, sizeof(double));
    phi[0] = 1.0;
    for (int i = 0; i < d; i++) {
        int rx0 = x->r[i], rx1 = x->r[i + 1], ry0 = y->r[i], ry1 = y->r[i + 1], n = x->n[i];
        double *res = (double *)calloc((size_t)rx0 * n * ry1 + 1, sizeof(double));     /* phi (rx0 x ry0) * Y (ry0 x n*ry1) */
        for (int c = 0; c < n * ry1; c++) for (int l = 0; l < ry0; l++) { double b = y->cores[i][l + (size_t)ry0 * c]; for (int a = 0; a < rx0; a++) res[a + (size_t)rx0 * c] += phi[a + (size_t)rx0 * l] * b; }
        double *np = (double *)calloc((size_t)rx1 * ry1 + 1, sizeof(double));          /* X^T (rx1 x rx0*n) * res (rx0*n x ry1) */
        for (int c = 0; c < ry1; c++) for (int a = 0; a < rx1; a++) { double s = 0.0; for (int l = 0; l < rx0 * n; l++) s += x->cores[i][l + (size_t)rx0 * n * a] * res[l + (size_t)rx0 * n * c]; np[a + (size_t)rx1 * c] = s; }
        free(phi); free(res); phi = np;
    }
    double v = phi[0];
    free(phi);
    return v;
}

/* lib/tt.f90:630-652 */
double ttxo_tt_ijk(const ttxo_tt *t, const int32_t *ind)
{
    const int d = t->d;
    double x[4096], z[4096];
    int r0 = t->r[d - 1];
    for (int a = 0; a < r0; a++) x[a] = t->cores[d - 1][a + (size_t)r0 * (ind[d - 1] - 1)];
    for (int i = d - 2; i >= 0; i--) {
        int q0 = t->r[i], q1 = t->r[i + 1], n = t->n[i];
        for (int a = 0; a < q0; a++) { double s = 0.0; for (int k = 0; k < q1; k++) s += t->cores[i][a + (size_t)q0 * ((ind[i] - 1) + (size_t)n * k)] * x[k]; z[a] = s; }
        memcpy(x, z, sizeof(double) * (size_t)q0);
    }
    return x[0];
}

/* helpers for the ctypes wrapper: C-heap allocation of a TT of given ranks (cores zero-filled) */
ttxo_tt *ttxo_tt_new(int d, const int32_t *n, const int32_t *r)
{
    ttxo_tt *b = (ttxo_tt *)calloc(1, sizeof(ttxo_tt));
    b->d = d;
    b->n = (int32_t *)malloc(sizeof(int32_t) * (size_t)d); memcpy(b->n, n, sizeof(int32_t) * (size_t)d);
    b->r = (int32_t *)malloc(sizeof(int32_t) * ((size_t)d + 1)); memcpy(b->r, r, sizeof(int32_t) * ((size_t)d + 1));
    b->cores = (double **)calloc((size_t)d, sizeof(double *));
    for (int k = 0; k < d; k++) b->cores[k] = (double *)calloc(csz(b, k) + 1, sizeof(double));
    return b;
}
void ttxo_tt_free(ttxo_tt *b) { if (b) tt_free(b); }

/* lib/dmrgg.f90:1418-1523 ztt_quad of a REAL tensor train with complex rank-1 weights (nproc = 1): per core
 * curr(:,k) = sum_j U(:,j,k) w_j (zgemv), running product prev*curr (zgemm), both in netlib order.
 * w: d blocks of n[k] complex weights, interleaved (re, im); out[0] = re, out[1] = im */
#include <complex.h>
void ttxo_tt_zquad(const ttxo_tt *t, const double *w, double *out)
{
    const int d = t->d;
    double complex *prev = NULL;
    int rf = t->r[0];
    size_t off = 0;
    for (int p = 0; p < d; p++) {
        int r0 = t->r[p], r1 = t->r[p + 1], n = t->n[p];
        double complex *curr = (double complex *)calloc((size_t)r0 * r1 + 1, sizeof(double complex));
        for (int k = 0; k < r1; k++)
            for (int j = 0; j < n; j++) {
                double complex temp = w[2 * (off + j)] + I * w[2 * (off + j) + 1];
                for (int i = 0; i < r0; i++) curr[i + (size_t)r0 * k] += temp * t->cores[p][i + (size_t)r0 * (j + (size_t)n * k)];
            }
        off += n;
        if (!prev) prev = curr;
        else {
            double complex *next = (double complex *)calloc((size_t)rf * r1 + 1, sizeof(double complex));
            for (int c = 0; c < r1; c++) for (int l = 0; l < r0; l++) { double complex temp = curr[l + (size_t)r0 * c]; for (int i = 0; i < rf; i++) next[i + (size_t)rf * c] += temp * prev[i + (size_t)rf * l]; }
            free(prev); free(curr); prev = next;
        }
    }
    out[0] = creal(prev[0]); out[1] = cimag(prev[0]);
    free(prev);
}
