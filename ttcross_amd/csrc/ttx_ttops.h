// ttx_ttops.h -- device kernels of the tt_lib utilities on the resident tensor train:
//   dtt_ort  (lib/tt.f90:130-198): left-to-right Householder QR of the tall-skinny (r0*n) x r1 unfoldings,
//            R pushed into the next core by an fp64-MFMA GEMM
//   dtt_svd  (lib/tt.f90:307-368): right-to-left truncated SVD of the r0 x (n*r1) unfoldings
//   dtt_norm (:1074-1092), dtt_dot (:1155-1175)
// All kernels work on COMPACT column-major work buffers (pack/unpack from the padded core layout of ttx_dev.h).
#pragma once
#include <hip/hip_runtime.h>
#include "ttx_dev.h"

typedef double dbl4 __attribute__((ext_vector_type(4)));

// padded core (i + RM*j + SS*s) -> compact (i + r0*(j + n*s)); tr: write the transpose of the r0 x (n*r1) unfolding
__global__ void k_pack_core(const double *core, double *w, int r0, int n, int r1, int RM, size_t SS, int tr)
{
    const size_t tot = (size_t)r0 * n * r1;
    for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < tot; x += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(x % r0); size_t c = x / r0; int j = (int)(c % n), s = (int)(c / n);
        double v = core[i + (size_t)RM * j + SS * s];
        if (tr) w[c + (size_t)n * r1 * i] = v; else w[x] = v;
    }
}
__global__ void k_unpack_core(double *core, const double *w, int r0, int n, int r1, int RM, size_t SS, int tr, double scale)
{
    const size_t tot = (size_t)r0 * n * r1;
    for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < tot; x += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(x % r0); size_t c = x / r0; int j = (int)(c % n), s = (int)(c / n);
        double v = tr ? w[c + (size_t)n * r1 * i] : w[x];
        core[i + (size_t)RM * j + SS * s] = scale * v;
    }
}

__device__ __forceinline__ double tt_block_sum(double v, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    for (int x = 0; x < nw; x++) r += sh[x];
    return r;
}

// Householder QR of A (m x n, compact, in place), LAPACK dgeqr2 / dlarfg / dlarf then dorg2r, one 1024-thread
// workgroup: the reflector lives in LDS; Rout (mn x n) gets the upper trapezoid (zeros below the diagonal); on exit A
// holds the first mn columns of Q.
// Round 2: the two rank-1 stages of every reflector (w = tau A^T v, A -= v w^T) map THREADS TO ROWS and loop over the
// columns in chunks of 32 -- every thread has up to 32 independent, coalesced loads in flight and 32 running sums --
// instead of one wave walking one column with a dependent load per 64 rows (that version was bound by the latency of
// ~2 m n^2 / 64 serial L2 round trips: 4.9 ms for a 1632 x 32 unfolding; now ~0.3 ms).
#define QR_CH 32
__device__ __forceinline__ void qr_apply_reflector(double *A, int m, int i, int len, int c_lo, int c_hi, double tau, const double *vsh,
                                                   double *wsh, double (*part)[QR_CH])
{
    // for columns c in [c_lo, c_hi): w_c = tau * v^T A(i:, c);  A(i:, c) -= v w_c
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    for (int c0 = c_lo; c0 < c_hi; c0 += QR_CH) {
        const int nc = min(QR_CH, c_hi - c0);
        double acc[QR_CH];
#pragma unroll
        for (int k = 0; k < QR_CH; k++) acc[k] = 0.0;
        for (int r = tid; r < len; r += nt) {
            const double v = vsh[r];
            const double *row = A + i + r + (size_t)m * c0;
#pragma unroll
            for (int k = 0; k < QR_CH; k++) if (k < nc) acc[k] += v * row[(size_t)m * k];
        }
#pragma unroll
        for (int k = 0; k < QR_CH; k++) {
            if (k < nc) {
                double q = acc[k];
                for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
                if (lane == 0) part[wv][k] = q;
            }
        }
        __syncthreads();
        if (tid < nc) { double q = 0.0; for (int x = 0; x < nw; x++) q += part[x][tid]; wsh[c0 + tid] = q * tau; }
        __syncthreads();
        for (int r = tid; r < len; r += nt) {
            const double v = vsh[r];
            double *row = A + i + r + (size_t)m * c0;
#pragma unroll
            for (int k = 0; k < QR_CH; k++) if (k < nc) row[(size_t)m * k] -= v * wsh[c0 + k];
        }
        __syncthreads();
    }
}
// the two rank-1 stages with ONE WAVE PER COLUMN (the panel is in LDS: a dependent LDS read per 64 rows is cheap)
__device__ __forceinline__ void qr_apply_reflector_cols(double *A, int m, int i, int len, int c_lo, int c_hi, double tau, const double *vsh, double *wsh)
{
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    for (int c = c_lo + wv; c < c_hi; c += nw) {
        const double *cc = A + i + (size_t)m * c;
        double q = 0.0;
        for (int r = lane; r < len; r += 64) q += vsh[r] * cc[r];
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        if (lane == 0) wsh[c] = q * tau;
    }
    __syncthreads();
    for (int c = c_lo + wv; c < c_hi; c += nw) {
        double *cc = A + i + (size_t)m * c;
        const double wc = wsh[c];
        for (int r = lane; r < len; r += 64) cc[r] -= vsh[r] * wc;
    }
    __syncthreads();
}
// Householder QR of an LDS-RESIDENT m x n matrix (column-major, ld = m; dgeqr2 / dlarfg / dlarf, then dorg2r) with THREE workgroup
// barriers per reflector of the factorisation and TWO per reflector of the generation (round 3; the phases used to be separated by
// seven and five).  The kernels are chains of short phases, so the barriers were most of their time:
//   * the squared norm that reflector i+1 needs is accumulated by the wave that updates column i+1 (no separate reduction phase);
//   * the dot products w = tau A' v take v on the fly from column i (v_r = x_r * scale, the values dlarfg stores), while all threads
//     write the same v into vsh for the update phase -- no phase of its own for v;
//   * column i is rewritten (v below the diagonal, beta on it; in dorg2r: the column of Q) during the update phase, which reads vsh.
// One wave per column in both phases.  After the first part the upper triangle holds R (emit(r, c, value) receives it, zeros
// below the diagonal), after the second A holds the first min(m, n) columns of Q.  tauv: n doubles of LDS.
__device__ __forceinline__ double jac_group_sum(double v, int tpp, int lane);
template <class EMIT>
__device__ __forceinline__ void qr_lds3(double *A, int m, int n, double *vsh, double *wsh, double *tauv, EMIT emit)
{
    __shared__ double s_xn2, s_tau, s_beta, s_scale;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6, mn = m < n ? m : n;
    // wave sum on the DPP data path (a __shfl_xor of a double is two ds_bpermute round trips: twelve of them per dot product)
    auto wsum = [&](double q) { return jac_group_sum(q, 64, lane); };
    if (wv == 0) {
        double q = 0.0;
        for (int r = 1 + lane; r < m; r += 64) q += A[r] * A[r];
        q = wsum(q);
        if (lane == 0) s_xn2 = q;
    }
    __syncthreads();
    for (int i = 0; i < mn; i++) {
        double *x = A + i + (size_t)m * i;
        const int len = m - i;
        if (tid == 0) {
            const double alpha = x[0], xn = sqrt(s_xn2);
            if (xn == 0.0) { s_tau = 0.0; s_beta = alpha; s_scale = 0.0; }
            else {
                const double beta = -copysign(hypot(alpha, xn), alpha);
                s_tau = (beta - alpha) / beta; s_beta = beta; s_scale = 1.0 / (alpha - beta);
            }
            tauv[i] = s_tau;
        }
        __syncthreads();                                                                        // (1) tau, beta, scale
        const double sc = s_scale, tau = s_tau;
        for (int r = tid; r < len; r += nt) vsh[r] = (r == 0) ? 1.0 : x[r] * sc;
        for (int c = i + 1 + wv; c < n; c += nw) {
            const double *cc = A + i + (size_t)m * c;
            double q = 0.0;
#pragma unroll 4
            for (int r = lane; r < len; r += 64) q += ((r == 0) ? 1.0 : x[r] * sc) * cc[r];
            q = wsum(q);
            if (lane == 0) wsh[c] = q * tau;
        }
        __syncthreads();                                                                        // (2) w, vsh
        for (int c = i + 1 + wv; c < n; c += nw) {
            double *cc = A + i + (size_t)m * c;
            const double wc = wsh[c];
            double acc = 0.0;
#pragma unroll 4
            for (int r = lane; r < len; r += 64) { const double nv = cc[r] - vsh[r] * wc; cc[r] = nv; if (r >= 2) acc += nv * nv; }
            if (c == i + 1) { acc = wsum(acc); if (lane == 0) s_xn2 = acc; }
        }
        for (int r = 1 + tid; r < len; r += nt) x[r] = vsh[r];
        if (tid == 0) x[0] = s_beta;
        __syncthreads();                                                                        // (3) trailing columns, column i, next norm
    }
    for (int t = tid; t < mn * n; t += nt) { const int r = t % mn, c = t / mn; emit(r, c, (r <= c) ? A[r + (size_t)m * c] : 0.0); }
    __syncthreads();
    for (int i = mn - 1; i >= 0; i--) {                                                         // dorg2r
        double *x = A + i + (size_t)m * i;
        const int len = m - i;
        const double tau = tauv[i];
        for (int r = tid; r < len; r += nt) vsh[r] = (r == 0) ? 1.0 : x[r];
        for (int c = i + 1 + wv; c < mn; c += nw) {
            const double *cc = A + i + (size_t)m * c;
            double q = 0.0;
#pragma unroll 4
            for (int r = lane; r < len; r += 64) q += ((r == 0) ? 1.0 : x[r]) * cc[r];
            q = wsum(q);
            if (lane == 0) wsh[c] = q * tau;
        }
        __syncthreads();
        for (int c = i + 1 + wv; c < mn; c += nw) {
            double *cc = A + i + (size_t)m * c;
            const double wc = wsh[c];
#pragma unroll 4
            for (int r = lane; r < len; r += 64) cc[r] -= vsh[r] * wc;
        }
        for (int r = tid; r < len; r += nt) x[r] = (r == 0) ? 1.0 - tau : -tau * vsh[r];
        for (int r = tid; r < i; r += nt) A[r + (size_t)m * i] = 0.0;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Householder QR with the matrix in REGISTERS and one workgroup barrier per reflector (round 3, second version).
// Wave w owns the columns c = w, w + nw, w + 2 nw, ...; lane l holds the rows l, l + 64, ... of them (a[NC][MR]).  Only the
// reflectors live in LDS (V, m x mn, and their taus).  Step i: the owner of column i+1 applies reflector i to that column FIRST,
// forms reflector i+1 from it (norm by a DPP wave sum, dlarfg scalars in every lane) and publishes it, then updates its other
// columns; every other wave applies reflector i to its columns; one barrier; next step.  The critical path per reflector is one
// LDS read of v, two wave sums and the dlarfg scalars -- no phase is separated from the next by more than that one barrier
// (qr_lds3 above needs three, and walks LDS for every element).  dorg2r needs no barrier at all: column c of Q is
// H_0 ... H_c e_c, formed by the owning wave in registers from the read-only reflectors and written straight to global memory.
// Panel p = blockIdx.x: rows [p rbs, p rbs + m) of M (leading dimension ldm) -> Q_p (m x mn) into the same rows of Qout (ldq; in
// place allowed), R_p (mn x n, zeros below the diagonal) at Rst + p rstep (ldr), taus at tau_out + p n (may be null).
// The host guarantees m <= 64 MR, n <= nw NC, (m mn + n + 2) doubles of dynamic LDS.
// ------------------------------------------------------------------------------------------------------------------
template <int MR, int NC>
__global__ __launch_bounds__(1024) void k_qr_own(int rows, int n, int rbs, const double *M, int ldm, double *Qout, int ldq, double *Rst, int ldr, int rstep,
                                                 double *tau_out)
{
    extern __shared__ __align__(16) double sm[];
    const int p = blockIdx.x, r0 = p * rbs, m = min(rbs, rows - r0), mn = m < n ? m : n;
    double *tauv = sm;
    double *V = sm + ((n + 2) & ~1);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
    double a[NC][MR];
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int c = wv + nw * k;
#pragma unroll
        for (int j = 0; j < MR; j++) { const int r = lane + 64 * j; a[k][j] = (c < n && r < m) ? M[r0 + r + (size_t)ldm * c] : 0.0; }
    }
    // reflector i from the (fully updated) column i held in slot ks of this wave: V(:, i), tauv[i]; the column keeps beta
    auto make_reflector = [&](int i, int ks) {
        double col[MR];
#pragma unroll
        for (int j = 0; j < MR; j++) { col[j] = 0.0;
#pragma unroll
            for (int k = 0; k < NC; k++) if (k == ks) col[j] = a[k][j]; }
        double cand = 0.0, q = 0.0;
#pragma unroll
        for (int j = 0; j < MR; j++) { const int r = lane + 64 * j; if (j == (i >> 6)) cand = col[j]; if (r > i) q += col[j] * col[j]; }
        // alpha from its lane (the index is wave-uniform: v_readlane, no LDS round trip), the norm by one DPP wave sum; dlapy2 as
        // a plain sqrt(alpha^2 + xn2) while that cannot over- or underflow (hypot costs more than the rest of the scalar chain)
        const int al_lane = i & 63;
        const long long cb = __double_as_longlong(cand);
        const double alpha = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(cb >> 32), al_lane) << 32) |
                                                  (unsigned int)__builtin_amdgcn_readlane((int)cb, al_lane));
        const double xn2 = jac_group_sum(q, 64, lane);
        double tau = 0.0, beta = alpha, sc = 0.0;
        if (xn2 != 0.0) {
            const double s2 = alpha * alpha + xn2;
            const double nrm = (s2 > 1e-280 && s2 < 1e280) ? sqrt(s2) : hypot(alpha, sqrt(xn2));
            beta = -copysign(nrm, alpha);
            tau = (beta - alpha) / beta; sc = 1.0 / (alpha - beta);
        }
#pragma unroll
        for (int j = 0; j < MR; j++) {
            const int r = lane + 64 * j;
            if (r >= i && r < m) V[r + (size_t)m * i] = (r == i) ? 1.0 : col[j] * sc;
            const double nv = (r == i) ? beta : (r > i) ? 0.0 : col[j];
#pragma unroll
            for (int k = 0; k < NC; k++) if (k == ks) a[k][j] = nv;
        }
        if (lane == 0) tauv[i] = tau;
    };
    auto apply = [&](int k, const double *v, double tau) {          // a[k] -= tau (v' a[k]) v
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < MR; j++) q += v[j] * a[k][j];
        q = jac_group_sum(q, 64, lane) * tau;
#pragma unroll
        for (int j = 0; j < MR; j++) a[k][j] -= v[j] * q;
    };
    if (wv == 0 && mn > 0) make_reflector(0, 0);
    __syncthreads();
    for (int i = 0; i < mn; i++) {
        double v[MR];
#pragma unroll
        for (int j = 0; j < MR; j++) { const int r = lane + 64 * j; v[j] = (r >= i && r < m) ? V[r + (size_t)m * i] : 0.0; }
        const double tau = tauv[i];
        const int nx = i + 1, kx = nx / nw;                              // the next column and its slot in its owner
        const bool mine = (nx < n) && (nx % nw == wv);
        if (mine) {
#pragma unroll
            for (int k = 0; k < NC; k++) if (k == kx) apply(k, v, tau);
            if (nx < mn) make_reflector(nx, kx);
        }
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const int c = wv + nw * k;
            if (c > i && c < n && !(mine && k == kx)) apply(k, v, tau);
        }
        __syncthreads();
    }
    // R: rows r <= c of column c are final since step r
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int c = wv + nw * k;
        if (c < n) {
#pragma unroll
            for (int j = 0; j < MR; j++) { const int r = lane + 64 * j; if (r < mn) Rst[(size_t)p * rstep + r + (size_t)ldr * c] = (r <= c) ? a[k][j] : 0.0; }
        }
    }
    if (tau_out) for (int x = tid; x < mn; x += blockDim.x) tau_out[(size_t)p * n + x] = tauv[x];
    // Q = H_0 ... H_{mn-1} (first mn columns), column by column in the owner's registers
    int cmax = -1;
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int c = wv + nw * k;
        if (c < mn) cmax = c;
#pragma unroll
        for (int j = 0; j < MR; j++) a[k][j] = (lane + 64 * j == c) ? 1.0 : 0.0;
    }
    for (int i = cmax; i >= 0; i--) {
        double v[MR];
#pragma unroll
        for (int j = 0; j < MR; j++) { const int r = lane + 64 * j; v[j] = (r >= i && r < m) ? V[r + (size_t)m * i] : 0.0; }
        const double tau = tauv[i];
#pragma unroll
        for (int k = 0; k < NC; k++) { const int c = wv + nw * k; if (c < mn && c >= i) apply(k, v, tau); }
    }
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int c = wv + nw * k;
        if (c < mn) {
#pragma unroll
            for (int j = 0; j < MR; j++) { const int r = lane + 64 * j; if (r < m) Qout[r0 + r + (size_t)ldq * c] = a[k][j]; }
        }
    }
}
__host__ __device__ inline size_t qr_own_lds_doubles(int m, int n) { return (size_t)((n + 2) & ~1) + (size_t)m * (m < n ? m : n); }

// INLDS: the whole unfolding is staged in LDS (m n doubles <= the budget the host checked), factored there and written back
template <bool INLDS>
__global__ __launch_bounds__(1024) void k_qr(int m, int n, double *Ag, double *Rout, double *tau_out)
{
    extern __shared__ __align__(16) double sm[];
    double *vsh = sm;                 // m
    double *wsh = sm + m;             // n  (INLDS: + n for the taus)
    double *A = INLDS ? sm + ((m + 2 * n + 1) & ~1) : Ag;
    __shared__ double red[16];
    __shared__ double part[16][QR_CH];
    __shared__ double s_tau, s_beta, s_scale;
    const int tid = threadIdx.x, nt = blockDim.x, mn = m < n ? m : n;
    if (INLDS) {
        for (size_t x = tid; x < (size_t)m * n; x += nt) A[x] = Ag[x];
        __syncthreads();
        double *tauv = wsh + n;                          // (the host sizes the LDS for m + 2 n + 4 doubles ahead of the matrix)
        qr_lds3(A, m, n, vsh, wsh, tauv, [&](int r, int c, double v) { Rout[r + (size_t)mn * c] = v; });
        for (int x = tid; x < mn; x += nt) tau_out[x] = tauv[x];
        for (size_t x = tid; x < (size_t)m * mn; x += nt) Ag[x] = A[x];
        return;
    }
    for (int i = 0; i < mn; i++) {
        double *x = A + i + (size_t)m * i;
        const int len = m - i;
        double p = 0.0;
        for (int r = 1 + tid; r < len; r += nt) p += x[r] * x[r];
        const double xn2 = tt_block_sum(p, red);
        if (tid == 0) {
            const double alpha = x[0], xn = sqrt(xn2);
            if (xn == 0.0) { s_tau = 0.0; s_beta = alpha; s_scale = 0.0; }
            else {
                const double beta = -copysign(hypot(alpha, xn), alpha);
                s_tau = (beta - alpha) / beta; s_beta = beta; s_scale = 1.0 / (alpha - beta);
            }
            tau_out[i] = s_tau;
        }
        __syncthreads();
        for (int r = tid; r < len; r += nt) { double v = (r == 0) ? 1.0 : x[r] * s_scale; vsh[r] = v; if (r > 0) x[r] = v; }
        __syncthreads();
        if (tid == 0) x[0] = s_beta;
        if (INLDS) qr_apply_reflector_cols(A, m, i, len, i + 1, n, s_tau, vsh, wsh);
        else qr_apply_reflector(A, m, i, len, i + 1, n, s_tau, vsh, wsh, part);
    }
    // R
    __syncthreads();
    for (int x = tid; x < mn * n; x += nt) { int r = x % mn, c = x / mn; Rout[x] = (r <= c) ? A[r + (size_t)m * c] : 0.0; }
    __syncthreads();
    // dorg2r
    for (int i = mn - 1; i >= 0; i--) {
        double *x = A + i + (size_t)m * i;
        const int len = m - i;
        const double tau = tau_out[i];
        for (int r = tid; r < len; r += nt) vsh[r] = (r == 0) ? 1.0 : x[r];
        __syncthreads();
        if (INLDS) qr_apply_reflector_cols(A, m, i, len, i + 1, mn, tau, vsh, wsh);
        else qr_apply_reflector(A, m, i, len, i + 1, mn, tau, vsh, wsh, part);
        for (int r = tid; r < len; r += nt) x[r] = (r == 0) ? 1.0 - tau : -tau * vsh[r];
        for (int r = tid; r < i; r += nt) A[r + (size_t)m * i] = 0.0;
        __syncthreads();
    }
    if (INLDS) { for (size_t x = tid; x < (size_t)m * mn; x += nt) Ag[x] = A[x]; }
}

// ------------------------------------------------------------------------------------------------------------------
// Tall-skinny QR (TSQR) for unfoldings that do not fit one CU's LDS (round 2): the rows are cut into panels of `rbs` rows,
// every panel is factored by its own workgroup ENTIRELY IN LDS (k_qr_panel, the arithmetic of k_qr<true>), the n x n
// triangles are stacked and factored again (recursively, host loop in qr_tsqr), and the explicit Q of a level is the
// product of its panels' Q with the n x n blocks of the level above (k_gemm_mfma_panels, fp64 MFMA).  One dependent
// single-workgroup pass over a 1632 x 32 unfolding streamed from L2 took 1.5 ms; the panels take ~0.1 ms side by side.
// Block p: rows [p rbs, p rbs + mp) of M (rows x n, leading dimension rows) -> Q_p in Qout (same layout), R_p (n x n, zeros
// below the diagonal) in rows [p n, p n + n) of Rst (leading dimension ldr).  The host guarantees mp >= n.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_qr_panel(int rows, int n, int rbs, const double *M, double *Qout, double *Rst, int ldr)
{
    extern __shared__ __align__(16) double sm[];
    const int p = blockIdx.x, r0 = p * rbs, m = min(rbs, rows - r0);
    double *vsh = sm;                 // rbs
    double *wsh = sm + rbs;           // n
    double *tauv = wsh + n;           // n
    double *A = sm + ((rbs + 2 * n + 1) & ~1);
    const int tid = threadIdx.x, nt = blockDim.x;
    for (size_t x = tid; x < (size_t)m * n; x += nt) { const int r = (int)(x % m), c = (int)(x / m); A[x] = M[r0 + r + (size_t)rows * c]; }
    __syncthreads();
    qr_lds3(A, m, n, vsh, wsh, tauv, [&](int r, int c, double v) { Rst[(size_t)p * n + r + (size_t)ldr * c] = v; });
    for (size_t x = tid; x < (size_t)m * n; x += nt) { const int r = (int)(x % m), c = (int)(x / m); Qout[r0 + r + (size_t)rows * c] = A[x]; }
}
__host__ __device__ inline size_t qr_panel_lds_doubles(int rbs, int n) { return (size_t)((rbs + 2 * n + 1) & ~1) + (size_t)rbs * n; }

// C_p (mp x n) = A_p (mp x n) * B_p (n x n) for the panels p = blockIdx.z of a level: A_p = rows [p rbs, ..) of A (lda), B_p = rows
// [p n, p n + n) of B (ldb), C_p = rows [p rbs, ..) of C (ldc); one wave per 16 x 16 tile, v_mfma_f64_16x16x4_f64 as in k_gemm_mfma
__global__ __launch_bounds__(256) void k_gemm_mfma_panels(int rows, int n, int rbs, const double *A, int lda, const double *B, int ldb, double *C, int ldc)
{
    const int p = blockIdx.z, r0 = p * rbs, M = min(rbs, rows - r0);
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int tm = blockIdx.y * 16, tn = (blockIdx.x * 4 + wave) * 16;
    if (tn >= n || tm >= M) return;
    const double *Ap = A + r0, *Bp = B + (size_t)p * n;
    double *Cp = C + r0;
    const int ar = tm + (l & 15), bc = tn + (l & 15), kq = l >> 4;
    dbl4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < n; k0 += 4) {
        const int k = k0 + kq;
        const double a = (ar < M && k < n) ? Ap[ar + (size_t)lda * k] : 0.0;
        const double b = (bc < n && k < n) ? Bp[k + (size_t)ldb * bc] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int row = tm + (l >> 4) + 4 * reg, col = tn + (l & 15);
        if (row < M && col < n) Cp[row + (size_t)ldc * col] = acc[reg];
    }
}

// C (M x N, ldc) = A (M x K, lda) * B (K x N, ldb), fp64 on the matrix cores: one wave per 16x16 tile of C,
// v_mfma_f64_16x16x4_f64 per k-step of 4.  Operand lane maps (cdna_hip_programming.md sec. 3): A[row l&15][k l>>4],
// B[k l>>4][col l&15]; C/D: col = l&15, row = (l>>4) + 4*reg.
__global__ __launch_bounds__(256) void k_gemm_mfma(int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc)
{
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int tm = blockIdx.y * 16, tn = (blockIdx.x * 4 + wave) * 16;
    if (tn >= N) return;
    const int ar = tm + (l & 15), bc = tn + (l & 15), kq = l >> 4;
    dbl4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + kq;
        const double a = (ar < M && k < K) ? A[ar + (size_t)lda * k] : 0.0;
        const double b = (bc < N && k < K) ? B[k + (size_t)ldb * bc] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int row = tm + (l >> 4) + 4 * reg, col = tn + (l & 15);
        if (row < M && col < N) C[row + (size_t)ldc * col] = acc[reg];
    }
}

// ||x||_2^2 of n doubles -> out[0] (single block)
__global__ __launch_bounds__(1024) void k_sumsq(size_t n, const double *x, double *out)
{
    __shared__ double red[16];
    double p = 0.0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) p += x[i] * x[i];
    p = tt_block_sum(p, red);
    if (threadIdx.x == 0) out[0] = p;
}
// norm equalisation of dtt_ort without a host round trip (lib/tt.f90:166-172, 184-188): nrm = ||x||_2; x *= 1/nrm;
// acc[0] += log(nrm).  scale_x = 0: x is left alone and 1/nrm is kept in acc[1] (the last core, scaled at the end).
__global__ __launch_bounds__(1024) void k_norm_log(size_t n, double *x, double *acc, int scale_x)
{
    __shared__ double red[16];
    __shared__ double s_inv;
    double p = 0.0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) p += x[i] * x[i];
    p = tt_block_sum(p, red);
    if (threadIdx.x == 0) {
        const double nrm = sqrt(p);
        s_inv = (nrm != 0.0) ? 1.0 / nrm : 1.0;
        if (nrm != 0.0) acc[0] += log(nrm);
        if (!scale_x) acc[1] = s_inv;
    }
    __syncthreads();
    if (scale_x && s_inv != 1.0) for (size_t i = threadIdx.x; i < n; i += blockDim.x) x[i] *= s_inv;
}
// final rescaling of every core by exp(acc[0]/d) (times acc[1] for the core whose norm was only measured)
__global__ void k_scal_core_acc(double *core, int r0, int n, int r1, int RM, size_t SS, const double *acc, int d, int extra)
{
    const double a = ttx_exp(acc[0] / d) * (extra ? acc[1] : 1.0);
    const size_t tot = (size_t)r0 * n * r1;
    for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < tot; x += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(x % r0); size_t c = x / r0; int j = (int)(c % n), s = (int)(c / n);
        core[i + (size_t)RM * j + SS * s] *= a;
    }
}
__global__ void k_fill_const(size_t n, double *x, double a)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = a;
}
__global__ void k_scal(size_t n, double *x, double a)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= a;
}
// scale every element of a padded core
__global__ void k_scal_core(double *core, int r0, int n, int r1, int RM, size_t SS, double a)
{
    const size_t tot = (size_t)r0 * n * r1;
    for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < tot; x += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(x % r0); size_t c = x / r0; int j = (int)(c % n), s = (int)(c / n);
        core[i + (size_t)RM * j + SS * s] *= a;
    }
}

// One-sided Jacobi (Hestenes) SVD of X (p x q, p >= q, compact, in place -> U), V (q x q) accumulated, s[q]; then
// ordering (perm[j] = column holding the j-th largest singular value) and the reference's chop (lib/mat.f90:433-458)
// info[0] = kept rank rr, info[1] = sweeps; sout[q] sorted singular values; one 1024-thread workgroup, the
// q/2 disjoint column pairs of a round-robin round rotate concurrently.
// sum over the tpp (16, 32 or 64) consecutive lanes of a pair's thread group on the DPP data path (ttx_kernels.h is included
// before this header): quad butterflies, two row rotations, then the row broadcasts -- a __shfl_xor of a double is two
// ds_bpermute round trips, 36 of them in sequence per rotation round made a round last ~2.8 us
__device__ __forceinline__ double jac_group_sum(double v, int tpp, int lane)
{
    v = v + dpp_d<0xb1, 0xf>(v);
    v = v + dpp_d<0x4e, 0xf>(v);
    v = v + dpp_d<0x124, 0xf>(v);
    v = v + dpp_d<0x128, 0xf>(v);                          // every lane: a sum of its row of 16 (association differs from quad to quad)
    if (tpp == 16)                                          // one value for the whole group: lane 15's
        return __longlong_as_double(__builtin_amdgcn_mov_dpp(__double_as_longlong(v), 0x15f, 0xf, 0xf, false));
    v = v + dpp_d<0x142, 0xa>(v);                          // rows 1 and 3: + the row before
    if (tpp == 32) {
        const double lo = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), 31) << 32) |
                                               (unsigned int)__builtin_amdgcn_readlane((int)__double_as_longlong(v), 31));
        const double hi = readlane63(v);
        return lane < 32 ? lo : hi;
    }
    v = v + dpp_d<0x143, 0xc>(v);                          // rows 2 and 3: + the first half
    return readlane63(v);
}
// in_lds: X and V live in dynamic LDS ((p + q) q doubles) for the whole iteration -- a rotation round is a dependent chain of
// column reads, a reduction and column writes, ~3.5 us per round out of L2 (0.9 ms for a 32 x 32 matrix), a tenth of that in LDS
__global__ __launch_bounds__(1024) void k_jacobi_svd(int p, int q, double *Xg, double *Vg, double *sout, int *perm, int *info,
                                                     int has_tol, double tol, int rmax, int in_lds)
{
    extern __shared__ __align__(16) double jsm[];
    __shared__ int s_rot;
    __shared__ double ssh[256];
    __shared__ int psh[256];
    const int tid = threadIdx.x, nt = blockDim.x;
    double *X = in_lds ? jsm : Xg, *V = in_lds ? jsm + (size_t)p * q : Vg;
    if (in_lds) for (int x = tid; x < p * q; x += nt) X[x] = Xg[x];
    for (int x = tid; x < q * q; x += nt) V[x] = ((x % q) == (x / q)) ? 1.0 : 0.0;
    __syncthreads();
    const int qq = (q + 1) & ~1;                 // players of the tournament (one dummy if q is odd)
    const int npair = qq / 2;
    int tpp = nt / npair; if (tpp > 64) tpp = 64; if (tpp < 1) tpp = 1;
    // threads per pair: a power of two <= 64 so that a pair never straddles a wave
    int t2 = 1; while (t2 * 2 <= tpp) t2 *= 2; tpp = t2;
    const int pairs_per_pass = nt / tpp;
    int ltpp = 0; while ((1 << ltpp) < tpp) ltpp++;
    // columns count as orthogonal at sqrt(p) eps (the criterion of LAPACK's one-sided Jacobi, dgesvj); a threshold BELOW eps (1e-16,
    // round 2) is met only by chance: 2 of the 62 cores of the D_64 train ran all 60 sweeps, the others 6-7
    const double jtol = 2.220446049250313e-16 * sqrt((double)p);
    int sweeps = 0;
    for (int sweep = 0; sweep < 60; sweep++) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int round = 0; round < qq - 1; round++) {
            for (int p0 = 0; p0 < npair; p0 += pairs_per_pass) {
                const int pi = p0 + (tid >> ltpp), sub = tid & (tpp - 1);
                if (pi < npair) {
                    // (round + pi) mod (qq - 1) and (round - pi) mod (qq - 1) by one conditional step each: an integer division is ~40
                    // instructions at the head of every round's dependent chain
                    int a = round + pi; if (a >= qq - 1) a -= qq - 1;
                    if (pi == 0) a = qq - 1;
                    int b = round - pi; if (b < 0) b += qq - 1;
                    if (a > b) { int t = a; a = b; b = t; }
                    if (b < q) {
                        double *xa = X + (size_t)p * a, *xb = X + (size_t)p * b;
                        double al = 0.0, be = 0.0, ga = 0.0;
                        for (int i = sub; i < p; i += tpp) { double u = xa[i], w = xb[i]; al += u * u; be += w * w; ga += u * w; }
                        if (tpp >= 16) { al = jac_group_sum(al, tpp, tid & 63); be = jac_group_sum(be, tpp, tid & 63); ga = jac_group_sum(ga, tpp, tid & 63); }
                        else for (int o = tpp >> 1; o > 0; o >>= 1) { al += __shfl_xor(al, o, 64); be += __shfl_xor(be, o, 64); ga += __shfl_xor(ga, o, 64); }
                        // the test and the rotation with three long operations (sqrt, division, rsqrt) in the dependent chain of a round
                        // instead of six: t = tan of the rotation angle, the smaller root of t^2 + 2 zeta t - 1 = 0 with
                        // zeta = (be - al) / (2 ga), written without forming zeta
                        const double ab = al * be;
                        const bool orth = (ab < 1e300) ? (ga * ga <= jtol * jtol * ab) : (fabs(ga) <= jtol * sqrt(al) * sqrt(be));
                        if (!(orth || ga == 0.0)) {
                            if (sub == 0) atomicAdd(&s_rot, 1);
                            const double dd = be - al, g2 = 2.0 * ga;
                            const double t = (dd >= 0.0 ? g2 : -g2) / (fabs(dd) + sqrt(dd * dd + g2 * g2));
                            const double c = rsqrt(1.0 + t * t), sn = c * t;
                            for (int i = sub; i < p; i += tpp) { double u = xa[i], w = xb[i]; xa[i] = c * u - sn * w; xb[i] = sn * u + c * w; }
                            double *va = V + (size_t)q * a, *vb = V + (size_t)q * b;
                            for (int i = sub; i < q; i += tpp) { double u = va[i], w = vb[i]; va[i] = c * u - sn * w; vb[i] = sn * u + c * w; }
                        }
                    }
                }
                __syncthreads();
            }
        }
        sweeps = sweep + 1;
        if (s_rot == 0) break;
        __syncthreads();
    }
    // singular values and normalised columns
    for (int j = tid / 64; j < q; j += nt / 64) {
        double *xj = X + (size_t)p * j; double s2 = 0.0;
        for (int i = tid & 63; i < p; i += 64) s2 += xj[i] * xj[i];
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const double s = sqrt(s2);
        if ((tid & 63) == 0) ssh[j] = s;
        if (s > 0.0) for (int i = tid & 63; i < p; i += 64) xj[i] /= s;
    }
    __syncthreads();
    if (in_lds) {
        for (int x = tid; x < p * q; x += nt) Xg[x] = X[x];
        for (int x = tid; x < q * q; x += nt) Vg[x] = V[x];
    }
    if (tid == 0) {
        for (int j = 0; j < q; j++) psh[j] = j;
        for (int j = 0; j < q - 1; j++) {               // selection sort, descending
            int mx = j;
            for (int k = j + 1; k < q; k++) if (ssh[psh[k]] > ssh[psh[mx]]) mx = k;
            int t = psh[j]; psh[j] = psh[mx]; psh[mx] = t;
        }
        for (int j = 0; j < q; j++) { perm[j] = psh[j]; sout[j] = ssh[psh[j]]; }
        // chop (lib/mat.f90:433-458)
        int r = q; double er2 = 0.0;
        if (rmax > 0 && rmax < r) { for (int i = rmax; i < r; i++) er2 += sout[i] * sout[i]; r = rmax; }
        if (has_tol) {
            double nrm = 0.0; for (int i = 0; i < q; i++) nrm += sout[i] * sout[i];
            const double bound = tol * tol * nrm;
            double er = er2 + sout[r - 1] * sout[r - 1];
            while (er < bound) { er2 = er; r--; er = er + sout[r - 1] * sout[r - 1]; }
        }
        info[0] = r; info[1] = sweeps;
    }
}

// rank and singular values of one core straight into pinned host memory (dtt_svd needs them to shape the next launches): the host
// polls host[0] for `seq` instead of two device-to-host copies and a stream synchronisation (30-40 us per core)
__global__ void k_svd_report(const double *sv, const int *info, int q, volatile double *host, double seq)
{
    for (int j = threadIdx.x; j < q; j += blockDim.x) host[3 + j] = sv[j];
    if (threadIdx.x == 0) { host[1] = (double)info[0]; host[2] = (double)info[1]; }
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); host[0] = seq; }
}
// out (rows x rr) <- in(:, perm[0..rr)) * diag(scale[c])   (scale == nullptr: 1)
__global__ void k_take_cols(int rows, int rr, const double *in, int ldin, const int *perm, const double *scale, double sdiv, double *out)
{
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < rows * rr; x += gridDim.x * blockDim.x) {
        int i = x % rows, c = x / rows;
        out[x] = in[i + (size_t)ldin * perm[c]] * (scale ? scale[c] * sdiv : 1.0);
    }
}
// out (rr x cols) <- transpose of in (cols x rr)
__global__ void k_transpose(int rows, int cols, const double *in, double *out)
{
    for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < (size_t)rows * cols; x += (size_t)gridDim.x * blockDim.x) {
        size_t i = x % rows, c = x / rows;
        out[c + (size_t)cols * i] = in[x];
    }
}

// ztt_quad, batched: T(i,k) = sum_j U(i,j,k) w_j per core and weight set (zgemv order), stored as (re, im) planes.
// grid (core, weight set), one block each; tq: [nf][d][2][RM*RM]
__global__ __launch_bounds__(256) void k_zquad_build(int d, int RM, int NM, size_t SS, const int *n1, const int *r, const double *const *cores,
                                                     const double *w, size_t wstride, double *tq)
{
    const int p = blockIdx.x + 1, f = blockIdx.y;
    if (!cores[p]) return;                           // a core of another process (multi-process job: every process builds its own)
    const int r0 = r[p - 1], r1 = r[p], n = n1[p];
    size_t off = 0;
    for (int c = 1; c < p; c++) off += n1[c];
    const double *wf = w + (size_t)f * wstride + 2 * off;
    const double *A = cores[p];
    double *tr = tq + (((size_t)f * (d + 1) + p) * 2) * RM * RM, *ti = tr + (size_t)RM * RM;
    for (int x = threadIdx.x; x < r0 * r1; x += blockDim.x) {
        const int i = x % r0, k = x / r0;
        const double *a = A + i + SS * k;
        double yr = 0.0, yi = 0.0;
        for (int j = 0; j < n; j++) { const double v = a[(size_t)RM * j]; yr = yr + wf[2 * j] * v; yi = yi + wf[2 * j + 1] * v; }
        tr[i + RM * k] = yr; ti[i + RM * k] = yi;
    }
}
// left-to-right chain of the complex matrices (zgemm 'n','n' order), one block per weight set
__global__ __launch_bounds__(256) void k_zquad_chain(int d, int RM, const int *r, const double *tq, double *out)
{
    extern __shared__ __align__(16) double sh[];     // 4 * RM*RM : prev(re,im), next(re,im)
    const int f = blockIdx.x, tid = threadIdx.x, mym = r[0];
    double *pr = sh, *pi = sh + RM * RM, *nr = pi + RM * RM, *ni = nr + RM * RM;
    const double *t1 = tq + (((size_t)f * (d + 1) + 1) * 2) * RM * RM;
    for (int x = tid; x < mym * r[1]; x += blockDim.x) { pr[(x % mym) + RM * (x / mym)] = t1[(x % mym) + RM * (x / mym)]; pi[(x % mym) + RM * (x / mym)] = t1[(size_t)RM * RM + (x % mym) + RM * (x / mym)]; }
    __syncthreads();
    for (int p = 2; p <= d; p++) {
        const double *tr = tq + (((size_t)f * (d + 1) + p) * 2) * RM * RM, *ti = tr + (size_t)RM * RM;
        const int r0 = r[p - 1], r1 = r[p];
        for (int x = tid; x < mym * r1; x += blockDim.x) {
            const int i = x % mym, j = x / mym;
            double cr = 0.0, ci = 0.0;
            for (int l = 0; l < r0; l++) {          // c += temp * a, temp = curr(l,j), a = prev(i,l)
                const double br = tr[l + RM * j], bi = ti[l + RM * j], ar = pr[i + RM * l], ai = pi[i + RM * l];
                cr = cr + (br * ar - bi * ai); ci = ci + (br * ai + bi * ar);
            }
            nr[i + RM * j] = cr; ni[i + RM * j] = ci;
        }
        __syncthreads();
        double *t = pr; pr = nr; nr = t; t = pi; pi = ni; ni = t;
    }
    if (tid == 0) { out[2 * f] = pr[0]; out[2 * f + 1] = pi[0]; }
}

// ztt_quad on a multi-process job (lib/dmrgg.f90:1418-1523 with mybonds: every rank multiplies the matrices of its own cores, the
// partial products are combined over the ranks): the chain over the cores [plo, phi] of THIS process, result (r[plo-1] x r[phi],
// re / im planes) into slot `slot` of part: [nf][nslots][2][RM*RM]; then k_zquad_fold multiplies the slots in rank order.
__global__ __launch_bounds__(256) void k_zquad_chain_seg(int d, int RM, const int *r, const double *tq, int plo, int phi, int slot, int nslots, double *part)
{
    extern __shared__ __align__(16) double sh[];     // 4 * RM*RM : prev(re,im), next(re,im)
    const int f = blockIdx.x, tid = threadIdx.x, mym = r[plo - 1];
    double *pr = sh, *pi = sh + RM * RM, *nr = pi + RM * RM, *ni = nr + RM * RM;
    const double *t1 = tq + (((size_t)f * (d + 1) + plo) * 2) * RM * RM;
    for (int x = tid; x < mym * r[plo]; x += blockDim.x) { pr[(x % mym) + RM * (x / mym)] = t1[(x % mym) + RM * (x / mym)]; pi[(x % mym) + RM * (x / mym)] = t1[(size_t)RM * RM + (x % mym) + RM * (x / mym)]; }
    __syncthreads();
    for (int p = plo + 1; p <= phi; p++) {
        const double *tr = tq + (((size_t)f * (d + 1) + p) * 2) * RM * RM, *ti = tr + (size_t)RM * RM;
        const int r0 = r[p - 1], r1 = r[p];
        for (int x = tid; x < mym * r1; x += blockDim.x) {
            const int i = x % mym, j = x / mym;
            double cr = 0.0, ci = 0.0;
            for (int l = 0; l < r0; l++) {
                const double br = tr[l + RM * j], bi = ti[l + RM * j], ar = pr[i + RM * l], ai = pi[i + RM * l];
                cr = cr + (br * ar - bi * ai); ci = ci + (br * ai + bi * ar);
            }
            nr[i + RM * j] = cr; ni[i + RM * j] = ci;
        }
        __syncthreads();
        double *t = pr; pr = nr; nr = t; t = pi; pi = ni; ni = t;
    }
    double *o = part + (((size_t)f * nslots + slot) * 2) * RM * RM;
    const int myn = r[phi];
    for (int x = tid; x < mym * myn; x += blockDim.x) { o[(x % mym) + RM * (x / mym)] = pr[(x % mym) + RM * (x / mym)]; o[(size_t)RM * RM + (x % mym) + RM * (x / mym)] = pi[(x % mym) + RM * (x / mym)]; }
}
// product of the slots in rank order; dims[2 s], dims[2 s + 1]: rows / columns of slot s; out[2 f], out[2 f + 1]
__global__ __launch_bounds__(256) void k_zquad_fold(int RM, int nslots, const int *dims, const double *part, double *out)
{
    extern __shared__ __align__(16) double sh[];
    const int f = blockIdx.x, tid = threadIdx.x, mym = dims[0];
    double *pr = sh, *pi = sh + RM * RM, *nr = pi + RM * RM, *ni = nr + RM * RM;
    const double *t1 = part + (((size_t)f * nslots) * 2) * RM * RM;
    for (int x = tid; x < mym * dims[1]; x += blockDim.x) { pr[(x % mym) + RM * (x / mym)] = t1[(x % mym) + RM * (x / mym)]; pi[(x % mym) + RM * (x / mym)] = t1[(size_t)RM * RM + (x % mym) + RM * (x / mym)]; }
    __syncthreads();
    for (int s = 1; s < nslots; s++) {
        const double *tr = part + (((size_t)f * nslots + s) * 2) * RM * RM, *ti = tr + (size_t)RM * RM;
        const int r0 = dims[2 * s], r1 = dims[2 * s + 1];
        for (int x = tid; x < mym * r1; x += blockDim.x) {
            const int i = x % mym, j = x / mym;
            double cr = 0.0, ci = 0.0;
            for (int l = 0; l < r0; l++) {
                const double br = tr[l + RM * j], bi = ti[l + RM * j], ar = pr[i + RM * l], ai = pi[i + RM * l];
                cr = cr + (br * ar - bi * ai); ci = ci + (br * ai + bi * ar);
            }
            nr[i + RM * j] = cr; ni[i + RM * j] = ci;
        }
        __syncthreads();
        double *t = pr; pr = nr; nr = t; t = pi; pi = ni; ni = t;
    }
    if (tid == 0) { out[2 * f] = pr[0]; out[2 * f + 1] = pi[0]; }
}
