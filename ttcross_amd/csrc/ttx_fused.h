// ttx_fused.h -- the whole sweep of one bond group in ONE launch (Ising C fast path).
//
// The multi-kernel path (ttx_kernels.h) spends most of a C-type run in per-launch overhead: six dependent
// launches per bond step, each re-reading the step state, re-staging the pivot tables and reducing across
// blocks.  For cheap integrands a bond step is small enough for one CU, so here one 1024-thread workgroup per
// bond group walks all bonds of its sweep (lib/dmrgg.f90:329-760) without leaving the kernel:
//   * the node/weight VALUES of both pivot tables (bond p-1 left, bond p+1 right) are staged once per bond in LDS
//     and serve the lottery candidates and every rook half-step;
//   * fibers (acol1 / arow1), the factor vector, the lottery lists and all arg-max reductions live in LDS;
//   * the neighbour fix-ups run as wave-shuffle wavefronts (no block barrier per LU step).
// Arithmetic and its order are identical to the multi-kernel path (and to the oracle): same device functions.
#pragma once
#include "ttx_kernels.h"

// 4-part index source as VALUES: dims 1..A from row (an, aw); dim A+1 = (s1n, s1w); dim A+2 = (s2n, s2w);
// dims A+3..m from row (bn, bw)
__device__ __forceinline__ double f_ising_c4v(int m, int A, const double *an, const double *aw, double s1n, double s1w,
                                              double s2n, double s2w, const double *bn, const double *bw)
{
    const int nb = m - A - 2;
    double v = 1.0, w = 1.0, vk = 1.0, wk = 1.0;
    auto vstep = [&](double xv) { vk = vk * xv; v = v + vk; };
    auto wstep = [&](double xv) { wk = wk * xv; w = w + wk; };
    chain8v<true>(bn, nb, vstep);
    vstep(s2n); vstep(s1n);
    chain8v<true>(an, A, vstep);
    chain8v<false>(an, A, wstep);
    wstep(s1n); wstep(s2n);
    chain8v<false>(bn, nb, wstep);
    double b = 1.0 / (v * w);
    double f = 2 * b;
    auto fstep = [&](double xv) { f = f * xv; };
    chain8v<false>(aw, A, fstep);
    fstep(s1w); fstep(s2w);
    chain8v<false>(bw, nb, fstep);
    return f;
}

#define FB 1024     // threads of the fused kernel

// block arg-max over FB threads, result broadcast to all threads through LDS
__device__ __forceinline__ void fused_argmax(double &a, double &v, int &idx, double *sha, double *shv, int *shi)
{
    block_argmax(a, v, idx, sha, shv, shi);
    if (threadIdx.x == 0) { sha[0] = a; shv[0] = v; shi[0] = idx; }
    __syncthreads();
    a = sha[0]; v = shv[0]; idx = shi[0];
    __syncthreads();
}

__global__ __launch_bounds__(FB) void k_sweep_fused(DevProb P, int dir, int nsteps)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ int zc[128], zr[128], zcs[128], zrs[128], keepc[128], keepr[128];
    __shared__ int nzc, nzr, nsc, nsr;
    __shared__ ttx_cdfseg segc[TTX_TABSEG], segr[TTX_TABSEG];
    __shared__ double sha[16], shv[16]; __shared__ int shi[16];
    __shared__ unsigned long long sA[2];
    if (P.ctl[0]) return;
    const int g = blockIdx.x, tid = threadIdx.x, m = P.d, RM = P.RM, NM = P.NM;
    const int lane = tid & 63, wv = tid >> 6;
    GroupState &gs = P.gs[g];
    const int first = gs.first, last = gs.last, nbonds = last - first + 1;
    int *r = P.r + (size_t)g * (m + 2);
    const int VS = ((m + 7) & ~7) + 8;
    const int n1m = P.n[1];
    // LDS carve-up
    double *par = dyn;
    double *XL = par + ((P.npar + 1) & ~1);            // RM rows x (VS node values, VS weight values)
    double *XR = XL + (size_t)RM * 2 * VS;
    double *acol = XR + (size_t)RM * 2 * VS;           // RM*NM
    double *arow = acol + (size_t)RM * NM;
    double *xs = arow + (size_t)RM * NM;               // RM
    int *lot = (int *)(xs + ((RM + 1) & ~1));          // 4 * nlotmax
    for (int x = tid; x < P.npar; x += FB) par[x] = P.par[x];
    if (tid == 0) {                                    // sweep start, :325-327
        gs.pivotmax = -1.0; gs.pivotmin = -1.0;
        int *rr = P.rr + (size_t)g * (m + 2);
        for (int s = 0; s <= m; s++) rr[s] = r[s];
    }
    __syncthreads();
    double amax = gs.amax, pivotmax = gs.pivotmax, pivotmin = gs.pivotmin;
    const double pivotmax_prev = gs.pivotmax_prev;
    long long neval = gs.neval;
    unsigned long long rngpos = gs.rngpos;
    double bytes_half = gs.bytes_half; long long n_resid = gs.n_resid;

    for (int pp = 1; pp <= nsteps; pp++) {
        if (pp > nbonds) break;
        const int p = (dir == 1) ? first + pp - 1 : last + 1 - pp;          // :330-331
        const int r0 = r[p - 1], r1 = r[p], r2 = r[p + 1], n1 = P.n[p], n2 = P.n[p + 1];
        const int nlot = r0 + n1 + n2 + r2;
        double *Cp = core_ptr(P, P.col, g, p, first), *Wq = core_ptr(P, P.row, g, p + 1, first);
        double *Ap = core_ptr(P, P.arg, g, p, first), *Aq = core_ptr(P, P.arg, g, p + 1, first);
        const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
        // ---- stage the value tables of both pivot sets ----
        for (int x = tid; x < r0 * VS; x += FB) {
            const int c = x / VS, o = x - c * VS;
            const int ix = (o < p - 1) ? (int)Lt[(size_t)o * RM + c] : 1;
            XL[(size_t)c * 2 * VS + o] = par[ix - 1]; XL[(size_t)c * 2 * VS + VS + o] = par[n1m + ix - 1];
        }
        for (int x = tid; x < r2 * VS; x += FB) {
            const int c = x / VS, o = x - c * VS;
            const int ix = (o < m - p - 1) ? (int)Rt[(size_t)o * RM + c] : 1;
            XR[(size_t)c * 2 * VS + o] = par[ix - 1]; XR[(size_t)c * 2 * VS + VS + o] = par[n1m + ix - 1];
        }
        // ---- lottery (:410-484) ----
        if (tid == 32) sA[0] = ttx_minstd_pow(2 * rngpos + 1);
        if (tid == 33) sA[1] = ttx_minstd_pow(2 * (rngpos + nlot) + 1);
        const unsigned long long bil = ttx_minstd_pow(2ull * tid);
        const int *vp = vip_ptr(P, g, p, first);
        if (tid < r1) {
            zc[tid] = (vp[4 * tid + 0] - 1) + r0 * (vp[4 * tid + 1] - 1) + 1;
            zr[tid] = (vp[4 * tid + 2] - 1) + n2 * (vp[4 * tid + 3] - 1) + 1;
        }
        __syncthreads();
        if (tid < r1) {
            int a = zc[tid], b = zr[tid], ra = 0, rb = 0;
            for (int u = 0; u < r1; u++) { ra += (zc[u] < a) || (zc[u] == a && u < tid); rb += (zr[u] < b) || (zr[u] == b && u < tid); }
            zcs[ra] = a; zrs[rb] = b;
        }
        __syncthreads();
        if (tid < r1) { keepc[tid] = (tid == 0) || (zcs[tid] != zcs[tid - 1]); keepr[tid] = (tid == 0) || (zrs[tid] != zrs[tid - 1]); }
        __syncthreads();
        if (tid < r1) {
            int pc = 0, pr = 0;
            for (int u = 0; u < tid; u++) { pc += keepc[u]; pr += keepr[u]; }
            if (keepc[tid]) zc[pc] = zcs[tid];
            if (keepr[tid]) zr[pr] = zrs[tid];
            if (tid == r1 - 1) { nzc = pc + keepc[tid]; nzr = pr + keepr[tid]; }
        }
        __syncthreads();
        const int Kc = r0 * n1 - nzc, Kr = n2 * r2 - nzr;
        if (tid < 64) { if (tid < P.cdf_ns[Kc]) segc[tid] = P.cdf_tab[(size_t)Kc * TTX_TABSEG + tid]; if (tid == 0) nsc = P.cdf_ns[Kc]; }
        else if (tid < 128) { const int t2 = tid - 64; if (t2 < P.cdf_ns[Kr]) segr[t2] = P.cdf_tab[(size_t)Kr * TTX_TABSEG + t2]; if (t2 == 0) nsr = P.cdf_ns[Kr]; }
        __syncthreads();
        double ma = 0.0, ba = -1.0, bv = 0.0; int bi = INT_MAX;
        if (tid < nlot) {
            const int il = tid;
            const double d1 = ttx_flang_from_word(ttx_mulmod31(sA[0], bil)), d2 = ttx_flang_from_word(ttx_mulmod31(sA[1], bil));
            const int x = ttx_lottery_index(segc, nsc, Kc, r0 * n1, zc, nzc, d1);
            const int y = ttx_lottery_index(segr, nsr, Kr, n2 * r2, zr, nzr, d2);
            const int i = (x - 1) % r0 + 1, j = (x - 1) / r0 + 1, k = (y - 1) % n2 + 1, q = (y - 1) / n2 + 1;
            lot[4 * il] = i; lot[4 * il + 1] = j; lot[4 * il + 2] = k; lot[4 * il + 3] = q;
            const double *rl = XL + (size_t)(i - 1) * 2 * VS, *rq = XR + (size_t)(q - 1) * 2 * VS;
            const double f = f_ising_c4v(m, p - 1, rl, rl + VS, par[j - 1], par[n1m + j - 1], par[k - 1], par[n1m + k - 1], rq, rq + VS);
            ma = fabs(f);
            const double *c = Cp + (i - 1) + (size_t)RM * (j - 1), *w = Wq + (k - 1) + (size_t)NM * (q - 1);
            double t = 0.0;
#pragma unroll 8
            for (int s = 0; s < r1; s++) t = t + c[P.SS * s] * w[P.SW * s];
            bv = f - t; ba = fabs(bv); bi = il;
        }
        ma = block_max(ma, sha);
        amax = fmax(amax, ma);
        fused_argmax(ba, bv, bi, sha, shv, shi);
        neval += nlot; rngpos += 2ull * nlot;
        if (bi == INT_MAX) bi = 0;                       // every residual a NaN: the first candidate, as idamax
        int ii = lot[4 * bi], jj = lot[4 * bi + 1], kk = lot[4 * bi + 2], qq = lot[4 * bi + 3];
        double pivot = bv;
        // ---- rook half-steps (:516-582) / piv = 0 (:492-513) ----
        int havecol = 0, haverow = 0, crs = 0, done = 0;
        const int H = (P.piv == 0) ? 2 : 2 * P.piv;
        for (int h = 0; h < H && !done; h++) {
            const bool iscol = (P.piv == 0) ? (h == 0) : (((h + (dir == 2 ? 1 : 0)) & 1) == 0);
            const int nf = iscol ? r0 * n1 : n2 * r2;
            double *fib = iscol ? acol : arow;
            if (iscol) for (int s = tid; s < r1; s += FB) xs[s] = Wq[(kk - 1) + (size_t)NM * (qq - 1) + P.SW * s];
            else       for (int s = tid; s < r1; s += FB) xs[s] = Cp[(ii - 1) + (size_t)RM * (jj - 1) + P.SS * s];
            crs++;
            if (iscol) havecol = 1; else haverow = 1;
            const int dn = (P.piv == 0) ? (h == 1) : (havecol && haverow && (crs >= 2 * P.piv));
            const bool resid = (P.piv != 0) && !dn;
            __syncthreads();
            double mx = 0.0, ab = -1.0, bb = 0.0; int ix = INT_MAX;
            for (int t = tid; t < nf; t += FB) {
                double a;
                if (iscol) {
                    const int i = t % r0, j = t / r0;
                    const double *rl = XL + (size_t)i * 2 * VS, *rq = XR + (size_t)(qq - 1) * 2 * VS;
                    a = f_ising_c4v(m, p - 1, rl, rl + VS, par[j], par[n1m + j], par[kk - 1], par[n1m + kk - 1], rq, rq + VS);
                    fib[t] = a;
                    if (resid) {
                        const double *c = Cp + i + (size_t)RM * j;
                        double b = a;
#pragma unroll 8
                        for (int s = 0; s < r1; s++) b = b + (-xs[s]) * c[P.SS * s];
                        const double aa = fabs(b);
                        if (aa > ab || (aa == ab && t < ix)) { ab = aa; bb = b; ix = t; }
                    }
                } else {
                    const int k = t % n2, q = t / n2;
                    const double *rl = XL + (size_t)(ii - 1) * 2 * VS, *rq = XR + (size_t)q * 2 * VS;
                    a = f_ising_c4v(m, p - 1, rl, rl + VS, par[jj - 1], par[n1m + jj - 1], par[k], par[n1m + k], rq, rq + VS);
                    fib[t] = a;
                    if (resid) {
                        const double *w = Wq + k + (size_t)NM * q;
                        double tt = 0.0;
#pragma unroll 8
                        for (int s = 0; s < r1; s++) tt = tt + w[P.SW * s] * xs[s];
                        const double b = a + (-1.0) * tt;
                        const double aa = fabs(b);
                        if (aa > ab || (aa == ab && t < ix)) { ab = aa; bb = b; ix = t; }
                    }
                }
                mx = fmax(mx, fabs(a));
            }
            mx = block_max(mx, sha);
            if (P.piv != 0) amax = fmax(amax, mx);           // the piv = 0 branch (:492-513) does not touch amax
            neval += nf;
            bytes_half += resid ? 8.0 * ((double)nf * r1 + r1 + 2.0 * nf) : 8.0 * nf;
            n_resid += resid ? 1 : 0;
            done = dn;
            if (resid) {
                fused_argmax(ab, bb, ix, sha, shv, shi);
                if (ix == INT_MAX) ix = 0;
                if (iscol) { const int i = ix % r0 + 1, j = ix / r0 + 1; done = havecol && haverow && (i == ii && j == jj); ii = i; jj = j; }
                else       { const int k = ix % n2 + 1, q = ix / n2 + 1; done = havecol && haverow && (k == kk && q == qq); kk = k; qq = q; }
                pivot = bb;
            }
        }
        __syncthreads();
        // ---- acceptance and in-place append (:598-758) ----
        int *tape = P.tape + ((size_t)g * (m + 2) + p) * 4;
        const bool upd = (fabs(pivot) > P.small_element * amax) && (fabs(pivot) > P.small_pivot * pivotmax_prev);
        if (!upd) {
            if (tid == 0) { tape[0] = tape[1] = tape[2] = tape[3] = -1; P.upd[(size_t)g * (m + 2) + p] = 0; }
        } else {
            const int i0 = ii - 1, j0 = jj - 1, k0 = kk - 1, q0 = qq - 1;
            double *gI = inv_ptr(P, g, p, first);
            // role E first part: packed LU from the OLD factors (:649-660)
            for (int s = tid; s < r1; s += FB) {
                gI[r1 * r1 + s] = Cp[i0 + (size_t)RM * j0 + P.SS * s];
                gI[r1 * r1 + r1 + s] = Wq[k0 + (size_t)NM * q0 + P.SW * s];
            }
            // role A: arg(p), col(p) new slab (:662-668, :701)
            for (int s = tid; s < r1; s += FB) xs[s] = Wq[k0 + (size_t)NM * q0 + P.SW * s];
            __syncthreads();
            for (int t = tid; t < r0 * n1; t += FB) {
                const int i = t % r0, j = t / r0; const size_t o = i + (size_t)RM * j;
                const double a = acol[t];
                Ap[o + P.SS * r1] = a;
                double y = a;
                for (int s = 0; s < r1; s++) y = y + (-xs[s]) * Cp[o + P.SS * s];
                Cp[o + P.SS * r1] = (1.0 / pivot) * y;
            }
            __syncthreads();
            // role B: arg(p+1), row(p+1) new row (:669-674, :702)
            for (int s = tid; s < r1; s += FB) xs[s] = Cp[i0 + (size_t)RM * j0 + P.SS * s];
            __syncthreads();
            for (int t = tid; t < n2 * r2; t += FB) {
                const int k = t % n2, q = t / n2;
                const double a = arow[t];
                Aq[r1 + (size_t)RM * k + P.SS * q] = a;
                const size_t o = k + (size_t)NM * q;
                double tt = 0.0;
                for (int s = 0; s < r1; s++) tt = tt + Wq[o + P.SW * s] * xs[s];
                Wq[o + P.SW * r1] = a + (-1.0) * tt;
            }
            // role C: row(p)(:, j, r1+1) = L(p-1)^-1 acol1(:, j)  (:715-728): one wave per column, shuffle wavefront
            if (p > first) {
                const double *gL = inv_ptr(P, g, p - 1, first);
                double *Wp = core_ptr(P, P.row, g, p, first);
                for (int j = wv; j < n1; j += FB / 64) {
                    const double a = (lane < r0) ? acol[lane + r0 * j] : 0.0;
                    double tmp = 0.0, xf = 0.0;
                    for (int s = 0; s < r0; s++) {
                        const double cand = (s == 0) ? a : a + (-1.0) * tmp;
                        const double xsv = __shfl(cand, s, 64);
                        if (lane == s) xf = xsv;
                        if (lane > s && lane < r0) tmp = tmp + xsv * gL[lane * lane + s];
                    }
                    if (lane < r0) Wp[j + (size_t)NM * r1 + P.SW * lane] = xf;
                }
            }
            // role D: col(p+1)(r1+1, k, :) = arow1(k, :) U(p+1)^-1  (:730-749)
            if (p < last) {
                const double *gU = inv_ptr(P, g, p + 1, first);
                double *Cq = core_ptr(P, P.col, g, p + 1, first);
                for (int k = wv; k < n2; k += FB / 64) {
                    double y = (lane < r2) ? arow[k + n2 * lane] : 0.0;
                    for (int s = 0; s < r2; s++) {
                        const double dg = gU[(s + 1) * (s + 1) - 1];
                        const double cand = (1.0 / dg) * y;
                        const double ys = __shfl(cand, s, 64);
                        if (lane == s) y = ys;
                        if (lane > s && lane < r2) y = y + (-gU[lane * lane + lane + s]) * ys;
                    }
                    if (lane < r2) Cq[r1 + (size_t)RM * k + P.SS * lane] = y;
                }
            }
            // role E: index tables, pivot set, scalars (:604-635)
            short *Ln = L_ptr(P, g, p, first), *Rn = R_ptr(P, g, p, first);
            for (int x = tid; x < p; x += FB) Ln[(size_t)x * RM + r1] = (x < p - 1) ? Lt[(size_t)x * RM + i0] : (short)(j0 + 1);
            for (int x = tid; x < m - p; x += FB) Rn[(size_t)x * RM + r1] = (x == 0) ? (short)(k0 + 1) : Rt[(size_t)(x - 1) * RM + q0];
            if (tid == 0) {
                gI[(r1 + 1) * (r1 + 1) - 1] = pivot;
                int *vq = vip_ptr(P, g, p, first) + 4 * r1;
                vq[0] = tape[0] = ii; vq[1] = tape[1] = jj; vq[2] = tape[2] = kk; vq[3] = tape[3] = qq;
                P.upd[(size_t)g * (m + 2) + p] = 1;
                r[p] = r1 + 1;                                                          // :752
            }
            const double ap = fabs(pivot);
            pivotmax = (pivotmax < 0.0) ? ap : fmax(pivotmax, ap);
            pivotmin = (pivotmin < 0.0) ? ap : fmin(pivotmin, ap);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) {
        gs.amax = amax; gs.pivotmax = pivotmax; gs.pivotmin = pivotmin; gs.neval = neval; gs.rngpos = rngpos;
        gs.bytes_half = bytes_half; gs.n_resid = n_resid;
    }
}
