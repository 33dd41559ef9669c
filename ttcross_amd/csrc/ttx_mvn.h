// ttx_mvn.h -- rook half-step and lottery of the multivariate-normal integrand (test_crs_mvn.f90:156-172,
// lib/mvn_pdf.f90:63-83) with one WAVE per varying pivot / per candidate.
//
// One evaluation is the quadratic form  ex = sum_i sum_j (d_i * S_ij) * d_j  accumulated in the reference's order (i outer,
// j inner): d^2 = 16 384 dependent fp64 additions at BASELINE config 4 (d = 128), each fed by two multiplies.  With one lane
// per element (k_halfstep) every term cost two LDS reads, a global read of S_ij, two multiplies and the add in a single
// lane: ~54 cycles per term, and a half-step lasted as long as one such chain (370 us).
//
// Here the lanes of a wave are the mode indices of ONE varying pivot, so the difference vector d is wave-uniform except for
// the one free dimension L.  For a row i /= L all terms with j /= L are the same for every lane: the 64 lanes compute
// 64 of them AT ONCE (T_j = (d_i S_ij) d_j, the reference's association), park them in LDS, and every lane then adds them
// to its own running sum in order as LDS broadcasts (1 add per term); the term j = L and the whole row i = L are taken
// per lane.  S is read one row ahead with coalesced loads.  Same operations, same order per lane: bit-identical.
#pragma once
#include "ttx_kernels.h"
#include "ttx_de.h"        // lds_sum_chain

#define MVN_MAXQ 8            // lanes hold ceil(m / 64) <= MVN_MAXQ entries of a row of S: m <= 512

// ex of lib/mvn_pdf.f90:74-80 for the elements of one wave.  dv[0..m): the difference vector in LDS, entry L is a
// placeholder; dL: this lane's difference in dimension L (L < 0: no free dimension, every lane gets the same sum).
// icT[j + m*i] = inv_cov(i, j).  tb: LDS scratch of 2*m doubles.
__device__ __forceinline__ double mvn_quadform_wave(int m, const double *dv, int L, double dL, const double *icT, double *tb, int lane)
{
    const int nq = (m + 63) >> 6;
    double sreg[MVN_MAXQ], dreg[MVN_MAXQ];             // this lane's entries of the current row of S and of d
#pragma unroll
    for (int q = 0; q < MVN_MAXQ; q++) { const int j = lane + 64 * q; dreg[q] = (q < nq && j < m) ? dv[j] : 0.0; sreg[q] = (q < nq && j < m) ? icT[j] : 0.0; }
    double ex = 0.0;
    for (int i = 0; i < m; i++) {
        double *t = tb + (size_t)(i & 1) * m;
        double snext[MVN_MAXQ];
        const double *nrow = icT + (size_t)m * (i + 1 < m ? i + 1 : i);
#pragma unroll
        for (int q = 0; q < MVN_MAXQ; q++) { const int j = lane + 64 * q; snext[q] = (q < nq && j < m) ? nrow[j] : 0.0; }    // next row of S, one row ahead
        if (i != L) {
            const double di = dv[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < MVN_MAXQ; q++) { const int j = lane + 64 * q; if (q < nq && j < m) t[j] = di * sreg[q] * dreg[q]; }   // (d_i S_ij) d_j
            __builtin_amdgcn_wave_barrier();
            if (L < 0) ex = lds_sum_chain(ex, t, m);
            else {
                ex = lds_sum_chain(ex, t, L);
                // the term with the free dimension: S_iL is uniform, the difference is this lane's
                const double siL = icT[(size_t)m * i + L];
                ex = ex + di * siL * dL;
                ex = lds_sum_chain(ex, t + L + 1, m - L - 1);
            }
        } else {
            // the row of the free dimension: d_L S_Lj d_j per lane, S_Lj and d_j uniform (j /= L) -- through LDS as well
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < MVN_MAXQ; q++) { const int j = lane + 64 * q; if (q < nq && j < m) { t[j] = sreg[q]; } }
            __builtin_amdgcn_wave_barrier();
            for (int j = 0; j < m; j++) {
                const double dj = (j == L) ? dL : dv[j];
                ex = ex + dL * t[j] * dj;
            }
        }
#pragma unroll
        for (int q = 0; q < MVN_MAXQ; q++) sreg[q] = snext[q];
    }
    return ex;
}

// The same for m a multiple of 64 WITHOUT LDS (round 2, later).  Stamps inside mvn_quadform_wave: 0.29 us per row for the S loads
// and the 128 terms, 0.9-1.1 us for adding them -- 7-8.4 ns per term although a dependent v_add_f64 takes 2.4 ns: the LDS
// broadcast reads are the cost (a ds_read2_b64 returns 1 KB to the wave and occupies the LDS return path ~32 cycles).  The terms
// are computed one per lane anyway (lane l holds T_{64q+l}); here they STAY in that register and every add takes its operand
// with two v_readlane_b32 into an SGPR pair (independent of the running sum) and a v_add_f64 with a scalar operand: three
// VALU instructions per term, no LDS traffic, no barrier.  The per-lane term of the free dimension replaces lane L's at its place
// in the order by a uniform select.  Operations and their order per lane are unchanged: bit-identical.
__device__ __forceinline__ double mvn_lane(double x, int k)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, k), hi = __builtin_amdgcn_readlane((int)(b >> 32), k);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sixteen lane values at a time into scalar registers first (so that the reads run ahead of the dependent adds), then the adds.
// The term of lane kL is replaced by `special`: the batch of sixteen that contains it (BL, a template parameter: one straight
// line of code per position, chosen by ONE wave-uniform branch per row) pays for the selects; a branch per batch instead kept the
// lane reads of a batch from running under the adds of the batch before (measured: 0.4 us per row of 128 terms).
template <int BL>
__device__ __forceinline__ double mvn_add64_at(double ex, double t, int kLL, double special)
{
#pragma unroll
    for (int b = 0; b < 64; b += 16) {
        double sv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) sv[k] = mvn_lane(t, b + k);
#pragma unroll
        for (int k = 0; k < 16; k++) ex = ex + ((b == 16 * BL && k == kLL) ? special : sv[k]);
    }
    return ex;
}
__device__ __forceinline__ double mvn_add64(double ex, double t, int kL, double special)
{
    switch (kL >> 4) {                                      // wave-uniform
        case 0: return mvn_add64_at<0>(ex, t, kL & 15, special);
        case 1: return mvn_add64_at<1>(ex, t, kL & 15, special);
        case 2: return mvn_add64_at<2>(ex, t, kL & 15, special);
        default: return mvn_add64_at<3>(ex, t, kL & 15, special);
    }
}
// the same without a replaced lane: no branch between the batches, so the lane reads of a batch run under the adds of the one before
__device__ __forceinline__ double mvn_add64_plain(double ex, double t)
{
#pragma unroll
    for (int b = 0; b < 64; b += 16) {
        double sv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) sv[k] = mvn_lane(t, b + k);
#pragma unroll
        for (int k = 0; k < 16; k++) ex = ex + sv[k];
    }
    return ex;
}
// NQ = m / 64 as a template parameter: the body is unrolled over the NQ registers a lane holds of a row, and unrolled for the
// largest possible count the kernel outgrew the instruction cache (measured: 2x slower half-steps)
template <int NQ>
__device__ __forceinline__ double mvn_quadform_lanes(int m, const double *dv, int L, double dL, const double *icT, int lane)
{
    double dreg[NQ], s0[NQ], s1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int j = lane + 64 * q;
        dreg[q] = dv[j];
        s0[q] = icT[j];
        s1[q] = icT[(size_t)m * (m > 1 ? 1 : 0) + j];
    }
    const int qL = L >> 6, kL = L & 63;
    double ex = 0.0;
    for (int i = 0; i < m; i++) {
        double s2[NQ];
        const double *row2 = icT + (size_t)m * (i + 2 < m ? i + 2 : m - 1);
#pragma unroll
        for (int q = 0; q < NQ; q++) s2[q] = row2[lane + 64 * q];              // S two rows ahead
        if (i != L) {
            const double di = dv[i];
            double special = 0.0;
            if (L >= 0) {                              // S_iL sits in lane kL of this row's registers
                double siL = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; q++) if (q == qL) siL = mvn_lane(s0[q], kL);
                special = di * siL * dL;               // the free dimension: this lane's own difference
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const double t = di * s0[q] * dreg[q];                        // (d_i S_ij) d_j, j = 64 q + lane
                if (L >= 0 && q == qL) ex = mvn_add64(ex, t, kL, special);    // wave-uniform
                else ex = mvn_add64_plain(ex, t);
            }
        } else {
            // the row of the free dimension: (d_L S_Lj) d_j per lane, S_Lj and d_j (j /= L) lane by lane (once per element)
            for (int j = 0; j < m; j++) {
                double sj = 0.0, dj = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; q++) if (q == (j >> 6)) { sj = mvn_lane(s0[q], j & 63); dj = mvn_lane(dreg[q], j & 63); }
                if (j == L) dj = dL;
                ex = ex + dL * sj * dj;
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) { s0[q] = s1[q]; s1[q] = s2[q]; }
    }
    return ex;
}
__device__ __forceinline__ double mvn_quadform(int m, const double *dv, int L, double dL, const double *icT, double *tb, int lane)
{
    if ((m & 63) == 0) {
        switch (m >> 6) {
            case 1: return mvn_quadform_lanes<1>(m, dv, L, dL, icT, lane);
            case 2: return mvn_quadform_lanes<2>(m, dv, L, dL, icT, lane);
            case 3: return mvn_quadform_lanes<3>(m, dv, L, dL, icT, lane);
            case 4: return mvn_quadform_lanes<4>(m, dv, L, dL, icT, lane);
            default: break;
        }
    }
    return mvn_quadform_wave(m, dv, L, dL, icT, tb, lane);
}

// grid = (RM * ceil(NM/64) wave slots, groups), 64 threads.  Same contract as k_halfstep (modes 0, 1, 2).
__global__ __launch_bounds__(64) void k_halfstep_mvn(DevProb P, int h, int dir, int mode)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ StepState cur;
    const int g = blockIdx.y, lane = threadIdx.x, m = P.d;
    GroupState &gs = P.gs[g];
    if (lane == 0) { cur = gs.S[h]; resolve_state(cur, gs.Pt[(h + 1) & 1]); }
    __syncthreads();
    if (!cur.active || cur.done) { if (blockIdx.x == 0 && lane == 0) gs.S[h + 1] = cur; return; }
    const bool iscol = (mode == 1 || mode == 2) ? (h == 0) : (((h + (dir == 2 ? 1 : 0)) & 1) == 0);    // :517,550
    const int p = cur.p, r0 = cur.r0, r1 = cur.r1, r2 = cur.r2, n1 = cur.n1, n2 = cur.n2, first = gs.first;
    const int nf = iscol ? r0 * n1 : n2 * r2;
    const int nv = iscol ? r0 : r2, nm = iscol ? n1 : n2, nch = (nm + 63) >> 6;
    const int npart = nv * nch;
    const int w = blockIdx.x;
    const int crs = cur.crs + 1;
    const int havecol = cur.havecol | (iscol ? 1 : 0), haverow = cur.haverow | (iscol ? 0 : 1);
    const int done = (mode == 1 || mode == 2) ? (h == 1) : (havecol && haverow && (crs >= 2 * P.piv));   // :534 / :567
    const bool resid = (mode == 0) && !done;
    if (w == 0 && lane == 0) {
        StepState nx = cur;
        nx.crs = crs; nx.havecol = havecol; nx.haverow = haverow; nx.done = done;
        nx.pending = resid ? (iscol ? 1 : 2) : 0;
        nx.npart = npart;
        gs.S[h + 1] = nx;
        if (mode != 2) gs.neval += nf;                                        // :527 / :560 / :509
        gs.bytes_half += resid ? 8.0 * ((double)nf * r1 + r1 + 2.0 * nf) : 8.0 * nf;
        gs.n_resid += resid ? 1 : 0;
    }
    if (w >= npart) return;
    const int pv = w / nch, vmode = (w - pv * nch) * 64 + lane;
    const bool live = vmode < nm;
    const int A = p - 1, B = m - p - 1;
    const int pl = iscol ? pv : cur.ii - 1, qr = iscol ? cur.qq - 1 : pv;
    const double *nodes = P.par, *mu = P.aux;
    double *dv = dyn, *tb = dyn + ((m + 1) & ~1);
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    const int L = iscol ? A : A + 1;                                   // the free dimension (0-based)
    for (int x = lane; x < m; x += 64) {
        int ix;
        if (x < A) ix = Lt[(size_t)x * P.RM + pl] - 1;
        else if (x == A) ix = iscol ? 0 : cur.jj - 1;
        else if (x == A + 1) ix = iscol ? cur.kk - 1 : 0;
        else ix = Rt[(size_t)(x - A - 2) * P.RM + qr] - 1;
        dv[x] = nodes[ix] - mu[x];
    }
    const double dL = nodes[live ? vmode : 0] - mu[L];
    __syncthreads();
    const double ex = mvn_quadform(m, dv, L, dL, P.auxT, tb, lane);
    const double a = ttx_exp(-0.5 * ex) / P.mvn_norm;                  // lib/mvn_pdf.f90:82
    // ---- fiber store, amax, residual, arg-max: as k_halfstep, on the fiber's linear index t ----
    const int u_ = iscol ? pv : vmode, v_ = iscol ? vmode : pv;
    const int t = iscol ? (u_ + r0 * v_) : (u_ + n2 * v_);
    if (live) (iscol ? P.acol : P.arow)[(size_t)g * P.RM * P.NM + t] = a;
    const double mx = wave_max(live ? fabs(a) : 0.0);
    if (lane == 0 && mode != 1) atomic_max_pos(&gs.amax, mx);
    if (resid) {
        const double *Cp = core_ptr(P, P.col, g, p, first), *Wq = core_ptr(P, P.row, g, p + 1, first);
        double bb = a, ab = -1.0; int bi = INT_MAX;
        if (live) {
            if (iscol) {
                const double *c = Cp + u_ + (size_t)P.RM * v_;
                const double *xq = Wq + (cur.kk - 1) + (size_t)P.NM * (cur.qq - 1);
#pragma unroll 8
                for (int s = 0; s < r1; s++) bb = bb + (-xq[P.SW * s]) * c[P.SS * s];
            } else {
                const double *wv = Wq + u_ + (size_t)P.NM * v_;
                const double *xc = Cp + (cur.ii - 1) + (size_t)P.RM * (cur.jj - 1);
                double tt = 0.0;
#pragma unroll 8
                for (int s = 0; s < r1; s++) tt = tt + wv[P.SW * s] * xc[P.SS * s];
                bb = bb + (-1.0) * tt;
            }
            ab = fabs(bb); bi = t;
        }
        wave_argmax(ab, bb, bi);
        if (lane == 0) { Partial pr; pr.absmax = ab; pr.val = bb; pr.idx = bi; pr.pad = 0; gs.Pt[h & 1][w] = pr; }
    }
}

// lottery candidates of the mvn integrand, one per wave (k_lottery phases 1 / 2 around it, as for Ising D/E)
__global__ __launch_bounds__(64) void k_lottery_eval_mvn(DevProb P)
{
    extern __shared__ __align__(16) double dyn[];
    const int g = blockIdx.y, il = blockIdx.x, lane = threadIdx.x, m = P.d;
    const GroupState &gs = P.gs[g];
    const StepState &st = gs.S[0];
    if (!st.active) return;
    const int p = st.p, first = gs.first;
    const int nlot = st.r0 + st.n1 + st.n2 + st.r2;
    if (il >= nlot) return;
    const int *cand = P.lotc + ((size_t)g * P.lot_max + il) * 4;
    const int ci = cand[0] - 1, cj = cand[1] - 1, ck = cand[2] - 1, cq = cand[3] - 1;
    const int A = p - 1;
    double *dv = dyn, *tb = dyn + ((m + 1) & ~1);
    const double *nodes = P.par, *mu = P.aux;
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = lane; x < m; x += 64) {
        const int ix = (x < A) ? Lt[(size_t)x * P.RM + ci] - 1 : (x == A) ? cj : (x == A + 1) ? ck : Rt[(size_t)(x - A - 2) * P.RM + cq] - 1;
        dv[x] = nodes[ix] - mu[x];
    }
    __syncthreads();
    const double ex = mvn_quadform(m, dv, -1, 0.0, P.auxT, tb, lane);
    if (lane == 0) P.lotf[(size_t)g * P.lot_max + il] = ttx_exp(-0.5 * ex) / P.mvn_norm;
}
