// ttx_de.h -- rook half-step for the Ising D / E integrands (test_crs_ising.f90:186-195): one WAVE per varying pivot.
//
// One evaluation of D/E is a chain of m(m+1)/2 pair factors multiplied in the reference's order -- 32 640 dependent
// fp64 multiplies at BASELINE config 5 (D_256) -- so a half-step lasts as long as one wave needs for its chain, and
// what counts is the number of instructions a wave issues per pair (fp64 VALU: one instruction per 4 cycles per wave).
// k_halfstep gave every lane its own pivot: the tabulated factors came from per-lane global loads (8 in flight:
// latency-bound, ~55 cycles per pair) and the node values of the bond-spanning pairs from per-lane LDS lookups.
//
// Here the 64 lanes of a wave are the MODE indices of one varying pivot (column half-step: left pivot pv, lanes = j;
// row half-step: right pivot pv, lanes = k), so everything except the one free mode index is wave-uniform:
//   * tabulated factors (TL of the left pivot, TR of the right pivot; k_de_tables lays a pivot's factors out in
//     multiplication order) are STREAMED: the wave loads 64 consecutive factors with one coalesced 512-byte load,
//     two batches ahead, parks them in a 64-entry LDS ring and multiplies them in as LDS broadcasts;
//   * the node values of the left / right dims and the running products UL are staged once per wave as LDS arrays
//     (broadcast reads, no index decoding);
//   * the IEEE divisions of the bond-spanning pairs use the exact short sequence fdiv_unit when the host has verified
//     that all nodes lie in [0,1] (P.de_unit), four instructions fewer than the general a/b;
// The arithmetic per element -- every product, difference, quotient and their order -- is that of f_ising_de /
// de_pairs_tab, hence of the oracle: results stay bit-identical.
#pragma once
#include "ttx_kernels.h"

// a = (..((a OP p[0]) OP p[1]) ..) OP p[len-1] for an LDS row read with a wave-uniform address (broadcasts), OP = * or +.
// A lone wave has nobody to hide the LDS latency behind (~130 cycles for four 16-byte broadcast reads), so the loop is
// software-pipelined by hand in batches of NB = 32: the reads of the NEXT batch are issued before the 32 dependent
// operations (~200 cycles) of the current one.  (With batches of 8 the compiler's s_waitcnt at the loop header -- forced by
// the address register doubling as a load destination -- exposed the full latency every 8 factors: 20+ cycles per factor.)
template <bool MUL, int NB>
__device__ __forceinline__ double lds_fold(double a, const double *p, int len)
{
    int c = 0;
    if (len >= NB) {
        double x[NB], y[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) x[q] = p[q];
        c = NB;
        for (;;) {                                   // two stages per trip, so that no register copies are needed
            if (c + NB > len) {
#pragma unroll
                for (int q = 0; q < NB; q++) a = MUL ? a * x[q] : a + x[q];
                break;
            }
#pragma unroll
            for (int q = 0; q < NB; q++) y[q] = p[c + q];
#pragma unroll
            for (int q = 0; q < NB; q++) a = MUL ? a * x[q] : a + x[q];
            c += NB;
            if (c + NB > len) {
#pragma unroll
                for (int q = 0; q < NB; q++) a = MUL ? a * y[q] : a + y[q];
                break;
            }
#pragma unroll
            for (int q = 0; q < NB; q++) x[q] = p[c + q];
#pragma unroll
            for (int q = 0; q < NB; q++) a = MUL ? a * y[q] : a + y[q];
            c += NB;
        }
    }
    if (len - c >= 8) {                              // tail: one batch of 8 at a time, then singles
        for (; c + 8 <= len; c += 8) {
            double z[8];
#pragma unroll
            for (int q = 0; q < 8; q++) z[q] = p[c + q];
#pragma unroll
            for (int q = 0; q < 8; q++) a = MUL ? a * z[q] : a + z[q];
        }
    }
    for (; c < len; c++) a = MUL ? a * p[c] : a + p[c];
    return a;
}
__device__ __forceinline__ double lds_chain(double a, const double *p, int len) { return lds_fold<true, 32>(a, p, len); }
__device__ __forceinline__ double lds_sum_chain(double s, const double *p, int len) { return lds_fold<false, 32>(s, p, len); }

// wave-uniform stream of doubles g[0..total) consumed in order by all lanes of ONE wave
struct WStream {
    const double *g; double *buf; double rA, rB; int total, nextb, avail, rd;
    __device__ __forceinline__ double ld(int b, int lane) const { const int ix = b * 64 + lane; return ix < total ? g[ix] : 1.0; }
    __device__ __forceinline__ void init(const double *g_, int total_, double *buf_, int lane)
    { g = g_; total = total_; buf = buf_; rA = ld(0, lane); rB = ld(1, lane); nextb = 2; avail = 0; rd = 0; }
    __device__ __forceinline__ void refill(int lane)
    {
        __builtin_amdgcn_wave_barrier();
        buf[lane] = rA; rA = rB; rB = ld(nextb, lane); nextb++; avail = 64; rd = 0;
        __builtin_amdgcn_wave_barrier();
    }
    // a = (..((a * g[pos]) * g[pos+1]) ..) over the next cnt entries
    __device__ __forceinline__ double chain(double a, int cnt, int lane)
    {
        while (cnt > 0) {
            if (avail == 0) refill(lane);
            const int mm = cnt < avail ? cnt : avail;
            a = lds_chain(a, buf + rd, mm);
            rd += mm; avail -= mm; cnt -= mm;
        }
        return a;
    }
};

// the bond-spanning tail of a row: pair with s2 (x2), then with the right dims xr[0..B) (LDS, wave-uniform)
template <bool FAST>
__device__ __forceinline__ void de_run(double &a, double u, double x2, const double *xr, int B)
{
    u = u * x2; a = a * de_t2<FAST>(u);
    int j = 0;
    if (B >= 4) {
        double y0 = xr[0], y1 = xr[1], y2 = xr[2], y3 = xr[3];
        for (; j + 4 <= B; j += 4) {                   // four independent divisions in flight; products in order
            const double x0 = y0, x1_ = y1, x2_ = y2, x3 = y3;
            if (j + 8 <= B) { y0 = xr[j + 4]; y1 = xr[j + 5]; y2 = xr[j + 6]; y3 = xr[j + 7]; }    // next batch's LDS reads fly under the divisions
            const double u1 = u * x0, u2 = u1 * x1_, u3 = u2 * x2_, u4 = u3 * x3;
            const double t1 = de_t2<FAST>(u1), t2 = de_t2<FAST>(u2), t3 = de_t2<FAST>(u3), t4 = de_t2<FAST>(u4);
            a = a * t1; a = a * t2; a = a * t3; a = a * t4;
            u = u4;
        }
    }
    for (; j < B; j++) { u = u * xr[j]; a = a * de_t2<FAST>(u); }
}

// grid = (RM * ceil(NM/64) wave slots, groups), 64 threads.  Same contract as k_halfstep (modes 0, 1, 2).
template <bool FAST>
__global__ __launch_bounds__(64) void k_halfstep_de(DevProb P, int h, int dir, int mode)
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
    const int pv = w / nch, vmode = (w - pv * nch) * 64 + lane;        // varying pivot, mode index (0-based)
    const bool live = vmode < nm;
    const int A = p - 1, B = m - p - 1;
    const int pl = iscol ? pv : cur.ii - 1, qr = iscol ? cur.qq - 1 : pv;         // left / right pivot of this wave
    const int n1m = P.n[1];
    const double *nodes = P.par, *weights = P.par + n1m;                          // 0-based here
    // LDS: UL[VS] | xl[VS] | wl[VS] | xr[VS] | wr[VS] | ring L[64] | ring R[64]
    const int VS = ((m + 7) & ~7) + 8;
    double *UL = dyn, *xl = UL + VS, *wl = xl + VS, *xr = wl + VS, *wr = xr + VS, *ringL = wr + VS, *ringR = ringL + 64;
    const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
    const double *TLg = P.deTL + (size_t)g * tsz + (size_t)pl * NP, *TRg = P.deTR + (size_t)g * tsz + (size_t)qr * NP;
    const double *ULg = P.deUL + ((size_t)g * P.RM + pl) * (m + 1);
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = lane; x < A; x += 64) { const int ix = Lt[(size_t)x * P.RM + pl] - 1; xl[x] = nodes[ix]; wl[x] = weights[ix]; }
    for (int x = lane; x < B; x += 64) { const int ix = Rt[(size_t)x * P.RM + qr] - 1; xr[x] = nodes[ix]; wr[x] = weights[ix]; }
    for (int x = lane; x <= A; x += 64) UL[x] = ULg[x];
    const int i1 = iscol ? (live ? vmode : 0) : cur.jj - 1, i2 = iscol ? cur.kk - 1 : (live ? vmode : 0);   // node index of dim p / p+1
    const double x1 = nodes[i1], x2 = nodes[i2], w1 = weights[i1], w2 = weights[i2];
    WStream sl, sr;
    sl.init(TLg, A * (A + 1) / 2, ringL, lane);
    sr.init(TRg, B * (B + 1) / 2, ringR, lane);
    __syncthreads();
    // ---- pair product (test_crs_ising.f90:186-195), order of de_pairs_tab ----
    double a = 1.0;
    for (int i = 0; i <= A; i++) {
        a = sl.chain(a, A - i, lane);
        double u = UL[i];
        u = u * x1; a = a * de_t2<FAST>(u);
        de_run<FAST>(a, u, x2, xr, B);
    }
    de_run<FAST>(a, 1.0, x2, xr, B);                                   // i = A+1: starts after dim p
    a = sr.chain(a, B * (B + 1) / 2, lane);
    // ---- b-part (id 2) and the weights (:197-218), order of de_finish ----
    const int id = P.ising_id;
    double b = 0.0;
    if (id == 2) {
        double v = 1.0, ww = 1.0, vk = 1.0, wk = 1.0;
        for (int j = B - 1; j >= 0; j--) { vk = vk * xr[j]; v = v + vk; }
        vk = vk * x2; v = v + vk;
        vk = vk * x1; v = v + vk;
        for (int j = A - 1; j >= 0; j--) { vk = vk * xl[j]; v = v + vk; }
        for (int j = 0; j < A; j++) { wk = wk * xl[j]; ww = ww + wk; }
        wk = wk * x1; ww = ww + wk;
        wk = wk * x2; ww = ww + wk;
        for (int j = 0; j < B; j++) { wk = wk * xr[j]; ww = ww + wk; }
        b = 1.0 / (v * ww);
    }
    double f = (id == 2) ? 2 * a * b : 2 * a;
    for (int j = 0; j < A; j++) f = f * wl[j];
    f = f * w1; f = f * w2;
    for (int j = 0; j < B; j++) f = f * wr[j];
    a = f;
    // ---- fiber store, amax, residual, arg-max: as k_halfstep, on the fiber's linear index t ----
    const int u_ = iscol ? pv : vmode, v_ = iscol ? vmode : pv;        // col: (i, j) ; row: (k, q), 0-based
    const int t = iscol ? (u_ + r0 * v_) : (u_ + n2 * v_);
    if (live) (iscol ? P.acol : P.arow)[(size_t)g * P.RM * P.NM + t] = a;
    const double mx = wave_max(live ? fabs(a) : 0.0);
    if (lane == 0 && mode != 1) atomic_max_pos(&gs.amax, mx);          // :531 / :564 (the piv = 0 branch :492-513 does not touch amax)
    if (resid) {
        const double *Cp = core_ptr(P, P.col, g, p, first), *Wq = core_ptr(P, P.row, g, p + 1, first);
        double bb = a, ab = -1.0; int bi = INT_MAX;
        if (live) {
            if (iscol) {   // dgemv 'n', alpha=-1 (:538): b += (-x_s) * col(:, s), x_s = row(p+1)(s, kk, qq)
                const double *c = Cp + u_ + (size_t)P.RM * v_;
                const double *xq = Wq + (cur.kk - 1) + (size_t)P.NM * (cur.qq - 1);
#pragma unroll 8
                for (int s = 0; s < r1; s++) bb = bb + (-xq[P.SW * s]) * c[P.SS * s];
            } else {       // dgemv 't', alpha=-1 (:571): b += -1 * sum_s row(s, kq) * x_s, x_s = col(p)(ii, jj, s)
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

// ------------------------------------------------------------------------------------------------------------------
// One element per WAVE (lottery candidates, boundary corners): the chip holds few of these evaluations at a time, and a
// lone lane needs ~32 640 x 20 cycles for one.  Here the 64 lanes share ONE element, so every operand is wave-uniform:
//   * division phase: lane r of a tile owns ROW i0+r of the pair triangle (rows are independent: each has its own
//     running product u) and writes its factors ((u-1)/(u+1))^2 to an LDS tile T2[r][.] -- RT rows per tile;
//   * product phase: all lanes multiply the tile's factors into `a` in the reference's order (row by row, LDS
//     broadcasts), with the tabulated factors of the pivots streamed in between (WStream) when tables exist.
// The value of `a` -- and everything after it -- is computed redundantly and identically by all lanes.
// ------------------------------------------------------------------------------------------------------------------
#define DE_RT 16                 // rows of pair factors per LDS tile
__host__ __device__ inline size_t de_wave_lds_doubles(int m) { const int VS = ((m + 7) & ~7) + 8; return (size_t)3 * VS + 128 + (size_t)DE_RT * (VS + 1); }

// b-part (id 2) and weights (test_crs_ising.f90:197-218) from per-dimension value arrays xv / wv (0-based dims)
__device__ __forceinline__ double de_finish_vals(int id, double a, int m, const double *xv, const double *wv)
{
    double b = 0.0;
    if (id == 2) {
        double v = 1.0, w = 1.0, vk = 1.0, wk = 1.0;
        for (int j = m - 1; j >= 0; j--) { vk = vk * xv[j]; v = v + vk; }
        for (int j = 0; j < m; j++) { wk = wk * xv[j]; w = w + wk; }
        b = 1.0 / (v * w);
    }
    double f = (id == 2) ? 2 * a * b : 2 * a;
    for (int j = 0; j < m; j++) f = f * wv[j];
    return f;
}

__device__ __forceinline__ double de_row_chain(double a, const double *row, int len) { return lds_chain(a, row, len); }

// pair product of ONE element without tables: every pair (i, j), 0 <= i < j <= m, by division.  xv[0..m) in LDS.
template <bool FAST>
__device__ __forceinline__ double de_elem_full(int m, const double *xv, double *T2, int lane)
{
    const int RS = ((m + 7) & ~7) + 9;                 // odd row stride in doubles: 16 rows hit 16 different bank pairs
    double a = 1.0;
    for (int i0 = 0; i0 < m; i0 += DE_RT) {
        const int nr = min(DE_RT, m - i0), maxlen = m - i0;
        __builtin_amdgcn_wave_barrier();
        if (lane < nr) {
            const int i = i0 + lane, len = m - i;      // row i: pairs (i, i+1..m), factors of dims i+1..m = xv[i..m)
            double u = 1.0;
            double *row = T2 + (size_t)lane * RS;
            for (int c = 0; c < maxlen; c++) {
                if (c < len) { u = u * xv[i + c]; row[c] = de_t2<FAST>(u); }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int r = 0; r < nr; r++) a = de_row_chain(a, T2 + (size_t)r * RS, m - (i0 + r));
    }
    return a;
}

// the same with the pivots' tables: rows 0..A+1 carry only the bond-spanning pairs (B+2 per row; row A+1 starts after
// dim p, its first slot holds the neutral 1.0), TL factors are streamed before each row, TR factors after the last
template <bool FAST>
__device__ __forceinline__ double de_elem_tab(int m, int A, const double *xv, const double *UL, WStream &sl, WStream &sr, double *T2, int lane)
{
    const int RS = ((m + 7) & ~7) + 9;
    const int B = m - A - 2, nrow = A + 2, len = B + 2;
    const double x1 = xv[A], x2 = xv[A + 1];
    const double *xr = xv + A + 2;
    double a = 1.0;
    for (int i0 = 0; i0 < nrow; i0 += DE_RT) {
        const int nr = min(DE_RT, nrow - i0);
        __builtin_amdgcn_wave_barrier();
        if (lane < nr) {
            const int i = i0 + lane;
            double *row = T2 + (size_t)lane * RS;
            double u = 1.0;
            if (i <= A) { u = UL[i] * x1; row[0] = de_t2<FAST>(u); } else row[0] = 1.0;
            u = u * x2; row[1] = de_t2<FAST>(u);
            int j = 0;
            for (; j + 4 <= B; j += 4) {
                const double u1 = u * xr[j], u2 = u1 * xr[j + 1], u3 = u2 * xr[j + 2], u4 = u3 * xr[j + 3];
                row[2 + j] = de_t2<FAST>(u1); row[3 + j] = de_t2<FAST>(u2); row[4 + j] = de_t2<FAST>(u3); row[5 + j] = de_t2<FAST>(u4);
                u = u4;
            }
            for (; j < B; j++) { u = u * xr[j]; row[2 + j] = de_t2<FAST>(u); }
        }
        __builtin_amdgcn_wave_barrier();
        for (int r = 0; r < nr; r++) {
            const int i = i0 + r;
            if (i <= A) a = sl.chain(a, A - i, lane);
            a = de_row_chain(a, T2 + (size_t)r * RS, len);
        }
    }
    return sr.chain(a, B * (B + 1) / 2, lane);
}

// lottery candidates (lib/dmrgg.f90:455-463) of the Ising D / E integrands: one candidate per wave.  grid = (nlot, groups).
// The candidates were drawn by k_lottery (phase 1) into P.lotc; values go to P.lotf for k_lottery (phase 2).
template <bool FAST>
__global__ __launch_bounds__(64) void k_lottery_eval_de(DevProb P)
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
    const int A = p - 1, B = m - p - 1, VS = ((m + 7) & ~7) + 8, n1m = P.n[1];
    double *xv = dyn, *wv = xv + VS, *UL = wv + VS, *ringL = UL + VS, *ringR = ringL + 64, *T2 = ringR + 64;
    const double *nodes = P.par, *weights = P.par + n1m;
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = lane; x < m; x += 64) {
        const int ix = (x < A) ? Lt[(size_t)x * P.RM + ci] - 1 : (x == A) ? cj : (x == A + 1) ? ck : Rt[(size_t)(x - A - 2) * P.RM + cq] - 1;
        xv[x] = nodes[ix]; wv[x] = weights[ix];
    }
    const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
    const double *ULg = P.deUL + ((size_t)g * P.RM + ci) * (m + 1);
    for (int x = lane; x <= A; x += 64) UL[x] = ULg[x];
    WStream sl, sr;
    sl.init(P.deTL + (size_t)g * tsz + (size_t)ci * NP, A * (A + 1) / 2, ringL, lane);
    sr.init(P.deTR + (size_t)g * tsz + (size_t)cq * NP, B * (B + 1) / 2, ringR, lane);
    __syncthreads();
    const double a = de_elem_tab<FAST>(m, A, xv, UL, sl, sr, T2, lane);
    const double f = de_finish_vals(P.ising_id, a, m, xv, wv);
    if (lane == 0) P.lotf[(size_t)g * P.lot_max + il] = f;
}

// ------------------------------------------------------------------------------------------------------------------
// k_halfstep_de4: the same half-step with the work of ONE (pivot, mode-chunk) spread over the four SIMDs of a CU.
// Measured unit costs (profiles/r02_fold_probe.txt): a bond-spanning pair with its exact division costs 39 ns of one wave's
// issue slots, an ordered multiply of a ready factor 2.6-5.8 ns -- and a launch holds ~1.25 waves per CU, three SIMDs idle.
// So a workgroup of FOUR waves shares the 64 elements (lane = mode index in every wave):
//   waves 1..3, the DIVIDERS: each runs the cheap running product u of the row (one multiply per pair) and performs the
//     divisions of every third pair -- twelve pairs per step, four independent divisions per divider -- writing
//     ((u-1)/(u+1))^2 into an LDS tile T2[pair][lane];
//   wave 0, the FOLDER: streams the tabulated factors (TL / TR) and multiplies the tiles' factors into the products `a`
//     in the reference's order; it alone owns `a`, the b-part, the weights, the residual and the arg-max record.
// One workgroup barrier per tile of 36 pairs; the tiles are double-buffered, so the dividers work on tile k+1 while the
// folder folds tile k.  Every product is still taken by one lane in the reference's order: bit-identical.
// ------------------------------------------------------------------------------------------------------------------
#define DE4_TJ 36                // pairs per tile (three steps of twelve)
__host__ __device__ inline size_t de4_lds_doubles(int m) { const int VS = ((m + 7) & ~7) + 8; return (size_t)5 * VS + 128 + (size_t)2 * DE4_TJ * 64; }

template <bool FAST>
__global__ __launch_bounds__(256) void k_halfstep_de4(DevProb P, int h, int dir, int mode)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ StepState cur;
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = P.d;
    GroupState &gs = P.gs[g];
    if (tid == 0) { cur = gs.S[h]; resolve_state(cur, gs.Pt[(h + 1) & 1]); }
    __syncthreads();
    if (!cur.active || cur.done) { if (blockIdx.x == 0 && tid == 0) gs.S[h + 1] = cur; return; }
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
    if (w == 0 && tid == 0) {
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
    const int n1m = P.n[1];
    const double *nodes = P.par, *weights = P.par + n1m;
    const int VS = ((m + 7) & ~7) + 8;
    double *UL = dyn, *xl = UL + VS, *wl = xl + VS, *xr = wl + VS, *wr = xr + VS, *ringL = wr + VS, *ringR = ringL + 64, *T2 = ringR + 64;
    const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
    const double *TLg = P.deTL + (size_t)g * tsz + (size_t)pl * NP, *TRg = P.deTR + (size_t)g * tsz + (size_t)qr * NP;
    const double *ULg = P.deUL + ((size_t)g * P.RM + pl) * (m + 1);
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = tid; x < A; x += 256) { const int ix = Lt[(size_t)x * P.RM + pl] - 1; xl[x] = nodes[ix]; wl[x] = weights[ix]; }
    for (int x = tid; x < B; x += 256) { const int ix = Rt[(size_t)x * P.RM + qr] - 1; xr[x] = nodes[ix]; wr[x] = weights[ix]; }
    for (int x = tid; x <= A; x += 256) UL[x] = ULg[x];
    const int i1 = iscol ? (live ? vmode : 0) : cur.jj - 1, i2 = iscol ? cur.kk - 1 : (live ? vmode : 0);
    const double x1 = nodes[i1], x2 = nodes[i2], w1 = weights[i1], w2 = weights[i2];
    __syncthreads();
    // The spanning pairs of row i (0 <= i <= A+1) as ONE sequence of factors applied to u0: [x1 (rows <= A only), x2, xr[0..B)].
    // Positions are numbered c = 0 .. len-1 with len = B+2 (rows <= A) or B+1 (row A+1); tile t of a row holds c in
    // [TJ t, TJ t + TJ).  Everybody walks rows and tiles in the same order and meets at ONE barrier per tile.
    double a = 1.0;
    WStream sl, sr;
    if (wv == 0) { sl.init(TLg, A * (A + 1) / 2, ringL, lane); sr.init(TRg, B * (B + 1) / 2, ringR, lane); }
    int gt = 0;                                              // global tile counter (selects the buffer)
    int prev_len = 0; const double *prev_buf = nullptr;      // folder: the tile handed over at the last barrier
    int prev_first_of_row = 0, prev_row = 0;
    for (int i = 0; i <= A + 1; i++) {
        const int len = (i <= A) ? B + 2 : B + 1;
        double u = (i <= A) ? UL[i] : 1.0;
        for (int c0 = 0; c0 < len; c0 += DE4_TJ, gt++) {
            const int tl = min(DE4_TJ, len - c0);
            double *buf = T2 + (size_t)(gt & 1) * DE4_TJ * 64;
            if (wv != 0) {
                // dividers: running product over the tile, every third pair divided here (divider wv-1 takes c = wv-1 mod 3)
                auto fac = [&](int c) -> double { const int k = (i <= A) ? c : c + 1; return (k == 0) ? x1 : (k == 1) ? x2 : xr[k - 2]; };
                int c = 0;
                for (; c + 12 <= tl; c += 12) {
                    double uu[12];
#pragma unroll
                    for (int q = 0; q < 12; q++) { u = u * fac(c0 + c + q); uu[q] = u; }
                    double mine[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) mine[q] = (wv == 1) ? uu[3 * q] : (wv == 2) ? uu[3 * q + 1] : uu[3 * q + 2];
                    double t2v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) t2v[q] = de_t2<FAST>(mine[q]);
#pragma unroll
                    for (int q = 0; q < 4; q++) buf[(size_t)(c + 3 * q + (wv - 1)) * 64 + lane] = t2v[q];
                }
                for (; c < tl; c++) {                         // tail of the row: pair by pair, round robin
                    u = u * fac(c0 + c);
                    if ((c % 3) == wv - 1) buf[(size_t)c * 64 + lane] = de_t2<FAST>(u);
                }
            } else if (prev_buf) {
                // folder: the tile of the previous barrier (its row's tabulated factors first, if it opens the row)
                if (prev_first_of_row && prev_row <= A) a = sl.chain(a, A - prev_row, lane);
#pragma unroll 4
                for (int c = 0; c < prev_len; c++) a = a * prev_buf[(size_t)c * 64 + lane];
            }
            __syncthreads();
            prev_buf = buf; prev_len = tl; prev_first_of_row = (c0 == 0); prev_row = i;
        }
    }
    if (wv != 0) return;                                     // dividers are done
    if (prev_buf) {
        if (prev_first_of_row && prev_row <= A) a = sl.chain(a, A - prev_row, lane);
        for (int c = 0; c < prev_len; c++) a = a * prev_buf[(size_t)c * 64 + lane];
    }
    a = sr.chain(a, B * (B + 1) / 2, lane);
    // ---- b-part (id 2) and the weights (:197-218), order of de_finish ----
    const int id = P.ising_id;
    double b = 0.0;
    if (id == 2) {
        double v = 1.0, ww = 1.0, vk = 1.0, wk = 1.0;
        for (int j = B - 1; j >= 0; j--) { vk = vk * xr[j]; v = v + vk; }
        vk = vk * x2; v = v + vk;
        vk = vk * x1; v = v + vk;
        for (int j = A - 1; j >= 0; j--) { vk = vk * xl[j]; v = v + vk; }
        for (int j = 0; j < A; j++) { wk = wk * xl[j]; ww = ww + wk; }
        wk = wk * x1; ww = ww + wk;
        wk = wk * x2; ww = ww + wk;
        for (int j = 0; j < B; j++) { wk = wk * xr[j]; ww = ww + wk; }
        b = 1.0 / (v * ww);
    }
    double f = (id == 2) ? 2 * a * b : 2 * a;
    for (int j = 0; j < A; j++) f = f * wl[j];
    f = f * w1; f = f * w2;
    for (int j = 0; j < B; j++) f = f * wr[j];
    a = f;
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
                const double *wvp = Wq + u_ + (size_t)P.NM * v_;
                const double *xc = Cp + (cur.ii - 1) + (size_t)P.RM * (cur.jj - 1);
                double tt = 0.0;
#pragma unroll 8
                for (int s = 0; s < r1; s++) tt = tt + wvp[P.SW * s] * xc[P.SS * s];
                bb = bb + (-1.0) * tt;
            }
            ab = fabs(bb); bi = t;
        }
        wave_argmax(ab, bb, bi);
        if (lane == 0) { Partial pr; pr.absmax = ab; pr.val = bb; pr.idx = bi; pr.pad = 0; gs.Pt[h & 1][w] = pr; }
    }
}
