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
    // two batches in flight (four measured no faster: D_256 half-steps 4.17 vs 4.20 s)
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

template <int K> __device__ __forceinline__ double rowbc(double f)
{
    return __longlong_as_double(__builtin_amdgcn_mov_dpp(__double_as_longlong(f), 0x150 + K, 0xf, 0xf, false));
}

#define TTX_RB8LO(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define TTX_RB8HI(M) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
// wave-uniform stream of doubles consumed in order by ONE wave, delivered by DPP row broadcasts: batches of 128 entries (two
// per lane and load, DEPTH batches in flight) are parked in an LDS ring and read back 64 at a time as four registers whose
// lane n (of every DPP row) holds entry 16 b + n.  A run that starts or ends inside a block of 16 multiplies the positions
// outside it by 1.0 (exact) instead of branching per factor.
template <int DEPTH>
struct WStreamD {
    const double *g; double *buf; double r[DEPTH][2], f0, f1, f2, f3, h0, h1, h2, h3; int total, nextb, rd; bool second;
    __device__ __forceinline__ void ld(int b, int lane, double &x, double &y) const
    { const int ix = b * 128 + 2 * lane; x = ix < total ? g[ix] : 1.0; y = ix + 1 < total ? g[ix + 1] : 1.0; }
    __device__ __forceinline__ void init(const double *g_, int total_, double *buf_, int lane)
    {
        g = g_; total = total_; buf = buf_; nextb = DEPTH; rd = 128; second = true; f0 = f1 = f2 = f3 = h0 = h1 = h2 = h3 = 1.0;
#pragma unroll
        for (int k = 0; k < DEPTH; k++) ld(k, lane, r[k][0], r[k][1]);
    }
    // both halves of the ring are read back right after it has been written: one LDS round trip per 128 entries
    __device__ __forceinline__ void take128(int lane)
    {
        const double *q = buf + (lane & 15);
        f0 = q[0]; f1 = q[16]; f2 = q[32]; f3 = q[48]; h0 = q[64]; h1 = q[80]; h2 = q[96]; h3 = q[112];
    }
    __device__ __forceinline__ void refill(int lane)
    {
        __builtin_amdgcn_wave_barrier();
        buf[2 * lane] = r[0][0]; buf[2 * lane + 1] = r[0][1];
#pragma unroll
        for (int k = 0; k + 1 < DEPTH; k++) { r[k][0] = r[k + 1][0]; r[k][1] = r[k + 1][1]; }
        ld(nextb, lane, r[DEPTH - 1][0], r[DEPTH - 1][1]); nextb++; rd = 0; second = false;
        __builtin_amdgcn_wave_barrier();
        take128(lane);
        __builtin_amdgcn_wave_barrier();
    }
    // one block of 16: entries off .. off+c-1 of register fb (positions outside the run are multiplied as 1.0); the
    // broadcasts of half a block are taken before its eight dependent multiplies
    __device__ __forceinline__ double fold(double a, double fb, int off, int c, int n) const
    {
        const double fm = (n >= off && n < off + c) ? fb : 1.0;
        if (off < 8) {
            const double b0 = rowbc<0>(fm), b1 = rowbc<1>(fm), b2 = rowbc<2>(fm), b3 = rowbc<3>(fm), b4 = rowbc<4>(fm), b5 = rowbc<5>(fm), b6 = rowbc<6>(fm), b7 = rowbc<7>(fm);
            a = a * b0; a = a * b1; a = a * b2; a = a * b3; a = a * b4; a = a * b5; a = a * b6; a = a * b7;
        }
        if (off + c > 8) {
            const double b0 = rowbc<8>(fm), b1 = rowbc<9>(fm), b2 = rowbc<10>(fm), b3 = rowbc<11>(fm), b4 = rowbc<12>(fm), b5 = rowbc<13>(fm), b6 = rowbc<14>(fm), b7 = rowbc<15>(fm);
            a = a * b0; a = a * b1; a = a * b2; a = a * b3; a = a * b4; a = a * b5; a = a * b6; a = a * b7;
        }
        return a;
    }
    // a whole block of 16 (no mask, no tests)
    __device__ __forceinline__ double fold16(double a, double fb) const
    {
        {
            const double b0 = rowbc<0>(fb), b1 = rowbc<1>(fb), b2 = rowbc<2>(fb), b3 = rowbc<3>(fb), b4 = rowbc<4>(fb), b5 = rowbc<5>(fb), b6 = rowbc<6>(fb), b7 = rowbc<7>(fb);
            a = a * b0; a = a * b1; a = a * b2; a = a * b3; a = a * b4; a = a * b5; a = a * b6; a = a * b7;
        }
        {
            const double b0 = rowbc<8>(fb), b1 = rowbc<9>(fb), b2 = rowbc<10>(fb), b3 = rowbc<11>(fb), b4 = rowbc<12>(fb), b5 = rowbc<13>(fb), b6 = rowbc<14>(fb), b7 = rowbc<15>(fb);
            a = a * b0; a = a * b1; a = a * b2; a = a * b3; a = a * b4; a = a * b5; a = a * b6; a = a * b7;
        }
        return a;
    }
    // one long run from a batch boundary of the stream (the tabulated tail TR, from position 0): whole batches of 128 factors
    // without a test, the rest through chain()
    __device__ __forceinline__ double chain_from_boundary(double a, int cnt, int lane)
    {
        while (cnt >= 128 && rd == 128) {
            refill(lane);
            a = fold16(a, f0); a = fold16(a, f1); a = fold16(a, f2); a = fold16(a, f3);
            a = fold16(a, h0); a = fold16(a, h1); a = fold16(a, h2); a = fold16(a, h3);
            rd = 128; second = true; cnt -= 128;
        }
        return chain(a, cnt, lane);
    }
    __device__ __forceinline__ double chain(double a, int cnt, int lane)
    {
        const int n = lane & 15;
        while (cnt > 0) {
            if (rd == 128) refill(lane);
            // the four registers are addressed statically (a run-time index would put them into scratch memory)
#define TTX_DETB(b, fb) if (((rd >> 4) & 3) == b && cnt > 0) { const int off = rd & 15, c = cnt < 16 - off ? cnt : 16 - off; a = (c == 16) ? fold16(a, fb) : fold(a, fb, off, c, n); rd += c; cnt -= c; }
            TTX_DETB(0, f0) TTX_DETB(1, f1) TTX_DETB(2, f2) TTX_DETB(3, f3)
#undef TTX_DETB
            if (rd == 64 && !second) { f0 = h0; f1 = h1; f2 = h2; f3 = h3; second = true; }
        }
        return a;
    }
};

// The same stream consumed by LDS BROADCASTS instead of DPP row broadcasts: every lane reads the ring buffer itself, two factors per
// ds_read_b128, one v_mul_f64 per factor.  A DPP broadcast of a double is two VALU instructions on top of the multiply; where a
// kernel keeps several such chains per SIMD busy (the lottery: one wave per candidate, 2.6 waves per SIMD) the VALU issue is the
// bound and the LDS port is idle -- the half-steps with at most one wave per SIMD stay with WStreamD.  Same factors, same order.
struct WStreamL {
    const double *g; double *buf; double r[2][2]; int total, nextb, rd;
    __device__ __forceinline__ void ld(int b, int lane, double &x, double &y) const
    { const int ix = b * 128 + 2 * lane; x = ix < total ? g[ix] : 1.0; y = ix + 1 < total ? g[ix + 1] : 1.0; }
    __device__ __forceinline__ void init(const double *g_, int total_, double *buf_, int lane)
    { g = g_; total = total_; buf = buf_; nextb = 2; rd = 128; ld(0, lane, r[0][0], r[0][1]); ld(1, lane, r[1][0], r[1][1]); }
    __device__ __forceinline__ void refill(int lane)
    {
        __builtin_amdgcn_wave_barrier();
        buf[2 * lane] = r[0][0]; buf[2 * lane + 1] = r[0][1];
        r[0][0] = r[1][0]; r[0][1] = r[1][1];
        ld(nextb, lane, r[1][0], r[1][1]); nextb++; rd = 0;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ double chain(double a, int cnt, int lane)
    {
        while (cnt > 0) {
            if (rd == 128) refill(lane);
            int c = cnt < 128 - rd ? cnt : 128 - rd;
            cnt -= c;
            if ((rd & 1) && c > 0) { a = a * buf[rd]; rd++; c--; }                 // up to a 16-byte boundary
            for (; c >= 8; c -= 8, rd += 8) {
                double f[8];
#pragma unroll
                for (int k = 0; k < 4; k++) { const Double2 q2 = *reinterpret_cast<const Double2 *>(buf + rd + 2 * k); f[2 * k] = q2.a; f[2 * k + 1] = q2.b; }
#pragma unroll
                for (int k = 0; k < 8; k++) a = a * f[k];
            }
            for (; c > 0; c--, rd++) a = a * buf[rd];
        }
        return a;
    }
    __device__ __forceinline__ double chain_from_boundary(double a, int cnt, int lane) { return chain(a, cnt, lane); }
};

#ifndef DE_RUNW
#define DE_RUNW 8        // division chains interleaved per batch (16 measured 1 % faster on D_256: not worth the registers)
#endif
// the bond-spanning tail of a row: pair with s2 (x2), then with the right dims xr[0..B) (LDS, wave-uniform), eight pairs
// at a time: running products, the eight divisions stage by stage (de_t2xw), then the factors into `a` in order
template <bool FAST>
__device__ __forceinline__ void de_run(double &a, double u, double x2, const double *xr, int B)
{
    u = u * x2; a = a * de_t2<FAST>(u);
    int j = 0;
    if (B >= DE_RUNW) {
        double y[DE_RUNW];
#pragma unroll
        for (int k = 0; k < DE_RUNW; k++) y[k] = xr[k];
        for (; j + DE_RUNW <= B; j += DE_RUNW) {
            double uu[DE_RUNW], t[DE_RUNW];
#pragma unroll
            for (int k = 0; k < DE_RUNW; k++) { u = u * y[k]; uu[k] = u; }
            if (j + 2 * DE_RUNW <= B) {                 // next batch's LDS reads fly under the divisions
#pragma unroll
                for (int k = 0; k < DE_RUNW; k++) y[k] = xr[j + DE_RUNW + k];
            }
            de_t2xw<FAST, DE_RUNW>(uu, t);
#pragma unroll
            for (int k = 0; k < DE_RUNW; k++) a = a * t[k];
        }
    }
    for (; j + 4 <= B; j += 4) {
        const double u1 = u * xr[j], u2 = u1 * xr[j + 1], u3 = u2 * xr[j + 2], u4 = u3 * xr[j + 3];
        double t1, t2, t3, t4;
        de_t2x4<FAST>(u1, u2, u3, u4, t1, t2, t3, t4);
        a = a * t1; a = a * t2; a = a * t3; a = a * t4;
        u = u4;
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
    // what steers the control flow is made wave-uniform explicitly (values read from LDS / global memory are per-lane
    // registers to the compiler: loop counters and branches would otherwise run on the vector unit)
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
    const int p = UNI(cur.p), r0 = UNI(cur.r0), r1 = UNI(cur.r1), r2 = UNI(cur.r2), n1 = UNI(cur.n1), n2 = UNI(cur.n2), first = UNI(gs.first);
    const int c_ii = UNI(cur.ii), c_jj = UNI(cur.jj), c_kk = UNI(cur.kk), c_qq = UNI(cur.qq);
    const int nf = iscol ? r0 * n1 : n2 * r2;
    const int nv = iscol ? r0 : r2, nm = iscol ? n1 : n2, nch = (nm + 63) >> 6;
    const int npart = nv * nch;
    const int w = blockIdx.x;
    const int crs = UNI(cur.crs) + 1;
    const int havecol = UNI(cur.havecol) | (iscol ? 1 : 0), haverow = UNI(cur.haverow) | (iscol ? 0 : 1);
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
    const int pl = iscol ? pv : c_ii - 1, qr = iscol ? c_qq - 1 : pv;         // left / right pivot of this wave
    const int n1m = UNI(P.n[1]);
    const double *nodes = P.par, *weights = P.par + n1m;                          // 0-based here
    // LDS: UL[VS] | xl[VS] | wl[VS] | xr[VS] | wr[VS] | ring L[128] | ring R[128]
    const int VS = ((m + 7) & ~7) + 8;
    double *UL = dyn, *xl = UL + VS, *wl = xl + VS, *xr = wl + VS, *wr = xr + VS, *ringL = wr + VS, *ringR = ringL + 128;
    const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
    const double *TLg = P.deTL + (size_t)g * tsz + (size_t)pl * NP, *TRg = P.deTR + (size_t)g * tsz + (size_t)qr * NP;
    const double *ULg = P.deUL + ((size_t)g * P.RM + pl) * (m + 1);
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = lane; x < A; x += 64) { const int ix = Lt[(size_t)x * P.RM + pl] - 1; xl[x] = nodes[ix]; wl[x] = weights[ix]; }
    for (int x = lane; x < B; x += 64) { const int ix = Rt[(size_t)x * P.RM + qr] - 1; xr[x] = nodes[ix]; wr[x] = weights[ix]; }
    for (int x = lane; x <= A; x += 64) UL[x] = ULg[x];
    const int i1 = iscol ? (live ? vmode : 0) : c_jj - 1, i2 = iscol ? c_kk - 1 : (live ? vmode : 0);   // node index of dim p / p+1
    const double x1 = nodes[i1], x2 = nodes[i2], w1 = weights[i1], w2 = weights[i2];
    WStreamD<2> sl, sr;                                                // tabulated factors by DPP row broadcasts (end of round 2; LDS broadcasts before)
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
    a = sr.chain_from_boundary(a, B * (B + 1) / 2, lane);
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
                const double *xq = Wq + (c_kk - 1) + (size_t)P.NM * (c_qq - 1);
#pragma unroll 8
                for (int s = 0; s < r1; s++) bb = bb + (-xq[P.SW * s]) * c[P.SS * s];
            } else {       // dgemv 't', alpha=-1 (:571): b += -1 * sum_s row(s, kq) * x_s, x_s = col(p)(ii, jj, s)
                const double *wv = Wq + u_ + (size_t)P.NM * v_;
                const double *xc = Cp + (c_ii - 1) + (size_t)P.RM * (c_jj - 1);
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
#undef UNI

#define UNI(x) __builtin_amdgcn_readfirstlane(x)
// the bond-spanning tail of a row as de_run, ended where every lane's running product has reached the unit cut (the factors of a
// lane that is already there are exactly 1)
__device__ __forceinline__ void de_run_cut(double &a, double u, double x2, const double *xr, int B)
{
    u = u * x2; a = a * de_t2<true>(u);
    int j = 0;
    while (j < B && __builtin_amdgcn_ballot_w64(u > 0x1p-54) != 0ull) {
        if (j + DE_RUNW <= B) {
            double uu[DE_RUNW], t[DE_RUNW];
#pragma unroll
            for (int k = 0; k < DE_RUNW; k++) { u = u * xr[j + k]; uu[k] = u; }
            de_t2xw<true, DE_RUNW>(uu, t);
#pragma unroll
            for (int k = 0; k < DE_RUNW; k++) a = a * t[k];
            j += DE_RUNW;
        } else {
            for (; j < B; j++) { u = u * xr[j]; a = a * de_t2<true>(u); }
        }
    }
}

// value of the fiber elements (left pivot pl | node i1 | node i2 | right pivot qr) of bond p, one per lane (pl, qr wave-uniform;
// i1, i2 per lane), from the compact tables: tabulated factors by DPP row broadcasts, the bond-spanning tails by division, both
// ended at the unit cut, everything in the reference's order.  All 64 lanes call it; dyn = the workgroup's dynamic LDS.
#ifndef DEC_HALF_STREAM
#define DEC_HALF_STREAM WStreamD<DEC_DEPTH>     // stream of the half-step's tabulated factors (WStreamL: LDS broadcasts)
#endif
#ifndef DEC_DEPTH
#define DEC_DEPTH 2      // batches of 128 tabulated factors in flight per stream
#endif
template <class STREAM>
__device__ __forceinline__ double dec_value(const DevProb &P, int g, int p, int first, int pl, int qr, int i1, int i2, double *dyn, int lane)
{
    const int m = P.d, A = p - 1, B = m - p - 1;
    const int n1m = UNI(P.n[1]);
    const double *nodes = P.par, *weights = P.par + n1m;                          // 0-based here
    // LDS: UL[VS] | xl[VS] | wl[VS] | xr[VS] | wr[VS] | ring L[128] | ring R[128] | row counts
    const int VS = ((m + 7) & ~7) + 8;
    double *UL = dyn, *xl = UL + VS, *wl = xl + VS, *xr = wl + VS, *wr = xr + VS, *ringL = wr + VS, *ringR = ringL + 128;
    int *cntL = (int *)(ringR + 128);
    const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
    const double *TLg = P.deTL + (size_t)g * tsz + (size_t)pl * NP, *TRg = P.deTR + (size_t)g * tsz + (size_t)qr * NP;
    const double *ULg = P.deUL + ((size_t)g * P.RM + pl) * (m + 1);
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = lane; x < A; x += 64) { const int ix = Lt[(size_t)x * P.RM + pl] - 1; xl[x] = nodes[ix]; wl[x] = weights[ix]; }
    for (int x = lane; x < B; x += 64) { const int ix = Rt[(size_t)x * P.RM + qr] - 1; xr[x] = nodes[ix]; wr[x] = weights[ix]; }
    for (int x = lane; x <= A; x += 64) UL[x] = ULg[x];
    const int *CLg = P.deCL + ((size_t)g * P.RM + pl) * (m + 1), *CRg = P.deCR + ((size_t)g * P.RM + qr) * (m + 1);
    for (int x = lane; x < A; x += 64) cntL[x] = CLg[x];
    const int totL = UNI(CLg[m]), totR = UNI(CRg[m]);
    const double x1 = nodes[i1], x2 = nodes[i2], w1 = weights[i1], w2 = weights[i2];
    STREAM sl, sr;                                                // tabulated factors by DPP row broadcasts
    sl.init(TLg, totL, ringL, lane);
    sr.init(TRg, totR, ringR, lane);
    __syncthreads();
    // ---- pair product (test_crs_ising.f90:186-195), order of de_pairs_ctab ----
    double a = 1.0;
    const int i0 = UNI((int)ULg[m]), pre = UNI(CLg[m - 1]);
    a = sl.chain_from_boundary(a, pre, lane);                          // rows 0 .. i0-1: they end inside the left pivot's dims
    for (int i = i0; i <= A; i++) {                                    // the rows that reach the bond
        if (i < A) a = sl.chain(a, UNI(cntL[i]), lane);
        double u = UL[i];
        u = u * x1; a = a * de_t2<true>(u);
        de_run_cut(a, u, x2, xr, B);
    }
    de_run_cut(a, 1.0, x2, xr, B);                                     // i = A+1: starts after dim p
    a = sr.chain_from_boundary(a, totR, lane);
    // ---- b-part (id 2) and the weights (:197-218), order of de_finish.  v >= 1 and the running product vk never grows: once
    //      vk <= 2^-54 every further v + vk returns v (half an ulp of v is at least 2^-53) -- the sums end there, bit for bit ----
    const int id = P.ising_id;
    double b = 0.0;
    if (id == 2) {
        double v = 1.0, ww = 1.0, vk = 1.0, wk = 1.0;
        bool on = true;
        for (int j = B - 1; j >= 0 && on; j--) { vk = vk * xr[j]; v = v + vk; on = __builtin_amdgcn_ballot_w64(vk > 0x1p-54) != 0ull; }
        if (on) { vk = vk * x2; v = v + vk; vk = vk * x1; v = v + vk; on = __builtin_amdgcn_ballot_w64(vk > 0x1p-54) != 0ull; }
        for (int j = A - 1; j >= 0 && on; j--) { vk = vk * xl[j]; v = v + vk; on = __builtin_amdgcn_ballot_w64(vk > 0x1p-54) != 0ull; }
        on = true;
        for (int j = 0; j < A && on; j++) { wk = wk * xl[j]; ww = ww + wk; on = __builtin_amdgcn_ballot_w64(wk > 0x1p-54) != 0ull; }
        if (on) { wk = wk * x1; ww = ww + wk; wk = wk * x2; ww = ww + wk; on = __builtin_amdgcn_ballot_w64(wk > 0x1p-54) != 0ull; }
        for (int j = 0; j < B && on; j++) { wk = wk * xr[j]; ww = ww + wk; on = __builtin_amdgcn_ballot_w64(wk > 0x1p-54) != 0ull; }
        b = 1.0 / (v * ww);
    }
    double f = (id == 2) ? 2 * a * b : 2 * a;
    for (int j = 0; j < A; j++) f = f * wl[j];
    f = f * w1; f = f * w2;
    for (int j = 0; j < B; j++) f = f * wr[j];
    return f;
}
// lottery candidates of the compact-table path: one wave per candidate (the lanes work redundantly: the value is a dependent chain)
// grid = (lot_max, groups), 64 threads; candidates from P.lotc (k_lottery phase 1), values to P.lotf (phase 2)
__global__ __launch_bounds__(64) void k_lottery_eval_dec(DevProb P)
{
    extern __shared__ __align__(16) double dyn[];
    const int g = blockIdx.y, lane = threadIdx.x, il = blockIdx.x;
    const GroupState &gs = P.gs[g];
    const StepState &st = gs.S[0];
    if (!UNI(st.active)) return;
    const int nlot = UNI(st.r0) + UNI(st.n1) + UNI(st.n2) + UNI(st.r2);
    if (il >= nlot) return;
    const int *c_ = P.lotc + ((size_t)g * P.lot_max + il) * 4;
    const int i = UNI(c_[0]), j = UNI(c_[1]), k = UNI(c_[2]), q = UNI(c_[3]);
    const double f = dec_value<WStreamL>(P, g, UNI(st.p), UNI(gs.first), i - 1, q - 1, j - 1, k - 1, dyn, lane);
    if (lane == 0) P.lotf[(size_t)g * P.lot_max + il] = f;
}

// lottery candidates of the unit-cut path: one wave per candidate, no tables -- a candidate is ONE multi-index, whose pair product a
// wave forms row-parallel (de_pairs_point_wave_cut: every lane walks a row of the triangle, the factors above the cut are then
// multiplied in order out of LDS); b-part and weights by de_finish_vals.  The chain of ~3 900 ordered multiplies is what remains.
// grid = (lot_max, groups), 64 threads, dynamic LDS 2 (m + 64) + 2048 doubles; candidates from P.lotc (k_lottery phase 1),
// values to P.lotf (phase 2)
__global__ __launch_bounds__(64) void k_lottery_eval_decp(DevProb P)
{
    extern __shared__ __align__(16) double dyn[];
    const int g = blockIdx.y, lane = threadIdx.x, il = blockIdx.x, m = P.d;
    const GroupState &gs = P.gs[g];
    const StepState &st = gs.S[0];
    if (!UNI(st.active)) return;
    const int nlot = UNI(st.r0) + UNI(st.n1) + UNI(st.n2) + UNI(st.r2);
    if (il >= nlot) return;
    const int *c_ = P.lotc + ((size_t)g * P.lot_max + il) * 4;
    const int ci = UNI(c_[0]) - 1, cj = UNI(c_[1]) - 1, ck = UNI(c_[2]) - 1, cq = UNI(c_[3]) - 1;
    const int p = UNI(st.p), first = UNI(gs.first), A = p - 1, n1m = UNI(P.n[1]);
    double *xv = dyn, *wv = dyn + (m + 64), *sf = wv + (m + 64);
    const double *nodes = P.par, *weights = P.par + n1m;
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = lane; x < m; x += 64) {
        const int ix = (x < A) ? Lt[(size_t)x * P.RM + ci] - 1 : (x == A) ? cj : (x == A + 1) ? ck : Rt[(size_t)(x - A - 2) * P.RM + cq] - 1;
        xv[x] = nodes[ix]; wv[x] = weights[ix];
    }
    __builtin_amdgcn_wave_barrier();
    const double f = de_finish_vals(P.ising_id, de_pairs_point_wave_cut(m, xv, sf, lane), m, xv, wv);
    if (lane == 0) P.lotf[(size_t)g * P.lot_max + il] = f;
}

// The same kernel on the COMPACT tables of k_de_ctables (P.de_cut: all nodes in [0,1]): every row of the pair triangle ends at the
// unit cut -- its tabulated part after CL[i] factors, its bond-spanning tail where the lanes' running products have all reached
// 2^-54 (a lane that is already there multiplies factors that are exactly 1).  ~12 % of the factors of D_256 are left, in the
// reference's order: the same bits.  LDS: as k_halfstep_de + the row counts.
__global__ __launch_bounds__(64) void k_halfstep_dec(DevProb P, int h, int dir, int mode)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ StepState cur;
    const int g = blockIdx.y, lane = threadIdx.x, m = P.d;
    GroupState &gs = P.gs[g];
    if (lane == 0) { cur = gs.S[h]; resolve_state(cur, gs.Pt[(h + 1) & 1]); }
    __syncthreads();
    if (!cur.active || cur.done) { if (blockIdx.x == 0 && lane == 0) gs.S[h + 1] = cur; return; }
    const bool iscol = (mode == 1 || mode == 2) ? (h == 0) : (((h + (dir == 2 ? 1 : 0)) & 1) == 0);    // :517,550
    // what steers the control flow is made wave-uniform explicitly (values read from LDS / global memory are per-lane
    // registers to the compiler: loop counters and branches would otherwise run on the vector unit)
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
    const int p = UNI(cur.p), r0 = UNI(cur.r0), r1 = UNI(cur.r1), r2 = UNI(cur.r2), n1 = UNI(cur.n1), n2 = UNI(cur.n2), first = UNI(gs.first);
    const int c_ii = UNI(cur.ii), c_jj = UNI(cur.jj), c_kk = UNI(cur.kk), c_qq = UNI(cur.qq);
    const int nf = iscol ? r0 * n1 : n2 * r2;
    const int nv = iscol ? r0 : r2, nm = iscol ? n1 : n2, nch = (nm + 63) >> 6;
    const int npart = nv * nch;
    const int w = blockIdx.x;
    const int crs = UNI(cur.crs) + 1;
    const int havecol = UNI(cur.havecol) | (iscol ? 1 : 0), haverow = UNI(cur.haverow) | (iscol ? 0 : 1);
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
    const int pl = iscol ? pv : c_ii - 1, qr = iscol ? c_qq - 1 : pv;         // left / right pivot of this wave
    const int i1 = iscol ? (live ? vmode : 0) : c_jj - 1, i2 = iscol ? c_kk - 1 : (live ? vmode : 0);   // node index of dim p / p+1
    double a = dec_value<DEC_HALF_STREAM>(P, g, p, first, pl, qr, i1, i2, dyn, lane);
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
                const double *xq = Wq + (c_kk - 1) + (size_t)P.NM * (c_qq - 1);
#pragma unroll 8
                for (int s = 0; s < r1; s++) bb = bb + (-xq[P.SW * s]) * c[P.SS * s];
            } else {       // dgemv 't', alpha=-1 (:571): b += -1 * sum_s row(s, kq) * x_s, x_s = col(p)(ii, jj, s)
                const double *wv = Wq + u_ + (size_t)P.NM * v_;
                const double *xc = Cp + (c_ii - 1) + (size_t)P.RM * (c_jj - 1);
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
#undef UNI

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


// ------------------------------------------------------------------------------------------------------------------
// k_halfstep_de5: the half-step of ONE (pivot, mode-chunk) by a RELAY of four waves (round 2; measured unit costs in
// profiles/r02_probe_dpp.txt).  A bond-spanning pair costs 35-42 ns of a lone wave but only ~29 ns of a SIMD's issue slots,
// and a launch of k_halfstep_de has at most r x ceil(n/64) x groups waves -- 624 at D_256, fewer than the chip's 1024 SIMDs.
// So the body of every row of the pair triangle (the B right dims) is dealt out in segments of 16 columns, round robin
// over four waves (lane = mode index in every wave).  Per round a wave
//   * moves its running product u over the other waves' 48 columns (one multiply per column, LDS broadcasts),
//   * performs the 16 divisions of its segment, eight at a time stage by stage, keeping the factors in 32 REGISTERS,
//   * then waits for the token -- the running product `a` of the element, 64 lanes x 8 bytes in an LDS mailbox --,
//     multiplies its factors in, in order, and hands the token to the next wave.
// Wave 0 additionally owns the head of a row (the tabulated factors TL, the pairs with dims p and p+1), the tabulated
// tail TR, the b-part, the weights, the residual and the arg-max record.  While the token travels, every wave is already
// dividing for its next segment; nothing but the token crosses waves (no tile traffic, no workgroup barrier).  Every product
// is still taken by one lane in the reference's order: bit-identical to k_halfstep_de and the oracle.
// Mailboxes: a slot is either the sentinel (a NaN pattern no arithmetic produces) or a value; the receiver polls its
// own slot, takes the value and restores the sentinel before it sends on -- LDS operations of one wave complete in order
// and the next value for this slot can only be produced after the token has passed through the receiver again.
// Every wait is bounded; a broken hand-over is reported to the host (ctl[3]), which repeats the run with k_halfstep_de.
// MEASURED AND NOT ADOPTED (TTX_DE_V5=1 selects it; bit-exact, in the tests): half-steps of D_256 5.05 s against 4.17 s of
// k_halfstep_de.  Every wave still walks every column of a row with its running product (an LDS-fed multiply, 5.8 ns), so
// four waves spend 4 x 5.8 ns per pair on top of the 29 ns of the division, and 214 registers allow two waves per SIMD; a
// token hop costs 140-165 ns.  Two waves with 32-column segments: 5.42 s; 64-column segments in registers (305 registers,
// one wave per SIMD): 8.5 s.
// ------------------------------------------------------------------------------------------------------------------
#define DE5_W 4
#define DE5_SEG 16               // columns of a row's body per wave and round
#define DE5_SENT 0xfff85a5a00000001ull
__host__ __device__ inline size_t de5_lds_doubles(int m) { const int VS = ((m + 7) & ~7) + 8 + DE5_SEG; return (size_t)5 * VS + 128 + (size_t)DE5_W * 64; }
__host__ __device__ inline bool de5_fits(int m) { return m >= 3; }

__device__ __forceinline__ void de5_send(unsigned long long *box, int lane, double a)
{
    __hip_atomic_store(&box[lane], (unsigned long long)__double_as_longlong(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// false: the token did not come (bounded wait) or another wave of the workgroup gave up -- the caller leaves the kernel
__device__ __forceinline__ bool de5_recv(unsigned long long *box, int lane, double &a, int *giveup)
{
    unsigned long long v; unsigned spins = 0;
    for (;;) {
        v = __hip_atomic_load(&box[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!__any(v == DE5_SENT)) break;
        if (++spins > (1u << 20) || ((spins & 255u) == 0 && __hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
            __hip_atomic_store(giveup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(&box[lane], DE5_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    a = __longlong_as_double((long long)v);
    return true;
}
// a relay that broke: stop the run after this sweep (ctl[0]) and tell the host (ctl[3]); ttx_run then repeats the run
// with k_halfstep_de
#define DE5_FAIL() do { if (lane == 0) { atomicAdd(&P.ctl[3], 1); P.ctl[0] = 1; } return; } while (0)

template <bool FAST>
__global__ __launch_bounds__(64 * DE5_W) void k_halfstep_de5(DevProb P, int h, int dir, int mode)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ StepState cur;
    __shared__ int giveup;
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = P.d;
    GroupState &gs = P.gs[g];
    if (tid == 0) { cur = gs.S[h]; resolve_state(cur, gs.Pt[(h + 1) & 1]); giveup = 0; }
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
    const int VS = ((m + 7) & ~7) + 8 + DE5_SEG;
    double *UL = dyn, *xl = UL + VS, *wl = xl + VS, *xr = wl + VS, *wr = xr + VS, *ringL = wr + VS, *ringR = ringL + 64;
    unsigned long long *box = reinterpret_cast<unsigned long long *>(ringR + 64);          // [DE5_W][64]
    const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
    const double *TLg = P.deTL + (size_t)g * tsz + (size_t)pl * NP, *TRg = P.deTR + (size_t)g * tsz + (size_t)qr * NP;
    const double *ULg = P.deUL + ((size_t)g * P.RM + pl) * (m + 1);
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = tid; x < A; x += 64 * DE5_W) { const int ix = Lt[(size_t)x * P.RM + pl] - 1; xl[x] = nodes[ix]; wl[x] = weights[ix]; }
    for (int x = tid; x < B; x += 64 * DE5_W) { const int ix = Rt[(size_t)x * P.RM + qr] - 1; xr[x] = nodes[ix]; wr[x] = weights[ix]; }
    for (int x = B + tid; x < B + DE5_SEG; x += 64 * DE5_W) xr[x] = 1.0;       // the division batches of eight may run past B
    for (int x = tid; x <= A; x += 64 * DE5_W) UL[x] = ULg[x];
    box[tid] = DE5_SENT;
    const int i1 = iscol ? (live ? vmode : 0) : cur.jj - 1, i2 = iscol ? cur.kk - 1 : (live ? vmode : 0);
    const double x1 = nodes[i1], x2 = nodes[i2], w1 = weights[i1], w2 = weights[i2];
    __syncthreads();
    // the body xr[0..B) of a row in segments of 16 columns, dealt round robin to nact waves; R rounds per row
    const int nact = max(1, min(DE5_W, (B + DE5_SEG - 1) / DE5_SEG));
    if (wv >= nact) return;                                   // no barrier below this line
    const int R = (B > 0) ? (B + nact * DE5_SEG - 1) / (nact * DE5_SEG) : 1;
    unsigned long long *mybox = box + (size_t)wv * 64, *nextbox = box + (size_t)((wv + 1 == nact) ? 0 : wv + 1) * 64;
    double a = 1.0;
    WStream sl, sr;
    if (wv == 0) { sl.init(TLg, A * (A + 1) / 2, ringL, lane); sr.init(TRg, B * (B + 1) / 2, ringR, lane); }
    for (int i = 0; i <= A + 1; i++) {
        const bool hasx1 = (i <= A);
        const double uh = hasx1 ? UL[i] * x1 : 1.0;
        const double u2 = uh * x2;
        double u = u2;
        if (wv > 0) u = lds_chain(u, xr, wv * DE5_SEG);     // the row's running product up to my first segment
        for (int r = 0; r < R; r++) {
            const int j0 = (r * nact + wv) * DE5_SEG, cnt = max(0, min(DE5_SEG, B - j0));
            double f[DE5_SEG];
            if (cnt > 0) {                                    // wave-uniform; xr reads as 1.0 from B to B + DE5_SEG
#pragma unroll
                for (int b8 = 0; b8 < DE5_SEG / 8; b8++) {
                    if (8 * b8 < cnt) {
                        double ua[8], fa[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) { u = u * xr[j0 + 8 * b8 + k]; ua[k] = u; }
                        de_t2xw<FAST, 8>(ua, fa);
#pragma unroll
                        for (int k = 0; k < 8; k++) f[8 * b8 + k] = fa[k];
                    }
                }
            }
            if (wv == 0 && r == 0) {
                const double t1 = hasx1 ? de_t2<FAST>(uh) : 1.0, t2 = de_t2<FAST>(u2);
                if (i > 0 && nact > 1 && !de5_recv(mybox, lane, a, &giveup)) DE5_FAIL();
                if (hasx1) { a = sl.chain(a, A - i, lane); a = a * t1; }
                a = a * t2;
            } else if (nact > 1 && !de5_recv(mybox, lane, a, &giveup)) DE5_FAIL();
            if (cnt == DE5_SEG) {
#pragma unroll
                for (int k = 0; k < DE5_SEG; k++) a = a * f[k];
            } else {
#pragma unroll
                for (int k = 0; k < DE5_SEG; k++) if (k < cnt) a = a * f[k];
            }
            if (nact > 1) de5_send(nextbox, lane, a);
            // over the other waves' segments to my next one (only if there is one)
            if (r + 1 < R && cnt == DE5_SEG && nact > 1) u = lds_chain(u, xr + j0 + DE5_SEG, (nact - 1) * DE5_SEG);
        }
    }
    if (wv != 0) return;
    if (nact > 1 && !de5_recv(mybox, lane, a, &giveup)) DE5_FAIL();
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
    double fv = (id == 2) ? 2 * a * b : 2 * a;
    for (int j = 0; j < A; j++) fv = fv * wl[j];
    fv = fv * w1; fv = fv * w2;
    for (int j = 0; j < B; j++) fv = fv * wr[j];
    a = fv;
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

// ------------------------------------------------------------------------------------------------------------------
// k_lottery_eval_de_rows: FOUR lottery candidates per wave, one per DPP row of 16 lanes (round 2).
// With one candidate per wave (k_lottery_eval_de) the 32 640 ordered multiplies of an element were LDS broadcasts done
// identically by all 64 lanes (5.8-8 ns per factor and wave), and only 16 lanes divided.  All candidates of a group
// sit at the same bond, so their pair triangles have the same shape and four of them can share a wave's control flow:
//   * lane n of a row holds ONE factor of a chunk of 16 (32 for the tabulated streams) consecutive factors;
//     `a = a * factor` takes the factor of lane k by a DPP row broadcast (v_mov_b64_dpp row_newbcast:k + v_mul_f64:
//     4.5 ns per factor for the four candidates together, profiles/r02_probe_dpp.txt), no LDS traffic;
//   * bond-spanning pairs: lane n owns column c0+n of the row of the triangle: it advances its own running product u by the
//     16 node values between two chunks (per-lane LDS reads, conflict-free) and divides once per chunk -- all 64 lanes
//     divide; the advance of chunk c+1, the division of chunk c and the ordered multiplies of chunk c-1 are independent
//     instruction streams in one loop body;
//   * tabulated factors TL / TR: lane n loads factors 2n, 2n+1 of a 32-factor chunk of ITS candidate's table row, four chunks
//     ahead.
// The order of all products is that of de_pairs_tab (the reference's row-major pair loop), the operations per factor those
// of de_t2: bit-identical.
// ------------------------------------------------------------------------------------------------------------------
#define TTX_RB16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
// a = a * (factor of lane off) * ... * (factor of lane off+c-1) of every DPP row; off, c wave-uniform
__device__ __forceinline__ double row_fold16(double a, double f, int off, int c)
{
    if (off == 0 && c == 16) {
#define TTX_RB_ALL(k) a = a * rowbc<k>(f);
        TTX_RB16(TTX_RB_ALL)
#undef TTX_RB_ALL
    } else {
#define TTX_RB_SOME(k) if (k >= off && k < off + c) a = a * rowbc<k>(f);
        TTX_RB16(TTX_RB_SOME)
#undef TTX_RB_SOME
    }
    return a;
}
// the same over a chunk of 32 factors held as (lo, hi) = factors (2 lane, 2 lane + 1)
__device__ __forceinline__ double row_fold32(double a, double lo, double hi, int off, int c)
{
    if (off == 0 && c == 32) {
#define TTX_RB_ALL(k) a = a * rowbc<k>(lo); a = a * rowbc<k>(hi);
        TTX_RB16(TTX_RB_ALL)
#undef TTX_RB_ALL
    } else {
#define TTX_RB_SOME(k) if (2 * k >= off && 2 * k < off + c) a = a * rowbc<k>(lo); if (2 * k + 1 >= off && 2 * k + 1 < off + c) a = a * rowbc<k>(hi);
        TTX_RB16(TTX_RB_SOME)
#undef TTX_RB_SOME
    }
    return a;
}
// per-DPP-row stream of tabulated factors g[0..total): every row has its own g, all rows the same total and the same takes
struct RStream {
    const double *g; int total, nextc, off; double lo0, hi0, lo1, hi1, lo2, hi2, lo3, hi3;
    __device__ __forceinline__ void ld(int c, int n, double &lo, double &hi) const
    {
        const int ix = 32 * c + 2 * n;
        lo = ix < total ? g[ix] : 1.0; hi = ix + 1 < total ? g[ix + 1] : 1.0;
    }
    __device__ __forceinline__ void init(const double *g_, int total_, int n)
    { g = g_; total = total_; ld(0, n, lo0, hi0); ld(1, n, lo1, hi1); ld(2, n, lo2, hi2); ld(3, n, lo3, hi3); nextc = 4; off = 0; }
    __device__ __forceinline__ double take(double a, int cnt, int n)
    {
        while (cnt > 0) {
            if (off == 32) { lo0 = lo1; hi0 = hi1; lo1 = lo2; hi1 = hi2; lo2 = lo3; hi2 = hi3; ld(nextc, n, lo3, hi3); nextc++; off = 0; }
            const int c = cnt < 32 - off ? cnt : 32 - off;
            a = row_fold32(a, lo0, hi0, off, c);
            off += c; cnt -= c;
        }
        return a;
    }
};
// LDS per DPP row: xv | wv | UL, each RSW doubles; RSW = 16 mod 32 so that the four rows' 16-lane reads hit disjoint bank halves
__host__ __device__ inline int de_rows_stride(int m) { const int s = m + 56; return ((s + 31) & ~31) + 16; }
__host__ __device__ inline size_t de_rows_lds_doubles(int m) { return (size_t)12 * de_rows_stride(m); }

// One row of the pair triangle for the four candidates of a wave: a = a * t(u0 xs[0]) * t(u0 xs[0] xs[1]) * ... (L factors,
// t(u) = ((u-1)/(u+1))^2); `neutral0`: the first column is the neutral 1.0 (the row that starts after dim p).
// xs is this DPP row's array of node values, readable (as 1.0) up to 47 entries past L.
template <bool FAST>
__device__ __forceinline__ double rows_span(double a, double u0, const double *xs, int L, bool neutral0, int n)
{
    const int nchunk = (L + 15) >> 4, cntlast = L - 16 * (nchunk - 1);
    // column n of chunk 0: u = u0 * xs[0] * .. * xs[n]
    double u = u0;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const double x = (s == 0 && neutral0) ? 1.0 : xs[s];
        u = u * ((s <= n) ? x : 1.0);
    }
    // steady state per chunk c: three independent chains -- the advance of u to chunk c+1 (16 multiplies by node values
    // read one chunk ahead), the division of chunk c, the ordered multiplies of chunk c-1 -- written INTERLEAVED,
    // statement by statement (the compiler otherwise emits them one after the other, every LDS latency exposed)
    double xq[16], xn[16];
#pragma unroll
    for (int s = 0; s < 16; s++) xq[s] = xs[n + 1 + s];
    double unx = u;
#pragma unroll
    for (int s = 0; s < 16; s++) unx = unx * xq[s];
#pragma unroll
    for (int s = 0; s < 16; s++) xq[s] = xs[16 + n + 1 + s];                 // node values for the advance to chunk 2
    double tprev = de_t2<FAST>(u);
    if (neutral0 && n == 0) tprev = 1.0;
    // one chunk: advance with the node values in `xc`, fetch those of the chunk after into `xf` (the two arrays swap
    // roles from chunk to chunk: no register copies)
    auto chunk = [&](const double (&xc)[16], double (&xf)[16], int c) {
        u = unx;
        const double *xa = xs + 16 * (c + 1) + n + 1;
#pragma unroll
        for (int s = 0; s < 16; s++) xf[s] = xa[s];                          // one chunk ahead: in flight under this chunk's work
        double tcur;
        if (FAST) {
            const double nn = u - 1.0, dd = u + 1.0;
            a = a * rowbc<0>(tprev);   unx = unx * xc[0];
            double r = __builtin_amdgcn_rcp(dd);
            a = a * rowbc<1>(tprev);   unx = unx * xc[1];
            double e = __builtin_fma(-dd, r, 1.0);
            a = a * rowbc<2>(tprev);   unx = unx * xc[2];
            r = __builtin_fma(r, e, r);
            a = a * rowbc<3>(tprev);   unx = unx * xc[3];
            e = __builtin_fma(-dd, r, 1.0);
            a = a * rowbc<4>(tprev);   unx = unx * xc[4];
            r = __builtin_fma(r, e, r);
            a = a * rowbc<5>(tprev);   unx = unx * xc[5];
            const double q = nn * r;
            a = a * rowbc<6>(tprev);   unx = unx * xc[6];
            const double rem = __builtin_fma(-dd, q, nn);
            a = a * rowbc<7>(tprev);   unx = unx * xc[7];
            const double sq = __builtin_fma(rem, r, q);
            a = a * rowbc<8>(tprev);   unx = unx * xc[8];
            tcur = sq * sq;
            a = a * rowbc<9>(tprev);   unx = unx * xc[9];
            a = a * rowbc<10>(tprev);  unx = unx * xc[10];
            a = a * rowbc<11>(tprev);  unx = unx * xc[11];
            a = a * rowbc<12>(tprev);  unx = unx * xc[12];
            a = a * rowbc<13>(tprev);  unx = unx * xc[13];
            a = a * rowbc<14>(tprev);  unx = unx * xc[14];
            a = a * rowbc<15>(tprev);  unx = unx * xc[15];
        } else {
#pragma unroll
            for (int s = 0; s < 16; s++) unx = unx * xc[s];
            tcur = de_t2<false>(u);
            a = row_fold16(a, tprev, 0, 16);
        }
        tprev = tcur;
    };
    int c = 1;
    for (; c + 1 < nchunk; c += 2) { chunk(xq, xn, c); chunk(xn, xq, c + 1); }
    if (c < nchunk) chunk(xq, xn, c);
    return row_fold16(a, tprev, 0, cntlast);
}

// TAB: tabulated factors of the pivots (k_de_tables) for the pairs on one side of the bond, divisions only for the spanning
// pairs; !TAB: every pair by division.  Measured at D_256: the table rows of the ~330 candidates of a group are 0.5 MB each
// and come from HBM (1.4 GB per launch, latency-bound at 1.5 TB/s: 909 us per launch), while a division costs this kernel
// 13.5 instructions per 16 factors of four candidates -- so the lottery divides everything and reads no tables.
template <bool FAST, bool TAB>
__global__ __launch_bounds__(64) void k_lottery_eval_de_rows(DevProb P)
{
    extern __shared__ __align__(16) double dyn[];
    const int g = blockIdx.y, lane = threadIdx.x, row = lane >> 4, n = lane & 15, m = P.d;
    const GroupState &gs = P.gs[g];
    const StepState &st = gs.S[0];
    if (!st.active) return;
    const int p = st.p, first = gs.first;
    const int nlot = st.r0 + st.n1 + st.n2 + st.r2;
    if (4 * (int)blockIdx.x >= nlot) return;
    const int il_raw = 4 * blockIdx.x + row, il = il_raw < nlot ? il_raw : nlot - 1;   // a row past the end repeats the last candidate
    const int *cand = P.lotc + ((size_t)g * P.lot_max + il) * 4;
    const int ci = cand[0] - 1, cj = cand[1] - 1, ck = cand[2] - 1, cq = cand[3] - 1;
    const int A = p - 1, B = m - p - 1, RSW = de_rows_stride(m), n1m = P.n[1];
    double *xv = dyn + (size_t)row * 3 * RSW, *wv = xv + RSW, *UL = wv + RSW;
    const double *nodes = P.par, *weights = P.par + n1m;
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = n; x < m; x += 16) {
        const int ix = (x < A) ? Lt[(size_t)x * P.RM + ci] - 1 : (x == A) ? cj : (x == A + 1) ? ck : Rt[(size_t)(x - A - 2) * P.RM + cq] - 1;
        xv[x] = nodes[ix]; wv[x] = weights[ix];
    }
    for (int x = m + n; x < m + 56; x += 16) xv[x] = 1.0;          // the advance of u runs up to 47 columns past a row's end
    double a = 1.0;
    if (TAB) {
        const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
        const double *ULg = P.deUL + ((size_t)g * P.RM + ci) * (m + 1);
        for (int x = n; x <= A; x += 16) UL[x] = ULg[x];
        RStream sl, sr;
        sl.init(P.deTL + (size_t)g * tsz + (size_t)ci * NP, A * (A + 1) / 2, n);
        sr.init(P.deTR + (size_t)g * tsz + (size_t)cq * NP, B * (B + 1) / 2, n);
        __syncthreads();
        // xs[c]: node of column c of a spanning row: c = 0 dim p, c = 1 dim p+1, c >= 2 right dims; 1.0 past the end
        for (int i = 0; i <= A + 1; i++) {
            const bool hasx1 = (i <= A);
            if (hasx1) a = sl.take(a, A - i, n);
            a = rows_span<FAST>(a, hasx1 ? UL[i] : 1.0, xv + A, B + 2, !hasx1, n);
        }
        a = sr.take(a, B * (B + 1) / 2, n);
    } else {
        __syncthreads();
        for (int i = 0; i < m; i++) a = rows_span<FAST>(a, 1.0, xv + i, m - i, false, n);     // row i: pairs with dims i+1 .. m
    }
    const double f = de_finish_vals(P.ising_id, a, m, xv, wv);
    if (n == 0 && il_raw < nlot) P.lotf[(size_t)g * P.lot_max + il] = f;
}

// ------------------------------------------------------------------------------------------------------------------
// k_halfstep_det: the half-step of ONE (pivot, mode-chunk) by a TEAM of 14 waves on one CU (end of round 2).
// k_halfstep_de keeps one wave per unit: while the ranks are small (a D_256 run spends most of its sweeps below rank 20)
// a launch has far fewer waves than the chip has SIMDs, and each of them divides at the lone-wave rate of 34 ns per pair.
// The divisions of a unit are independent of each other -- only the running product `a` (and, along one row of the pair
// triangle, the running node product u) is a chain -- so the team splits the roles:
//   wave 1      (U) walks the rows of the triangle and writes u of every bond-spanning pair (one multiply per pair; the node
//                   values come 16 at a time in one register and are picked by DPP row broadcasts),
//   waves 2..13 (D) turn tiles of four consecutive u into the factors ((u-1)/(u+1))^2 in place, stage by stage,
//   wave 0      (M) multiplies the factors into `a` in the reference's order, the tabulated factors of the pivots
//                   (TL before every row, TR at the end) in between -- DPP row broadcasts out of a register that holds 16 of
//                   them instead of LDS broadcasts --, and then finishes the element exactly as k_halfstep_de does (b-part,
//                   weights, store, residual, arg-max record).
// The three roles work on three consecutive chunks of 48 slots (three blocks of 16) in three LDS buffers and meet at ONE
// workgroup barrier per chunk: no flags, no polling, nothing that can hang.  A row of the triangle is padded to whole blocks:
// a padding slot (and the absent first pair of the last row) carries u = 0, whose factor is exactly 1, so M multiplies every
// slot without a test.  Per element every product, difference and quotient and their order are those of k_halfstep_de:
// bit-identical.
// ------------------------------------------------------------------------------------------------------------------
// NBK blocks of 16 slots per chunk, four dividing waves per block: 3 (14 waves, one team per CU) or 1 (6 waves, several teams per CU)
__host__ __device__ inline int det_vs(int m) { return ((m + 7) & ~7) + 8; }
__host__ __device__ inline size_t det_lds_doubles(int m, int nbk) { return (size_t)4 * det_vs(m) + 256 + (det_vs(m) + 48) + (size_t)3 * nbk * 16 * 64; }

#ifdef TTX_STAMPS
#ifndef DET_STAMPG
#define DET_STAMPG 3
#endif
#define DET_T0() long long st_t = wall_clock64()
#define DET_ACC(slot) do { const long long st_n = wall_clock64(); if (st_on && lane == 0) atomicAdd((unsigned long long *)&P.gs[0].stamp[1][slot], (unsigned long long)(st_n - st_t)); st_t = st_n; } while (0)
#else
#define DET_T0() do {} while (0)
#define DET_ACC(slot) do {} while (0)
#endif
template <bool FAST, int NBK>
__global__ __launch_bounds__(64 * (4 * NBK + 2)) void k_halfstep_det(DevProb P, int h, int dir, int mode)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ StepState cur;
    constexpr int ND = 4 * NBK, CHS = ND * 4, NT_ = 64 * (ND + 2);
    // everything that steers the control flow is made wave-uniform explicitly (values read from LDS / global memory are
    // per-lane registers to the compiler: loop counters and branches would otherwise run on the vector unit)
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = UNI(tid >> 6), m = P.d;
    GroupState &gs = P.gs[g];
    if (tid == 0) { cur = gs.S[h]; resolve_state(cur, gs.Pt[(h + 1) & 1]); }
    __syncthreads();
    if (!cur.active || cur.done) { if (blockIdx.x == 0 && tid == 0) gs.S[h + 1] = cur; return; }
    const bool iscol = (mode == 1 || mode == 2) ? (h == 0) : (((h + (dir == 2 ? 1 : 0)) & 1) == 0);    // :517,550
    const int p = UNI(cur.p), r0 = UNI(cur.r0), r1 = UNI(cur.r1), r2 = UNI(cur.r2), n1 = UNI(cur.n1), n2 = UNI(cur.n2), first = UNI(gs.first);
    const int c_ii = UNI(cur.ii), c_jj = UNI(cur.jj), c_kk = UNI(cur.kk), c_qq = UNI(cur.qq);
    const int nf = iscol ? r0 * n1 : n2 * r2;
    const int nv = iscol ? r0 : r2, nm = iscol ? n1 : n2, nch = (nm + 63) >> 6;
    const int npart = nv * nch;
    const int w = blockIdx.x;
    const int crs = UNI(cur.crs) + 1;
    const int havecol = UNI(cur.havecol) | (iscol ? 1 : 0), haverow = UNI(cur.haverow) | (iscol ? 0 : 1);
    const int done = (mode == 1 || mode == 2) ? (h == 1) : (havecol && haverow && (crs >= 2 * P.piv));   // :534 / :567
    const bool resid = (mode == 0) && !done;
    if (npart > (int)gridDim.x) {       // the host sized the grid from its bound on the ranks: never expected; the run is repeated without teams
        if (w == 0 && tid == 0) { atomicAdd(&P.ctl[3], 1); P.ctl[0] = 1; }
        return;
    }
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
    const int pv = w / nch, vmode = (w - pv * nch) * 64 + lane;        // varying pivot, mode index (0-based)
    const bool live = vmode < nm;
    const int A = p - 1, B = m - p - 1;
    const int pl = iscol ? pv : c_ii - 1, qr = iscol ? c_qq - 1 : pv;         // left / right pivot of this unit
    const int n1m = UNI(P.n[1]);
    const double *nodes = P.par, *weights = P.par + n1m;                          // 0-based here
    // rows i = 0..A+1 of the triangle, RL slots each (slot 0: pair with dim p -- absent in row A+1 --, slot 1: with dim p+1,
    // slot 2+j: with right dim j), padded to BPR blocks of 16 slots; chunk k = blocks [k NBK, (k+1) NBK) of the (A+2) BPR blocks
    const int RL = B + 2, BPR = (RL + 15) >> 4, NBT = (A + 2) * BPR, NCH = (NBT + NBK - 1) / NBK;
    // LDS: UL[VS] | xl[VS] | wl[VS] | wr[VS] | ring L[128] | ring R[128] | V[VS+48] (node of slot c of a row: V[2+j] = xr[j], 0 from RL on) | 3 chunks
    const int VS = det_vs(m);
    double *UL = dyn, *xl = UL + VS, *wl = xl + VS, *wr = wl + VS, *ringL = wr + VS, *ringR = ringL + 128, *V = ringR + 128, *xr = V + 2;
    double *chunks = V + VS + 48;
    const size_t NP = (size_t)P.de_npair, tsz = NP * P.RM;
    const double *TLg = P.deTL + (size_t)g * tsz + (size_t)pl * NP, *TRg = P.deTR + (size_t)g * tsz + (size_t)qr * NP;
    const double *ULg = P.deUL + ((size_t)g * P.RM + pl) * (m + 1);
    const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
    for (int x = tid; x < A; x += NT_) { const int ix = Lt[(size_t)x * P.RM + pl] - 1; xl[x] = nodes[ix]; wl[x] = weights[ix]; }
    for (int x = tid; x < B; x += NT_) { const int ix = Rt[(size_t)x * P.RM + qr] - 1; xr[x] = nodes[ix]; wr[x] = weights[ix]; }
    for (int x = RL + tid; x < 16 * BPR + 32; x += NT_) V[x] = 0.0;
    if (tid < 2) V[tid] = 0.0;
    for (int x = tid; x <= A + 2; x += NT_) UL[x] = (x <= A) ? ULg[x] : 1.0;
    const int i1 = iscol ? (live ? vmode : 0) : c_jj - 1, i2 = iscol ? c_kk - 1 : (live ? vmode : 0);   // node index of dim p / p+1
    const double x1 = nodes[i1], x2 = nodes[i2];
    const int n16 = lane & 15;
    WStreamD<2> sl;
    double a = 1.0, u = 0.0, uln = 0.0;
    int ti = 0, bc = 0;                                                // U and M: row / block-in-row of the next block to visit
    int dbc = (wv >= 2) ? ((wv - 2) >> 2) : 0;                         // D: block-in-row of its block of the chunk
    while (dbc >= BPR) dbc -= BPR;
    if (wv == 0) sl.init(TLg, A * (A + 1) / 2, ringL, lane);
    __syncthreads();
    if (wv == 1) uln = UL[0];
    if (wv < 2) __builtin_amdgcn_s_setprio(3);                        // the two serial roles go first on their SIMDs
#ifdef TTX_STAMPS
    const bool st_on = (g == DET_STAMPG % (int)gridDim.y) && w == 0;
    if (st_on && tid == 0) atomicAdd((unsigned long long *)&P.gs[0].nstamp[1], 1ull);
#endif
    DET_T0();
    // one loop per role (the barrier is the same hardware barrier wherever a wave executes it): registers are allocated per role
    if (wv == 1) {
        for (int R = 0; R < NCH + 2; R++) {
            if (R < NCH) {                                             // U: the running node products of chunk R
                // a chunk holds its slots in PAIRS: (slot 2j, slot 2j+1) of lane l at double2 index 64 j + l -- every role moves
                // 16 bytes per lane and LDS instruction (a wave gets ~20 cycles per LDS instruction whatever its width)
                double2 *cb = reinterpret_cast<double2 *>(chunks + (size_t)(R % 3) * CHS * 64) + lane;
                const int nb = (NBT - R * NBK) < NBK ? (NBT - R * NBK) : NBK;
                // all LDS reads of the round first (node values of its three blocks, the next row's start value): one wait per round
                int lbc = bc;
                const double fA = V[lbc * 16 + n16]; if (++lbc == BPR) lbc = 0;
                const double fB = V[lbc * 16 + n16]; if (++lbc == BPR) lbc = 0;
                const double fC = V[lbc * 16 + n16];
                const double ulA = UL[ti + 1], ulB = UL[ti + 2], ulC = UL[ti + 3];
                const int ti0 = ti;
                for (int kb = 0; kb < nb; kb++) {
                    const double f = (kb == 0) ? fA : (kb == 1) ? fB : fC;
                    const bool head = (bc == 0), lastrow = (ti == A + 1);
                    double2 *out = cb + (size_t)kb * 8 * 64;
                    if (head) { u = uln; uln = (ti == ti0) ? ulA : (ti == ti0 + 1) ? ulB : ulC; }   // UL[A+1] = 1: the row that starts after dim p
                    // the broadcasts of half a block are named values taken before its eight dependent multiplies (written inline the
                    // compiler funnels them through one temporary: move, multiply, move, multiply ... each waiting for the other)
                    const double c0 = rowbc<0>(f), c1 = rowbc<1>(f), c2 = rowbc<2>(f), c3 = rowbc<3>(f), c4 = rowbc<4>(f), c5 = rowbc<5>(f), c6 = rowbc<6>(f), c7 = rowbc<7>(f);
                    const double m0 = head ? (lastrow ? 1.0 : x1) : c0;
                    const double m1 = head ? x2 : c1;
                    // the dividers of the six-slot tiles continue the chain themselves from slot 3 and slot 9, which U leaves in the
                    // first pair of THEIR tile (pairs 2 and 5; the tiles of other waves are overwritten in the same round).  The four
                    // stores go out together at the end of the block from registers of their own: a multiply that overwrites the
                    // data register of an LDS store still in flight waits for it (~20 cycles per store, profiles/r02_probe_dpp.txt)
                    u = u * m0; const double s0 = (head && lastrow) ? 0.0 : u;
                    u = u * m1; const double s1 = u;
                    u = u * c2; const double s2 = u;
                    u = u * c3; const double s3 = u;
                    u = u * c4; u = u * c5; u = u * c6; u = u * c7; const double s7 = u;
                    const double d0 = rowbc<8>(f), d1 = rowbc<9>(f), d2 = rowbc<10>(f), d3 = rowbc<11>(f), d4 = rowbc<12>(f), d5 = rowbc<13>(f), d6 = rowbc<14>(f), d7 = rowbc<15>(f);
                    u = u * d0; u = u * d1; const double s9 = u;
                    u = u * d2; u = u * d3; const double s11 = u;
                    u = u * d4; u = u * d5; u = u * d6; u = u * d7;
                    out[0] = make_double2(s0, s1); out[64] = make_double2(s2, s3); out[128] = make_double2(s3, 0.0);
                    if (NBK == 1) { out[256] = make_double2(s7, 0.0); out[384] = make_double2(s11, 0.0); }      // four equal tiles (below)
                    else out[320] = make_double2(s9, 0.0);
                    if (++bc == BPR) { bc = 0; ti++; }
                }
            }
            DET_ACC(3);
            __syncthreads();
            DET_ACC(4);
        }
        return;
    }
    if (wv >= 2) {
        for (int R = 0; R < NCH + 2; R++) {
            // waves go round-robin to the four SIMDs: waves 4k and 4k+1 share theirs with M and U and take two slots of a
            // block, waves 4k+2 and 4k+3 six (blocks of 16 slots = 2 + 2 + 6 + 6)
            const int dv = wv - 2, bq = dv >> 2, role = wv & 3;         // role 0/1: pair 0 / 1 ; role 2/3: pairs 2-4 / 5-7 of the block
            const int blk = (R - 1) * NBK + bq;
            if (R >= 1 && blk < NBT) {                                 // D: its slots of block bq of chunk R-1, in place
                double2 *cb = reinterpret_cast<double2 *>(chunks + ((size_t)((R - 1) % 3) * CHS + (size_t)bq * 16) * 64) + lane;
                if (NBK == 1) {
                    // six-wave teams share their CU with other teams, no wave has a SIMD to itself: four equal tiles of four slots;
                    // tile 0 takes its two pairs from U, the others continue the chain from slot 3 / 7 / 11 (first pair of their tile)
                    double2 *cp = cb + (size_t)(2 * role) * 64;
                    double uu[4], t[4];
                    if (role == 0) { const double2 v0 = cp[0], v1 = cp[64]; uu[0] = v0.x; uu[1] = v0.y; uu[2] = v1.x; uu[3] = v1.y; }
                    else {
                        const double fV = V[dbc * 16 + n16];
                        const double us = cp[0].x;
                        if (role == 1) { uu[0] = us * rowbc<4>(fV); uu[1] = uu[0] * rowbc<5>(fV); uu[2] = uu[1] * rowbc<6>(fV); uu[3] = uu[2] * rowbc<7>(fV); }
                        else if (role == 2) { uu[0] = us * rowbc<8>(fV); uu[1] = uu[0] * rowbc<9>(fV); uu[2] = uu[1] * rowbc<10>(fV); uu[3] = uu[2] * rowbc<11>(fV); }
                        else { uu[0] = us * rowbc<12>(fV); uu[1] = uu[0] * rowbc<13>(fV); uu[2] = uu[1] * rowbc<14>(fV); uu[3] = uu[2] * rowbc<15>(fV); }
                    }
                    de_t2xw<FAST, 4>(uu, t);
                    cp[0] = make_double2(t[0], t[1]); cp[64] = make_double2(t[2], t[3]);
                } else if (role < 2) {
                    double2 *cp = cb + (size_t)role * 64;
                    const double2 v = cp[0];
                    const double uu[2] = {v.x, v.y}; double t[2];
                    de_t2xw<FAST, 2>(uu, t);
                    cp[0] = make_double2(t[0], t[1]);
                } else {
                    // the running products of its six slots from the last one U left behind (slot 3 / slot 9, in the first pair of its own tile),
                    // by the node values of this block -- the same multiplications U performs for its own chain
                    const double fV = V[dbc * 16 + n16];
                    double2 *cp = cb + (size_t)(2 + (role - 2) * 3) * 64;
                    double uu[6], t[6];
                    if (role == 2) {
                        const double us = cp[0].x;                     // slot 3
                        uu[0] = us * rowbc<4>(fV); uu[1] = uu[0] * rowbc<5>(fV); uu[2] = uu[1] * rowbc<6>(fV);
                        uu[3] = uu[2] * rowbc<7>(fV); uu[4] = uu[3] * rowbc<8>(fV); uu[5] = uu[4] * rowbc<9>(fV);
                    } else {
                        const double us = cp[0].x;                     // slot 9
                        uu[0] = us * rowbc<10>(fV); uu[1] = uu[0] * rowbc<11>(fV); uu[2] = uu[1] * rowbc<12>(fV);
                        uu[3] = uu[2] * rowbc<13>(fV); uu[4] = uu[3] * rowbc<14>(fV); uu[5] = uu[4] * rowbc<15>(fV);
                    }
                    de_t2xw<FAST, 6>(uu, t);
                    cp[0] = make_double2(t[0], t[1]); cp[64] = make_double2(t[2], t[3]); cp[128] = make_double2(t[4], t[5]);
                }
            }
            if (R >= 1) { dbc += NBK; while (dbc >= BPR) dbc -= BPR; }
            if (wv == 2) DET_ACC(5);
            __syncthreads();
            if (wv == 2) DET_ACC(6);
        }
        return;
    }
    for (int R = 0; R < NCH + 2; R++) {
        if (R >= 2) {                                                  // M: the factors of chunk R-2 into `a`, in order
            const double2 *cb = reinterpret_cast<const double2 *>(chunks + (size_t)((R - 2) % 3) * CHS * 64) + lane;
            const int nb = (NBT - (R - 2) * NBK) < NBK ? (NBT - (R - 2) * NBK) : NBK;
            // four buffers of half a block: the first four are loaded at the top of the round (under the tabulated run of a row
            // head, if there is one), the last two 16 multiplies ahead -- the LDS answers slowly while 14 waves use it
            double q0[8], q1[8], q2[8], q3[8];
#define DET_LD(q, hb) _Pragma("unroll") for (int x = 0; x < 4; x++) { const double2 v = cb[(size_t)((hb) * 4 + x) * 64]; q[2 * x] = v.x; q[2 * x + 1] = v.y; }
#define DET_F(q) _Pragma("unroll") for (int x = 0; x < 8; x++) a = a * q[x];
#define DET_HEAD() if (bc == 0 && ti <= A) { DET_ACC(1); a = sl.chain(a, A - ti, lane); DET_ACC(0); }
#define DET_NEXT() if (++bc == BPR) { bc = 0; ti++; }
            DET_LD(q0, 0) DET_LD(q1, 1) if (NBK > 1) { DET_LD(q2, 2) DET_LD(q3, 3) }
            DET_HEAD() DET_F(q0) if (NBK > 2) { DET_LD(q0, 4) } DET_F(q1) if (NBK > 2) { DET_LD(q1, 5) } DET_NEXT()
            if (NBK > 1 && nb > 1) { DET_HEAD() DET_F(q2) DET_F(q3) DET_NEXT() }
            if (NBK > 2 && nb > 2) { DET_HEAD() DET_F(q0) DET_F(q1) DET_NEXT() }
#undef DET_LD
#undef DET_F
#undef DET_HEAD
#undef DET_NEXT
            DET_ACC(1);
        }
        __syncthreads();
        DET_ACC(2);
    }
    WStreamD<6> sr;                                                    // alone on the CU by now: six batches (768 factors) in flight
    sr.init(TRg, B * (B + 1) / 2, ringR, lane);
    a = sr.chain_from_boundary(a, B * (B + 1) / 2, lane);
    DET_ACC(8);
    const double w1 = weights[i1], w2 = weights[i2];
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
    // ---- fiber store, amax, residual, arg-max: as k_halfstep_de ----
    const int u_ = iscol ? pv : vmode, v_ = iscol ? vmode : pv;        // col: (i, j) ; row: (k, q), 0-based
    const int t = iscol ? (u_ + r0 * v_) : (u_ + n2 * v_);
    if (live) (iscol ? P.acol : P.arow)[(size_t)g * P.RM * P.NM + t] = a;
    const double mx = wave_max(live ? fabs(a) : 0.0);
    if (lane == 0 && mode != 1) atomic_max_pos(&gs.amax, mx);          // :531 / :564
    if (resid) {
        const double *Cp = core_ptr(P, P.col, g, p, first), *Wq = core_ptr(P, P.row, g, p + 1, first);
        double bb = a, ab = -1.0; int bi = INT_MAX;
        if (live) {
            if (iscol) {   // dgemv 'n', alpha=-1 (:538)
                const double *c = Cp + u_ + (size_t)P.RM * v_;
                const double *xq = Wq + (c_kk - 1) + (size_t)P.NM * (c_qq - 1);
#pragma unroll 8
                for (int s = 0; s < r1; s++) bb = bb + (-xq[P.SW * s]) * c[P.SS * s];
            } else {       // dgemv 't', alpha=-1 (:571)
                const double *wvp = Wq + u_ + (size_t)P.NM * v_;
                const double *xc = Cp + (c_ii - 1) + (size_t)P.RM * (c_jj - 1);
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
    DET_ACC(9);
}
#undef UNI
