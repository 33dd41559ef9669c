// ttx_cdf.h -- exact O(#binades) evaluation of the lottery's cumulative weights (host + device).
//
// lottery2 (reference lib/rnd.f90:116-119) builds pcol(i) = pcol(i-1) + |w_i|/sum(w) by SEQUENTIAL fp64
// accumulation and bisects it with find_d (lib/rnd.f90:128-143).  In dtt_dmrgg the weights are 1 except 0
// at the rows/columns that already hold a pivot (lib/dmrgg.f90:425-439), so the sequence over the K
// non-zero weights is a_k = fl(a_{k-1} + c), a_0 = 0, c = fl(1/K).  A serial K-step chain (K up to r*n =
// 6464) would sit on the critical path of every bond step, so it is evaluated in closed form per binade:
// inside one binade [2^e, 2^(e+1)) every a_k is a multiple of u = 2^(e-52) and fl(a + c) = a + delta with
// a CONSTANT delta once two additions have been made inside the binade (round-to-nearest-even: a tie
// c/u = q + 1/2 yields even multiples from the first in-binade addition on, after which the increment is
// the even one of q, q+1).  Real fp64 additions are made only for the <= 3 values after each binade
// crossing; the rest are arithmetic progressions.  tests/test_cdf.py checks this against the plain
// sequential loop for every K in 1..8192.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define TTX_HD __host__ __device__ inline
#else
#define TTX_HD static inline
#endif

#define TTX_MAXSEG 128

struct ttx_cdfseg {
    double a0;     // a_{k0}
    double alast;  // a_{k0+cnt-1}
    double delta;  // a_{k0+t} = a0 + t*delta (exact), t in [0, cnt)
    int32_t k0;
    int32_t cnt;
};

TTX_HD uint64_t ttx_dbits(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
TTX_HD double ttx_bitsd(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }
TTX_HD int ttx_bexp(double x) { return (int)((ttx_dbits(x) >> 52) & 0x7ff); }  // biased exponent
// x / 2^(be-1075) as integer, x a positive normal multiple of ulp(be)
TTX_HD int64_t ttx_units(double x, int be)
{
    uint64_t b = ttx_dbits(x);
    int xe = (int)((b >> 52) & 0x7ff);
    int64_t mant = (int64_t)((b & 0xFFFFFFFFFFFFFULL) | (1ULL << 52));
    return (xe >= be) ? (mant << (xe - be)) : (mant >> (be - xe));
}
TTX_HD double ttx_from_units(int64_t v, int be)
{
    // v * 2^(be-1075), v < 2^54, result normal
    double d = (double)v;                        // exact for v < 2^53; v == 2^53 exact too
    uint64_t b = ttx_dbits(d);
    int de = (int)((b >> 52) & 0x7ff);           // biased exponent of (double)v : value = mant*2^(de-1075)
    b = (b & ~(0x7ffULL << 52)) | ((uint64_t)(de + be - 1075) << 52);
    return ttx_bitsd(b);
}

// floor(N / D) for 0 <= N < 2^54, 0 < D < 2^54 without 64-bit integer division (slow on the device): fp64
// quotient, then exact correction with integer multiplies
TTX_HD int64_t ttx_divfloor(int64_t N, int64_t D)
{
    int64_t q = (int64_t)((double)N / (double)D);
    while (q * D > N) q--;
    while ((q + 1) * D <= N) q++;
    return q;
}

// segments covering k = 1..K of a_k = fl(a_{k-1} + c), a_0 = 0, c = 1.0/K; returns their number
TTX_HD int ttx_cdf_build(int K, ttx_cdfseg *seg)
{
    if (K <= 0) return 0;
    const double c = 1.0 / (double)K;
    double a = 0.0, am1 = 0.0;                   // a_k, a_{k-1}
    int k = 0, ns = 0, inb = 0, lastexp = -1;
    while (k < K && ns < TTX_MAXSEG - 2) {
        double x = a + c;                        // the real sequential step
        k++;
        int xe = ttx_bexp(x);
        inb = (xe == lastexp) ? inb + 1 : 1;     // consecutive values inside binade xe
        lastexp = xe;
        seg[ns].a0 = x; seg[ns].alast = x; seg[ns].delta = 0.0; seg[ns].k0 = k; seg[ns].cnt = 1; ns++;
        am1 = a; a = x;
        if (inb >= 3 && k < K) {
            // a_{k-2}, a_{k-1}, a_k share the binade: the increment D = a_k - a_{k-1} is steady until 2^(e+1)
            int64_t Z = ttx_units(a, xe), D = Z - ttx_units(am1, xe);
            const int64_t B = (int64_t)1 << 53;  // 2^(e+1) in units of u = 2^(e-52)
            if (D > 0) {
                int64_t tmax = ttx_divfloor(B - Z - 1, D);  // number of t >= 1 with Z + t*D < B
                if (tmax > (int64_t)(K - k)) tmax = K - k;
                if (tmax > 0) {
                    seg[ns].a0 = ttx_from_units(Z + D, xe);
                    seg[ns].alast = ttx_from_units(Z + tmax * D, xe);
                    seg[ns].delta = ttx_from_units(D, xe);
                    seg[ns].k0 = k + 1; seg[ns].cnt = (int32_t)tmax; ns++;
                    am1 = ttx_from_units(Z + (tmax - 1) * D, xe);
                    a = ttx_from_units(Z + tmax * D, xe);
                    k += (int)tmax;
                }
            }
        }
    }
    return ns;
}

// largest k in [0, K] with a_k <= y  (binary search over the segments, then exact integer arithmetic)
TTX_HD int ttx_cdf_kmax(const ttx_cdfseg *seg, int ns, double y)
{
    int lo = -1, hi = ns;                         // last segment with a0 <= y
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (seg[mid].a0 <= y) lo = mid; else hi = mid; }
    if (lo < 0) return 0;
    const ttx_cdfseg sg = seg[lo];
    if (sg.cnt == 1 || y >= sg.alast) return sg.k0 + sg.cnt - 1;
    int be = ttx_bexp(sg.a0);
    int64_t A = ttx_units(y, be) - ttx_units(sg.a0, be);   // y in [a0, alast): same binade, exact
    int64_t D = ttx_units(sg.delta, be);
    return sg.k0 + (int)ttx_divfloor(A, D);
}

// 1-based position of the kth (1-based) non-zero weight, given the ascending distinct 1-based zero positions:
// kth + #{t : zeros[t] - t <= kth} (zeros[t] - t is non-decreasing, so the count is a binary search)
TTX_HD int ttx_select_nonzero(int kth, const int32_t *zeros, int nz)
{
    int lo = 0, hi = nz;                          // first t with zeros[t] - t > kth
    while (lo < hi) { int mid = (lo + hi) >> 1; if (zeros[mid] - mid <= kth) lo = mid + 1; else hi = mid; }
    return kth + lo;
}

// lottery index (1-based in 1..m) for uniform draw y: lib/rnd.f90:122-123 incl. the clamp to m
TTX_HD int ttx_lottery_index(const ttx_cdfseg *seg, int ns, int K, int m, const int32_t *zeros, int nz, double y)
{
    int kmax = ttx_cdf_kmax(seg, ns, y);
    if (kmax >= K) return m;
    return ttx_select_nonzero(kmax + 1, zeros, nz);
}

// minstd power 48271^e mod (2^31-1)
TTX_HD uint64_t ttx_minstd_pow(uint64_t e)
{
    uint64_t base = 48271ULL, w = 1;
    // (mulmod31 is declared below; forward use is fine inside the header's single translation unit)
    while (e) {
        if (e & 1) { uint64_t x = w * base; x = (x & 2147483647ULL) + (x >> 31); x = (x & 2147483647ULL) + (x >> 31); w = (x >= 2147483647ULL) ? x - 2147483647ULL : x; }
        uint64_t y = base * base; y = (y & 2147483647ULL) + (y >> 31); y = (y & 2147483647ULL) + (y >> 31); base = (y >= 2147483647ULL) ? y - 2147483647ULL : y;
        e >>= 1;
    }
    return w;
}
// the double made from the generator word w1 = 48271^(2k+1) (and its successor), as ttx_flang_draw(k)
TTX_HD double ttx_flang_from_word(uint64_t w1)
{
    uint64_t x = w1 * 48271ULL; x = (x & 2147483647ULL) + (x >> 31); x = (x & 2147483647ULL) + (x >> 31);
    uint64_t w2 = (x >= 2147483647ULL) ? x - 2147483647ULL : x;
    uint64_t f = ((w1 << 30) | ((w2 - 1) & ((1ULL << 30) - 1))) >> 7;
    return (double)f * 5.5511151231257827e-17;
}

// flang run-time random_number (unseeded): minstd 48271 mod 2^31-1 from seed 1, two words per double
TTX_HD uint64_t ttx_mulmod31(uint64_t a, uint64_t b)
{   // (a*b) mod (2^31-1) for a,b < 2^31, by Mersenne folding (no 64-bit division on the device)
    uint64_t x = a * b;
    x = (x & 2147483647ULL) + (x >> 31);
    x = (x & 2147483647ULL) + (x >> 31);
    return (x >= 2147483647ULL) ? x - 2147483647ULL : x;
}
TTX_HD double ttx_flang_draw(uint64_t k)
{
    uint64_t e = 2 * k + 1, base = 48271ULL, w1 = 1;
    while (e) { if (e & 1) w1 = ttx_mulmod31(w1, base); base = ttx_mulmod31(base, base); e >>= 1; }
    uint64_t w2 = ttx_mulmod31(w1, 48271ULL);
    uint64_t f = ((w1 << 30) | ((w2 - 1) & ((1ULL << 30) - 1))) >> 7;   // 54 significant bits
    // (double)f * 2^-54 with round-to-nearest-even of the 54-bit integer, as ldexp((double)f, -54)
    return (double)f * 5.5511151231257827e-17;
}
