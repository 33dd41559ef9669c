// ttx_cluster.h -- the whole sweep of one bond group in ONE launch by a CLUSTER of workgroups (Ising C fast path).
//
// The multi-kernel path pays six dependent launches per bond step and the single-workgroup kernel (ttx_fused.h)
// pushes every fiber through one CU.  Here NB workgroups of 256 threads share a bond group and stay resident for
// the whole sweep (lib/dmrgg.f90:329-760):
//   * the blocks of a group are placed on ONE XCD (workgroups are dealt round-robin to the 8 XCDs -- checked by
//     ttx_k_xcc_map -- so block ids g, g+8, g+16, ... share an L2): what they exchange stays in that L2;
//   * each block owns a slice of the mode index (j of the column fiber, k of the row fiber): it evaluates, keeps
//     (LDS) and later appends exactly that slice, so fibers never travel between blocks;
//   * the small serial parts (lottery draw and its arg-max, acceptance test, pivot lists) are computed redundantly and
//     identically by every block, so the only cross-block traffic per rook half-step is ONE tagged 32-byte record per
//     block (its partial arg-max), written with one 16-byte store pair and polled with L1-bypassing loads -- no fence,
//     no counter; once per bond step a counter barrier with release/acquire publishes the appended slabs;
//   * everything that is fixed during a bond step is staged in LDS once: node/weight VALUES of both pivot sets, the
//     prefix states of the integrand's two running sums per pivot row, the neighbour LU factors, the factor rows at
//     the current pivot, and -- for the whole launch -- the sorted pivot lists of every own bond;
//   * every wait is bounded: a block that waits too long raises the abort flag (pinned host memory), all blocks leave,
//     and the host reports the failure instead of hanging the GPU.
// All blocks of all groups must be resident at once.  The host takes this path only when the grid fits HALF of what the
// occupancy calculator says the device can hold for this kernel's LDS footprint (room for the forked quadrature and a
// second engine); TTX_CLUSTER_COOP=1 additionally launches it as a COOPERATIVE kernel (the runtime itself then refuses a
// grid that cannot be co-resident; ~30 us per launch dearer).  Should a wait still time out, the kernel stops every later
// kernel of the sweep (ctl[0]) and ttx_run replays the run on the multi-kernel chain, which gives the identical result
// (ttx_cluster_fallbacks).
// Arithmetic and its order are identical to the other two paths (and to the oracle): same device functions, the
// first-max rule is applied on global fiber positions, reused partial results are bit-identical by construction.
#pragma once
#include "ttx_fused.h"
#include "ttx_mvn.h"       // mvn_lane: one double of a wave by v_readlane

#define CB 256      // threads of a cluster workgroup
#ifndef TTX_CL_CF
#define TTX_CL_CF 0    // factor entries of the residual requested before the evaluation (measured: 16 -> +1.4 % run time, registers)
#endif
#define F_ISING_CL f_ising_c4p
#ifndef TTX_CL_UNR
#define TTX_CL_UNR 16   // factor loads of the residual in flight per batch (8: +0.6 % run time, 32: no further gain)
#endif

// Ising C integrand from VALUE rows, with the two running sums resumed from per-row prefix states: (pv, pvk) is the
// state of the descending sum after the right row bn (dims A+3..m), (pw, pwk) the state of the ascending sum after
// the left row an (dims 1..A).  The states are produced by exactly the operations f_ising_c4v would perform, so the
// result is bit-identical; per element this saves one of the three passes over the m dimensions.
__device__ __forceinline__ double f_ising_c4p(int m, int A, const double *an, const double *aw, double s1n, double s1w,
                                              double s2n, double s2w, const double *bn, const double *bw,
                                              double pv, double pvk, double pw, double pwk)
{
    const int nb = m - A - 2;
    double v = pv, w = pw, vk = pvk, wk = pwk;
    auto vstep = [&](double xv) { vk = vk * xv; v = v + vk; };
    auto wstep = [&](double xv) { wk = wk * xv; w = w + wk; };
#ifndef TTX_CL_PIPE
#define TTX_CL_PIPE 0      // 1: chain8p (LDS reads one chunk ahead of the dependent steps; measured 4 % slower: registers), 0: chain8v
#endif
#if TTX_CL_PIPE
#define CHAIN8 chain8p
#else
#define CHAIN8 chain8v
#endif
    vstep(s2n); vstep(s1n);
    CHAIN8<true>(an, A, vstep);
    wstep(s1n); wstep(s2n);
    CHAIN8<false>(bn, nb, wstep);
    double b = 1.0 / (v * w);
    double f = 2 * b;
    auto fstep = [&](double xv) { f = f * xv; };
    CHAIN8<false>(aw, A, fstep);
    fstep(s1w); fstep(s2w);
    CHAIN8<false>(bw, nb, fstep);
    return f;
}

// arg-max over a wave by the first-max rule (larger |.| wins, ties go to the lower position) in two cheap passes on the DPP
// path: the maximum of |.| (an fp64 max per step), then the smallest (position, sign) key among the lanes that hold it (an
// integer min per step) -- a third of the instructions of the three-operand compare-and-select of wave_argmax.  a is never NaN
// (a NaN residual never replaces the start value -1), so fmax and == are exact here.  Result in every lane.
__device__ __forceinline__ int wave_min_i(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0xb1, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x4e, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x124, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x128, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ void wave_argmax2(double &a, double &v, int &idx)
{
    const double amx = wave_max(a);
    const int key = (a == amx && idx != INT_MAX) ? ((idx << 1) | (signbit(v) ? 1 : 0)) : INT_MAX;
    const int kmin = wave_min_i(key);
    a = amx;
    if (kmin == INT_MAX) { idx = INT_MAX; v = 0.0; }
    else { idx = kmin >> 1; v = (kmin & 1) ? -amx : amx; }
}

// TTX_ARITH=fast for Ising C: the two running sums of test_crs_ising.f90:197-204 are affine in their start state, so with the
// prefix states of the pivot rows (pv, pvk after the right row; pw, pwk after the left row), the sums SL / SR of the partial
// products that start at the bond and the weight products WL / WR an element is a closed form of its two free nodes:
//   v = pv + pvk xk (1 + xj (1 + SL)),   w = pw + pwk xj (1 + xk (1 + SR)),   f = 2 / (v w) * WL wj wk WR
// -- fifteen operations instead of a dependent chain of 2 m fed from LDS; equal to the exact value to rounding.
__device__ __forceinline__ double f_ising_cfast(double xj, double wj, double xk, double wk, double pv, double pvk, double pw, double pwk,
                                                double SL, double WL, double SR, double WR)
{
    const double v = pv + pvk * xk * (1.0 + xj * (1.0 + SL));
    const double w = pw + pwk * xj * (1.0 + xk * (1.0 + SR));
    return 2.0 / (v * w) * ((WL * wj) * (wk * WR));
}

// 16-byte records exchanged between the blocks of a cluster: one store / one load instruction each, agent scope
// (sc1: the store goes through to the point of coherence, the load does not hit in the vector L1)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
struct ClRec { u4 a, b; };
static_assert(sizeof(ClRec) == sizeof(ClPart), "record buffers are sized as ClPart: [2][G][TTX_CLREC] u4 arg-max records, then as many running-maximum records");
__device__ __forceinline__ void st16(void *p, u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u4 ld16(const void *p)
{
    u4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// XCC_ID hardware register (id 20, 4 bits): the XCD this wave runs on
__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 15; }
__global__ void k_xcc_probe(int *out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// Barrier across the workgroups of one cluster with release/acquire semantics for ordinary global data; false: timed
// out / aborted (every thread of the block gets the same answer).  Every wave first waits for its own stores to be
// acknowledged by the L2; after the block barrier ONE wave performs the agent-scope release (L2 write-back), the
// counter increment, the bounded spin and the agent-scope acquire: cache maintenance acts on the CU's L1 and the XCD's
// L2, not on a wave, so repeating it per wave only multiplies its cost (measured: 12.0 -> 9.0 ms per C_64 run).
__device__ __forceinline__ bool cluster_sync(unsigned *ctr, unsigned target, int *abortflag, int *s_ok)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned spins = 0;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = 0; __hip_atomic_store(abortflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            if ((spins & 1023u) == 0 && __hip_atomic_load(abortflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { ok = 0; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

#ifdef TTX_STAMPS
#define CST_DECL const bool t_me = (threadIdx.x == 0 && g == 0 && cb == 0); long long t_prev = wall_clock64()
#define CST(k) do { if (t_me) { long long t_now = wall_clock64(); gs.stamp[0][k] += t_now - t_prev; t_prev = t_now; } } while (0)
#define CST_END() do { if (t_me) gs.nstamp[0]++; } while (0)
// per-wave timeline of ONE bond step (launch 8, step 4, group 0): slot k of half-step h of wave (cb, wv)
#define WST(k) do { if (P.dbg && g == 0 && epoch == 8 && pp == 4 && lane == 0 && h < 8) P.dbg[((size_t)(h * 64 + cb * 4 + wv)) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define WST(k)
#define CST_DECL
#define CST(k)
#define CST_END()
#endif

__global__ __launch_bounds__(CB) void k_sweep_cluster(DevProb P, int dir, int nsteps, int NB, int ldsinv, int epoch, int zkeep)
{
    extern __shared__ __align__(16) double dyn[];
    __shared__ int zc[128], zr[128], zcs[128], zrs[128], keepc[128], keepr[128];
    __shared__ int nzc, nzr, nsc, nsr, s_ok;
    __shared__ ttx_cdfseg segc[TTX_TABSEG], segr[TTX_TABSEG];
    __shared__ double sha[8], shv[8], shm[8]; __shared__ int shi[8];
    __shared__ double pLw[64], pLk[64], pRv[64], pRk[64];   // prefix states of the two running sums per left / right row
    // TTX_ARITH=fast: per pivot row the sum of the partial products that start at the bond (left: descending, right: ascending) and
    // the product of its weights -- with them the integrand is a closed form of the two free nodes (f_ising_cfast)
    __shared__ double fSL[64], fWL[64], fSR[64], fWR[64];
    const bool cfast = P.arith != 0;
    // results of the rook loop, handed from wave 0 to the waves of the workgroup that own no fiber element at the current ranks
    __shared__ struct { int ii, jj, kk, qq, hcount, rc_k, rc_q, rr_i, rr_j; double pivot, amax, bytes_half; long long neval, n_resid; } s_rook;
    const int bid = blockIdx.x;
    const int g = (bid & 7) + 8 * (bid / (8 * NB)), cb = (bid >> 3) % NB;     // cluster of group g lives on XCD g % 8
    if (g >= P.G || P.ctl[0]) return;
    if (P.cl_test_abort && P.cl_test_abort == epoch && g == 0 && cb == 0) {    // test hook (TTX_CLUSTER_TEST_ABORT): a block that never arrives
        if (threadIdx.x == 0) { __hip_atomic_store(P.cl_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); P.ctl[0] = 1; }
        return;
    }
    const int tid = threadIdx.x, m = P.d, RM = P.RM, NM = P.NM;
    const int lane = tid & 63, wv = tid >> 6;
    GroupState &gs = P.gs[g];
    const int first = gs.first, last = gs.last, nbonds = last - first + 1;
    int *r = P.r + (size_t)g * (m + 2);
    const int VS = ((m + 7) & ~7) + 8;
    // row stride of the value tables: 2*VS doubles would put every row on the same LDS banks (a multiple of 128 bytes), and
    // a column half-step reads 32 DIFFERENT rows with one 16-byte load per lane -- a 32-way bank conflict on every load
    // of the integrand chain.  Two extra doubles shift consecutive rows by 16 bytes: 8 rows tile the 32 banks.
    const int RS = 2 * VS + 2;
    const int n1m = P.n[1];
    const int SL = RM * ((NM + NB - 1) / NB + 1);       // most fiber entries one block owns
    unsigned *ctr = P.cl_ctr + g;
    // the barrier counter and the record tags are never reset: launch number `epoch` (1, 2, ...) of a run starts from
    // where launch epoch-1 stopped (every launch passes exactly one counter barrier per own bond)
    unsigned nbar = (unsigned)(epoch - 1) * (unsigned)min(nsteps, nbonds);
    ClPart *part = P.cl_part;
    // LDS carve-up
    double *par = dyn;
    double *XL = par + ((P.npar + 1) & ~1);            // RM rows x (VS node values, VS weight values)
    double *XR = XL + (size_t)RM * RS;
    double *acol = XR + (size_t)RM * RS;           // own slice of the column fiber
    double *arow = acol + SL;                          // own slice of the row fiber
    double *resc = arow + SL;                          // residuals of the own column / row slice at the last half-step
    double *resr = resc + SL;                          // that computed them (reused by the append, roles A and B)
    double *xsc = resr + SL;                           // RM: row-factor entries at the pivot column (kk, qq) ...
    double *xsr = xsc + ((RM + 1) & ~1);               // RM: column-factor entries at the pivot row (ii, jj)
    int *lot = (int *)(xsr + ((RM + 1) & ~1));         // 4 * nlotmax
    double *GL = (double *)(lot + 4 * ((2 * RM + 2 * NM + 4 + 1) & ~1));   // packed LU of bonds p-1 and p+1 (ldsinv only)
    double *GU = GL + (size_t)RM * RM;
    // zkeep: the sorted distinct pivot rows / columns of EVERY own bond stay in LDS for the whole launch as keys
    // (j << 16 | i), (q << 16 | k): built once, extended by insertion when a pivot is accepted.  Their order does not
    // depend on the ranks, only the flattened positions do (recomputed per bond step).
    int *ZK = (int *)(ldsinv ? GU + (size_t)RM * RM : GL);   // [nsteps][2][RM]
    int *ZN = ZK + (size_t)nsteps * 2 * RM;                  // [nsteps][2]
    for (int x = tid; x < P.npar; x += CB) par[x] = P.par[x];
    if (cb == 0 && tid == 0) {                         // sweep start, :325-327
        int *rr = P.rr + (size_t)g * (m + 2);
        for (int s = 0; s <= m; s++) rr[s] = r[s];
    }
    __syncthreads();
    if (zkeep) {
        int *TMP = (int *)XL;                          // XL is staged later; free scratch for now: [nsteps][2][RM]
        const int E = nbonds * 2 * RM;
        for (int x = tid; x < E; x += CB) {            // raw keys
            const int b = x / (2 * RM), side = (x / RM) & 1, u = x % RM, pb = first + b;
            if (u < r[pb]) { const int *vp = vip_ptr(P, g, pb, first) + 4 * u; TMP[x] = (vp[2 * side + 1] << 16) | vp[2 * side]; }
        }
        __syncthreads();
        for (int x = tid; x < E; x += CB) {            // rank among the bond's keys (ties by position), scatter
            const int b = x / (2 * RM), u = x % RM, nb_ = r[first + b];
            if (u < nb_) {
                const int *src = TMP + (x - u); const int a = src[u]; int ra = 0;
                for (int t = 0; t < nb_; t++) ra += (src[t] < a) || (src[t] == a && t < u);
                ZK[(x - u) + ra] = a;
            }
        }
        __syncthreads();
        for (int x = tid; x < E; x += CB) {            // drop repeats: position among the kept ones
            const int b = x / (2 * RM), u = x % RM, nb_ = r[first + b];
            if (u < nb_) {
                const int *src = ZK + (x - u); int pc = 0;
                for (int t = 1; t <= u; t++) pc += (src[t] != src[t - 1]);
                const bool keep = (u == 0) || (src[u] != src[u - 1]);
                TMP[x] = keep ? pc : -1;
                if (u == nb_ - 1) ZN[x / RM] = pc + 1;
            }
        }
        __syncthreads();
        int mykey = 0, mypos = -1;
        for (int x0 = 0; x0 < E; x0 += CB) {
            const int x = x0 + tid;
            mypos = -1;
            if (x < E) { const int b = x / (2 * RM), u = x % RM; if (u < r[first + b]) { mypos = TMP[x]; mykey = ZK[x]; } }
            __syncthreads();
            if (mypos >= 0) ZK[(x - x % RM) + mypos] = mykey;
            __syncthreads();
        }
    }
    double amax = gs.amax, pivotmax = -1.0, pivotmin = -1.0;
    const double pivotmax_prev = gs.pivotmax_prev;
    long long neval = gs.neval;
    unsigned long long rngpos = gs.rngpos;
    double bytes_half = gs.bytes_half; long long n_resid = gs.n_resid;
    int hcount = 0;                                    // half-steps exchanged so far (selects the record buffer)
    const unsigned long long bil0 = ttx_minstd_pow(2ull * tid);     // RNG jump of this thread's first lottery candidate
    unsigned long long sA0 = ttx_minstd_pow(2 * rngpos + 1);
    CST_DECL;

    for (int pp = 1; pp <= nsteps; pp++) {
        if (pp > nbonds) break;
        const int p = (dir == 1) ? first + pp - 1 : last + 1 - pp;          // :330-331
        const int r0 = r[p - 1], r1 = r[p], r2 = r[p + 1], n1 = P.n[p], n2 = P.n[p + 1];
        const int nlot = r0 + n1 + n2 + r2;
        const int jlo = (int)((long long)cb * n1 / NB), jhi = (int)((long long)(cb + 1) * n1 / NB);   // own columns j of acol1
        const int klo = (int)((long long)cb * n2 / NB), khi = (int)((long long)(cb + 1) * n2 / NB);   // own rows k of arow1
        const int nj = jhi - jlo, nk = khi - klo;
        double *Cp = core_ptr(P, P.col, g, p, first), *Wq = core_ptr(P, P.row, g, p + 1, first);
        double *Ap = core_ptr(P, P.arg, g, p, first), *Aq = core_ptr(P, P.arg, g, p + 1, first);
        const short *Lt = L_ptr(P, g, p - 1, first), *Rt = R_ptr(P, g, p + 1, first);
        // ---- stage the value tables of both pivot sets (only the dimensions that exist on either side, rounded up to
        //      the 8-wide chunks the integrand reads; the walk over (row, dim) needs no division) ----
        {
            const int AL = min(VS, (p - 1 + 7) & ~7), AR = min(VS, (m - p - 1 + 7) & ~7);
            if (AL > 0) {
                int c = tid / AL, o = tid - c * AL;
                const int dc = CB / AL, dO = CB - dc * AL;
                while (c < r0) {
                    const int ix = (o < p - 1) ? (int)Lt[(size_t)o * RM + c] : 1;
                    XL[(size_t)c * RS + o] = par[ix - 1]; XL[(size_t)c * RS + VS + o] = par[n1m + ix - 1];
                    c += dc; o += dO; if (o >= AL) { o -= AL; c++; }
                }
            }
            if (AR > 0) {
                int c = tid / AR, o = tid - c * AR;
                const int dc = CB / AR, dO = CB - dc * AR;
                while (c < r2) {
                    const int ix = (o < m - p - 1) ? (int)Rt[(size_t)o * RM + c] : 1;
                    XR[(size_t)c * RS + o] = par[ix - 1]; XR[(size_t)c * RS + VS + o] = par[n1m + ix - 1];
                    c += dc; o += dO; if (o >= AR) { o -= AR; c++; }
                }
            }
        }
        if (ldsinv) {                                  // neighbour LU factors for the fix-ups of the append (roles C, D)
            if (p > first) { const double *gL = inv_ptr(P, g, p - 1, first); for (int x = tid; x < r0 * r0; x += CB) GL[x] = gL[x]; }
            if (p < last)  { const double *gU = inv_ptr(P, g, p + 1, first); for (int x = tid; x < r2 * r2; x += CB) GU[x] = gU[x]; }
        }
        CST(0);
        // ---- lottery (:410-484): every block draws and scores all candidates (identical results, no traffic) ----
        // generator words of the two draw columns: 48271^(2*rngpos+1) and 48271^(2*(rngpos+nlot)+1), advanced from
        // step to step by the small power 48271^(2*nlot) (every thread keeps them; no shared state, no barrier)
        if (tid == 0) s_ok = 1;                            // (read behind the rook loop; several barriers lie in between)
        const unsigned long long stepmul = ttx_minstd_pow(2ull * nlot);
        const unsigned long long sA1 = ttx_mulmod31(sA0, stepmul);
        if (zkeep) {
            const int b = p - first;
            const int *kc = ZK + (size_t)b * 2 * RM, *kr = kc + RM;
            if (tid == 0) { nzc = ZN[2 * b]; nzr = ZN[2 * b + 1]; }
            if (tid < ZN[2 * b]) zc[tid] = ((kc[tid] & 0xffff) - 1) + r0 * ((kc[tid] >> 16) - 1) + 1;
            if (tid < ZN[2 * b + 1]) zr[tid] = ((kr[tid] & 0xffff) - 1) + n2 * ((kr[tid] >> 16) - 1) + 1;
            __syncthreads();
        } else {
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
        }
        CST(1);
        const int Kc = r0 * n1 - nzc, Kr = n2 * r2 - nzr;
        if (tid < 64) { if (tid < P.cdf_ns[Kc]) segc[tid] = P.cdf_tab[(size_t)Kc * TTX_TABSEG + tid]; if (tid == 0) nsc = P.cdf_ns[Kc]; }
        else if (tid < 128) { const int t2 = tid - 64; if (t2 < P.cdf_ns[Kr]) segr[t2] = P.cdf_tab[(size_t)Kr * TTX_TABSEG + t2]; if (t2 == 0) nsr = P.cdf_ns[Kr]; }
        if (tid < r0) {                                // ascending sum over the left row (dims 1..p-1)
            double w = 1.0, wk = 1.0;
            auto wstep = [&](double xv) { wk = wk * xv; w = w + wk; };
            chain8v<false>(XL + (size_t)tid * RS, p - 1, wstep);
            pLw[tid] = w; pLk[tid] = wk;
            if (cfast) {
                double sk = 1.0, ss = 0.0, ww = 1.0;
                auto sstep = [&](double xv) { sk = sk * xv; ss = ss + sk; };
                auto pstep = [&](double xv) { ww = ww * xv; };
                chain8v<true>(XL + (size_t)tid * RS, p - 1, sstep);
                chain8v<false>(XL + (size_t)tid * RS + VS, p - 1, pstep);
                fSL[tid] = ss; fWL[tid] = ww;
            }
        } else if (tid >= 64 && tid < 64 + r2) {       // descending sum over the right row (dims p+2..m)
            const int c = tid - 64;
            double v = 1.0, vk = 1.0;
            auto vstep = [&](double xv) { vk = vk * xv; v = v + vk; };
            chain8v<true>(XR + (size_t)c * RS, m - p - 1, vstep);
            pRv[c] = v; pRk[c] = vk;
            if (cfast) {
                double sk = 1.0, ss = 0.0, ww = 1.0;
                auto sstep = [&](double xv) { sk = sk * xv; ss = ss + sk; };
                auto pstep = [&](double xv) { ww = ww * xv; };
                chain8v<false>(XR + (size_t)c * RS, m - p - 1, sstep);
                chain8v<false>(XR + (size_t)c * RS + VS, m - p - 1, pstep);
                fSR[c] = ss; fWR[c] = ww;
            }
        }
        __syncthreads();
        CST(2);
        double ma = 0.0, ba = -1.0, bv = 0.0; int bi = INT_MAX;
        for (int il = tid; il < nlot; il += CB) {
            const unsigned long long bil = (il == tid) ? bil0 : ttx_minstd_pow(2ull * il);
            const double d1 = ttx_flang_from_word(ttx_mulmod31(sA0, bil)), d2 = ttx_flang_from_word(ttx_mulmod31(sA1, bil));
            const int x = ttx_lottery_index(segc, nsc, Kc, r0 * n1, zc, nzc, d1);
            const int y = ttx_lottery_index(segr, nsr, Kr, n2 * r2, zr, nzr, d2);
            CST(3);
            const int i = (x - 1) % r0 + 1, j = (x - 1) / r0 + 1, k = (y - 1) % n2 + 1, q = (y - 1) / n2 + 1;
            lot[4 * il] = i; lot[4 * il + 1] = j; lot[4 * il + 2] = k; lot[4 * il + 3] = q;
            const double *rl = XL + (size_t)(i - 1) * RS, *rq = XR + (size_t)(q - 1) * RS;
            const double f = cfast ? f_ising_cfast(par[j - 1], par[n1m + j - 1], par[k - 1], par[n1m + k - 1], pRv[q - 1], pRk[q - 1], pLw[i - 1], pLk[i - 1], fSL[i - 1], fWL[i - 1], fSR[q - 1], fWR[q - 1])
                                   : F_ISING_CL(m, p - 1, rl, rl + VS, par[j - 1], par[n1m + j - 1], par[k - 1], par[n1m + k - 1], rq, rq + VS, pRv[q - 1], pRk[q - 1], pLw[i - 1], pLk[i - 1]);
            ma = fmax(ma, fabs(f));
            CST(4);
            const double *c = Cp + (i - 1) + (size_t)RM * (j - 1), *w = Wq + (k - 1) + (size_t)NM * (q - 1);
            double t = 0.0;
#pragma unroll 8
            for (int s = 0; s < r1; s++) t = t + c[P.SS * s] * w[P.SW * s];
            const double b = f - t, aa = fabs(b);
            if (aa > ba || (aa == ba && il < bi)) { ba = aa; bv = b; bi = il; }
        }
        ma = wave_max(ma);
        wave_argmax(ba, bv, bi);
        if (lane == 0) { shm[wv] = ma; sha[wv] = ba; shv[wv] = bv; shi[wv] = bi; }
        __syncthreads();
        ma = shm[0]; ba = sha[0]; bv = shv[0]; bi = shi[0];
        for (int x = 1; x < CB / 64; x++) {
            ma = fmax(ma, shm[x]);
            if (sha[x] > ba || (sha[x] == ba && shi[x] < bi)) { ba = sha[x]; bv = shv[x]; bi = shi[x]; }
        }
        amax = fmax(amax, ma);
        neval += nlot; rngpos += 2ull * nlot;
        sA0 = ttx_mulmod31(sA1, stepmul);
        if (bi == INT_MAX) bi = 0;                       // every residual a NaN: the first candidate, as idamax
        int ii = lot[4 * bi], jj = lot[4 * bi + 1], kk = lot[4 * bi + 2], qq = lot[4 * bi + 3];
        double pivot = bv;
        CST(5);
        // ---- rook half-steps (:516-582) / piv = 0 (:492-513): own slice, then one record per block ----
        int havecol = 0, haverow = 0, crs = 0, done = 0;
        int rc_k = -1, rc_q = -1, rr_i = -1, rr_j = -1;     // pivot at which resc / resr were computed (-1: not valid)
        const int H = (P.piv == 0) ? 2 : 2 * P.piv;
        const int r1u = __builtin_amdgcn_readfirstlane(r1);
        double mxrun = 0.0;                                 // this wave's largest |value| over the half-steps of this bond step
        // Only the waves that own fiber elements at the current ranks (and wave 0 of every workgroup) take part in the rook loop and
        // its record exchange: at low ranks that is one wave per workgroup, and 24 idle waves polling the same four cache lines for
        // the whole evaluation delayed the records of the working ones (2900 cycles from the last store to the end of the poll,
        // against ~1000 in isolation: profiles/probes/probe_exchange.hip).  The others wait at the barrier behind the loop and take
        // the results from LDS.  wact(c): participating waves of workgroup c -- the same number on every wave of the cluster.
        auto wact = [&](int c) {
            const int njc = (int)((long long)(c + 1) * n1 / NB) - (int)((long long)c * n1 / NB), nkc = (int)((long long)(c + 1) * n2 / NB) - (int)((long long)c * n2 / NB);
            const int e = max(r0 * njc, nkc * r2);
            return min(CB / 64, max(1, (e + 63) >> 6));
        };
        const bool rook_active = wv < wact(cb);
        const bool slot_on = (lane < NB * (CB / 64)) && ((lane & (CB / 64 - 1)) < wact(lane / (CB / 64)));      // lane = record slot
        if (rook_active) {
        for (int h = 0; h < H && !done; h++) {
            const bool iscol = (P.piv == 0) ? (h == 0) : (((h + (dir == 2 ? 1 : 0)) & 1) == 0);
            const int nf = iscol ? r0 * n1 : n2 * r2;
            const int nsl = iscol ? r0 * nj : nk * r2;
            double *fib = iscol ? acol : arow;
            // The r1 factor entries at the current pivot, ONE PER LANE of every wave (r1 <= 64): requested here, consumed after the
            // evaluation (lane s of the wave supplies entry s to the ordered residual sum through v_readlane).  No LDS, no
            // workgroup barrier: the waves of a cluster run the rook loop independently and meet only in the record exchange.
            double xsv = 0.0;
            if (lane < r1) xsv = iscol ? Wq[(kk - 1) + (size_t)NM * (qq - 1) + P.SW * lane] : Cp[(ii - 1) + (size_t)RM * (jj - 1) + P.SS * lane];
            crs++;
            if (iscol) havecol = 1; else haverow = 1;
            const int dn = (P.piv == 0) ? (h == 1) : (havecol && haverow && (crs >= 2 * P.piv));
            const bool resid = (P.piv != 0) && !dn;
            if (iscol) { rc_k = resid ? kk : -1; rc_q = qq; } else { rr_i = resid ? ii : -1; rr_j = jj; }
            CST(6);
            WST(0);
            double mx = 0.0, ab = -1.0, bb = 0.0; int ix = INT_MAX;
            for (int u = tid; u < nsl; u += CB) {
                double a;
                if (iscol) {
                    const int i = u % r0, j = jlo + u / r0, t = i + r0 * j;
                    const double *rl = XL + (size_t)i * RS, *rq = XR + (size_t)(qq - 1) * RS;
                    const double *c = Cp + i + (size_t)RM * j;
#if TTX_CL_CF
                    double cf[TTX_CL_CF];                // the factor entries of this element's residual, requested BEFORE the evaluation
                    if (resid) {
#pragma unroll
                        for (int s = 0; s < TTX_CL_CF; s++) if (s < r1u) cf[s] = c[P.SS * s];
                    }
#endif
                    a = cfast ? f_ising_cfast(par[j], par[n1m + j], par[kk - 1], par[n1m + kk - 1], pRv[qq - 1], pRk[qq - 1], pLw[i], pLk[i], fSL[i], fWL[i], fSR[qq - 1], fWR[qq - 1])
                              : F_ISING_CL(m, p - 1, rl, rl + VS, par[j], par[n1m + j], par[kk - 1], par[n1m + kk - 1], rq, rq + VS, pRv[qq - 1], pRk[qq - 1], pLw[i], pLk[i]);
                    fib[u] = a;
                    if (resid) {
                        double b = a;
#if TTX_CL_CF
#pragma unroll
                        for (int s = 0; s < TTX_CL_CF; s++) if (s < r1u) b = b + (-mvn_lane(xsv, s)) * cf[s];
#endif
#pragma unroll TTX_CL_UNR
                        for (int s = TTX_CL_CF; s < r1u; s++) b = b + (-mvn_lane(xsv, s)) * c[P.SS * s];
                        resc[u] = b;
                        const double aa = fabs(b);
                        if (aa > ab || (aa == ab && t < ix)) { ab = aa; bb = b; ix = t; }
                    }
                } else {
                    const int k = klo + u % nk, q = u / nk, t = k + n2 * q;
                    const double *rl = XL + (size_t)(ii - 1) * RS, *rq = XR + (size_t)q * RS;
                    const double *w = Wq + k + (size_t)NM * q;
#if TTX_CL_CF
                    double cf[TTX_CL_CF];
                    if (resid) {
#pragma unroll
                        for (int s = 0; s < TTX_CL_CF; s++) if (s < r1u) cf[s] = w[P.SW * s];
                    }
#endif
                    a = cfast ? f_ising_cfast(par[jj - 1], par[n1m + jj - 1], par[k], par[n1m + k], pRv[q], pRk[q], pLw[ii - 1], pLk[ii - 1], fSL[ii - 1], fWL[ii - 1], fSR[q], fWR[q])
                              : F_ISING_CL(m, p - 1, rl, rl + VS, par[jj - 1], par[n1m + jj - 1], par[k], par[n1m + k], rq, rq + VS, pRv[q], pRk[q], pLw[ii - 1], pLk[ii - 1]);
                    fib[u] = a;
                    if (resid) {
                        double tt = 0.0;
#if TTX_CL_CF
#pragma unroll
                        for (int s = 0; s < TTX_CL_CF; s++) if (s < r1u) tt = tt + cf[s] * mvn_lane(xsv, s);
#endif
#pragma unroll TTX_CL_UNR
                        for (int s = TTX_CL_CF; s < r1u; s++) tt = tt + w[P.SW * s] * mvn_lane(xsv, s);
                        const double b = a + (-1.0) * tt;
                        resr[u] = b;
                        const double aa = fabs(b);
                        if (aa > ab || (aa == ab && t < ix)) { ab = aa; bb = b; ix = t; }
                    }
                }
                mx = fmax(mx, fabs(a));
            }
            CST(7);
            WST(1);
            mx = wave_max(mx);
            wave_argmax2(ab, bb, ix);
            WST(2);
            // exchange: every WAVE publishes ONE tagged 16-byte record {|b|max, position, tag | sign} of its own elements and then
            // polls the 4 NB records of this half-step itself (L1-bypassing 16-byte loads, one per lane) and reduces them -- every
            // wave of the cluster arrives at the same result without a workgroup barrier.  A second record per exchange doubles its
            // cost (profiles/probes/probe_exchange.hip: 1.33 us per round with two records, 0.63 us with one, 0.49 us without the
            // sleep between polls), so the fiber maximum -- needed only by the acceptance test after the rook loop -- travels as a
            // RUNNING maximum in a record of its own that is stored here but read once, after the loop.  No fence: nothing but the
            // records crosses blocks here (fibers stay in LDS, factors were published at the end of earlier bond steps).  Records
            // are double-buffered by half-step parity (a wave cannot be two exchanges ahead of another: it needs that wave's record
            // of the exchange in between); the tag carries the launch number.
            u4 *bufA = (u4 *)part + ((size_t)(hcount & 1) * P.G + g) * TTX_CLREC;
            u4 *bufB = bufA + (size_t)2 * P.G * TTX_CLREC;
            const unsigned gen = ((unsigned)epoch << 20) | (unsigned)(hcount + 1);
            mxrun = fmax(mxrun, mx);
            if (lane == 0) {
                const unsigned long long ua = (unsigned long long)__double_as_longlong(ab), um = (unsigned long long)__double_as_longlong(mxrun);
                u4 ra, rb;
                rb.x = (unsigned)um; rb.y = (unsigned)(um >> 32); rb.z = gen; rb.w = 0;
                st16(&bufB[cb * (CB / 64) + wv], rb);
                ra.x = (unsigned)ua; ra.y = (unsigned)(ua >> 32); ra.z = (unsigned)ix; ra.w = gen | (signbit(bb) ? 0x80000000u : 0u);
                st16(&bufA[cb * (CB / 64) + wv], ra);
            }
            CST(8);
            WST(3);
            int ok = 1;
            ab = -1.0; bb = 0.0; ix = INT_MAX;
            if (slot_on) {
                unsigned spins = 0;
                for (;;) {
                    const u4 ra = ld16(&bufA[lane]);
                    if ((ra.w & 0x7fffffffu) == gen) {
                        ab = __longlong_as_double((long long)(((unsigned long long)ra.y << 32) | ra.x));
                        ix = (int)ra.z;
                        bb = (ra.w & 0x80000000u) ? -ab : ab;
                        break;
                    }
                    if (++spins > 64u) __builtin_amdgcn_s_sleep(1);
                    if (spins > (1u << 22)) { ok = 0; break; }
                    if ((spins & 1023u) == 0 && __hip_atomic_load(P.cl_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { ok = 0; break; }
                }
            }
            WST(4);
            ok = __all(ok);
            wave_argmax2(ab, bb, ix);
            WST(5);
            if (!ok) {                                   // aborted: the tail kernels of this sweep do nothing
                if (lane == 0) { __hip_atomic_store(P.cl_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); P.ctl[0] = 1; *(volatile int *)&s_ok = 0; }
                return;
            }
            CST(9);
            hcount++;
            neval += nf;
            bytes_half += resid ? 8.0 * ((double)nf * r1 + r1 + 2.0 * nf) : 8.0 * nf;
            n_resid += resid ? 1 : 0;
            done = dn;
            if (resid) {
                if (ix == INT_MAX) ix = 0;
                if (iscol) { const int i = ix % r0 + 1, j = ix / r0 + 1; done = havecol && haverow && (i == ii && j == jj); ii = i; jj = j; }
                else       { const int k = ix % n2 + 1, q = ix / n2 + 1; done = havecol && haverow && (k == kk && q == qq); kk = k; qq = q; }
                pivot = bb;
            }
        }
        if (P.piv != 0) {            // the fiber maxima of the bond step (:531 / :564; the piv = 0 branch :492-513 does not touch amax):
            // the running maxima that every wave stored next to its record of the LAST exchange
            u4 *bufB = (u4 *)part + ((size_t)((hcount - 1) & 1) * P.G + g) * TTX_CLREC + (size_t)2 * P.G * TTX_CLREC;
            const unsigned gen = ((unsigned)epoch << 20) | (unsigned)hcount;
            double mx = 0.0; int ok = 1;
            if (slot_on) {
                unsigned spins = 0;
                for (;;) {
                    const u4 rb = ld16(&bufB[lane]);
                    if (rb.z == gen) { mx = __longlong_as_double((long long)(((unsigned long long)rb.y << 32) | rb.x)); break; }
                    if (++spins > 64u) __builtin_amdgcn_s_sleep(1);
                    if (spins > (1u << 22)) { ok = 0; break; }
                    if ((spins & 1023u) == 0 && __hip_atomic_load(P.cl_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { ok = 0; break; }
                }
            }
            ok = __all(ok);
            if (!ok) {
                if (lane == 0) { __hip_atomic_store(P.cl_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); P.ctl[0] = 1; *(volatile int *)&s_ok = 0; }
                return;
            }
            amax = fmax(amax, wave_max(mx));
        }
        if (tid == 0) {
            s_rook.ii = ii; s_rook.jj = jj; s_rook.kk = kk; s_rook.qq = qq; s_rook.hcount = hcount; s_rook.rc_k = rc_k; s_rook.rc_q = rc_q;
            s_rook.rr_i = rr_i; s_rook.rr_j = rr_j; s_rook.pivot = pivot; s_rook.amax = amax; s_rook.bytes_half = bytes_half; s_rook.neval = neval; s_rook.n_resid = n_resid;
        }
        }   // rook_active
        __syncthreads();
        if (!*(volatile int *)&s_ok) return;                // the working waves gave up (bounded wait expired) and have left
        if (!rook_active) {
            ii = s_rook.ii; jj = s_rook.jj; kk = s_rook.kk; qq = s_rook.qq; hcount = s_rook.hcount; rc_k = s_rook.rc_k; rc_q = s_rook.rc_q;
            rr_i = s_rook.rr_i; rr_j = s_rook.rr_j; pivot = s_rook.pivot; amax = s_rook.amax; bytes_half = s_rook.bytes_half; neval = s_rook.neval; n_resid = s_rook.n_resid;
        }
        CST(10);
        // ---- acceptance and in-place append (:598-758): every block appends its own slice ----
        int *tape = P.tape + ((size_t)g * (m + 2) + p) * 4;
        const bool upd = (fabs(pivot) > P.small_element * amax) && (fabs(pivot) > P.small_pivot * pivotmax_prev);
        if (!upd) {
            if (cb == 0 && tid == 0) { tape[0] = tape[1] = tape[2] = tape[3] = -1; P.upd[(size_t)g * (m + 2) + p] = 0; }
        } else {
            const int i0 = ii - 1, j0 = jj - 1, k0 = kk - 1, q0 = qq - 1;
            double *gI = inv_ptr(P, g, p, first);
            // the factor entries at the final pivot (the half-steps keep theirs in registers)
            for (int s = tid; s < r1; s += CB) { xsc[s] = Wq[k0 + (size_t)NM * q0 + P.SW * s]; xsr[s] = Cp[i0 + (size_t)RM * j0 + P.SS * s]; }
            __syncthreads();
            // role E first part: packed LU from the OLD factors (:649-660)
            if (cb == 0)
                for (int s = tid; s < r1; s += CB) { gI[r1 * r1 + s] = xsr[s]; gI[r1 * r1 + r1 + s] = xsc[s]; }
            // role A: arg(p), col(p) new slab (:662-668, :701)
            const bool reuseA = (rc_k == kk && rc_q == qq), reuseB = (rr_i == ii && rr_j == jj);
            for (int u = tid; u < r0 * nj; u += CB) {
                const int i = u % r0, j = jlo + u / r0; const size_t o = i + (size_t)RM * j;
                const double a = acol[u];
                Ap[o + P.SS * r1] = a;
                double y;
                if (reuseA) y = resc[u];             // the same sum, in the same order, as the last column half-step
                else {
                    y = a;
#pragma unroll 16
                    for (int s = 0; s < r1; s++) y = y + (-xsc[s]) * Cp[o + P.SS * s];
                }
                Cp[o + P.SS * r1] = (1.0 / pivot) * y;
            }
            // role B: arg(p+1), row(p+1) new row (:669-674, :702)
            for (int u = tid; u < nk * r2; u += CB) {
                const int k = klo + u % nk, q = u / nk;
                const double a = arow[u];
                Aq[r1 + (size_t)RM * k + P.SS * q] = a;
                const size_t o = k + (size_t)NM * q;
                if (reuseB) Wq[o + P.SW * r1] = resr[u];
                else {
                    double tt = 0.0;
#pragma unroll 16
                    for (int s = 0; s < r1; s++) tt = tt + Wq[o + P.SW * s] * xsr[s];
                    Wq[o + P.SW * r1] = a + (-1.0) * tt;
                }
            }
            CST(11);
            // roles C and D are triangular solves along the rank index, one per own column j / row k.  A solve occupies
            // r0 (r2) lanes: with ranks up to 32 two of them share a wave (shuffles of width 32).
            // role C: row(p)(:, j, r1+1) = L(p-1)^-1 acol1(:, j)  (:715-728)
            if (p > first) {
                const double *gL = ldsinv ? GL : inv_ptr(P, g, p - 1, first);
                double *Wp = core_ptr(P, P.row, g, p, first);
                const int pk = (r0 <= 32) ? 2 : 1, lw = 64 / pk, l = lane & (lw - 1), sub = lane / lw;
                for (int base = wv * pk; base < nj; base += (CB / 64) * pk) {
                    const int jl = base + sub, j = jlo + jl;
                    const bool on = (jl < nj) && (l < r0);
                    const double a = on ? acol[l + r0 * jl] : 0.0;
                    double tmp = 0.0, xf = 0.0;
                    for (int s = 0; s < r0; s++) {
                        const double cand = (s == 0) ? a : a + (-1.0) * tmp;
                        const double xsv = __shfl(cand, s, lw);
                        if (l == s) xf = xsv;
                        if (l > s && l < r0) tmp = tmp + xsv * gL[l * l + s];
                    }
                    if (on) Wp[j + (size_t)NM * r1 + P.SW * l] = xf;
                }
            }
            CST(12);
            // role D: col(p+1)(r1+1, k, :) = arow1(k, :) U(p+1)^-1  (:730-749)
            if (p < last) {
                const double *gU = ldsinv ? GU : inv_ptr(P, g, p + 1, first);
                double *Cq = core_ptr(P, P.col, g, p + 1, first);
                const int pk = (r2 <= 32) ? 2 : 1, lw = 64 / pk, l = lane & (lw - 1), sub = lane / lw;
                const double rdg = (l < r2) ? 1.0 / gU[(l + 1) * (l + 1) - 1] : 0.0;   // 1/U(s,s) held by lane s
                for (int base = wv * pk; base < nk; base += (CB / 64) * pk) {
                    const int kl = base + sub, k = klo + kl;
                    const bool on = (kl < nk) && (l < r2);
                    double y = on ? arow[kl + nk * l] : 0.0;
                    for (int s = 0; s < r2; s++) {
                        const double cand = rdg * y;          // only lane s's product is used: (1.0 / U(s,s)) * y_s
                        const double ys = __shfl(cand, s, lw);
                        if (l == s) y = ys;
                        if (l > s && l < r2) y = y + (-gU[l * l + l + s]) * ys;
                    }
                    if (on) Cq[r1 + (size_t)RM * k + P.SS * l] = y;
                }
            }
            CST(13);
            // role E: index tables, pivot set, scalars (:604-635)
            if (cb == 0) {
                short *Ln = L_ptr(P, g, p, first), *Rn = R_ptr(P, g, p, first);
                for (int x = tid; x < p; x += CB) Ln[(size_t)x * RM + r1] = (x < p - 1) ? Lt[(size_t)x * RM + i0] : (short)(j0 + 1);
                for (int x = tid; x < m - p; x += CB) Rn[(size_t)x * RM + r1] = (x == 0) ? (short)(k0 + 1) : Rt[(size_t)(x - 1) * RM + q0];
                if (tid == 0) {
                    gI[(r1 + 1) * (r1 + 1) - 1] = pivot;
                    int *vq = vip_ptr(P, g, p, first) + 4 * r1;
                    vq[0] = tape[0] = ii; vq[1] = tape[1] = jj; vq[2] = tape[2] = kk; vq[3] = tape[3] = qq;
                    P.upd[(size_t)g * (m + 2) + p] = 1;
                    r[p] = r1 + 1;                                                      // :752
                }
            }
            if (zkeep && wv < 2) {                     // keep the sorted distinct lists of this bond current:
                const int b = p - first;               // wave 0 inserts the pivot row, wave 1 the pivot column (RM <= 64)
                int *kl = ZK + (size_t)b * 2 * RM + (size_t)wv * RM;
                const int nl = ZN[2 * b + wv];
                const int key = wv == 0 ? ((jj << 16) | ii) : ((qq << 16) | kk);
                const int v = (lane < nl) ? kl[lane] : INT_MAX;
                const bool dup = __ballot(v == key) != 0ull;
                const int pos = __popcll(__ballot(v < key));
                if (!dup) {
                    if (lane < nl && v > key) kl[lane + 1] = v;      // every lane has read its entry before any lane writes
                    if (lane == 0) { kl[pos] = key; ZN[2 * b + wv] = nl + 1; }
                }
            }
            const double ap = fabs(pivot);
            pivotmax = (pivotmax < 0.0) ? ap : fmax(pivotmax, ap);
            pivotmin = (pivotmin < 0.0) ? ap : fmin(pivotmin, ap);
        }
        CST(14);
        // end of the bond step: appends and the new rank become visible to the whole cluster (every block read r[] of
        // this step before its first half-step barrier, so block 0 may already have overwritten r[p])
        if (!cluster_sync(ctr, (++nbar) * (unsigned)NB, P.cl_abort, &s_ok)) { if (tid == 0) P.ctl[0] = 1; return; }
        CST(15);
        CST_END();
    }
    if (cb == 0) {
        if (tid == 0) {
            gs.amax = amax; gs.pivotmax = pivotmax; gs.pivotmin = pivotmin; gs.neval = neval; gs.rngpos = rngpos;
            gs.bytes_half = bytes_half; gs.n_resid = n_resid;
        }
        __syncthreads();
        exch_pack_group(P, g);          // the group's outgoing boundary messages (saves the separate k_exch_pack launch)
    }
}
