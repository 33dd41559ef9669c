// Development probe (not part of the library): unit costs behind the round-2 D/E kernels on one MI355X.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/probes/probe_dpp.hip -o /tmp/probe_dpp && /tmp/probe_dpp
//  1. dependent product chain fed by DPP row broadcasts (v_mov_b64_dpp row_newbcast + v_mul_f64), factors from global memory
//  2. latency of one token hop between two waves of a workgroup through LDS (value + sentinel polling)
//  3. division throughput per SIMD with 1, 2 and 4 resident waves ((u-1)/(u+1))^2 * a, the D/E pair factor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ double bc(double f)
{
    double r;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(f), "n"(N));
    return r;
}
__device__ __forceinline__ double fold16(double a, double f)
{
    a = a * bc<0>(f); a = a * bc<1>(f); a = a * bc<2>(f); a = a * bc<3>(f);
    a = a * bc<4>(f); a = a * bc<5>(f); a = a * bc<6>(f); a = a * bc<7>(f);
    a = a * bc<8>(f); a = a * bc<9>(f); a = a * bc<10>(f); a = a * bc<11>(f);
    a = a * bc<12>(f); a = a * bc<13>(f); a = a * bc<14>(f); a = a * bc<15>(f);
    return a;
}
__global__ __launch_bounds__(64) void k_dpp_chain(const double *g, int nbatch, int reps, double *out, double *chk)
{
    const int lane = threadIdx.x;
    double a = 1.0;
    long long t0 = wall_clock64();
    for (int r = 0; r < reps; r++) {
        const double *p = g + (lane & 15);
        double f0 = p[0], f1 = p[16], f2 = p[32], f3 = p[48];
        for (int b = 0; b < nbatch; b += 4) {
            const double *q = p + 16 * (b + 4);
            const double n0 = q[0], n1 = q[16], n2 = q[32], n3 = q[48];
            a = fold16(a, f0); a = fold16(a, f1); a = fold16(a, f2); a = fold16(a, f3);
            f0 = n0; f1 = n1; f2 = n2; f3 = n3;
        }
    }
    long long t1 = wall_clock64();
    if (lane == 0 && blockIdx.x == 0) out[0] = 10.0 * (double)(t1 - t0) / ((double)reps * nbatch * 16);
    if (blockIdx.x == 0) chk[lane] = a;
}
// (1b) the U-wave pattern of k_halfstep_det: the chain value after every factor is also stored to LDS (one 512-byte row per factor)
template <int STORE>
__global__ __launch_bounds__(64) void k_dpp_store(const double *g, int nbatch, int reps, double *out, double *chk)
{
    __shared__ double buf[48 * 64];
    const int lane = threadIdx.x;
    double a = 1.0;
    long long t0 = wall_clock64();
    for (int r = 0; r < reps; r++) {
        const double *p = g + (lane & 15);
        double f = p[0];
        for (int b = 0; b < nbatch; b++) {
            const double fn = p[16 * (b + 1)];
            double *o = buf + (size_t)(b % 3) * 16 * 64 + lane;
#define ST(k) a = a * bc<k>(f); if (STORE) o[k * 64] = a;
            ST(0) ST(1) ST(2) ST(3) ST(4) ST(5) ST(6) ST(7) ST(8) ST(9) ST(10) ST(11) ST(12) ST(13) ST(14) ST(15)
#undef ST
            f = fn;
        }
    }
    long long t1 = wall_clock64();
    if (lane == 0 && blockIdx.x == 0) out[0] = 10.0 * (double)(t1 - t0) / ((double)reps * nbatch * 16);
    if (blockIdx.x == 0) chk[lane] = a + buf[lane];
}
// reference for (1): the same product as a plain sequential loop on one lane
__global__ void k_seq_chain(const double *g, int n, double *o) { double a = 1.0; for (int i = 0; i < n; i++) a = a * g[i]; *o = a; }

#define SENT 0xfff85a5a00000001ull
__global__ __launch_bounds__(128) void k_hop(int hops, double *out)
{
    __shared__ unsigned long long box[2][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    box[0][lane] = SENT; box[1][lane] = SENT;
    __syncthreads();
    double a = 1.0 + lane;
    long long t0 = wall_clock64();
    for (int h = 0; h < hops; h++) {
        if ((h & 1) == wv) {                    // my turn to send
            box[wv][lane] = (unsigned long long)__double_as_longlong(a);
        } else {                                // wait for the partner's value, reset the slot
            unsigned long long v; unsigned spins = 0;
            do { v = __hip_atomic_load(&box[wv ^ 1][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (__any(v == SENT) && ++spins < (1u << 24));
            box[wv ^ 1][lane] = SENT;
            a = __longlong_as_double((long long)v) * 1.0000001;
        }
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = 10.0 * (double)(t1 - t0) / hops;
    if (threadIdx.x == 64) out[1] = a;
}

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
__device__ __forceinline__ double t2(double u) { const double n = u - 1.0, d = u + 1.0; const double t = fdiv_unit(n, d); return t * t; }
// WPS waves per SIMD: block of 4*WPS waves on one CU; every wave runs `len` pairs of the chain u *= x, a *= t2(u), 4 at a time
__global__ void k_div(int len, int reps, const double *x, double *out, double *sink)
{
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    long long t0 = wall_clock64();
    for (int r = 0; r < reps; r++) {
        double u = 0.9 + 1e-4 * lane, a = 1.0;
        for (int j = 0; j + 4 <= len; j += 4) {
            const double u1 = u * x[j], u2 = u1 * x[j + 1], u3 = u2 * x[j + 2], u4 = u3 * x[j + 3];
            const double s1 = t2(u1), s2 = t2(u2), s3 = t2(u3), s4 = t2(u4);
            a = a * s1; a = a * s2; a = a * s3; a = a * s4;
            u = u4;
        }
        acc += a;
    }
    __syncthreads();                           // the block's time: until its slowest wave is done
    long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 10.0 * (double)(t1 - t0) / ((double)reps * len);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}


// (4) ONE wave, W independent pair factors written stage by stage (W = 1, 2, 4, 8): does interleaving the nine-deep
//     division chains at source level let a lone wave reach the SIMD's issue rate?
template <int W>
__global__ __launch_bounds__(64) void k_divw(int len, int reps, const double *x, double *out, double *sink)
{
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    long long t0 = wall_clock64();
    for (int r = 0; r < reps; r++) {
        double u = 0.9 + 1e-4 * lane, a = 1.0;
        for (int j = 0; j + W <= len; j += W) {
            double uu[W], n[W], d[W], rr[W], e[W], q[W], m_[W], s[W];
#pragma unroll
            for (int k = 0; k < W; k++) { u = u * x[j + k]; uu[k] = u; }
#pragma unroll
            for (int k = 0; k < W; k++) { n[k] = uu[k] - 1.0; d[k] = uu[k] + 1.0; }
#pragma unroll
            for (int k = 0; k < W; k++) rr[k] = __builtin_amdgcn_rcp(d[k]);
#pragma unroll
            for (int k = 0; k < W; k++) e[k] = __builtin_fma(-d[k], rr[k], 1.0);
#pragma unroll
            for (int k = 0; k < W; k++) rr[k] = __builtin_fma(rr[k], e[k], rr[k]);
#pragma unroll
            for (int k = 0; k < W; k++) e[k] = __builtin_fma(-d[k], rr[k], 1.0);
#pragma unroll
            for (int k = 0; k < W; k++) rr[k] = __builtin_fma(rr[k], e[k], rr[k]);
#pragma unroll
            for (int k = 0; k < W; k++) q[k] = n[k] * rr[k];
#pragma unroll
            for (int k = 0; k < W; k++) m_[k] = __builtin_fma(-d[k], q[k], n[k]);
#pragma unroll
            for (int k = 0; k < W; k++) s[k] = __builtin_fma(m_[k], rr[k], q[k]);
#pragma unroll
            for (int k = 0; k < W; k++) a = a * (s[k] * s[k]);
        }
        acc += a;
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 10.0 * (double)(t1 - t0) / ((double)reps * len);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// (5) issue rate of independent three-register fp64 FMAs on one wave (8 streams), and of v_rcp_f64 followed by a use
__global__ __launch_bounds__(64) void k_fma3(int reps, double *out, double *sink)
{
    const int lane = threadIdx.x;
    double f[8], g[8], h[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { f[q] = 1.0 + 1e-9 * (lane + q); g[q] = 1.0 - 1e-9 * (lane + 3 * q); h[q] = 1e-9 * q; }
    long long t0 = wall_clock64();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int q = 0; q < 8; q++) f[q] = __builtin_fma(f[q], g[q], h[q]);
    }
    long long t1 = wall_clock64();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int q = 0; q < 8; q++) f[q] = __builtin_fma(-g[q], f[q], 1.0);
    }
    long long t2 = wall_clock64();
    if (lane == 0) { out[0] = 10.0 * (double)(t1 - t0) / (8.0 * reps); out[1] = 10.0 * (double)(t2 - t1) / (8.0 * reps); }
    double acc = 0; for (int q = 0; q < 8; q++) acc += f[q];
    sink[lane] = acc;
}

// (6) shader clock while a launch of `blocks` single-wave workgroups runs a dependent fp64 chain for ~ms: core-clock counter
//     (s_memtime) against the constant 100 MHz counter (s_memrealtime)
__global__ __launch_bounds__(64) void k_clock(int iters, double *out, double *sink)
{
    double a = 1.0 + 1e-9 * threadIdx.x;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) a = __builtin_fma(a, 1.0000001, 1e-9);
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (double)(c1 - c0) / ((double)(w1 - w0) * 10.0); out[1] = (double)(w1 - w0) * 10.0 / iters; }
    sink[blockIdx.x * 64 + threadIdx.x] = a;
}

// (7) latency of DEPENDENT fp64 operations whose second operand comes from a register array (one wave): add, mul, fma(x, 1.0, acc)
//     (= the same sum with one rounding), fma with a register multiplier
template <int OP>
__global__ __launch_bounds__(64) void k_dep(int reps, const double *g, double *out, double *sink)
{
    double x[32];
#pragma unroll
    for (int q = 0; q < 32; q++) x[q] = g[q * 64 + threadIdx.x];
    double a = (OP == 1) ? 1.0 : 0.0;
    const long long t0 = wall_clock64();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int q = 0; q < 32; q++) {
            if (OP == 0) a = a + x[q];
            if (OP == 1) a = a * x[q];
            if (OP == 2) a = __builtin_fma(x[q], 1.0, a);
            if (OP == 3) a = __builtin_fma(x[q], x[(q + 1) & 31], a);
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = 10.0 * (double)(t1 - t0) / (32.0 * reps);
    sink[threadIdx.x] = a;
}

int main()
{
    const int NB = 512, N = NB * 16;
    double *hg = (double *)malloc(sizeof(double) * (N + 256));
    for (int i = 0; i < N + 256; i++) hg[i] = 1.0 - 1e-7 * ((i * 2654435761u) % 1000);
    double *g, *out, *chk, *seq, *sink;
    CK(hipMalloc(&g, sizeof(double) * (N + 256))); CK(hipMalloc(&out, 64)); CK(hipMalloc(&chk, 512)); CK(hipMalloc(&seq, 8));
    CK(hipMalloc(&sink, sizeof(double) * 1024 * 1024));
    CK(hipMemcpy(g, hg, sizeof(double) * (N + 256), hipMemcpyHostToDevice));
    double o[8], c[64], s;
    for (int blocks : {1, 1024, 4096}) {
        hipLaunchKernelGGL(k_dpp_chain, dim3(blocks), dim3(64), 0, 0, g, NB, 20, out, chk);
        hipLaunchKernelGGL(k_dpp_chain, dim3(blocks), dim3(64), 0, 0, g, NB, 20, out, chk);
        CK(hipMemcpy(o, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, chk, 512, hipMemcpyDeviceToHost));
        printf("dpp chain, %4d blocks of one wave: %.2f ns per factor\n", blocks, o[0]);
    }
    {
        hipLaunchKernelGGL(k_dpp_store<0>, dim3(1), dim3(64), 0, 0, g, NB, 20, out, chk);
        hipLaunchKernelGGL(k_dpp_store<0>, dim3(1), dim3(64), 0, 0, g, NB, 20, out, chk);
        CK(hipMemcpy(o, out, 8, hipMemcpyDeviceToHost));
        printf("dpp chain, one register of 16 factors at a time, no store: %.2f ns per factor\n", o[0]);
        hipLaunchKernelGGL(k_dpp_store<1>, dim3(1), dim3(64), 0, 0, g, NB, 20, out, chk);
        hipLaunchKernelGGL(k_dpp_store<1>, dim3(1), dim3(64), 0, 0, g, NB, 20, out, chk);
        CK(hipMemcpy(o, out, 8, hipMemcpyDeviceToHost));
        printf("the same with the running product stored to LDS after every factor: %.2f ns per factor\n", o[0]);
    }
    // exactness: 20 repetitions of the N-factor product on lane 0 vs the plain loop
    {
        double *h2 = (double *)malloc(sizeof(double) * N * 20);
        for (int r = 0; r < 20; r++) for (int i = 0; i < N; i++) h2[r * N + i] = hg[i];
        double *g2; CK(hipMalloc(&g2, sizeof(double) * N * 20)); CK(hipMemcpy(g2, h2, sizeof(double) * N * 20, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_seq_chain, dim3(1), dim3(1), 0, 0, g2, N * 20, seq);
        CK(hipMemcpy(&s, seq, 8, hipMemcpyDeviceToHost));
        int same = 1; for (int l = 0; l < 64; l++) same &= (c[l] == s);
        printf("dpp chain product %.17g, sequential %.17g, all 64 lanes identical to it: %s\n", c[0], s, same ? "yes" : "NO");
    }
    hipLaunchKernelGGL(k_hop, dim3(1), dim3(128), 0, 0, 20000, out);
    hipLaunchKernelGGL(k_hop, dim3(1), dim3(128), 0, 0, 20000, out);
    CK(hipMemcpy(o, out, 16, hipMemcpyDeviceToHost));
    printf("token hop between two waves through LDS: %.1f ns per hop\n", o[0]);
    for (int wps : {1, 2, 3, 4}) {
        for (int blocks : {1, 256}) {
            hipLaunchKernelGGL(k_div, dim3(blocks), dim3(256 * wps), 0, 0, 128, 200, g, out, sink);
            hipLaunchKernelGGL(k_div, dim3(blocks), dim3(256 * wps), 0, 0, 128, 200, g, out, sink);
            CK(hipMemcpy(o, out, 8, hipMemcpyDeviceToHost));
            printf("division run, %d wave(s) per SIMD, %3d blocks: %.2f ns per pair per wave = %.2f ns per pair per SIMD\n", wps, blocks, o[0], o[0] / wps);
        }
    }
    {
        auto run = [&](auto kern, int W) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, 128, 200, g, out, sink);
            hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, 128, 200, g, out, sink);
            CK(hipMemcpy(o, out, 8, hipMemcpyDeviceToHost));
            printf("one wave, %d division chains interleaved stage by stage: %.2f ns per pair\n", W, o[0]);
        };
        run(k_divw<1>, 1); run(k_divw<2>, 2); run(k_divw<4>, 4); run(k_divw<8>, 8);
        hipLaunchKernelGGL(k_fma3, dim3(1), dim3(64), 0, 0, 2000, out, sink);
        hipLaunchKernelGGL(k_fma3, dim3(1), dim3(64), 0, 0, 2000, out, sink);
        CK(hipMemcpy(o, out, 16, hipMemcpyDeviceToHost));
        printf("one wave, 8 independent streams: fma(v,v,v) %.2f ns per instruction, fma(-v,v,1.0) %.2f ns\n", o[0], o[1]);
    }
    {
        const char *nm[4] = {"a + x", "a * x", "fma(x, 1.0, a)", "fma(x, y, a)"};
        auto run = [&](auto kern, int op) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, 2000, g, out, sink);
            hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, 2000, g, out, sink);
            CK(hipMemcpy(o, out, 8, hipMemcpyDeviceToHost));
            printf("one wave, dependent %-16s with register operands: %.2f ns per operation\n", nm[op], o[0]);
        };
        run(k_dep<0>, 0); run(k_dep<1>, 1); run(k_dep<2>, 2); run(k_dep<3>, 3);
    }
    for (int blocks : {1, 64, 200, 1024, 8192}) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(64), 0, 0, 400000, out, sink);
        CK(hipMemcpy(o, out, 16, hipMemcpyDeviceToHost));
        printf("shader clock with %4d single-wave workgroups in flight: %.2f GHz (dependent fma: %.2f ns)\n", blocks, o[0], o[1]);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
