// Development probe: cost of one record exchange between the waves of a cluster (8 workgroups x 4 waves on one XCD), as in
// k_sweep_cluster: every wave publishes a tagged record with sc1 stores and polls the records of all waves with sc1 loads.
//   hipcc --offload-arch=gfx950 -O3 -o probe_exchange probe_exchange.hip && ./probe_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(void *p, u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u4 ld16(const void *p) { u4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void ld16x2(const void *p, const void *q, u4 &a, u4 &b)
{ asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p), "v"(q) : "memory"); }

// mode 0: 2 x 16-byte records, two dependent loads per poll (the kernel's scheme); 1: both loads in flight; 2: one 16-byte record;
// 3: mode 2 without s_sleep; 4: mode 2, only wave 0 of each block publishes/polls + __syncthreads (the round-2 scheme)
__global__ __launch_bounds__(256) void k_probe(u4 *rec, int NB, int iters, int mode, int work, long long *out)
{
    const int bid = blockIdx.x, g = bid & 7, cb = bid >> 3, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ double sh[4];
    u4 *buf0 = rec + (size_t)g * 2 * 2 * 64;
    const int nrec = (mode == 4) ? NB : NB * 4, me = (mode == 4) ? cb : cb * 4 + wv;
    double acc = 1.0 + 1e-9 * threadIdx.x;
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        for (int k = 0; k < work; k++) acc = acc * 1.0000001 + 1e-12;          // stand-in for the evaluation
        u4 *buf = buf0 + (size_t)(it & 1) * 2 * 64;
        const unsigned gen = (unsigned)(it + 1);
        if (mode == 4) { if (lane == 0) sh[wv] = acc; __syncthreads(); }
        if (mode != 4 || wv == 0) {
            if (lane == 0) { u4 r; r.x = __double_as_longlong(acc) & 0xffffffff; r.y = 1; r.z = me; r.w = gen; st16(buf + 2 * me, r); if (mode < 2) st16(buf + 2 * me + 1, r); }
            if (lane < nrec) {
                for (;;) {
                    u4 a, b;
                    if (mode == 0) { a = ld16(buf + 2 * lane); b = ld16(buf + 2 * lane + 1); }
                    else if (mode == 1) ld16x2(buf + 2 * lane, buf + 2 * lane + 1, a, b);
                    else { a = ld16(buf + 2 * lane); b = a; }
                    if (a.w == gen && b.w == gen) break;
                    if (mode != 3) __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        if (mode == 4) __syncthreads();
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0 && g == 0 && cb == 0) { out[0] = t1 - t0; out[1] = (long long)acc; }
}
int main()
{
    u4 *rec; long long *out;
    hipMalloc((void **)&rec, 8 * 2 * 2 * 64 * sizeof(u4)); hipMalloc((void **)&out, 16);
    const int NB = 8, iters = 2000;
    for (int work : {0, 400})
        for (int mode = 0; mode < 5; mode++) {
            hipMemset(rec, 0, 8 * 2 * 2 * 64 * sizeof(u4));
            long long h[2];
            for (int rep = 0; rep < 2; rep++) {
                hipMemset(rec, 0, 8 * 2 * 2 * 64 * sizeof(u4));
                hipLaunchKernelGGL(k_probe, dim3(8 * NB), dim3(256), 0, 0, rec, NB, iters, mode, work, out);
                hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            }
            printf("work %4d mode %d: %.3f us per exchange round\n", work, mode, 0.01 * (double)h[0] / iters);
        }
    return 0;
}
