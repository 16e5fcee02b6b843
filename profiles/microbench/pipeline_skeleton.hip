// what makes the pipelined sweep slow?  5 waves: 3 producers (4/6/6 points), 1 integrator (16 points), 1 idle;
// flags switch on LDS writes, LDS reads, the per-chunk barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#define BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int CNT, bool LDSW, bool BARR>
__device__ __forceinline__ double producer(double* prod, int iters, double E, double R2, double d2p4, int lane, int off)
{
    double acc = 0, x = 1.0 + lane * 1e-3, y = 0.999;
    int ps = 0;
    for (int it = 0; it < iters; ++it) {
        double* P = prod + ps * 16 * 192 + off * 192 + lane;
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            const double f = (x + k - E) * R2 * y + d2p4;
            const double d = 1. - (1. / 12.) * f;
            double rr = __builtin_amdgcn_rcp(d);
            double e = __builtin_fma(-d, rr, 1.0);
            rr = __builtin_fma(rr, e, rr);
            e = __builtin_fma(-d, rr, 1.0);
            rr = __builtin_fma(rr, e, rr);
            if (LDSW) { P[k * 192] = f; P[k * 192 + 64] = d; P[k * 192 + 128] = rr; } else acc += f + d + rr;
        }
        x += 1e-9;
        ps = ps == 2 ? 0 : ps + 1;
        if (BARR) BAR();
    }
    return acc;
}
template <bool LDSR, bool BARR, int PREF>
__device__ __forceinline__ double integrator(const double* prod, int iters, int lane)
{
    double w = 1.0 + lane * 1e-3, wprev = 0.99, u = 1.0, fprev = 1e-9;
    int ls = 0;
    double F[16], D[16], R[16];
    for (int k = 0; k < 16; ++k) { F[k] = 1e-9; D[k] = 0.9999999; R[k] = 1.0000001; }
    for (int it = 0; it < iters; ++it) {
        const double* P = prod + ls * 16 * 192 + lane;
        double Fn[16], Dn[16], Rn[16];
        if (LDSR) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { Fn[k] = P[k * 192]; Dn[k] = P[k * 192 + 64]; Rn[k] = P[k * 192 + 128]; }
            if (PREF == 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double wnext = __builtin_fma(2., w, -wprev) + u * fprev;
            wprev = w; w = wnext;
            const double q = wnext * R[k];
            const double rem = __builtin_fma(-D[k], q, wnext);
            u = __builtin_fma(rem, R[k], q);
            fprev = F[k];
            if (PREF == 2 && LDSR) {
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
            }
        }
        if (LDSR) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { F[k] = Fn[k]; D[k] = Dn[k]; R[k] = Rn[k]; }
        }
        ls = ls == 2 ? 0 : ls + 1;
        if (BARR) BAR();
    }
    return u + w;
}
template <bool LDSW, bool LDSR, bool BARR, int PREF>
__global__ __launch_bounds__(320) void k(double* out, int iters, double E, double R2, double d2p4, int mask)
{
    __shared__ double prod[3 * 16 * 192 + 64];
    const int lane = threadIdx.x & 63, role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 3 * 16 * 192; i += 320) prod[i] = 1.0;
    __syncthreads();
    double r = 0;
    const bool on = (mask >> role) & 1;
    if (!on) { if (BARR) for (int it = 0; it < iters; ++it) BAR(); }
    else if (role == 0) r = producer<4, LDSW, BARR>(prod, iters, E, R2, d2p4, lane, 0);
    else if (role == 1) r = producer<6, LDSW, BARR>(prod, iters, E, R2, d2p4, lane, 4);
    else if (role == 3) r = producer<6, LDSW, BARR>(prod, iters, E, R2, d2p4, lane, 10);
    else if (role == 2) r = integrator<LDSR, BARR, PREF>(prod, iters, lane);
    else { if (BARR) for (int it = 0; it < iters; ++it) BAR(); }
    out[threadIdx.x] = r;
}
template <bool LDSW, bool LDSR, bool BARR, int PREF = 1>
void run(const char* name, double* d, int mask)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000; float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<LDSW, LDSR, BARR, PREF>), dim3(1), dim3(320), 0, 0, d, iters, -100.0, 1e-8, 2.5e-9, mask);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-34s mask 0x%02x: %7.1f ns per chunk (%.1f ns/pt)\n", name, mask, ms * 1e6 / iters, ms * 1e6 / iters / 16); fflush(stdout);
}
int main()
{
    double* d; (void)hipMalloc(&d, 8 * 320);
    run<false, false, false>("compute only", d, 0x04);
    run<false, false, false>("compute only", d, 0x02);
    run<false, false, false>("compute only", d, 0x0f);
    run<true, false, false>("+LDS writes", d, 0x02);
    run<true, false, false>("+LDS writes", d, 0x0b);
    run<false, true, false>("+LDS reads", d, 0x04);
    run<true, true, false>("+LDS writes+reads", d, 0x0f);
    run<false, false, true>("+barrier", d, 0x0f);
    run<true, true, true>("+LDS writes+reads+barrier", d, 0x0f);
    run<true, true, true>("+LDS writes+reads+barrier", d, 0x04);
    run<true, true, true>("+LDS writes+reads+barrier", d, 0x0b);
    run<false, true, false, 0>("reads, compiler order", d, 0x04);
    run<false, true, false, 2>("reads, interleaved 6:2", d, 0x04);
    run<true, true, true, 0>("all, compiler order", d, 0x0f);
    run<true, true, true, 2>("all, interleaved 6:2", d, 0x0f);
    return 0;
}
