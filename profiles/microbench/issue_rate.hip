// per-wave cost of a producer-like fp64 instruction mix vs number of waves in the workgroup (one workgroup, one CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_prod(double* out, int iters, double E, double R2, double d2p4, int active_mask)
{
    const int wave = threadIdx.x >> 6;
    if (!((active_mask >> wave) & 1)) return;
    double acc = 0;
    double x0 = 1.0 + threadIdx.x * 1e-3, x1 = x0 * 1.1, x2 = x0 * 1.2, x3 = x0 * 1.3, y = 0.999;
    for (int i = 0; i < iters; ++i) {
        double xs[4] = {x0, x1, x2, x3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double f = (xs[k] - E) * R2 * y + d2p4;
            const double d = 1. - (1. / 12.) * f;
            double rr = __builtin_amdgcn_rcp(d);
            double e = __builtin_fma(-d, rr, 1.0);
            rr = __builtin_fma(rr, e, rr);
            e = __builtin_fma(-d, rr, 1.0);
            rr = __builtin_fma(rr, e, rr);
            acc += rr;
        }
        x0 += 1e-9; x1 += 1e-9; x2 += 1e-9; x3 += 1e-9;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// integrator-like dependent chain
__global__ void k_chain(double* out, int iters, double r, double d, double f, int active_mask)
{
    const int wave = threadIdx.x >> 6;
    if (!((active_mask >> wave) & 1)) return;
    double w = 1.0 + threadIdx.x * 1e-3, wprev = 0.99, u = 1.0, fprev = f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double wnext = __builtin_fma(2., w, -wprev) + u * fprev;
            wprev = w; w = wnext;
            const double q = wnext * r;
            const double rem = __builtin_fma(-d, q, wnext);
            u = __builtin_fma(rem, r, q);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = u + w;
}
int main()
{
    double* d; hipMalloc(&d, 8 * 1024 * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int masks[] = {0x1, 0x3, 0xf, 0x11, 0x33, 0x3f, 0xff, 0x5, 0x55};
    for (int which = 0; which < 2; ++which)
    for (int m : masks) {
        const int iters = 100000;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(k_prod, dim3(1), dim3(512), 0, 0, d, iters, -100.0, 1e-8, 2.5e-9, m);
            else            hipLaunchKernelGGL(k_chain, dim3(1), dim3(512), 0, 0, d, iters, 1.0000001, 0.9999999, 1e-9, m);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double instr = which == 0 ? 4 * 12.0 + 4 : 16 * 6.0;
        printf("%s wave mask 0x%02x: %.3f ms -> %.2f ns per loop trip, %.2f ns per VALU instr\n", which == 0 ? "producer" : "chain   ", m, ms,
               ms * 1e6 / iters, ms * 1e6 / iters / instr);
        fflush(stdout);
    }
    return 0;
}
