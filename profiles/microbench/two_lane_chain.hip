// What does a step of k_match's integrator cost?  One wave, lanes 0 and 1 live: the Numerov recurrence (6 fp64
// instructions per step) fed from LDS (16-byte reads of {f, r}, 8-byte d) and writing u back to LDS, in groups of 16.
// Variants: full / without the LDS writes / without the LDS reads / arithmetic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int groups, double dseed)
{
    __shared__ v2d fr[2][64];
    __shared__ double dd[2][64];
    __shared__ double uo[2][64];
    const int lane = threadIdx.x;
    fr[0][lane] = v2d{1e-9 * lane, 1.0000001};
    fr[1][lane] = v2d{2e-9 * lane, 0.9999999};
    dd[0][lane] = dd[1][lane] = dseed;
    __syncthreads();
    double w = 1.0 + lane * 1e-3, wprev = 0.99, u = 1.0, fprev = 1e-9, acc = 0;
    if (lane < 2) {
        const v2d* mine = &fr[lane][0];
        const double* mined = &dd[lane][0];
        double* outp = &uo[lane][0];
        for (int g = 0; g < groups; ++g) {
            const int k0 = (g & 3) * 16;
            v2d in16[16];
            double d16[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (MODE == 0 || MODE == 1) { in16[q] = mine[k0 + q]; d16[q] = mined[k0 + q]; }
                else { in16[q] = v2d{1e-9 * q + fprev * 1e-30, 1.0000001}; d16[q] = dseed; }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double wnext = __builtin_fma(2., w, -wprev) + u * fprev;
                wprev = w; w = wnext;
                const double qq = wnext * in16[q].y;
                const double rem = __builtin_fma(-d16[q], qq, wnext);
                u = __builtin_fma(rem, in16[q].y, qq);
                fprev = in16[q].x;
                if (MODE == 0 || MODE == 2) outp[k0 + q] = u; else acc += u * 1e-300;
            }
        }
    }
    out[lane] = u + w + acc + uo[0][lane & 63];
}
int main()
{
    double* d; (void)hipMalloc(&d, 8 * 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[4] = {"reads + recurrence + writes", "reads + recurrence         ", "recurrence + writes        ", "recurrence only            "};
    for (int mode = 0; mode < 4; ++mode) {
        const int groups = 200000; float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, groups, 0.9999999);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, groups, 0.9999999);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, groups, 0.9999999);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, groups, 0.9999999);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s: %.2f ns per step\n", names[mode], ms * 1e6 / (groups * 16.0));
        fflush(stdout);
    }
    return 0;
}
