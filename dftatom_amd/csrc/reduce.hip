// reduce.hip -- DFT::Integral (Integral.h:11-155) on the device, in the reference's summation order.
#include <hip/hip_runtime.h>

#include <vector>

#include "internal.h"
#include "ordered_sum.h"

namespace {

__global__ __launch_bounds__(64) void k_simpson38_ordered(const double* __restrict__ vals, int n, size_t stride,
                                                          double* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double lds[dfta::kTile];
    const double* v = vals + (size_t)blockIdx.x * stride;
    const double r = dfta::wave_simpson38(v, n, 1.0, lds);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// Trapezoid / SimpsonOneThird / Simpson38 / Boole (Integral.h:11-104): one wave
__global__ __launch_bounds__(64) void k_newton_cotes(const double* __restrict__ v, int sz, double delta, int rule,
                                                     double* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double lds[dfta::kTile];
    double res = 0;
    const long cnt = static_cast<long>(sz) - 2;   // interior points i = 1 .. sz-2
    if (rule == DFTA_INT_TRAPEZOID) {
        double acc[3] = {0.5 * (v[0] + v[sz - 1]), 0, 0};     // Integral.h:15: the running sum starts from the end terms
        const int cls[1] = {0};
        dfta::wave_ordered_sums<1>(v, 1, 1, cnt, cls, acc, lds);
        res = acc[0] * delta;
    } else if (rule == DFTA_INT_SIMPSON13) {
        double acc[3] = {0, 0, 0};                              // acc0 = sum4 (odd i), acc1 = sum2 (even i)
        const int cls[2] = {0, 1};
        dfta::wave_ordered_sums<2>(v, 1, 1, cnt, cls, acc, lds);
        double sum = v[0] + v[sz - 1];
        sum += 4. * acc[0] + 2. * acc[1];
        constexpr double coef = 1. / 3.;
        res = sum * delta * coef;
    } else if (rule == DFTA_INT_SIMPSON38) {
        res = dfta::wave_simpson38(v, sz, delta, lds);
    } else {                                                    // Boole, Integral.h:75-104
        double acc[3] = {0, 0, 0};                              // acc0 = sum32 (odd i), acc1 = sum12 (i%4==2), acc2 = sum14 (i%4==0)
        const int cls[4] = {0, 1, 0, 2};
        dfta::wave_ordered_sums<4>(v, 1, 1, cnt, cls, acc, lds);
        double sum = 7. * (v[0] + v[sz - 1]);
        sum += 32. * acc[0] + 12. * acc[1] + 14. * acc[2];
        constexpr double coef = 2. / 45.;
        res = sum * delta * coef;
    }
    if (threadIdx.x == 0) out[0] = res;
}

// Romberg (Integral.h:106-155): the trapezoid refinements are strided sums; level i (1-based) adds
// values[n], values[n + oldStep], ... with n = numPoints >> i, oldStep = 2n.  One wave per level computes
// its sum in order; the extrapolation table is then filled by one thread exactly as the reference does.
__global__ __launch_bounds__(64) void k_romberg_sums(const double* __restrict__ v, int numPoints, double* __restrict__ sums)
{
    __shared__ __attribute__((aligned(16))) double lds[dfta::kTile];
    const int i = blockIdx.x + 1;
    const long n = numPoints >> i;
    const long oldStep = numPoints >> (i - 1);   // Integral.h:128-129: oldStep = n before the shift (== 2n only for even n)
    double acc[3] = {0, 0, 0};
    const int cls[1] = {0};
    if (n > 0) {
        const long count = (numPoints - 1 - n) / oldStep + 1;   // j = n, n+oldStep, ... < numPoints
        dfta::wave_ordered_sums<1>(v, n, oldStep, count, cls, acc, lds);
    }
    if (threadIdx.x == 0) sums[i] = acc[0];
}

__global__ void k_romberg_table(const double* __restrict__ v, int numPoints, int cnt, double delta, double err, int minSteps,
                                const double* __restrict__ sums, double* __restrict__ work, double* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double* Rprev = work;
    double* Rcur = work + cnt;
    for (int k = 0; k < cnt; ++k) { Rprev[k] = 0; Rcur[k] = 0; }
    double h = delta * numPoints;
    Rprev[0] = 0.5 * h * (v[0] + v[numPoints]);
    for (int i = 1; i < cnt; ++i) {
        const double sum = sums[i];
        h *= 0.5;
        Rcur[0] = 0.5 * Rprev[0] + h * sum;
        double nk = 1;
        for (int m = 1; m <= i; ++m) {
            nk *= 4;
            Rcur[m] = Rcur[m - 1] + (Rcur[m - 1] - Rprev[m - 1]) / (nk - 1);
        }
        if (i >= minSteps && fabs(Rcur[i] - Rprev[i - 1]) < err) { out[0] = Rcur[i]; return; }
        double* t = Rcur; Rcur = Rprev; Rprev = t;
    }
    out[0] = Rprev[cnt - 1];
}

}  // namespace

int dfta_launch_simpson38_ordered(dfta_ctx* ctx, const double* dVals, int n, int nvec, size_t stride, double* dOut)
{
    hipLaunchKernelGGL(k_simpson38_ordered, dim3(nvec), dim3(64), 0, ctx->stream, dVals, n, stride, dOut);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

extern "C" int dfta_integrate(dfta_ctx* ctx, int rule, double delta, const double* values, int sz, double* result)
{
    if (!ctx) return DFTA_ERR_INVALID;
    if (int rc_ = dfta_use(ctx)) return rc_;
    DFTA_REQUIRE(ctx, values && result && sz >= 3 && rule >= 0 && rule <= DFTA_INT_ROMBERG, "integrate arguments");
    // the reference asserts these shapes (Integral.h:13,27-28,52-53,77-78,110)
    if (rule == DFTA_INT_SIMPSON13 || rule == DFTA_INT_SIMPSON38) DFTA_REQUIRE(ctx, sz >= 5 && sz % 2 == 1, "size");
    if (rule == DFTA_INT_BOOLE) DFTA_REQUIRE(ctx, sz > 4 && sz % 4 == 1, "size");
    if (rule == DFTA_INT_ROMBERG) DFTA_REQUIRE(ctx, sz % 2 == 1, "size");
    hipStream_t st = ctx->stream;
    DevBuf<double> dV, dOut, dSums, dWork;
    DFTA_HIP(ctx, dV.alloc(sz));
    DFTA_HIP(ctx, dOut.alloc(1));
    DFTA_HIP(ctx, hipMemcpyAsync(dV.p, values, sizeof(double) * sz, hipMemcpyHostToDevice, st));
    if (rule != DFTA_INT_ROMBERG) {
        hipLaunchKernelGGL(k_newton_cotes, dim3(1), dim3(64), 0, st, dV.p, sz, delta, rule, dOut.p);
        DFTA_CHECK_LAUNCH(ctx);
    } else {
        const int numPoints = sz - 1;
        int cnt = 0;
        for (int n = numPoints; n; n >>= 1) ++cnt;
        DFTA_HIP(ctx, dSums.alloc(cnt + 1));
        DFTA_HIP(ctx, dWork.alloc(2 * (size_t)cnt));
        if (cnt > 1) {
            hipLaunchKernelGGL(k_romberg_sums, dim3(cnt - 1), dim3(64), 0, st, dV.p, numPoints, dSums.p);
            DFTA_CHECK_LAUNCH(ctx);
        }
        hipLaunchKernelGGL(k_romberg_table, dim3(1), dim3(1), 0, st, dV.p, numPoints, cnt, delta, 1E-18, 3, dSums.p, dWork.p, dOut.p);
        DFTA_CHECK_LAUNCH(ctx);
    }
    DFTA_HIP(ctx, hipMemcpyAsync(result, dOut.p, sizeof(double), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return DFTA_OK;
}
