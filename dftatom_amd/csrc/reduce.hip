// reduce.hip -- DFT::Integral (Integral.h:11-155) on the device, in the reference's summation order.
#include <hip/hip_runtime.h>

#include <vector>

#include "internal.h"
#include "ordered_sum.h"

namespace {

// one block of two waves per vector: out[k] = Integral::<rule>(delta, vals + k * stride); Simpson 3/8 -- the live rule -- on both
// waves (its two sums are independent chains), the other rules on the first
__global__ __launch_bounds__(128) void k_integrate(const double* __restrict__ vals, int n, size_t stride, int rule, double delta,
                                                   double* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double lds[dfta::kTile];
    __shared__ double rtab[64];
    const double* v = vals + (size_t)blockIdx.x * stride;
    double r = 0;
    if (rule == DFTA_INT_SIMPSON38) r = dfta::block_simpson38(v, n, delta, lds, rtab);
    else if (threadIdx.x < 64)      r = dfta::wave_integrate(rule, v, n, delta, lds, rtab);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

}  // namespace

int dfta_launch_integrate_ordered(dfta_ctx* ctx, int rule, double delta, const double* dVals, int n, int nvec, size_t stride, double* dOut)
{
    hipLaunchKernelGGL(k_integrate, dim3(nvec), dim3(128), 0, ctx->stream, dVals, n, stride, rule, delta, dOut);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

// the reference asserts these shapes (Integral.h:13,27-28,52-53,77-78,110)
int dfta_integral_shape_ok(int rule, int sz)
{
    if (rule < 0 || rule > DFTA_INT_ROMBERG || sz < 3) return 0;
    if (rule == DFTA_INT_SIMPSON13 || rule == DFTA_INT_SIMPSON38) return sz >= 5 && sz % 2 == 1;
    if (rule == DFTA_INT_BOOLE) return sz > 4 && sz % 4 == 1;
    if (rule == DFTA_INT_ROMBERG) return sz % 2 == 1;
    return 1;
}

extern "C" int dfta_integrate(dfta_ctx* ctx, int rule, double delta, const double* values, int sz, double* result)
{
    if (!ctx) return DFTA_ERR_INVALID;
    if (int rc_ = dfta_use(ctx)) return rc_;
    DFTA_REQUIRE(ctx, values && result && sz >= 3 && rule >= 0 && rule <= DFTA_INT_ROMBERG, "integrate arguments");
    DFTA_REQUIRE(ctx, dfta_integral_shape_ok(rule, sz), "size");
    hipStream_t st = ctx->stream;
    DevBuf<double> dV, dOut;
    DFTA_HIP(ctx, dV.alloc(sz));
    DFTA_HIP(ctx, dOut.alloc(1));
    DFTA_HIP(ctx, hipMemcpyAsync(dV.p, values, sizeof(double) * sz, hipMemcpyHostToDevice, st));
    if (int rc = dfta_launch_integrate_ordered(ctx, rule, delta, dV.p, sz, 1, (size_t)sz, dOut.p)) return rc;
    DFTA_HIP(ctx, hipMemcpyAsync(result, dOut.p, sizeof(double), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return DFTA_OK;
}
