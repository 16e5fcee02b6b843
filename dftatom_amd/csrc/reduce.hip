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

// Simpson 3/8 with parallel sums: one block per vector, thread t adds the nodes t + 1, t + 257, ... to sum1 (i % 3 != 0) or sum2 (i % 3 == 0)
__global__ __launch_bounds__(256) void k_integrate_simpson38_par(const double* __restrict__ vals, int n, size_t stride, double delta, double* __restrict__ out)
{
    __shared__ double red[8];
    const double* v = vals + (size_t)blockIdx.x * stride;
    double s1 = 0, s2 = 0;
    for (int i = 1 + threadIdx.x; i < n - 1; i += 256) {
        const double x = v[i];
        if (i % 3 == 0) s2 += x; else s1 += x;
    }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s1; red[4 + (threadIdx.x >> 6)] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sum1 = (red[0] + red[1]) + (red[2] + red[3]), sum2 = (red[4] + red[5]) + (red[6] + red[7]);
        double sum = v[0] + v[n - 1];
        sum += 3. * sum1 + 2. * sum2;
        constexpr double coef = 3. / 8.;
        out[blockIdx.x] = sum * delta * coef;
    }
}

// Attainable HBM bandwidth of this device: the second denominator of every roofline figure (SURVEY.md 8d asks for the spec
// figure AND a stream measurement on the box).  Plain grid-stride kernels, 16 bytes per lane and access, buffers far larger
// than the 256 MB Infinity Cache; launch shape from profiles/microbench/hbm_stream.hip (results_r03.txt).
__global__ __launch_bounds__(256) void k_stream_copy(const double2* __restrict__ a, double2* __restrict__ c, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) c[i] = a[i];
}
__global__ __launch_bounds__(256) void k_stream_triad(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c,
                                                      double s, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const double2 x = b[i], y = c[i];
        a[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
    }
}

}  // namespace

extern "C" int dfta_ctx_measure_hbm(dfta_ctx* ctx, size_t doubles_per_array, int reps, double* copy_gbs, double* triad_gbs)
{
    if (!ctx) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, doubles_per_array >= (1u << 20) && doubles_per_array % 2 == 0 && reps >= 1 && reps <= 1000, "measure_hbm arguments");
    const size_t n = doubles_per_array, n2 = n / 2;
    DevBuf<double> A, B, Cc;
    DFTA_HIP(ctx, A.alloc(n)); DFTA_HIP(ctx, B.alloc(n)); DFTA_HIP(ctx, Cc.alloc(n));
    hipStream_t st = ctx->stream;
    DFTA_HIP(ctx, hipMemsetAsync(A.p, 0, n * sizeof(double), st));
    DFTA_HIP(ctx, hipMemsetAsync(B.p, 0, n * sizeof(double), st));
    DFTA_HIP(ctx, hipMemsetAsync(Cc.p, 0, n * sizeof(double), st));
    const int blocks = ctx->num_cu * 4;       // the best shape of profiles/microbench/hbm_stream.hip on MI355X: 5.5 TB/s copy
    double best[2] = {0, 0};
    for (int which = 0; which < 2; ++which) {
        for (int r = 0; r < reps + 1; ++r) {            // first repetition: warm-up
            DFTA_HIP(ctx, hipEventRecord(ctx->ev[0], st));
            if (which == 0) hipLaunchKernelGGL(k_stream_copy, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const double2*>(A.p), reinterpret_cast<double2*>(Cc.p), n2);
            else hipLaunchKernelGGL(k_stream_triad, dim3(blocks), dim3(256), 0, st, reinterpret_cast<double2*>(A.p), reinterpret_cast<const double2*>(B.p),
                                    reinterpret_cast<const double2*>(Cc.p), 3.0, n2);
            DFTA_CHECK_LAUNCH(ctx);
            DFTA_HIP(ctx, hipEventRecord(ctx->ev[1], st));
            DFTA_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
            float ms = 0;
            DFTA_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
            const double gbs = (which == 0 ? 2.0 : 3.0) * n * sizeof(double) / (ms * 1e-3) / 1e9;
            if (r > 0 && gbs > best[which]) best[which] = gbs;
        }
    }
    if (copy_gbs) *copy_gbs = best[0];
    if (triad_gbs) *triad_gbs = best[1];
    return DFTA_OK;
}

int dfta_launch_integrate_ordered(dfta_ctx* ctx, int rule, double delta, const double* dVals, int n, int nvec, size_t stride, double* dOut)
{
    hipLaunchKernelGGL(k_integrate, dim3(nvec), dim3(128), 0, ctx->stream, dVals, n, stride, rule, delta, dOut);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

int dfta_launch_integrate_simpson38_parallel(dfta_ctx* ctx, double delta, const double* dVals, int n, int nvec, size_t stride, double* dOut)
{
    hipLaunchKernelGGL(k_integrate_simpson38_par, dim3(nvec), dim3(256), 0, ctx->stream, dVals, n, stride, delta, dOut);
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
    DFTA_ENTER(ctx);
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
