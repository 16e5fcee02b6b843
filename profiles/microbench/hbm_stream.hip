// Attainable HBM bandwidth of the device: copy (c = a) and triad (a = b + s c) over arrays far larger than the 256 MB Infinity
// Cache, for several launch shapes (blocks per CU x 16-byte accesses in flight per lane).  The best copy figure is the "measured"
// denominator that bench.py quotes next to the 8 TB/s spec figure (the library's dfta_ctx_measure_hbm uses the best shape found here).
//   hipcc --offload-arch=gfx950 -O3 -o hbm_stream hbm_stream.hip && ./hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U>
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ c, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2 x[U];
#pragma unroll
        for (int q = 0; q < U; ++q) x[q] = a[i + q * stride];
#pragma unroll
        for (int q = 0; q < U; ++q) c[i + q * stride] = x[q];
    }
    for (; i < n2; i += stride) c[i] = a[i];
}
template <int U>
__global__ __launch_bounds__(256) void k_triad(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c, double s, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2 x[U], y[U];
#pragma unroll
        for (int q = 0; q < U; ++q) { x[q] = b[i + q * stride]; y[q] = c[i + q * stride]; }
#pragma unroll
        for (int q = 0; q < U; ++q) a[i + q * stride] = make_double2(x[q].x + s * y[q].x, x[q].y + s * y[q].y);
    }
    for (; i < n2; i += stride) { const double2 x = b[i], y = c[i]; a[i] = make_double2(x.x + s * y.x, x.y + s * y.y); }
}
template <int U>
static void run(double2* a, double2* b, double2* c, size_t n2, int ncu, int bpc)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best[2] = {0, 0};
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(k_copy<U>, dim3(ncu * bpc), dim3(256), 0, 0, a, c, n2);
            else hipLaunchKernelGGL(k_triad<U>, dim3(ncu * bpc), dim3(256), 0, 0, a, b, c, 3.0, n2);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            const double gbs = (which == 0 ? 2.0 : 3.0) * n2 * 16 / (ms * 1e-3) / 1e9;
            if (rep > 0 && gbs > best[which]) best[which] = gbs;
        }
    printf("blocks/CU %3d  x%d in flight: copy %7.1f GB/s  triad %7.1f GB/s\n", bpc, U, best[0], best[1]);
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const size_t n2 = (size_t)1 << 26;             // 1 GiB per array
    double2 *a, *b, *c;
    (void)hipMalloc(&a, n2 * 16); (void)hipMalloc(&b, n2 * 16); (void)hipMalloc(&c, n2 * 16);
    (void)hipMemset(a, 0, n2 * 16); (void)hipMemset(b, 0, n2 * 16); (void)hipMemset(c, 0, n2 * 16);
    printf("%s, %d CUs, arrays of 1 GiB\n", p.name, p.multiProcessorCount);
    for (int bpc : {2, 4, 8, 16, 32}) { run<1>(a, b, c, n2, p.multiProcessorCount, bpc); run<2>(a, b, c, n2, p.multiProcessorCount, bpc); run<4>(a, b, c, n2, p.multiProcessorCount, bpc); }
    return 0;
}
