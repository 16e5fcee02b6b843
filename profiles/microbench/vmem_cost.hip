// per-CU cost of vector memory instructions for one workgroup of 4 (or 8) waves streaming coalesced rows (L2-resident)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(double* buf, double* out, int iters, int rows)
{
    const unsigned t = threadIdx.x;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        const int r0 = (it * 8) % rows;
        if (MODE == 0) {            // 8 x global_load_dwordx2, rows of blockDim doubles
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (buf + (size_t)(r0 + q) * blockDim.x)[t];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        } else if (MODE == 1) {     // 4 x global_load_dwordx4 (same bytes)
            v2d v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ((const v2d*)(buf + (size_t)(r0 + 2 * q) * blockDim.x))[t];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[q].x + v[q].y;
        } else if (MODE == 2) {     // 8 x global_store_dwordx2
#pragma unroll
            for (int q = 0; q < 8; ++q) (buf + (size_t)(r0 + q) * blockDim.x)[t] = acc + q;
            acc += 1;
        } else if (MODE == 3) {     // 4 x global_store_dwordx4
#pragma unroll
            for (int q = 0; q < 4; ++q) { v2d x; x.x = acc; x.y = acc + q; ((v2d*)(buf + (size_t)(r0 + 2 * q) * blockDim.x))[t] = x; }
            acc += 1;
        }
    }
    out[blockIdx.x * blockDim.x + t] = acc;
}
template <int MODE>
void run(const char* name, double* buf, double* out, int threads, int rows, double bytes_per_iter_per_lane, int ninstr)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000; float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(threads), 0, 0, buf, out, iters, rows);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double ns = ms * 1e6 / iters;
    printf("%-28s threads %3d rows %5d: %7.1f ns per batch, %.2f ns per instr per wave, %.1f B/ns per CU\n", name, threads, rows, ns,
           ns / ninstr, threads * bytes_per_iter_per_lane / ns);
    fflush(stdout);
}
int main()
{
    double *buf, *out; (void)hipMalloc(&buf, 8ull * 512 * 16384); (void)hipMalloc(&out, 8 * 512);
    (void)hipMemset(buf, 0, 8ull * 512 * 16384);
    for (int threads : {64, 256, 512}) for (int rows : {64, 8192}) {
        run<0>("8 x load dwordx2", buf, out, threads, rows, 64, 8);
        run<1>("4 x load dwordx4", buf, out, threads, rows, 64, 4);
        run<2>("8 x store dwordx2", buf, out, threads, rows, 64, 8);
        run<3>("4 x store dwordx4", buf, out, threads, rows, 64, 4);
    }
    return 0;
}
