// cost of LDS instructions for a single wave (and for 2/4 waves on different SIMDs)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, int nwaves, int lanes)
{
    __shared__ __attribute__((aligned(16))) double buf[16 * 1024];   // 128 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += 256) buf[i] = 1.0 + i;
    __syncthreads();
    if (wave >= nwaves || lane >= lanes) return;
    double acc = 0;
    const double* base = buf + wave * 4096;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 32 x ds_read_b64, lane-contiguous
            double v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = base[k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 32; ++k) acc += v[k];
        } else if (MODE == 1) {   // 16 x ds_read_b128, lane-contiguous 16 B
            v2d v[16];
            const v2d* b2 = (const v2d*)base;
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = b2[k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k].x + v[k].y;
        } else if (MODE == 2) {   // 32 x ds_write_b64
#pragma unroll
            for (int k = 0; k < 32; ++k) ((double*)base)[k * 64 + lane] = acc + k;
            acc += 1;
        } else if (MODE == 3) {   // 16 x ds_write_b128
            v2d* b2 = (v2d*)base;
#pragma unroll
            for (int k = 0; k < 16; ++k) { v2d t; t.x = acc; t.y = acc + k; b2[k * 64 + lane] = t; }
            acc += 1;
        } else if (MODE == 4) {   // 32 x broadcast ds_read_b64 (same address in all lanes)
            double v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = base[k * 64 + (lane >> 6)];
#pragma unroll
            for (int k = 0; k < 32; ++k) acc += v[k];
        } else if (MODE == 5) {   // 32 x ds_read_b32
            float v[32];
            const float* bf = (const float*)base;
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = bf[k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 32; ++k) acc += v[k];
        }
        asm volatile("" ::: "memory");
    }
    out[threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name, double* d, int nwaves, int lanes, int ninstr, int bytes_per_lane, int valu)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000; float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(256), 0, 0, d, iters, nwaves, lanes);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double ns = ms * 1e6 / iters;
    printf("%-28s waves %d lanes %2d: %7.1f ns per batch; minus %d VALU adds (1.86 ns): %.2f ns per LDS instr, %.1f B/ns per wave\n", name, nwaves,
           lanes, ns, valu, (ns - valu * 1.86) / ninstr, lanes * bytes_per_lane * ninstr / (ns - valu * 1.86));
    fflush(stdout);
}
int main()
{
    double* d; (void)hipMalloc(&d, 8 * 256);
    for (int nw : {1, 2, 4}) for (int lanes : {64, 32}) {
        run<0>("ds_read_b64 x32", d, nw, lanes, 32, 8, 32);
        run<1>("ds_read_b128 x16", d, nw, lanes, 16, 16, 32);
        run<2>("ds_write_b64 x32", d, nw, lanes, 32, 8, 33);
        run<3>("ds_write_b128 x16", d, nw, lanes, 16, 16, 17);
        run<4>("ds_read_b64 broadcast x32", d, nw, lanes, 32, 8, 32);
        run<5>("ds_read_b32 x32", d, nw, lanes, 32, 4, 64);
    }
    return 0;
}
