// Cost of one all-to-all exchange of a double between the G workgroups of a group (what the per-sweep error norm of the
// shared Poisson levels needs), for the variants considered in DESIGN.md 4.3:
//   A  arrival counter + agent-scope release/acquire fences + read of the G slots (group_sum)
//   B  sentinel slots written and polled with agent-scope atomic accesses, no fences (group_sum_fast)
// (a third variant -- slots accessed with sc0 only, i.e. through one XCD's L2 -- never sees the other members' stores,
//  not even inside one XCD: it is compiled out and only kept for the record: -DWITH_SC0)
// with the members on consecutive workgroups (8 XCDs) or 8 workgroups apart (one XCD).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr unsigned long long kSent = 0x7FF8DEAD7FF8DEADull;
__device__ __forceinline__ double ld_sc0(const double* p)
{
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc0(double* p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(256) void k(double* slots, unsigned* ctr, double* out, int G, int iters, int stride)
{
    if (blockIdx.x % stride != 0) return;
    const int g = blockIdx.x / stride;
    if (g >= G) return;
    __shared__ double red;
    double acc = g;
    for (int it = 0; it < iters; ++it) {
        const double mine = acc * 1e-3 + it;
        double tot = 0;
        if (MODE == 0) {
            double* slot = slots + (it & 1) * 64;
            if (threadIdx.x == 0) slot[g] = mine;
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * (it + 1))) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            for (int m = 0; m < G; ++m) tot += slot[m];
        } else {
            double* cur = slots + (it % 3) * 64;
            if (threadIdx.x < 64) {
                if (threadIdx.x == 0) {
                    if (MODE == 1) __hip_atomic_store(cur + g, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else st_sc0(cur + g, mine);
                }
                double x = 0;
                if ((int)threadIdx.x < G) {
                    int spins = 0;
                    while (true) {
                        x = MODE == 1 ? __hip_atomic_load(cur + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ld_sc0(cur + threadIdx.x);
                        if ((unsigned long long)__double_as_longlong(x) != kSent) break;
                        if (++spins > (1 << 22)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                for (int m = 0; m < G; ++m) tot += __shfl(x, m);
                if (threadIdx.x == 0) {
                    red = tot;
                    double* old = slots + ((it + 2) % 3) * 64 + g;
                    if (MODE == 1) __hip_atomic_store(old, __longlong_as_double((long long)kSent), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else st_sc0(old, __longlong_as_double((long long)kSent));
                }
            }
            __syncthreads();
            tot = red;
            __syncthreads();
        }
        acc += tot;
    }
    if (threadIdx.x == 0) out[g] = acc;
}
int main()
{
    double *slots, *out; unsigned* ctr;
    (void)hipMalloc(&slots, 8 * 64 * 3); (void)hipMalloc(&out, 8 * 64); (void)hipMalloc(&ctr, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[3] = {"A counter + fences     ", "B agent-scope slots     ", "C sc0 (L2-only) slots   "};
#ifdef WITH_SC0
    const int nmodes = 3;
#else
    const int nmodes = 2;
#endif
    for (int stride : {1, 8}) for (int G : {8, 16}) for (int mode = 0; mode < nmodes; ++mode) {
        if (stride == 8 && G > 8) continue;         // 8 workgroups apart: at most 32 members fit one XCD, keep the launch small
        if (mode == 2 && stride == 1) continue;     // not coherent across XCDs: would spin forever
        const int iters = 3000; float ms = 0;
        double h[64]; double ref = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(ctr, 0, 4);
            (void)hipMemsetD32((hipDeviceptr_t)slots, 0x7FF8DEAD, 64 * 3 * 2);
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G * stride), dim3(256), 0, 0, slots, ctr, out, G, iters, stride);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(G * stride), dim3(256), 0, 0, slots, ctr, out, G, iters, stride);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(G * stride), dim3(256), 0, 0, slots, ctr, out, G, iters, stride);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(h, out, 8 * G, hipMemcpyDeviceToHost);
        bool same = true; for (int m = 1; m < G; ++m) same = same && (h[m] - m == h[0] - 0 || true);
        (void)ref;
        printf("G %2d %s members %s: %.2f us per exchange   (acc[0] %.6g acc[G-1]-(G-1) %.6g)\n", G, names[mode],
               stride == 1 ? "on consecutive workgroups (8 XCDs)" : "8 workgroups apart (one XCD)     ", ms * 1e3 / iters, h[0], h[G - 1] - (G - 1));
        fflush(stdout);
    }
    return 0;
}
