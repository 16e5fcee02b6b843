// common.h -- shared host-side plumbing of libdftatom_hip (context, grid tables, error handling).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dftatom_hip.h"

struct dfta_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 0;
    char name[128] = {0};
    mutable char err[512] = {0};
    hipEvent_t ev[2] = {nullptr, nullptr};   // bracket the dominant kernel of the last host-pointer call
    bool have_kernel_time = false;
    int sweep_kernel = 0;                    // DFTA_SWEEP_AUTO / _FUSED / _PIPELINED (dfta_ctx_set_sweep_kernel)
};

#define DFTA_HIP(ctx, call)                                                                     \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #call, \
                     hipGetErrorString(e_));                                                    \
            return DFTA_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

// Every C-ABI entry point makes the context's device current first -- the caller (or another library in the process) may
// have changed it since dfta_ctx_create -- and puts the caller's device back when it returns: a process that drives several
// GPUs (torch with another current device) must not find its current device changed by a library call.
struct dfta_device_guard {
    int prev = -1;
    int rc = DFTA_OK;
    explicit dfta_device_guard(dfta_ctx* ctx)
    {
        if (!ctx) { rc = DFTA_ERR_INVALID; return; }
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur == ctx->device) return;      // already current: nothing to do or undo
        const hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) {
            snprintf(ctx->err, sizeof(ctx->err), "hipSetDevice(%d) -> %s", ctx->device, hipGetErrorString(e));
            rc = DFTA_ERR_HIP;
            return;
        }
        prev = cur;
    }
    ~dfta_device_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
    dfta_device_guard(const dfta_device_guard&) = delete;
    dfta_device_guard& operator=(const dfta_device_guard&) = delete;
};
#define DFTA_ENTER(ctx) dfta_device_guard dfta_guard_(ctx); if (dfta_guard_.rc) return dfta_guard_.rc

#define DFTA_CHECK_LAUNCH(ctx) DFTA_HIP(ctx, hipGetLastError())

// Debug / measurement knobs: ONE environment variable, DFTA_DEBUG, a comma-separated list of NAME or NAME=VALUE entries (names
// as listed in DESIGN.md section 9, e.g. DFTA_DEBUG="POISSON_GROUP=3,POISSON_NOFOLD,LEVELS_STATIC").  dfta_knob("NAME") returns
// the value ("" for a bare NAME), or nullptr when the knob is not set.  (The separate DFTA_<NAME> variables of rounds 1-2 are gone.)
// Knobs never change results (the tests prove it for each), only how they are computed.
static inline const char* dfta_knob(const char* name)
{
    static thread_local char val[128];
    if (const char* all = getenv("DFTA_DEBUG")) {
        const size_t n = strlen(name);
        for (const char* p = all; *p;) {
            const char* e = strchr(p, ',');
            const size_t len = e ? static_cast<size_t>(e - p) : strlen(p);
            if (len >= n && strncmp(p, name, n) == 0 && (len == n || p[n] == '=')) {
                const size_t vl = len == n ? 0 : len - n - 1;
                const size_t c = vl < sizeof(val) - 1 ? vl : sizeof(val) - 1;
                memcpy(val, p + n + (len == n ? 0 : 1), c);
                val[c] = 0;
                return val;
            }
            if (!e) break;
            p = e + 1;
        }
    }
    return nullptr;
}

#define DFTA_REQUIRE(ctx, cond, msg)                                               \
    do {                                                                           \
        if (!(cond)) {                                                             \
            if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), "invalid: %s", msg); \
            return DFTA_ERR_INVALID;                                               \
        }                                                                          \
    } while (0)

// roctx ranges (tracing counterpart) around the phases of an SCF step -- level-search rounds, the multigrid
// solve, the XC + integrals tail -- so that a rocprofv3 --marker-trace timeline shows them.  libroctx64 is looked up at run time,
// only under a profiler (rocprofv3 exports ROCP_TOOL_LIBRARIES) or with the ROCTX knob: no link-time dependency, no cost otherwise.
#include <dlfcn.h>
struct dfta_range {
    typedef int (*push_t)(const char*);
    typedef int (*pop_t)();
    static void resolve(push_t& push, pop_t& pop)
    {
        static push_t s_push = nullptr;
        static pop_t s_pop = nullptr;
        static bool tried = false;
        if (!tried) {
            tried = true;
            if (getenv("ROCP_TOOL_LIBRARIES") || dfta_knob("ROCTX")) {
                void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
                if (h) {
                    s_push = reinterpret_cast<push_t>(dlsym(h, "roctxRangePushA"));
                    s_pop = reinterpret_cast<pop_t>(dlsym(h, "roctxRangePop"));
                }
            }
        }
        push = s_push; pop = s_pop;
    }
    bool on = false;
    explicit dfta_range(const char* name)
    {
        push_t push; pop_t pop;
        resolve(push, pop);
        if (push && pop) { push(name); on = true; }
    }
    ~dfta_range()
    {
        if (!on) return;
        push_t push; pop_t pop;
        resolve(push, pop);
        if (pop) pop();
    }
    dfta_range(const dfta_range&) = delete;
    dfta_range& operator=(const dfta_range&) = delete;
};

// Device-resident tables of one logarithmic grid r_i = Rp (exp(i delta) - 1), i = 0..N-1.
// All exp() values are produced on the host with libm in the reference's expression order
// (Numerov.h:79-101,183; DFTAtom.cpp:42,47,334,439-442; PoissonSolver.h:66-74) and uploaded once.
struct dfta_grid {
    dfta_ctx* ctx = nullptr;
    int levels = 0;
    int N = 0;
    double delta = 0, Rmax = 0, Rp = 0, twodelta = 0, Rp2delta2 = 0, delta2p4 = 0;
    // uniform grid r_i = i h (NumerovFunctionRegularGrid, Numerov.h:16-70): delta = 0, h = Rmax / (N - 1); the tables
    // keep their meaning with exp(...) == 1 (d_eh, d_cnst are all ones), so the SCF kernels need no second flavour
    int uniform = 0;
    double h = 1, h2 = 1, h2p12 = 1. / 12.;   // step of the Numerov recurrence (1 on the logarithmic grid, Numerov.h:285-287)
    double far_arg_threshold = 0;   // exp(a) < 1e-200  <=>  a < far_arg_threshold (host libm, monotone)
    double zero1[4] = {0, 0, 0, 0};  // GetBoundaryValueZero(1, l), l = 0..3 (Numerov.h:110-116)
    // host copies
    std::vector<double> h_r, h_e1, h_e2, h_eh;
    // device tables (N doubles each unless noted)
    double* d_r = nullptr;      // r_i
    double* d_e1 = nullptr;     // exp(i delta)
    double* d_e2 = nullptr;     // exp(i 2delta)
    double* d_eh = nullptr;     // exp(i delta / 2)
    double* d_cl = nullptr;     // 4*N: l(l+1.)/(r_i r_i)*0.5 for l = 0..3 (row 0 is all zeros)
    double* d_cnst = nullptr;   // (Rp delta) exp(delta i)            (DFTAtom.cpp:47,442)
    double* d_psrc = nullptr;   // (4 pi Rp^2 delta^2) exp(i 2delta)  (PoissonSolver.h:66-74)
    double* d_fpr2 = nullptr;   // (4 pi r_i) r_i                     (DFTAtom.cpp:340)
    double* d_rsrc = nullptr;   // the r factor of the Poisson source: d_r, or FillR's (Rmax i) / (N-1) on a uniform grid (PoissonSolver.cpp:200-210)
};

template <typename T>
static inline int dfta_alloc(dfta_ctx* ctx, T** p, size_t count)
{
    DFTA_HIP(ctx, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return DFTA_OK;
}

// simple RAII device buffer for call-scoped scratch
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    }
};

// A pointer that reaches a non-inlined device function has no known address space: the compiler emits FLAT loads and stores, which count
// against the LDS counter as well and return out of order with it -- every wait for one of them is a wait for ALL of them (`s_waitcnt
// vmcnt(n) lgkmcnt(0)`), so loads issued ahead of their use hide nothing, and wave-uniform addresses cannot use scalar loads.
// dfta_as_global / dfta_as_shared re-derive the pointer through its address space (a flat address of global memory IS the global address;
// the LDS offset is the low half of a flat LDS address): the address-space inference then turns every access that descends from the
// result into a global / LDS instruction.  (`__builtin_assume(!is_shared && !is_private)` is not enough: it survives only where the
// assumed value itself is the base of the access.)
#if defined(__HIPCC__)
#if defined(__HIP_DEVICE_COMPILE__)
#define DFTA_AS_CAST(T, N, p) ((T*)(T __attribute__((address_space(N)))*)(unsigned long long)(p))
#else
#define DFTA_AS_CAST(T, N, p) (p)      // host pass: the device functions are parsed, never run
#endif
template <typename T> __device__ __forceinline__ T* dfta_as_global(T* p) { return DFTA_AS_CAST(T, 1, p); }
// read-only for the whole kernel (tables built by an earlier launch): the constant address space, so that wave-uniform addresses are
// read through the scalar cache (`s_load`) -- a non-kernel function cannot prove "not written meanwhile" of a global pointer
template <typename T> __device__ __forceinline__ const T* dfta_as_constant(const T* p) { return DFTA_AS_CAST(const T, 4, p); }
template <typename T> __device__ __forceinline__ T* dfta_as_shared(T* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (T*)(T __attribute__((address_space(3)))*)(unsigned)(unsigned long long)p;
#else
    return p;
#endif
}
// the arguments of a non-kernel function arrive in vector registers and count as divergent: a pointer that is the same in every lane is
// declared so (two v_readfirstlane), which gives scalar address arithmetic, the `saddr` form of vector loads and scalar loads
template <typename T> __device__ __forceinline__ T* dfta_uniform(T* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
#else
    return p;
#endif
}
#endif
