// poisson.hip -- DFT::PoissonSolver (PoissonSolver.h:15-171, PoissonSolver.cpp) on gfx950.
//
// One PERSISTENT 256-thread workgroup (one wave per SIMD of a CU) owns one atom and runs the whole
// FullCycle (Initialize, FMG ramp, up to 100 V-cycles: ~10^4 smoother sweeps over 17 levels) in a single
// launch; the batch dimension (atoms) is the grid.  A launch per sweep would cost ~3*10^4 launches per solve.
//
// Gauss-Seidel is a first-order recurrence in i (PoissonSolver.cpp:48-61):
//     x_i = 0.5 * (S_i + x_{i-1} + x_{i+1}^old - d*(x_{i+1}^old - x_{i-1})*0.5),     |dx_i/dx_{i-1}| = (1+d/2)/2
// Lanes own contiguous chunks of C points; each lane starts W = 96 points early from the OLD values, so the
// error of its start value has decayed by ((1+d/2)/2)^96 < 2^-90 before its first owned point and its chunk
// equals what the sequential sweep computes.  Levels with fewer than 257 nodes (where d grows towards and
// beyond 2) are swept sequentially by one lane -- exactly the reference's loop.
//
// Layout.  A level with n = C*T + 1 nodes is stored lane-interleaved: node i = t*C + k lives at k*T + t
// (node n-1 at C*T), so that at step k the T lanes touch consecutive addresses -- the warm-up, restriction
// and prolongation accesses are coalesced too.  Phi is double-buffered (the sweep reads old right
// neighbours while other lanes overwrite them).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "internal.h"

namespace {

constexpr int kMaxLevels = 24;
constexpr int kThreads = 256;
constexpr int kWarm = 96;   // start-value error decays by <= 0.52^96 < 2^-90: chunked sweep == sequential sweep bit for bit
constexpr int kSeqBelow = 257;   // levels with n < 257 nodes: one lane, sequential

struct Lvl {
    int n;        // nodes
    int logC;     // chunk = 1 << logC
    int logT;     // lanes = 1 << logT   (n - 1 == C * T)
    int seq;      // 1: swept by a single lane in natural order
    long off;     // offset of this level inside the per-atom level storage
    double d;     // deltaGridLevel[l]
};

struct MgDesc {
    int levels;
    long per_atom;   // doubles per atom and per array (sum of n)
    Lvl lv[kMaxLevels];
};

__device__ __forceinline__ long addr(const Lvl& L, int i)
{
    if (i == L.n - 1) return L.off + (L.n - 1);
    return L.off + ((long)(i & ((1 << L.logC) - 1)) << L.logT) + (i >> L.logC);
}
// inverse: storage index -> node
__device__ __forceinline__ int node_of(const Lvl& L, int idx)
{
    if (idx == L.n - 1) return idx;
    return ((idx & ((1 << L.logT) - 1)) << L.logC) + (idx >> L.logT);
}

struct Atom {
    double* phi0;     // two copies of every level
    double* phi1;
    double* src;
    unsigned cur;     // bit l: which copy of level l is current (identical in all threads)
    __device__ __forceinline__ double* cur_phi(int l) const { return ((cur >> l) & 1u) ? phi1 : phi0; }
    __device__ __forceinline__ double* other_phi(int l) const { return ((cur >> l) & 1u) ? phi0 : phi1; }
};

__device__ __forceinline__ double block_sum(double v, double* red)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ double gs_point(double s, double xm, double xp, double dh)
{
    // PoissonSolver.cpp:56-57; d * t * 0.5 == (0.5 d) * t exactly
    return 0.5 * (s + xm + xp - dh * (xp - xm));
}

// one lexicographic Gauss-Seidel sweep of level l: PoissonSolver::GaussSeidel (PoissonSolver.cpp:40-64).
// returns ||dPhi||_2 (same value in every thread)
__device__ double gauss_seidel(const MgDesc& D, Atom& A, int l, double* red)
{
    const Lvl L = D.lv[l];
    const double* __restrict__ S = A.src;
    const double* __restrict__ pin = A.cur_phi(l);
    double* __restrict__ pout = A.other_phi(l);
    const double dh = L.d * 0.5;
    double err2 = 0;
    const int tid = threadIdx.x;
    if (L.seq) {
        if (tid == 0) {
            const long o = L.off;
            double xm = pin[o];
            pout[o] = xm;
            const int limit = L.n - 1;
            for (int i = 1; i < limit; ++i) {
                const double old = pin[o + i];
                const double x = gs_point(S[o + i], xm, pin[o + i + 1], dh);
                const double dif = old - x;
                err2 += dif * dif;
                pout[o + i] = x;
                xm = x;
            }
            pout[o + limit] = pin[o + limit];
        }
    } else {
        const int T = 1 << L.logT, C = 1 << L.logC;
        if (tid < T) {
            const int lo = tid << L.logC;
            const int first = lo > 1 ? lo : 1;
            const int last = lo + C - 1;                       // <= n - 2
            const int i0 = (lo - kWarm) > 1 ? (lo - kWarm) : 1;
            double xm = pin[addr(L, i0 - 1)];                  // old value (exact boundary value when i0 == 1)
            double old = pin[addr(L, i0)];
#pragma unroll 4
            for (int i = i0; i <= last; ++i) {
                const double xp = pin[addr(L, i + 1)];
                const double x = gs_point(S[addr(L, i)], xm, xp, dh);
                if (i >= first) {
                    const double dif = old - x;
                    err2 += dif * dif;
                    pout[addr(L, i)] = x;
                }
                xm = x;
                old = xp;
            }
        }
        if (tid == 0) {
            pout[addr(L, 0)] = pin[addr(L, 0)];
            pout[addr(L, L.n - 1)] = pin[addr(L, L.n - 1)];
        }
    }
    A.cur ^= (1u << l);
    const double tot = block_sum(err2, red);   // also orders the global writes of this sweep before the next phase
    return sqrt(tot);
}

// PoissonSolver::IterateGaussSeidel (PoissonSolver.cpp:66-77)
__device__ double iterate_gs(const MgDesc& D, Atom& A, int l, double errorMin, int iterno, double* red, long* nsweeps)
{
    double err = 1E10;
    for (int i = 0; i < iterno; ++i) {
        err = gauss_seidel(D, A, l, red);
        ++*nsweeps;
        if (err < errorMin) break;
    }
    return err;
}

// PoissonSolver::Restrict(lvl) (PoissonSolver.cpp:126-157): fine = lvl-1 -> coarse = lvl
__device__ void restrict_to(const MgDesc& D, Atom& A, int lvl)
{
    const Lvl Lc = D.lv[lvl], Lf = D.lv[lvl - 1];
    const double* __restrict__ Pf = A.cur_phi(lvl - 1);
    double* __restrict__ Pc = A.cur_phi(lvl);
    double* __restrict__ S = A.src;
    const int lim = Lc.n - 1;
    for (int idx = threadIdx.x; idx < Lc.n; idx += kThreads) {
        const int i = node_of(Lc, idx);
        Pc[Lc.off + idx] = 0;
        double s = 0;
        if (i > 0 && i < lim) {
            const int twoi = 2 * i;
            const double pm = Pf[addr(Lf, twoi - 1)], p0 = Pf[addr(Lf, twoi)], pp = Pf[addr(Lf, twoi + 1)];
            s = 4. * (S[addr(Lf, twoi)] + pm - 2. * p0 + pp) - Lc.d * (pp - pm);
        }
        S[Lc.off + idx] = s;
    }
    __syncthreads();
}

// PoissonSolver::Prolong (PoissonSolver.cpp:110-123): coarse = lvl -> fine = lvl-1 (additive)
__device__ void prolong_from(const MgDesc& D, Atom& A, int lvl)
{
    const Lvl Lc = D.lv[lvl], Lf = D.lv[lvl - 1];
    const double* __restrict__ Pc = A.cur_phi(lvl);
    double* __restrict__ Pf = A.cur_phi(lvl - 1);
    for (int idx = threadIdx.x; idx < Lc.n; idx += kThreads) {
        const int i = node_of(Lc, idx);
        const double c = Pc[Lc.off + idx];
        Pf[addr(Lf, 2 * i)] += c;
        if (i > 0) Pf[addr(Lf, 2 * i - 1)] += 0.5 * (Pc[addr(Lc, i - 1)] + c);
    }
    __syncthreads();
}

struct Counters { long sweeps, vcycles; };

__device__ void ascend(const MgDesc& D, Atom& A, int from, int to, double errorMin, int iterno, double* red, Counters& c)
{   // PoissonSolver.cpp:162-171
    for (int i = from; i < to;) {
        iterate_gs(D, A, i, errorMin, iterno, red, &c.sweeps);
        restrict_to(D, A, ++i);
    }
    iterate_gs(D, A, to, errorMin, iterno, red, &c.sweeps);
}

__device__ double descend(const MgDesc& D, Atom& A, int from, int to, double errorMin, int iterno, double* red, Counters& c)
{   // PoissonSolver.cpp:173-186
    double err = 1E10;
    for (int i = from; i > to;) {
        const int im1 = i - 1;
        prolong_from(D, A, i);
        err = iterate_gs(D, A, im1, errorMin, iterno, red, &c.sweeps);
        i = im1;
    }
    return err;
}

// PoissonSolver::Initialize (PoissonSolver.cpp:80-106)
__device__ void initialize(const MgDesc& D, Atom& A, double lowB, double highB, double errorMin, double* red, Counters& c)
{
    double* __restrict__ S = A.src;
    A.cur = 0;
    {
        const Lvl L0 = D.lv[0];
        for (int idx = threadIdx.x; idx < L0.n; idx += kThreads) A.phi0[L0.off + idx] = 0;
    }
    for (int l = 1; l < D.levels; ++l) {
        const Lvl L = D.lv[l], Lf = D.lv[l - 1];
        __syncthreads();
        for (int idx = threadIdx.x; idx < L.n; idx += kThreads) {
            const int p = node_of(L, idx);
            double s = 0;
            if (p > 0 && p < L.n - 1) s = 4 * S[addr(Lf, 2 * p)];
            S[L.off + idx] = s;
            A.phi0[L.off + idx] = 0;
        }
    }
    __syncthreads();
    const int cl = D.levels - 1;
    if (threadIdx.x == 0) {
        A.phi0[addr(D.lv[cl], 0)] = lowB;
        A.phi0[addr(D.lv[cl], D.lv[cl].n - 1)] = highB;
    }
    __syncthreads();
    iterate_gs(D, A, cl, errorMin, 15, red, &c.sweeps);
}

// PoissonSolver::FullCycle(1E-3, 1E-14) (PoissonSolver.h:89-124)
__device__ double full_cycle(const MgDesc& D, Atom& A, double lowB, double highB, double errorMin, double errorMinLast,
                             double* red, Counters& c)
{
    const int numSweeps = 3;
    const int last = D.levels - 1;
    initialize(D, A, lowB, highB, errorMin, red, c);
    for (int i = D.levels - 2; i > 0; --i) {
        descend(D, A, last, i, errorMin, numSweeps, red, c);
        ascend(D, A, i, last, errorMin, numSweeps, red, c);
    }
    descend(D, A, last, 0, errorMinLast, numSweeps, red, c);
    double err = 0;
    for (int i = 0; i < 100; ++i) {
        ascend(D, A, 0, last, errorMinLast, numSweeps, red, c);          // VCycle, PoissonSolver.h:155-159
        err = descend(D, A, last, 0, errorMinLast, numSweeps, red, c);
        ++c.vcycles;
        if (err < errorMinLast) break;
    }
    return err;
}

// SolvePoissonNonUniform (PoissonSolver.h:51-81): one block per atom
__global__ __launch_bounds__(kThreads) void k_poisson_solve(MgDesc D, double* __restrict__ phi0, double* __restrict__ phi1,
                                                            double* __restrict__ src, const int* __restrict__ Z,
                                                            const double* __restrict__ density, const double* __restrict__ r,
                                                            const double* __restrict__ psrc, double* __restrict__ U,
                                                            int* __restrict__ vcycles, double* __restrict__ errs,
                                                            unsigned long long* __restrict__ total_vcycles)
{
    __shared__ double red[4];
    const int a = blockIdx.x;
    Atom A;
    A.phi0 = phi0 + (size_t)a * D.per_atom;
    A.phi1 = phi1 + (size_t)a * D.per_atom;
    A.src = src + (size_t)a * D.per_atom;
    A.cur = 0;
    const Lvl L0 = D.lv[0];
    const int N = L0.n;
    const double* rho = density + (size_t)a * N;
    // source: Source[i] = r_i; Source[i] *= (4 pi Rp^2 delta^2) exp(2 i delta) * density[i], 1 <= i <= N-2
    for (int idx = threadIdx.x; idx < N; idx += kThreads) {
        const int i = node_of(L0, idx);
        double s = r[i];
        if (i > 0 && i < N - 1) s *= psrc[i] * rho[i];
        A.src[L0.off + idx] = s;
    }
    __syncthreads();
    Counters c{0, 0};
    const double err = full_cycle(D, A, 0.0, (double)Z[a], 1E-3, 1E-14, red, c);
    const double* __restrict__ P = A.cur_phi(0);
    for (int i = threadIdx.x; i < N; i += kThreads) U[(size_t)a * N + i] = P[addr(L0, i)];
    if (threadIdx.x == 0) {
        if (vcycles) vcycles[a] = (int)c.vcycles;
        if (errs) errs[a] = err;
        if (total_vcycles) atomicAdd(total_vcycles, (unsigned long long)c.vcycles);
    }
}

// unit-parity kernels on atom 0 ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_unit(MgDesc D, double* phi0, double* phi1, double* src, int* cur, int op,
                                                   int lvl, int sweeps, double* out)
{
    __shared__ double red[4];
    Atom A;
    A.phi0 = phi0; A.phi1 = phi1; A.src = src;
    A.cur = 0;
    for (int l = 0; l < D.levels; ++l) A.cur |= (cur[l] ? 1u : 0u) << l;
    Counters c{0, 0};
    if (op == 0) {
        for (int s = 0; s < sweeps; ++s) {
            const double e = gauss_seidel(D, A, lvl, red);
            if (threadIdx.x == 0) out[s] = e;
        }
    } else if (op == 1) restrict_to(D, A, lvl);
    else if (op == 2) prolong_from(D, A, lvl);
    else if (op == 3) {
        const int last = D.levels - 1;
        ascend(D, A, 0, last, 1E-14, 3, red, c);
        const double e = descend(D, A, last, 0, 1E-14, 3, red, c);
        if (threadIdx.x == 0) out[0] = e;
    }
    __syncthreads();
    if (threadIdx.x == 0) for (int l = 0; l < D.levels; ++l) cur[l] = (A.cur >> l) & 1u;
}

}  // namespace

struct dfta_poisson {
    dfta_ctx* ctx = nullptr;
    const dfta_grid* g = nullptr;
    int batch = 0;
    MgDesc D;
    double *d_phi0 = nullptr, *d_phi1 = nullptr, *d_src = nullptr;
    int* d_cur = nullptr;           // unit hooks: current buffer per level (atom 0)
    std::vector<int> h_cur;
    unsigned long long* d_total_vcycles = nullptr;
};

static long host_addr(const Lvl& L, int i)
{
    if (i == L.n - 1) return L.off + (L.n - 1);
    return L.off + ((long)(i & ((1 << L.logC) - 1)) << L.logT) + (i >> L.logC);
}

int dfta_poisson_solve_launch(dfta_poisson* p, const int* dZ, const double* dDensity, double* dU, int* dVcycles, double* dErr)
{
    dfta_ctx* ctx = p->ctx;
    hipLaunchKernelGGL(k_poisson_solve, dim3(p->batch), dim3(kThreads), 0, ctx->stream, p->D, p->d_phi0, p->d_phi1, p->d_src, dZ,
                       dDensity, p->g->d_r, p->g->d_psrc, dU, dVcycles, dErr, p->d_total_vcycles);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

int dfta_poisson_take_vcycles(dfta_poisson* p, unsigned long long* out)   // reads and clears the V-cycle counter
{
    dfta_ctx* ctx = p->ctx;
    DFTA_HIP(ctx, hipMemcpyAsync(out, p->d_total_vcycles, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DFTA_HIP(ctx, hipMemsetAsync(p->d_total_vcycles, 0, sizeof(unsigned long long), ctx->stream));
    return DFTA_OK;
}

extern "C" {

int dfta_poisson_create(dfta_ctx* ctx, const dfta_grid* g, int batch, dfta_poisson** out)
{
    if (!ctx || !g || !out) return DFTA_ERR_INVALID;
    DFTA_REQUIRE(ctx, batch >= 1 && g->levels <= kMaxLevels, "poisson batch/levels");
    dfta_poisson* p = new dfta_poisson();
    p->ctx = ctx; p->g = g; p->batch = batch;
    MgDesc& D = p->D;
    D.levels = g->levels;
    long off = 0;
    double d = g->delta;                       // PoissonSolver.cpp:21-26
    int n = g->N;                              // finest level first
    for (int l = 0; l < D.levels; ++l) {
        Lvl& L = D.lv[l];
        L.n = n; L.off = off; L.d = d;
        int lg = 0;
        while ((1 << lg) < n - 1) ++lg;        // n - 1 == 2^lg
        if (n < kSeqBelow) { L.seq = 1; L.logT = 0; L.logC = lg; }
        else { L.seq = 0; L.logT = std::min(lg, 8); L.logC = lg - L.logT; }
        off += n;
        n = (n + 1) / 2;
        d *= 2;
    }
    D.per_atom = off;
    const size_t tot = (size_t)off * batch;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->d_phi0), tot * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_phi1), tot * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_src), tot * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_cur), kMaxLevels * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_total_vcycles), sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(p->d_phi0, 0, tot * sizeof(double), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_phi1, 0, tot * sizeof(double), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_src, 0, tot * sizeof(double), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_cur, 0, kMaxLevels * sizeof(int), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_total_vcycles, 0, sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "poisson alloc: %s", hipGetErrorString(e));
        dfta_poisson_destroy(p);
        return DFTA_ERR_HIP;
    }
    p->h_cur.assign(kMaxLevels, 0);
    *out = p;
    return DFTA_OK;
}

void dfta_poisson_destroy(dfta_poisson* p)
{
    if (!p) return;
    void* ptrs[] = {p->d_phi0, p->d_phi1, p->d_src, p->d_cur, p->d_total_vcycles};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

int dfta_poisson_solve(dfta_poisson* p, const int* Z, const double* density, double* U, int* vcycles_out, double* err_out)
{
    if (!p) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = p->ctx;
    DFTA_REQUIRE(ctx, Z && density && U, "null input");
    const int N = p->g->N, B = p->batch;
    hipStream_t st = ctx->stream;
    DevBuf<int> dZ, dVc;
    DevBuf<double> dRho, dU, dErr;
    DFTA_HIP(ctx, dZ.alloc(B)); DFTA_HIP(ctx, dVc.alloc(B)); DFTA_HIP(ctx, dErr.alloc(B));
    DFTA_HIP(ctx, dRho.alloc((size_t)B * N)); DFTA_HIP(ctx, dU.alloc((size_t)B * N));
    DFTA_HIP(ctx, hipMemcpyAsync(dZ.p, Z, sizeof(int) * B, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dRho.p, density, sizeof(double) * (size_t)B * N, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    int rc = dfta_poisson_solve_launch(p, dZ.p, dRho.p, dU.p, dVc.p, dErr.p);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    ctx->have_kernel_time = true;
    DFTA_HIP(ctx, hipMemcpyAsync(U, dU.p, sizeof(double) * (size_t)B * N, hipMemcpyDeviceToHost, st));
    if (vcycles_out) DFTA_HIP(ctx, hipMemcpyAsync(vcycles_out, dVc.p, sizeof(int) * B, hipMemcpyDeviceToHost, st));
    if (err_out) DFTA_HIP(ctx, hipMemcpyAsync(err_out, dErr.p, sizeof(double) * B, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return DFTA_OK;
}

int dfta_poisson_solve_dev(dfta_poisson* p, const int* dZ, const double* dDensity, double* dU)
{
    if (!p) return DFTA_ERR_INVALID;
    DFTA_REQUIRE(p->ctx, dZ && dDensity && dU, "null input");
    return dfta_poisson_solve_launch(p, dZ, dDensity, dU, nullptr, nullptr);
}

int dfta_poisson_level_size(const dfta_poisson* p, int lvl)
{
    if (!p || lvl < 0 || lvl >= p->D.levels) return -1;
    return p->D.lv[lvl].n;
}

int dfta_poisson_set_level(dfta_poisson* p, int lvl, const double* Phi, const double* Src)
{
    if (!p) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = p->ctx;
    DFTA_REQUIRE(ctx, lvl >= 0 && lvl < p->D.levels, "level");
    const Lvl& L = p->D.lv[lvl];
    std::vector<double> tmp(L.n);
    hipStream_t st = ctx->stream;
    if (Phi) {
        for (int i = 0; i < L.n; ++i) tmp[host_addr(L, i) - L.off] = Phi[i];
        double* dst = (p->h_cur[lvl] ? p->d_phi1 : p->d_phi0) + L.off;
        DFTA_HIP(ctx, hipMemcpyAsync(dst, tmp.data(), sizeof(double) * L.n, hipMemcpyHostToDevice, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
    }
    if (Src) {
        for (int i = 0; i < L.n; ++i) tmp[host_addr(L, i) - L.off] = Src[i];
        DFTA_HIP(ctx, hipMemcpyAsync(p->d_src + L.off, tmp.data(), sizeof(double) * L.n, hipMemcpyHostToDevice, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
    }
    return DFTA_OK;
}

int dfta_poisson_get_level(dfta_poisson* p, int lvl, double* Phi, double* Src)
{
    if (!p) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = p->ctx;
    DFTA_REQUIRE(ctx, lvl >= 0 && lvl < p->D.levels, "level");
    const Lvl& L = p->D.lv[lvl];
    std::vector<double> tmp(L.n);
    hipStream_t st = ctx->stream;
    if (Phi) {
        const double* s = (p->h_cur[lvl] ? p->d_phi1 : p->d_phi0) + L.off;
        DFTA_HIP(ctx, hipMemcpyAsync(tmp.data(), s, sizeof(double) * L.n, hipMemcpyDeviceToHost, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
        for (int i = 0; i < L.n; ++i) Phi[i] = tmp[host_addr(L, i) - L.off];
    }
    if (Src) {
        DFTA_HIP(ctx, hipMemcpyAsync(tmp.data(), p->d_src + L.off, sizeof(double) * L.n, hipMemcpyDeviceToHost, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
        for (int i = 0; i < L.n; ++i) Src[i] = tmp[host_addr(L, i) - L.off];
    }
    return DFTA_OK;
}

static int unit_op(dfta_poisson* p, int op, int lvl, int sweeps, double* out_host, int nout)
{
    dfta_ctx* ctx = p->ctx;
    hipStream_t st = ctx->stream;
    DevBuf<double> dOut;
    DFTA_HIP(ctx, dOut.alloc(std::max(nout, 1)));
    DFTA_HIP(ctx, hipMemcpyAsync(p->d_cur, p->h_cur.data(), sizeof(int) * kMaxLevels, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_unit, dim3(1), dim3(kThreads), 0, st, p->D, p->d_phi0, p->d_phi1, p->d_src, p->d_cur, op, lvl, sweeps, dOut.p);
    DFTA_CHECK_LAUNCH(ctx);
    DFTA_HIP(ctx, hipMemcpyAsync(p->h_cur.data(), p->d_cur, sizeof(int) * kMaxLevels, hipMemcpyDeviceToHost, st));
    if (out_host && nout > 0) DFTA_HIP(ctx, hipMemcpyAsync(out_host, dOut.p, sizeof(double) * nout, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return DFTA_OK;
}

int dfta_poisson_gauss_seidel(dfta_poisson* p, int lvl, int sweeps, double* err_out)
{
    if (!p) return DFTA_ERR_INVALID;
    DFTA_REQUIRE(p->ctx, lvl >= 0 && lvl < p->D.levels && sweeps >= 1 && sweeps <= 1024, "level/sweeps");
    return unit_op(p, 0, lvl, sweeps, err_out, err_out ? sweeps : 0);
}
int dfta_poisson_restrict(dfta_poisson* p, int lvl)
{
    if (!p) return DFTA_ERR_INVALID;
    DFTA_REQUIRE(p->ctx, lvl >= 1 && lvl < p->D.levels, "level");
    return unit_op(p, 1, lvl, 0, nullptr, 0);
}
int dfta_poisson_prolong(dfta_poisson* p, int lvl_src)
{
    if (!p) return DFTA_ERR_INVALID;
    DFTA_REQUIRE(p->ctx, lvl_src >= 1 && lvl_src < p->D.levels, "level");
    return unit_op(p, 2, lvl_src, 0, nullptr, 0);
}
int dfta_poisson_vcycle(dfta_poisson* p, double* err_out)
{
    if (!p) return DFTA_ERR_INVALID;
    return unit_op(p, 3, 0, 0, err_out, err_out ? 1 : 0);
}

}  // extern "C"
