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
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "internal.h"
#include "xc.h"

namespace {

constexpr int kMaxLevels = 24;
constexpr int kThreads = 256;
#ifndef DFTA_KWARM
#define DFTA_KWARM 96
#endif
constexpr int kWarm = DFTA_KWARM;   // start-value error decays by <= 0.52^96 < 2^-90: chunked sweep == sequential sweep bit for bit
constexpr int kSeqBelow = 129;   // levels with n < 129 nodes: one lane, sequential, LDS-resident
constexpr int kWaveMaxN = 1025;  // staged levels up to this size are swept by the first wave alone (64 lanes: a quarter of the LDS traffic per warm-up step)
constexpr int kSeqCap = 144;     // LDS doubles per array for the sequential levels (65+33+17+9+5+3 = 132)
constexpr int kPF = 8;           // register prefetch depth of the chunked sweep
constexpr int kFuseMinLogC = 9;  // fuse the three sweeps of a level visit when every lane owns >= 512 nodes
constexpr int kStageMaxLogC = 5;   // chunked levels with <= 32 nodes per lane are swept from a copy in LDS (Phi, S)
constexpr int kStagePad = 128;     // one workgroup: doubles in front of each staged array (warm-up reads of the first lanes)
constexpr int kStageH = 24;        // group members: halo columns in front of every staged row (>= 96/C lanes, C >= 4)
constexpr int kStageRS = kThreads + kStageH;          // row stride of a member's staged part
constexpr int kStageArr = 8992;    // doubles per staged array: >= kStagePad + 8193 and >= kStageH + 32*kStageRS + 1
constexpr int kPad = 320;        // doubles of padding in front of every atom's level storage (warm-up reads of lane 0)

struct Lvl {
    int n;        // nodes
    int logC;     // chunk = 1 << logC
    int logT;     // lanes = 1 << logT   (n - 1 == C * T)
    int seq;      // 1: swept by a single lane in natural order
    int stage;    // swept from a copy in LDS (the CU's vector-memory path is the bound otherwise): 1 = one workgroup's level, 2 = shared level
    long off;     // offset of this level inside the per-atom level storage
    long soff;    // sequential levels: offset inside the LDS-resident copy
    double d;     // deltaGridLevel[l]
};

struct MgDesc {
    int levels;
    int G;           // workgroups per atom (power of two); 1: the whole solve runs in one workgroup
    int logG;
    int dbg;         // $DFTA_POISSON_DBG, measurements only (results are garbage): 1 = the coarse workgroup skips its sweeps, 2 = the members skip their passes, 4 = no restriction / prolongation on the shared levels
    int res_kres;    // > 0: resident group (k_poisson_solve_res): levels 0 .. res_kres-1 live in the members' LDS; the level layout is that of G = 1
    int res_logC0;   // log2(nodes per lane) of level 0 in a member's stretch (kResG members x kResNT lanes)
    int fuse3;    // visits of three sweeps on staged levels of one workgroup run as ONE fused pass (gs_lds3); 0: $DFTA_POISSON_NOFUSE3
    int nofold;   // DFTA_POISSON_NOFOLD: restriction / prolongation as separate passes even where they could be folded into a staged copy-in
    int kcoop;       // levels 0 .. kcoop-1 are swept by all G workgroups together (256 G lanes), the others by workgroup 0
    int spin_max;    // bound of the group barriers' spin loops (Atom::spin_max)
    long per_atom;   // doubles per atom and per array (sum of n)
    // Coarse section (coarse_section below): levels cs_top .. levels-1 of a V-cycle are handled by the first wave of
    // workgroup 0 alone, entirely in LDS.  -1: off.  cs_phi / cs_src: offsets of a level's arrays inside the staging memory
    // (doubles), cs_lc: log2(nodes per lane) of its 64-lane interleaved layout, or -1 for natural order.
    int cs_top;
    int cs_phi[kMaxLevels], cs_src[kMaxLevels], cs_lc[kMaxLevels];
    Lvl lv[kMaxLevels];
};

// storage index of node i RELATIVE to the start of its level
__device__ __forceinline__ int addr(const Lvl& L, int i)
{
    if (i == L.n - 1) return L.n - 1;
    return ((i & ((1 << L.logC) - 1)) << L.logT) + (i >> L.logC);
}
// inverse: storage index -> node
__device__ __forceinline__ int node_of(const Lvl& L, int idx)
{
    if (idx == L.n - 1) return idx;
    return ((idx & ((1 << L.logT) - 1)) << L.logC) + (idx >> L.logT);
}

// Level storage of one atom.  The chunked levels live in global memory (L2-resident: 6.3 MB per atom at 17
// levels); the sequential levels (n < 257, 261 nodes in total) live in LDS for the whole solve -- they are visited
// 6 times per V-cycle by a single lane and would otherwise pay a global-memory round trip per node.
constexpr unsigned long long kFastSentinel = 0x7FF8DEAD7FF8DEADull;
constexpr int kXchg = 128;       // doubles per member and buffer of the boundary exchange (<= 96 halo nodes; the first node in the last one)
// per atom: [6 G + 2] partial sums of the members and the published state, [3 G] slots of the fast sum, [3 G kXchg] boundary
// nodes exchanged between neighbours in the middle of a staged visit
__host__ __device__ constexpr size_t group_part_doubles(int G) { return (size_t)9 * G + 2 + (size_t)3 * G * kXchg; }

struct Atom {
    double* phi0;     // two copies of every level (global)
    double* phi1;
    double* src;
    double* lds;      // shared memory: [phi copy 0 | phi copy 1 | src], kSeqCap doubles each
    double* stage;    // shared memory: two arrays of kStageArr doubles (Phi, S) for the staged sweeps of one level visit
    unsigned cur;     // bit l: which copy of level l is current (identical in all threads)
    // group of G workgroups that share the fine levels of this atom (G == 1: none of this is touched)
    int g, G;               // member index, group size
    unsigned* ctr;          // monotonic arrival counter of the group (zeroed before the launch)
    unsigned bar;           // barriers passed so far
    double* part;           // [2][3][G] partial sums of the members (double-buffered by barrier parity) + [1] published `cur`
    double* fslot;          // [3][G] slots of the fast sum (group_sum_fast), sentinel-filled before the launch
    unsigned fseq;          // fast sums taken so far
    double* xchg;           // [3][G][kXchg] boundary nodes of the fast exchange, sentinel-filled before the launch
    int pend;               // > 0: the prolongation from this level is folded into the staged copy-in of the level below it
    int pend_r;             // > 0: the restriction TO this level is folded into its staged copy-in
    int pend_z;             // > 0: this level's Phi is zero and its source complete in global memory (written by the members of a resident group): staged as such
    int spin_max;           // polls of a group barrier before the waiting member gives up and raises the abort flag
    bool gave_up;           // this thread has timed out on a slot of group_sum_fast: it does not wait for that member again
    __device__ __forceinline__ int lane() const { return g * kThreads + static_cast<int>(threadIdx.x); }
    // pointer to storage element 0 of the level (generic address space: LDS for sequential levels, global otherwise)
    __device__ __forceinline__ double* cur_phi(int l, const Lvl& L) const
    {
        const unsigned c = (cur >> l) & 1u;
        return L.seq ? lds + c * kSeqCap + L.soff : (c ? phi1 : phi0) + L.off;
    }
    __device__ __forceinline__ double* other_phi(int l, const Lvl& L) const
    {
        const unsigned c = ((cur >> l) & 1u) ^ 1u;
        return L.seq ? lds + c * kSeqCap + L.soff : (c ? phi1 : phi0) + L.off;
    }
    __device__ __forceinline__ double* src_of(const Lvl& L) const { return L.seq ? lds + 2 * kSeqCap + L.soff : src + L.off; }
};

// Barrier of the G workgroups of an atom: every store made before it by any member is visible to every member after it
// (agent-scope release before the arrival, acquire after the last one; MI355X_MICROARCH.md: the per-XCD L2s are not
// coherent).  ~2 us for 8 members (profiles/microbench).  The spin is bounded so that a lost member cannot hang the GPU.
__device__ __forceinline__ void group_sync(Atom& A)
{
    if (A.G == 1) { __syncthreads(); return; }
    __syncthreads();
    ++A.bar;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(A.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = A.bar * static_cast<unsigned>(A.G);
        // a member that never arrives (its workgroup was not scheduled: the group does not fit the free compute units)
        // must not hang the GPU: after ~1 s the waiting member raises the group's abort flag (the counter's top bit),
        // which releases every later barrier at once; the host sees the flag and reports the failure
        int spins = 0;
        unsigned seen;
        while ((seen = __hip_atomic_load(A.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > A.spin_max) { __hip_atomic_fetch_or(A.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __forceinline__ double block_sum(double v, double* red)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Sum over all lanes of the group (cooperative levels): the members' block sums are exchanged through global memory around
// a group barrier and added in member order by everybody, so that every member takes the same decisions.
// `which` = 0..2 selects one of three concurrent sums (the fused three-sweep pass is never cooperative, kept for symmetry).
__device__ __forceinline__ double group_sum(Atom& A, double v, double* red)
{
    const double mine = block_sum(v, red);
    if (A.G == 1) return mine;
    double* slot = A.part + ((A.bar + 1) & 1u) * 3 * A.G;      // parity of the barrier that follows
    if (threadIdx.x == 0) slot[A.g] = mine;
    group_sync(A);
    double tot = 0;
    for (int m = 0; m < A.G; ++m) tot += slot[m];
    return tot;
}

// -DDFTA_POISSON_PROF: time (s_memtime ticks of workgroup 0) per operation kind and level, printed when the solver is destroyed
#ifdef DFTA_POISSON_PROF
__device__ unsigned long long g_prof[8 * 24];
#define PROF_T0() const long long prof_t0 = clock64()
__device__ unsigned long long g_prof_member[64 * 4];     // per member of atom 0: ticks in the exchange / in the LDS sweeps (all shared levels)
#define PROF_ADD(cat, lvl) do { if (threadIdx.x == 0) { const unsigned long long dt_ = clock64() - prof_t0; \
        if (blockIdx.x == 0) g_prof[(cat) * 24 + (lvl)] += dt_; \
        if (blockIdx.x < 64 && ((cat) == 3 || (cat) == 4 || (cat) == 5)) g_prof_member[blockIdx.x * 4 + (cat) - 3] += dt_; } } while (0)
#else
#define PROF_T0()
#define PROF_ADD(cat, lvl)
#endif

// Sum over the group for the sweeps in the middle of a staged visit, where the members exchange only a partial sum and the
// <= 97 nodes next to their parts' boundaries.  Everything travels through agent-scope atomic stores into slots that hold a
// sentinel (a NaN with a payload no arithmetic produces) and is polled with agent-scope atomic loads until the sentinel is
// gone: coherent without cache maintenance, and every datum validates itself -- an agent-scope store can overtake an earlier
// one on its way to another XCD (measured), so "the sum has arrived" says nothing about the nodes.  One round trip for
// everything: while the first wave polls the G sums, the second and third poll the left neighbour's nodes straight into
// the halo columns and the fourth the right neighbour's first node (round 1: arrival counter, poll, read of the sums, then
// the nodes: three trips and two fences).  Three buffers rotate: when the sums of exchange s are complete everybody has
// read what it needed of exchange s-1 (they have all published s), so each member resets its part of that buffer; it is
// used again in exchange s+2, a whole sweep later.  Sums are added in member order, as in group_sum.
__device__ __forceinline__ double exchange_poll(Atom& A, const double* p)
{
    int spins = A.gave_up ? A.spin_max : 0;
    while (true) {
        const double x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (static_cast<unsigned long long>(__double_as_longlong(x)) != kFastSentinel) return x;
        if (++spins > A.spin_max) {      // a lost member must not hang the GPU: raise the group's abort flag
            __hip_atomic_fetch_or(A.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            A.gave_up = true;
            return 0.0;
        }
        // somebody else has already given up on this group (the abort bit of the arrival counter): do not spin out the
        // whole bound again in every thread at every later exchange
        if ((spins & 1023) == 0 && (__hip_atomic_load(A.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x80000000u)) {
            A.gave_up = true;
            return 0.0;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// this member's node `idx` of the exchange that travels with the next fast sum
__device__ __forceinline__ void exchange_store(Atom& A, int idx, double v)
{
    __hip_atomic_store(A.xchg + (static_cast<size_t>(A.fseq % 3u) * A.G + A.g) * kXchg + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// PP: the staged part (element 0 = its first node); nhalo = Hc << logC nodes of the left neighbour go to the halo columns
// (node idx: row idx & (C-1), column -Hc + (idx >> logC)), the right neighbour's first node to PP[C * RS]
// FENCED: the sum that ends a visit, after the member has written its part out with plain stores -- a release before the sum
// is published and an acquire after the last one has arrived make those stores visible to every member (the next operation
// reads other members' columns with plain loads); no nodes travel with it.  One trip instead of the three of group_sum
// (arrival counter, poll, read of the slots).
template <int RS, bool FENCED = false>
__device__ __forceinline__ double group_sum_fast(Atom& A, double v, double* red, double* PP, int logC, int Hc)
{
    const double mine = block_sum(v, red);      // its barriers wait for every store of this member issued so far
    if (A.G == 1) return mine;
    const unsigned s = A.fseq++;
    const int tid = threadIdx.x, C = 1 << logC;
    if (tid < 64) {
        double* cur = A.fslot + (s % 3u) * A.G;
        if (FENCED && tid == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (tid == 0) __hip_atomic_store(cur + A.g, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double x = 0;
        if (tid < A.G) x = exchange_poll(A, cur + tid);
        double tot = 0;
        for (int m = 0; m < A.G; ++m) tot += __shfl(x, m);
        if (tid == 0) {
            if (FENCED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            red[18] = tot;
            __hip_atomic_store(A.fslot + ((s + 2u) % 3u) * A.G + A.g, __longlong_as_double(static_cast<long long>(kFastSentinel)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (FENCED) {
    } else if (tid < 192) {
        const int idx = tid - 64;
        if (A.g > 0 && idx < (Hc << logC))
            PP[(idx & (C - 1)) * RS + (idx >> logC) - Hc] = exchange_poll(A, A.xchg + (static_cast<size_t>(s % 3u) * A.G + A.g - 1) * kXchg + idx);
    } else if (tid == 192) {
        if (A.g < A.G - 1) PP[C * RS] = exchange_poll(A, A.xchg + (static_cast<size_t>(s % 3u) * A.G + A.g + 1) * kXchg + kXchg - 1);
    }
    __syncthreads();
    if (tid < kXchg)
        __hip_atomic_store(A.xchg + (static_cast<size_t>((s + 2u) % 3u) * A.G + A.g) * kXchg + tid,
                           __longlong_as_double(static_cast<long long>(kFastSentinel)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return red[18];
}

__device__ __forceinline__ double gs_point(double s, double xm, double xp, double dh)
{
    // PoissonSolver.cpp:56-57; d * t * 0.5 == (0.5 d) * t exactly
    return 0.5 * (s + xm + xp - dh * (xp - xm));
}

// The same update with the recurrence carried as y = 2x: 0.5*y is exact, so fma(y, 0.5, s) = fl(s + x) and
// fma(y, -0.5, xp) = fl(xp - x) are the reference's roundings, and the dependent chain per node is three fp64 operations
// instead of four (the sweeps that run out of LDS are bound by exactly that latency: ~16 cycles per dependent operation).
// Returns 2 * gs_point(s, 0.5*y, xp, dh) before its (exact) halving.
__device__ __forceinline__ double gs_point2(double s, double y, double xp, double dh)
{
    const double t1 = __builtin_fma(y, 0.5, s);
    const double t2 = __builtin_fma(y, -0.5, xp);
    return (t1 + xp) - dh * t2;
}

// Chunked sweep of one level (S, pin, pout: storage element 0 of the level's source / current / other copy, in global
// memory or in LDS); LOGT = log2(lanes) as a compile-time constant (row strides become instruction offsets) or -1 for any
// value.  Returns this thread's share of sum dPhi^2.
template <int LOGT>
__device__ __forceinline__ double gs_chunked(const Lvl& L, const double* __restrict__ S, const double* __restrict__ pin,
                                             double* __restrict__ pout, const int tid, const double dh)
{
    double err2 = 0;
    {
        // chunked sweep: lane t owns nodes [t*C, t*C + C), all lanes step through their chunk in lockstep.  With
        // m = step number (m - W < 0: warm-up inside the previous lanes' chunks) the node of lane t is
        //     i = t*C + (m - W) = (t + tu)*C + ku,   ku = (m - W) & (C-1),  tu = (m - W) >> logC   (both wave-uniform)
        // so its storage index is ku*T + tu + t: a uniform base plus the lane id -- no per-lane address arithmetic.
        // Loads run kPF steps ahead of the recurrence in registers (two buffers), all of them unconditional: lanes
        // whose node index is still < 1 read in-bounds padding (kPad) and skip the update.
        const int logT = LOGT >= 0 ? LOGT : L.logT;
        const int T = 1 << logT, C = 1 << L.logC, logC = L.logC;
        if (tid < T) {
            const int lo = tid << logC;
            const int one_minus_lo = 1 - lo;
            // first node of this lane's run and its left neighbour (old value; the exact boundary value for node 0)
            const int i0 = (lo - kWarm) > 1 ? (lo - kWarm) : 1;
            double xm, old;
            {
                const int a0 = i0 - 1, a1 = i0;
                xm = pin[((a0 & (C - 1)) << logT) + (a0 >> logC)];
                old = pin[((a1 & (C - 1)) << logT) + (a1 >> logC)];
            }
            // right neighbour of the last owned node: node (t+1)*C, which is node n-1 (stored at C*T) for the last lane
            const double xp_end = pin[(tid == T - 1) ? (C << logT) : (tid + 1)];
            const int Cm1 = C - 1;
            const unsigned tu = static_cast<unsigned>(tid);
            // Every access is "wave-uniform row pointer + lane id" (scalar base register + 32-bit lane offset in the
            // load/store instruction): no per-lane address arithmetic on the VALU.
            auto row = [&](const double* base, int r) -> const double* {
                return base + (static_cast<long>((r & Cm1)) << logT) + (r >> logC);
            };
            // loads of step r (= m - W): S at node i(r), Phi_old at node i(r) + 1 = i(r+1); indices clamped to r <= C-1
            auto load8 = [&](double (&X)[kPF], double (&SV)[kPF], int rbase) {
                // a batch that neither wraps into the neighbouring lane's column nor runs past the chunk is 9 consecutive
                // rows: one base pointer per array, the row strides are instruction offsets
                if (LOGT >= 0 && C >= kPF && rbase + kPF <= Cm1 && ((rbase + kPF) & Cm1) != 0) {
                    const double* __restrict__ bs = row(S, rbase);
                    const double* __restrict__ bp = row(pin, rbase) + T;
#pragma unroll
                    for (int q = 0; q < kPF; ++q) {
                        SV[q] = bs[(q << logT) + tu];
                        X[q] = bp[(q << logT) + tu];
                    }
                    return;
                }
#pragma unroll
                for (int q = 0; q < kPF; ++q) {
                    int r0 = rbase + q;
                    r0 = r0 < Cm1 ? r0 : Cm1;
                    SV[q] = row(S, r0)[tu];
                    X[q] = row(pin, r0 + 1)[tu];
                }
            };
            // Only the first lanes of the workgroup ever see a node index < 1 (lo + r < 1: during the warm-up, and lane 0
            // at r = 0).  Their wave takes the `careful` variants, which keep xm with a select instead of an exec-mask
            // branch per node (a mask that depends on a compare costs a VALU -> SGPR -> EXEC round trip per step).
            const bool careful = __builtin_amdgcn_readfirstlane(lo) <= kWarm;
            // warm-up steps (r < 0): recurrence only
            auto warm8 = [&](auto CAREFUL, const double (&X)[kPF], const double (&SV)[kPF], int rbase) {
#pragma unroll
                for (int q = 0; q < kPF; ++q) {
                    const double x = gs_point(SV[q], xm, X[q], dh);
                    if (decltype(CAREFUL)::value) xm = (rbase + q >= one_minus_lo) ? x : xm;   // node index lo + r >= 1
                    else xm = x;
                }
            };
            // owned steps (0 <= r < C): recurrence, error norm, store.  FIRST: the batch that holds r = 0 (node 0 of lane 0
            // is a boundary value, not an unknown); LAST: the batch whose last node has xp_end as right neighbour
            auto main8 = [&](auto FIRST, auto LAST, const double (&X)[kPF], const double (&SV)[kPF], int rbase) {
                double* __restrict__ bo = pout + (static_cast<long>(rbase) << logT);   // rows rbase .. rbase+7 of the own chunk
#pragma unroll
                for (int q = 0; q < kPF; ++q) {
                    const int r0 = rbase + q;
                    const double xp = (decltype(LAST)::value && q == kPF - 1) ? xp_end : X[q];
                    const double x = gs_point(SV[q], xm, xp, dh);
                    double dif = old - x;
                    if (decltype(FIRST)::value && q == 0) {
                        const bool live = r0 >= one_minus_lo;
                        dif = live ? dif : 0.0;
                        xm = live ? x : xm;
                    } else {
                        xm = x;
                    }
                    err2 += dif * dif;
                    if (LOGT >= 0) bo[(q << logT) + tu] = x;             // node 0 (lane 0, r = 0) is rewritten below
                    else const_cast<double*>(row(pout, r0))[tu] = x;     // tu == 0 inside the own chunk
                    old = xp;
                }
            };
            using std::true_type;
            using std::false_type;
            double ax[kPF], as[kPF], bx[kPF], bs[kPF];
            static_assert(kWarm % (2 * kPF) == 0, "warm-up must be a whole number of A/B rounds");
            load8(ax, as, -kWarm);
            if (careful) {
                for (int r = -kWarm; r < 0; r += 2 * kPF) {
                    load8(bx, bs, r + kPF);
                    warm8(true_type{}, ax, as, r);
                    load8(ax, as, r + 2 * kPF);                          // the last one already fetches r = 0 .. kPF-1
                    warm8(true_type{}, bx, bs, r + kPF);
                }
            } else {
                for (int r = -kWarm; r < 0; r += 2 * kPF) {
                    load8(bx, bs, r + kPF);
                    warm8(false_type{}, ax, as, r);
                    load8(ax, as, r + 2 * kPF);
                    warm8(false_type{}, bx, bs, r + kPF);
                }
            }
            // Phi_old at the first owned node (the last warm-up step's right neighbour); lane 0 keeps the value loaded above
            if (lo >= 1) old = bx[kPF - 1];
            if (C >= 2 * kPF) {
                for (int r = 0; r < C; r += 2 * kPF) {
                    load8(bx, bs, r + kPF);
                    if (r == 0) main8(true_type{}, false_type{}, ax, as, r);
                    else        main8(false_type{}, false_type{}, ax, as, r);
                    load8(ax, as, r + 2 * kPF);
                    if (r + 2 * kPF >= C) main8(false_type{}, true_type{}, bx, bs, r + kPF);
                    else                  main8(false_type{}, false_type{}, bx, bs, r + kPF);
                }
            } else {
                // C = 1, 2, 4 or 8 owned nodes: ax/as hold r = 0 .. min(C, kPF) - 1 (clamped beyond)
                for (int r0 = 0; r0 < C; ++r0) {
                    if (r0 >= one_minus_lo) {
                        const int r1 = r0 + 1;
                        const double sv = S[(r0 << logT) + tid];
                        const double xp = (r0 == Cm1) ? xp_end : pin[(r1 << logT) + tid];
                        const double x = gs_point(sv, xm, xp, dh);
                        const double dif = old - x;
                        err2 += dif * dif;
                        pout[(r0 << logT) + tid] = x;
                        xm = x;
                        old = xp;
                    }
                }
            }
        }
        if (tid == 0) {
            pout[0] = pin[0];                                   // node 0
            pout[C << logT] = pin[C << logT];                   // node n-1
        }
    }
    return err2;
}

// Chunked sweep, IN PLACE, of (a workgroup's part of) a level staged in LDS: NT lanes, C = 2^LOGC nodes per lane; node
// t*C + k of the part lives at k*RS + t relative to SSbase / PPbase, lanes t < 0 (the 96 nodes in front of the part: the
// previous lanes' columns for RS == 256, halo columns otherwise) included; Phi of the node behind the part at C*RS.  Same
// arithmetic as gs_chunked.  A lone wave on a SIMD issues one instruction of any kind per ~4.5 cycles, so the step is
// priced in instructions: with C a compile-time constant every LDS access of a 16-step block is "per-lane base register +
// immediate offset" (the uniform part of the index, (q & (C-1))*RS + (q >> LOGC), is known at compile time; the base
// advances by 16/C per block) -- 2 reads + 6 flops per warm-up step, nothing else.  Everything a lane reads from other
// lanes' nodes (warm-up, start values, right neighbour of its last node) is read before the barrier in the middle, the
// owned nodes are overwritten after it.  lo_g = index of the lane's first node within the level.
// NT = lanes that sweep (256: the workgroup; 64: its first wave alone -- the barrier is then a wave-level fence).
template <int LOGC, int RS, int NT = kThreads>
__device__ __forceinline__ double gs_lds(const double* __restrict__ SSbase, double* __restrict__ PPbase, const int tid,
                                         const int lo_g, const double dh)
{
    constexpr int C = 1 << LOGC, Cm1 = C - 1, kH = 8;
    auto uoff = [](int q) constexpr -> int { return (q & Cm1) * RS + (q >> LOGC); };    // q >= 0
    // explicit LDS pointers, each pinned in its own register: otherwise the compiler rebuilds every address from one
    // base plus a literal (the arrays are > 64 KB apart, beyond the instruction offset) -- one VALU add per access
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) const double lds_cf64;
    lds_cf64* ps = (lds_cf64*)(SSbase) + tid;
    lds_f64* pp = (lds_f64*)(PPbase) + tid;
    asm volatile("" : "+v"(ps), "+v"(pp));
    // The warm-up covers the 95 nodes lo-95 .. lo-1, in blocks of BS = max(16, C) steps; step q of the block with base
    // node lo + Rb (Rb = -96, -96+BS, ..: multiples of C) handles node lo + Rb + q + 1.  It starts from the old value of
    // node lo-96.  Lanes whose run would begin below node 1 (the first 96/C lanes of the level) restart from the exact
    // boundary value at node 1: with the blocks shifted by one node against the chunk rows that can only happen at steps
    // with (q % C) == 0 -- one select per C steps instead of one per step.
    const int neg_lo = -lo_g;
    const double Y0 = 2.0 * PPbase[0];          // node 0 of the level (meaningful where a restart can happen at all)
    double y = 2.0 * pp[-(kWarm >> LOGC)];      // the recurrence is carried as y = 2x (gs_point2)
    const double xp_end = (tid == NT - 1) ? PPbase[C * RS] : pp[1];
    const double node0 = pp[0];
    lds_cf64* bs = ps - (kWarm >> LOGC);
    lds_cf64* bp = pp - (kWarm >> LOGC);
    double ax[kH], as[kH], bx[kH], bv[kH];
    constexpr int BS = C > 2 * kH ? C : 2 * kH;
    static_assert(kWarm % BS == 0, "warm-up must be a whole number of blocks");
    static_assert(BS <= 4 * kH, "chunks of more than 32 nodes are not staged");
    using std::integral_constant;
    auto load = [&](double (&X)[kH], double (&SV)[kH], auto HH) {
        constexpr int h = decltype(HH)::value;
#pragma unroll
        for (int q = 0; q < kH; ++q) { SV[q] = bs[uoff(h * kH + q + 1)]; X[q] = bp[uoff(h * kH + q + 2)]; }
    };
    auto load_own = [&](double (&X)[kH], double (&SV)[kH]) {     // rows 0 .. 7 of the own chunk (bs == ps, bp == pp by then)
#pragma unroll
        for (int q = 0; q < kH; ++q) { SV[q] = bs[uoff(q)]; X[q] = bp[uoff(q + 1)]; }
    };
    auto warm = [&](const double (&X)[kH], const double (&SV)[kH], int Rb, auto HH, auto NSTEPS) {
        constexpr int h = decltype(HH)::value;
#pragma unroll
        for (int q = 0; q < decltype(NSTEPS)::value; ++q) {
            if ((h * kH + q) % C == 0) y = (Rb + (h * kH + q) == neg_lo) ? Y0 : y;      // this step is node 1
            y = gs_point2(SV[q], y, X[q], dh);
        }
    };
    auto block = [&](int Rb, auto LAST) {
        // halves 0, 2 in A, 1, 3 in B; the reads stay a whole half block ahead of their use
        constexpr bool last = decltype(LAST)::value;
        auto pair = [&](auto H0) {
            constexpr int h = decltype(H0)::value;
            constexpr bool wraps = (h + 2) * kH >= BS;
            load(bx, bv, integral_constant<int, h + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            warm(ax, as, Rb, integral_constant<int, h>{}, integral_constant<int, kH>{});
            if constexpr (!wraps) {
                load(ax, as, integral_constant<int, h + 2>{});
            } else {
                bs += BS >> LOGC;
                bp += BS >> LOGC;
                asm volatile("" : "+v"(bs), "+v"(bp));
                if constexpr (last) load_own(ax, as);
                else load(ax, as, integral_constant<int, 0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            warm(bx, bv, Rb, integral_constant<int, h + 1>{}, integral_constant<int, (last && wraps) ? kH - 1 : kH>{});
        };
        pair(integral_constant<int, 0>{});
        if constexpr (BS > 2 * kH) pair(integral_constant<int, 2>{});
    };
    load(ax, as, integral_constant<int, 0>{});
    for (int Rb = -kWarm; Rb < -BS; Rb += BS) block(Rb, std::false_type{});
    block(-BS, std::true_type{});
    double old = bx[kH - 2];                    // Phi_old at the first owned node: the right neighbour of the last warm-up step
    // node 1 as a lane's FIRST node (C == 1, lane 1): its restart falls on the step that the warm-up leaves to the owned part
    if constexpr (C == 1) y = (lo_g == 1) ? Y0 : y;
    // all reads of other lanes' old values are done
    if constexpr (NT == kThreads) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    double err2 = 0;
    auto own = [&](const double (&X)[kH], const double (&SV)[kH], int r0) {
#pragma unroll
        for (int q = 0; q < (C < kH ? C : kH); ++q) {
            const int r = r0 + q;
            const double xp = (r == Cm1) ? xp_end : X[q];
            const double yn = gs_point2(SV[q], y, xp, dh);
            const double x = 0.5 * yn;
            double dif = old - x;
            if (r == 0) {
                const bool live = lo_g >= 1;    // node 0 is a boundary value, not an unknown
                dif = live ? dif : 0.0;
                y = live ? yn : Y0;             // the lane of node 0 continues from the boundary value
            } else {
                y = yn;
            }
            err2 += dif * dif;
            pp[r * RS] = x;                     // node 0 is restored below
            old = xp;
        }
    };
    if constexpr (C <= kH) {
        own(ax, as, 0);
    } else {
        // rows r .. r+7 in A (fetched by the last warm-up block for r = 0), r+8 .. r+15 in B
#pragma unroll
        for (int r = 0; r < C; r += 2 * kH) {
#pragma unroll
            for (int q = 0; q < kH; ++q) { bv[q] = ps[(r + kH + q) * RS]; bx[q] = pp[((r + kH + q + 1) & Cm1) * RS]; }
            __builtin_amdgcn_sched_barrier(0);
            own(ax, as, r);
            if (r + 2 * kH < C) {
#pragma unroll
                for (int q = 0; q < kH; ++q) { as[q] = ps[(r + 2 * kH + q) * RS]; ax[q] = pp[(r + 2 * kH + q + 1) * RS]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            own(bx, bv, r + kH);
        }
    }
    if (lo_g == 0) pp[0] = node0;
    return err2;
}

// ---- fused visit -------------------------------------------------------------------------------------------------
// The three sweeps of IterateGaussSeidel(level, errorMin, 3) (PoissonSolver.cpp:66-77) in ONE in-place pass over a part staged in
// LDS.  Sweep k+1 at node j needs sweep k at node j+1, so the three sweeps run as a software pipeline: at the step with stage-1
// node tau the lane computes
//     x1[tau]   = gs(S[tau],   x1[tau-1], x0[tau+1])
//     x2[tau-1] = gs(S[tau-1], x2[tau-2], x1[tau])
//     x3[tau-2] = gs(S[tau-2], x3[tau-3], x2[tau-1])
// with the arithmetic of gs_point2, i.e. exactly what three sequential sweeps compute.  All three stages start kWarm3 nodes in front
// of the lane's chunk from old values: a start-value error e decays as e b^j in stage 1, (1 + j/4) e b^j in stage 2 (which also
// feeds on stage 1's error) and (1 + j/4 + j^2/32) e b^j in stage 3, b = (1 + d/2)/2; at j = 112 the third factor is 2^8.7 and
// b^(112-95) <= 2^-16: a wider margin than the single sweep's 95 nodes.  The lane runs two nodes past its chunk (x1, x2 of the right
// neighbour's first nodes, from old values read before the barrier).  One LDS read pair per step instead of three, no barrier and --
// for a part shared by a group -- no exchange between the sweeps.  KST = which sweep's values are stored (3; 1 or 2 when the
// reference would have stopped early: the caller restores the old values and repeats the pass).
// Layout as gs_lds: node lo + r of lane tid at (r & (C-1)) * RS + (r >> LOGC) relative to the lane's column; the three nodes
// behind the chunk of lane NT-1 are rows 0..2 of column NT (a right halo column) unless `last_lane`: then the chunk ends at the
// level's last node, whose value is xN.  Returns the lane's shares of the three sums of dPhi^2.
constexpr int kWarm3 = 112;
// gs_point2 with the right neighbour given as yp = 2 xp as well (stages 2 and 3 of the fused pass, which take it from the stage in
// front of them): halving commutes with every rounding involved -- fl(xp - x) = fl(yp - y) / 2, fl(t1 + xp) = fma(yp, 0.5, t1),
// dh fl(xp - x) = (dh / 2) fl(yp - y) -- so the result is gs_point2(s, y, yp / 2, dh) bit for bit, one instruction less.
__device__ __forceinline__ double gs_point2y(double s, double y, double yp, double dh2)
{
    const double t1 = __builtin_fma(y, 0.5, s);
    const double u = yp - y;
    const double a = __builtin_fma(yp, 0.5, t1);
    return a - dh2 * u;
}
template <int LOGC, int RS, int NT, int KST, bool CAREFUL>
__device__ __forceinline__ void gs_lds3(const double* __restrict__ SSbase, double* __restrict__ PPbase, const int tid, const int lo_g,
                                        const bool last_lane, const double xN, const double dh, double& e1, double& e2, double& e3)
{
    constexpr int C = 1 << LOGC, Cm1 = C - 1, kH = 8, W = kWarm3;
    static_assert(C >= 4 && C <= 32 && W % (2 * kH) == 0, "fused pass: 4..32 nodes per lane");
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) const double lds_cf64;
    const bool active = tid < NT;
    lds_cf64* ps = (lds_cf64*)(SSbase) + tid;
    lds_f64* pp = (lds_f64*)(PPbase) + tid;
    asm volatile("" : "+v"(ps), "+v"(pp));
    auto off = [](int r) constexpr -> int { return (r & Cm1) * RS + (r >> LOGC); };       // any r (arithmetic shift = floor)
    const int neg_lo = -lo_g;
    const double dh2 = 0.5 * dh;
    const double Y0 = 2.0 * PPbase[0];          // node 0 of the level (meaningful where a restart can happen at all)
    double a1 = 0, a2 = 0, a3 = 0;              // a2, a3 are accumulated on doubled differences: four times the sums (exactly)
    double y1 = 0, y2 = 0, y3 = 0, sA = 0, sB = 0, o0 = 0, y1p = 0, y2p = 0;
    double xr0 = 0, xr1 = 0, xr2 = 0, sr0 = 0, sr1 = 0;
    double ax[kH], as[kH], bx[kH], bv[kH];
    // loads of the batch of 8 steps whose stage-1 nodes are lo + r0 .. lo + r0 + 7 (r0 a multiple of 8): S at the node, old Phi at
    // the node behind it.  r0 is wave-uniform: one base per array and batch, the rows are instruction offsets.
    auto load = [&](double (&X)[kH], double (&SV)[kH], const int r0) {
        const int col = r0 >> LOGC;
        if constexpr (C <= kH) {
            lds_cf64* bs = ps + col;
            lds_cf64* bp = (lds_cf64*)pp + col;
#pragma unroll
            for (int q = 0; q < kH; ++q) { SV[q] = bs[off(q)]; X[q] = bp[off(q + 1)]; }
        } else {
            const int k0 = r0 & Cm1;
            lds_cf64* bs = ps + (k0 * RS + col);
            lds_cf64* bp = (lds_cf64*)pp + (k0 * RS + col);
#pragma unroll
            for (int q = 0; q < kH; ++q) SV[q] = bs[q * RS];
#pragma unroll
            for (int q = 0; q < kH - 1; ++q) X[q] = bp[(q + 1) * RS];
            lds_cf64* bw = (k0 + kH == C) ? (lds_cf64*)pp + (col + 1) : bp + kH * RS;     // the next column's first row
            X[kH - 1] = *bw;
        }
    };
    // warm-up batch: recurrences only (15 instructions per step)
    auto warm = [&](const double (&X)[kH], const double (&SV)[kH], const int r0) {
#pragma unroll
        for (int q = 0; q < kH; ++q) {
            double t = gs_point2(SV[q], y1, X[q], dh);
            if (CAREFUL && ((q & (C < kH ? Cm1 : kH - 1)) == 0)) t = (r0 + q == neg_lo) ? Y0 : t;           // stage 1 stands on node 0
            double u = gs_point2y(sA, y2, t, dh2);
            if (CAREFUL && (((q - 1) & (C < kH ? Cm1 : kH - 1)) == 0)) u = (r0 + q - 1 == neg_lo) ? Y0 : u;
            double v = gs_point2y(sB, y3, u, dh2);
            if (CAREFUL && (((q - 2) & (C < kH ? Cm1 : kH - 1)) == 0)) v = (r0 + q - 2 == neg_lo) ? Y0 : v;
            y1 = t; y2 = u; y3 = v;
            sB = sA; sA = SV[q];
        }
        o0 = X[kH - 1];
    };
    // own batch: recurrences, error norms, the store of sweep KST's values
    auto own = [&](auto FIRST, const double (&X)[kH], const double (&SV)[kH], const int r0) {
        constexpr bool first = decltype(FIRST)::value;
        constexpr int nq = C < kH ? C : kH;
        lds_f64* po = pp + r0 * RS;
#pragma unroll
        for (int q = 0; q < nq; ++q) {
            const double xn = (q == nq - 1) ? ((r0 + nq == C) ? xr0 : X[q]) : X[q];
            double t = gs_point2(SV[q], y1, xn, dh);
            if (CAREFUL && first && q == 0) t = (lo_g == 0) ? Y0 : t;
            { const double d = __builtin_fma(t, -0.5, o0); a1 = __builtin_fma(d, d, a1); }
            double u = gs_point2y(sA, y2, t, dh2);
            if (CAREFUL && first && q == 1) u = (lo_g == 0) ? Y0 : u;
            if (!(first && q == 0)) { const double d = y1p - u; a2 = __builtin_fma(d, d, a2); }
            double v = gs_point2y(sB, y3, u, dh2);
            if (CAREFUL && first && q == 2) v = (lo_g == 0) ? Y0 : v;
            if (!(first && q < 2)) { const double d = y2p - v; a3 = __builtin_fma(d, d, a3); }
            if (KST == 1) po[q * RS] = 0.5 * t;
            if (KST == 2 && !(first && q == 0)) po[(q - 1) * RS] = 0.5 * u;
            if (KST == 3 && !(first && q < 2)) po[(q - 2) * RS] = 0.5 * v;
            y1 = t; y2 = u; y3 = v;
            sB = sA; sA = SV[q];
            o0 = xn; y1p = t; y2p = u;
        }
    };
    if (active) {
        y1 = 2.0 * pp[off(-W - 1)]; y2 = 2.0 * pp[off(-W - 2)]; y3 = 2.0 * pp[off(-W - 3)];
        sA = ps[off(-W - 1)]; sB = ps[off(-W - 2)];
        load(ax, as, -W);
        for (int r0 = -W; r0 < 0; r0 += 2 * kH) {
            load(bx, bv, r0 + kH);
            __builtin_amdgcn_sched_barrier(0);
            warm(ax, as, r0);
            load(ax, as, r0 + 2 * kH);                  // the last one fetches the first own batch (still old values)
            __builtin_amdgcn_sched_barrier(0);
            warm(bx, bv, r0 + kH);
        }
        y1p = y1; y2p = y2;
        // old values behind the chunk: the right neighbour overwrites them after the barrier
        xr0 = last_lane ? xN : pp[off(C)];
        xr1 = pp[off(C + 1)]; xr2 = pp[off(C + 2)];
        sr0 = ps[off(C)]; sr1 = ps[off(C + 1)];
    }
    // all reads of other lanes' old values are done
    if constexpr (NT == 64) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
    if (active) {
        if constexpr (C <= kH) {
            own(std::true_type{}, ax, as, 0);
        } else {
            for (int r0 = 0; r0 < C; r0 += 2 * kH) {
                load(bx, bv, r0 + kH);
                __builtin_amdgcn_sched_barrier(0);
                if (r0 == 0) own(std::true_type{}, ax, as, 0); else own(std::false_type{}, ax, as, r0);
                if (r0 + 2 * kH < C) load(ax, as, r0 + 2 * kH);
                __builtin_amdgcn_sched_barrier(0);
                own(std::false_type{}, bx, bv, r0 + kH);
            }
        }
        if (KST >= 2) {
            // two steps behind the chunk: x1 (x2) of the right neighbour's first node(s) from the old values read above; a chunk that
            // ends at the level's last node finds the boundary value there
            const double Y_N = 2.0 * xN;
            double t = gs_point2(sr0, y1, xr1, dh);
            t = last_lane ? Y_N : t;
            double u = gs_point2y(sA, y2, t, dh2);
            { const double d = y1p - u; a2 = __builtin_fma(d, d, a2); }
            if (KST == 2) pp[Cm1 * RS] = 0.5 * u;
            if (KST == 3) {
                double v = gs_point2y(sB, y3, u, dh2);
                { const double d = y2p - v; a3 = __builtin_fma(d, d, a3); }
                pp[(C - 2) * RS] = 0.5 * v;
                y1 = t; y2 = u; y3 = v; sB = sA; sA = sr0; y2p = u;
                t = gs_point2(sr1, y1, xr2, dh);
                u = gs_point2y(sA, y2, t, dh2);
                u = last_lane ? Y_N : u;
                v = gs_point2y(sB, y3, u, dh2);
                { const double d = y2p - v; a3 = __builtin_fma(d, d, a3); }
                pp[Cm1 * RS] = 0.5 * v;
            }
        }
    }
    e1 = a1; e2 = 0.25 * a2; e3 = 0.25 * a3;
}

// One sweep of an LDS-resident level by ONE thread, in the reference's order.  The loads of a batch (right neighbours,
// sources) are independent of the recurrence: they are issued together ahead of it, so that the chain is not one LDS
// round trip per node.  Returns sum dPhi^2.
__device__ __forceinline__ double seq_sweep(const double* __restrict__ S, const double* __restrict__ pin, double* __restrict__ pout,
                                            const int n, const double dh)
{
    double err2 = 0;
    const double x0 = pin[0];
    pout[0] = x0;
    double y = 2.0 * x0;                // the recurrence is carried as y = 2x (gs_point2: a shorter dependent chain)
    const int limit = n - 1;
    double old = pin[1];
    constexpr int kB = 8;
    int i = 1;
    for (; i + kB <= limit; i += kB) {
        double xp[kB], sv[kB], xo[kB];
#pragma unroll
        for (int q = 0; q < kB; ++q) { xp[q] = pin[i + q + 1]; sv[q] = S[i + q]; }
#pragma unroll
        for (int q = 0; q < kB; ++q) {
            y = gs_point2(sv[q], y, xp[q], dh);
            const double x = 0.5 * y;
            const double dif = old - x;
            err2 += dif * dif;
            xo[q] = x;
            old = xp[q];
        }
#pragma unroll
        for (int q = 0; q < kB; ++q) pout[i + q] = xo[q];
    }
    for (; i < limit; ++i) {
        const double xp = pin[i + 1];
        y = gs_point2(S[i], y, xp, dh);
        const double x = 0.5 * y;
        const double dif = old - x;
        err2 += dif * dif;
        pout[i] = x;
        old = xp;
    }
    pout[limit] = pin[limit];
    return err2;
}

// one lexicographic Gauss-Seidel sweep of level l: PoissonSolver::GaussSeidel (PoissonSolver.cpp:40-64).
// returns ||dPhi||_2 (same value in every thread)
__device__ __forceinline__ double gauss_seidel(const MgDesc& D, Atom& A, int l, double* red)
{
    const Lvl L = D.lv[l];
    const double dh = L.d * 0.5;
    double err2 = 0;
    const bool coop = l < D.kcoop;         // swept by the whole group: lane ids run over all members
    const int tid = coop ? A.lane() : static_cast<int>(threadIdx.x);
    if (L.seq) {
        if (tid == 0) err2 = seq_sweep(A.src_of(L), A.cur_phi(l, L), A.other_phi(l, L), L.n, dh);
    } else {
        const double* S = A.src + L.off;
        const double* pin = (((A.cur >> l) & 1u) ? A.phi1 : A.phi0) + L.off;
        double* pout = (((A.cur >> l) & 1u) ? A.phi0 : A.phi1) + L.off;
        err2 = gs_chunked<-1>(L, S, pin, pout, tid, dh);
    }
    A.cur ^= (1u << l);
    // also orders the writes of this sweep before the next phase (group barrier inside for cooperative levels)
    const double tot = coop ? group_sum(A, err2, red) : block_sum(err2, red);
    return sqrt(tot);
}

// Three consecutive Gauss-Seidel sweeps of a chunked level in ONE pass over memory (the smoother is bound by the
// CU's vector-memory pipe: fusing cuts its traffic to a third).  Sweep k+1 at node j needs sweep k at node j+1, so the
// three sweeps run as a software pipeline: at step tau the lane computes
//     x1[tau]   = gs(S[tau],   x1[tau-1], x0[tau+1])
//     x2[tau-1] = gs(S[tau-1], x2[tau-2], x1[tau])
//     x3[tau-2] = gs(S[tau-2], x3[tau-3], x2[tau-1])
// i.e. exactly the arithmetic of three sequential sweeps.  Each lane starts 3W nodes before its chunk (x1 is exact
// after W nodes, x2 after 2W, x3 after 3W: start-value errors contract by (1+d/2)/2 per node) and runs 2 nodes past it
// (recomputing what its right neighbour computes).  The input copy stays untouched (the result goes to the other
// copy), so when the reference would have stopped after the first or second sweep (err < errorMin) the caller redoes
// the visit with single sweeps.  Returns the three error norms of PoissonSolver::GaussSeidel.
__device__ __forceinline__ void gs_fused3(const MgDesc& D, Atom& A, int l, double* red, double& e1, double& e2, double& e3)
{
    const Lvl L = D.lv[l];
    const double dh = L.d * 0.5;
    const int tid = threadIdx.x;
    const int T = 1 << L.logT, C = 1 << L.logC, logC = L.logC, logT = L.logT, Cm1 = C - 1;
    const int n = L.n;
    const double* __restrict__ S = A.src + L.off;
    const double* __restrict__ pin = (((A.cur >> l) & 1u) ? A.phi1 : A.phi0) + L.off;
    double* __restrict__ pout = (((A.cur >> l) & 1u) ? A.phi0 : A.phi1) + L.off;
    double a1 = 0, a2 = 0, a3 = 0;
    if (tid < T) {
        const int lo = tid << logC;
        const int nsteps = (C + 3 * kWarm + 2 + kPF - 1) / kPF * kPF;      // whole prefetch batches; extra steps = extra warm-up
        const int rstart = (C + 1) - (nsteps - 1);                           // stage-1 node offset of the first step (<= -3W)
        const double b0 = pin[0], bN = pin[C << logT];                        // fixed values at node 0 and node n-1
        auto old_at = [&](int j) -> double {
            if (j <= 0) return b0;
            if (j >= n - 1) return bN;
            return pin[((j & Cm1) << logT) + (j >> logC)];
        };
        auto src_at = [&](int j) -> double {
            if (j <= 0 || j >= n - 1) return 0.0;
            return S[((j & Cm1) << logT) + (j >> logC)];
        };
        // pipeline state: left neighbours = values of the previous step; old values and sources of the last three nodes
        const int t0 = lo + rstart;                         // stage-1 node of the first step
        double x1m = old_at(t0 - 1), x2m = old_at(t0 - 2), x3m = old_at(t0 - 3);
        double o0 = old_at(t0), o1 = old_at(t0 - 1), o2 = old_at(t0 - 2);
        double sA = src_at(t0 - 1), sB = src_at(t0 - 2);
        // Fixed nodes.  Left: a stage is inactive (its value stays b0) while its node index is < 1, i.e. while
        // r - (k-1) < rs with rs = 1 - lo; only lanes whose warm-up reaches below node 1 are concerned and only in the
        // first batches of their wave.  Right: only the last lane, in the last two steps (nodes n-1, n).  Batches that
        // touch neither run the unguarded step.
        const int rs = 1 - lo;
        if (t0 - 1 < 1) { x1m = b0; }                       // stage values at nodes <= 0 are the boundary value
        if (t0 - 2 < 1) { x2m = b0; }
        if (t0 - 3 < 1) { x3m = b0; }
        const int wave_first_lo = (tid & ~63) << logC;
        const int rs_wave = 1 - wave_first_lo;             // largest rs in this wave
        const bool wave_has_last = ((tid | 63) >= T - 1) && ((tid & ~63) <= T - 1);

        // loads of the step with stage-1 offset r: S at node lo+r, old Phi at node lo+r+1 (rows are wave-uniform)
        auto load8 = [&](double (&X)[kPF], double (&SV)[kPF], int rbase) {
#pragma unroll
            for (int q = 0; q < kPF; ++q) {
                int r0 = rbase + q;
                r0 = r0 < C + 1 ? r0 : C + 1;
                const int r1 = r0 + 1;
                SV[q] = S[(((r0 & Cm1) << logT) + (r0 >> logC)) + tid];
                X[q] = pin[(((r1 & Cm1) << logT) + (r1 >> logC)) + tid];
            }
        };
        // MODE 0: no fixed node in this batch; 1: left end only (stage activation by predicate); 2: anything (selects)
        auto step8 = [&](const double (&X)[kPF], const double (&SV)[kPF], int rbase, const int mode) {
#pragma unroll
            for (int q = 0; q < kPF; ++q) {
                const int r = rbase + q;               // stage-1 node offset (wave-uniform)
                double xn = X[q];                      // old value at node tau + 1
                const double s0 = SV[q];
                double x1, x2, x3;
                bool u1 = true, u2 = true, u3 = true;
                if (mode == 0) {
                    x1 = gs_point(s0, x1m, xn, dh);
                    x2 = gs_point(sA, x2m, x1, dh);
                    x3 = gs_point(sB, x3m, x2, dh);
                } else if (mode == 1) {
                    u1 = (r >= rs); u2 = (r - 1 >= rs); u3 = (r - 2 >= rs);
                    x1 = x1m; x2 = x2m; x3 = x3m;      // == b0 while the stage has not started
                    if (u1) x1 = gs_point(s0, x1m, xn, dh);
                    if (u2) x2 = gs_point(sA, x2m, x1, dh);
                    if (u3) x3 = gs_point(sB, x3m, x2, dh);
                } else {
                    const int tau = lo + r;
                    if (tau + 1 >= n - 1) xn = bN;
                    if (tau + 1 <= 0) xn = b0;
                    u1 = (tau >= 1) && (tau <= n - 2);
                    u2 = (tau - 1 >= 1) && (tau - 1 <= n - 2);
                    u3 = (tau - 2 >= 1) && (tau - 2 <= n - 2);
                    x1 = u1 ? gs_point(s0, x1m, xn, dh) : ((tau <= 0) ? b0 : ((tau >= n - 1) ? bN : o0));
                    x2 = u2 ? gs_point(sA, x2m, x1, dh) : ((tau - 1 <= 0) ? b0 : ((tau - 1 >= n - 1) ? bN : o1));
                    x3 = u3 ? gs_point(sB, x3m, x2, dh) : ((tau - 2 <= 0) ? b0 : ((tau - 2 >= n - 1) ? bN : o2));
                }
                // error norms and the store, own nodes only (wave-uniform ranges); x1m / x2m still hold the stage values
                // of the previous step, which are x1[tau-1] and x2[tau-2]
                if (r >= 0 && r <= Cm1) { if (u1) { const double dd = o0 - x1; a1 = __builtin_fma(dd, dd, a1); } }
                if (r >= 1 && r <= C) { if (u2) { const double dd = x1m - x2; a2 = __builtin_fma(dd, dd, a2); } }
                if (r >= 2 && r <= C + 1) {
                    if (u3) { const double dd = x2m - x3; a3 = __builtin_fma(dd, dd, a3); pout[(((r - 2) & Cm1) << logT) + tid] = x3; }
                }
                x1m = x1; x2m = x2; x3m = x3;
                sB = sA; sA = s0;
                o2 = o1; o1 = o0; o0 = xn;
            }
        };
        auto mode_of = [&](int rbase) -> int {
            const bool left = (rbase - 2 < rs_wave);                          // some stage of some lane below node 1
            const bool right = wave_has_last && (rbase + kPF - 1 >= Cm1);      // last lane at / beyond node n-2
            return right ? 2 : (left ? 1 : 0);
        };
        double ax[kPF], as[kPF], bx[kPF], bs[kPF];
        load8(ax, as, rstart);
        for (int r = rstart; r <= C + 1; r += 2 * kPF) {
            load8(bx, bs, r + kPF);
            {
                const int m = mode_of(r);
                if (m == 0) step8(ax, as, r, 0); else if (m == 1) step8(ax, as, r, 1); else step8(ax, as, r, 2);
            }
            if (r + kPF > C + 1) break;
            load8(ax, as, r + 2 * kPF);
            {
                const int m = mode_of(r + kPF);
                if (m == 0) step8(bx, bs, r + kPF, 0); else if (m == 1) step8(bx, bs, r + kPF, 1); else step8(bx, bs, r + kPF, 2);
            }
        }
    }
    if (tid == 0) {
        pout[0] = pin[0];                                   // node 0
        pout[(1 << L.logC) << L.logT] = pin[(1 << L.logC) << L.logT];   // node n-1
    }
    // one block reduction for the three norms (also orders this pass's writes before the next phase)
    for (int off = 32; off > 0; off >>= 1) { a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); a3 += __shfl_xor(a3, off); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a1; red[4 + (threadIdx.x >> 6)] = a2; red[8 + (threadIdx.x >> 6)] = a3; }
    __syncthreads();
    e1 = sqrt((red[0] + red[1]) + (red[2] + red[3]));
    e2 = sqrt((red[4] + red[5]) + (red[6] + red[7]));
    e3 = sqrt((red[8] + red[9]) + (red[10] + red[11]));
}

// Copy C rows of 256 lanes between global memory and LDS (row strides in doubles).  The loads of up to 8 rows are issued
// together: a row-by-row loop pays one memory round trip per row.
template <typename Dst, typename Src>
__device__ __forceinline__ void copy_rows(Dst* dst, int dstride, const Src* src, int sstride, int C)
{
    const int tid = threadIdx.x;
    int k = 0;
    for (; k + 8 <= C; k += 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[(k + q) * sstride + tid];
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[(k + q) * dstride + tid] = v[q];
    }
    for (; k < C; ++k) dst[k * dstride + tid] = src[k * sstride + tid];
}

// Two arrays at once, 16 rows of each in flight: the copy into the staging memory is a chain of round trips to L2 / the
// Infinity Cache (the level was last written by this or by another compute unit), not a bandwidth problem -- 32 loads per
// lane and trip instead of 8 cut a 32-row copy from eight trips to two.
template <typename DA, typename SA>
__device__ __forceinline__ void copy_rows2(DA* dstA, DA* dstB, int dstride, const SA* srcA, const SA* srcB, int sstride, int C)
{
    const int tid = threadIdx.x;
    int k = 0;
    for (; k + 16 <= C; k += 16) {
        double a[16], b[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { a[q] = srcA[(k + q) * sstride + tid]; b[q] = srcB[(k + q) * sstride + tid]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) { dstA[(k + q) * dstride + tid] = a[q]; dstB[(k + q) * dstride + tid] = b[q]; }
    }
    for (; k + 4 <= C; k += 4) {
        double a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = srcA[(k + q) * sstride + tid]; b[q] = srcB[(k + q) * sstride + tid]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { dstA[(k + q) * dstride + tid] = a[q]; dstB[(k + q) * dstride + tid] = b[q]; }
    }
    for (; k < C; ++k) { dstA[k * dstride + tid] = srcA[k * sstride + tid]; dstB[k * dstride + tid] = srcB[k * sstride + tid]; }
}

// PoissonSolver::IterateGaussSeidel (PoissonSolver.cpp:66-77)
__device__ __forceinline__ double iterate_gs(const MgDesc& D, Atom& A, int l, double errorMin, int iterno, double* red, long* nsweeps)
{
    // the fused pass pays off only where the chunk is long compared with its 3x96-node warm-up (measured: C >= 512)
    if (iterno == 3 && !D.lv[l].seq && D.lv[l].logC >= kFuseMinLogC && l >= D.kcoop) {
        double e1, e2, e3;
        gs_fused3(D, A, l, red, e1, e2, e3);
        if (!(e1 < errorMin) && !(e2 < errorMin)) {          // the reference runs all three sweeps
            A.cur ^= (1u << l);
            *nsweeps += 3;
            return e3;
        }
        __syncthreads();                                      // rare: it stops early -- redo from the untouched input copy
    }
    if (D.lv[l].seq) {
        // LDS-resident level: the sweeps of the visit by one thread, no workgroup barrier between them; the others wait
        // for the result (the error norm of one thread's partial sums equals the workgroup sum: the others add zeros)
        const Lvl L = D.lv[l];
        if (threadIdx.x == 0) {
            const double dh = L.d * 0.5;
            unsigned cur = A.cur;
            double err = 1E10;
            int done = 0;
            for (int i = 0; i < iterno; ++i) {
                const unsigned c = (cur >> l) & 1u;
                const double err2 = seq_sweep(A.lds + 2 * kSeqCap + L.soff, A.lds + c * kSeqCap + L.soff,
                                              A.lds + (c ^ 1u) * kSeqCap + L.soff, L.n, dh);
                cur ^= (1u << l);
                err = sqrt(err2);
                ++done;
                if (err < errorMin) break;
            }
            red[16] = err;
            red[17] = done;
        }
        __syncthreads();
        const double err = red[16];
        const int done = static_cast<int>(red[17]);
        if (done & 1) A.cur ^= (1u << l);
        *nsweeps += done;
        __syncthreads();
        return err;
    }
    if (D.lv[l].stage == 3) {
        // A small level of one workgroup (129 ... 1025 nodes): staged like the others but laid out over 64 lanes and swept
        // by the first wave alone -- a quarter of the LDS traffic per warm-up step, no workgroup barrier between the
        // sweeps.  All four waves copy in and out.
        const Lvl L = D.lv[l];
        const double dh = L.d * 0.5;
        const int tid = threadIdx.x;
        const unsigned c0 = (A.cur >> l) & 1u;
        double* G0 = (c0 ? A.phi1 : A.phi0) + L.off;
        double* G1 = (c0 ? A.phi0 : A.phi1) + L.off;
        const double* Sg = A.src + L.off;
        double* PP = A.stage + kStagePad;
        double* SS = PP + kStageArr;
        const int lc = L.logC + L.logT - 6;                   // log2(nodes per lane) with 64 lanes
        const int nm1 = L.n - 1;
        const int Cg1 = (1 << L.logC) - 1;
        // LDS index j = k*64 + t <-> node i = t*2^lc + k <-> global storage index (i & (C-1))*T + (i >> logC)
        auto gidx = [&](int j) { const int i = ((j & 63) << lc) + (j >> 6); return ((i & Cg1) << L.logT) + (i >> L.logC); };
        {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = tid + q * kThreads;
                if (j < nm1) { const int gi = gidx(j); a[q] = G0[gi]; b[q] = Sg[gi]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = tid + q * kThreads;
                if (j < nm1) { PP[j] = a[q]; SS[j] = b[q]; }
            }
            if (tid == 0) PP[nm1] = G0[nm1];
        }
        __syncthreads();
        if (tid < 64) {
            const int lo_g = tid << lc;
            double err = 1E10;
            int done = 0;
            for (int i = 0; i < iterno; ++i) {
                double err2;
                switch (lc) {
                    case 1:  err2 = gs_lds<1, 64, 64>(SS, PP, tid, lo_g, dh); break;
                    case 2:  err2 = gs_lds<2, 64, 64>(SS, PP, tid, lo_g, dh); break;
                    case 3:  err2 = gs_lds<3, 64, 64>(SS, PP, tid, lo_g, dh); break;
                    default: err2 = gs_lds<4, 64, 64>(SS, PP, tid, lo_g, dh); break;
                }
                for (int off = 32; off > 0; off >>= 1) err2 += __shfl_xor(err2, off);
                err = sqrt(err2);
                ++done;
                if (err < errorMin) break;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (tid == 0) { red[16] = err; red[17] = done; }
        }
        __syncthreads();
        const double err = red[16];
        const int done = static_cast<int>(red[17]);
        *nsweeps += done;
        double* Gout = (done & 1) ? G1 : G0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = tid + q * kThreads;
            if (j < nm1) Gout[gidx(j)] = PP[j];
        }
        if (tid == 0) Gout[nm1] = PP[nm1];
        if (done & 1) A.cur ^= (1u << l);
        __syncthreads();
        return err;
    }
    if (D.lv[l].stage == 1) {
        // A level of one workgroup with <= 8193 nodes: the current copy and the source are copied to LDS once per visit
        // (same interleaved layout), the sweeps of the visit run there in place, the result goes back once.  A smoother
        // step then costs LDS issue slots instead of the 12 vector-memory instructions that bound it in HBM/L2.
        const Lvl L = D.lv[l];
        const double dh = L.d * 0.5;
        const int tid = threadIdx.x;
        const unsigned c0 = (A.cur >> l) & 1u;
        double* G0 = (c0 ? A.phi1 : A.phi0) + L.off;
        double* G1 = (c0 ? A.phi0 : A.phi1) + L.off;
        const double* Sg = A.src + L.off;
        double* PP = A.stage + kStagePad;
        double* SS = PP + kStageArr;
        const int C1 = 1 << L.logC;
        const bool fold = (A.pend == l + 1);                  // prolongation from level l+1 taken in while staging (do_prolong)
        const bool fold_r = (l > 0 && A.pend_r == l);         // restriction from level l-1 computed while staging (do_restrict)
        const bool fold_z = (A.pend_z == l);                  // resident group: the members have written the source, Phi = 0
        A.pend = 0;
        A.pend_r = 0;
        A.pend_z = 0;
        auto stage_in = [&]() { PROF_T0();
        if (fold_z) {
            for (int k0 = 0; k0 < C1; k0 += 8) {
                double b[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) b[q] = Sg[(k0 + q) * kThreads + tid];
#pragma unroll
                for (int q = 0; q < 8; ++q) { SS[(k0 + q) * kThreads + tid] = b[q]; PP[(k0 + q) * kThreads + tid] = 0; }
            }
            if (tid == 0) PP[L.n - 1] = 0;
        } else if (fold_r) {
            // PoissonSolver::Restrict (PoissonSolver.cpp:126-157) from level l-1 (same workgroup, same lane columns: coarse node
            // (t, k) sits under the fine nodes (t, 2k-1 .. 2k+1)) straight into the staging memory: Phi starts from 0, the
            // source also goes to the level's global array for the visit on the way back up.  Arithmetic of restrict_to.
            const Lvl Lf = D.lv[l - 1];
            const int logT = L.logT, Cf = 1 << Lf.logC;
            const double* __restrict__ pf = (((A.cur >> (l - 1)) & 1u) ? A.phi1 : A.phi0) + Lf.off;
            const double* __restrict__ sf = A.src + Lf.off;
            double* __restrict__ sc = A.src + L.off;
            const double dc = L.d;
            double pm = pf[((Cf - 1) << logT) + (tid > 0 ? tid - 1 : 0)];
#pragma unroll 4
            for (int k = 0; k < C1; ++k) {
                const double p0 = pf[((2 * k) << logT) + tid];
                const double pp = pf[((2 * k + 1) << logT) + tid];
                const double s0 = sf[((2 * k) << logT) + tid];
                double sv = 4. * (s0 + pm - 2. * p0 + pp) - dc * (pp - pm);
                if (k == 0 && tid == 0) sv = 0;                    // coarse node 0
                sc[(k << logT) + tid] = sv;
                SS[k * kThreads + tid] = sv;
                PP[k * kThreads + tid] = 0;
                pm = pp;
            }
            if (tid == 0) { sc[C1 << logT] = 0; PP[L.n - 1] = 0; }  // coarse node n-1
        } else if (fold) {
            // PoissonSolver::Prolong (PoissonSolver.cpp:110-123) from level l+1 added to what is staged (arithmetic of
            // prolong_from: fine(2j) += coarse(j), fine(2j-1) += 0.5 (coarse(j-1) + coarse(j))); the level's global copy is
            // brought up to date by the write-out at the end of the visit
            const Lvl Lc = D.lv[l + 1];
            const double* __restrict__ Pc = A.cur_phi(l + 1, Lc);
            auto corr = [&](int i, bool odd) -> double {
                if (!odd) return Pc[addr(Lc, i >> 1)];
                return 0.5 * (Pc[addr(Lc, (i - 1) >> 1)] + Pc[addr(Lc, (i + 1) >> 1)]);
            };
            for (int k0 = 0; k0 < C1; k0 += 4) {                    // C1 >= 8 on these levels
                double a[4], b[4], cadd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = k0 + q;
                    a[q] = G0[k * kThreads + tid];
                    b[q] = Sg[k * kThreads + tid];
                    cadd[q] = corr((tid << L.logC) + k, (q & 1) != 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    PP[(k0 + q) * kThreads + tid] = a[q] + cadd[q];
                    SS[(k0 + q) * kThreads + tid] = b[q];
                }
            }
            if (tid == 0) PP[L.n - 1] = G0[L.n - 1] + corr(L.n - 1, false);
        } else {
            copy_rows2(PP, SS, kThreads, G0, Sg, kThreads, C1);
            if (tid == 0) PP[L.n - 1] = G0[L.n - 1];
        }
        __syncthreads();
        PROF_ADD(5, l); };
        stage_in();
        const int lo_g = tid << L.logC;
        double err = 1E10;
        int done = 0;
        PROF_T0();
        bool fused_done = false;
        if (D.fuse3 && iterno == 3 && L.logC >= 2) {
            // the whole visit as one fused pass (gs_lds3); if the reference would have stopped after the first or the second sweep
            // (rare: err < errorMin), the level is staged again and the visit runs sweep by sweep below
            double f1, f2, f3;
            const bool lastl = tid == kThreads - 1;
            const double xN = PP[L.n - 1];
            const bool careful = (__builtin_amdgcn_readfirstlane(lo_g) <= kWarm3);
#define DFTA_GS3(LC) do { if (careful) gs_lds3<LC, kThreads, kThreads, 3, true>(SS, PP, tid, lo_g, lastl, xN, dh, f1, f2, f3); \
                          else gs_lds3<LC, kThreads, kThreads, 3, false>(SS, PP, tid, lo_g, lastl, xN, dh, f1, f2, f3); } while (0)
            switch (L.logC) {
                case 2:  DFTA_GS3(2); break;
                case 3:  DFTA_GS3(3); break;
                case 4:  DFTA_GS3(4); break;
                default: DFTA_GS3(5); break;
            }
#undef DFTA_GS3
            for (int off = 32; off > 0; off >>= 1) { f1 += __shfl_xor(f1, off); f2 += __shfl_xor(f2, off); f3 += __shfl_xor(f3, off); }
            __syncthreads();
            if ((tid & 63) == 0) { red[tid >> 6] = f1; red[4 + (tid >> 6)] = f2; red[8 + (tid >> 6)] = f3; }
            __syncthreads();
            const double e1 = sqrt((red[0] + red[1]) + (red[2] + red[3]));
            const double e2 = sqrt((red[4] + red[5]) + (red[6] + red[7]));
            const double e3 = sqrt((red[8] + red[9]) + (red[10] + red[11]));
            if (!(e1 < errorMin) && !(e2 < errorMin)) {
                err = e3;
                done = 3;
                *nsweeps += 3;
                fused_done = true;
            } else {
                __syncthreads();
                stage_in();
            }
        }
        if (!fused_done)
        for (int i = 0; i < iterno; ++i) {
            double err2;
            switch (L.logC) {
                case 0:  err2 = gs_lds<0, kThreads>(SS, PP, tid, lo_g, dh); break;
                case 1:  err2 = gs_lds<1, kThreads>(SS, PP, tid, lo_g, dh); break;
                case 2:  err2 = gs_lds<2, kThreads>(SS, PP, tid, lo_g, dh); break;
                case 3:  err2 = gs_lds<3, kThreads>(SS, PP, tid, lo_g, dh); break;
                case 4:  err2 = gs_lds<4, kThreads>(SS, PP, tid, lo_g, dh); break;
                default: err2 = gs_lds<5, kThreads>(SS, PP, tid, lo_g, dh); break;
            }
            err = sqrt(block_sum(err2, red));
            ++done;
            ++*nsweeps;
            if (err < errorMin) break;
        }
        PROF_ADD(7, l);
        // after `done` sweeps the current copy is G1 for odd counts, G0 for even ones
        double* Gout = (done & 1) ? G1 : G0;
        { PROF_T0();
        copy_rows(Gout, kThreads, PP, kThreads, C1);
        if (tid == 0) Gout[L.n - 1] = PP[L.n - 1];
        if (done & 1) A.cur ^= (1u << l);
        __syncthreads();
        PROF_ADD(6, l); }
        return err;
    }
    if (D.lv[l].stage == 2) {
        // A level shared by the G workgroups of the atom (lane ids run over the group: member g owns the columns
        // g*256 .. g*256+255 of every row): each member stages its columns plus kStageH halo columns in front of them (the
        // previous member's last lanes: the warm-up reads up to 96 nodes = 96/C lanes back) and Phi of the first node of
        // the next member.  After every sweep the members exchange the new values of those boundary nodes through the
        // level's other global copy (exactly where the unstaged sweep would have put them) around the group barrier that
        // the error norm needs anyway; the last sweep of the visit writes the whole part out instead.
        const Lvl L = D.lv[l];
        const double dh = L.d * 0.5;
        const int tid = threadIdx.x;
        const int C = 1 << L.logC, logT = L.logT, g = A.g;
        const int Hc = kWarm >> L.logC;                       // lanes in the halo that the warm-up touches
        const unsigned c0 = (A.cur >> l) & 1u;
        double* Gin = (c0 ? A.phi1 : A.phi0) + L.off;
        double* Gout = (c0 ? A.phi0 : A.phi1) + L.off;
        const double* Sg = A.src + L.off;
        double* PP = A.stage + kStageH;
        double* SS = PP + kStageArr;
        const int col0 = g * kThreads;
        const int end_g = (g == A.G - 1) ? (C << logT) : (col0 + kThreads);     // node behind this member's part
        const bool fold = (A.pend == l + 1);
        const bool fold_r = (l > 0 && A.pend_r == l);
        A.pend = 0;
        A.pend_r = 0;
        PROF_T0();
        if (fold_r) {
            // PoissonSolver::Restrict (PoissonSolver.cpp:126-157) from level l-1 folded into the copy: the source of every
            // staged node -- own columns and halo columns -- comes straight from the fine level (complete and visible since
            // its last barrier), Phi starts from 0.  The own columns of the source also go to the level's global array, where
            // the later visits of this V-cycle and the next restriction read them (after this visit's barriers).
            const Lvl Lf = D.lv[l - 1];
            const double* __restrict__ Pf = (((A.cur >> (l - 1)) & 1u) ? A.phi1 : A.phi0) + Lf.off;
            const double* __restrict__ Sf = A.src + Lf.off;
            double* __restrict__ Sgw = A.src + L.off;
            const int lim = L.n - 1;
            const double dc = L.d;
            auto src_of_node = [&](int i) -> double {
                if (i <= 0 || i >= lim) return 0.0;
                const int twoi = 2 * i;
                const double pm = Pf[addr(Lf, twoi - 1)], p0 = Pf[addr(Lf, twoi)], pp = Pf[addr(Lf, twoi + 1)];
                return 4. * (Sf[addr(Lf, twoi)] + pm - 2. * p0 + pp) - dc * (pp - pm);
            };
            const int lane0 = col0 + tid;
            for (int k0 = 0; k0 < C; k0 += 4) {
                double b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q] = src_of_node((lane0 << L.logC) + k0 + q);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    PP[(k0 + q) * kStageRS + tid] = 0;
                    SS[(k0 + q) * kStageRS + tid] = b[q];
                    Sgw[((k0 + q) << logT) + lane0] = b[q];
                }
            }
            for (int idx = tid; idx < C * kStageH; idx += kThreads) {
                const int k = idx / kStageH, c = idx % kStageH - kStageH;
                const int lanec = col0 + c;
                PP[k * kStageRS + c] = 0;
                SS[k * kStageRS + c] = lanec >= 0 ? src_of_node((lanec << L.logC) + k) : 0.0;
            }
            if (tid == 0) {
                PP[C * kStageRS] = 0;
                if (g == A.G - 1) Sgw[C << logT] = 0;          // source of node n-1
            }
        } else if (!fold) {
            copy_rows2(PP, SS, kStageRS, Gin + col0, Sg + col0, 1 << logT, C);
            for (int idx = tid; idx < C * kStageH; idx += kThreads) {       // halo columns (member 0: in-bounds padding / the
                const int k = idx / kStageH, c = idx % kStageH - kStageH;   // previous level's tail, never used)
                PP[k * kStageRS + c] = Gin[(k << logT) + col0 + c];
                SS[k * kStageRS + c] = Sg[(k << logT) + col0 + c];
            }
            if (tid == 0) PP[C * kStageRS] = Gin[end_g];
        } else {
            // PoissonSolver::Prolong (PoissonSolver.cpp:110-123) from level l+1 folded into the copy: every member adds the
            // correction to what it stages -- its own columns, the halo columns and the node behind its part -- from the
            // coarse level, which is complete and visible since that level's last barrier.  The separate pass over the fine
            // level and the group barrier after it are gone; the level's global copy is brought up to date by the write-out
            // at the end of this visit.
            const Lvl Lc = D.lv[l + 1];
            const double* __restrict__ Pc = (((A.cur >> (l + 1)) & 1u) ? A.phi1 : A.phi0) + Lc.off;
            auto corr = [&](int i, bool odd) -> double {               // fine node i (odd-ness known to the caller)
                if (!odd) return Pc[addr(Lc, i >> 1)];                  // fine(2j) += coarse(j)
                return 0.5 * (Pc[addr(Lc, (i - 1) >> 1)] + Pc[addr(Lc, (i + 1) >> 1)]);   // fine(2j-1) += 0.5 (coarse(j-1) + coarse(j))
            };
            const int lane0 = col0 + tid;
            for (int k0 = 0; k0 < C; k0 += 4) {                         // C >= 4, a multiple of 4
                double a[4], b[4], cadd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = k0 + q, gi = (k << logT) + lane0;
                    a[q] = Gin[gi];
                    b[q] = Sg[gi];
                    cadd[q] = corr((lane0 << L.logC) + k, (q & 1) != 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    PP[(k0 + q) * kStageRS + tid] = a[q] + cadd[q];
                    SS[(k0 + q) * kStageRS + tid] = b[q];
                }
            }
            for (int idx = tid; idx < C * kStageH; idx += kThreads) {
                const int k = idx / kStageH, c = idx % kStageH - kStageH;
                const int lanec = col0 + c, gi = (k << logT) + lanec;
                double v = Gin[gi];
                if (lanec >= 0) v += corr((lanec << L.logC) + k, (k & 1) != 0);
                PP[k * kStageRS + c] = v;
                SS[k * kStageRS + c] = Sg[gi];
            }
            if (tid == 0) {
                const int i_end = (g == A.G - 1) ? (L.n - 1) : ((col0 + kThreads) << L.logC);
                PP[C * kStageRS] = Gin[end_g] + corr(i_end, false);    // even node in both cases
            }
        }
        __syncthreads();
        PROF_ADD(5, l);
        auto write_out = [&]() {
            copy_rows(Gout + col0, 1 << logT, PP, kStageRS, C);
            if (g == A.G - 1 && tid == 0) Gout[C << logT] = PP[C * kStageRS];
        };
        const int lo_g = (col0 + tid) << L.logC;
        double err = 1E10;
        int done = 0;
        for (int i = 0; i < iterno; ++i) {
            double err2;
            { PROF_T0();
            switch (L.logC) {
                case 2:  err2 = gs_lds<2, kStageRS>(SS, PP, tid, lo_g, dh); break;
                case 3:  err2 = gs_lds<3, kStageRS>(SS, PP, tid, lo_g, dh); break;
                case 4:  err2 = gs_lds<4, kStageRS>(SS, PP, tid, lo_g, dh); break;
                default: err2 = gs_lds<5, kStageRS>(SS, PP, tid, lo_g, dh); break;
            }
            PROF_ADD(4, l); }
            ++done;
            ++*nsweeps;
            const bool last = (i == iterno - 1);
            __syncthreads();
            if (last) { PROF_T0(); write_out(); PROF_ADD(6, l); }
            else {
                // into the sentinel-filled slots of the exchange that travels with the fast sum (group_sum_fast)
                if (tid < (Hc << L.logC)) {                    // the last Hc lanes' nodes: what the next member's warm-up reads
                    const int k = tid & (C - 1), c = kThreads - Hc + (tid >> L.logC);
                    exchange_store(A, tid, PP[k * kStageRS + c]);
                }
                if (tid == 0) exchange_store(A, kXchg - 1, PP[0]);   // the first node: right neighbour of the previous member's last one
            }
            { PROF_T0(); err = sqrt(last ? group_sum_fast<kStageRS, true>(A, err2, red, PP, L.logC, Hc) : group_sum_fast<kStageRS>(A, err2, red, PP, L.logC, Hc)); PROF_ADD(3, l); }
            if (last) break;
            if (err < errorMin) {                              // the reference stops here: publish everything, meet once more
                write_out();
                group_sync(A);
                break;
            }
            // (the neighbours' nodes arrived with the sum: halo columns and PP[C * kStageRS] are up to date)
            { double* t = Gin; Gin = Gout; Gout = t; }
        }
        if (done & 1) A.cur ^= (1u << l);
        return err;
    }
    double err = 1E10;
    for (int i = 0; i < iterno; ++i) {
        err = gauss_seidel(D, A, l, red);
        ++*nsweeps;
        if (err < errorMin) break;
    }
    return err;
}

// PoissonSolver::Restrict(lvl) (PoissonSolver.cpp:126-157): fine = lvl-1 -> coarse = lvl
__device__ __forceinline__ void restrict_to(const MgDesc& D, Atom& A, int lvl)
{
    const Lvl Lc = D.lv[lvl], Lf = D.lv[lvl - 1];
    const double* Pf = A.cur_phi(lvl - 1, Lf);
    double* Pc = A.cur_phi(lvl, Lc);
    const double* Sf = A.src_of(Lf);
    double* Sc = A.src_of(Lc);
    const int lim = Lc.n - 1;
    const bool coop = lvl < D.kcoop;       // both levels are shared by the group
    if (!Lc.seq && !Lf.seq && Lc.logT == Lf.logT && Lc.logC >= 1) {
        // both levels chunked over the same lanes: coarse node (t, k) sits under fine nodes (t, 2k-1 .. 2k+1), so lane t
        // streams its own column -- all rows are wave-uniform, every access is coalesced, 4 rows per batch in flight
        const int T = 1 << Lc.logT, Cc = 1 << Lc.logC, Cf = 1 << Lf.logC, logT = Lc.logT;
        const int t = coop ? A.lane() : static_cast<int>(threadIdx.x);
        const double* __restrict__ pf = (((A.cur >> (lvl - 1)) & 1u) ? A.phi1 : A.phi0) + Lf.off;
        const double* __restrict__ sf = A.src + Lf.off;
        double* __restrict__ pc = (((A.cur >> lvl) & 1u) ? A.phi1 : A.phi0) + Lc.off;
        double* __restrict__ sc = A.src + Lc.off;
        const double dc = Lc.d;
        if (t < T) {
            // row -1 of lane t is the last row of lane t-1 (node t*Cf - 1); lane 0 never uses it (coarse node 0 is fixed)
            double pm = pf[((Cf - 1) << logT) + (t > 0 ? t - 1 : 0)];
#pragma unroll 4
            for (int k = 0; k < Cc; ++k) {
                const double p0 = pf[((2 * k) << logT) + t];
                const double pp = pf[((2 * k + 1) << logT) + t];
                const double s0 = sf[((2 * k) << logT) + t];
                double s = 4. * (s0 + pm - 2. * p0 + pp) - dc * (pp - pm);
                if (k == 0 && t == 0) s = 0;                       // coarse node 0
                sc[(k << logT) + t] = s;
                pc[(k << logT) + t] = 0;
                pm = pp;
            }
        }
        if (t == 0) { sc[Cc << logT] = 0; pc[Cc << logT] = 0; }    // coarse node n-1
        if (coop) group_sync(A); else __syncthreads();
        return;
    }
    // the fine level is shared but the coarse one is workgroup 0's (lvl == kcoop): the members compute it together all the
    // same (their copy of the coarse level's `cur` bit dates from the last prolongation out of it and is still valid)
    const bool share = coop || (lvl - 1 < D.kcoop && A.G > 1);
    for (int idx = share ? A.lane() : static_cast<int>(threadIdx.x); idx < Lc.n; idx += share ? kThreads * A.G : kThreads) {
        const int i = node_of(Lc, idx);
        Pc[idx] = 0;
        double s = 0;
        if (i > 0 && i < lim) {
            const int twoi = 2 * i;
            const double pm = Pf[addr(Lf, twoi - 1)], p0 = Pf[addr(Lf, twoi)], pp = Pf[addr(Lf, twoi + 1)];
            s = 4. * (Sf[addr(Lf, twoi)] + pm - 2. * p0 + pp) - Lc.d * (pp - pm);
        }
        Sc[idx] = s;
    }
    if (share) group_sync(A); else __syncthreads();
}

// PoissonSolver::Prolong (PoissonSolver.cpp:110-123): coarse = lvl -> fine = lvl-1 (additive)
__device__ __forceinline__ void prolong_from(const MgDesc& D, Atom& A, int lvl)
{
    const Lvl Lc = D.lv[lvl], Lf = D.lv[lvl - 1];
    const double* Pc = A.cur_phi(lvl, Lc);
    double* Pf = A.cur_phi(lvl - 1, Lf);
    const bool coop = lvl - 1 < D.kcoop;   // the fine level is shared by the group (the coarse one may be workgroup 0's)
    if (!Lc.seq && !Lf.seq && Lc.logT == Lf.logT && Lc.logC >= 1) {
        // same lane-column structure as in restrict_to: fine (t, 2k) += coarse (t, k); fine (t, 2k-1) += 0.5 (coarse (t, k-1) + coarse (t, k))
        const int T = 1 << Lc.logT, Cc = 1 << Lc.logC, Cf = 1 << Lf.logC, logT = Lc.logT;
        const int t = coop ? A.lane() : static_cast<int>(threadIdx.x);
        const double* __restrict__ pc = (((A.cur >> lvl) & 1u) ? A.phi1 : A.phi0) + Lc.off;
        double* __restrict__ pf = (((A.cur >> (lvl - 1)) & 1u) ? A.phi1 : A.phi0) + Lf.off;
        if (t < T) {
            // coarse value left of this lane's first node: last row of lane t-1 (unused by lane 0: fine node -1 does not exist)
            double cm = pc[((Cc - 1) << logT) + (t > 0 ? t - 1 : 0)];
#pragma unroll 4
            for (int k = 0; k < Cc; ++k) {
                const double c = pc[(k << logT) + t];
                pf[((2 * k) << logT) + t] += c;
                if (k > 0) pf[((2 * k - 1) << logT) + t] += 0.5 * (cm + c);
                else if (t > 0) pf[((Cf - 1) << logT) + t - 1] += 0.5 * (cm + c);      // fine node t*Cf - 1 lives in lane t-1's column
                cm = c;
            }
            if (t == T - 1) {                                                            // coarse node n-1 and the fine node below it
                const double cN = pc[Cc << logT];
                pf[Cf << logT] += cN;
                pf[((Cf - 1) << logT) + t] += 0.5 * (cm + cN);
            }
        }
        if (coop) group_sync(A); else __syncthreads();
        return;
    }
    for (int idx = coop ? A.lane() : static_cast<int>(threadIdx.x); idx < Lc.n; idx += coop ? kThreads * A.G : kThreads) {
        const int i = node_of(Lc, idx);
        const double c = Pc[idx];
        Pf[addr(Lf, 2 * i)] += c;
        if (i > 0) Pf[addr(Lf, 2 * i - 1)] += 0.5 * (Pc[addr(Lc, i - 1)] + c);
    }
    if (coop) group_sync(A); else __syncthreads();
}

struct Counters { long sweeps, vcycles; };

// PoissonSolver::Initialize without its final smoothing (PoissonSolver.cpp:80-103)
__device__ __forceinline__ void initialize(const MgDesc& D, Atom& A, double lowB, double highB)
{
    A.cur = 0;
    {
        const Lvl L0 = D.lv[0];
        double* P0 = A.cur_phi(0, L0);
        const bool coop = D.kcoop > 0;
        if (coop || A.g == 0)
            for (int idx = coop ? A.lane() : static_cast<int>(threadIdx.x); idx < L0.n; idx += coop ? kThreads * A.G : kThreads) P0[idx] = 0;
    }
    for (int l = 1; l < D.levels; ++l) {
        const Lvl L = D.lv[l], Lf = D.lv[l - 1];
        const bool coop = l < D.kcoop;
        // the source of level l-1 must be complete: it was written by the whole group for l <= kcoop
        if (A.G > 1 && l <= D.kcoop) group_sync(A); else __syncthreads();
        if (!coop && A.g != 0) continue;
        for (int idx = coop ? A.lane() : static_cast<int>(threadIdx.x); idx < L.n; idx += coop ? kThreads * A.G : kThreads) {
            const int p = node_of(L, idx);
            double s = 0;
            if (p > 0 && p < L.n - 1) s = 4 * A.src_of(Lf)[addr(Lf, 2 * p)];
            A.src_of(L)[idx] = s;
            A.cur_phi(l, L)[idx] = 0;
        }
    }
    __syncthreads();
    const int cl = D.levels - 1;
    if (threadIdx.x == 0 && A.g == 0) {
        double* Pc = A.cur_phi(cl, D.lv[cl]);
        Pc[addr(D.lv[cl], 0)] = lowB;
        Pc[addr(D.lv[cl], D.lv[cl].n - 1)] = highB;
    }
    __syncthreads();
}

// The whole cycle structure of PoissonSolver::FullCycle (PoissonSolver.h:89-124) as ONE loop over "legs", so that
// the smoother, restriction and prolongation are each inlined exactly once:
//   step 0                      : the 15 sweeps on the coarsest level that end Initialize (PoissonSolver.cpp:105)
//   steps 1 .. 2*nramp          : for i = levels-2 .. 1: Descend(last -> i), Ascend(i -> last)     (FMG ramp)
//   step 2*nramp+1              : Descend(last -> 0, errorMinLast)
//   then pairs                  : VCycle = Ascend(0 -> last), Descend(last -> 0); stop on err < errorMinLast or 100 cycles
// Ascend(from,to): { GS(from); Restrict(from+1); GS(from+1); ... ; GS(to) }      (PoissonSolver.cpp:162-171)
// Descend(from,to): { Prolong(from); GS(from-1); ... ; GS(to) }                  (PoissonSolver.cpp:173-186)
// The operations as the members of a group execute them: levels below kcoop by everybody, the others by workgroup 0
// alone (the other members skip them and meet workgroup 0 again at the barrier in front of the first shared operation).
// may_fold: the caller visits level lvl next (the V-cycle driver); a staged shared level then computes its source -- own
// columns and halo -- while it is copied to LDS (iterate_gs), and the pass and the group barrier below are skipped
__device__ __forceinline__ void do_restrict(const MgDesc& D, Atom& A, int lvl, bool may_fold = false)
{
    if (may_fold && lvl < D.kcoop && D.lv[lvl].stage == 2 && !D.nofold) { A.pend_r = lvl; return; }
    // both levels workgroup 0's, staged, on the same lane columns: folded into the copy-in likewise (iterate_gs, stage 1)
    if (may_fold && lvl - 1 >= D.kcoop && D.lv[lvl].stage == 1 && D.lv[lvl - 1].stage == 1 && D.lv[lvl].logT == D.lv[lvl - 1].logT &&
        D.lv[lvl].logC >= 2 && !D.nofold) {
        if (A.g == 0) A.pend_r = lvl;
        return;
    }
    if (lvl - 1 < D.kcoop || A.g == 0) restrict_to(D, A, lvl);
}

// may_fold: the caller visits level lvl-1 next (the V-cycle driver); a staged shared level then takes the correction in
// while it is copied to LDS (iterate_gs) and the pass below is skipped
__device__ __forceinline__ void do_prolong(const MgDesc& D, Atom& A, int lvl, bool may_fold = false)
{
    if (lvl - 1 < D.kcoop) {
        if (lvl >= D.kcoop && A.G > 1) {
            // the coarse level is workgroup 0's: wait for it, and learn which of its two copies is current
            unsigned* pub = reinterpret_cast<unsigned*>(A.part + 6 * A.G);
            if (A.g == 0 && threadIdx.x == 0) *pub = A.cur;
            group_sync(A);
            const unsigned shared = (1u << D.kcoop) - 1u;
            A.cur = (A.cur & shared) | (*pub & ~shared);
        }
        if (may_fold && D.lv[lvl - 1].stage == 2 && !D.nofold) { A.pend = lvl; return; }
        prolong_from(D, A, lvl);
    } else if (may_fold && D.lv[lvl - 1].stage == 1 && D.lv[lvl - 1].logC >= 2 && !D.lv[lvl].seq && !D.nofold) {
        if (A.g == 0) A.pend = lvl;                            // workgroup 0's own staged level: folded into its copy-in
    } else if (A.g == 0) prolong_from(D, A, lvl);
}

__device__ __forceinline__ double do_iterate(const MgDesc& D, Atom& A, int l, double errorMin, int iterno, double* red, long* nsweeps)
{
    if (l < D.kcoop || A.g == 0) return iterate_gs(D, A, l, errorMin, iterno, red, nsweeps);
    return 1E10;
}

// ---- coarse section -----------------------------------------------------------------------------------------------
// The levels with <= 1025 nodes cost a V-cycle more in waiting than in arithmetic: every visit pays a copy-in from global
// memory, workgroup barriers and a copy-out for ~1 us of sweeps.  From the first V-cycle on, the part of a cycle below
// level cs_top -- restrict .. iterate down to the coarsest level and prolong .. iterate back up -- therefore runs in the
// first wave of workgroup 0 alone, on copies that stay in LDS (the staging memory is idle meanwhile): no barrier, no
// global round trip until the wave hands level cs_top back.  Same operations in the same order on the same values as the
// level-by-level code (the error norms are summed in a different order, as everywhere).  Nothing on these levels
// carries over from one V-cycle to the next: the restriction rewrites the sources and zeroes Phi on the way down.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// index of node i inside the LDS copy of a level (n nodes, lc as in MgDesc::cs_lc)
__device__ __forceinline__ int cs_idx(int n, int lc, int i)
{
    if (lc < 0 || i == n - 1) return i;
    return ((i & ((1 << lc) - 1)) << 6) + (i >> lc);
}

// One in-place sweep of a level in natural order by ONE thread (PoissonSolver.cpp:40-64); returns sum dPhi^2.  Explicit LDS
// pointers; the loads of a batch are issued together ahead of the recurrence (the right neighbours are read before the
// batch overwrites them).
__device__ __forceinline__ double seq_sweep_inplace(const double* __restrict__ Sg, double* __restrict__ Pg, const int n, const double dh)
{
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) const double lds_cf64;
    lds_cf64* S = (lds_cf64*)(Sg);
    lds_f64* P = (lds_f64*)(Pg);
    double err2 = 0;
    double y = 2.0 * P[0];
    const int limit = n - 1;
    double old = P[1];
    constexpr int kB = 8;
    int i = 1;
    for (; i + kB <= limit; i += kB) {
        double xp[kB], sv[kB], xo[kB];
#pragma unroll
        for (int q = 0; q < kB; ++q) { xp[q] = P[i + q + 1]; sv[q] = S[i + q]; }
#pragma unroll
        for (int q = 0; q < kB; ++q) {
            y = gs_point2(sv[q], y, xp[q], dh);
            const double x = 0.5 * y;
            const double dif = old - x;
            err2 += dif * dif;
            xo[q] = x;
            old = xp[q];
        }
#pragma unroll
        for (int q = 0; q < kB; ++q) P[i + q] = xo[q];
    }
    for (; i < limit; ++i) {
        const double xp = P[i + 1];
        y = gs_point2(S[i], y, xp, dh);
        const double x = 0.5 * y;
        const double dif = old - x;
        err2 += dif * dif;
        P[i] = x;
        old = xp;
    }
    return err2;
}

// lane t <- v of lane t-1 (DPP wave_shr:1 on both halves), lane 0 keeps `keep`
__device__ __forceinline__ double lane_shr1(double keep, double v)
{
    const long long k = __builtin_bit_cast(long long, keep), x = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp((int)k, (int)x, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(k >> 32), (int)(x >> 32), 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// The same sweep (n <= 65 nodes, natural order) by the 64 lanes of a wave: lane t keeps S, the old right neighbour and the
// old value of node t + 1 in registers and recomputes its node in every step from the candidate of lane t - 1, handed on
// through a DPP lane shift (lane 0: from the boundary value).  The candidate of node i is final from step i on -- its
// left neighbour's is final one step earlier and its other inputs never change -- so after n - 2 steps every lane holds
// exactly what the sequential loop computes: the same n - 2 dependent updates, but 7 instructions each (5 flops, 2 lane
// shifts) instead of a scalar loop with its LDS traffic (~46 ns per node).  Returns this lane's share of sum dPhi^2.
__device__ __forceinline__ double seq_sweep_wave(const double* __restrict__ Sg, double* __restrict__ Pg, const int n, const double dh)
{
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) const double lds_cf64;
    lds_cf64* S = (lds_cf64*)(Sg);
    lds_f64* P = (lds_f64*)(Pg);
    const int i = (threadIdx.x & 63) + 1;             // this lane's node
    const bool mine = i < n - 1;
    const double s = mine ? S[i] : 0.0;
    const double xp = mine ? P[i + 1] : 0.0;
    const double old = mine ? P[i] : 0.0;
    double yin = 2.0 * P[0];                          // stays the boundary value in lane 0
    double cand = 0;
    // (extra steps are harmless -- the candidates are at their fixed point -- so the trip count is rounded up to the unroll)
    for (int k = 0; k < n - 2; k += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cand = gs_point2(s, yin, xp, dh);
            yin = lane_shr1(yin, cand);
        }
    }
    const double x = 0.5 * cand;
    const double dif = old - x;
    if (mine) P[i] = x;
    return mine ? dif * dif : 0.0;
}

// IterateGaussSeidel on the LDS copy of level l; all 64 lanes of the wave call
__device__ __forceinline__ double cs_iterate(const MgDesc& D, Atom& A, int l, double errorMin, int iterno, long* nsweeps)
{
    const Lvl L = D.lv[l];
    const double dh = L.d * 0.5;
    const int lane = threadIdx.x;
    double* P = A.stage + D.cs_phi[l];
    const double* S = A.stage + D.cs_src[l];
    const int lc = D.cs_lc[l];
    double err = 1E10;
    for (int i = 0; i < iterno; ++i) {
        double err2 = 0;
        if (lc >= 0) {
            switch (lc) {
                case 1:  err2 = gs_lds<1, 64, 64>(S, P, lane, lane << lc, dh); break;
                case 2:  err2 = gs_lds<2, 64, 64>(S, P, lane, lane << lc, dh); break;
                case 3:  err2 = gs_lds<3, 64, 64>(S, P, lane, lane << lc, dh); break;
                default: err2 = gs_lds<4, 64, 64>(S, P, lane, lane << lc, dh); break;
            }
            for (int off = 32; off > 0; off >>= 1) err2 += __shfl_xor(err2, off);
        } else {
#ifdef DFTA_POISSON_SEQ_ONE_LANE
            if (lane == 0) err2 = seq_sweep_inplace(S, P, L.n, dh);
            err2 = __shfl(err2, 0);
#else
            err2 = seq_sweep_wave(S, P, L.n, dh);
            for (int off = 32; off > 0; off >>= 1) err2 += __shfl_xor(err2, off);
#endif
        }
        err = sqrt(err2);
        ++*nsweeps;
        wave_sync();
        if (err < errorMin) break;
    }
    return err;
}

// PoissonSolver::Restrict into the LDS copy of level lc; the fine level is read through `fine(i)` / `fsrc(i)` (node index)
template <int NT, typename FP, typename FS>
__device__ __forceinline__ void cs_restrict(const MgDesc& D, Atom& A, int lvl, FP fine, FS fsrc)
{
    const Lvl Lc = D.lv[lvl];
    double* P = A.stage + D.cs_phi[lvl];
    double* S = A.stage + D.cs_src[lvl];
    const int lc = D.cs_lc[lvl], lim = Lc.n - 1;
    // four nodes per thread and pass, all their loads in flight together
    for (int i0 = threadIdx.x; i0 < Lc.n; i0 += 4 * NT) {
        double pm[4], p0[4], pp[4], s0[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * NT;
            if (i > 0 && i < lim) { pm[q] = fine(2 * i - 1); p0[q] = fine(2 * i); pp[q] = fine(2 * i + 1); s0[q] = fsrc(2 * i); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * NT;
            if (i < Lc.n) {
                double sv = 0;
                if (i > 0 && i < lim) sv = 4. * (s0[q] + pm[q] - 2. * p0[q] + pp[q]) - Lc.d * (pp[q] - pm[q]);
                const int idx = cs_idx(Lc.n, lc, i);
                S[idx] = sv;
                P[idx] = 0;
            }
        }
    }
    if (NT == 64) wave_sync();
}

// The whole part of a V-cycle below level cs_top (first wave of workgroup 0; the level above it, cs_top-1, is current in
// global memory on entry, and level cs_top is current in global memory on return).
// entry, by the whole workgroup: the restriction from level cs_top-1 (global) into the LDS copy of level cs_top
__device__ __forceinline__ void coarse_section_enter(const MgDesc& D, Atom& A)
{
    const int top = D.cs_top;
    const Lvl Lf = D.lv[top - 1];
    const double* __restrict__ Pf = (((A.cur >> (top - 1)) & 1u) ? A.phi1 : A.phi0) + Lf.off;
    const double* __restrict__ Sf = A.src + Lf.off;
    cs_restrict<kThreads>(D, A, top, [&](int i) { return Pf[addr(Lf, i)]; }, [&](int i) { return Sf[addr(Lf, i)]; });
}

// exit, by the whole workgroup: level cs_top goes back to its current global copy
__device__ __forceinline__ void coarse_section_leave(const MgDesc& D, Atom& A)
{
    const int top = D.cs_top;
    const Lvl L = D.lv[top];
    double* __restrict__ G = (((A.cur >> top) & 1u) ? A.phi1 : A.phi0) + L.off;
    const double* P = A.stage + D.cs_phi[top];
    const int lc = D.cs_lc[top];
    for (int i = threadIdx.x; i < L.n; i += kThreads) G[addr(L, i)] = P[cs_idx(L.n, lc, i)];
}

__device__ __forceinline__ void coarse_section(const MgDesc& D, Atom& A, double errorMin, int iterno, long* nsweeps)
{
    const int top = D.cs_top, last = D.levels - 1;
    // down: iterate, then restrict / iterate
    cs_iterate(D, A, top, errorMin, iterno, nsweeps);
    for (int l = top + 1; l <= last; ++l) {
        const int nf = D.lv[l - 1].n, lcf = D.cs_lc[l - 1];
        const double* Pf = A.stage + D.cs_phi[l - 1];
        const double* Sf = A.stage + D.cs_src[l - 1];
        cs_restrict<64>(D, A, l, [&](int i) { return Pf[cs_idx(nf, lcf, i)]; }, [&](int i) { return Sf[cs_idx(nf, lcf, i)]; });
        cs_iterate(D, A, l, errorMin, iterno, nsweeps);
    }
    // up: prolong (PoissonSolver.cpp:110-123), iterate
    for (int l = last - 1; l >= top; --l) {
        const int nc = D.lv[l + 1].n, lcc = D.cs_lc[l + 1], nf = D.lv[l].n, lcf = D.cs_lc[l];
        const double* Pc = A.stage + D.cs_phi[l + 1];
        double* Pf = A.stage + D.cs_phi[l];
        for (int i = threadIdx.x; i < nc; i += 64) {
            const double c = Pc[cs_idx(nc, lcc, i)];
            Pf[cs_idx(nf, lcf, 2 * i)] += c;
            if (i > 0) Pf[cs_idx(nf, lcf, 2 * i - 1)] += 0.5 * (Pc[cs_idx(nc, lcc, i - 1)] + c);
        }
        wave_sync();
        cs_iterate(D, A, l, errorMin, iterno, nsweeps);
    }
}

template <bool CS>
__device__ __forceinline__ double run_cycles(const MgDesc& D, Atom& A, int first_step, int max_vcycles, double errorMin,
                                             double errorMinLast, double* red, Counters& c)
{
    const int last = D.levels - 1;
    const int nramp = D.levels - 2 > 0 ? D.levels - 2 : 0;
    double err = 0;
    bool cs_skip = false;
    for (int step = first_step;; ++step) {
        int from, to, iterno = 3;
        double emin = errorMin;
        bool vleg_down = false;
        if (step == 0) { from = to = last; iterno = 15; }
        else if (step <= 2 * nramp) {
            const int q = (step - 1) >> 1;
            const int i = D.levels - 2 - q;
            if ((step - 1) & 1) { from = i; to = last; } else { from = last; to = i; }
        } else if (step == 2 * nramp + 1) { from = last; to = 0; emin = errorMinLast; }
        else {
            emin = errorMinLast;
            if ((step - (2 * nramp + 2)) & 1) { from = last; to = 0; vleg_down = true; } else { from = 0; to = last; }
        }
        const int dir = (from > to) ? -1 : 1;
        err = 1E10;
        if (!(dir < 0 && from == to)) {
            int first_lvl = (dir > 0) ? from : from - 1;
            if (CS && cs_skip) { first_lvl = D.cs_top - 1; cs_skip = false; }       // the section has done last .. cs_top
            for (int lvl = first_lvl;; lvl += dir) {
                if (CS && D.cs_top > 0 && dir > 0 && lvl == D.cs_top && lvl > from && to == last && step > 2 * nramp + 1) {
                    // the rest of this leg and the beginning of the next one (coarse_section)
                    if (A.g == 0) {
                        PROF_T0();
                        long nsw = 0;
                        coarse_section_enter(D, A);
                        __syncthreads();
                        if (threadIdx.x < 64) coarse_section(D, A, emin, iterno, &nsw);
                        if (threadIdx.x == 0) red[17] = static_cast<double>(nsw);
                        __syncthreads();
                        c.sweeps += static_cast<long>(red[17]);
                        coarse_section_leave(D, A);
                        __syncthreads();
                        PROF_ADD(5, 20);
                    }
                    cs_skip = true;
                    break;
                }
                {
                    PROF_T0();
                    if (dir > 0) { if (lvl > from) do_restrict(D, A, lvl, true); }
                    else do_prolong(D, A, lvl + 1, true);
                    PROF_ADD(dir > 0 ? 0 : 1, lvl);
                }
                {
                    PROF_T0();
                    err = do_iterate(D, A, lvl, emin, iterno, red, &c.sweeps);
                    PROF_ADD(2, lvl);
                }
                if (lvl == to) break;
            }
        }
        if (vleg_down) {
            ++c.vcycles;
            if (err < errorMinLast || c.vcycles >= max_vcycles) break;
        }
    }
    return err;
}

// ==== resident shared levels =======================================================================================
// The grouped solve above stages a shared level from global memory for every visit and writes it back: per V-cycle and
// shared level two copies through the coherent level behind the XCDs' L2s, three exchanges and a full barrier (70 % of a
// single-atom solve was hand-over).  Here a group of kResG member workgroups keeps its stretch of EVERY shared
// level in LDS for the whole solve (node range [m, m+1) * 128 * C_l of level l: 128 lanes x C_l nodes, C_l = 32, 16, 8, 4 at 131073
// nodes), with halo columns: the 115 nodes in front of the stretch (what the fused visit's warm-up reads) and up to ten nodes
// behind it.  A visit is ONE fused pass (gs_lds3) followed by ONE exchange in sentinel-validated slots that carries the three
// partial sums of dPhi^2, the new values of the nodes the neighbours keep in their halos and -- on the way down -- the tail
// of the next level's source, so that restriction and prolongation are local (PoissonSolver.cpp:110-157 on own nodes and
// halos).  One more workgroup per atom, the coarse workgroup, runs the levels below (<= 8193 nodes) with the code above and
// meets the members at the two hand-overs of a cycle, through the global arrays of the first coarse level.
#ifdef DFTA_POISSON_RPROF
__device__ unsigned long long g_rprof[2 * 8 * 8];    // [role][category][level & 7], ticks of member 0 / of the coarse workgroup of atom 0
#define RPROF_T0() const long long rprof_t0 = clock64()
#define RPROF_ADD(cat, lvl) do { if (threadIdx.x == 0) R.prof[(cat) * 8 + ((lvl) & 7)] += clock64() - rprof_t0; } while (0)   /* LDS: a global read-modify-write costs microseconds */
#else
#define RPROF_T0()
#define RPROF_ADD(cat, lvl)
#endif
constexpr int kResNT = 128;                    // sweeping lanes of a member (its first two waves; all four move data)
constexpr int kResG = 32;                      // members per atom
constexpr int kResWG = kResG + 1;              // + the coarse workgroup (participant kResG of every exchange)
constexpr int kResX = 272;                     // payload doubles per participant and buffer
constexpr int kResMaxShared = 4;
constexpr int kResXTail = 0, kResXSrc = 128, kResXHead = 256, kResXS0 = 266;
__host__ __device__ constexpr size_t res_slot_doubles() { return (size_t)3 * kResWG * 4 + (size_t)3 * kResWG * kResX; }

template <int LOGC> struct ResLay {
    static constexpr int C = 1 << LOGC;
    static constexpr int H = (kWarm3 + 3 + C - 1) / C;       // halo columns in front of the stretch
    static constexpr int HN = H * C;                         // = nodes in the left halo
    static constexpr int RS = H + kResNT + 1;                // + one column behind it
    static constexpr int N = kResNT * C;                     // own nodes
    // nodes behind the stretch that are kept up to date: what the fused pass reads (3 of Phi, 2 of the source) on the last
    // shared level, and what the restriction of those needs on the levels above it
    static constexpr int NRP = LOGC >= 5 ? 10 : (LOGC == 4 ? 6 : (LOGC == 3 ? 4 : 3));
    static constexpr int NRS = LOGC >= 5 ? 9 : (LOGC == 4 ? 5 : (LOGC == 3 ? 3 : 2));
    static constexpr int doubles = C * RS;
    // node j relative to the member's first node (-HN <= j < N + C) -> index relative to column 0, row 0
    __device__ static __forceinline__ int off(int j) { return (j & (C - 1)) * RS + (j >> LOGC); }
};
static_assert(ResLay<5>::HN <= 128 && ResLay<4>::HN <= 128 && ResLay<3>::HN <= 128 && ResLay<2>::HN <= 128, "payload layout");

struct Res {
    int role;                 // 0: member, 1: the coarse workgroup
    int m;                    // member index
    int kres;                 // shared levels 0 .. kres-1
    int logC0;                // log2(nodes per lane) on level 0
    double* shm;              // the member's LDS
    int pat[kResMaxShared];   // column 0 of level l's Phi array inside it (the source follows C * RS doubles later); read through
    int sat[kResMaxShared];   // res_pp / res_ss with compile-time indices only -- a dynamic index would move the struct to scratch
    double* xs;               // [3][kResWG][4] sums
    double* xp;               // [3][kResWG][kResX] payload
    unsigned seq;             // exchanges so far
#ifdef DFTA_POISSON_RPROF
    unsigned long long* prof; // [8][8] ticks per category and level, in LDS
#endif
};

// a value that is the same in every lane (read from LDS after a barrier), moved to scalar registers: what is derived from it --
// the decisions of the cycle -- then stays scalar, and the loops that depend on it keep their state out of the vector registers
__device__ __forceinline__ double res_uniform(const double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readfirstlane(static_cast<int>(b));
    const int hi = __builtin_amdgcn_readfirstlane(static_cast<int>(b >> 32));
    return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
}
// the lane id behind an optimisation barrier: the per-lane addresses of an operation are computed where it runs.  (Hoisted out of
// the cycle loop -- dozens of inlined operations -- they would be live across everything and spill to scratch memory.)
__device__ __forceinline__ int res_tid()
{
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ double* res_pp(const Res& R, const int l)
{
    return R.shm + (l == 0 ? R.pat[0] : (l == 1 ? R.pat[1] : (l == 2 ? R.pat[2] : R.pat[3])));
}
__device__ __forceinline__ double* res_ss(const Res& R, const int l)
{
    return R.shm + (l == 0 ? R.sat[0] : (l == 1 ? R.sat[1] : (l == 2 ? R.sat[2] : R.sat[3])));
}
__device__ __forceinline__ void res_store(double* p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double* res_mine(const Res& R, int idx)     // slot idx of this participant's payload in the NEXT exchange
{
    return R.xp + ((size_t)(R.seq % 3u) * kResWG + (R.role ? kResG : R.m)) * kResX + idx;
}

// One exchange between the kResWG participants of an atom.  Everybody publishes four doubles (members: the partial sums of a
// visit; the coarse workgroup: zeros and, in [3], the number it hands over) and receives their totals (added in a fixed tree
// order, the same for everybody, so that all participants take the same decisions).  Members also receive up to two payload
// values per thread from their neighbours' slots (published before the call with res_store(res_mine())): value v = tid and
// v = tid + 256 with  v < 128: tail of the left neighbour (nt), 128 <= v < 256: source tail of the left neighbour (ns),
// 256 <= v: head of the right neighbour (nh).  want_cb (coarse workgroup): thread 64 + m, m >= 1, receives in cb[] the four
// values around the boundary between members m-1 and m (last node of m-1, first two nodes and first source value of m).
// FENCED: release before publishing, acquire after the last arrival (plain stores / loads of global level storage around it).
template <bool FENCED>
__device__ __forceinline__ void res_exchange(Atom& A, Res& R, const double m0, const double m1, const double m2, const double m3,
                                             const int nt, const int ns, const int nh, const int cb_hn, double* red,
                                             double (&tot)[4], double& r0, double& r1, double (&cb)[4])
{
    const unsigned s = R.seq++;
    const int b = s % 3u, tid = res_tid();
    const int me = R.role ? kResG : R.m;
    double* slots = R.xs + (size_t)b * kResWG * 4;
    const double* pay = R.xp + (size_t)b * kResWG * kResX;
    if (tid == 0) {
        if (FENCED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        res_store(slots + me * 4 + 0, m0); res_store(slots + me * 4 + 1, m1);
        res_store(slots + me * 4 + 2, m2); res_store(slots + me * 4 + 3, m3);
    }
    const double* p0 = nullptr;
    const double* p1 = nullptr;
    if (R.role == 0) {
        if (tid < 128) { if (R.m > 0 && tid < nt) p0 = pay + (size_t)(R.m - 1) * kResX + tid; }
        else { if (R.m > 0 && tid - 128 < ns) p0 = pay + (size_t)(R.m - 1) * kResX + tid; }
        if (tid < nh && R.m < kResG - 1) p1 = pay + (size_t)(R.m + 1) * kResX + kResXHead + tid;
    }
    const double* pc[4] = {nullptr, nullptr, nullptr, nullptr};
    if (R.role == 1 && cb_hn > 0 && tid > 64 && tid < 64 + kResG) {
        const int m = tid - 64;
        pc[0] = pay + (size_t)(m - 1) * kResX + kResXTail + cb_hn - 1;
        pc[1] = pay + (size_t)m * kResX + kResXHead;
        pc[2] = pay + (size_t)m * kResX + kResXHead + 1;
        pc[3] = pay + (size_t)m * kResX + kResXS0;
    }
    auto ld = [](const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto is_sent = [](double x) { return static_cast<unsigned long long>(__double_as_longlong(x)) == kFastSentinel; };
    // Up to ten values per thread.  Every round re-issues the loads of ALL values that have not arrived yet together (one round
    // trip per round, not per value); a lost participant must not hang the GPU: after spin_max rounds the group's abort flag is
    // raised (and honoured at once by everybody who sees it), the host repeats the solve with one workgroup per atom.
    const double sentv = __longlong_as_double(static_cast<long long>(kFastSentinel));
    double q[4] = {0, 0, 0, 0}, v0 = 0, v1 = 0, c4[4] = {0, 0, 0, 0};
    const bool sums = tid < kResWG;             // first wave
    const double* pq = slots + (sums ? tid : 0) * 4;
    if (sums) { q[0] = sentv; q[1] = sentv; q[2] = sentv; q[3] = sentv; }
    if (p0) v0 = sentv;
    if (p1) v1 = sentv;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (pc[k]) c4[k] = sentv;
    for (int spins = A.gave_up ? A.spin_max : 0;; ++spins) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (is_sent(q[k])) q[k] = ld(pq + k);
        if (is_sent(v0)) v0 = ld(p0);
        if (is_sent(v1)) v1 = ld(p1);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (is_sent(c4[k])) c4[k] = ld(pc[k]);
        const bool missing = is_sent(q[0]) || is_sent(q[1]) || is_sent(q[2]) || is_sent(q[3]) || is_sent(v0) || is_sent(v1) ||
                             is_sent(c4[0]) || is_sent(c4[1]) || is_sent(c4[2]) || is_sent(c4[3]);
        if (!missing) break;
        bool give_up = spins > A.spin_max;
        if (!give_up && (spins & 255) == 255) give_up = (__hip_atomic_load(A.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x80000000u) != 0;
        if (give_up) {
            __hip_atomic_fetch_or(A.ctr, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            A.gave_up = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (is_sent(q[k])) q[k] = 0; if (is_sent(c4[k])) c4[k] = 0; }
            if (is_sent(v0)) v0 = 0;
            if (is_sent(v1)) v1 = 0;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (tid < 64) {
        const double extra = __shfl(q[3], kResG);             // the coarse workgroup's fourth value
        if (tid >= kResG) q[3] = 0;
        for (int off = 32; off > 0; off >>= 1) {
            q[0] += __shfl_xor(q[0], off); q[1] += __shfl_xor(q[1], off); q[2] += __shfl_xor(q[2], off); q[3] += __shfl_xor(q[3], off);
        }
        if (tid == 0) {
            if (FENCED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            red[20] = q[0]; red[21] = q[1]; red[22] = q[2]; red[23] = extra;
        }
    }
    __syncthreads();
    tot[0] = res_uniform(red[20]); tot[1] = res_uniform(red[21]); tot[2] = res_uniform(red[22]); tot[3] = res_uniform(red[23]);
    // everybody has published exchange s, i.e. has read what it needed of exchange s-1: this participant's part of that buffer
    // is reset (by the threads that will store into it again in exchange s+2)
    {
        const double sent = __longlong_as_double(static_cast<long long>(kFastSentinel));
        const int bo = (s + 2u) % 3u;
        if (tid < 4) res_store(R.xs + ((size_t)bo * kResWG + me) * 4 + tid, sent);
        double* mine = R.xp + ((size_t)bo * kResWG + me) * kResX;
        res_store(mine + tid, sent);
        if (tid + 256 < kResX) res_store(mine + tid + 256, sent);
    }
    r0 = v0; r1 = v1;
    cb[0] = c4[0]; cb[1] = c4[1]; cb[2] = c4[2]; cb[3] = c4[3];
    __syncthreads();                                           // red[] may be reused
}

__device__ __forceinline__ double res_restrict_value(double s0, double pm, double p0, double pq, double dc)
{
    return 4. * (s0 + pm - 2. * p0 + pq) - dc * (pq - pm);     // PoissonSolver.cpp:126-157, the expression of restrict_to
}

// IterateGaussSeidel(l, errorMin, 3) on a shared level by a member.  down: the caller restricts to level l+1 next.
template <int LOGC>
__device__ __forceinline__ double res_visit(const MgDesc& D, Atom& A, Res& R, const int l, const double errorMin, const bool down,
                                            double* red, long* nsweeps)
{
    using Y = ResLay<LOGC>;
    using Yc = ResLay<(LOGC > 2 ? LOGC - 1 : 2)>;
    constexpr int C = Y::C, RS = Y::RS;
    const Lvl L = D.lv[l];
    const double dh = L.d * 0.5;
    const int tid = res_tid();
    double* PP = res_pp(R, l);
    double* SS = res_ss(R, l);
    const bool active = tid < kResNT;
    const int lo_g = (R.m * kResNT + tid) << LOGC;
    const bool lastl = (R.m == kResG - 1) && tid == kResNT - 1;
    const double xN = PP[Y::off(Y::N)];
    const bool careful = R.m == 0 && (__builtin_amdgcn_readfirstlane(lo_g) <= kWarm3);
    const bool to_shared = down && (l + 1 < R.kres);
    const bool to_coarse = down && (l + 1 == R.kres);
    // the old values of the stretch, in case the reference stops after the first or second sweep: parked in the level's (otherwise
    // unused) second global copy -- stores that nobody waits for; a thread reads back only what it wrote itself
    double* __restrict__ park = A.phi1 + L.off + (size_t)R.m * Y::N;
    RPROF_T0();
    if (active) {
#pragma unroll
        for (int k = 0; k < C; ++k) park[k * kResNT + tid] = PP[k * RS + tid];
    }
    double f1 = 0, f2 = 0, f3 = 0;
    if (D.dbg & 2) { f1 = 1; f2 = 1; f3 = 1; }
    else if (careful) gs_lds3<LOGC, RS, kResNT, 3, true>(SS, PP, tid, lo_g, lastl, xN, dh, f1, f2, f3);
    else gs_lds3<LOGC, RS, kResNT, 3, false>(SS, PP, tid, lo_g, lastl, xN, dh, f1, f2, f3);
    if (!active) { f1 = 0; f2 = 0; f3 = 0; }
    for (int off = 32; off > 0; off >>= 1) { f1 += __shfl_xor(f1, off); f2 += __shfl_xor(f2, off); f3 += __shfl_xor(f3, off); }
    __syncthreads();                                           // the pass's stores are visible
    if ((tid & 63) == 0 && active) { red[tid >> 6] = f1; red[4 + (tid >> 6)] = f2; red[8 + (tid >> 6)] = f3; }
    // what the neighbours (and, at the hand-over, the coarse workgroup) need of this stretch
    auto publish = [&]() {
        if (tid < Y::HN) res_store(res_mine(R, kResXTail + tid), PP[Y::off(Y::N - Y::HN + tid)]);
        if (tid < Y::NRP) res_store(res_mine(R, kResXHead + tid), PP[Y::off(tid)]);
        if (tid == kResXS0 - 256) res_store(res_mine(R, kResXS0), SS[Y::off(0)]);
        if constexpr (LOGC > 2) {
            if (to_shared && tid >= 128 && tid - 128 < Yc::HN) {
                // the tail of the next level's source: PoissonSolver::Restrict on nodes that lie inside this stretch
                const int jc = Yc::N - Yc::HN + (tid - 128), j = 2 * jc;
                res_store(res_mine(R, kResXSrc + tid - 128),
                          res_restrict_value(SS[Y::off(j)], PP[Y::off(j - 1)], PP[Y::off(j)], PP[Y::off(j + 1)], D.lv[l + 1].d));
            }
        }
        if (to_coarse && active) {
            // the first level of the coarse workgroup: this stretch's part of its source goes to global memory (the exchange
            // that follows is fenced); the node under the boundary to the left neighbour is the coarse workgroup's
            const Lvl Lk = D.lv[l + 1];
#pragma unroll
            for (int k = 0; k < C / 2; ++k) {
                if (tid == 0 && k == 0) continue;
                const int j = 2 * (tid * (C / 2) + k);
                const int ic = (R.m * Y::N) / 2 + tid * (C / 2) + k;
                A.src[Lk.off + addr(Lk, ic)] = res_restrict_value(SS[Y::off(j)], PP[Y::off(j - 1)], PP[Y::off(j)], PP[Y::off(j + 1)], Lk.d);
            }
        }
    };
    RPROF_ADD(0, l);
    { RPROF_T0();
    publish();
    __syncthreads();                                           // red[] written
    RPROF_ADD(1, l); }
    const double s1 = (red[0] + red[1]), s2 = (red[4] + red[5]), s3 = (red[8] + red[9]);
    double tot[4], r0, r1, cb[4];
    const int ns = to_shared ? Yc::HN : 0;
    { RPROF_T0();
    if (to_coarse) res_exchange<true>(A, R, s1, s2, s3, 0.0, Y::HN, ns, Y::NRP, 0, red, tot, r0, r1, cb);
    else res_exchange<false>(A, R, s1, s2, s3, 0.0, Y::HN, ns, Y::NRP, 0, red, tot, r0, r1, cb);
    RPROF_ADD(2, l); }
    const double e1 = sqrt(tot[0]), e2 = sqrt(tot[1]), e3 = sqrt(tot[2]);
    const int K = (e1 < errorMin) ? 1 : ((e2 < errorMin) ? 2 : 3);
    if (K < 3) {
        // the reference stops after sweep K: back to the old values (the halos have not been touched yet), the same pass
        // storing sweep K's values, and the exchange once more
        RPROF_T0();
        if (active) {
#pragma unroll
            for (int k = 0; k < C; ++k) PP[k * RS + tid] = park[k * kResNT + tid];
        }
        __syncthreads();
        if (K == 1) gs_lds3<LOGC, RS, kResNT, 1, true>(SS, PP, tid, lo_g, lastl, xN, dh, f1, f2, f3);
        else gs_lds3<LOGC, RS, kResNT, 2, true>(SS, PP, tid, lo_g, lastl, xN, dh, f1, f2, f3);
        __syncthreads();
        publish();
        if (to_coarse) res_exchange<true>(A, R, 0.0, 0.0, 0.0, 0.0, Y::HN, ns, Y::NRP, 0, red, tot, r0, r1, cb);
        else res_exchange<false>(A, R, 0.0, 0.0, 0.0, 0.0, Y::HN, ns, Y::NRP, 0, red, tot, r0, r1, cb);
        RPROF_ADD(7, l);
    }
    // the neighbours' new values go into the halos
    { RPROF_T0();
    if (R.m > 0) {
        if (tid < Y::HN) PP[Y::off(-Y::HN + tid)] = r0;
        if constexpr (LOGC > 2) {
            if (to_shared && tid >= 128 && tid - 128 < Yc::HN) res_ss(R, l + 1)[Yc::off(-Yc::HN + tid - 128)] = r0;
        }
    }
    if (R.m < kResG - 1 && tid < Y::NRP) PP[Y::off(Y::N + tid)] = r1;
    __syncthreads();
    RPROF_ADD(3, l); }
    *nsweeps += K;
    return K == 1 ? e1 : (K == 2 ? e2 : e3);
}

// The coarse workgroup's part in a member visit: it follows the exchange(s) to take the same decisions; at the hand-over it
// completes the source of its first level (the nodes under the members' boundaries) and marks it for a zero-Phi stage-in.
__device__ __forceinline__ double res_visit_passive(const MgDesc& D, Atom& A, Res& R, const int l, const double errorMin, const bool down,
                                                    double* red, long* nsweeps)
{
    const bool to_coarse = down && (l + 1 == R.kres);
    const int logC = R.logC0 - l;
    const int C = 1 << logC;
    const int hn = ((kWarm3 + 3 + C - 1) / C) * C;
    double tot[4], r0, r1, cb[4];
    RPROF_T0();
    if (to_coarse) res_exchange<true>(A, R, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, hn, red, tot, r0, r1, cb);
    else res_exchange<false>(A, R, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, red, tot, r0, r1, cb);
    RPROF_ADD(0, l);
    const double e1 = sqrt(tot[0]), e2 = sqrt(tot[1]), e3 = sqrt(tot[2]);
    const int K = (e1 < errorMin) ? 1 : ((e2 < errorMin) ? 2 : 3);
    if (K < 3) {
        if (to_coarse) res_exchange<true>(A, R, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, hn, red, tot, r0, r1, cb);
        else res_exchange<false>(A, R, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, red, tot, r0, r1, cb);
    }
    if (to_coarse) {
        const Lvl Lk = D.lv[l + 1];
        const int tid = res_tid();
        const int per = (kResNT << logC) / 2;                  // coarse nodes per member
        if (tid > 64 && tid < 64 + kResG) {
            const int m = tid - 64;
            A.src[Lk.off + addr(Lk, m * per)] = res_restrict_value(cb[3], cb[0], cb[1], cb[2], Lk.d);
        }
        if (tid == 0) { A.src[Lk.off + addr(Lk, 0)] = 0; A.src[Lk.off + addr(Lk, Lk.n - 1)] = 0; }
        A.pend_z = l + 1;
        __syncthreads();
    }
    *nsweeps += 0;      // the members count the sweeps of the shared levels
    return K == 1 ? e1 : (K == 2 ? e2 : e3);
}

// PoissonSolver::Restrict from shared level lc-1 (LOGC nodes per lane) to shared level lc on the member's stretch: own nodes and the
// nodes behind it; the nodes in front of it arrived with the last exchange.  Phi of the coarse level starts from zero everywhere.
template <int LOGC>
__device__ __forceinline__ void res_restrict_local(const MgDesc& D, Res& R, const int lc)
{
    using Yf = ResLay<LOGC>;
    using Yc = ResLay<LOGC - 1>;
    const double* __restrict__ Pf = res_pp(R, lc - 1);
    const double* __restrict__ Sf = res_ss(R, lc - 1);
    double* __restrict__ Pc = res_pp(R, lc);
    double* __restrict__ Sc = res_ss(R, lc);
    const double dc = D.lv[lc].d;
    const int tid = res_tid(), t = tid & (kResNT - 1), half = tid >> 7;
#pragma unroll
    for (int kk = 0; kk < Yc::C / 2; ++kk) {
        const int k = half * (Yc::C / 2) + kk;
        const int jc = t * Yc::C + k, j = 2 * jc;
        double sv = res_restrict_value(Sf[Yf::off(j)], Pf[Yf::off(j - 1)], Pf[Yf::off(j)], Pf[Yf::off(j + 1)], dc);
        if (R.m == 0 && jc == 0) sv = 0;                       // coarse node 0
        Sc[k * Yc::RS + t] = sv;
        Pc[k * Yc::RS + t] = 0;
    }
    // behind the stretch (for the last member only the level's last node: source 0)
    if (tid < Yc::NRS) {
        const int jc = Yc::N + tid, j = 2 * jc;
        double sv = 0;
        if (R.m < kResG - 1) sv = res_restrict_value(Sf[Yf::off(j)], Pf[Yf::off(j - 1)], Pf[Yf::off(j)], Pf[Yf::off(j + 1)], dc);
        Sc[Yc::off(jc)] = sv;
    }
    if (tid < Yc::C) Pc[Yc::off(Yc::N + tid)] = 0;
    for (int i = tid; i < Yc::HN; i += kThreads) Pc[Yc::off(-Yc::HN + i)] = 0;
    __syncthreads();
}

// PoissonSolver::Prolong (PoissonSolver.cpp:110-123) into shared level lf (LOGC nodes per lane) on the member's stretch and its halos.
// coarse(i): Phi of coarse node i RELATIVE to the member's first coarse node.
template <int LOGC, typename CF>
__device__ __forceinline__ void res_prolong(Res& R, const int lf, CF coarse)
{
    using Y = ResLay<LOGC>;
    double* __restrict__ Pf = res_pp(R, lf);
    const int tid = res_tid(), t = tid & (kResNT - 1), half = tid >> 7;
    auto corr = [&](int j) -> double {                         // fine node j relative to the stretch
        if ((j & 1) == 0) return coarse(j >> 1);
        return 0.5 * (coarse((j - 1) >> 1) + coarse((j + 1) >> 1));
    };
#pragma unroll 4
    for (int kk = 0; kk < Y::C / 2; ++kk) {
        const int k = half * (Y::C / 2) + kk;
        Pf[k * Y::RS + t] += corr(t * Y::C + k);
    }
    if (R.m > 0) for (int i = tid; i < Y::HN; i += kThreads) Pf[Y::off(-Y::HN + i)] += corr(-Y::HN + i);
    if (tid < Y::NRP && (R.m < kResG - 1 || tid == 0)) Pf[Y::off(Y::N + tid)] += corr(Y::N + tid);
    __syncthreads();
}

// shared levels of a member at the start of a solve (PoissonSolver.h:55-74, PoissonSolver.cpp:80-103): Phi = 0, Source_l[p] =
// 4 Source_{l-1}[2p] = 4^l Source_0[2^l p] (exact scalings) on own nodes and halos, straight from the density
template <int LOGC>
__device__ __forceinline__ void res_init_level(const MgDesc& D, Atom& A, Res& R, const int l, const double* __restrict__ rho,
                                               const double* __restrict__ r, const double* __restrict__ psrc, const int src_all)
{
    using Y = ResLay<LOGC>;
    double* PP = res_pp(R, l);
    double* SS = res_ss(R, l);
    const int n = D.lv[l].n, N0 = D.lv[0].n;
    const int a = R.m * Y::N;
    const Lvl L0 = D.lv[0];
    for (int idx = threadIdx.x; idx < Y::C * Y::RS; idx += kThreads) {
        const int row = idx / Y::RS, col = idx % Y::RS - Y::H;
        const int i = a + col * Y::C + row;                    // node of level l
        double sv = 0;
        if (l == 0) {
            if (i >= 0 && i < N0) {
                sv = r[i];
                if (src_all || (i > 0 && i < N0 - 1)) sv *= psrc[i] * rho[i];
                // level 0 of the global storage: the source for the unit hooks (dfta_poisson_full_cycle repeats the cycle on it)
                if (col >= 0 && col < kResNT) A.src[L0.off + addr(L0, i)] = sv;
                else if (i == N0 - 1) A.src[L0.off + addr(L0, i)] = sv;
            }
        } else if (i > 0 && i < n - 1) {
            const int i0 = i << l;
            sv = r[i0];
            sv *= psrc[i0] * rho[i0];
            for (int q = 0; q < l; ++q) sv = 4 * sv;
        }
        PP[row * Y::RS + col] = 0;
        SS[row * Y::RS + col] = sv;
    }
}

// the coarse workgroup's levels at the start of a solve
__device__ __forceinline__ void res_init_coarse(const MgDesc& D, Atom& A, const Res& R, const double* __restrict__ rho,
                                                const double* __restrict__ r, const double* __restrict__ psrc, const double lowB, const double highB)
{
    A.cur = 0;
    for (int l = R.kres; l < D.levels; ++l) {
        const Lvl L = D.lv[l];
        for (int idx = threadIdx.x; idx < L.n; idx += kThreads) {
            const int p = node_of(L, idx);
            double sv = 0;
            if (p > 0 && p < L.n - 1) {
                const int i0 = p << l;
                sv = r[i0];
                sv *= psrc[i0] * rho[i0];
                for (int q = 0; q < l; ++q) sv = 4 * sv;
            }
            A.src_of(L)[idx] = sv;
            A.cur_phi(l, L)[idx] = 0;
        }
    }
    __syncthreads();
    const int cl = D.levels - 1;
    if (threadIdx.x == 0) {
        double* Pc = A.cur_phi(cl, D.lv[cl]);
        Pc[addr(D.lv[cl], 0)] = lowB;
        Pc[addr(D.lv[cl], D.lv[cl].n - 1)] = highB;
    }
    __syncthreads();
}

#define DFTA_RES_LEVEL(lc, CALL)                     \
    switch (lc) {                                    \
        case 5: { constexpr int LC = 5; CALL; } break; \
        case 4: { constexpr int LC = 4; CALL; } break; \
        case 3: { constexpr int LC = 3; CALL; } break; \
        default: { constexpr int LC = 2; CALL; } break; \
    }

// PoissonSolver::FullCycle (PoissonSolver.h:89-124) for both roles of a resident group: the leg structure of run_cycles, every
// operation carried out by whoever owns the level (members: levels < kres, coarse workgroup: the others), the two roles meeting
// in the exchanges of the members' visits and at the hand-over from the first coarse level back to the last shared one.
__device__ __forceinline__ double res_cycles(const MgDesc& D, Atom& A, Res& R, const int max_vcycles, const double errorMin,
                                             const double errorMinLast, double* red, Counters& c)
{
    const int last = D.levels - 1, kres = R.kres;
    const int nramp = D.levels - 2 > 0 ? D.levels - 2 : 0;
    const bool coarse = R.role == 1;
    double err = 0;
    bool cs_skip = false;
    for (int step = 0;; ++step) {
        int from, to, iterno = 3;
        double emin = errorMin;
        bool vleg_down = false;
        if (step == 0) { from = to = last; iterno = 15; }
        else if (step <= 2 * nramp) {
            const int q = (step - 1) >> 1;
            const int i = D.levels - 2 - q;
            if ((step - 1) & 1) { from = i; to = last; } else { from = last; to = i; }
        } else if (step == 2 * nramp + 1) { from = last; to = 0; emin = errorMinLast; }
        else {
            emin = errorMinLast;
            if ((step - (2 * nramp + 2)) & 1) { from = last; to = 0; vleg_down = true; } else { from = 0; to = last; }
        }
        const int dir = (from > to) ? -1 : 1;
        err = 1E10;
        if (!(dir < 0 && from == to)) {
            int first_lvl = (dir > 0) ? from : from - 1;
            if (cs_skip) { first_lvl = D.cs_top - 1; cs_skip = false; }
            for (int lvl = first_lvl;; lvl += dir) {
                if (D.cs_top > 0 && dir > 0 && lvl == D.cs_top && lvl > from && to == last && step > 2 * nramp + 1) {
                    if (coarse && !(D.dbg & 1)) {
                        RPROF_T0();
                        long nsw = 0;
                        coarse_section_enter(D, A);
                        __syncthreads();
                        if (threadIdx.x < 64) coarse_section(D, A, emin, iterno, &nsw);
                        if (threadIdx.x == 0) red[17] = static_cast<double>(nsw);
                        __syncthreads();
                        c.sweeps += static_cast<long>(res_uniform(red[17]));
                        coarse_section_leave(D, A);
                        __syncthreads();
                        RPROF_ADD(3, 0);
                    }
                    cs_skip = true;
                    break;
                }
                // transfer into level lvl
                { RPROF_T0();
                if (dir > 0) {
                    if (lvl > from) {                                          // Restrict(lvl): fine lvl-1 -> coarse lvl
                        if (lvl < kres) { if (!coarse && !(D.dbg & 4)) { DFTA_RES_LEVEL(R.logC0 - (lvl - 1), (res_restrict_local<(LC > 2 ? LC : 3)>(D, R, lvl))) } }
                        else if (lvl > kres) { if (coarse) do_restrict(D, A, lvl, true); }
                        // lvl == kres: the members have written the source during their visit of level kres-1
                    }
                } else {                                                       // Prolong: coarse lvl+1 -> fine lvl
                    if (lvl + 1 < kres) {
                        if (!coarse && !(D.dbg & 4)) {
                            DFTA_RES_LEVEL(R.logC0 - lvl, ({
                                using Yc = ResLay<(LC > 2 ? LC - 1 : 2)>;
                                const double* Pc = res_pp(R, lvl + 1);
                                res_prolong<(LC > 2 ? LC : 3)>(R, lvl, [&](int i) { return Pc[Yc::off(i)]; });
                            }))
                        }
                    } else if (lvl + 1 == kres) {
                        // hand-over: the coarse workgroup has written the first coarse level back to its current global copy
                        double tot[4], r0, r1, cb[4];
                        res_exchange<true>(A, R, 0.0, 0.0, 0.0, coarse ? static_cast<double>((A.cur >> kres) & 1u) : 0.0, 0, 0, 0, 0, red, tot, r0, r1, cb);
                        if (!coarse) {
                            const Lvl Lk = D.lv[kres];
                            const double* __restrict__ Pg = ((tot[3] != 0.0) ? A.phi1 : A.phi0) + Lk.off;
                            DFTA_RES_LEVEL(R.logC0 - lvl, ({
                                using Y = ResLay<LC>;
                                const int a_c = (R.m * Y::N) / 2;
                                const int nck = Lk.n;
                                res_prolong<LC>(R, lvl, [&](int i) { const int ic = a_c + i; return (ic >= 0 && ic < nck) ? Pg[addr(Lk, ic)] : 0.0; });
                            }))
                        }
                    } else if (coarse) do_prolong(D, A, lvl + 1, true);
                }
                RPROF_ADD(dir > 0 ? 4 : ((lvl + 1 == kres) ? 6 : 5), lvl); }
                // IterateGaussSeidel(lvl)
                if (lvl < kres) {
                    const bool down = dir > 0 && lvl < to;
                    if (coarse) err = res_visit_passive(D, A, R, lvl, emin, down, red, &c.sweeps);
                    else { DFTA_RES_LEVEL(R.logC0 - lvl, (err = res_visit<LC>(D, A, R, lvl, emin, down, red, &c.sweeps))) }
                } else if (coarse && !(D.dbg & 1)) {
                    RPROF_T0();
                    err = res_uniform(iterate_gs(D, A, lvl, emin, iterno, red, &c.sweeps));
                    RPROF_ADD(2, lvl - kres);
                }
                if (lvl == to) break;
            }
        }
        if (vleg_down) {
            ++c.vcycles;
            if (err < errorMinLast || c.vcycles >= max_vcycles) break;
        }
    }
    return err;
}

// SolvePoissonNonUniform (PoissonSolver.h:51-81) by a resident group: kResWG blocks per atom
__global__ __launch_bounds__(kThreads) void k_poisson_solve_res(const MgDesc* __restrict__ Dp, double* __restrict__ phi0, double* __restrict__ phi1,
                                                                double* __restrict__ src, const int* __restrict__ Z,
                                                                const double* __restrict__ density, const double* __restrict__ r,
                                                                const double* __restrict__ psrc, double* __restrict__ U,
                                                                int* __restrict__ vcycles, double* __restrict__ errs,
                                                                unsigned long long* __restrict__ total_vcycles,
                                                                unsigned* __restrict__ group_ctr, double* __restrict__ res_slots,
                                                                const int* __restrict__ skip, int fault, int src_all)
{
    __shared__ double red[32];
    __shared__ double shm[3 * kSeqCap + 2 * kStageArr];
    const MgDesc& D = *Dp;
    const int a = blockIdx.x / kResWG, g = blockIdx.x % kResWG;
    if (skip && skip[a]) return;
    if (fault && g == kResG - 1) return;                           // fault injection (tests): this member never arrives
    Atom A;
    A.phi0 = phi0 + (size_t)a * D.per_atom;
    A.phi1 = phi1 + (size_t)a * D.per_atom;
    A.src = src + (size_t)a * D.per_atom;
    A.lds = shm;
    A.stage = shm + 3 * kSeqCap;
    A.cur = 0;
    A.G = 1;                       // the code of the non-shared levels runs in the coarse workgroup alone
    A.g = (g == kResG) ? 0 : 1;
    A.ctr = group_ctr + a;
    A.bar = 0;
    A.part = nullptr; A.fslot = nullptr; A.xchg = nullptr;
    A.fseq = 0;
    A.pend = 0; A.pend_r = 0; A.pend_z = 0;
    A.spin_max = D.spin_max;
    A.gave_up = false;
    Res R;
    R.role = (g == kResG) ? 1 : 0;
    R.m = g;
    R.kres = D.res_kres;
    R.logC0 = D.res_logC0;
    R.xs = res_slots + (size_t)a * res_slot_doubles();
    R.xp = R.xs + (size_t)3 * kResWG * 4;
    R.seq = 0;
#ifdef DFTA_POISSON_RPROF
    __shared__ unsigned long long prof_acc[64];
    if (threadIdx.x < 64) prof_acc[threadIdx.x] = 0;
    R.prof = prof_acc;
    __syncthreads();
#endif
    R.shm = shm;
    {
        int at = 0;
#pragma unroll
        for (int l = 0; l < kResMaxShared; ++l) {
            const int lc = R.logC0 - l;
            const int C = 1 << (lc > 0 ? lc : 0);
            const int H = (kWarm3 + 3 + C - 1) / C, RS = H + kResNT + 1;
            R.pat[l] = at + H;
            R.sat[l] = at + C * RS + H;
            if (l < R.kres) at += 2 * C * RS;
        }
    }
    const int N = D.lv[0].n;
    const double* rho = density + (size_t)a * N;
    Counters c{0, 0};
    if (R.role == 0) {
        for (int l = 0; l < R.kres; ++l) { DFTA_RES_LEVEL(R.logC0 - l, (res_init_level<LC>(D, A, R, l, rho, r, psrc, src_all))) }
        __syncthreads();
    } else {
        res_init_coarse(D, A, R, rho, r, psrc, 0.0, (double)Z[a]);
    }
    const double err = res_cycles(D, A, R, 100, 1E-3, 1E-14, red, c);      // FullCycle(1E-3, 1E-14), PoissonSolver.h:78
    if (R.role == 0) {
        // U = PhiLevels[0] (PoissonSolver.h:80); also into level 0 of the global storage (copy 0) for the unit hooks
        const Lvl L0 = D.lv[0];
        DFTA_RES_LEVEL(R.logC0, ({
            using Y = ResLay<LC>;
            const double* P = res_pp(R, 0);
            const int a0 = R.m * Y::N;
            for (int j = threadIdx.x; j < Y::N; j += kThreads) {
                const double v = P[Y::off(j)];
                U[(size_t)a * N + a0 + j] = v;
                A.phi0[L0.off + addr(L0, a0 + j)] = v;
            }
            if (R.m == kResG - 1 && threadIdx.x == 0) {
                const double v = P[Y::off(Y::N)];
                U[(size_t)a * N + N - 1] = v;
                A.phi0[L0.off + addr(L0, N - 1)] = v;
            }
        }))
    } else if (threadIdx.x == 0) {
        if (vcycles) vcycles[a] = (int)c.vcycles;
        if (errs) errs[a] = err;
        if (total_vcycles) atomicAdd(total_vcycles, (unsigned long long)c.vcycles);
    }
#ifdef DFTA_POISSON_RPROF
    __syncthreads();
    if (a == 0 && (R.role == 1 || R.m == 0) && threadIdx.x < 64) g_rprof[R.role * 64 + threadIdx.x] += prof_acc[threadIdx.x];
#endif
}

// SolvePoissonNonUniform (PoissonSolver.h:51-81): one block per atom
__global__ __launch_bounds__(kThreads) void k_poisson_solve(const MgDesc* __restrict__ Dp, double* __restrict__ phi0, double* __restrict__ phi1,
                                                            double* __restrict__ src, const int* __restrict__ Z,
                                                            const double* __restrict__ density, const double* __restrict__ r,
                                                            const double* __restrict__ psrc, double* __restrict__ U,
                                                            int* __restrict__ vcycles, double* __restrict__ errs,
                                                            unsigned long long* __restrict__ total_vcycles,
                                                            unsigned* __restrict__ group_ctr, double* __restrict__ group_part,
                                                            const int* __restrict__ skip, int fault, int src_all)
{
    __shared__ double red[20];
    __shared__ double seqmem[3 * kSeqCap];
    __shared__ double stagemem[2 * kStageArr];
    const MgDesc& D = *Dp;
    // consecutive blocks are the members of one group (they land on different XCDs, where the barrier is cheapest)
    const int a = blockIdx.x >> D.logG;
    if (skip && skip[a]) return;                                  // a frozen atom of the batch (finished SCF): U stays as it is
    if (fault && D.G > 1 && (blockIdx.x & (D.G - 1)) == D.G - 1) return;   // fault injection (tests): this member never arrives
    Atom A;
    A.phi0 = phi0 + (size_t)a * D.per_atom;
    A.phi1 = phi1 + (size_t)a * D.per_atom;
    A.src = src + (size_t)a * D.per_atom;
    A.lds = seqmem;
    A.stage = stagemem;
    A.cur = 0;
    A.G = D.G;
    A.g = blockIdx.x & (D.G - 1);
    A.ctr = group_ctr + a;
    A.bar = 0;
    A.part = group_part + (size_t)a * group_part_doubles(D.G);
    A.fslot = A.part + 6 * D.G + 2;
    A.xchg = A.part + 9 * D.G + 2;
    A.fseq = 0;
    A.pend = 0;
    A.pend_r = 0;
    A.pend_z = 0;
    A.spin_max = D.spin_max;
    A.gave_up = false;
    const Lvl L0 = D.lv[0];
    const int N = L0.n;
    const double* rho = density + (size_t)a * N;
    const bool coop0 = D.kcoop > 0;        // level 0 is shared by the group
    // source: Source[i] = r_i; Source[i] *= (4 pi Rp^2 delta^2) exp(2 i delta) * density[i], 1 <= i <= N-2
    if (coop0 || A.g == 0)
    for (int idx = coop0 ? A.lane() : static_cast<int>(threadIdx.x); idx < N; idx += coop0 ? kThreads * A.G : kThreads) {
        const int i = node_of(L0, idx);
        double s = r[i];
        if (src_all || (i > 0 && i < N - 1)) s *= psrc[i] * rho[i];      // PoissonSolver.h:72-74; uniform grid: every node (PoissonSolver.h:39-40)
        A.src[L0.off + idx] = s;   // level 0 is never sequential: plain global storage
    }
    __syncthreads();
    Counters c{0, 0};
    PROF_T0();
    initialize(D, A, 0.0, (double)Z[a]);
    const double err = run_cycles<true>(D, A, 0, 100, 1E-3, 1E-14, red, c);      // FullCycle(1E-3, 1E-14), PoissonSolver.h:78
    PROF_ADD(5, 21);
    const double* __restrict__ P = A.cur_phi(0, L0);
    if (coop0 || A.g == 0)
        for (int i = coop0 ? A.lane() : static_cast<int>(threadIdx.x); i < N; i += coop0 ? kThreads * A.G : kThreads)
            U[(size_t)a * N + i] = P[addr(L0, i)];
    if (threadIdx.x == 0 && A.g == 0) {
        if (vcycles) vcycles[a] = (int)c.vcycles;
        if (errs) errs[a] = err;
        if (total_vcycles) atomicAdd(total_vcycles, (unsigned long long)c.vcycles);
    }
}

// unit-parity kernels on atom 0 ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_unit(const MgDesc* __restrict__ Dp, double* phi0, double* phi1, double* src, int* cur, int op,
                                                   int lvl, int sweeps, double* out, unsigned* group_ctr, double* group_part)
{
    __shared__ double red[20];
    __shared__ double seqmem[3 * kSeqCap];
    __shared__ double stagemem[2 * kStageArr];
    const MgDesc& D = *Dp;
    Atom A;
    A.phi0 = phi0; A.phi1 = phi1; A.src = src; A.lds = seqmem;
    A.stage = stagemem;
    A.G = D.G;
    A.g = blockIdx.x;
    A.ctr = group_ctr;
    A.bar = 0;
    A.part = group_part;
    A.fslot = A.part + 6 * D.G + 2;
    A.xchg = A.part + 9 * D.G + 2;
    A.fseq = 0;
    A.pend = 0;
    A.pend_r = 0;
    A.pend_z = 0;
    A.spin_max = D.spin_max;
    A.gave_up = false;
    A.cur = 0;
    for (int l = 0; l < D.levels; ++l) A.cur |= (cur[l] ? 1u : 0u) << l;
    // sequential levels: global -> LDS (the solve kernel initialises them itself); they are workgroup 0's
    for (int l = 0; l < D.levels && A.g == 0; ++l) {
        const Lvl L = D.lv[l];
        if (!L.seq) continue;
        for (int idx = threadIdx.x; idx < L.n; idx += kThreads) {
            seqmem[L.soff + idx] = phi0[L.off + idx];
            seqmem[kSeqCap + L.soff + idx] = phi1[L.off + idx];
            seqmem[2 * kSeqCap + L.soff + idx] = src[L.off + idx];
        }
    }
    __syncthreads();
    Counters c{0, 0};
    const bool lead = A.g == 0;
    if (op == 0) {
        if (lvl < D.kcoop || lead)
            for (int s = 0; s < sweeps; ++s) {
                const double e = gauss_seidel(D, A, lvl, red);
                if (threadIdx.x == 0 && lead) out[s] = e;
            }
    } else if (op == 1) do_restrict(D, A, lvl);
    else if (op == 2) do_prolong(D, A, lvl);
    else if (op == 4) {                                   // IterateGaussSeidel(lvl, errorMin = out[0], iterno = sweeps)
        const double emin = out[0];
        group_sync(A);                                    // everybody has read out[0]
        const double e = do_iterate(D, A, lvl, emin, sweeps, red, &c.sweeps);
        if (threadIdx.x == 0 && lead) { out[0] = e; out[1] = (double)c.sweeps; }
    }
    else if (op == 5) {                                   // FullCycle(errorMin = out[0], errorMinLast = out[1]) with boundaries out[2], out[3]
        const double e1 = out[0], e2 = out[1], lowB = out[2], highB = out[3];
        group_sync(A);                                    // everybody has read out[]
        initialize(D, A, lowB, highB);
        const double e = run_cycles<true>(D, A, 0, 100, e1, e2, red, c);
        if (threadIdx.x == 0 && lead) { out[0] = e; out[1] = (double)c.vcycles; }
    }
    else if (op == 3) {
        const int nramp = D.levels - 2 > 0 ? D.levels - 2 : 0;
        const double e = run_cycles<false>(D, A, 2 * nramp + 2, 1, 1E-14, 1E-14, red, c);   // one VCycle(last, 1E-14, 3)
        if (threadIdx.x == 0 && lead) out[0] = e;
    }
    __syncthreads();
    for (int l = 0; l < D.levels && lead; ++l) {
        const Lvl L = D.lv[l];
        if (!L.seq) continue;
        for (int idx = threadIdx.x; idx < L.n; idx += kThreads) {
            phi0[L.off + idx] = seqmem[L.soff + idx];
            phi1[L.off + idx] = seqmem[kSeqCap + L.soff + idx];
            src[L.off + idx] = seqmem[2 * kSeqCap + L.soff + idx];
        }
    }
    if (threadIdx.x == 0 && lead) for (int l = 0; l < D.levels; ++l) cur[l] = (A.cur >> l) & 1u;
}

}  // namespace

struct dfta_poisson {
    dfta_ctx* ctx = nullptr;
    const dfta_grid* g = nullptr;
    int batch = 0;
    MgDesc D;
    MgDesc* d_desc = nullptr;       // device copy of D (read with scalar loads)
    double *d_phi0 = nullptr, *d_phi1 = nullptr, *d_src = nullptr;
    int* d_cur = nullptr;           // unit hooks: current buffer per level (atom 0)
    std::vector<int> h_cur;
    unsigned long long* d_total_vcycles = nullptr;
    unsigned* d_group_ctr = nullptr;    // per atom: arrival counter of its group of workgroups (zeroed before every launch)
    double* d_group_part = nullptr;     // per atom: 6 G + 2 doubles (partial sums of the members, published state)
    // Groups of workgroups wait for each other, so every workgroup of a launch has to be resident: the launch is a
    // COOPERATIVE one (the runtime refuses it when the grid cannot be co-resident), the barriers' spins are bounded, and
    // dfta_poisson_finish() inspects the abort flag after every solve.  If a launch is refused or a group gives up, the
    // solve is repeated by `fallback` -- the same solver with one workgroup per atom (no cross-workgroup waits, results
    // bit-identical) -- and this solver stays degraded to it.
    dfta_poisson* fallback = nullptr;
    bool degraded = false;
    int aborts = 0;                 // solves that had to be repeated
    int fault = 0;                  // $DFTA_FAULT_POISSON_MEMBER (tests): the last member of every group never arrives
    bool plain_launch = false;      // groups started with an ordinary launch instead of a cooperative one (profilers, see poisson_create_impl)
    bool resident = false;          // k_poisson_solve_res: kResWG workgroups per atom, the shared levels live in the members' LDS
    double* d_res_slots = nullptr;  // per atom: res_slot_doubles() exchange slots (sentinel-filled before every launch)
    bool grouped() const { return D.G > 1 || resident; }
};

static int poisson_create_impl(dfta_ctx* ctx, const dfta_grid* g, int batch, int force_logG, dfta_poisson** out);

static long host_addr(const Lvl& L, int i)
{
    if (i == L.n - 1) return L.off + (L.n - 1);
    return L.off + ((long)(i & ((1 << L.logC) - 1)) << L.logT) + (i >> L.logC);
}

static int degrade(dfta_poisson* p)
{
    if (!p->fallback) {
        int rc = poisson_create_impl(p->ctx, p->g, p->batch, 0, &p->fallback);
        if (rc) return rc;
    }
    p->degraded = true;
    return DFTA_OK;
}

// dSkip (device, per atom, may be null): atoms with a non-zero entry are left untouched (frozen atoms of an SCF batch)
int dfta_poisson_solve_launch(dfta_poisson* p, const int* dZ, const double* dDensity, double* dU, int* dVcycles, double* dErr,
                              const int* dSkip)
{
    dfta_ctx* ctx = p->ctx;
    if (p->degraded) return dfta_poisson_solve_launch(p->fallback, dZ, dDensity, dU, dVcycles, dErr, dSkip);
    DFTA_HIP(ctx, hipMemsetAsync(p->d_group_ctr, 0, sizeof(unsigned) * p->batch, ctx->stream));
    DFTA_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p->d_group_part), 0x7FF8DEAD, (size_t)p->batch * group_part_doubles(p->D.G) * 2, ctx->stream));   // group_sum_fast's sentinel
    if (p->resident) {
        DFTA_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p->d_res_slots), 0x7FF8DEAD, (size_t)p->batch * res_slot_doubles() * 2, ctx->stream));
        const MgDesc* a0 = p->d_desc;
        const double *a_r = p->g->d_rsrc, *a_psrc = p->g->d_psrc;
        int fault = p->fault, src_all = p->g->uniform;
        if (p->plain_launch) {
            hipLaunchKernelGGL(k_poisson_solve_res, dim3(p->batch * kResWG), dim3(kThreads), 0, ctx->stream, a0, p->d_phi0, p->d_phi1, p->d_src, dZ,
                               dDensity, a_r, a_psrc, dU, dVcycles, dErr, p->d_total_vcycles, p->d_group_ctr, p->d_res_slots, dSkip, fault, src_all);
            DFTA_CHECK_LAUNCH(ctx);
            return DFTA_OK;
        }
        void* args[] = {&a0, &p->d_phi0, &p->d_phi1, &p->d_src, &dZ, &dDensity, &a_r, &a_psrc, &dU, &dVcycles, &dErr, &p->d_total_vcycles,
                        &p->d_group_ctr, &p->d_res_slots, &dSkip, &fault, &src_all};
        const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_poisson_solve_res), dim3(p->batch * kResWG), dim3(kThreads),
                                                        args, 0, ctx->stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            int rc = degrade(p);
            if (rc) return rc;
            ++p->aborts;
            return dfta_poisson_solve_launch(p->fallback, dZ, dDensity, dU, dVcycles, dErr, dSkip);
        }
        return DFTA_OK;
    }
    if (p->D.G == 1) {
        hipLaunchKernelGGL(k_poisson_solve, dim3(p->batch), dim3(kThreads), 0, ctx->stream, p->d_desc, p->d_phi0, p->d_phi1, p->d_src, dZ,
                           dDensity, p->g->d_rsrc, p->g->d_psrc, dU, dVcycles, dErr, p->d_total_vcycles, p->d_group_ctr, p->d_group_part,
                           dSkip, 0, p->g->uniform);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    if (p->plain_launch) {           // under a profiler (see poisson_create_impl): same kernel, ordinary launch
        hipLaunchKernelGGL(k_poisson_solve, dim3(p->batch * p->D.G), dim3(kThreads), 0, ctx->stream, p->d_desc, p->d_phi0, p->d_phi1, p->d_src, dZ,
                           dDensity, p->g->d_rsrc, p->g->d_psrc, dU, dVcycles, dErr, p->d_total_vcycles, p->d_group_ctr, p->d_group_part,
                           dSkip, p->fault, p->g->uniform);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    const MgDesc* a0 = p->d_desc;
    const double *a_r = p->g->d_rsrc, *a_psrc = p->g->d_psrc;
    int fault = p->fault, src_all = p->g->uniform;
    void* args[] = {&a0, &p->d_phi0, &p->d_phi1, &p->d_src, &dZ, &dDensity, &a_r, &a_psrc, &dU, &dVcycles, &dErr, &p->d_total_vcycles,
                    &p->d_group_ctr, &p->d_group_part, &dSkip, &fault, &src_all};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_poisson_solve), dim3(p->batch * p->D.G), dim3(kThreads),
                                                    args, 0, ctx->stream);
    if (e != hipSuccess) {
        // the grid cannot be co-resident right now (or cooperative launches are unavailable): one workgroup per atom instead
        (void)hipGetLastError();
        int rc = degrade(p);
        if (rc) return rc;
        ++p->aborts;
        return dfta_poisson_solve_launch(p->fallback, dZ, dDensity, dU, dVcycles, dErr, dSkip);
    }
    return DFTA_OK;
}

// after a solve has completed: did a group of workgroups give up on one of its barriers (a member was never scheduled)?
static int check_groups(dfta_poisson* p)
{
    dfta_ctx* ctx = p->ctx;
    if (!p->grouped() || p->degraded) return DFTA_OK;
    std::vector<unsigned> h(p->batch);
    DFTA_HIP(ctx, hipMemcpyAsync(h.data(), p->d_group_ctr, sizeof(unsigned) * p->batch, hipMemcpyDeviceToHost, ctx->stream));
    DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (unsigned v : h)
        if (v & 0x80000000u) {
            const int G = p->resident ? kResWG : p->D.G;
            snprintf(ctx->err, sizeof(ctx->err), "poisson: a group of %d workgroups lost a member at a barrier (the %d workgroups of "
                     "the launch were not all resident)", G, p->batch * G);
            return DFTA_ERR_HIP;
        }
    return DFTA_OK;
}

// Completes the solve launched last (synchronises the stream).  If a group of workgroups gave up, the solve is repeated
// with one workgroup per atom, in this process and on the same stream, and every later solve of `p` takes that path.
int dfta_poisson_finish(dfta_poisson* p, const int* dZ, const double* dDensity, double* dU, int* dVcycles, double* dErr,
                        const int* dSkip)
{
    dfta_ctx* ctx = p->ctx;
    if (!p->grouped() || p->degraded) { DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream)); return DFTA_OK; }
    if (check_groups(p) == DFTA_OK) return DFTA_OK;
    ++p->aborts;
    DFTA_HIP(ctx, hipMemsetAsync(p->d_total_vcycles, 0, sizeof(unsigned long long), ctx->stream));   // the aborted solve's count
    int rc = degrade(p);
    if (rc) return rc;
    rc = dfta_poisson_solve_launch(p->fallback, dZ, dDensity, dU, dVcycles, dErr, dSkip);
    if (rc) return rc;
    DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DFTA_OK;
}

int dfta_poisson_take_vcycles(dfta_poisson* p, unsigned long long* out)   // reads and clears the V-cycle counter
{
    dfta_ctx* ctx = p->ctx;
    unsigned long long a = 0, b = 0;
    DFTA_HIP(ctx, hipMemcpyAsync(&a, p->d_total_vcycles, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    if (p->fallback) DFTA_HIP(ctx, hipMemcpyAsync(&b, p->fallback->d_total_vcycles, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DFTA_HIP(ctx, hipMemsetAsync(p->d_total_vcycles, 0, sizeof(unsigned long long), ctx->stream));
    if (p->fallback) DFTA_HIP(ctx, hipMemsetAsync(p->fallback->d_total_vcycles, 0, sizeof(unsigned long long), ctx->stream));
    *out = a + b;
    return DFTA_OK;
}

int dfta_poisson_group_state(const dfta_poisson* p, int* G, int* degraded, int* aborts)
{
    if (!p) return DFTA_ERR_INVALID;
    if (G) *G = p->resident ? kResWG : p->D.G;
    if (degraded) *degraded = p->degraded ? 1 : 0;
    if (aborts) *aborts = p->aborts;
    return DFTA_OK;
}

extern "C" {

int dfta_poisson_create(dfta_ctx* ctx, const dfta_grid* g, int batch, dfta_poisson** out)
{
    if (!ctx || !g || !out) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    return poisson_create_impl(ctx, g, batch, -1, out);
}

}  // extern "C"

// force_logG >= 0: that many doublings of the workgroups per atom (0: one workgroup per atom); -1: chosen from the batch size
static int poisson_create_impl(dfta_ctx* ctx, const dfta_grid* g, int batch, int force_logG, dfta_poisson** out)
{
    DFTA_REQUIRE(ctx, batch >= 1 && g->levels <= kMaxLevels, "poisson batch/levels");
    dfta_poisson* p = new dfta_poisson();
    p->ctx = ctx; p->g = g; p->batch = batch;
    MgDesc& D = p->D;
    D.levels = g->levels;
    // Workgroups per atom: a solve is bound by ONE compute unit's vector-memory path, so while the batch leaves compute
    // units idle the fine levels of every atom are shared by a group of G workgroups (all of them must be resident:
    // batch * G <= 256 CUs).  A level is shared when every lane of the group still owns >= 8 nodes (>= 4 for G = 16: the
    // same four levels at 131073 nodes, with 32 nodes per lane on the finest one -- the most that is staged in LDS).
    // (measured: G = 16 wins up to 4 atoms; beyond that the barriers of 16 members on a nearly full chip cost more than the
    // shorter chunks save)
    int logG = batch <= 4 ? 4 : (batch <= 32 ? 3 : (batch <= 64 ? 2 : (batch <= 128 ? 1 : 0)));
    if (const char* e = getenv("DFTA_POISSON_GROUP")) {      // measurements: force log2 of the group size
        const int v = atoi(e);
        if (v >= 0 && v <= 4 && (batch << v) <= 256) logG = v;
    }
    if (force_logG >= 0) logG = force_logG;
    if (const char* e = getenv("DFTA_FAULT_POISSON_MEMBER")) p->fault = atoi(e) != 0;
    // Resident group (k_poisson_solve_res): where the batch would get 16 workgroups per atom and level 0 gives every lane of
    // kResG x kResNT lanes 4 .. 32 nodes (16385 .. 131073 nodes); $DFTA_POISSON_RES = 0 / 1 switches it off / on (for batches
    // up to 7 atoms), a forced group size (DFTA_POISSON_GROUP, force_logG) selects the staged groups above
    int res_kres = 0, res_logC0 = 0;
    {
        bool want = logG == 4 && force_logG < 0 && !getenv("DFTA_POISSON_GROUP");
        if (const char* e = getenv("DFTA_POISSON_RES")) want = atoi(e) != 0 && force_logG < 0 && batch * kResWG <= 256;
        const int lanes = kResG * kResNT;
        if (want && (g->N - 1) % lanes == 0) {
            const int C0 = (g->N - 1) / lanes;
            int lc = 0;
            while ((1 << lc) < C0) ++lc;
            if ((1 << lc) == C0 && lc >= 2 && lc <= 5 && g->levels >= lc + 4) { res_logC0 = lc; res_kres = lc - 1; }
        }
        if (res_kres > 0) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_poisson_solve_res, kThreads, 0) != hipSuccess) per_cu = 0;
            if (batch * kResWG > per_cu * ctx->num_cu) res_kres = 0;
        }
        if (res_kres > 0) { logG = 0; p->resident = true; }
    }
    // rocprofiler-sdk (ROCm 7.2) crashes in an exit handler of a process that has made a cooperative launch -- after its
    // output is written, but the profiled command returns 139.  Under the profiler (rocprofv3 exports ROCP_TOOL_LIBRARIES), or
    // when DFTA_POISSON_PLAIN_LAUNCH is set, the groups are therefore started with an ordinary launch: same kernel, same
    // results and timing; co-residency then rests on the occupancy query of this function, the bounded spins and the abort
    // flag (dfta_poisson_finish) as in round 1.
    p->plain_launch = getenv("DFTA_POISSON_PLAIN_LAUNCH") != nullptr || getenv("ROCP_TOOL_LIBRARIES") != nullptr;
    D.spin_max = p->fault ? (1 << 12) : (1 << 23);
    {
        // every workgroup of the launch must be resident at once (the members wait for each other)
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_poisson_solve, kThreads, 0) != hipSuccess) per_cu = 1;
        while (logG > 0 && (batch << logG) > per_cu * ctx->num_cu) --logG;
    }
    D.kcoop = 0;
    {
        int n = g->N;
        for (int l = 0; l < D.levels; ++l, n = (n + 1) / 2)
            if (logG > 0 && (n - 1) >= (kThreads << logG) * (logG >= 4 ? 4 : 8)) D.kcoop = l + 1;
    }
    if (D.kcoop == 0) logG = 0;
    D.logG = logG;
    D.nofold = getenv("DFTA_POISSON_NOFOLD") ? 1 : 0;
    D.fuse3 = getenv("DFTA_POISSON_NOFUSE3") ? 0 : 1;
    D.dbg = getenv("DFTA_POISSON_DBG") ? atoi(getenv("DFTA_POISSON_DBG")) : 0;
    D.res_kres = res_kres;
    D.res_logC0 = res_logC0;
    D.G = 1 << logG;
    long off = kPad, soff = 0;
    double d = g->delta;                       // PoissonSolver.cpp:21-26 (0 on a uniform grid: PoissonSolver(levels), DFTAtom.cpp:89)
    int n = g->N;                              // finest level first
    for (int l = 0; l < D.levels; ++l) {
        Lvl& L = D.lv[l];
        L.n = n; L.off = off; L.d = d;
        int lg = 0;
        while ((1 << lg) < n - 1) ++lg;        // n - 1 == 2^lg
        L.stage = 0;
        if (n < kSeqBelow) { L.seq = 1; L.logT = 0; L.logC = lg; L.soff = soff; soff += n; }
        else {
            L.seq = 0; L.logT = std::min(lg, l < D.kcoop ? 8 + logG : 8); L.logC = lg - L.logT; L.soff = -1;
            if (!getenv("DFTA_POISSON_NOSTAGE")) {
                if (l >= D.kcoop && n <= kWaveMaxN && n >= 129 && !getenv("DFTA_POISSON_NOSTAGE_WAVE")) L.stage = 3;
                else if (l >= D.kcoop && L.logT == 8 && L.logC <= kStageMaxLogC) L.stage = 1;
                else if (l < D.kcoop && D.G > 1 && L.logT == 8 + logG && L.logC >= 2 && L.logC <= kStageMaxLogC &&
                         !getenv("DFTA_POISSON_NOSTAGE_SHARED")) L.stage = 2;
            }
        }
        off += n;
        n = (n + 1) / 2;
        d *= 2;
    }
    D.per_atom = off;
    // coarse section: from the first one-wave level down, if everything below is one-wave or sequential and fits the staging memory
    D.cs_top = -1;
    if (!getenv("DFTA_POISSON_NOCOARSE")) {
        int top = -1;
        for (int l = 1; l < D.levels; ++l)
            if (D.lv[l].stage == 3) { top = l; break; }
        bool ok = top >= 1 && top >= D.kcoop + 1 && top <= D.levels - 2;
        int at = 0;
        for (int l = top; ok && l < D.levels; ++l) {
            const Lvl& L = D.lv[l];
            if (L.stage == 3) {
                int lc = 0;
                while ((64 << lc) < L.n - 1) ++lc;                 // n - 1 == 64 * 2^lc
                if (lc < 1 || lc > 4) { ok = false; break; }
                D.cs_lc[l] = lc;
                D.cs_phi[l] = at + kStagePad; at += kStagePad + L.n + 8;
                D.cs_src[l] = at + kStagePad; at += kStagePad + L.n + 8;
            } else if (L.seq) {
                D.cs_lc[l] = -1;
                D.cs_phi[l] = at; at += L.n + 1;
                D.cs_src[l] = at; at += L.n + 1;
            } else ok = false;
        }
        if (ok && at <= 2 * kStageArr) D.cs_top = top;
    }
    if (soff > kSeqCap) { delete p; snprintf(ctx->err, sizeof(ctx->err), "sequential levels exceed LDS budget"); return DFTA_ERR_INVALID; }
    const size_t tot = (size_t)off * batch;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->d_phi0), tot * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_phi1), tot * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_src), tot * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_cur), kMaxLevels * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_desc), sizeof(MgDesc));
    if (e == hipSuccess) e = hipMemcpyAsync(p->d_desc, &p->D, sizeof(MgDesc), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_total_vcycles), sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_group_ctr), sizeof(unsigned) * batch);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->d_group_part), sizeof(double) * (size_t)batch * group_part_doubles(D.G));
    if (e == hipSuccess && p->resident) e = hipMalloc(reinterpret_cast<void**>(&p->d_res_slots), sizeof(double) * (size_t)batch * res_slot_doubles());
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

extern "C" {

void dfta_poisson_destroy(dfta_poisson* p)
{
    if (!p) return;
    if (p->fallback) dfta_poisson_destroy(p->fallback);
#ifdef DFTA_POISSON_PROF
    {
        unsigned long long h[8 * 24];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof(h)) == hipSuccess) {
            const char* names[8] = {"restrict", "prolong ", "iterate ", "grp_sum ", "gs_lds  ", "copy_in ", "copy_out", "sweeps1 "};
            unsigned long long tot = 0;
            for (int c = 0; c < 8; ++c) {
                fprintf(stderr, "[poisson prof] %s:", names[c]);
                for (int l = 0; l < 22; ++l) { fprintf(stderr, " %llu", h[c * 24 + l]); tot += h[c * 24 + l]; }
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "[poisson prof] total ticks %llu\n", tot);
            unsigned long long hm[64 * 4];
            if (hipMemcpyFromSymbol(hm, HIP_SYMBOL(g_prof_member), sizeof(hm)) == hipSuccess) {
                for (int m = 0; m < 16; ++m) fprintf(stderr, "[poisson prof] member %2d: exchange %llu  sweeps %llu  copy-in %llu\n", m, hm[m * 4], hm[m * 4 + 1], hm[m * 4 + 2]);
                unsigned long long zz[64 * 4] = {0};
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_member), zz, sizeof(zz));
            }
            unsigned long long z[8 * 24] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
        }
    }
#endif
#ifdef DFTA_POISSON_RPROF
    {
        unsigned long long hr[2 * 8 * 8];
        if (p->resident && hipMemcpyFromSymbol(hr, HIP_SYMBOL(g_rprof), sizeof(hr)) == hipSuccess) {
            const char* mn[8] = {"pass    ", "publish ", "exchange", "commit  ", "restrict", "prolong ", "handover", "redo    "};
            const char* cn[8] = {"passive ", "-       ", "iterate ", "coarsesc", "restrict", "prolong ", "handover", "-       "};
            for (int role = 0; role < 2; ++role)
                for (int c = 0; c < 8; ++c) {
                    unsigned long long t = 0;
                    fprintf(stderr, "[res prof] %s %s:", role ? "coarse wg" : "member 0 ", role ? cn[c] : mn[c]);
                    for (int l = 0; l < 8; ++l) { fprintf(stderr, " %llu", hr[(role * 8 + c) * 8 + l]); t += hr[(role * 8 + c) * 8 + l]; }
                    fprintf(stderr, "  = %llu\n", t);
                }
            unsigned long long zz[2 * 8 * 8] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rprof), zz, sizeof(zz));
        }
    }
#endif
    void* ptrs[] = {p->d_phi0, p->d_phi1, p->d_src, p->d_cur, p->d_total_vcycles, p->d_desc, p->d_group_ctr, p->d_group_part, p->d_res_slots};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

int dfta_poisson_solve(dfta_poisson* p, const int* Z, const double* density, double* U, int* vcycles_out, double* err_out)
{
    if (!p) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = p->ctx;
    DFTA_ENTER(ctx);
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
    int rc = dfta_poisson_solve_launch(p, dZ.p, dRho.p, dU.p, dVc.p, dErr.p, nullptr);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    ctx->have_kernel_time = true;
    rc = dfta_poisson_finish(p, dZ.p, dRho.p, dU.p, dVc.p, dErr.p, nullptr);
    if (rc) return rc;
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
    DFTA_ENTER(p->ctx);
    // synchronises: the group barriers' abort flag is inspected after every solve (and the solve repeated with one
    // workgroup per atom if it was raised), so a DFTA_OK always means a completed solve
    int rc = dfta_poisson_solve_launch(p, dZ, dDensity, dU, nullptr, nullptr, nullptr);
    if (rc) return rc;
    return dfta_poisson_finish(p, dZ, dDensity, dU, nullptr, nullptr, nullptr);
}

int dfta_poisson_group_info(const dfta_poisson* p, int* G, int* degraded, int* aborts)
{
    return dfta_poisson_group_state(p, G, degraded, aborts);
}

int dfta_poisson_level_size(const dfta_poisson* p, int lvl)
{
    if (!p || lvl < 0 || lvl >= p->D.levels) return -1;
    return p->D.lv[lvl].n;
}

int dfta_poisson_set_level(dfta_poisson* p, int lvl, const double* Phi, const double* Src)
{
    if (!p) return DFTA_ERR_INVALID;
    if (p->degraded) return dfta_poisson_set_level(p->fallback, lvl, Phi, Src);     // the solves run on the fallback's storage
    dfta_ctx* ctx = p->ctx;
    DFTA_ENTER(ctx);
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
    if (p->degraded) return dfta_poisson_get_level(p->fallback, lvl, Phi, Src);
    dfta_ctx* ctx = p->ctx;
    DFTA_ENTER(ctx);
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

// the unit hooks run on atom 0 with the solver's own group of G workgroups (cooperative launch, like the solve)
static int launch_unit(dfta_poisson* p, int op, int lvl, int sweeps, double* dOut)
{
    dfta_ctx* ctx = p->ctx;
    if (p->D.G == 1) {
        hipLaunchKernelGGL(k_unit, dim3(1), dim3(kThreads), 0, ctx->stream, p->d_desc, p->d_phi0, p->d_phi1, p->d_src, p->d_cur, op, lvl, sweeps,
                           dOut, p->d_group_ctr, p->d_group_part);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    if (p->plain_launch) {
        hipLaunchKernelGGL(k_unit, dim3(p->D.G), dim3(kThreads), 0, ctx->stream, p->d_desc, p->d_phi0, p->d_phi1, p->d_src, p->d_cur, op, lvl, sweeps,
                           dOut, p->d_group_ctr, p->d_group_part);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    const MgDesc* a0 = p->d_desc;
    void* args[] = {&a0, &p->d_phi0, &p->d_phi1, &p->d_src, &p->d_cur, &op, &lvl, &sweeps, &dOut, &p->d_group_ctr, &p->d_group_part};
    DFTA_HIP(ctx, hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_unit), dim3(p->D.G), dim3(kThreads), args, 0, ctx->stream));
    return DFTA_OK;
}

static int unit_op(dfta_poisson* p, int op, int lvl, int sweeps, double* out_host, int nout)
{
    if (p->degraded) return unit_op(p->fallback, op, lvl, sweeps, out_host, nout);
    dfta_ctx* ctx = p->ctx;
    DFTA_ENTER(ctx);
    hipStream_t st = ctx->stream;
    DevBuf<double> dOut;
    DFTA_HIP(ctx, dOut.alloc(std::max(nout, 1)));
    DFTA_HIP(ctx, hipMemcpyAsync(p->d_cur, p->h_cur.data(), sizeof(int) * kMaxLevels, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemsetAsync(p->d_group_ctr, 0, sizeof(unsigned), st));
    DFTA_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p->d_group_part), 0x7FF8DEAD, (size_t)group_part_doubles(p->D.G) * 2, st));
    if (int rc = launch_unit(p, op, lvl, sweeps, dOut.p)) return rc;
    DFTA_HIP(ctx, hipMemcpyAsync(p->h_cur.data(), p->d_cur, sizeof(int) * kMaxLevels, hipMemcpyDeviceToHost, st));
    if (out_host && nout > 0) DFTA_HIP(ctx, hipMemcpyAsync(out_host, dOut.p, sizeof(double) * nout, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return check_groups(p);          // a member lost at a barrier of a unit launch is an error, not a silent wrong answer
}

int dfta_poisson_gauss_seidel(dfta_poisson* p, int lvl, int sweeps, double* err_out)
{
    if (!p) return DFTA_ERR_INVALID;
    DFTA_REQUIRE(p->ctx, lvl >= 0 && lvl < p->D.levels && sweeps >= 1 && sweeps <= 1024, "level/sweeps");
    return unit_op(p, 0, lvl, sweeps, err_out, err_out ? sweeps : 0);
}
int dfta_poisson_iterate_gs(dfta_poisson* p, int lvl, double errorMin, int iterno, double* err_out, int* sweeps_out)
{
    if (!p) return DFTA_ERR_INVALID;
    if (p->degraded) return dfta_poisson_iterate_gs(p->fallback, lvl, errorMin, iterno, err_out, sweeps_out);
    dfta_ctx* ctx = p->ctx;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, lvl >= 0 && lvl < p->D.levels && iterno >= 1 && iterno <= 1024, "level/iterno");
    hipStream_t st = ctx->stream;
    DevBuf<double> dOut;
    DFTA_HIP(ctx, dOut.alloc(2));
    DFTA_HIP(ctx, hipMemcpyAsync(dOut.p, &errorMin, sizeof(double), hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(p->d_cur, p->h_cur.data(), sizeof(int) * kMaxLevels, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemsetAsync(p->d_group_ctr, 0, sizeof(unsigned), st));
    DFTA_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p->d_group_part), 0x7FF8DEAD, (size_t)group_part_doubles(p->D.G) * 2, st));
    if (int rc = launch_unit(p, 4, lvl, iterno, dOut.p)) return rc;
    double out[2] = {0, 0};
    DFTA_HIP(ctx, hipMemcpyAsync(p->h_cur.data(), p->d_cur, sizeof(int) * kMaxLevels, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipMemcpyAsync(out, dOut.p, sizeof(out), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    if (err_out) *err_out = out[0];
    if (sweeps_out) *sweeps_out = (int)out[1];
    return check_groups(p);
}
// PoissonSolver::FullCycle (PoissonSolver.h:89-124) on atom 0's level storage: Initialize from the level-0 source that
// the last solve (or dfta_poisson_set_level) left there and from the boundary values, FMG ramp, up to 100 V-cycles
int dfta_poisson_full_cycle(dfta_poisson* p, double lowBoundary, double highBoundary, double errorMin, double errorMinLast,
                            double* err_out, int* vcycles_out)
{
    if (!p) return DFTA_ERR_INVALID;
    if (p->degraded) return dfta_poisson_full_cycle(p->fallback, lowBoundary, highBoundary, errorMin, errorMinLast, err_out, vcycles_out);
    dfta_ctx* ctx = p->ctx;
    DFTA_ENTER(ctx);
    hipStream_t st = ctx->stream;
    DevBuf<double> dOut;
    DFTA_HIP(ctx, dOut.alloc(4));
    double io[4] = {errorMin, errorMinLast, lowBoundary, highBoundary};
    DFTA_HIP(ctx, hipMemcpyAsync(dOut.p, io, sizeof(io), hipMemcpyHostToDevice, st));
    std::fill(p->h_cur.begin(), p->h_cur.end(), 0);       // Initialize starts from copy 0 of every level
    DFTA_HIP(ctx, hipMemcpyAsync(p->d_cur, p->h_cur.data(), sizeof(int) * kMaxLevels, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemsetAsync(p->d_group_ctr, 0, sizeof(unsigned), st));
    DFTA_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p->d_group_part), 0x7FF8DEAD, (size_t)group_part_doubles(p->D.G) * 2, st));
    if (int rc = launch_unit(p, 5, 0, 0, dOut.p)) return rc;
    DFTA_HIP(ctx, hipMemcpyAsync(p->h_cur.data(), p->d_cur, sizeof(int) * kMaxLevels, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipMemcpyAsync(io, dOut.p, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    if (err_out) *err_out = io[0];
    if (vcycles_out) *vcycles_out = (int)io[1];
    return check_groups(p);
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
