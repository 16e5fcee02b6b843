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
// kWarm (the chunked sweeps' warm-up) and kWarm3 (the fused visit's) are set per inclusion of poisson_kernels.inc:
//   mg_exact: 96 / 112 nodes -- start-value error decays by <= 0.52^96 < 2^-90: chunked sweep == sequential sweep bit for bit
//   mg_tol  : 32 / 32 nodes  -- the opt-in tolerance mode (DFTA_POISSON_TOLERANCE): start values good to 1e-9 of a sweep's change
constexpr int kSeqBelow = 129;   // levels with n < 129 nodes: one lane, sequential, LDS-resident
constexpr int kWaveMaxN = 1025;  // staged levels up to this size are swept by the first wave alone (64 lanes: a quarter of the LDS traffic per warm-up step)
constexpr int kSeqCap = 144;     // LDS doubles per array for the sequential levels (65+33+17+9+5+3 = 132)
constexpr int kPF = 8;           // register prefetch depth of the chunked sweep
constexpr int kFuseMinLogC = 6;  // fuse the three sweeps of a visit of a global-memory level when every lane owns >= 64 nodes (knob POISSON_FUSE_MIN_LOGC)
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
    int fuse_min_logc;   // global-memory levels of one workgroup: fused visits (gs_fused3) from this many nodes per lane on
    int fuse_coop;       // ... and on the levels the G workgroups of an atom share (0: $DFTA_DEBUG POISSON_NOFUSE_COOP)
    int fuse3;    // visits of three sweeps on staged levels of one workgroup run as ONE fused pass (gs_lds3); 0: $DFTA_POISSON_NOFUSE3
    int fuse3w;   // ... and those of the one-wave levels with 257 .. 1025 nodes of the coarse section (cs_visit3); 0: POISSON_NOFUSE3 / POISSON_NOFUSE3_WAVE
    int nofold;   // DFTA_POISSON_NOFOLD: restriction / prolongation as separate passes even where they could be folded into a staged copy-in
    int fold_lds; // the folded restriction reads the finer level from the staging memory where its visit has just left it (POISSON_NOFOLD_LDS: from global)
    int kcoop;       // levels 0 .. kcoop-1 are swept by all G workgroups together (256 G lanes), the others by workgroup 0
    int spin_max;    // bound of the group barriers' spin loops (Atom::spin_max)
    long per_atom;   // doubles per atom and per array (sum of n)
    // Coarse section (coarse_section below): levels cs_top .. levels-1 of a V-cycle are handled by the first wave of
    // workgroup 0 alone, entirely in LDS.  -1: off.  cs_phi / cs_src: offsets of a level's arrays inside the staging memory
    // (doubles), cs_lc: log2(nodes per lane) of its 64-lane interleaved layout, or -1 for natural order.
    int cs_top;
    int xw_top;      // exact mode: levels xw_top .. levels-1 (65, 33, 17, 9, 5, 3 nodes) of the coarse section with their nodes in registers (xw_section); -1: off
    int rc_src[6];   // ... offsets (doubles, inside the staging memory) of the sources of levels rc_top .. rc_top + 5 (256 C entries each)
    int adaptive;    // DFTA_POISSON_ADAPTIVE: stop the V-cycles at the round-off floor (run_cycles / res_cycles)
    int rc_top;      // tolerance mode, resident groups: the coarse workgroup runs levels rc_top .. levels-1 of a V-cycle with their nodes in registers (coarse_resident_cycle); -1: off
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
// per atom: [6 G + 2] partial sums of the members and the published state, [kGrpBuf G] slots of the fast sum, [kGrpBuf G kXchg] boundary
// nodes exchanged between neighbours in the middle of a staged visit
// fast-sum slots and boundary-exchange buffers rotate over kGrpBuf sets (round 3: 8, was 3): a member resets its part of the set half a
// rotation away, so that a reset has several exchanges to land before the slot is used again (see kResBuf below)
constexpr unsigned kGrpBuf = 8;
__host__ __device__ constexpr size_t group_part_doubles(int G) { return (size_t)(6 + kGrpBuf) * G + 2 + (size_t)kGrpBuf * G * kXchg; }

constexpr int kResNT = 128;                    // sweeping lanes of a member (its first two waves; all four move data)
constexpr int kResG = 32;                      // members per atom
constexpr int kResWG = kResG + 1;              // + the coarse workgroup (participant kResG of every exchange)
constexpr int kResX = 272;                     // payload doubles per participant and buffer
constexpr int kResMaxShared = 4;
constexpr int kResXTail = 0, kResXSrc = 128, kResXHead = 256, kResXS0 = 266;
// Exchange buffers in rotation.  A slot holds a sentinel until its datum arrives; the owner resets its slots of buffer (s + kResBuf / 2)
// while exchange s completes -- a buffer nobody has touched for kResBuf / 2 exchanges and nobody will for as many.  (Three buffers, as
// in the staged groups above, leave one exchange between a reset and the slot's next use: an agent-scope store can overtake an
// earlier one on its way to another XCD, and a reader that still saw the datum of three exchanges ago took it for the new one --
// observed as rare run-to-run differences of the V-cycle count for He at 16385 nodes, where exchanges follow each other within 3 us.)
constexpr unsigned kResBuf = 16;
__host__ __device__ constexpr size_t res_slot_doubles() { return (size_t)kResBuf * kResWG * 4 + (size_t)kResBuf * kResWG * kResX; }

}  // namespace

namespace {
namespace mg_exact {
#define DFTA_MG_KWARM DFTA_KWARM
#define DFTA_MG_KWARM3 112
#include "poisson_kernels.inc"
#undef DFTA_MG_KWARM
#undef DFTA_MG_KWARM3
}  // namespace mg_exact
// The resident group's second configuration (exact mode): 16 members of 256 lanes + the coarse workgroup per atom, level 0 and the other
// shared levels taking turns in the members' LDS -- up to 15 atoms per launch (poisson_kernels.inc: DFTA_MG_RES16)
namespace mg_exact16 {
#define DFTA_MG_KWARM DFTA_KWARM
#define DFTA_MG_KWARM3 112
#define DFTA_MG_RES16 1
#include "poisson_kernels.inc"
#undef DFTA_MG_RES16
#undef DFTA_MG_KWARM
#undef DFTA_MG_KWARM3
}  // namespace mg_exact16
// Tolerance mode (opt-in, DFTA_POISSON_TOLERANCE): the same kernels with 32-node warm-ups.  A lane's start value then carries
// 0.52^32 ~ 1e-9 of the change its start node undergoes in that sweep -- a perturbation of the ITERATION, not of its fixed point:
// the cycle still converges to the solution of the same discrete equations, to the same round-off floor (tests: U within
// 2e-9 Z of the exact solve, SCF energies within 1e-9 of the reference's), but a sweep is no longer the sequential sweep bit for bit.
namespace mg_tol {
#define DFTA_MG_KWARM 32
#define DFTA_MG_KWARM3 32
#ifndef DFTA_MG_NO_SCAN_COARSE
#define DFTA_MG_SCAN_COARSE 1      // the coarse section's sweeps as affine scans (poisson_kernels.inc: cs_sweep_scan)
#endif
#include "poisson_kernels.inc"
#undef DFTA_MG_KWARM
#undef DFTA_MG_KWARM3
}  // namespace mg_tol
namespace mg_tol16 {               // tolerance mode, the resident group's second configuration (8 .. 15 atoms)
#define DFTA_MG_KWARM 32
#define DFTA_MG_KWARM3 32
#define DFTA_MG_RES16 1
#include "poisson_kernels.inc"
#undef DFTA_MG_RES16
#undef DFTA_MG_KWARM
#undef DFTA_MG_KWARM3
}  // namespace mg_tol16
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
    bool tol = false;               // tolerance mode: the kernels of namespace mg_tol (32-node warm-ups) instead of mg_exact
    bool adaptive = false;          // DFTA_POISSON_ADAPTIVE: tolerance mode + the V-cycles stop at the round-off floor
    bool resident = false;          // k_poisson_solve_res: kResWG workgroups per atom, the shared levels live in the members' LDS
    double* d_res_slots = nullptr;  // per atom: res_slot_doubles() exchange slots (sentinel-filled before every launch)
    bool res16 = false;             // resident, second configuration (mg_exact16): 17 workgroups per atom, level 0 spilled to d_res_spill in turns
    double* d_res_spill = nullptr;  // res16: per member the two LDS images of level 0
    int res_wg() const { return res16 ? mg_exact16::kResWG : kResWG; }
    bool grouped() const { return D.G > 1 || resident; }
};

static int poisson_create_impl(dfta_ctx* ctx, const dfta_grid* g, int batch, int force_logG, int mode, dfta_poisson** out);
#define K_SOLVE(p) ((p)->tol ? mg_tol::k_poisson_solve : mg_exact::k_poisson_solve)
#define K_SOLVE_RES(p) ((p)->res16 ? ((p)->tol ? mg_tol16::k_poisson_solve_res : mg_exact16::k_poisson_solve_res) \
                                   : ((p)->tol ? mg_tol::k_poisson_solve_res : mg_exact::k_poisson_solve_res))
#define K_UNIT(p) ((p)->tol ? mg_tol::k_unit : mg_exact::k_unit)

static long host_addr(const Lvl& L, int i)
{
    if (i == L.n - 1) return L.off + (L.n - 1);
    return L.off + ((long)(i & ((1 << L.logC) - 1)) << L.logT) + (i >> L.logC);
}

static int degrade(dfta_poisson* p)
{
    if (!p->fallback) {
        int rc = poisson_create_impl(p->ctx, p->g, p->batch, 0, dfta_poisson_mode(p), &p->fallback);
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
            hipLaunchKernelGGL(K_SOLVE_RES(p), dim3(p->batch * p->res_wg()), dim3(kThreads), 0, ctx->stream, a0, p->d_phi0, p->d_phi1, p->d_src, dZ,
                               dDensity, a_r, a_psrc, dU, dVcycles, dErr, p->d_total_vcycles, p->d_group_ctr, p->d_res_slots, dSkip, fault, src_all,
                               p->d_res_spill);
            DFTA_CHECK_LAUNCH(ctx);
            return DFTA_OK;
        }
        void* args[] = {&a0, &p->d_phi0, &p->d_phi1, &p->d_src, &dZ, &dDensity, &a_r, &a_psrc, &dU, &dVcycles, &dErr, &p->d_total_vcycles,
                        &p->d_group_ctr, &p->d_res_slots, &dSkip, &fault, &src_all, &p->d_res_spill};
        const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(K_SOLVE_RES(p)), dim3(p->batch * p->res_wg()), dim3(kThreads),
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
        hipLaunchKernelGGL(K_SOLVE(p), dim3(p->batch), dim3(kThreads), 0, ctx->stream, p->d_desc, p->d_phi0, p->d_phi1, p->d_src, dZ,
                           dDensity, p->g->d_rsrc, p->g->d_psrc, dU, dVcycles, dErr, p->d_total_vcycles, p->d_group_ctr, p->d_group_part,
                           dSkip, 0, p->g->uniform);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    if (p->plain_launch) {           // under a profiler (see poisson_create_impl): same kernel, ordinary launch
        hipLaunchKernelGGL(K_SOLVE(p), dim3(p->batch * p->D.G), dim3(kThreads), 0, ctx->stream, p->d_desc, p->d_phi0, p->d_phi1, p->d_src, dZ,
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
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(K_SOLVE(p)), dim3(p->batch * p->D.G), dim3(kThreads),
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
            const int G = p->resident ? p->res_wg() : p->D.G;
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
    if (G) *G = p->resident ? p->res_wg() : p->D.G;
    if (degraded) *degraded = p->degraded ? 1 : 0;
    if (aborts) *aborts = p->aborts;
    return DFTA_OK;
}

extern "C" {

int dfta_poisson_create_ex(dfta_ctx* ctx, const dfta_grid* g, int batch, int mode, dfta_poisson** out)
{
    if (!ctx || !g || !out) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, mode == DFTA_POISSON_EXACT || mode == DFTA_POISSON_TOLERANCE || mode == DFTA_POISSON_ADAPTIVE, "poisson mode");
    return poisson_create_impl(ctx, g, batch, -1, mode, out);
}

int dfta_poisson_create(dfta_ctx* ctx, const dfta_grid* g, int batch, dfta_poisson** out)
{
    // DFTA_DEBUG="POISSON_MODE=tolerance" (or adaptive): measurements and tests of the opt-in modes through callers that do not pass a mode
    const char* e = dfta_knob("POISSON_MODE");
    return dfta_poisson_create_ex(ctx, g, batch, (e && e[0] == 't') ? DFTA_POISSON_TOLERANCE : ((e && e[0] == 'a') ? DFTA_POISSON_ADAPTIVE : DFTA_POISSON_EXACT), out);
}

int dfta_poisson_mode(const dfta_poisson* p) { return p ? (p->adaptive ? DFTA_POISSON_ADAPTIVE : (p->tol ? DFTA_POISSON_TOLERANCE : DFTA_POISSON_EXACT)) : -1; }

}  // extern "C"

// force_logG >= 0: that many doublings of the workgroups per atom (0: one workgroup per atom); -1: chosen from the batch size
static int poisson_create_impl(dfta_ctx* ctx, const dfta_grid* g, int batch, int force_logG, int mode, dfta_poisson** out)
{
    DFTA_REQUIRE(ctx, batch >= 1 && g->levels <= kMaxLevels, "poisson batch/levels");
    dfta_poisson* p = new dfta_poisson();
    p->ctx = ctx; p->g = g; p->batch = batch;
    p->tol = mode == DFTA_POISSON_TOLERANCE || mode == DFTA_POISSON_ADAPTIVE;
    p->adaptive = mode == DFTA_POISSON_ADAPTIVE;
    MgDesc& D = p->D;
    D.levels = g->levels;
    // Workgroups per atom: a solve is bound by ONE compute unit's vector-memory path, so while the batch leaves compute
    // units idle the fine levels of every atom are shared by a group of G workgroups (all of them must be resident:
    // batch * G <= 256 CUs).  A level is shared when every lane of the group still owns >= 8 nodes (>= 4 for G = 16: the
    // same four levels at 131073 nodes, with 32 nodes per lane on the finest one -- the most that is staged in LDS).
    // round 3, re-measured with the fused visits in place (131073 nodes, ms per solve): 8 atoms 39.8 (G = 16) / 48.7 (8); 12: 42.4 / 50.5;
    // 16: 45.0 / 51.4; 32: 59.2 (8) / 71.6 (4); 64: 99.6 (4) / 124 (2) / 149 (1); 96: 150 (2) / 156 (1); 128: 184 (2) / 171 (1).
    // End of round 3, with fused visits on the global levels too (gs_fused3, also on shared levels): 16 atoms 45.1 (16) / 50.1 (8);
    // 32: 53.6 (8) / 64.7 (4); 64: 77.3 (4) / 99.5 (2) / 137 (1); 96: 113 (2) / 145 (1); 112: 118 / 154; 128: 129 (2) / 153 (1)
    int logG = batch <= 16 ? 4 : (batch <= 32 ? 3 : (batch <= 64 ? 2 : ((batch <= 128 && 2 * batch <= std::max(ctx->num_cu, 1)) ? 1 : 0)));
    // 1 048 577 nodes, up to four atoms: 32 workgroups per atom (measured: 95.3 -> 84.0 ms for one atom, 99.5 -> 92.0 for four; 64 workgroups
    // 88.5; at eight atoms, and at 131 073 nodes, 16 remain faster: the barrier of a larger group costs more than the shorter chunks save)
    if (batch <= 4 && g->N - 1 >= (1 << 20)) logG = 5;
    if (const char* e = dfta_knob("POISSON_GROUP")) {      // measurements: force log2 of the group size
        const int v = atoi(e);
        if (v >= 0 && v <= 6 && (batch << v) <= 256) logG = v;
    }
    if (force_logG >= 0) logG = force_logG;
    if (const char* e = dfta_knob("FAULT_POISSON_MEMBER")) p->fault = atoi(e) != 0;
    // Resident group (k_poisson_solve_res): where the batch leaves 33 compute units per atom (up to 7 atoms) and level 0 gives every
    // lane of kResG x kResNT lanes 4 .. 32 nodes (16385 .. 131073 nodes); the knob POISSON_RES = 0 / 1 switches it off / on, a forced
    // group size (POISSON_GROUP, force_logG) selects the staged groups above
    int res_kres = 0, res_logC0 = 0;
    {
        // every atom of the batch gets its 33 workgroups at once: up to 7 atoms on 256 compute units (measured: 28.6 .. 29.0 ms per
        // 131073-node solve for 5 .. 7 atoms against 46 .. 48 ms with staged groups of 8)
        bool want = batch * kResWG <= ctx->num_cu && force_logG < 0 && !dfta_knob("POISSON_GROUP") && !dfta_knob("POISSON_NOSTAGE");   // (the hand-over needs the first coarse level staged)
        if (const char* e = dfta_knob("POISSON_RES")) want = atoi(e) != 0 && force_logG < 0 && batch * kResWG <= 256 && !dfta_knob("POISSON_NOSTAGE");
        // 8 .. 15 atoms: 17 workgroups per atom, 16 members of 256 lanes whose level 0 takes turns with their other shared levels in
        // LDS (mg_exact16 / mg_tol16); POISSON_RES16 = 0 / 1 switches it off / on (1: for any batch of up to 15 atoms)
        bool want16 = !want && batch * mg_exact16::kResWG <= ctx->num_cu && force_logG < 0 && !dfta_knob("POISSON_GROUP") &&
                      !dfta_knob("POISSON_NOSTAGE") && !dfta_knob("POISSON_RES");
        if (const char* e = dfta_knob("POISSON_RES16"))
            want16 = atoi(e) != 0 && force_logG < 0 && batch * mg_exact16::kResWG <= ctx->num_cu && !dfta_knob("POISSON_NOSTAGE");
        if (want16) { want = true; p->res16 = true; }
        const int lanes = kResG * kResNT;          // (the same 4096 lanes in both configurations)
        if (want && (g->N - 1) % lanes == 0) {
            const int C0 = (g->N - 1) / lanes;
            int lc = 0;
            while ((1 << lc) < C0) ++lc;
            if ((1 << lc) == C0 && lc >= 2 && lc <= 5 && g->levels >= lc + 4) { res_logC0 = lc; res_kres = lc - 1; }
        }
        if (res_kres > 0) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, K_SOLVE_RES(p), kThreads, 0) != hipSuccess) per_cu = 0;
            if (batch * p->res_wg() > per_cu * ctx->num_cu) res_kres = 0;
        }
        if (res_kres > 0) { logG = 0; p->resident = true; }
        else p->res16 = false;
    }
    // rocprofiler-sdk (ROCm 7.2) crashes in an exit handler of a process that has made a cooperative launch -- after its
    // output is written, but the profiled command returns 139.  Under the profiler (rocprofv3 exports ROCP_TOOL_LIBRARIES), or
    // when DFTA_POISSON_PLAIN_LAUNCH is set, the groups are therefore started with an ordinary launch: same kernel, same
    // results and timing; co-residency then rests on the occupancy query of this function, the bounded spins and the abort
    // flag (dfta_poisson_finish) as in round 1.
    p->plain_launch = dfta_knob("POISSON_PLAIN_LAUNCH") != nullptr || getenv("ROCP_TOOL_LIBRARIES") != nullptr;
    D.spin_max = p->fault ? (1 << 12) : (1 << 23);
    {
        // every workgroup of the launch must be resident at once (the members wait for each other)
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, K_SOLVE(p), kThreads, 0) != hipSuccess) per_cu = 1;
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
    D.nofold = dfta_knob("POISSON_NOFOLD") ? 1 : 0;
    D.fold_lds = dfta_knob("POISSON_NOFOLD_LDS") ? 0 : 1;
    D.fuse3 = dfta_knob("POISSON_NOFUSE3") ? 0 : 1;
    D.fuse3w = (D.fuse3 && !dfta_knob("POISSON_NOFUSE3_WAVE")) ? 1 : 0;
    D.fuse_min_logc = kFuseMinLogC;
    D.fuse_coop = dfta_knob("POISSON_NOFUSE_COOP") ? 0 : 1;
    if (const char* e = dfta_knob("POISSON_FUSE_MIN_LOGC")) D.fuse_min_logc = std::max(kFuseMinLogC, atoi(e));   // measurements (99: never; staged levels -- <= 32 nodes per lane -- have their own fused pass)
    D.dbg = dfta_knob("POISSON_DBG") ? atoi(dfta_knob("POISSON_DBG")) : 0;
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
            if (!dfta_knob("POISSON_NOSTAGE")) {
                // (round 6: 2 049 nodes tried -- level 6 inside the coarse section, 32 nodes per lane: the solve got 0.5 ms slower)
                if (l >= D.kcoop && n <= kWaveMaxN && n >= 129 && !dfta_knob("POISSON_NOSTAGE_WAVE")) L.stage = 3;
                else if (l >= D.kcoop && L.logT == 8 && L.logC <= kStageMaxLogC) L.stage = 1;
                else if (l < D.kcoop && D.G > 1 && L.logT == 8 + logG && L.logC >= 2 && L.logC <= kStageMaxLogC &&
                         !dfta_knob("POISSON_NOSTAGE_SHARED")) L.stage = 2;
            }
        }
        off += n;
        n = (n + 1) / 2;
        d *= 2;
    }
    D.per_atom = off;
    // coarse section: from the first one-wave level down, if everything below is one-wave or sequential and fits the staging memory
    D.cs_top = -1;
    if (!dfta_knob("POISSON_NOCOARSE")) {
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
                // exact mode: the 129-node level is laid out 4 nodes per lane on 32 lanes (rows of 64 all the same: its last node sits behind
                // the four rows, cs_idx) so that its visits run as the fused three-sweep pass too (gs_lds3 needs >= 4 nodes per lane)
                if (lc == 1 && !p->tol && !dfta_knob("POISSON_NOFUSE3") && !dfta_knob("POISSON_NOFUSE3_WAVE") && !dfta_knob("POISSON_NOHALF129")) lc = 2;
                const int span = std::max(L.n, (64 << lc) + 1);    // the last node's slot: (2^lc) << 6
                D.cs_lc[l] = lc;
                D.cs_phi[l] = at + kStagePad; at += kStagePad + span + 8;
                D.cs_src[l] = at + kStagePad; at += kStagePad + span + 8;
            } else if (L.seq) {
                D.cs_lc[l] = -1;
                D.cs_phi[l] = at; at += L.n + 1;
                D.cs_src[l] = at; at += L.n + 1;
            } else ok = false;
        }
        if (ok && at <= 2 * kStageArr - 64) D.cs_top = top;
    }
    // exact mode: the six coarsest levels of the coarse section in registers (poisson_kernels.inc: xw_section), entered from the 129-node level
    D.xw_top = -1;
    if (!p->tol && D.cs_top > 0 && D.levels >= 8 && D.levels - 6 > D.cs_top && D.lv[D.levels - 6].n == 65 && D.lv[D.levels - 1].n == 3 &&
        D.cs_lc[D.levels - 6] < 0 && (D.cs_lc[D.levels - 7] == 1 || D.cs_lc[D.levels - 7] == 2) && D.lv[D.levels - 7].n == 129 && !dfta_knob("POISSON_NOXW"))
        D.xw_top = D.levels - 6;
    // tolerance mode: the sub-cycle from the 8193-node level down in registers (poisson_kernels.inc: coarse_resident_cycle) -- 32 nodes per
    // thread on its first level, the levels down to 257 nodes halve the chunk, the 129-node level and below run in one wave.  Resident
    // groups: the coarse workgroup's levels; staged groups and one workgroup per atom: workgroup 0's, from the first level it does not share
    // (8193 nodes for groups of 8 and 16 and for a lone workgroup, 4097 / 2049 nodes -- 16 / 8 per thread -- for groups of 4 / 2)
    D.adaptive = p->adaptive ? 1 : 0;
    D.rc_top = -1;
    int k8193 = -1;
    for (int l = 0; l < D.levels; ++l) if (D.lv[l].n == 8193) k8193 = l;
    int sft = 0;                                    // the cycle starts `sft` levels below the 8193-node level
    if (k8193 > 0 && res_kres == 0) while (sft < 2 && k8193 + sft < D.kcoop) ++sft;
    if (p->tol && k8193 > 0 && D.cs_top > 0 && (res_kres > 0 ? k8193 == res_kres : k8193 + sft >= D.kcoop) && !dfta_knob("POISSON_NORC")) {
        const int kt = k8193 + sft;
        const Lvl& Lk = D.lv[kt];
        bool ok = !Lk.seq && Lk.logT == 8 && Lk.logC == 5 - sft && k8193 + 6 < D.levels && D.lv[k8193 + 5].n == 257 &&
                  D.lv[k8193 + 6].n == 129 && k8193 + 6 >= D.cs_top && D.cs_lc[k8193 + 6] == 1;
        // a wave's scan end value goes to the next wave without what entered the wave itself: a^(64 C) of it, largest on the 257-node level
        // (C = 1, a = (1 + delta_l / 2) / 2) -- 3e-19 on the grids of BASELINE.md; a grid coarse enough to make it matter stays level by level
        if (ok) ok = std::pow(0.5 * (1.0 + 0.5 * D.lv[k8193 + 5].d), 64.0) < 1e-16;
        if (ok) {
            // the sources of the six register levels live in the staging memory around the coarse section's arrays of the levels it
            // still runs (129 nodes and below): 8192 + 4096 behind them, 2048 + 1024 + 512 + 256 in front (where the section's own
            // copies of the 1025 .. 257-node levels would be)
            int first10 = 1 << 30, end_cs = 0;
            for (int l = k8193 + 6; l < D.levels; ++l) {
                first10 = std::min(first10, std::min(D.cs_phi[l], D.cs_src[l]) - (D.cs_lc[l] >= 0 ? kStagePad : 0));
                end_cs = std::max(end_cs, std::max(D.cs_phi[l], D.cs_src[l]) + D.lv[l].n + 8);
            }
            const int cap = 2 * kStageArr - 64;
            ok = first10 >= 3840 && end_cs + 12288 <= cap;
            if (ok) {
                const int slot[6] = {end_cs, end_cs + 8192, 0, 2048, 3072, 3584};      // by level: 8193, 4097, 2049, 1025, 513, 257 nodes
                for (int j = 0; j < 6; ++j) D.rc_src[j] = j + sft < 6 ? slot[j + sft] : 0;          // index: level - rc_top
                D.rc_top = kt;
            }
        }
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
    if (e == hipSuccess && p->res16) {
        // per member: Phi and source of level 0 as they lie in LDS (C0 rows of H0 + 256 + 1 columns each)
        const int C0 = 1 << res_logC0, RS0 = ((p->tol ? mg_tol16::kWarm3 : mg_exact16::kWarm3) + 3 + C0 - 1) / C0 + mg_exact16::kResNT + 1;
        e = hipMalloc(reinterpret_cast<void**>(&p->d_res_spill), sizeof(double) * (size_t)batch * mg_exact16::kResG * 2 * C0 * RS0);
    }
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
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(mg_exact::g_prof), sizeof(h)) == hipSuccess) {
            const char* names[8] = {"restrict", "prolong ", "iterate ", "grp_sum ", "gs_lds  ", "copy_in ", "copy_out", "sweeps1 "};
            unsigned long long tot = 0;
            for (int c = 0; c < 8; ++c) {
                fprintf(stderr, "[poisson prof] %s:", names[c]);
                for (int l = 0; l < 22; ++l) { fprintf(stderr, " %llu", h[c * 24 + l]); tot += h[c * 24 + l]; }
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "[poisson prof] total ticks %llu\n", tot);
            unsigned long long hm[64 * 4];
            if (hipMemcpyFromSymbol(hm, HIP_SYMBOL(mg_exact::g_prof_member), sizeof(hm)) == hipSuccess) {
                for (int m = 0; m < 16; ++m) fprintf(stderr, "[poisson prof] member %2d: exchange %llu  sweeps %llu  copy-in %llu\n", m, hm[m * 4], hm[m * 4 + 1], hm[m * 4 + 2]);
                unsigned long long zz[64 * 4] = {0};
                (void)hipMemcpyToSymbol(HIP_SYMBOL(mg_exact::g_prof_member), zz, sizeof(zz));
            }
            unsigned long long z[8 * 24] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(mg_exact::g_prof), z, sizeof(z));
        }
    }
#endif
#ifdef DFTA_POISSON_RPROF
    {
        unsigned long long hr[2 * 8 * 8];
        if (p->resident && (p->res16 ? (p->tol ? hipMemcpyFromSymbol(hr, HIP_SYMBOL(mg_tol16::g_rprof), sizeof(hr)) : hipMemcpyFromSymbol(hr, HIP_SYMBOL(mg_exact16::g_rprof), sizeof(hr))) : (p->tol ? hipMemcpyFromSymbol(hr, HIP_SYMBOL(mg_tol::g_rprof), sizeof(hr)) : hipMemcpyFromSymbol(hr, HIP_SYMBOL(mg_exact::g_rprof), sizeof(hr)))) == hipSuccess) {
            const char* mn[8] = {"pass    ", "publish ", "exchange", "commit  ", "restrict", "prolong ", "handover", "redo    "};
            const char* cn[8] = {"passive ", "cs sweep", "iterate ", "coarsesc", "restrict", "prolong ", "handover", "cs r/p/xw/enter/leave"};
            for (int role = 0; role < 2; ++role)
                for (int c = 0; c < 8; ++c) {
                    unsigned long long t = 0;
                    fprintf(stderr, "[res prof] %s %s:", role ? "coarse wg" : "member 0 ", role ? cn[c] : mn[c]);
                    for (int l = 0; l < 8; ++l) { fprintf(stderr, " %llu", hr[(role * 8 + c) * 8 + l]); t += hr[(role * 8 + c) * 8 + l]; }
                    fprintf(stderr, "  = %llu\n", t);
                }
            unsigned long long zz[2 * 8 * 8] = {0};
            if (p->res16 && p->tol) (void)hipMemcpyToSymbol(HIP_SYMBOL(mg_tol16::g_rprof), zz, sizeof(zz));
            else if (p->res16) (void)hipMemcpyToSymbol(HIP_SYMBOL(mg_exact16::g_rprof), zz, sizeof(zz));
            else if (p->tol) (void)hipMemcpyToSymbol(HIP_SYMBOL(mg_tol::g_rprof), zz, sizeof(zz));
            else (void)hipMemcpyToSymbol(HIP_SYMBOL(mg_exact::g_rprof), zz, sizeof(zz));
        }
    }
#endif
    void* ptrs[] = {p->d_phi0, p->d_phi1, p->d_src, p->d_cur, p->d_total_vcycles, p->d_desc, p->d_group_ctr, p->d_group_part, p->d_res_slots, p->d_res_spill};
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
        hipLaunchKernelGGL(K_UNIT(p), dim3(1), dim3(kThreads), 0, ctx->stream, p->d_desc, p->d_phi0, p->d_phi1, p->d_src, p->d_cur, op, lvl, sweeps,
                           dOut, p->d_group_ctr, p->d_group_part);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    if (p->plain_launch) {
        hipLaunchKernelGGL(K_UNIT(p), dim3(p->D.G), dim3(kThreads), 0, ctx->stream, p->d_desc, p->d_phi0, p->d_phi1, p->d_src, p->d_cur, op, lvl, sweeps,
                           dOut, p->d_group_ctr, p->d_group_part);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    const MgDesc* a0 = p->d_desc;
    void* args[] = {&a0, &p->d_phi0, &p->d_phi1, &p->d_src, &p->d_cur, &op, &lvl, &sweeps, &dOut, &p->d_group_ctr, &p->d_group_part};
    DFTA_HIP(ctx, hipLaunchCooperativeKernel(reinterpret_cast<const void*>(K_UNIT(p)), dim3(p->D.G), dim3(kThreads), args, 0, ctx->stream));
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
