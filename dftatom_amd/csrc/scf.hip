// scf.hip -- one SCF iteration for a batch of atoms, state resident in HBM.
//
// Replaces the body of DFTAtom::CalculateNonUniformLDA (DFTAtom.cpp:346-491) and
// DFTAtom::CalculateNonUniformLSDA (DFTAtom.cpp:847-1022): flat-density start, level search for every
// (atom, spin), density mixing, multigrid Poisson, VWN, the pointwise potential/integrand pass, the five
// Simpson 3/8 integrals (reference summation order) and the energy assembly + convergence test.
// Potentials are stored as V[(atom * nspin + spin) * N + i].
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "internal.h"
#include "levels.h"
#include "ordered_sum.h"
#include "xc.h"

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double fourM_PI = 4. * kPi;
constexpr double kTotalEnergyErr = 1E-11;   // DFTAtom.cpp:349

struct AtomState {          // per atom, device
    int Z;
    int nAlphaE, nBetaE;    // electrons per spin (LSDA), DFTAtom.cpp:611-638
    int job_off, job_end;   // jobs of this atom in the level solver (alpha levels first, then beta)
    int lastTimeConverged;
    int finished;
    int steps;
    double Eold;
    dfta_energies e;
};

// flat start density (DFTAtom.cpp:371-376 / 874-884)
__global__ void k_init_density(const AtomState* __restrict__ atoms, int lsda, int N, double MaxR, double* __restrict__ density,
                               double* __restrict__ dA, double* __restrict__ dB)
{
    const int a = blockIdx.y;
    const double volume = fourM_PI / 3. * MaxR * MaxR * MaxR;
    const AtomState s = atoms[a];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)a * N + i;
        if (!lsda) density[o] = (i == 0) ? 0. : s.Z / volume;
        else {
            const double cA = s.nAlphaE / volume, cB = s.nBetaE / volume;
            dA[o] = (i == 0) ? 0. : cA;
            dB[o] = (i == 0) ? 0. : cB;
            density[o] = (i == 0) ? 0. : cA + cB;
        }
    }
}

// potential from U and v_xc (DFTAtom.cpp:387-392 / 895-904, also the first statement of the tail loop)
__global__ void k_potential(const AtomState* __restrict__ atoms, int lsda, int N, const double* __restrict__ r,
                            const double* __restrict__ U, const double* __restrict__ Vexc, const double* __restrict__ va,
                            const double* __restrict__ vb, double* __restrict__ V)
{
    const int a = blockIdx.y;
    const int Z = atoms[a].Z;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)a * N + i;
        if (!lsda) V[o] = (i == 0) ? 0. : (-Z + U[o]) / r[i] + Vexc[o];
        else {
            const double u = (i == 0) ? 0. : (-Z + U[o]) / r[i];
            V[((size_t)2 * a) * N + i] = (i == 0) ? 0. : u + va[o];
            V[((size_t)2 * a + 1) * N + i] = (i == 0) ? 0. : u + vb[o];
        }
    }
}

// newDensity /= 4 pi r^2; density = alpha density + (1-alpha) newDensity (DFTAtom.cpp:332-342); LSDA total (DFTAtom.cpp:933-934)
__global__ void k_mix(int lsda, int N, double alpha, double oneMinusAlpha, const double* __restrict__ fpr2,
                      double* __restrict__ newDensity, double* __restrict__ density, double* __restrict__ dA, double* __restrict__ dB,
                      const int* __restrict__ fin)
{
    const int a = blockIdx.y;
    if (fin[a]) return;        // a finished atom is frozen (the reference leaves its loop: DFTAtom.cpp:474-479)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        if (i == 0) continue;
        const size_t o = (size_t)a * N + i;
        if (!lsda) {
            double nd = newDensity[o];
            nd /= fpr2[i];
            newDensity[o] = nd;
            density[o] = alpha * density[o] + oneMinusAlpha * nd;
        } else {
            const size_t oa = ((size_t)2 * a) * N + i, ob = ((size_t)2 * a + 1) * N + i;
            double na = newDensity[oa], nb = newDensity[ob];
            na /= fpr2[i];
            nb /= fpr2[i];
            newDensity[oa] = na;
            newDensity[ob] = nb;
            const double x = alpha * dA[o] + oneMinusAlpha * na;
            const double y = alpha * dB[o] + oneMinusAlpha * nb;
            dA[o] = x;
            dB[o] = y;
            density[o] = x + y;
        }
    }
}

// new potential + the five integrands (DFTAtom.cpp:437-457 / 956-983); integrands: [atom][5][N]
__global__ void k_tail(const AtomState* __restrict__ atoms, int lsda, int N, const double* __restrict__ r,
                       const double* __restrict__ cnst, const double* __restrict__ density, const double* __restrict__ dA,
                       const double* __restrict__ dB, const double* __restrict__ U, const double* __restrict__ Vexc,
                       const double* __restrict__ va, const double* __restrict__ vb, const double* __restrict__ eexc,
                       double* __restrict__ V, double* __restrict__ integrands, int uniform)
{
    const int a = blockIdx.y;
    if (atoms[a].finished) return;
    const int Z = atoms[a].Z;
    double* nuclear = integrands + ((size_t)a * 5 + 0) * N;
    double* exccor = integrands + ((size_t)a * 5 + 1) * N;
    double* eexcD = integrands + ((size_t)a * 5 + 2) * N;
    double* hartree = integrands + ((size_t)a * 5 + 3) * N;
    double* potentiale = integrands + ((size_t)a * 5 + 4) * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)a * N + i;
        if (i == 0) {
            if (!lsda) V[o] = 0; else { V[((size_t)2 * a) * N] = 0; V[((size_t)2 * a + 1) * N] = 0; }
            nuclear[0] = 0; exccor[0] = 0; eexcD[0] = 0; hartree[0] = 0; potentiale[0] = 0;
            continue;
        }
        const double position = r[i];
        const double c = cnst[i];
        const double rho = density[o];
        if (!lsda) {
            const double pot = (-Z + U[o]) / position + Vexc[o];
            V[o] = pot;
            const double positiondensity = position * rho * c;
            nuclear[i] = Z * positiondensity;
            const double position2density = position * position * rho * c;
            exccor[i] = position2density * Vexc[o];
            eexcD[i] = position2density * eexc[o];
            hartree[i] = positiondensity * U[o];
            potentiale[i] = position2density * pot;
        } else {
            const double u = (-Z + U[o]) / position;
            const double pa = u + va[o], pb = u + vb[o];
            V[((size_t)2 * a) * N + i] = pa;
            V[((size_t)2 * a + 1) * N + i] = pb;
            const double positioncnst = position * c;
            const double positiondensity = positioncnst * rho;
            nuclear[i] = Z * positiondensity;
            const double position2cnst = position * positioncnst;
            const double position2density = position2cnst * rho;
            const double p2a = position2cnst * dA[o];
            const double p2b = position2cnst * dB[o];
            exccor[i] = position2density * Vexc[o];
            eexcD[i] = position2density * eexc[o];
            hartree[i] = positiondensity * U[o];
            // DFTAtom.cpp:981 on the logarithmic grid; the uniform-grid loop groups the product differently (DFTAtom.cpp:799)
            potentiale[i] = uniform ? (position * position) * (dA[o] * pa + dB[o] * pb) : p2a * pa + p2b * pb;
        }
    }
}

// energy assembly and the reference's stop test (DFTAtom.cpp:459-481 / 985-1006); one thread per atom
__global__ void k_energies(AtomState* __restrict__ atoms, int natoms, const dfta::Job* __restrict__ jobs,
                           const double* __restrict__ integrals, double* __restrict__ records, int* __restrict__ fin)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= natoms) return;
    AtomState s = atoms[a];
    if (s.finished) return;     // energies, eigenvalues, step count and record stay those of the finishing step
    double Eelectronic = 0;
    bool conv = true;
    for (int k = s.job_off; k < s.job_end; ++k) {
        Eelectronic += jobs[k].occ * jobs[k].E;                    // DFTAtom.cpp:561
        conv = conv && (jobs[k].converged != 0);
    }
    const double* I = integrals + (size_t)a * 5;
    const double Enuclear = -fourM_PI * I[0];
    double Exc = fourM_PI * I[1];
    const double eExcDif = fourM_PI * I[2];
    Exc += eExcDif;
    const double Ehartree = -2 * kPi * I[3];
    const double Epotential = fourM_PI * I[4];
    const double Ekinetic = Eelectronic - Epotential;
    const double Etotal = Eelectronic + Ehartree + eExcDif;
    s.e.Etotal = Etotal; s.e.Ekinetic = Ekinetic; s.e.Ecoul = -Ehartree; s.e.Enuclear = Enuclear; s.e.Exc = Exc;
    s.e.Eelectronic = Eelectronic; s.e.Ehartree = Ehartree; s.e.eExcDif = eExcDif; s.e.Epotential = Epotential;
    s.steps++;
    if (fabs((s.Eold - Etotal) / Etotal) < kTotalEnergyErr && conv && s.lastTimeConverged) s.finished = 1;
    else { s.Eold = Etotal; s.lastTimeConverged = conv ? 1 : 0; }
    atoms[a] = s;
    fin[a] = s.finished;
    if (records) {
        double* R = records + (size_t)a * DFTA_RECORD_DOUBLES;
        R[0] = s.Z; R[1] = Etotal; R[2] = Ekinetic; R[3] = -Ehartree; R[4] = Enuclear; R[5] = Exc; R[6] = s.finished;
        R[7] = s.steps; R[8] = s.job_end - s.job_off; R[9] = conv ? 1. : 0.;
        int m = 10;
        for (int k = s.job_off; k < s.job_end && m < DFTA_RECORD_DOUBLES; ++k) R[m++] = jobs[k].E;
        for (; m < DFTA_RECORD_DOUBLES; ++m) R[m] = 0;
    }
}

}  // namespace

// rows idx[y] of src -> row y of dst, and back (the live atoms of a batch, see dfta_scf::live_solver)
__global__ void k_gather_rows(const double* __restrict__ src, const int* __restrict__ idx, int N, double* __restrict__ dst)
{
    const double* s = src + (size_t)idx[blockIdx.y] * N;
    double* d = dst + (size_t)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void k_scatter_rows(const double* __restrict__ src, const int* __restrict__ idx, int N, double* __restrict__ dst)
{
    const double* s = src + (size_t)blockIdx.y * N;
    double* d = dst + (size_t)idx[blockIdx.y] * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) d[i] = s[i];
}

constexpr int kLiveClasses = 6;
constexpr int kLiveClassAtoms[kLiveClasses] = {7, 15, 16, 32, 64, 128};     // (7 / 15: the resident multigrid's two configurations)

struct dfta_scf {
    dfta_ctx* ctx = nullptr;
    const dfta_grid* g = nullptr;
    int lsda = 0, natoms = 0, nspin = 1, nV = 0;
    double alpha = 0.5;
    dfta::LevelSolver solver;
    dfta_poisson* poisson = nullptr;
    // Finished atoms are frozen, and the multigrid's workgroups per atom are a function of the batch size (1 for > 128 atoms ... 16 for
    // <= 16, the resident groups' 33 for <= 7): once the LIVE atoms of a batch fit a smaller size class, their solve goes to a second
    // solver made for that class, on gathered copies of their densities (results scattered back; the same bits -- every grouping of
    // the multigrid is bit-identical to one workgroup per atom).
    struct LiveSolver { dfta_poisson* p = nullptr; int G = 0; bool tried = false; };
    LiveSolver live_solver[kLiveClasses];
    int live_cap = 0;                     // atoms the gather buffers hold
    int* d_liveZ = nullptr;               // live_cap charges, live_cap skip flags, live_cap atom indices
    double *d_liveRho = nullptr, *d_liveU = nullptr;
    std::vector<AtomState> h_atoms;
    std::vector<double> h_bottom0;        // per potential: -Z^2-1 (DFTAtom.cpp:407)
    std::vector<double> h_job_bottom;     // per job: bracket start of the next level solve
    std::vector<dfta::Job> h_jobs;        // job results of the last step
    std::vector<unsigned char> h_frozen;  // per job: its atom has finished
    int* d_fin = nullptr;                 // per atom: finished (device copy for the kernels that skip frozen atoms)
    bool debug_levels = false;            // $DFTA_DEBUG_LEVELS
    int integ_rule = DFTA_INT_SIMPSON38;  // dfta_scf_set_integrator
    int functional = DFTA_XC_VWN;         // dfta_scf_options::functional
    int fallbacks_seen = 0;               // level-search fallbacks already reported in a step's statistics
    int levels_mode = DFTA_LEVELS_BATCHED;
    int steps_done = 0;
    std::vector<int> spin_nlev[2];      // per atom number of levels per spin
    AtomState* d_atoms = nullptr;
    int* d_Z = nullptr;
    double *d_density = nullptr, *d_dA = nullptr, *d_dB = nullptr, *d_V = nullptr, *d_U = nullptr, *d_Vexc = nullptr,
           *d_va = nullptr, *d_vb = nullptr, *d_eexc = nullptr, *d_newDensity = nullptr, *d_integrands = nullptr,
           *d_integrals = nullptr, *d_records = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

static int scf_xc(dfta_scf* s)
{
    const size_t sz = (size_t)s->natoms * s->g->N;
    if (!s->lsda && s->functional != DFTA_XC_VWN)
        return dfta_launch_chachiyo_lda(s->ctx, s->functional == DFTA_XC_CHACHIYO_IMPROVED, s->d_density, sz, s->d_Vexc, s->d_eexc);
    if (!s->lsda) return dfta_launch_vwn_lda(s->ctx, s->d_density, sz, s->d_Vexc, s->d_eexc);
    return dfta_launch_vwn_lsda(s->ctx, s->d_dA, s->d_dB, sz, s->d_Vexc, s->d_va, s->d_vb, s->d_eexc);
}

extern "C" {

void dfta_scf_destroy(dfta_scf* s)
{
    if (!s) return;
    void* ptrs[] = {s->d_fin, s->d_atoms, s->d_Z, s->d_density, s->d_dA, s->d_dB, s->d_V, s->d_U, s->d_Vexc, s->d_va, s->d_vb, s->d_eexc,
                    s->d_newDensity, s->d_integrands, s->d_integrals, s->d_records};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t e : s->ev) if (e) (void)hipEventDestroy(e);
    if (s->poisson) dfta_poisson_destroy(s->poisson);
    for (auto& ls : s->live_solver) if (ls.p) dfta_poisson_destroy(ls.p);
    for (void* p : {(void*)s->d_liveZ, (void*)s->d_liveRho, (void*)s->d_liveU}) if (p) (void)hipFree(p);
    delete s;
}

int dfta_abi_version(void) { return DFTA_ABI_VERSION; }

int dfta_scf_create(dfta_ctx* ctx, const dfta_grid* g, int lsda, int natoms, const int* Z, double alpha, int levels_mode,
                    int tree_depth, dfta_scf** out)
{
    return dfta_scf_create_ex(ctx, g, lsda, natoms, Z, alpha, levels_mode, tree_depth, nullptr, out);
}

int dfta_scf_create_ex(dfta_ctx* ctx, const dfta_grid* g, int lsda, int natoms, const int* Z, double alpha, int levels_mode,
                       int tree_depth, const dfta_scf_options* options, dfta_scf** out)
{
    if (!ctx || !g || !out) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, natoms >= 1 && Z && alpha >= 0 && alpha <= 1, "scf arguments");
    dfta_scf_options opt = {(int)sizeof(dfta_scf_options), DFTA_INT_SIMPSON38, DFTA_XC_VWN, DFTA_AUFBAU_REFERENCE, -1, DFTA_SWEEPS_EXACT};
    if (options) {
        // the caller's struct may be shorter (an older header) or longer (a newer one) than this library's: read what both know
        const int sz = options->struct_size;
        DFTA_REQUIRE(ctx, sz >= (int)(2 * sizeof(int)) && sz % (int)sizeof(int) == 0 && sz <= 4096, "dfta_scf_options::struct_size (set it to sizeof(dfta_scf_options))");
        dfta_scf_options in = {};
        memcpy(&in, options, std::min<size_t>((size_t)sz, sizeof(in)));
        // members the caller's struct does not have keep the defaults above; those it has are taken as they are (0 = the reference's behaviour)
        const size_t have = std::min<size_t>((size_t)sz, sizeof(in));
        memcpy(&opt, &in, have);
        opt.struct_size = (int)sizeof(dfta_scf_options);
    }
    DFTA_REQUIRE(ctx, opt.poisson_mode >= -1 && opt.poisson_mode <= DFTA_POISSON_ADAPTIVE, "poisson mode");
    DFTA_REQUIRE(ctx, dfta_integral_shape_ok(opt.integrator, g->N), "integration rule / grid size");
    DFTA_REQUIRE(ctx, opt.functional >= DFTA_XC_VWN && opt.functional <= DFTA_XC_CHACHIYO_IMPROVED, "functional");
    DFTA_REQUIRE(ctx, opt.functional == DFTA_XC_VWN || !lsda, "the Chachiyo functional is LDA only (ExcCor.h)");
    DFTA_REQUIRE(ctx, opt.aufbau == DFTA_AUFBAU_REFERENCE || opt.aufbau == DFTA_AUFBAU_TRANSITION_METALS, "aufbau");
    DFTA_REQUIRE(ctx, opt.sweep_mode == DFTA_SWEEPS_EXACT || opt.sweep_mode == DFTA_SWEEPS_TOLERANCE, "sweep mode");
    DFTA_REQUIRE(ctx, opt.sweep_mode == DFTA_SWEEPS_EXACT || dfta_scan_supported(g), "the tolerance mode of the sweeps needs a logarithmic grid of 12 .. 20 multigrid levels");
    dfta_scf* s = new dfta_scf();
    s->solver.sweep_mode = opt.sweep_mode;
    s->integ_rule = opt.integrator;
    s->solver.integ_rule = opt.integrator;
    s->functional = opt.functional;
    s->ctx = ctx; s->g = g; s->lsda = lsda ? 1 : 0; s->natoms = natoms; s->nspin = lsda ? 2 : 1; s->nV = natoms * s->nspin;
    s->alpha = alpha;
    const int N = g->N;
    std::vector<dfta::JobSpec> specs;
    s->h_atoms.resize(natoms);
    s->h_bottom0.resize(s->nV);
    s->spin_nlev[0].resize(natoms); s->spin_nlev[1].assign(natoms, 0);
    for (int a = 0; a < natoms; ++a) {
        DFTA_REQUIRE(ctx, Z[a] >= 1 && Z[a] <= 118, "Z out of range");
        AtomState& st = s->h_atoms[a];
        memset(&st, 0, sizeof(st));
        st.Z = Z[a];
        st.job_off = (int)specs.size();
        int an[32], al[32], ao[32], bn[32], bl[32], bo[32], nA = 0, nB = 0;
        if (!lsda) {
            nA = dfta_get_subshells_ex(Z[a], opt.aufbau, an, al, ao, 32);
            if (nA < 0) { dfta_scf_destroy(s); return DFTA_ERR_INVALID; }
            for (int k = 0; k < nA; ++k) specs.push_back({a, an[k], al[k], ao[k]});
            s->h_bottom0[a] = -double(Z[a]) * Z[a] - 1.;                                   // DFTAtom.cpp:407
        } else {
            if (dfta_split_spin_ex(Z[a], opt.aufbau, &nA, &nB, an, al, ao, bn, bl, bo, 32) != DFTA_OK) { dfta_scf_destroy(s); return DFTA_ERR_INVALID; }
            int ne = 0;
            for (int k = 0; k < nA; ++k) { specs.push_back({2 * a, an[k], al[k], ao[k]}); ne += ao[k]; }
            for (int k = 0; k < nB; ++k) specs.push_back({2 * a + 1, bn[k], bl[k], bo[k]});
            st.nAlphaE = ne;
            st.nBetaE = Z[a] - ne;
            s->h_bottom0[2 * a] = s->h_bottom0[2 * a + 1] = -double(Z[a]) * Z[a] - 1.;     // DFTAtom.cpp:919,929
        }
        s->spin_nlev[0][a] = nA;
        s->spin_nlev[1][a] = nB;
        st.job_end = (int)specs.size();
    }
    s->levels_mode = levels_mode;
    s->h_job_bottom.resize(specs.size());
    for (size_t k = 0; k < specs.size(); ++k) s->h_job_bottom[k] = s->h_bottom0[specs[k].v];
    int rc = s->solver.setup(ctx, g, levels_mode, tree_depth, s->nV, specs);
    if (rc) { dfta_scf_destroy(s); return rc; }
    rc = opt.poisson_mode < 0 ? dfta_poisson_create(ctx, g, natoms, &s->poisson) : dfta_poisson_create_ex(ctx, g, natoms, opt.poisson_mode, &s->poisson);
    if (rc) { dfta_scf_destroy(s); return rc; }

    hipStream_t st = ctx->stream;
    const size_t aN = (size_t)natoms * N;
    hipError_t e = hipSuccess;
    auto al = [&](double** p, size_t cnt) { if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(p), cnt * sizeof(double)); };
    al(&s->d_density, aN); al(&s->d_dA, aN); al(&s->d_dB, aN); al(&s->d_V, aN * s->nspin); al(&s->d_U, aN); al(&s->d_Vexc, aN);
    al(&s->d_va, aN); al(&s->d_vb, aN); al(&s->d_eexc, aN); al(&s->d_newDensity, aN * s->nspin); al(&s->d_integrands, aN * 5);
    al(&s->d_integrals, (size_t)natoms * 5); al(&s->d_records, (size_t)natoms * DFTA_RECORD_DOUBLES);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_atoms), sizeof(AtomState) * natoms);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_Z), sizeof(int) * natoms);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_fin), sizeof(int) * natoms);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_fin, 0, sizeof(int) * natoms, st);
    for (auto& ev : s->ev) if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_atoms, s->h_atoms.data(), sizeof(AtomState) * natoms, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(s->d_Z, Z, sizeof(int) * natoms, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(s->d_records, 0, sizeof(double) * natoms * DFTA_RECORD_DOUBLES, st);
    if (e != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "scf alloc: %s", hipGetErrorString(e));
        dfta_scf_destroy(s);
        return DFTA_ERR_HIP;
    }
    // DFTAtom.cpp:371-392 / 874-904: flat density, Poisson, v_xc, start potential
    const dim3 grid(std::min(64, (N + 255) / 256), natoms), block(256);
    hipLaunchKernelGGL(k_init_density, grid, block, 0, st, s->d_atoms, s->lsda, N, g->Rmax, s->d_density, s->d_dA, s->d_dB);
    rc = dfta_poisson_solve_launch(s->poisson, s->d_Z, s->d_density, s->d_U, nullptr, nullptr, nullptr);
    if (!rc) rc = dfta_poisson_finish(s->poisson, s->d_Z, s->d_density, s->d_U, nullptr, nullptr, nullptr);
    if (!rc) rc = scf_xc(s);
    if (rc) { dfta_scf_destroy(s); return rc; }
    hipLaunchKernelGGL(k_potential, grid, block, 0, st, s->d_atoms, s->lsda, N, g->d_r, s->d_U, s->d_Vexc, s->d_va, s->d_vb, s->d_V);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "scf init: %s", hipGetErrorString(e));
        dfta_scf_destroy(s);
        return DFTA_ERR_HIP;
    }
    unsigned long long dummy;
    rc = dfta_poisson_take_vcycles(s->poisson, &dummy);
    if (rc) { dfta_scf_destroy(s); return rc; }
    s->debug_levels = dfta_knob("DEBUG_LEVELS") != nullptr;
    s->h_frozen.assign(specs.size(), 0);
    *out = s;
    return DFTA_OK;
}

int dfta_scf_step(dfta_scf* s, dfta_step_stats* stats)
{
    if (!s) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = s->ctx;
    DFTA_ENTER(ctx);
    const dfta_grid* g = s->g;
    const int N = g->N, natoms = s->natoms;
    hipStream_t st = ctx->stream;
    const dim3 grid(std::min(64, (N + 255) / 256), natoms), block(256);

    DFTA_HIP(ctx, hipEventRecord(s->ev[0], st));
    DFTA_HIP(ctx, hipMemsetAsync(s->d_newDensity, 0, sizeof(double) * (size_t)s->nV * N, st));
    dfta::LevelStats ls;
    // bracket starts: CHAINED hands E-3 from level to level exactly as DFTAtom.cpp:541 (levels one after the other);
    // BATCHED starts every level concurrently from max(-Z^2-1, min Veff_l) (LevelSolver::clamp_bottoms)
    const int run_mode = s->levels_mode;
    // atoms that have met the reference's stop test (DFTAtom.cpp:474-479) are frozen: no level search, no mixing, no
    // Poisson solve, no new energies -- every atom of a batch ends in the state of its own last step
    bool any_frozen = false;
    for (int a = 0; a < natoms; ++a) {
        const AtomState& as = s->h_atoms[a];
        for (int k = as.job_off; k < as.job_end; ++k) s->h_frozen[k] = as.finished ? 1 : 0;
        any_frozen = any_frozen || as.finished;
    }
    dfta_range* r_levels = new dfta_range("dfta: level search (LoopOverLevels: all subshells of the batch)");
    int rc = s->solver.run(s->d_V, s->h_job_bottom.data(), run_mode, s->d_newDensity, stats ? &ls : nullptr,
                           any_frozen ? s->h_frozen.data() : nullptr);
    delete r_levels;
    if (rc) return rc;
    {
        std::vector<dfta::Job>& jobs = s->h_jobs;
        rc = s->solver.fetch_jobs(jobs);
        if (rc) return rc;
        for (size_t k = 0; k < jobs.size(); ++k) {
            s->h_job_bottom[k] = s->h_bottom0[jobs[k].v];
        }
        s->steps_done++;
        if (s->debug_levels) {
            for (size_t k = 0; k < jobs.size(); ++k)
                fprintf(stderr, "step %d job %zu n%d l%d: trust %d %d %d  len %d %d %d  pred_len %d %d %d  sweeps %d %d\n", s->steps_done, k, jobs[k].n,
                        jobs[k].l, jobs[k].trust[0], jobs[k].trust[1], jobs[k].trust[2], jobs[k].cur_len[0], jobs[k].cur_len[1],
                        jobs[k].cur_len[2], jobs[k].pred_len[0], jobs[k].pred_len[1], jobs[k].pred_len[2], jobs[k].n_count, jobs[k].n_zero);
        }
    }
    hipLaunchKernelGGL(k_mix, grid, block, 0, st, s->lsda, N, s->alpha, 1. - s->alpha, g->d_fpr2, s->d_newDensity, s->d_density,
                       s->d_dA, s->d_dB, s->d_fin);
    DFTA_CHECK_LAUNCH(ctx);
    DFTA_HIP(ctx, hipEventRecord(s->ev[1], st));
    // live atoms of this step (finished ones are frozen, see above) and the solver of their size class
    std::vector<int> live_atoms;
    for (int a = 0; a < natoms; ++a) if (!s->h_atoms[a].finished) live_atoms.push_back(a);
    const int nlive = static_cast<int>(live_atoms.size());
    dfta_poisson* pl = nullptr;
    int cls_atoms = 0;
    if (nlive > 0 && dfta_knob("SCF_NOLIVE") == nullptr) {
        for (int c = 0; c < kLiveClasses && !pl; ++c) {
            if (kLiveClassAtoms[c] < nlive) continue;
            if (kLiveClassAtoms[c] >= natoms) break;              // the batch's own solver is of that class already
            dfta_scf::LiveSolver& ls = s->live_solver[c];
            if (!ls.tried) {
                ls.tried = true;
                int Gbig = 0;
                dfta_poisson_group_state(s->poisson, &Gbig, nullptr, nullptr);
                // an optimisation only: if the class solver cannot be made (memory), the batch's own solver does the work --
                // the level solve and the mixing of this step have already run, so nothing may fail here
                rc = dfta_poisson_create_ex(ctx, g, kLiveClassAtoms[c], dfta_poisson_mode(s->poisson), &ls.p);
                if (rc) { ls.p = nullptr; ctx->err[0] = 0; (void)hipGetLastError(); rc = DFTA_OK; }
                if (ls.p) {
                    dfta_poisson_group_state(ls.p, &ls.G, nullptr, nullptr);
                    if (ls.G <= Gbig) { dfta_poisson_destroy(ls.p); ls.p = nullptr; }     // nothing to gain on this grid
                }
            }
            if (ls.p) { pl = ls.p; cls_atoms = kLiveClassAtoms[c]; }
            break;                                                  // the smallest class that holds the live atoms, or the batch's solver
        }
    }
    bool use_live = pl != nullptr;
    std::vector<int> h_live;
    if (use_live && s->live_cap < cls_atoms) {
        for (void* q : {(void*)s->d_liveZ, (void*)s->d_liveRho, (void*)s->d_liveU}) if (q) (void)hipFree(q);
        s->d_liveZ = nullptr; s->d_liveRho = nullptr; s->d_liveU = nullptr;
        s->live_cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->d_liveZ), sizeof(int) * 3 * cls_atoms);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_liveRho), sizeof(double) * (size_t)cls_atoms * N);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_liveU), sizeof(double) * (size_t)cls_atoms * N);
        if (e == hipSuccess) s->live_cap = cls_atoms;
        else {                                             // no room for the gathered copies: the batch's own solver (same bits)
            for (void* q : {(void*)s->d_liveZ, (void*)s->d_liveRho, (void*)s->d_liveU}) if (q) (void)hipFree(q);
            s->d_liveZ = nullptr; s->d_liveRho = nullptr; s->d_liveU = nullptr;
            (void)hipGetLastError();
            use_live = false;
            pl = nullptr;
        }
    }
    if (use_live) {
        const int cap = s->live_cap;
        h_live.assign(3 * cap, 0);
        for (int i = 0; i < cap; ++i) {
            const bool on = i < nlive;
            h_live[i] = on ? s->h_atoms[live_atoms[i]].Z : 1;
            h_live[cap + i] = on ? 0 : 1;
            h_live[2 * cap + i] = on ? live_atoms[i] : 0;
        }
        DFTA_HIP(ctx, hipMemcpyAsync(s->d_liveZ, h_live.data(), sizeof(int) * h_live.size(), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_gather_rows, dim3(std::min(256, (N + 255) / 256), nlive), dim3(256), 0, st, s->d_density, s->d_liveZ + 2 * cap, N, s->d_liveRho);
        DFTA_CHECK_LAUNCH(ctx);
    }
    {
        dfta_range r_poisson("dfta: multigrid Poisson solve (FullCycle)");
        dfta_poisson* ps = use_live ? pl : s->poisson;
        const int* pZ = use_live ? s->d_liveZ : s->d_Z;
        const double* pRho = use_live ? s->d_liveRho : s->d_density;
        double* pU = use_live ? s->d_liveU : s->d_U;
        const int* pSkip = use_live ? s->d_liveZ + s->live_cap : s->d_fin;
        rc = dfta_poisson_solve_launch(ps, pZ, pRho, pU, nullptr, nullptr, pSkip);
        if (rc) return rc;
        DFTA_HIP(ctx, hipEventRecord(s->ev[2], st));   // ev[1]..ev[2] brackets exactly the persistent multigrid kernel
        // synchronises and inspects the group barriers' abort flag on EVERY step; an aborted solve is repeated with one
        // workgroup per atom before anything reads U
        rc = dfta_poisson_finish(ps, pZ, pRho, pU, nullptr, nullptr, pSkip);
        if (rc) return rc;
        if (use_live) {
            hipLaunchKernelGGL(k_scatter_rows, dim3(std::min(256, (N + 255) / 256), nlive), dim3(256), 0, st, s->d_liveU, s->d_liveZ + 2 * s->live_cap, N, s->d_U);
            DFTA_CHECK_LAUNCH(ctx);
        }
    }
    dfta_range r_tail("dfta: XC + integrands + ordered integrals + energies");
    rc = scf_xc(s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tail, grid, block, 0, st, s->d_atoms, s->lsda, N, g->d_r, g->d_cnst, s->d_density, s->d_dA, s->d_dB, s->d_U,
                       s->d_Vexc, s->d_va, s->d_vb, s->d_eexc, s->d_V, s->d_integrands, g->uniform);
    DFTA_CHECK_LAUNCH(ctx);
    // Simpson38(1, .) on the logarithmic grid (DFTAtom.cpp:459-467), Simpson38(h, .) on the uniform one (DFTAtom.cpp:167-177)
    // (tolerance mode of the sweeps: the two sums of Simpson 3/8 in parallel, as the normalisation integral of scan_match takes them --
    // 0.44 ms of ordered additions per step otherwise; the energies move by a few 1e-16 relative)
    if (s->solver.sweep_mode == DFTA_SWEEPS_TOLERANCE && s->integ_rule == DFTA_INT_SIMPSON38 && !dfta_knob("SCF_ORDERED_SUMS"))
        rc = dfta_launch_integrate_simpson38_parallel(ctx, g->uniform ? g->h : 1.0, s->d_integrands, N, natoms * 5, (size_t)N, s->d_integrals);
    else
        rc = dfta_launch_integrate_ordered(ctx, s->integ_rule, g->uniform ? g->h : 1.0, s->d_integrands, N, natoms * 5, (size_t)N, s->d_integrals);
    if (rc) return rc;
    hipLaunchKernelGGL(k_energies, dim3((natoms + 63) / 64), dim3(64), 0, st, s->d_atoms, natoms, s->solver.d_jobs, s->d_integrals,
                       s->d_records, s->d_fin);
    DFTA_CHECK_LAUNCH(ctx);
    DFTA_HIP(ctx, hipEventRecord(s->ev[3], st));
    // the step ends synchronised with the atoms' state on the host: the next step freezes what has finished
    DFTA_HIP(ctx, hipMemcpyAsync(s->h_atoms.data(), s->d_atoms, sizeof(AtomState) * natoms, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    if (stats) {
        // the caller's struct may be an older (shorter) or a newer (longer) one: fill this library's, hand over what both know
        dfta_step_stats* const user = stats;
        const int user_size = user->struct_size;
        DFTA_REQUIRE(ctx, user_size >= (int)(4 * sizeof(int)) && user_size % (int)sizeof(int) == 0 && user_size <= 4096,
                     "dfta_step_stats::struct_size (set it to sizeof(dfta_step_stats) before the call)");
        dfta_step_stats mine;
        stats = &mine;
        DFTA_HIP(ctx, hipEventSynchronize(s->ev[3]));
        memset(stats, 0, sizeof(*stats));
        stats->struct_size = (int)sizeof(dfta_step_stats);
        stats->levels_fallbacks = (s->solver.scan_fallbacks + s->solver.persist_fallbacks) - s->fallbacks_seen;
        s->fallbacks_seen = s->solver.scan_fallbacks + s->solver.persist_fallbacks;
        DFTA_HIP(ctx, hipEventElapsedTime(&stats->ms_levels, s->ev[0], s->ev[1]));
        DFTA_HIP(ctx, hipEventElapsedTime(&stats->ms_poisson, s->ev[1], s->ev[2]));
        DFTA_HIP(ctx, hipEventElapsedTime(&stats->ms_tail, s->ev[2], s->ev[3]));
        stats->ms_poisson_kernel = stats->ms_poisson;
        stats->ms_sweep_kernels = ls.ms_sweep;
        stats->rounds = ls.rounds;
        stats->levels_layout = ls.layout;
        dfta_poisson_group_state(use_live ? pl : s->poisson, &stats->poisson_groups, nullptr, nullptr);
        stats->sweeps_issued = ls.sweeps_issued;
        stats->points_traversed = ls.points_traversed;
        long ref = 0;
        long long pts = 0, skipped = 0;
        for (const auto& j : s->h_jobs) if (!j.frozen) {
            ref += j.n_count + j.n_zero + 1;   // + the matched solve of each level
            pts += j.n_points;
            if (j.nodes == 0) skipped += j.cur_len[1];     // the second bisection of a node-less level is pure arithmetic
            skipped += j.n_fixed;                            // a third bisection on its fixed point (walk_job)
        }
        stats->sweeps_reference = ref;
        stats->sweeps_reference_executed = ref - (long)skipped;
        stats->points_reference = (long)pts;
        unsigned long long vc = 0;
        rc = dfta_poisson_take_vcycles(s->poisson, &vc);
        if (rc) return rc;
        for (auto& ls : s->live_solver) if (ls.p) {
            unsigned long long vl = 0;
            rc = dfta_poisson_take_vcycles(ls.p, &vl);
            if (rc) return rc;
            vc += vl;
        }
        stats->vcycles = (long)vc;
        memcpy(user, &mine, std::min<size_t>((size_t)user_size, sizeof(mine)));
        user->struct_size = user_size;
    }
    return DFTA_OK;
}

int dfta_scf_get_energies(dfta_scf* s, dfta_energies* e, int* finished)
{
    if (!s) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = s->ctx;
    DFTA_ENTER(ctx);
    for (int a = 0; a < s->natoms; ++a) {      // h_atoms is current: every step ends with its copy
        if (e) e[a] = s->h_atoms[a].e;
        if (finished) finished[a] = s->h_atoms[a].finished;
    }
    return DFTA_OK;
}

int dfta_scf_info(const dfta_scf* s, int* tree_depth, int* njobs, long* trials_per_round)
{
    if (!s) return DFTA_ERR_INVALID;
    if (tree_depth) *tree_depth = s->solver.depth;
    if (njobs) *njobs = s->solver.njobs;
    if (trials_per_round) *trials_per_round = s->solver.ntrials;
    return DFTA_OK;
}

int dfta_scf_set_integrator(dfta_scf* s, int rule)
{
    if (!s) return DFTA_ERR_INVALID;
    DFTA_REQUIRE(s->ctx, dfta_integral_shape_ok(rule, s->g->N), "integration rule / grid size");
    s->integ_rule = rule;
    s->solver.integ_rule = rule;
    return DFTA_OK;
}

int dfta_scf_poisson_info(const dfta_scf* s, int* G, int* degraded, int* aborts)
{
    if (!s) return DFTA_ERR_INVALID;
    return dfta_poisson_group_state(s->poisson, G, degraded, aborts);
}

int dfta_scf_num_levels(const dfta_scf* s, int atom, int spin)
{
    if (!s || atom < 0 || atom >= s->natoms || spin < 0 || spin > 1) return -1;
    return s->spin_nlev[spin][atom];
}

int dfta_scf_get_levels(dfta_scf* s, int atom, int spin, int* n, int* l, int* occ, double* E, int* converged)
{
    if (!s) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = s->ctx;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, atom >= 0 && atom < s->natoms && spin >= 0 && spin < s->nspin, "atom/spin");
    const std::vector<dfta::Job>& jobs = s->h_jobs;
    DFTA_REQUIRE(ctx, !jobs.empty(), "no SCF step has run yet");
    int k0 = s->h_atoms[atom].job_off + (spin ? s->spin_nlev[0][atom] : 0);
    const int cnt = s->spin_nlev[spin][atom];
    for (int k = 0; k < cnt; ++k) {
        const dfta::Job& j = jobs[k0 + k];
        if (n) n[k] = j.n;
        if (l) l[k] = j.l;
        if (occ) occ[k] = j.occ;
        if (E) E[k] = j.E;
        if (converged) converged[k] = j.converged;
    }
    return DFTA_OK;
}

int dfta_scf_get_level_status(dfta_scf* s, int atom, int spin, int* status, int* n_count, int* n_zero)
{
    if (!s) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = s->ctx;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, atom >= 0 && atom < s->natoms && spin >= 0 && spin < s->nspin, "atom/spin");
    const std::vector<dfta::Job>& jobs = s->h_jobs;
    DFTA_REQUIRE(ctx, !jobs.empty(), "no SCF step has run yet");
    const int k0 = s->h_atoms[atom].job_off + (spin ? s->spin_nlev[0][atom] : 0);
    for (int k = 0; k < s->spin_nlev[spin][atom]; ++k) {
        if (status) status[k] = jobs[k0 + k].status;
        if (n_count) n_count[k] = jobs[k0 + k].n_count;
        if (n_zero) n_zero[k] = jobs[k0 + k].n_zero;
    }
    return DFTA_OK;
}

int dfta_scf_get_array(dfta_scf* s, int atom, int which, double* out)
{
    if (!s) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = s->ctx;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, atom >= 0 && atom < s->natoms && out, "atom/out");
    const int N = s->g->N;
    const double* src = nullptr;
    switch (which) {
    case 0: src = s->d_density + (size_t)atom * N; break;
    case 1: src = (s->lsda ? s->d_dA : s->d_density) + (size_t)atom * N; break;
    case 2: src = (s->lsda ? s->d_dB : s->d_density) + (size_t)atom * N; break;
    case 3: src = s->d_V + (size_t)atom * s->nspin * N; break;
    case 4: src = s->d_V + ((size_t)atom * s->nspin + (s->lsda ? 1 : 0)) * N; break;
    case 5: src = s->d_U + (size_t)atom * N; break;
    default: DFTA_REQUIRE(ctx, false, "which");
    }
    DFTA_HIP(ctx, hipMemcpyAsync(out, src, sizeof(double) * N, hipMemcpyDeviceToHost, ctx->stream));
    DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DFTA_OK;
}

int dfta_scf_get_records_dev(dfta_scf* s, double* dRecords)
{
    if (!s || !dRecords) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = s->ctx;
    DFTA_ENTER(ctx);
    DFTA_HIP(ctx, hipMemcpyAsync(dRecords, s->d_records, sizeof(double) * (size_t)s->natoms * DFTA_RECORD_DOUBLES,
                                 hipMemcpyDeviceToDevice, ctx->stream));
    return DFTA_OK;
}

}  // extern "C"
