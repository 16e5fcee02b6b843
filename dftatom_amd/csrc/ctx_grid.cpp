// ctx_grid.cpp -- context, grid tables and the integer-only Aufbau helper of libdftatom_hip.
//
// Grid tables replace the per-point exp() calls of NumerovFunctionNonUniformGrid (Numerov.h:76-101,
// 181-184) and of the SCF pointwise loops (DFTAtom.cpp:42,47,334,439-442; PoissonSolver.h:66-74).
// They are evaluated on the host with libm in exactly the reference's expression order, so device
// kernels work from bit-identical f(i) inputs and never call exp() on grid quantities.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <algorithm>

#include "common.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

extern "C" {

const char* dfta_version(void) { return "dftatom_amd 0.1 (gfx950)"; }

int dfta_ctx_create(int device, void* hip_stream, dfta_ctx** out)
{
    if (!out) return DFTA_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DFTA_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return DFTA_ERR_NO_DEVICE;
    dfta_ctx* c = new dfta_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return DFTA_ERR_NO_DEVICE; }
    c->num_cu = prop.multiProcessorCount;
    snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
    if (hip_stream) {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream);
        c->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return DFTA_ERR_NO_DEVICE; }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev[0]) != hipSuccess || hipEventCreate(&c->ev[1]) != hipSuccess) { delete c; return DFTA_ERR_NO_DEVICE; }
    if (const char* e = dfta_knob("SWEEP_KERNEL"))
        c->sweep_kernel = !strcmp(e, "fused") ? DFTA_SWEEP_FUSED : (!strcmp(e, "pipe") ? DFTA_SWEEP_PIPELINED : DFTA_SWEEP_AUTO);
    *out = c;
    return DFTA_OK;
}

int dfta_ctx_set_sweep_kernel(dfta_ctx* ctx, int which)
{
    if (!ctx) return DFTA_ERR_INVALID;
    if (which < DFTA_SWEEP_AUTO || which > DFTA_SWEEP_PIPELINED) {
        snprintf(ctx->err, sizeof(ctx->err), "dfta_ctx_set_sweep_kernel: unknown kernel %d", which);
        return DFTA_ERR_INVALID;
    }
    ctx->sweep_kernel = which;
    return DFTA_OK;
}

int dfta_ctx_last_kernel_ms(dfta_ctx* ctx, float* ms)
{
    if (!ctx || !ms) return DFTA_ERR_INVALID;
    DFTA_REQUIRE(ctx, ctx->have_kernel_time, "no timed kernel yet");
    DFTA_HIP(ctx, hipEventSynchronize(ctx->ev[1]));
    DFTA_HIP(ctx, hipEventElapsedTime(ms, ctx->ev[0], ctx->ev[1]));
    return DFTA_OK;
}

void dfta_ctx_destroy(dfta_ctx* ctx)
{
    if (!ctx) return;
    for (hipEvent_t e : ctx->ev) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int dfta_ctx_synchronize(dfta_ctx* ctx)
{
    if (!ctx) return DFTA_ERR_INVALID;
    DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DFTA_OK;
}

const char* dfta_last_error(const dfta_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int dfta_ctx_device_info(const dfta_ctx* ctx, int* num_cu, char* name, int name_cap)
{
    if (!ctx) return DFTA_ERR_INVALID;
    if (num_cu) *num_cu = ctx->num_cu;
    if (name && name_cap > 0) { strncpy(name, ctx->name, static_cast<size_t>(name_cap) - 1); name[name_cap - 1] = 0; }
    return DFTA_OK;
}

int dfta_num_nodes(int mg_levels)   // PoissonSolver.h:127-135 with Ncoarse = 3
{
    int size = 3;
    for (int i = 0; i < mg_levels - 1; ++i) size = size * 2 - 1;
    return size;
}

// smallest double a with exp(a) >= 1e-200 (libm); the reference's cut-off test `exp(arg) < 1E-200`
// (Numerov.h:129) is then `arg < a` for a monotone exp.
static double far_threshold()
{
    auto key = [](double x) { int64_t b; memcpy(&b, &x, 8); return b < 0 ? INT64_MIN - b : b; };   // monotone map
    auto unkey = [](int64_t k) { int64_t b = k < 0 ? INT64_MIN - k : k; double x; memcpy(&x, &b, 8); return x; };
    int64_t lo = key(-461.0), hi = key(-460.0);   // exp(lo) < 1e-200 <= exp(hi)
    while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (exp(unkey(mid)) < 1E-200) lo = mid; else hi = mid;
    }
    return unkey(hi);
}

int dfta_grid_create(dfta_ctx* ctx, int mg_levels, double delta, double Rmax, dfta_grid** out)
{
    if (!ctx || !out) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, mg_levels >= 3 && mg_levels <= 24 && delta > 0 && Rmax > 0, "grid parameters");
    dfta_grid* g = new dfta_grid();
    g->ctx = ctx;
    g->levels = mg_levels;
    const int N = dfta_num_nodes(mg_levels);
    g->N = N;
    g->delta = delta;
    g->Rmax = Rmax;
    // Numerov.h:79-86
    g->Rp = Rmax / (exp((static_cast<double>(N) - 1.) * delta) - 1.);
    const double Rp2 = g->Rp * g->Rp;
    g->twodelta = 2. * delta;
    const double delta2 = delta * delta;
    g->Rp2delta2 = Rp2 * delta2;
    g->delta2p4 = delta2 * 0.25;
    g->far_arg_threshold = far_threshold();

    std::vector<double>&r = g->h_r, &e1 = g->h_e1, &e2 = g->h_e2, &eh = g->h_eh;
    r.resize(N); e1.resize(N); e2.resize(N); eh.resize(N);
    std::vector<double> cl(static_cast<size_t>(4) * N, 0.0), cnst(N), psrc(N), fpr2(N);
    const double fourM_PI = 4. * M_PI;
    const double fourM_PIRp2delta2 = fourM_PI * g->Rp2delta2;   // PoissonSolver.h:64-70 (Rp*Rp*delta2grid == Rp2*delta2)
    for (int i = 0; i < N; ++i) {
        e1[i] = exp(static_cast<double>(i) * delta);              // Numerov.h:183, DFTAtom.cpp:334,439
        r[i] = g->Rp * (e1[i] - 1.);
        e2[i] = exp(static_cast<double>(i) * g->twodelta);         // Numerov.h:100, PoissonSolver.h:74
        eh[i] = exp(i * delta * 0.5);                               // DFTAtom.cpp:42
        cnst[i] = g->Rp * delta * e1[i];                            // DFTAtom.cpp:47,442
        psrc[i] = fourM_PIRp2delta2 * e2[i];                        // PoissonSolver.h:74
        fpr2[i] = fourM_PI * r[i] * r[i];                           // DFTAtom.cpp:340
        if (i > 0)
            for (unsigned l = 1; l < 4; ++l)
                cl[static_cast<size_t>(l) * N + i] = l * (l + 1.) / (r[i] * r[i]) * 0.5;   // Numerov.h:93
    }
    for (unsigned l = 0; l < 4; ++l)   // Numerov.h:110-116 at position = 1
        g->zero1[l] = pow(r[1], static_cast<double>(l) + 1) * exp(-1.0 * delta * 0.5);

    struct Up { double** d; const double* h; size_t n; } ups[] = {
        {&g->d_r, r.data(), (size_t)N}, {&g->d_e1, e1.data(), (size_t)N}, {&g->d_e2, e2.data(), (size_t)N},
        {&g->d_eh, eh.data(), (size_t)N}, {&g->d_cl, cl.data(), (size_t)4 * N}, {&g->d_cnst, cnst.data(), (size_t)N},
        {&g->d_psrc, psrc.data(), (size_t)N}, {&g->d_fpr2, fpr2.data(), (size_t)N}};
    for (auto& u : ups) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(u.d), u.n * sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(*u.d, u.h, u.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            snprintf(ctx->err, sizeof(ctx->err), "grid upload: %s", hipGetErrorString(e));
            dfta_grid_destroy(g);
            return DFTA_ERR_HIP;
        }
    }
    DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    g->d_rsrc = g->d_r;
    *out = g;
    return DFTA_OK;
}

// Uniform grid r_i = i h, h = MaxR / NumSteps (DFTAtom.cpp:66-68, NumerovFunctionRegularGrid Numerov.h:16-70).
int dfta_grid_create_uniform(dfta_ctx* ctx, int mg_levels, double Rmax, dfta_grid** out)
{
    if (!ctx || !out) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, mg_levels >= 3 && mg_levels <= 24 && Rmax > 0, "grid parameters");
    dfta_grid* g = new dfta_grid();
    g->ctx = ctx;
    g->levels = mg_levels;
    const int N = dfta_num_nodes(mg_levels);
    g->N = N;
    g->uniform = 1;
    g->delta = 0;
    g->Rmax = Rmax;
    g->Rp = 0;
    const long steps = N - 1;
    g->h = Rmax / steps;                         // Numerov.h:276 (h = startPoint / steps), DFTAtom.cpp:68
    g->h2 = g->h * g->h;
    g->h2p12 = g->h2 / 12.;
    g->Rp2delta2 = 1; g->delta2p4 = 0; g->twodelta = 0;
    g->far_arg_threshold = far_threshold();
    std::vector<double>&r = g->h_r, &e1 = g->h_e1, &e2 = g->h_e2, &eh = g->h_eh;
    r.resize(N); e1.assign(N, 1.0); e2.assign(N, 1.0); eh.assign(N, 1.0);
    std::vector<double> cl(static_cast<size_t>(4) * N, 0.0), cnst(N, 1.0), psrc(N), fpr2(N), rsrc(N);
    const double fourM_PI = 4. * M_PI;
    {
        // PoissonSolver::FillR(Source, 0, maxRadius) (PoissonSolver.cpp:200-210) and the factor of PoissonSolver.h:26-40
        const size_t Nn = static_cast<size_t>(N) - 1;
        for (size_t i = 0; i < static_cast<size_t>(N); ++i) rsrc[i] = (0.0 * (Nn - i) + Rmax * i) / Nn;
        const double dl = rsrc[1] - rsrc[0];
        const double delta2fourM_PI = (dl * dl) * fourM_PI;
        for (int i = 0; i < N; ++i) psrc[i] = delta2fourM_PI;
    }
    for (int i = 0; i < N; ++i) {
        r[i] = g->h * i;                                           // Numerov.h:313 (position = h * i), DFTAtom.cpp:102,126
        fpr2[i] = fourM_PI * r[i] * r[i];                          // DFTAtom.cpp:132
        if (i > 0)
            for (unsigned l = 1; l < 4; ++l)
                cl[static_cast<size_t>(l) * N + i] = l * (l + 1.) / (r[i] * r[i]) * 0.5;   // Numerov.h:23
    }
    for (unsigned l = 0; l < 4; ++l) g->zero1[l] = pow(g->h, static_cast<double>(l) + 1.);   // Numerov.h:38-41 at position = h (sweeps keep h; the match solve re-derives it)
    struct Up { double** d; const double* h; size_t n; } ups[] = {
        {&g->d_r, r.data(), (size_t)N}, {&g->d_e1, e1.data(), (size_t)N}, {&g->d_e2, e2.data(), (size_t)N},
        {&g->d_eh, eh.data(), (size_t)N}, {&g->d_cl, cl.data(), (size_t)4 * N}, {&g->d_cnst, cnst.data(), (size_t)N},
        {&g->d_psrc, psrc.data(), (size_t)N}, {&g->d_fpr2, fpr2.data(), (size_t)N}, {&g->d_rsrc, rsrc.data(), (size_t)N}};
    for (auto& u : ups) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(u.d), u.n * sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(*u.d, u.h, u.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            snprintf(ctx->err, sizeof(ctx->err), "grid upload: %s", hipGetErrorString(e));
            dfta_grid_destroy(g);
            return DFTA_ERR_HIP;
        }
    }
    DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = g;
    return DFTA_OK;
}

void dfta_grid_destroy(dfta_grid* g)
{
    if (!g) return;
    if (g->d_rsrc == g->d_r) g->d_rsrc = nullptr;
    double* ptrs[] = {g->d_r, g->d_e1, g->d_e2, g->d_eh, g->d_cl, g->d_cnst, g->d_psrc, g->d_fpr2, g->d_rsrc};
    for (double* p : ptrs) if (p) (void)hipFree(p);
    delete g;
}

int dfta_grid_is_uniform(const dfta_grid* g) { return g ? g->uniform : 0; }

int dfta_grid_num_nodes(const dfta_grid* g) { return g ? g->N : 0; }
double dfta_grid_rp(const dfta_grid* g) { return g ? g->Rp : 0.0; }
int dfta_grid_get_r(const dfta_grid* g, double* r_host)
{
    if (!g || !r_host) return DFTA_ERR_INVALID;
    memcpy(r_host, g->h_r.data(), sizeof(double) * g->h_r.size());
    return DFTA_OK;
}

// ---- Aufbau (integer only; AufbauPrinciple.h:36-75,101-117 and DFTAtom.cpp:367, 611-638) -----------------
static void adjust_f_block(int& nrElectrons, int Z, int N, int L)
{
    if (L == 3) {
        if ((Z == 57 || Z == 58 || Z == 64) && N == 3) --nrElectrons;      // La, Ce, Gd: one 4f electron goes to 5d
        else if (N == 4) {
            if (Z == 89 || Z == 90) nrElectrons = 0;                        // Ac, Th
            else if (Z == 91 || Z == 92 || Z == 93 || Z == 96) --nrElectrons; // Pa, U, Np, Cm
        }
    } else if (Z == 103 && N == 5 && L == 2) nrElectrons = 0;              // Lr
}

// AufbauPrinciple.h:78-99,119-127: the s shell under a d shell that the Madelung rule leaves one (Pd: two) short hands
// the electron(s) over.  The reference defines this adjustment but never calls it (Cr comes out as 3d4 4s2); here it is an
// option, applied once, after the clamp to the electrons that are left (an s shell is never clamped, so once is enough).
static void adjust_transition_metal(int& nrElectrons, int Z, int N, int L)
{
    if (L != 0) return;
    const bool one = (Z == 24 || Z == 29 || Z == 41 || Z == 42 || Z == 44 || Z == 45 || Z == 47 || Z == 78 || Z == 79);
    if (one) {
        const int outer_s = Z <= 29 ? 3 : (Z <= 47 ? 4 : 5);              // 4s, 5s, 6s
        if (N == outer_s) --nrElectrons;
    } else if (Z == 46 && N == 4) nrElectrons -= 2;                        // Pd: 4d10 5s0
}

int dfta_get_subshells(int Z, int* n, int* l, int* occ, int cap) { return dfta_get_subshells_ex(Z, DFTA_AUFBAU_REFERENCE, n, l, occ, cap); }

int dfta_get_subshells_ex(int Z, int aufbau, int* n, int* l, int* occ, int cap)
{
    if (Z < 1 || !n || !l || !occ) return -1;
    struct S { int n, l, occ; };
    std::vector<S> lv;
    int electronCount = 0;
    bool stop = false;
    for (int NplusL = 0; !stop && NplusL < 10; ++NplusL)
        for (int N = 0; N <= NplusL; ++N) {
            const int L = NplusL - N;
            if (L > N) continue;
            int e = 2 * (2 * L + 1);
            adjust_f_block(e, Z, N, L);
            if (Z - electronCount < e) e = Z - electronCount;
            adjust_f_block(e, Z, N, L);
            if (aufbau == DFTA_AUFBAU_TRANSITION_METALS) adjust_transition_metal(e, Z, N, L);
            if (e > 0) { electronCount += e; lv.push_back({N, L, e}); }
            if (electronCount == Z) { stop = true; break; }
        }
    std::sort(lv.begin(), lv.end(), [](const S& a, const S& b) { return a.n < b.n || (a.n == b.n && a.l < b.l); });
    if (static_cast<int>(lv.size()) > cap) return -1;
    for (size_t i = 0; i < lv.size(); ++i) { n[i] = lv[i].n; l[i] = lv[i].l; occ[i] = lv[i].occ; }
    return static_cast<int>(lv.size());
}

int dfta_split_spin(int Z, int* nA, int* nB, int* an, int* al, int* aocc, int* bn, int* bl, int* bocc, int cap)
{
    return dfta_split_spin_ex(Z, DFTA_AUFBAU_REFERENCE, nA, nB, an, al, aocc, bn, bl, bocc, cap);
}

int dfta_split_spin_ex(int Z, int aufbau, int* nA, int* nB, int* an, int* al, int* aocc, int* bn, int* bl, int* bocc, int cap)
{
    if (!nA || !nB) return DFTA_ERR_INVALID;
    const int cnt = dfta_get_subshells_ex(Z, aufbau, an, al, aocc, cap);
    if (cnt < 0) return DFTA_ERR_INVALID;
    int m = 0;
    for (int i = 0; i < cnt; ++i) {
        const int maxe = 2 * al[i] + 1;
        int b;
        if (aocc[i] >= maxe) { b = aocc[i] - maxe; aocc[i] = maxe; }
        else b = 0;
        if (b != 0) { bn[m] = an[i]; bl[m] = al[i]; bocc[m] = b; ++m; }
    }
    *nA = cnt;
    *nB = m;
    return DFTA_OK;
}

}  // extern "C"
