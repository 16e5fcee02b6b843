// xc.hip -- Vosko-Wilk-Nusair exchange-correlation, LDA and LSDA, as coalesced pointwise kernels.
//
// Replaces VWNExchCor::Vexc / eexcDif (VWNExcCor.h:73-128, LDA) and the spin-polarised pair
// (VWNExcCor.h:134-312, LSDA) with ExcCorBase::f / df (ExcCorBase.h:14-26).  Expressions are written in the
// reference's operation order; pow/log/atan come from the ROCm device library, so results agree with the
// glibc-based reference to rounding of those functions (parity tolerance: 1e-13 relative, see tests).
// HBM-bound by construction: LDA reads 8 B and writes 16 B per point, LSDA reads 16 B and writes 32 B.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "internal.h"
#include "xc.h"

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kFourPi = 4. * kPi;
constexpr double kThird = 1. / 3.;                                  // ExcCorBase.h:12

// One Pade-like fit of VWN: eps(y) = A { ln(y^2/Y) + 2b/Q atan(Q/(2y+b)) - b y0/Y0 [ ln((y-y0)^2/Y) + 2(b+2y0)/Q atan(Q/(2y+b)) ] },
// y = sqrt(rs), Y(y) = y^2 + b y + c, Q = sqrt(4c - b^2).  Three parameter sets (VWNExcCor.h:23-41): paramagnetic,
// ferromagnetic, and the spin stiffness alpha_c.
struct VwnFit { double A, y0, b, c; };
constexpr VwnFit kPara{0.0310907, -0.10498, 3.72744, 12.93532};
constexpr VwnFit kFerro{0.01554535, -0.325, 7.06042, 18.0578};
constexpr VwnFit kStiff{-1. / (6. * kPi * kPi), -0.0047584, 1.13107, 13.0045};
constexpr double poly_at(const VwnFit& p, double y) { return y * y + p.b * y + p.c; }

struct FitValue { double eps, slope; };     // eps: the fit itself (VWNExcCor.h:43-50); slope: its rs-derivative term (VWNExcCor.h:52-55)

// NESTED selects how Y(y) is rounded: the LDA routines write y*y + b*y + c (VWNExcCor.h:89,119), the LSDA ones y*(y + b) + c
// (VWNExcCor.h:182,187,192) -- both kept so that each path rounds like the reference's
template <bool NESTED>
__device__ __forceinline__ FitValue eval_fit(const VwnFit p, double y)
{
    const double Y = NESTED ? y * (y + p.b) + p.c : y * y + p.b * y + p.c;
    const double Y0 = poly_at(p, p.y0);
    const double dy = y - p.y0;
    const double Q = sqrt(4 * p.c - p.b * p.b);
    const double at = atan(Q / (2. * y + p.b));
    FitValue v;
    v.eps = p.A * (log(y * y / Y) + 2. * p.b / Q * at - p.b * p.y0 / Y0 * (log(dy * dy / Y) + 2. * (p.b + 2. * p.y0) / Q * at));
    v.slope = p.A * (p.c * dy - p.b * p.y0 * y) / (dy * Y);
    return v;
}

__device__ __forceinline__ double wigner_seitz(double rho) { return pow(3. / (kFourPi * rho), kThird); }   // rs, eq 2 of the NIST note

// LDA (VWNExcCor.h:73-128): v_xc and the double-counting term eps_xc - v_xc, both from one evaluation of the paramagnetic fit
__global__ void k_vwn_lda(const double* __restrict__ n, size_t sz, double* __restrict__ vexc, double* __restrict__ eexc, double cx)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < sz; i += (size_t)gridDim.x * blockDim.x) {
        const double rho = n[i];
        double v = 0., e = 0.;
        if (!(rho < 1E-18)) {                                         // VWNExcCor.h:82,112
            const double rs = wigner_seitz(rho);
            const FitValue c = eval_fit<false>(kPara, sqrt(rs));
            v = -cx / rs + c.eps - kThird * c.slope;                  // VWNExcCor.h:94-97
            e = (0.25 * cx) / rs + kThird * c.slope;                  // VWNExcCor.h:123-124
        }
        if (vexc) vexc[i] = v;
        if (eexc) eexc[i] = e;
    }
}

// LSDA (VWNExcCor.h:134-312).  With g(zeta) the spin interpolation of ExcCorBase.h:14-19, g''(0) = gdd and
//     w(zeta) = g / gdd * (1 + beta zeta^4),   beta = gdd (eps_F - eps_P) / alpha_c - 1
// the correlation energy is eps_P + alpha_c w (eqs 7-10 of the NIST note); the kernel forms its rs-derivative (`drs`) and
// its zeta-derivative (`dz`) and assembles the common term and the two spin potentials from them.
__global__ void k_vwn_lsda(const double* __restrict__ na, const double* __restrict__ nb, size_t sz, double* __restrict__ res,
                           double* __restrict__ va, double* __restrict__ vb, double* __restrict__ eexc, double cx, double cbrt2)
{
    const double gdd = 4. / (9. * (cbrt2 - 1.));
    const double gscale = 1. / (2. * (cbrt2 - 1.)), dgscale = 2. / (3. * (cbrt2 - 1.));       // ExcCorBase.h:16,23
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < sz; i += (size_t)gridDim.x * blockDim.x) {
        const double up = na[i], dn = nb[i];
        const double tot = up + dn;
        double common = 0., vup = 0., vdn = 0., e = 0.;
        if (!(tot < 1E-18)) {                                         // VWNExcCor.h:160,260
            const double rs = wigner_seitz(tot);
            const double z = (up - dn) / tot;
            const double z3 = z * z * z, z4 = z3 * z;
            const double g = gscale * (pow(1. + z, 4. * kThird) + pow(1. - z, 4. * kThird) - 2.);
            const double dg = dgscale * (pow(1. + z, kThird) - pow(1. - z, kThird));
            const double y = sqrt(rs);
            const FitValue P = eval_fit<true>(kPara, y), F = eval_fit<true>(kFerro, y), S = eval_fit<true>(kStiff, y);

            const double gap = F.eps - P.eps;                                             // VWNExcCor.h:202
            const double beta = gdd * gap / S.eps - 1.;
            const double env = 1. + beta * z4;
            const double w = g / gdd * env;
            const double dbeta = gdd / S.eps * (F.slope - P.slope - S.slope * gap / S.eps);   // VWNExcCor.h:208
            const double dw = g / gdd * z4 * dbeta;
            const double drs = kThird * (P.slope + S.slope * w + S.eps * dw);             // VWNExcCor.h:212-213
            const double dz = S.eps / gdd * (4. * beta * z3 * g + env * dg);               // VWNExcCor.h:216

            const double xP = -cx / rs;                                                    // exchange of the unpolarised gas ...
            const double xF = cbrt2 * xP;                                                  // ... and of the fully polarised one
            common = P.eps + S.eps * w - drs;                                              // VWNExcCor.h:219-231
            vup = -(cx * cbrt2) / wigner_seitz(up) + common + (1. - z) * dz;               // VWNExcCor.h:233
            vdn = -(cx * cbrt2) / wigner_seitz(dn) + common - (1. + z) * dz;               // VWNExcCor.h:234
            common += (xP + (xF - xP) * g);                                                // VWNExcCor.h:236

            const double xPd = (0.25 * cx) / rs;                                           // VWNExcCor.h:271-272
            e = xPd + (cbrt2 * xPd - xPd) * g + drs;                                       // VWNExcCor.h:306-308
        }
        if (res) res[i] = common;
        if (va) va[i] = vup;
        if (vb) vb[i] = vdn;
        if (eexc) eexc[i] = e;
    }
}

// Chachiyo's one-line correlation (ExcCor.h:27-95; the reference keeps it next to VWN, all its call sites commented out:
// DFTAtom.cpp:383,412,421): eps_c = a ln(1 + b/rs + b/rs^2), a = (ln 2 - 1) / (2 pi^2).  LDA only, as in the reference.
__global__ void k_chachiyo_lda(const double* __restrict__ n, size_t sz, double* __restrict__ vexc, double* __restrict__ eexc, double cx,
                               double a, double b)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < sz; i += (size_t)gridDim.x * blockDim.x) {
        const double rho = n[i];
        double v = 0., e = 0.;
        if (!(rho < 1E-18)) {                                         // ExcCor.h:50,79
            const double rs = wigner_seitz(rho);
            const double q1 = b / rs, q2 = q1 / rs;
            const double tail = a / (1. + q1 + q2) * (q1 + 2. * q2) * rs / 3.;
            v = -cx / rs + a * log(1. + q1 + q1 / rs) - tail;         // ExcCor.h:60-63
            e = (0.25 * cx) / rs + tail;                              // ExcCor.h:89-91
        }
        if (vexc) vexc[i] = v;
        if (eexc) eexc[i] = e;
    }
}

}  // namespace

// the two irrational constants are evaluated once on the host with libm, as the reference does (VWNExcCor.h:75,139-140)
static double host_X1() { return pow(3. / (2. * kPi), 2. * kThird); }
static double host_X2() { return pow(2., kThird); }

int dfta_launch_chachiyo_lda(dfta_ctx* ctx, int improved, const double* dN, size_t sz, double* dVexc, double* dEexc)
{
    const double a = (M_LN2 - 1.) / (2. * kPi * kPi);                 // ExcCor.h:32
    const double b = improved ? 21.7392245 : 20.4562557;              // ExcCor.h:16,24
    const int blocks = (int)std::min<size_t>((sz + 255) / 256, 2048);
    hipLaunchKernelGGL(k_chachiyo_lda, dim3(blocks), dim3(256), 0, ctx->stream, dN, sz, dVexc, dEexc, host_X1(), a, b);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

int dfta_launch_vwn_lda(dfta_ctx* ctx, const double* dN, size_t sz, double* dVexc, double* dEexc)
{
    const int blocks = (int)std::min<size_t>((sz + 255) / 256, 2048);
    hipLaunchKernelGGL(k_vwn_lda, dim3(blocks), dim3(256), 0, ctx->stream, dN, sz, dVexc, dEexc, host_X1());
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

int dfta_launch_vwn_lsda(dfta_ctx* ctx, const double* dNa, const double* dNb, size_t sz, double* dRes, double* dVa, double* dVb,
                         double* dEexc)
{
    const int blocks = (int)std::min<size_t>((sz + 255) / 256, 2048);
    hipLaunchKernelGGL(k_vwn_lsda, dim3(blocks), dim3(256), 0, ctx->stream, dNa, dNb, sz, dRes, dVa, dVb, dEexc, host_X1(), host_X2());
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

extern "C" int dfta_vwn_lda(dfta_ctx* ctx, const double* n, size_t sz, double* vexc, double* eexcdif)
{
    if (!ctx) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, n && (vexc || eexcdif), "null input");
    if (sz == 0) return DFTA_OK;
    hipStream_t st = ctx->stream;
    DevBuf<double> dN, dV, dE;
    DFTA_HIP(ctx, dN.alloc(sz)); DFTA_HIP(ctx, dV.alloc(sz)); DFTA_HIP(ctx, dE.alloc(sz));
    DFTA_HIP(ctx, hipMemcpyAsync(dN.p, n, sizeof(double) * sz, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    int rc = dfta_launch_vwn_lda(ctx, dN.p, sz, dV.p, dE.p);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    ctx->have_kernel_time = true;
    if (vexc) DFTA_HIP(ctx, hipMemcpyAsync(vexc, dV.p, sizeof(double) * sz, hipMemcpyDeviceToHost, st));
    if (eexcdif) DFTA_HIP(ctx, hipMemcpyAsync(eexcdif, dE.p, sizeof(double) * sz, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return DFTA_OK;
}

extern "C" int dfta_vwn_lsda(dfta_ctx* ctx, const double* na, const double* nb, size_t sz, double* vexc, double* va, double* vb,
                             double* eexcdif)
{
    if (!ctx) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, na && nb, "null input");
    if (sz == 0) return DFTA_OK;
    hipStream_t st = ctx->stream;
    DevBuf<double> dA, dB, dR, dVa, dVb, dE;
    DFTA_HIP(ctx, dA.alloc(sz)); DFTA_HIP(ctx, dB.alloc(sz)); DFTA_HIP(ctx, dR.alloc(sz));
    DFTA_HIP(ctx, dVa.alloc(sz)); DFTA_HIP(ctx, dVb.alloc(sz)); DFTA_HIP(ctx, dE.alloc(sz));
    DFTA_HIP(ctx, hipMemcpyAsync(dA.p, na, sizeof(double) * sz, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dB.p, nb, sizeof(double) * sz, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    int rc = dfta_launch_vwn_lsda(ctx, dA.p, dB.p, sz, dR.p, dVa.p, dVb.p, dE.p);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    ctx->have_kernel_time = true;
    if (vexc) DFTA_HIP(ctx, hipMemcpyAsync(vexc, dR.p, sizeof(double) * sz, hipMemcpyDeviceToHost, st));
    if (va) DFTA_HIP(ctx, hipMemcpyAsync(va, dVa.p, sizeof(double) * sz, hipMemcpyDeviceToHost, st));
    if (vb) DFTA_HIP(ctx, hipMemcpyAsync(vb, dVb.p, sizeof(double) * sz, hipMemcpyDeviceToHost, st));
    if (eexcdif) DFTA_HIP(ctx, hipMemcpyAsync(eexcdif, dE.p, sizeof(double) * sz, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return DFTA_OK;
}

extern "C" int dfta_chachiyo_lda(dfta_ctx* ctx, int improved, const double* n, size_t sz, double* vexc, double* eexcdif)
{
    if (!ctx) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, n && (vexc || eexcdif), "null input");
    if (sz == 0) return DFTA_OK;
    hipStream_t st = ctx->stream;
    DevBuf<double> dN, dV, dE;
    DFTA_HIP(ctx, dN.alloc(sz)); DFTA_HIP(ctx, dV.alloc(sz)); DFTA_HIP(ctx, dE.alloc(sz));
    DFTA_HIP(ctx, hipMemcpyAsync(dN.p, n, sizeof(double) * sz, hipMemcpyHostToDevice, st));
    int rc = dfta_launch_chachiyo_lda(ctx, improved, dN.p, sz, dV.p, dE.p);
    if (rc) return rc;
    if (vexc) DFTA_HIP(ctx, hipMemcpyAsync(vexc, dV.p, sizeof(double) * sz, hipMemcpyDeviceToHost, st));
    if (eexcdif) DFTA_HIP(ctx, hipMemcpyAsync(eexcdif, dE.p, sizeof(double) * sz, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return DFTA_OK;
}
