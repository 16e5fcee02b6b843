// xc.hip -- Vosko-Wilk-Nusair exchange-correlation, LDA and LSDA, as coalesced pointwise kernels.
//
// Replaces VWNExchCor::Vexc / eexcDif (VWNExcCor.h:73-128, LDA) and the spin-polarised pair
// (VWNExcCor.h:134-312, LSDA) with ExcCorBase::f / df (ExcCorBase.h:14-26).  Expressions are written in the
// reference's operation order; pow/log/atan come from the ROCm device library, so results agree with the
// glibc-based reference to rounding of those functions (parity tolerance: 1e-13 relative, see tests).
// HBM-bound by construction: LDA reads 8 B and writes 16 B per point, LSDA reads 16 B and writes 32 B.
#include <hip/hip_runtime.h>

#include <cmath>

#include "internal.h"
#include "xc.h"

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double fourM_PI = 4. * kPi;
constexpr double aThird = 1. / 3.;                                  // ExcCorBase.h:12
// VWNExcCor.h:23-41
constexpr double AP = 0.0310907, y0P = -0.10498, bP = 3.72744, cP = 12.93532;
constexpr double Y0P = y0P * y0P + bP * y0P + cP;
constexpr double AF = 0.01554535, y0F = -0.325, bF = 7.06042, cF = 18.0578;
constexpr double Y0F = y0F * y0F + bF * y0F + cF;
constexpr double Aalpha = -1. / (6. * kPi * kPi);
constexpr double y0alpha = -0.0047584, balpha = 1.13107, calpha = 13.0045;
constexpr double Y0alpha = y0alpha * y0alpha + balpha * y0alpha + calpha;

__device__ __forceinline__ double vwnF(double y, double dify, double A, double y0, double b, double c, double Y0, double Y)
{   // VWNExcCor.h:43-50 (B.5)
    const double Q = sqrt(4 * c - b * b);
    const double twoyb = 2. * y + b;
    const double atanQ = atan(Q / twoyb);
    return A * (log(y * y / Y) + 2. * b / Q * atanQ - b * y0 / Y0 * (log(dify * dify / Y) + 2. * (b + 2. * y0) / Q * atanQ));
}

__device__ __forceinline__ double vwnEcDif(double y, double dify, double A, double y0, double b, double c, double Y)
{   // VWNExcCor.h:52-55 (B.6)
    return A * (c * dify - b * y0 * y) / (dify * Y);
}

__device__ __forceinline__ double spin_f(double zeta, double p2third)
{   // ExcCorBase.h:14-19
    const double mul = 1. / (2. * (p2third - 1.));
    return mul * (pow(1. + zeta, 4. * aThird) + pow(1. - zeta, 4. * aThird) - 2.);
}

__device__ __forceinline__ double spin_df(double zeta, double p2third)
{   // ExcCorBase.h:21-26
    const double mul = 2. / (3. * (p2third - 1.));
    return mul * (pow(1. + zeta, aThird) - pow(1. - zeta, aThird));
}

__global__ void k_vwn_lda(const double* __restrict__ n, size_t sz, double* __restrict__ vexc, double* __restrict__ eexc, double X1)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < sz; i += (size_t)gridDim.x * blockDim.x) {
        const double ro = n[i];
        double v = 0., e = 0.;
        if (!(ro < 1E-18)) {                                          // VWNExcCor.h:82,112
            const double rs = pow(3. / (fourM_PI * ro), aThird);
            const double y = sqrt(rs);
            const double Y = y * y + bP * y + cP;
            const double dify = y - y0P;
            const double ecd = vwnEcDif(y, dify, AP, y0P, bP, cP, Y);
            v = -X1 / rs + vwnF(y, dify, AP, y0P, bP, cP, Y0P, Y) - aThird * ecd;       // VWNExcCor.h:94-97
            e = (0.25 * X1) / rs + aThird * ecd;                                          // VWNExcCor.h:123-124
        }
        if (vexc) vexc[i] = v;
        if (eexc) eexc[i] = e;
    }
}

__global__ void k_vwn_lsda(const double* __restrict__ na, const double* __restrict__ nb, size_t sz, double* __restrict__ res,
                           double* __restrict__ va, double* __restrict__ vb, double* __restrict__ eexc, double X1, double X2)
{
    const double X12 = X1 * X2;
    const double fdd = 4. / (9. * (X2 - 1.));
    const double X1d = 0.25 * X1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < sz; i += (size_t)gridDim.x * blockDim.x) {
        const double roa = na[i];
        const double rob = nb[i];
        const double n = roa + rob;
        double r = 0., a = 0., b = 0., e = 0.;
        if (!(n < 1E-18)) {                                           // VWNExcCor.h:160,260
            const double rs = pow(3. / (fourM_PI * n), aThird);
            const double rsa = pow(3. / (fourM_PI * roa), aThird);
            const double rsb = pow(3. / (fourM_PI * rob), aThird);

            const double exp_ = -X1 / rs;
            const double exf = X2 * exp_;
            const double exdif = exf - exp_;
            const double exfa = -X12 / rsa;
            const double exfb = -X12 / rsb;

            const double zeta = (roa - rob) / n;
            const double zeta3 = zeta * zeta * zeta;
            const double zeta4 = zeta3 * zeta;
            const double fval = spin_f(zeta, X2);
            const double dfval = spin_df(zeta, X2);
            const double y = sqrt(rs);

            const double YP = y * (y + bP) + cP;
            const double difyP = y - y0P;
            const double ecp = vwnF(y, difyP, AP, y0P, bP, cP, Y0P, YP);
            const double YF = y * (y + bF) + cF;
            const double difyF = y - y0F;
            const double ecf = vwnF(y, difyF, AF, y0F, bF, cF, Y0F, YF);
            const double YA = y * (y + balpha) + calpha;
            const double difyA = y - y0alpha;
            const double eca = vwnF(y, difyA, Aalpha, y0alpha, balpha, calpha, Y0alpha, YA);

            const double ecpd = vwnEcDif(y, difyP, AP, y0P, bP, cP, YP);
            const double ecfd = vwnEcDif(y, difyF, AF, y0F, bF, cF, YF);
            const double ecad = vwnEcDif(y, difyA, Aalpha, y0alpha, balpha, calpha, YA);

            const double deltaecfp = ecf - ecp;
            const double beta = fdd * deltaecfp / eca - 1.;
            const double opbz4 = 1. + beta * zeta4;
            const double interp = fval / fdd * opbz4;
            const double deltaec = eca * interp;
            const double betad = fdd / eca * (ecfd - ecpd - ecad * deltaecfp / eca);
            const double interpd = fval / fdd * zeta4 * betad;
            const double deriv = aThird * (ecpd + ecad * interp + eca * interpd);
            const double dterm = eca / fdd * (4. * beta * zeta3 * fval + opbz4 * dfval);

            r = ecp + deltaec - deriv;                                 // VWNExcCor.h:219-231
            a = exfa + r + (1. - zeta) * dterm;                        // VWNExcCor.h:233
            b = exfb + r - (1. + zeta) * dterm;                        // VWNExcCor.h:234
            r += (exp_ + exdif * fval);                                // VWNExcCor.h:236

            const double expd = X1d / rs;                              // VWNExcCor.h:271-272
            const double exfd = X2 * expd;
            e = expd + (exfd - expd) * fval + deriv;                   // VWNExcCor.h:306-308
        }
        if (res) res[i] = r;
        if (va) va[i] = a;
        if (vb) vb[i] = b;
        if (eexc) eexc[i] = e;
    }
}

}  // namespace

// X1 = pow(3/(2 pi), 2/3), X2 = pow(2, 1/3) are evaluated once on the host with libm (VWNExcCor.h:75,139-140)
static double host_X1() { return pow(3. / (2. * kPi), 2. * aThird); }
static double host_X2() { return pow(2., aThird); }

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
    if (int rc_ = dfta_use(ctx)) return rc_;
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
    if (int rc_ = dfta_use(ctx)) return rc_;
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
