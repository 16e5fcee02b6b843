// xc.h -- device-pointer launchers of the VWN kernels (xc.hip) and of the Poisson solve (poisson.hip)
#pragma once
#include "common.h"

int dfta_launch_vwn_lda(dfta_ctx* ctx, const double* dN, size_t sz, double* dVexc, double* dEexc);
int dfta_launch_vwn_lsda(dfta_ctx* ctx, const double* dNa, const double* dNb, size_t sz, double* dRes, double* dVa, double* dVb,
                         double* dEexc);
int dfta_poisson_solve_launch(dfta_poisson* p, const int* dZ, const double* dDensity, double* dU, int* dVcycles, double* dErr);
int dfta_poisson_take_vcycles(dfta_poisson* p, unsigned long long* out);
