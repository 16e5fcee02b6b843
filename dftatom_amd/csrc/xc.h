// xc.h -- device-pointer launchers of the VWN kernels (xc.hip) and of the Poisson solve (poisson.hip)
#pragma once
#include "common.h"

int dfta_launch_vwn_lda(dfta_ctx* ctx, const double* dN, size_t sz, double* dVexc, double* dEexc);
int dfta_launch_vwn_lsda(dfta_ctx* ctx, const double* dNa, const double* dNb, size_t sz, double* dRes, double* dVa, double* dVb,
                         double* dEexc);
int dfta_launch_chachiyo_lda(dfta_ctx* ctx, int improved, const double* dN, size_t sz, double* dVexc, double* dEexc);
// poisson.hip: launch (asynchronous) / finish (synchronises, inspects the group barriers' abort flag and repeats the solve with
// one workgroup per atom if it was raised).  dSkip: per atom, non-zero = leave this atom alone (may be null).
int dfta_poisson_solve_launch(dfta_poisson* p, const int* dZ, const double* dDensity, double* dU, int* dVcycles, double* dErr,
                              const int* dSkip);
int dfta_poisson_finish(dfta_poisson* p, const int* dZ, const double* dDensity, double* dU, int* dVcycles, double* dErr,
                        const int* dSkip);
int dfta_poisson_take_vcycles(dfta_poisson* p, unsigned long long* out);
int dfta_poisson_group_state(const dfta_poisson* p, int* G, int* degraded, int* aborts);
