// ordered_sum.h -- wave-cooperative summation in the REFERENCE'S ORDER (device code).
//
// Integral::Simpson38 & co. (Integral.h:11-155) accumulate `sum += values[i]` sequentially.  A tree
// reduction would differ in the last bits, so the sums here keep the sequential order exactly: the 64
// lanes of a wave fetch a tile with coalesced loads into LDS, then every lane redundantly performs the
// same chain of fp64 adds while reading the tile back as LDS broadcasts (48 elements per batch of reads).
// The dependent-add chain costs one issue slot (4 cycles on gfx950) per element plus the LDS reads;
// independent vectors go to different waves.
#pragma once
#include <hip/hip_runtime.h>

namespace dfta {

constexpr int kTile = 768;   // doubles per tile: divisible by 1,2,3,4 (period of every Newton-Cotes rule here) and 64

// Adds v[first + m*stride], m = 0..count-1, in increasing m, to up to three accumulators selected by
// cls[m % P] (0,1,2).  `lds` points to kTile doubles private to the calling wave.  All 64 lanes must call.
template <int P>
__device__ __forceinline__ void wave_ordered_sums(const double* __restrict__ v, long first, long stride, long count,
                                                  const int (&cls)[P], double (&acc)[3], double* lds)
{
    const int lane = threadIdx.x & 63;
    // the next tile's loads (coalesced for stride 1, a strided gather otherwise) are in flight while the chain of the
    // current one runs: a round trip to memory takes about as long as the 768 dependent adds of a tile
    double nxt[kTile / 64];
    auto fetch = [&](long base) {
#pragma unroll
        for (int k = 0; k < kTile / 64; ++k) {
            const long j = base + k * 64 + lane;
            nxt[k] = j < count ? v[first + j * stride] : 0.0;
        }
    };
    fetch(0);
    for (long base = 0; base < count; base += kTile) {
        const int nt = (count - base) < kTile ? static_cast<int>(count - base) : kTile;
#pragma unroll
        for (int k = 0; k < kTile / 64; ++k) lds[k * 64 + lane] = nxt[k];
        if (base + kTile < count) fetch(base + kTile);
        __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0) only: the tile is in LDS, the next one stays in flight
        __builtin_amdgcn_wave_barrier();
        // sequential chain; base is a multiple of kTile and kTile % P == 0, so the class pattern restarts per tile.
        // Blocks of kBlk elements are read back with 16-byte broadcasts that are all in flight before the first add:
        // the chain then costs one dependent v_add_f64 per element instead of one LDS round trip.
        constexpr int kBlk = 48;                // multiple of every P and of 2; kTile % kBlk == 0
        typedef double v2 __attribute__((ext_vector_type(2)));
        int j = 0;
        for (; j + kBlk <= nt; j += kBlk) {
            v2 x[kBlk / 2];
            const v2* __restrict__ src = reinterpret_cast<const v2*>(lds + j);
#pragma unroll
            for (int q = 0; q < kBlk / 2; ++q) x[q] = src[q];
#pragma unroll
            for (int q = 0; q < kBlk; ++q) {
                const double xv = (q & 1) ? x[q >> 1].y : x[q >> 1].x;
                const int c = cls[q % P];
                if (c == 0) acc[0] += xv;
                else if (c == 1) acc[1] += xv;
                else acc[2] += xv;
            }
        }
        for (; j + P <= nt; j += P) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const double x = lds[j + q];
                if (cls[q] == 0) acc[0] += x;
                else if (cls[q] == 1) acc[1] += x;
                else acc[2] += x;
            }
        }
        for (int q = 0; j < nt; ++j, ++q) {
            const double x = lds[j];
            if (cls[q] == 0) acc[0] += x;
            else if (cls[q] == 1) acc[1] += x;
            else acc[2] += x;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Integral::Simpson38(delta, values) (Integral.h:50-73); every lane returns the same value.
__device__ __forceinline__ double wave_simpson38(const double* __restrict__ v, int sz, double delta, double* lds)
{
    double acc[3] = {0.0, 0.0, 0.0};            // acc[0] = sum1 (i % 3 != 0), acc[1] = sum2 (i % 3 == 0)
    const int cls[3] = {0, 0, 1};               // i = 1,2,3,...  ->  sum1, sum1, sum2
    wave_ordered_sums<3>(v, 1, 1, static_cast<long>(sz) - 2, cls, acc, lds);
    double sum = v[0] + v[sz - 1];
    sum += 3. * acc[0] + 2. * acc[1];
    constexpr double coef = 3. / 8.;
    return sum * delta * coef;
}

// Integral::Simpson38 by TWO waves: its two running sums (sum1 over i % 3 != 0, sum2 over i % 3 == 0) are independent chains,
// each in the reference's order.  Threads 0..127 of the block fetch a tile and store it de-interleaved -- sum1's elements in
// lds[0..511], sum2's in lds[512..767] -- then wave 0 adds up the first region while wave 1 adds up the second: the longer chain
// has 2/3 of the elements of the single-wave version.  EVERY thread of the block must call (block-wide barriers); all return
// the same value.  lds: kTile doubles, xch: 2 doubles.
// (`load(i)` delivers element i: an array, or an integrand evaluated on the fly.  FETCH0: the first of the 128 threads that fetch the tiles
// -- 0: the two waves that also run the chains; 128 in a block of >= 256 threads: two other waves, so that the chain waves touch LDS only
// and an integrand of several loads and flops per element costs the chains nothing)
template <int FETCH0 = 0, typename Load>
__device__ __forceinline__ double block_simpson38_of(Load load, int sz, double delta, double* lds, double* xch)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    const int wave = threadIdx.x >> 6;
    const int tid = static_cast<int>(threadIdx.x) - FETCH0;      // fetch lane: 0 .. 127 fetch
    const long count = static_cast<long>(sz) - 2;       // elements j = 0 .. count-1 are the nodes i = j + 1
    constexpr int kPer = kTile / 128;
    double nxt[kPer];
    auto fetch = [&](long base) {
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const long j = base + k * 128 + tid;
            nxt[k] = (tid >= 0 && tid < 128 && j < count) ? load(1 + j) : 0.0;
        }
    };
    // one chain: n elements of a region, added in order; reads as 16-byte broadcasts, 32 elements in flight ahead of the adds
    auto chain = [&](const double* region, int n, double acc) -> double {
        constexpr int kBlk = 32;
        int j = 0;
        for (; j + kBlk <= n; j += kBlk) {
            v2 x[kBlk / 2];
            const v2* __restrict__ src = reinterpret_cast<const v2*>(region + j);
#pragma unroll
            for (int q = 0; q < kBlk / 2; ++q) x[q] = src[q];
#pragma unroll
            for (int q = 0; q < kBlk / 2; ++q) { acc += x[q].x; acc += x[q].y; }
        }
        for (; j < n; ++j) acc += region[j];
        return acc;
    };
    double acc = 0;
    fetch(0);
    for (long base = 0; base < count; base += kTile) {
        const int nt = (count - base) < kTile ? static_cast<int>(count - base) : kTile;
        if (tid >= 0 && tid < 128) {
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int j = k * 128 + tid, t = j / 3, r = j - 3 * t;       // base is a multiple of 3: the pattern restarts per tile
                lds[r < 2 ? 2 * t + r : 512 + t] = nxt[k];
            }
        }
        if (base + kTile < count) fetch(base + kTile);
        __syncthreads();
        const int triples = nt / 3, rest = nt - 3 * triples;
        if (wave == 0)      acc = chain(lds, 2 * triples + (rest < 2 ? rest : 2), acc);
        else if (wave == 1) acc = chain(lds + 512, triples, acc);
        __syncthreads();
    }
    if (threadIdx.x == 0) xch[0] = acc;
    if (threadIdx.x == 64) xch[1] = acc;
    __syncthreads();
    double sum = load(0) + load(static_cast<long>(sz) - 1);
    sum += 3. * xch[0] + 2. * xch[1];
    constexpr double coef = 3. / 8.;
    return sum * delta * coef;
}
__device__ __forceinline__ double block_simpson38(const double* __restrict__ v, int sz, double delta, double* lds, double* xch)
{
    return block_simpson38_of([v](long i) { return v[i]; }, sz, delta, lds, xch);
}

// Integral::Romberg(delta, values, 1E-18, 3) (Integral.h:106-155): the trapezoid refinement of level i adds
// values[n], values[n + oldStep], ... (n = numPoints >> i, oldStep = numPoints >> (i-1)) in that order; the extrapolation
// table is filled exactly as the reference does.  rtab: 2 * 32 doubles of LDS private to the wave.  All 64 lanes call
// and return the same value.
__device__ __forceinline__ double wave_romberg(const double* __restrict__ v, int sz, double delta, double* lds, double* rtab)
{
    const int numPoints = sz - 1;
    int cnt = 0;
    for (int n = numPoints; n; n >>= 1) ++cnt;
    double* Rprev = rtab;
    double* Rcur = rtab + 32;
    const int lane = threadIdx.x & 63;
    if (lane < 32) { Rprev[lane] = 0; Rcur[lane] = 0; }
    __builtin_amdgcn_wave_barrier();
    double h = delta * numPoints;
    if (lane == 0) Rprev[0] = 0.5 * h * (v[0] + v[numPoints]);
    double result = 0;
    bool done = false;
    for (int i = 1; i < cnt && !done; ++i) {
        const long n = numPoints >> i;
        const long oldStep = numPoints >> (i - 1);
        double acc[3] = {0, 0, 0};
        const int cls[1] = {0};
        if (n > 0) wave_ordered_sums<1>(v, n, oldStep, (numPoints - 1 - n) / oldStep + 1, cls, acc, lds);
        h *= 0.5;
        int stop = 0;
        if (lane == 0) {
            Rcur[0] = 0.5 * Rprev[0] + h * acc[0];
            double nk = 1;
            for (int m = 1; m <= i; ++m) {
                nk *= 4;
                Rcur[m] = Rcur[m - 1] + (Rcur[m - 1] - Rprev[m - 1]) / (nk - 1);
            }
            if (i >= 3 && fabs(Rcur[i] - Rprev[i - 1]) < 1E-18) stop = 1;
        }
        stop = __shfl(stop, 0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (stop) { result = Rcur[i]; done = true; }
        double* t = Rcur; Rcur = Rprev; Rprev = t;
    }
    if (!done) result = Rprev[cnt - 1];
    return result;
}

// Integral::{Trapezoid, SimpsonOneThird, Simpson38, Boole, Romberg}(delta, values) (Integral.h:11-155) by one wave;
// rule = DFTA_INT_* (include/dftatom_hip.h).  lds: kTile doubles, rtab: 64 doubles (Romberg only).
__device__ __forceinline__ double wave_integrate(int rule, const double* __restrict__ v, int sz, double delta, double* lds, double* rtab)
{
    const long cnt = static_cast<long>(sz) - 2;   // interior points i = 1 .. sz-2
    if (rule == 0) {                                            // Trapezoid, Integral.h:11-23
        double acc[3] = {0.5 * (v[0] + v[sz - 1]), 0, 0};       // the running sum starts from the end terms
        const int cls[1] = {0};
        wave_ordered_sums<1>(v, 1, 1, cnt, cls, acc, lds);
        return acc[0] * delta;
    }
    if (rule == 1) {                                            // SimpsonOneThird, Integral.h:25-48
        double acc[3] = {0, 0, 0};                              // acc0 = sum4 (odd i), acc1 = sum2 (even i)
        const int cls[2] = {0, 1};
        wave_ordered_sums<2>(v, 1, 1, cnt, cls, acc, lds);
        double sum = v[0] + v[sz - 1];
        sum += 4. * acc[0] + 2. * acc[1];
        constexpr double coef = 1. / 3.;
        return sum * delta * coef;
    }
    if (rule == 2) return wave_simpson38(v, sz, delta, lds);
    if (rule == 3) {                                            // Boole, Integral.h:75-104
        double acc[3] = {0, 0, 0};                              // acc0 = sum32 (odd i), acc1 = sum12 (i%4==2), acc2 = sum14 (i%4==0)
        const int cls[4] = {0, 1, 0, 2};
        wave_ordered_sums<4>(v, 1, 1, cnt, cls, acc, lds);
        double sum = 7. * (v[0] + v[sz - 1]);
        sum += 32. * acc[0] + 12. * acc[1] + 14. * acc[2];
        constexpr double coef = 2. / 45.;
        return sum * delta * coef;
    }
    return wave_romberg(v, sz, delta, lds, rtab);
}

}  // namespace dfta
