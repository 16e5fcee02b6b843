// call_stream_check.cpp -- CPU check of dftatom_amd/compat/call_stream.h without a device: a synthetic "sweep" (node count = thresholds below
// E, u(0) = a polynomial with its sign changes at the eigenvalues) behind the same serve-a-call logic as DFT::Numerov::one_trial, driven by
// (a) the level-search protocol of the reference's orchestration (LocateInterval + u(0) bisection, DFTAtom.cpp:493-604) over six chained
// levels, (b) a caller with another energyErr, (c) a caller asking arbitrary energies.  Every answer must equal the direct evaluation;
// for (a) the launches must be a small fraction of the calls.
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../dftatom_amd/compat/call_stream.h"

using dfta_compat::CallStream;

static double kEig[] = {-3204.75642, -535.87331, -512.1183, -130.2447, -118.90112, -99.3301, -30.118, -24.6, -12.0, -3.3};
static int count_of(double E, int limit)
{
    int c = 0;
    for (double e : kEig) if (E > e) ++c;
    return c > limit ? limit + 1 : c;
}
static double u0_of(double E)
{
    double p = 1;
    for (double e : kEig) p *= (E - e) * 1e-2;
    return p;
}

struct Server {
    CallStream cs;
    long launches = 0, hits = 0, trials = 0, wrong = 0;
    std::vector<double> Es;
    CallStream::Value serve(int kind, int l, int limit, double E)
    {
        CallStream::Value v{0, 0.0};
        cs.sync(kind, l, limit, E);
        if (!cs.lookup(kind, l, limit, E, v)) {
            cs.plan(kind, E, 13, 8191, Es);
            std::vector<int> c(Es.size());
            std::vector<double> u(Es.size());
            for (size_t q = 0; q < Es.size(); ++q) { c[q] = count_of(Es[q], limit); u[q] = u0_of(Es[q]); }
            cs.store(kind, l, limit, Es, kind == DFTA_SWEEP_COUNT ? c.data() : nullptr, kind == DFTA_SWEEP_ZERO ? u.data() : nullptr);
            v = CallStream::Value{kind == DFTA_SWEEP_COUNT ? c[0] : 0, kind == DFTA_SWEEP_ZERO ? u[0] : 0.0};
            ++launches;
            trials += static_cast<long>(Es.size());
        } else ++hits;
        cs.advance(v);
        if (kind == DFTA_SWEEP_COUNT ? v.count != count_of(E, limit) : v.u0 != u0_of(E)) ++wrong;
        return v;
    }
};

// the protocol of the reference's level search with the given energyErr; returns the number of calls
static long search(Server& s, int nlevels, double err, double bottom, std::vector<double>& found)
{
    long calls = 0;
    for (int k = 0; k < nlevels; ++k) {
        const int l = k % 3, nodes = k;            // (l only labels the stream here)
        double hi = 50, lo = bottom;
        while (hi - lo > err) { const double E = (hi + lo) / 2; ++calls; if (s.serve(DFTA_SWEEP_COUNT, l, nodes, E).count > nodes) hi = E; else lo = E; }
        const double top = hi;
        lo = bottom;
        while (hi - lo > err) { const double E = (hi + lo) / 2; ++calls; if (s.serve(DFTA_SWEEP_COUNT, l, nodes, E).count < nodes) lo = E; else hi = E; }
        double B = hi, T = top;
        const bool sB = s.serve(DFTA_SWEEP_ZERO, l, 0, B).u0 > 0; ++calls;
        for (int i = 0; i < 500; ++i) {
            const double E = (T + B) / 2;
            const double u = s.serve(DFTA_SWEEP_ZERO, l, 0, E).u0; ++calls;
            if ((u > 0) == sB) B = E; else T = E;
            const double a = std::fabs(u);
            if (T - B < err && !std::isnan(a) && a < 1E15) break;
        }
        found.push_back(B);
        bottom = B - 3;
    }
    return calls;
}

int main()
{
    std::vector<double> a, b;
    Server s1;
    const long c1 = search(s1, 6, 1E-12, -86.0 * 86.0 - 1.0, a);
    std::printf("reference protocol: calls %ld launches %ld hits %ld trials %ld wrong %ld\n", c1, s1.launches, s1.hits, s1.trials, s1.wrong);
    for (size_t k = 0; k < a.size(); ++k) std::printf("level %zu E %.17g\n", k, a[k]);
    // the same six levels again and again with the eigenvalues moving a little, as from one SCF step to the next (a new Server = a new Numerov):
    // from the fourth search on the history of the level end points gives the launches a spine of predicted decisions
    const double base[10] = {kEig[0], kEig[1], kEig[2], kEig[3], kEig[4], kEig[5], kEig[6], kEig[7], kEig[8], kEig[9]};
    double shift = 2e-6;
    long first_launches = s1.launches, last_launches = 0, wrong_steps = s1.wrong;
    for (int step = 1; step <= 6; ++step) {
        for (int q = 0; q < 10; ++q) kEig[q] = base[q] * (1.0 + shift);
        shift *= 0.5;
        Server ss;
        std::vector<double> e;
        const long c = search(ss, 6, 1E-12, -86.0 * 86.0 - 1.0, e);
        std::printf("step %d: calls %ld launches %ld hits %ld wrong %ld\n", step, c, ss.launches, ss.hits, ss.wrong);
        last_launches = ss.launches;
        wrong_steps += ss.wrong;
    }
    std::printf("history: launches first search %ld last search %ld wrong %ld\n", first_launches, last_launches, wrong_steps);
    for (int q = 0; q < 10; ++q) kEig[q] = base[q];
    Server s2;                                      // another energyErr: the mirror loses the caller where the loops end -- answers stay right
    const long c2 = search(s2, 4, 1E-9, -86.0 * 86.0 - 1.0, b);
    std::printf("other energyErr: calls %ld launches %ld hits %ld wrong %ld\n", c2, s2.launches, s2.hits, s2.wrong);
    Server s3;                                      // no pattern at all
    long c3 = 0;
    for (int i = 0; i < 400; ++i) { const double E = -3000.0 + 7.77 * i; s3.serve(i % 2, i % 4, 3, E); ++c3; }
    std::printf("arbitrary caller: calls %ld launches %ld hits %ld wrong %ld trials %ld\n", c3, s3.launches, s3.hits, s3.wrong, s3.trials);
    return (s1.wrong || s2.wrong || s3.wrong || wrong_steps) ? 1 : 0;
}
