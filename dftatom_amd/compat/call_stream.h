// call_stream.h -- speculation for the PER-CALL surface of DFT::Numerov (round 5).
//
// The reference's own orchestration (DFTAtom::LoopOverLevels / LocateInterval, DFTAtom.cpp:493-604) asks for ONE trial energy per call:
// a 131 073-step dependent recurrence, 4 ms on the device whatever the machine could do beside it.  But its call stream is a function of
// the answers alone: three bisections per level whose next energy is (toe + boe) / 2 of bounds that are themselves earlier energies (or
// 50, or the previous level's eigenvalue - 3).  CallStream mirrors that state machine from the calls it sees and the answers it gives, and
// when a call is not in its cache it has the whole tree of energies the caller can ask for in its next `depth` calls integrated in the
// SAME launch (a sweep of 4 095 trials costs what a sweep of one costs).  The following calls are answered from the cache.
//
// Across SCF steps (the reference builds a new Numerov for every step): the three end points of every level -- top and bottom of its band, its
// eigenvalue -- are remembered per (l, nodes) in a process-wide history; the next time that level is searched, the bisection is followed
// along the side of those values (a SPINE of predicted decisions, one trial each, while the midpoints stay outside a bracket as wide as
// the end point has moved lately) and the tree of all answers hangs at the spine's end: ~30 decisions per launch instead of 12.  A
// prediction that turns out wrong costs a launch and switches the spines off for the rest of that level.
//
// Nothing here can change an answer: a cached value is returned only for the bit-identical (kind, l, nodesLimit, E) of a trial that was
// integrated by the same kernels on the same potential; a caller that is not the reference's loop (or uses another energyErr) simply
// misses the cache and gets its single trial as before.  The cache is dropped when the potential changes (Numerov.h compares it on every
// call, as the reference re-reads it on every call) and when a new level starts.
#pragma once

#include <math.h>

#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/dftatom_hip.h"

namespace dfta_compat {

// end points of the levels searched so far in this process: [0] top, [1] bottom of the band, [2] eigenvalue; newest first, three deep
class LevelHistory {
public:
    static LevelHistory& instance()
    {
        static LevelHistory h;
        return h;
    }
    void push(int l, int nodes, double top, double bottom, double E)
    {
        auto& q = m_h[std::make_pair(l, nodes)];
        q.push_front(Ends{{top, bottom, E}});
        if (q.size() > 3) q.pop_back();
    }
    // where will end point `which` of (l, nodes) be this time?  c +- w with c = the newest value and w = twice the larger of its distances to
    // the two before it (LDA: the last two movements; LSDA: the other spin's value and the movement) + the round-off floor
    bool bracket(int l, int nodes, int which, double& lo, double& hi) const
    {
        auto it = m_h.find(std::make_pair(l, nodes));
        if (it == m_h.end() || it->second.size() < 3) return false;
        const double c = it->second[0].T[which], a = fabs(c - it->second[1].T[which]), b = fabs(c - it->second[2].T[which]);
        const double w = 2 * (a > b ? a : b) + 1e-10 * fabs(c) + 64e-12;
        lo = c - w;
        hi = c + w;
        return true;
    }
    void clear() { m_h.clear(); }

private:
    struct Ends { double T[3]; };
    std::map<std::pair<int, int>, std::deque<Ends>> m_h;
};

class CallStream {
public:
    enum { KIND_COUNT = DFTA_SWEEP_COUNT, KIND_ZERO = DFTA_SWEEP_ZERO };

    struct Value { int count; double u0; };

    void reset()
    {
        m_cache.clear();
        m_pred.clear();
        m_m = Mirror();
        m_in_sync = false;
    }
    bool lookup(int kind, int l, int limit, double E, Value& v)
    {
        auto it = m_cache.find(Key{kind, l, limit, bits(E)});
        if (it == m_cache.end()) return false;
        v = it->second;
        ++m_hits_since_launch;
        return true;
    }
    // Every call starts here: does it continue the mirrored loop, or is it the first call of LocateInterval for another level (then the
    // mirror starts over with it and what was integrated ahead for the last level is dropped)?  Returns whether the mirror follows the caller.
    bool sync(int kind, int l, int limit, double E)
    {
        if (fits(m_m, kind, l, limit, E)) return m_in_sync = true;
        Mirror m = m_m;
        if (start_level(m, kind, l, limit, E)) {
            m_cache.clear();
            m_m = m;
            return m_in_sync = true;
        }
        m_m = Mirror();
        return m_in_sync = false;
    }
    // The energies to integrate for a call that is not cached: E itself first, then -- when the mirror follows the caller -- every energy
    // the reference's loop can ask for in its next `depth` calls of the same kind (at most `cap` in all).
    void plan(int kind, double E, int depth, size_t cap, std::vector<double>& out)
    {
        out.clear();
        out.push_back(E);
        m_pred.clear();
        // a caller that only LOOKS like the reference's loop (every call "starts a level", nothing integrated ahead is ever asked for) must
        // not pay for trees: after three launches without a single hit the speculation pauses, for 16, 32, 64 ... calls
        if (m_last_launch_speculated) m_useless = m_hits_since_launch == 0 ? m_useless + 1 : 0;
        m_hits_since_launch = 0;
        m_last_launch_speculated = false;
        if (!m_in_sync) return;
        if (m_useless >= 3) {
            if (m_pause == 0) { m_pause = m_pause_len; m_pause_len = m_pause_len < (1 << 20) ? 2 * m_pause_len : m_pause_len; }
            if (--m_pause > 0) return;
            m_useless = 2;                       // one more try; a hit resets everything
        } else if (m_useless == 0) m_pause_len = 16;
        m_last_launch_speculated = true;
        std::unordered_map<uint64_t, char> seen;
        seen.emplace(bits(E), 0);
        // the spine: while the side of the expected end point decides the pending call, follow it (one trial per decision)
        Mirror cur = m_m;
        if (m_spines && !cur.nospine) {
            const LevelHistory& H = LevelHistory::instance();
            for (int steps = 0; steps < 56 && out.size() < cap / 2; ++steps) {
                int k;
                double Ep;
                if (!pending(cur, k, Ep) || k != kind) break;
                if (seen.emplace(bits(Ep), 0).second) out.push_back(Ep);
                if (cur.ph == P3_FIRST) { apply_outcome(cur, 1); continue; }        // no decision: either sign leads to the same next energy
                int below = -1;                                                     // 1: the lower bound moves to Ep (Ep lies below the end point), 0: the upper one
                if (cur.ph == P2 && cur.nodes == 0) below = 0;                      // "count < 0" never holds: certain
                else {
                    double lo, hi;
                    if (!H.bracket(cur.l, cur.nodes, cur.ph == P1 ? 0 : (cur.ph == P2 ? 1 : 2), lo, hi)) break;
                    if (Ep <= lo) below = 1; else if (Ep >= hi) below = 0; else break;
                    m_pred[bits(Ep)] = static_cast<char>(below);
                }
                apply_side(cur, below == 1);
            }
        }
        // the tree of every answer from there on
        {
            int k;
            double Ep;
            if (pending(cur, k, Ep) && k == kind && seen.emplace(bits(Ep), 0).second) out.push_back(Ep);
        }
        std::vector<Mirror> frontier(1, cur), next;
        for (int d = 0; d < depth && !frontier.empty() && out.size() < cap; ++d) {
            if (out.size() + 2 * frontier.size() > cap && d > 0) break;                  // a level of the tree that does not fit whole is of little use
            next.clear();
            for (const Mirror& f : frontier) {
                for (int outcome = 0; outcome < 2 && out.size() < cap; ++outcome) {      // both answers the caller can get for f's pending call
                    Mirror c = f;
                    if (!apply_outcome(c, outcome)) continue;
                    int k2;
                    double E2;
                    if (!pending(c, k2, E2) || k2 != kind) continue;          // the level's next bisection is of the other kind: not in this launch
                    if (seen.emplace(bits(E2), 0).second) out.push_back(E2);
                    next.push_back(c);
                }
                if (out.size() >= cap) break;
            }
            frontier.swap(next);
        }
    }
    void store(int kind, int l, int limit, const std::vector<double>& Es, const int* counts, const double* u0)
    {
        for (size_t q = 0; q < Es.size(); ++q) m_cache[Key{kind, l, limit, bits(Es[q])}] = Value{counts ? counts[q] : 0, u0 ? u0[q] : 0.0};
    }
    // the caller is given this answer: follow it
    void advance(const Value& v)
    {
        if (!m_in_sync) return;
        int k;
        double E;
        if (!m_pred.empty() && pending(m_m, k, E)) {
            auto it = m_pred.find(bits(E));
            if (it != m_pred.end()) {
                const bool below = m_m.ph == P1 ? !(v.count > m_m.nodes) : (m_m.ph == P2 ? v.count < m_m.nodes : ((v.u0 > 0) == m_m.sgn));
                if (below != (it->second == 1)) { m_m.nospine = true; m_pred.clear(); }      // the history misled: plain trees for the rest of this level
            }
        }
        const int before = m_m.ph;
        apply(m_m, v.count, v.u0);
        if (before != LEVEL_DONE && m_m.ph == LEVEL_DONE) LevelHistory::instance().push(m_m.l, m_m.nodes, m_m.top, m_m.band_bottom, m_m.e_final);
    }
    void set_spines(bool on) { m_spines = on; }
    size_t cached() const { return m_cache.size(); }

private:
    static constexpr double kErr = 1E-12;         // energyErr of DFTAtom.cpp:369 (every caller of LoopOverLevels passes it)
    enum Phase { IDLE = 0, P1, P2, P3_FIRST, P3_LOOP, LEVEL_DONE };
    struct Mirror {
        int ph = IDLE;
        int l = 0, nodes = 0;
        double toe = 0, boe = 0;       // P1 / P2: LocateInterval's; P3: TopEnergy / BottomEnergy of LoopOverLevels
        double b_entry = 0, top = 0;   // BottomEnergy at the entry of LocateInterval; TopEnergy after its first loop
        bool boe_is_entry = false;     // boe still holds b_entry, which is only inferred until a call depends on it
        double b_alt[4] = {0, 0, 0, 0};
        int n_alt = 0;                 // other values of b_entry that explain the first call equally well
        int iter = 0;
        bool sgn = false;
        double e_final = 0;
        bool have_final = false;
        double band_bottom = 0;        // BottomEnergy as LocateInterval returned it
        bool nospine = false;          // a predicted decision of this level was wrong
    };
    struct Key {
        int kind, l, limit;
        uint64_t e;
        bool operator==(const Key& o) const { return kind == o.kind && l == o.l && limit == o.limit && e == o.e; }
    };
    struct KeyHash {
        size_t operator()(const Key& k) const
        {
            uint64_t h = k.e * 0x9E3779B97F4A7C15ull;
            h ^= (static_cast<uint64_t>(k.kind) << 1) ^ (static_cast<uint64_t>(k.l) << 8) ^ (static_cast<uint64_t>(static_cast<unsigned>(k.limit)) << 16);
            return static_cast<size_t>(h ^ (h >> 29));
        }
    };
    static uint64_t bits(double x)
    {
        uint64_t u;
        memcpy(&u, &x, sizeof u);
        return u;
    }

    // the call the reference's loop makes next in state m
    static bool pending(const Mirror& m, int& kind, double& E)
    {
        switch (m.ph) {
        case P1: case P2: kind = KIND_COUNT; E = (m.toe + m.boe) / 2; return true;               // DFTAtom.cpp:573,591
        case P3_FIRST: kind = KIND_ZERO; E = m.boe; return true;                                 // DFTAtom.cpp:513
        case P3_LOOP: kind = KIND_ZERO; E = (m.toe + m.boe) / 2; return true;                    // DFTAtom.cpp:519
        default: return false;
        }
    }
    static bool fits(Mirror& m, int kind, int l, int limit, double E)
    {
        int k;
        double Ep;
        if (!pending(m, k, Ep) || k != kind || l != m.l) return false;
        if (kind == KIND_COUNT && limit != m.nodes) return false;
        if (bits(Ep) == bits(E)) return true;
        // the inferred BottomEnergy may be one of its neighbours: does another candidate explain this call?
        if ((m.ph == P1 || m.ph == P2) && m.boe_is_entry)
            for (int q = 0; q < m.n_alt; ++q)
                if (bits((m.toe + m.b_alt[q]) / 2) == bits(E)) {
                    m.b_entry = m.boe = m.b_alt[q];
                    m.n_alt = 0;
                    return true;
                }
        return false;
    }
    // a CountNodes call that does not continue the running level: the first call of LocateInterval for the next level (toe = 50)?
    static bool start_level(Mirror& m, int kind, int l, int limit, double E)
    {
        if (kind != KIND_COUNT) return false;
        Mirror n;
        n.ph = P1; n.l = l; n.nodes = limit; n.toe = 50;                                           // DFTAtom.cpp:499
        n.boe_is_entry = true;
        bool found = false;
        if (m.have_final) {                                                                       // BottomEnergy = level.E - 3 (DFTAtom.cpp:541)
            const double b = m.e_final - 3;
            if (bits((50 + b) / 2) == bits(E)) { n.b_entry = b; found = true; }
        }
        if (!found) {
            // (50 + b) / 2 == E: b = 2 E - 50 up to the rounding of the sum -- the neighbours that give the same E are kept as alternatives
            const double b0 = 2 * E - 50;
            double cand[9];
            int nc = 0;
            double lo = b0, hi = b0;
            cand[nc++] = b0;
            for (int q = 0; q < 4; ++q) { lo = nextafter(lo, -INFINITY); hi = nextafter(hi, INFINITY); cand[nc++] = lo; cand[nc++] = hi; }
            for (int q = 0; q < nc; ++q) {
                if (bits((50 + cand[q]) / 2) != bits(E)) continue;
                if (!found) { n.b_entry = cand[q]; found = true; }
                else if (n.n_alt < 4) n.b_alt[n.n_alt++] = cand[q];
            }
        }
        if (!found || !(50 - n.b_entry > kErr)) return false;
        n.boe = n.b_entry;
        m = n;
        return true;
    }
    // the reference's update for the answer (count, u0) to the pending call
    static void apply(Mirror& m, int count, double u0)
    {
        int kind;
        double E;
        if (!pending(m, kind, E)) return;
        switch (m.ph) {
        case P1:
            if (count > m.nodes) m.toe = E; else { m.boe = E; m.boe_is_entry = false; }          // DFTAtom.cpp:578-581
            if (!(m.toe - m.boe > kErr)) {
                m.top = m.toe;                                                                    // :585
                m.boe = m.b_entry;                                                                // :587
                m.boe_is_entry = true;
                m.ph = P2;
                if (!(m.toe - m.boe > kErr)) enter_p3(m);
            }
            break;
        case P2:
            if (count < m.nodes) { m.boe = E; m.boe_is_entry = false; } else m.toe = E;           // :596-599
            if (!(m.toe - m.boe > kErr)) enter_p3(m);
            break;
        case P3_FIRST:
            m.sgn = u0 > 0;                                                                       // :514
            m.iter = 0;
            m.ph = P3_LOOP;
            break;
        case P3_LOOP: {
            if ((u0 > 0) == m.sgn) m.boe = E; else m.toe = E;                                     // :522-525
            ++m.iter;
            const double a = fabs(u0);
            if ((m.toe - m.boe < kErr && !isnan(a) && a < 1E15) || m.iter >= 500) {               // :527-532, the loop's cap :517
                m.e_final = m.boe;                                                                // :534
                m.have_final = true;
                m.ph = LEVEL_DONE;
            }
            break;
        }
        default: break;
        }
    }
    static void enter_p3(Mirror& m)
    {
        const double bottom = m.toe;       // BottomEnergy = toe (DFTAtom.cpp:603)
        m.band_bottom = bottom;
        m.toe = m.top;
        m.boe = bottom;
        m.boe_is_entry = false;
        m.ph = P3_FIRST;
    }
    // one of the two answers the pending call can get (speculation: representative values); false: this answer cannot occur
    static bool apply_outcome(Mirror& m, int outcome)
    {
        switch (m.ph) {
        case P1: apply(m, outcome ? m.nodes + 1 : m.nodes, 0); return true;
        case P2:
            if (outcome && m.nodes == 0) return false;           // "count < 0" never holds
            apply(m, outcome ? m.nodes - 1 : m.nodes, 0);
            return true;
        case P3_FIRST: case P3_LOOP: apply(m, 0, outcome ? 1.0 : -1.0); return true;
        default: return false;
        }
    }

    // the answer that moves the lower bound (below) or the upper one to the pending energy
    static void apply_side(Mirror& m, bool below)
    {
        switch (m.ph) {
        case P1: apply_outcome(m, below ? 0 : 1); break;                 // count <= nodes: boe = E
        case P2: apply_outcome(m, below ? 1 : 0); break;                 // count < nodes: boe = E
        case P3_LOOP: apply_outcome(m, below == m.sgn ? 1 : 0); break;   // (u0 > 0) == sgnBottom: BottomEnergy = E
        default: break;
        }
    }

    Mirror m_m;
    bool m_in_sync = false;
    bool m_spines = true;
    long m_hits_since_launch = 0;
    bool m_last_launch_speculated = false;
    int m_useless = 0, m_pause = 0, m_pause_len = 16;
    std::unordered_map<uint64_t, char> m_pred;      // energies of the last plan's spine -> the side it predicted
    std::unordered_map<Key, Value, KeyHash> m_cache;
};

}  // namespace dfta_compat
