// Host check of k_fill's streamed slow-body flags (poreseq_amd/csrc/ps_slowmask.h + the chunk loop of fill_body, ps_kernels.hip)
// against the rule they implement: a body of 8 anti-diagonals starting at s0 runs the SLOW variant iff the band resumes
// (lo(t-1) < 0 <= lo(t)) on some anti-diagonal t in [s0 - 6, s0 + 13] — the rule the plain layout's LDS bitmap marks up front.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "../../poreseq_amd/csrc/ps_slowmask.h"

static unsigned long long rnd_state = 88172645463325252ull;
static unsigned long long rnd() { rnd_state ^= rnd_state << 13; rnd_state ^= rnd_state >> 7; rnd_state ^= rnd_state << 17; return rnd_state; }

int main(int argc, char** argv) {
    const long trials = argc > 1 ? atol(argv[1]) : 200000;
    long bad = 0;
    // 1. one chunk's mask against the windows, bit by bit
    for (long it = 0; it < trials; it++) {
        unsigned long long rm = 0;
        const int nbits = (int)(rnd() % 4);                       // resumes are rare: 0-3 per chunk, plus dense masks now and then
        for (int k = 0; k < nbits; k++) rm |= 1ull << (rnd() % 64);
        if (it % 97 == 0) rm = rnd();
        unsigned want = 0;
        for (int t = 0; t < 64; t++) {
            if (!((rm >> t) & 1)) continue;
            if (t <= 5) want |= 0x80u;                            // body starting at -8 of the chunk: window [-14, 5]
            if (t >= 58) want |= 0x10000u;                        // body starting at 64: window [58, 77]
            for (int b = 0; b < 8; b++) if (t >= 8 * b - 6 && t <= 8 * b + 13) want |= 0x100u << b;
        }
        if (resume_spread(rm) != want) { if (bad++ < 5) printf("mask %016llx: got %05x want %05x\n", rm, resume_spread(rm), want); }
    }
    // 2. the chunk loop: flags streamed from the lo chunks == the up-front rule, over whole sweeps
    const int FB = 8, FCH = 64, s_first = 2 - FB;
    for (long it = 0; it < trials / 200 + 50; it++) {
        const int S = 20 + (int)(rnd() % 3000);
        std::vector<int> LO(S + 400, -1);
        int t = 2;
        while (t < S) {                                            // stretches with a band, stretches without
            const int on = 1 + (int)(rnd() % (it % 3 ? 400 : 12)), off = (int)(rnd() % (it % 2 ? 30 : 3));
            for (int k = 0; k < on && t < S; k++, t++) LO[t] = 1 + (int)(rnd() % 1000);
            t += off;
        }
        auto lo_at = [&](int s) { return s < 0 ? -1 : LO[s]; };
        // the rule (what the LDS bitmap of the plain layout holds)
        const int nbodies = (S - s_first + FB - 1) / FB + 1;
        std::vector<char> want(nbodies + 16, 0);
        for (int r = 2; r < S; r++)
            if (LO[r] >= 0 && LO[r - 1] < 0) {
                const int b0 = std::max(0, (r - 6 - s_first) >> 3), b1 = (r + 6 - s_first) >> 3;
                for (int b = b0; b <= b1; b++) want[b] = 1;
            }
        // the stream (fill_body's chunk loop): lane k of a chunk register holds lo of step cb + k
        auto chunk_bits = [&](int cb) {
            unsigned long long rm = 0;
            for (int k = 0; k < 64; k++) {
                const int cur = lo_at(cb + k), prev = lo_at(cb + k - 1);   // (lane 0 takes the step before the chunk: its last lane, or -1)
                if (cur >= 0 && prev < 0) rm |= 1ull << k;
            }
            return rm;
        };
        unsigned slow = resume_spread(chunk_bits(s_first)) >> 8;
        int nb = 0;
        for (int cb = s_first; cb < S; cb += FCH) {
            slow |= resume_spread(chunk_bits(cb + FCH));
            for (int o = 0; o < FCH && cb + o < S; o += FB, nb++) {
                const int got = (slow >> (o >> 3)) & 1;
                if (got != want[nb]) { if (bad++ < 5) printf("sweep %ld body %d (s0 %d): got %d want %d\n", it, nb, cb + o, got, (int)want[nb]); }
            }
            slow >>= 8;
        }
    }
    printf("mismatches=%ld\n", bad);
    return bad != 0;
}
