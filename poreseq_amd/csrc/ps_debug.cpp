// ps_debug.cpp — ps_debug_fill: dump one event's dense DP matrices (test hook).
#include <cmath>

#include "ps_host.h"
#include "ps_sweep_body.h"   // layout and predicate bits of the strip sweeps' code bytes

namespace ps {

// Alignment::update (cpp/Alignment.cpp:63-73) for one event, then un-skew the device matrices.
int debug_fill(Runtime* rt, Align* a, int ev, int dir, double* main, double* stay, uint8_t* sm, uint8_t* ss) {
    std::vector<JobSpec> specs(1);
    specs[0].a = a; specs[0].ev = ev; specs[0].states = &a->states; specs[0].out = a->d_out + ev;
    specs[0].ra = a->d_ra + a->off[ev]; specs[0].rl = a->d_rl + a->off[ev]; specs[0].ri = a->d_ri + a->off[ev];
    Batch b;
    PS_TRY(b.build(rt, specs, 2, 0));
    PS_TRY(realign(rt, b));
    a->host_refs_valid = false;
    const JobD& J = b.jobs[0];
    const int n0 = J.n0, C = J.C, P = J.P;
    const size_t ld = (size_t)C + 1, tot = ((size_t)n0 + 1) * ld;
    const double nan = std::nan("");
    for (size_t k = 0; k < tot; k++) { main[k] = nan; if (stay) stay[k] = nan; if (sm) sm[k] = 0; if (ss) ss[k] = 0; }
    JobOut o;
    PS_HIP(hipMemcpyAsync(&o, a->d_out + ev, sizeof(o), hipMemcpyDeviceToHost, rt->stream));
    std::vector<int> lb(J.lbn);
    PS_HIP(hipMemcpyAsync(lb.data(), b.d.lb + J.lb_off, J.lbn * sizeof(int), hipMemcpyDeviceToHost, rt->stream));
    // skewed matrices of k_fill (REC[i + j][i mod P], u16 step words) or strip matrices of k_sweep2 (REC[j + q][r][q mod NL], one code byte per cell)
    const int K = J.K;
    const SweepJob sj0 = K ? b.sjobs[0] : SweepJob();
    const int NL = K ? J.NL : 64;   // lanes of a strip sweep (64 per wavefront)
    const size_t ncell = K ? (size_t)sj0.T * NL * K : (size_t)J.S * P;
    std::vector<double2> rec(ncell);
    std::vector<unsigned short> flg(K ? 0 : ncell);
    std::vector<unsigned char> codes(K ? ncell : 0);
    PS_HIP(hipMemcpyAsync(rec.data(), b.d.rec + J.mat_off[dir], rec.size() * sizeof(double2), hipMemcpyDeviceToHost, rt->stream));
    if (!K) PS_HIP(hipMemcpyAsync(flg.data(), b.d.flg + J.mat_off[dir], flg.size() * sizeof(unsigned short), hipMemcpyDeviceToHost, rt->stream));
    else PS_HIP(hipMemcpyAsync(codes.data(), b.sd.codes + sj0.codes_off, codes.size(), hipMemcpyDeviceToHost, rt->stream));
    PS_HIP(hipStreamSynchronize(rt->stream));
    auto band_of_col = [&](int c, int& i0, int& i1) {   // rows of column c (0: the blank column, all rows)
        if (c < 1) { i0 = 0; i1 = n0; return; }
        int ce;
        if (dir == 0) { int v = lb[c]; ce = v < 0 ? 1 : v; }
        else { int v = lb[C - c + 1]; ce = v < 0 ? 1 : n0 - v + 1; }
        ce = std::min(std::max(ce, 1), n0);
        i0 = std::max(1, ce - J.W); i1 = std::min(n0, ce + J.W);
    };
    // column 0 is the blank column: rows 0..n0, all zero (cpp/Alignment.cpp:42-43)
    for (int i = 0; i <= n0; i++) { main[(size_t)i * ld] = 0.0; if (stay) stay[(size_t)i * ld] = 0.0; }
    if (o.inert) return PS_OK;
    for (int c = 1; c <= C; c++) {
        int ce;
        if (dir == 0) { int v = lb[c]; ce = v < 0 ? 1 : v; }
        else { int v = lb[C - c + 1]; ce = v < 0 ? 1 : n0 - v + 1; }
        ce = std::min(std::max(ce, 1), n0);
        const int i0 = std::max(1, ce - J.W), i1 = std::min(n0, ce + J.W);
        for (int i = i0; i <= i1; i++) {
            const size_t to = (size_t)i * ld + c;
            if (K) {
                const int q = (i - 1) / K, r = (i - 1) % K;
                const size_t at = ((size_t)(c + q) * K + r) * NL + (q & (NL - 1));
                main[to] = rec[at].x;
                if (stay) stay[to] = rec[at].y;
                if (dir == 0 && (sm || ss)) {
                    // raw predicate byte -> the reference's step codes (ps_sweep_body.h); a column without a 5-mer has no steps
                    const unsigned by = a->states[c - 1] < 0 ? 0u : code_fetch(codes.data() + (size_t)(c + q) * NL * K, K, q & (NL - 1), r, NL);
                    int p0, p1;
                    band_of_col(c - 1, p0, p1);
                    if (sm) sm[to] = (uint8_t)code_main_step(by, i > p0 && i <= p1);
                    if (ss) ss[to] = (uint8_t)code_stay_step(by);
                }
                continue;
            }
            const size_t at = (size_t)(i + c) * P + (i % P);
            main[to] = rec[at].x;
            if (stay) stay[to] = rec[at].y;
            // back-pointer codes exist for the forward matrix only (nothing reads the backward ones)
            if (sm && dir == 0) sm[to] = (uint8_t)(flg[at] & 255);
            if (ss && dir == 0) ss[to] = (uint8_t)((flg[at] >> 8) & 7);   // bits 14/15 are the backtrace's sign flags
        }
    }
    return PS_OK;
}

}  // namespace ps
