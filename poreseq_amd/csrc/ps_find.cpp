// ps_find.cpp — FindMutations (cpp/FindMutations.cpp:24-186), MapAlignments (cpp/EventUtil.cpp:12-55),
// fillinds (cpp/swlib.cpp:342-365) and the host part of ViterbiMutate (cpp/Viterbi.cpp:239-426).
// The alignments, Smith-Waterman matrices and the Viterbi recursion run on the GPU; what is left
// here is the reference's list / index bookkeeping.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "ps_host.h"
#include "ps_sw.h"

namespace ps {

// cap on the DP-matrix bytes of one seed batch (PORESEQ_MAX_BATCH_GB, default 48): lower it when several host
// threads share one GPU
static double max_batch_bytes() {
    static const double v = [] { const char* e = getenv("PORESEQ_MAX_BATCH_GB"); const double g = e ? atof(e) : 48.0; return (g > 0 ? g : 48.0) * 1e9; }();
    return v;
}

static void fillinds(SwResult& al) {  // cpp/swlib.cpp:342-365
    if (al.a.empty()) return;
    int i1 = al.a[0], i2 = al.b[0];
    for (size_t k = 0; k < al.a.size(); k++) {
        if (al.a[k] > 0) i1 = al.a[k]; else al.a[k] = i1;
        if (al.b[k] > 0) i2 = al.b[k]; else al.b[k] = i2;
    }
}

static int argmax(const std::vector<double>& v) { return (int)(std::max_element(v.begin(), v.end()) - v.begin()); }

int find_mutations(Runtime* rt, Align* a, const std::vector<std::string>& seeds, std::vector<Mut>* out) {
    out->clear();
    const size_t L = a->bases.size();
    // re-align to the current sequence, keeping per-base cumulative likelihoods (cpp/FindMutations.cpp:28-29)
    std::vector<double> base(std::max<size_t>(L, 4) + 1, 0.0), sc(std::max(a->E, 1));
    Tick tk("find_mutations");
    const int S = (int)seeds.size();
    // Smith-Waterman of the current sequence against every seed (MapAlignments, cpp/EventUtil.cpp:16):
    // independent of the re-alignment below, so it runs concurrently on the second stream
    std::vector<std::pair<const std::string*, const std::string*>> pairs;
    for (const std::string& s : seeds) pairs.push_back({&a->bases, &s});
    SwJob swjob;
    PS_TRY(sw_launch(rt, pairs, &swjob));
    tk.lap("sw enqueue");
    const int rc_base = score_alignments(rt, a, sc.data(), base.data());
    tk.lap("base realign");
    std::vector<SwResult> als;
    PS_TRY(sw_finish(rt, &swjob, &als));   // always drain the second stream, even on failure above
    PS_TRY(rc_base);
    if (!S) return PS_OK;
    for (SwResult& al : als) fillinds(al);
    tk.lap("smith-waterman");
    // seeds whose likelihood vector is not cached yet get (seed x event) alignment jobs
    std::vector<int> need;
    {
        std::map<std::string, int> seen;
        for (int k = 0; k < S; k++) {
            if (!a->seqlikes[seeds[k]].empty()) continue;
            if (seen.count(seeds[k])) continue;
            seen[seeds[k]] = k;
            need.push_back(k);
        }
    }
    if (!need.empty() && a->E) {
        PS_TRY(a->refs_to_host(rt));
        std::vector<std::vector<int>> sstates(need.size());
        for (size_t q = 0; q < need.size(); q++) sstates[q] = states_of(seeds[need[q]]);
        // chunk the seeds so the DP matrices of one batch stay below ~48 GB
        const int W = a->par.realign_width;
        const int P = std::max(64, ((W + 1 + 63) / 64) * 64);  // typical anti-diagonal footprint is about half the band
        size_t q0 = 0;
        while (q0 < need.size()) {
            size_t q1 = q0;
            double bytes = 0;
            while (q1 < need.size()) {
                double add = 0;
                for (int e = 0; e < a->E; e++) add += ((double)a->n[e] + sstates[q1].size() + 1) * P * 26.0;
                if (q1 > q0 && bytes + add > max_batch_bytes()) break;
                bytes += add; q1++;
            }
            const size_t ns = q1 - q0;
            // remapped ref_align of every (seed, event) job, cpp/EventUtil.cpp:22-51
            const size_t nref = (size_t)ns * a->ntot;
            const size_t stage_mark = rt->stage.mark();
            double* h_ra = (double*)rt->stage.alloc(std::max<size_t>(nref, 1) * sizeof(double));   // pinned: plain enqueue below
            if (!h_ra) return fail(PS_ERR_NOMEM, "hipHostMalloc (staging arena)");
            auto remap_seed = [&](size_t q) {
                const SwResult& al = als[need[q]];
                double* dst = h_ra + (q - q0) * a->ntot;
                for (int64_t t = 0; t < a->ntot; t++) {
                    const int ra = (int)a->h_ra[t];
                    double v = 0.0;
                    if (!al.a.empty() && !(ra < al.a.front() || ra > al.a.back())) {
                        const size_t k = std::lower_bound(al.a.begin(), al.a.end(), ra) - al.a.begin();
                        v = k < al.b.size() ? (double)al.b[k] : 0.0;
                    }
                    dst[t] = v;
                }
            };
            {   // one host thread per seed (pure index arithmetic on disjoint outputs)
                std::vector<std::thread> th;
                for (size_t q = q0 + 1; q < q1; q++) th.emplace_back(remap_seed, q);
                remap_seed(q0);
                for (std::thread& x : th) x.join();
            }
            tk.lap("remap (host)");
            DBuf& rb = rt->buf("seed_refs");
            PS_TRY(rb.ensure((size_t)3 * ns * std::max<int64_t>(a->ntot, 1) * sizeof(double)));
            double* d_ra = rb.as<double>();
            double* d_rl = d_ra + ns * a->ntot;
            double* d_ri = d_rl + ns * a->ntot;
            if (nref) PS_HIP(hipMemcpyAsync(d_ra, h_ra, nref * sizeof(double), hipMemcpyHostToDevice, rt->stream));
            PS_HIP(hipMemsetAsync(d_rl, 0, ns * a->ntot * sizeof(double), rt->stream));
            std::vector<JobSpec> specs;
            for (size_t q = q0; q < q1; q++)
                for (int e = 0; e < a->E; e++) {
                    JobSpec s;
                    s.a = a; s.ev = e; s.states = &sstates[q];
                    const size_t o = (q - q0) * a->ntot + a->off[e];
                    s.ra = d_ra + o; s.rl = d_rl + o; s.ri = d_ri + o;
                    specs.push_back(s);
                }
            DBuf& ob = rt->buf("seed_out");
            PS_TRY(ob.ensure(specs.size() * sizeof(JobOut)));   // (before the out pointers are taken: ensure() may move the buffer)
            PS_HIP(hipMemsetAsync(ob.p, 0, specs.size() * sizeof(JobOut), rt->stream));
            for (size_t k = 0; k < specs.size(); k++) specs[k].out = ob.as<JobOut>() + k;
            Batch b;
            PS_TRY(b.build(rt, specs, 1, 0));
            PS_TRY(launch_updaterefs(rt, b.d));  // MapAlignments ends with updaterefs (cpp/EventUtil.cpp:51)
            tk.lap("seed batch build");
            PS_TRY(realign(rt, b));
            PS_HIP(hipStreamSynchronize(rt->stream));
            tk.lap("seed realign");
            double *r_ra = nullptr, *r_rl = nullptr;
            PS_TRY(rt->down(&r_ra, d_ra, nref));
            PS_TRY(rt->down(&r_rl, d_rl, nref));
            PS_HIP(hipStreamSynchronize(rt->stream));
            tk.lap("seed D2H");
            std::vector<std::vector<double>> lks(ns);
            auto likes_seed = [&](size_t q) {
                const std::string& sd = seeds[need[q]];
                std::vector<double>& lk = lks[q - q0];
                lk.assign(std::max<size_t>(sd.size(), 4) + 1, 0.0);
                for (int e = 0; e < a->E; e++) {
                    const size_t o = (q - q0) * a->ntot + a->off[e];
                    accumulate_likes(r_ra + o, r_rl + o, a->n[e], (int)sstates[q].size(), lk.data());
                }
                lk.resize(sd.size());
            };
            {
                std::vector<std::thread> th;
                for (size_t q = q0 + 1; q < q1; q++) th.emplace_back(likes_seed, q);
                likes_seed(q0);
                for (std::thread& x : th) x.join();
            }
            for (size_t q = q0; q < q1; q++) a->seqlikes[seeds[need[q]]] = std::move(lks[q - q0]);
            rt->stage.release(stage_mark);   // the stream is idle: this batch's staging memory can be reused
            q0 = q1;
        }
    } else if (!need.empty()) {
        for (int k : need) a->seqlikes[seeds[k]] = std::vector<double>(seeds[k].size(), 0.0);
    }
    tk.lap("seed likes");
    // per seed: likelihood differences along the pairwise alignment -> clamped CUSUM (cpp/FindMutations.cpp:51-98)
    std::vector<std::vector<double>> dl(S);
    for (int k = 0; k < S; k++) {
        SwResult& al = als[k];
        const std::vector<double>& rl = a->seqlikes[seeds[k]];
        for (size_t q = 0; q < al.a.size(); q++) { al.a[q] -= 2; al.b[q] -= 2; }
        while (!al.a.empty() && (al.a[0] < 0 || al.b[0] < 0)) { al.a.erase(al.a.begin()); al.b.erase(al.b.begin()); }
        const size_t n = al.a.size();
        std::vector<double> x(n), y(n);
        for (size_t q = 0; q < n; q++) {
            x[q] = (size_t)al.a[q] < base.size() ? base[al.a[q]] : 0.0;
            y[q] = (size_t)al.b[q] < rl.size() ? rl[al.b[q]] : 0.0;
        }
        for (size_t q = n; q-- > 1;) { x[q] -= x[q - 1]; y[q] -= y[q - 1]; }
        if (n) { x[0] = 0; y[0] = 0; }
        std::vector<double>& cs = dl[k];
        cs.resize(n);
        double run = 0;
        for (size_t q = 0; q < n; q++) {
            run += y[q] - x[q];
            if (run < 0) run = 0;
            cs[q] = run;
            if (std::fabs(x[q] - y[q]) < 1e-5) cs[q] = 0;
        }
    }
    tk.lap("cusum");
    // greedy extraction of candidate edits (cpp/FindMutations.cpp:111-183).  The reference rescans every
    // seed's vector for its maximum on each round; here per-seed block maxima (128 entries per block)
    // are kept current instead — same first-maximum semantics (std::max_element), same output.
    const size_t BLK = 128;
    std::vector<std::vector<double>> bmax(S);
    auto block_refresh = [&](int k, size_t blk) {
        const std::vector<double>& v = dl[k];
        const size_t lo = blk * BLK, hi = std::min(v.size(), lo + BLK);
        double m = v[lo];
        for (size_t q = lo + 1; q < hi; q++) if (v[q] > m) m = v[q];
        bmax[k][blk] = m;
    };
    auto seed_argmax = [&](int k) -> int {   // index of the first maximum of dl[k]
        const std::vector<double>& bm = bmax[k];
        size_t bb = 0;
        for (size_t q = 1; q < bm.size(); q++) if (bm[q] > bm[bb]) bb = q;
        const std::vector<double>& v = dl[k];
        const size_t lo = bb * BLK, hi = std::min(v.size(), lo + BLK);
        size_t at = lo;
        for (size_t q = lo + 1; q < hi; q++) if (v[q] > v[at]) at = q;
        return (int)at;
    };
    std::vector<double> top(S, 0.0);
    std::vector<int> topi(S, 0);
    for (int k = 0; k < S; k++) {
        if (dl[k].empty()) continue;
        bmax[k].resize((dl[k].size() + BLK - 1) / BLK);
        for (size_t blk = 0; blk < bmax[k].size(); blk++) block_refresh(k, blk);
        topi[k] = seed_argmax(k);
        top[k] = dl[k][topi[k]];
    }
    while (out->size() < L / 3) {
        const int w = argmax(top);
        std::vector<double>& v = dl[w];
        if (v.empty()) break;
        const int ind = topi[w];
        if (v[ind] < 0.25) break;
        int i1 = (int)(std::find(v.begin() + ind, v.end(), 0.0) - v.begin());
        int i0 = -1;
        for (int q = ind; q >= 0; q--) if (v[q] == 0) { i0 = q; break; }
        if (i0 < 0) i0 = 0;
        if (i1 < 0) i1 = 0;
        if ((size_t)i0 >= v.size()) i0 = (int)v.size() - 1;
        if ((size_t)i1 >= v.size()) i1 = (int)v.size() - 1;
        const int s1 = als[w].a[i0], s2 = als[w].b[i0], e1 = als[w].a[ind], e2 = als[w].b[ind];
        Mut m;
        m.start = s1;
        if ((size_t)s1 > a->bases.size() || (size_t)s2 > seeds[w].size())
            return fail(PS_ERR_BAD_ARG, "FindMutations: alignment index outside the sequence");
        m.orig = a->bases.substr(s1, (size_t)(e1 - s1));
        m.mut = seeds[w].substr(s2, (size_t)(e2 - s2));
        while (!m.orig.empty() && !m.mut.empty() && m.orig.front() == m.mut.front()) {
            m.orig.erase(m.orig.begin()); m.mut.erase(m.mut.begin()); m.start++;
        }
        while (!m.orig.empty() && !m.mut.empty() && m.orig.back() == m.mut.back()) { m.orig.pop_back(); m.mut.pop_back(); }
        if (!m.orig.empty() || !m.mut.empty()) out->push_back(m);
        std::fill(v.begin() + i0, v.begin() + i1 + 1, 0.0);
        for (size_t blk = (size_t)i0 / BLK; blk <= (size_t)i1 / BLK; blk++) block_refresh(w, blk);
        topi[w] = seed_argmax(w);
        top[w] = v[topi[w]];
    }
    tk.lap("extract");
    return PS_OK;
}

// ---------------------------------------------------------------------------------------------
inline int succ(int st, int k, int j) { return ((st << (2 * j)) & (NS - 1)) + k; }   // cpp/Viterbi.h:31-32
inline char base_at(int st, int k) { return "ACGT"[3 & (st >> (2 * (4 - k)))]; }    // cpp/Viterbi.h:35-39

// StatesToSequence, cpp/Viterbi.cpp:171-237
static std::string path_to_bases(const std::vector<int>& st) {
    std::string s;
    int cur = st[0];
    s.push_back(base_at(cur, 0));
    for (size_t i = 1; i < st.size(); i++) {
        if (cur == st[i]) continue;
        bool hit = false;
        for (int n = 1; n <= 4 && !hit; n++)
            for (int k = 0; k < (1 << (2 * n)); k++)
                if (succ(cur, k, n) == st[i]) {
                    for (int b = 1; b <= n; b++) s.push_back(base_at(cur, b));
                    cur = st[i]; hit = true; break;
                }
        if (!hit) { cur = st[i]; s.push_back(base_at(cur, 0)); }
    }
    for (int b = 1; b <= 4; b++) s.push_back(base_at(cur, b));
    return s;
}

// rand() of a fresh process, per host thread (include/poreseq_hip.h, ps_srand): glibc's reentrant random_r() on
// a 128-byte TYPE_3 state is the generator behind rand() minus the process-wide lock.
namespace {
struct RandState {
    random_data rd;
    char st[128];
    bool init = false;
};
thread_local RandState t_rand;
}  // namespace
void rand_seed(unsigned seed) {
    memset(&t_rand.rd, 0, sizeof(t_rand.rd));
    initstate_r(seed, t_rand.st, sizeof(t_rand.st), &t_rand.rd);
    t_rand.init = true;
}
int rand_next() {
    if (!t_rand.init) rand_seed(1);
    int32_t r = 0;
    random_r(&t_rand.rd, &r);
    return (int)r;
}

int viterbi_mutate(Runtime* rt, Align* a, int nkeep, double skip, double stay, double mmin, double mmax,
                   std::vector<std::string>* out) {
    out->clear();
    Tick tk("viterbi_mutate");
    const int E = a->E;
    // host mirrors of ref_align / ref_index / refstart / refend
    PS_TRY(a->refs_to_host(rt));
    double* h_ri = nullptr;
    JobOut* info = nullptr;
    PS_TRY(rt->down(&h_ri, a->d_ri, (size_t)a->ntot));
    PS_TRY(rt->down(&info, a->d_out, (size_t)E));
    PS_HIP(hipStreamSynchronize(rt->stream));
    tk.lap("refs D2H");
    // first level whose ref_index equals an integer position (std::find in getrefstates, cpp/EventData.h:192),
    // as a flat table per event indexed by position (positions outside [0, maxpos] never match)
    // (extrapolated ref_index values past refend can be integers too and do take part, so the table
    // spans every integer-valued entry)
    int maxpos = 0;
    for (int e = 0; e < E; e++) {
        if (!info[e].has_index) continue;
        const double* ri = h_ri + a->off[e];
        for (int t = 0; t < a->n[e]; t++) {
            const double v = ri[t];
            if (v >= 0.0 && v < 1e9 && v == std::floor(v)) maxpos = std::max(maxpos, (int)v);
        }
    }
    std::vector<std::vector<int>> first(E);
    for (int e = 0; e < E; e++) {
        if (!info[e].has_index) continue;  // empty ref_index: getrefstates finds nothing
        first[e].assign((size_t)maxpos + 1, -1);
        const double* ri = h_ri + a->off[e];
        for (int t = 0; t < a->n[e]; t++) {
            const double v = ri[t];
            if (v >= 0.0 && v <= (double)maxpos && v == std::floor(v)) {
                int& slot = first[e][(size_t)v];
                if (slot < 0) slot = t;
            }
        }
    }
    tk.lap("first-index maps");
    auto rstart = [&](int e) { return info[e].has_index ? info[e].refstart : -1; };
    auto rend = [&](int e) { return info[e].has_index ? info[e].refend : -1; };
    int refind = rstart(0);
    for (int e = 0; e < E; e++) refind = std::min(refind, rstart(e));
    std::vector<double> obsin;
    int T = 0;
    while (true) {
        int nl = 0, nal = 0;
        const size_t at = obsin.size();
        obsin.resize(at + (size_t)E * 4, 0.0);
        for (int e = 0; e < E; e++) {
            if (refind < 0 || refind > maxpos || first[e].empty() || first[e][refind] < 0) continue;
            const double* ra = a->h_ra.data() + a->off[e];
            const double* mean = a->h_mean.data() + a->off[e];
            const double* stdv = a->h_stdv.data() + a->off[e];
            int t = first[e][refind], cnt = 1;
            double lvl = mean[t], sd = stdv[t];
            // getrefstates keeps following levels while ref_align <= refind, using those > 0 (cpp/EventData.h:197-201)
            double lsum = 0, ssum = 0;
            lsum += mean[t]; ssum += stdv[t];
            for (t++; t < a->n[e] && ra[t] <= refind; t++)
                if (ra[t] > 0) { lsum += mean[t]; ssum += stdv[t]; cnt++; }
            lvl = lsum / cnt; sd = ssum / cnt;
            nl++;
            double* o = obsin.data() + at + (size_t)e * 4;
            o[0] = lvl; o[1] = sd; o[2] = std::log(sd); o[3] = 1.0;
        }
        for (int e = 0; e < E; e++) if (refind >= rstart(e) && refind <= rend(e)) nal++;
        if (nl <= nal * 0.2) {
            obsin.resize(at);
            if (nal == 0) break;
            refind++;
            continue;
        }
        T++;
        refind++;
    }
    tk.lap("gather levels");
    if (T == 0) return PS_OK;
    // uniform deviates in the reference's call order: for each kept path, one per back-step (cpp/Viterbi.cpp:108)
    auto draw = [](double* rnd, size_t n) { for (size_t k = 0; k < n; k++) rnd[k] = rand_next() / (double(RAND_MAX) + 1); };
    std::vector<std::vector<int>> paths;
    PS_TRY(viterbi_device(rt, E, T, obsin.data(), a->d_model, nkeep, skip, stay, mmin, mmax, draw, &paths));
    tk.lap("device");
    for (auto& p : paths) out->push_back(path_to_bases(p));
    tk.lap("paths to bases");
    return PS_OK;
}

}  // namespace ps
