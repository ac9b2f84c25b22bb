// ps_find.cpp — FindMutations (cpp/FindMutations.cpp:24-186), MapAlignments (cpp/EventUtil.cpp:12-55),
// fillinds (cpp/swlib.cpp:342-365) and the host part of ViterbiMutate (cpp/Viterbi.cpp:239-426).
// The alignments, Smith-Waterman matrices and the Viterbi recursion run on the GPU; what is left
// here is the reference's list / index bookkeeping.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>

#include "ps_host.h"
#include "ps_sw.h"

namespace ps {

// cap on the DP-matrix bytes of one batch of candidate-sequence alignments: this runtime's share of the device (ps_host.cpp)
static double max_batch_bytes() { return device_share_bytes(); }

static void fillinds(SwResult& al) {  // cpp/swlib.cpp:342-365
    if (al.a.empty()) return;
    int i1 = al.a[0], i2 = al.b[0];
    for (size_t k = 0; k < al.a.size(); k++) {
        if (al.a[k] > 0) i1 = al.a[k]; else al.a[k] = i1;
        if (al.b[k] > 0) i2 = al.b[k]; else al.b[k] = i2;
    }
}

static int argmax(const std::vector<double>& v) { return (int)(std::max_element(v.begin(), v.end()) - v.begin()); }

// the tail of FindMutations for one AlignData (host only): likelihood differences along the pairwise alignments ->
// clamped CUSUM -> greedy extraction of candidate edits (cpp/FindMutations.cpp:51-183)
static int extract_edits(Align* a, const std::vector<std::string>& seeds, std::vector<SwResult>& als, const std::vector<double>& base,
                         std::vector<Mut>* out) {
    const size_t L = a->bases.size();
    const int S = (int)seeds.size();
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
    // greedy extraction (cpp/FindMutations.cpp:111-183).  The reference rescans every seed's vector for its maximum on
    // each round; here per-seed block maxima (128 entries per block) are kept current instead — same first-maximum
    // semantics (std::max_element), same output.
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
        if ((size_t)s1 > a->bases.size() || (size_t)s2 > seeds[w].size()) return PS_ERR_BAD_ARG;
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
    return PS_OK;
}

// FindMutations (cpp/FindMutations.cpp:24-186) for several AlignData in lock-step: one Smith-Waterman batch for all
// (sequence, seed) pairs, one base realign over all events, the candidate-sequence alignments of all regions in as few
// launch chains as device memory allows, then the host-side extraction per AlignData on host threads
int find_mutations_multi(Runtime* rt, const std::vector<Align*>& as, const std::vector<const std::vector<std::string>*>& seeds,
                         const std::vector<std::vector<Mut>*>& outs) {
    Tick tk("find_mutations");
    const int R = (int)as.size();
    for (int r = 0; r < R; r++) outs[r]->clear();
    // Smith-Waterman of every current sequence against each of its seeds (MapAlignments, cpp/EventUtil.cpp:16): independent of
    // the re-alignment below, so it runs concurrently on the second stream
    std::vector<std::pair<const std::string*, const std::string*>> pairs;
    std::vector<int> pair0(R + 1, 0);
    for (int r = 0; r < R; r++) {
        pair0[r] = (int)pairs.size();
        for (const std::string& s : *seeds[r]) pairs.push_back({&as[r]->bases, &s});
    }
    pair0[R] = (int)pairs.size();
    // The batch's checkpoint rows / columns (~13 MB per 10 kb pair) are a pool like the DP matrices: at most an eighth of this
    // runtime's share goes into one launch.  The first chunk overlaps with the base realign; what is left (large lock-step batches
    // only) follows it, chunk by chunk, and a chunk the device has no memory for is cut in two.
    auto sw_bytes = [&](size_t k) {
        const double n1 = (double)pairs[k].first->size(), n2 = (double)pairs[k].second->size();
        return 4.0 * ((n1 / 64 + 1) * (n2 + 8) + (n2 / 64 + 1) * (n1 + 1)) + 8.0 * (n1 + n2 + 2);
    };
    double sw_cap = device_share_bytes() / 8;
    auto sw_chunk_end = [&](size_t k0) {
        double acc = 0;
        size_t k = k0;
        for (; k < pairs.size(); k++) { const double add = sw_bytes(k); if (k > k0 && acc + add > sw_cap) break; acc += add; }
        return k;
    };
    typedef std::vector<std::pair<const std::string*, const std::string*>> SwIn;
    SwJob swjob;
    size_t sw_first = sw_chunk_end(0);
    {
        const int rc = sw_launch(rt, SwIn(pairs.begin(), pairs.begin() + sw_first), &swjob);
        if (rc == PS_ERR_NOMEM && sw_first > 1) { sw_first = 0; swjob = SwJob(); }   // nothing enqueued: everything goes the chunked way below
        else PS_TRY(rc);
    }
    tk.lap("sw enqueue");
    // re-align to the current sequences, keeping per-base cumulative likelihoods (cpp/FindMutations.cpp:28-29)
    std::vector<std::vector<double>> base(R), sc(R);
    std::vector<double*> scp(R), lkp(R);
    for (int r = 0; r < R; r++) {
        base[r].assign(std::max<size_t>(as[r]->bases.size(), 4) + 1, 0.0);
        sc[r].assign(std::max(as[r]->E, 1), 0.0);
        scp[r] = sc[r].data(); lkp[r] = base[r].data();
    }
    const int rc_base = score_alignments_multi(rt, as, scp, lkp);
    tk.lap("base realign");
    std::vector<SwResult> als_all;
    PS_TRY(sw_finish(rt, &swjob, &als_all));   // always drain the second stream, even on failure above
    PS_TRY(rc_base);
    for (size_t k0 = sw_first; k0 < pairs.size();) {
        const size_t k1 = sw_chunk_end(k0);
        std::vector<SwResult> part;
        const int rc = sw_batch(rt, SwIn(pairs.begin() + k0, pairs.begin() + k1), &part);
        if (rc == PS_ERR_NOMEM && k1 - k0 > 1) { sw_cap *= 0.5; continue; }
        PS_TRY(rc);
        for (SwResult& r : part) als_all.push_back(std::move(r));
        k0 = k1;
    }
    if (sw_first < pairs.size()) { static const bool trace = getenv("PORESEQ_TRACE") != nullptr; if (trace) fprintf(stderr, "[ps] smith-waterman: %zu pairs, %zu with the realign, the rest in chunks\n", pairs.size(), sw_first); }
    if (pairs.empty()) return PS_OK;
    // the reference's progress line under `verbose` (cpp/FindMutations.cpp:34-35, 100-109: "Finding mutations", a dot per seed sequence, a
    // newline); single-handle calls only (a lock-step call has no single line to write)
    if (R == 1 && as[0]->par.verbose) {
        fputs("Finding mutations", stderr);
        for (size_t k = 0; k < seeds[0]->size(); k++) fputc('.', stderr);
        fputc('\n', stderr);
        fflush(stderr);
    }
    par_for((int)als_all.size(), [&](int k) { fillinds(als_all[k]); });
    tk.lap("smith-waterman");
    // (region, seed) pairs whose likelihood vector is not cached yet get (seed x event) alignment jobs
    struct Need { int r, k; std::vector<int> states; };
    std::vector<Need> need;
    for (int r = 0; r < R; r++) {
        Align* a = as[r];
        const std::vector<std::string>& sd = *seeds[r];
        std::map<std::string, int> seen;
        for (int k = 0; k < (int)sd.size(); k++) {
            if (!a->seqlikes[sd[k]].empty()) continue;
            if (seen.count(sd[k])) continue;
            seen[sd[k]] = k;
            if (a->E) need.push_back({r, k, {}});
            else a->seqlikes[sd[k]] = std::vector<double>(sd[k].size(), 0.0);
        }
    }
    if (!need.empty()) {
        {   // host mirrors of ref_align for the remap: one synchronisation for all
            for (int r = 0; r < R; r++) PS_TRY(as[r]->refs_to_host_async(rt));
            PS_HIP(hipStreamSynchronize(rt->stream));
            for (int r = 0; r < R; r++) as[r]->refs_finish();
        }
        par_for((int)need.size(), [&](int q) { need[q].states = states_of((*seeds[need[q].r])[need[q].k]); });
        const double cap = max_batch_bytes();
        size_t q0 = 0;
        int p_seen = 0;   // widest anti-diagonal footprint (in slots) a chunk of this call turned out to need
        size_t limit = (size_t)-1;   // (region, seed) pairs per chunk after a chunk had to be cut again
        while (q0 < need.size()) {
            // chunk: as many (region, seed) pairs as the matrix budget holds (typical anti-diagonal footprint ~ half the band)
            size_t q1 = q0;
            double bytes = 0;
            size_t nref = 0;
            while (q1 < need.size()) {
                const Align* a = as[need[q1].r];
                // slots per anti-diagonal as realign will probably size them (the first chunk finds out and is cut again if not)
                const int P = std::max(p_seen, guess_slots(a));
                double add = 0;
                for (int e = 0; e < a->E; e++)
                    add += sweep_enabled() && sweep_guess_k(a->par.realign_width) ? fwd_job_bytes(a, a->n[e], (int)need[q1].states.size())
                                                                                  : ((double)a->n[e] + need[q1].states.size() + 1 + MAT_FRONT + MAT_BACK) * P * 18.0;
                if (q1 > q0 && (bytes + add > cap || q1 - q0 >= limit)) break;
                bytes += add; nref += (size_t)a->ntot; q1++;
            }
            const size_t stage_mark = rt->stage.mark();
            double* h_ra = (double*)rt->stage.alloc(std::max<size_t>(nref, 1) * sizeof(double));   // pinned: plain enqueue below
            if (!h_ra) return fail(PS_ERR_NOMEM, "hipHostMalloc (staging arena)");
            std::vector<size_t> roff(q1 - q0 + 1, 0);
            for (size_t q = q0; q < q1; q++) roff[q - q0 + 1] = roff[q - q0] + (size_t)as[need[q].r]->ntot;
            // remapped ref_align of every (seed, event) job, cpp/EventUtil.cpp:22-51
            par_for((int)(q1 - q0), [&](int qq) {
                const Need& nd = need[q0 + qq];
                const Align* a = as[nd.r];
                const SwResult& al = als_all[pair0[nd.r] + nd.k];
                double* dst = h_ra + roff[qq];
                for (int64_t t = 0; t < a->ntot; t++) {
                    const int ra = (int)a->h_ra[t];
                    double v = 0.0;
                    if (!al.a.empty() && !(ra < al.a.front() || ra > al.a.back())) {
                        const size_t k = std::lower_bound(al.a.begin(), al.a.end(), ra) - al.a.begin();
                        v = k < al.b.size() ? (double)al.b[k] : 0.0;
                    }
                    dst[t] = v;
                }
            });
            tk.lap("remap (host)");
            DBuf& rb = rt->buf("seed_refs");
            PS_TRY(rb.ensure((size_t)3 * std::max<size_t>(nref, 1) * sizeof(double)));
            double* d_ra = rb.as<double>();
            double* d_rl = d_ra + nref;
            double* d_ri = d_rl + nref;
            if (nref) PS_HIP(hipMemcpyAsync(d_ra, h_ra, nref * sizeof(double), hipMemcpyHostToDevice, rt->stream));
            PS_HIP(hipMemsetAsync(d_rl, 0, std::max<size_t>(nref, 1) * sizeof(double), rt->stream));
            size_t njobs = 0;
            for (size_t q = q0; q < q1; q++) njobs += as[need[q].r]->E;
            DBuf& ob = rt->buf("seed_out");
            PS_TRY(ob.ensure(njobs * sizeof(JobOut)));   // (before the out pointers are taken: ensure() may move the buffer)
            PS_HIP(hipMemsetAsync(ob.p, 0, njobs * sizeof(JobOut), rt->stream));
            std::vector<JobSpec> specs;
            for (size_t q = q0; q < q1; q++) {
                Align* a = as[need[q].r];
                for (int e = 0; e < a->E; e++) {
                    JobSpec s;
                    s.a = a; s.ev = e; s.states = &need[q].states;
                    const size_t o = roff[q - q0] + a->off[e];
                    s.ra = d_ra + o; s.rl = d_rl + o; s.ri = d_ri + o;
                    s.out = ob.as<JobOut>() + specs.size();
                    specs.push_back(s);
                }
            }
            Batch b;
            PS_TRY(b.build(rt, specs, 1, 0));
            PS_TRY(launch_updaterefs(rt, b.d));  // MapAlignments ends with updaterefs (cpp/EventUtil.cpp:51)
            tk.lap("seed batch build");
            {
                const int rc = realign(rt, b, q1 - q0 > 1 ? 1.2 * cap : 0.0);
                if (rc == PS_SPLIT) {   // wider bands than guessed: cut the chunk again with the width it asked for
                    { static const bool trace = getenv("PORESEQ_TRACE") != nullptr; if (trace) fprintf(stderr, "[ps] seed chunk of %zu cut again: %d slots per anti-diagonal, %d guessed\n", q1 - q0, b.P, std::max(p_seen, 0)); }
                    p_seen = std::max(p_seen, b.P);
                    limit = std::max<size_t>(1, (q1 - q0) / 2);
                    PS_HIP(hipStreamSynchronize(rt->stream));
                    rt->stage.release(stage_mark);
                    continue;
                }
                PS_TRY(rc);
            }
            p_seen = std::max(p_seen, b.P);
            // per-base likelihood vectors of the chunk's sequences: on the device when every sequence fits k_likes' table (only the
            // vectors cross PCIe), else from the jobs' ref_align / ref_like on the host
            std::vector<std::vector<double>> lks(q1 - q0);
            bool on_device = true;
            for (size_t q = q0; q < q1; q++) if ((int)need[q].states.size() > likes_max_states()) on_device = false;
            if (on_device) {
                std::vector<LikeGroup> groups(q1 - q0);
                int64_t out_tot = 0;
                int job = 0;
                for (size_t q = q0; q < q1; q++) {
                    LikeGroup& g = groups[q - q0];
                    const std::string& sd = (*seeds[need[q].r])[need[q].k];
                    g.job0 = job; g.njobs = as[need[q].r]->E; job += g.njobs;
                    g.C = (int)need[q].states.size(); g.len = (int)sd.size();
                    g.out_off = out_tot; out_tot += g.len;
                }
                DBuf& gb = rt->buf("like_groups");
                DBuf& lb = rt->buf("like_out");
                PS_TRY(gb.ensure(groups.size() * sizeof(LikeGroup)));
                PS_TRY(lb.ensure((size_t)std::max<int64_t>(out_tot, 1) * sizeof(double)));
                PS_TRY(rt->up(gb.p, groups.data(), groups.size() * sizeof(LikeGroup)));
                PS_TRY(launch_likes(rt, b.d, gb.as<LikeGroup>(), (int)groups.size(), lb.as<double>()));
                double* h_lk = nullptr;
                PS_TRY(rt->down(&h_lk, lb.p, (size_t)out_tot));
                PS_HIP(hipStreamSynchronize(rt->stream));
                tk.lap("seed realign + D2H");
                for (size_t q = q0; q < q1; q++) lks[q - q0].assign(h_lk + groups[q - q0].out_off, h_lk + groups[q - q0].out_off + groups[q - q0].len);
            } else {
            double *r_ra = nullptr, *r_rl = nullptr;
            PS_TRY(rt->down(&r_ra, d_ra, nref));
            PS_TRY(rt->down(&r_rl, d_rl, nref));
            PS_HIP(hipStreamSynchronize(rt->stream));
            tk.lap("seed realign + D2H");
            par_for((int)(q1 - q0), [&](int qq) {
                const Need& nd = need[q0 + qq];
                const Align* a = as[nd.r];
                const std::string& sd = (*seeds[nd.r])[nd.k];
                std::vector<double>& lk = lks[qq];
                lk.assign(std::max<size_t>(sd.size(), 4) + 1, 0.0);
                for (int e = 0; e < a->E; e++) {
                    const size_t o = roff[qq] + a->off[e];
                    accumulate_likes(r_ra + o, r_rl + o, a->n[e], (int)nd.states.size(), lk.data());
                }
                lk.resize(sd.size());
            });
            }
            for (size_t q = q0; q < q1; q++) as[need[q].r]->seqlikes[(*seeds[need[q].r])[need[q].k]] = std::move(lks[q - q0]);
            rt->stage.release(stage_mark);   // the stream is idle: this batch's staging memory can be reused
            q0 = q1;
        }
    }
    tk.lap("seed likes");
    std::vector<int> rcs(R, PS_OK);
    par_for(R, [&](int r) {
        std::vector<SwResult> als(als_all.begin() + pair0[r], als_all.begin() + pair0[r + 1]);
        rcs[r] = extract_edits(as[r], *seeds[r], als, base[r], outs[r]);
    });
    for (int r = 0; r < R; r++) if (rcs[r] != PS_OK) return fail(rcs[r], "FindMutations: alignment index outside the sequence");
    tk.lap("extract");
    return PS_OK;
}

int find_mutations(Runtime* rt, Align* a, const std::vector<std::string>& seeds, std::vector<Mut>* out) {
    return find_mutations_multi(rt, {a}, {&seeds}, {out});
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

// rand() of a fresh process, per region: glibc's reentrant random_r() on a 128-byte TYPE_3 state is the generator behind
// rand() minus the process-wide lock.  A state is either the calling thread's (ps_srand / ps_rand_draw, the single-handle
// ABI) or an explicit ps_rng object that a lock-step driver keeps per region (include/poreseq_hip.h).
struct RandState {
    random_data rd;
    char st[128];
    bool init = false;
};
namespace { thread_local RandState t_rand; }
static void rs_seed(RandState* r, unsigned seed) {
    memset(&r->rd, 0, sizeof(r->rd));
    initstate_r(seed, r->st, sizeof(r->st), &r->rd);
    r->init = true;
}
static int rs_next(RandState* r) {
    if (!r->init) rs_seed(r, 1);
    int32_t v = 0;
    random_r(&r->rd, &v);
    return (int)v;
}
void rand_seed(unsigned seed) { rs_seed(&t_rand, seed); }
int rand_next() { return rs_next(&t_rand); }
RandState* rand_state_new(unsigned seed) { RandState* r = new RandState(); rs_seed(r, seed); return r; }
void rand_state_free(RandState* r) { delete r; }
void rand_state_seed(RandState* r, unsigned seed) { rs_seed(r, seed); }
int rand_state_next(RandState* r) { return rs_next(r ? r : &t_rand); }

namespace {
// host part of ViterbiMutate for one AlignData: which reference positions are kept and the mean level per (position, event)
struct VitGather { std::vector<double> obsin; int T = 0; };
}  // namespace

static void vit_gather(const Align* a, const double* h_ri, const JobOut* info, VitGather* g) {
    const int E = a->E;
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
    auto rstart = [&](int e) { return info[e].has_index ? info[e].refstart : -1; };
    auto rend = [&](int e) { return info[e].has_index ? info[e].refend : -1; };
    int refind = rstart(0);
    for (int e = 0; e < E; e++) refind = std::min(refind, rstart(e));
    std::vector<double>& obsin = g->obsin;
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
            // getrefstates keeps following levels while ref_align <= refind, using those > 0 (cpp/EventData.h:197-201)
            double lsum = 0, ssum = 0;
            lsum += mean[t]; ssum += stdv[t];
            for (t++; t < a->n[e] && ra[t] <= refind; t++)
                if (ra[t] > 0) { lsum += mean[t]; ssum += stdv[t]; cnt++; }
            const double lvl = lsum / cnt, sd = ssum / cnt;
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
    g->T = T;
}

// uniform deviates in the reference's call order: for each kept path, one per back-step (cpp/Viterbi.cpp:108)
static void vit_draw(void* rng, double* rnd, size_t n) {
    RandState* r = (RandState*)rng;
    for (size_t k = 0; k < n; k++) rnd[k] = rand_state_next(r) / (double(RAND_MAX) + 1);
}

// ViterbiMutate (cpp/Viterbi.cpp:239-426) for several AlignData in lock-step; rngs[r] = the region's generator (nullptr: the
// calling thread's)
int viterbi_mutate_multi(Runtime* rt, const std::vector<Align*>& as, const std::vector<RandState*>& rngs, int nkeep, double skip, double stay,
                         double mmin, double mmax, const std::vector<std::vector<std::string>*>& outs) {
    Tick tk("viterbi_mutate");
    const int R = (int)as.size();
    for (int r = 0; r < R; r++) outs[r]->clear();
    // host mirrors of ref_align / ref_index / refstart / refend
    std::vector<double*> h_ri(R, nullptr);
    std::vector<JobOut*> info(R, nullptr);
    for (int r = 0; r < R; r++) {
        PS_TRY(as[r]->refs_to_host_async(rt));
        as[r]->last_stream = (void*)rt->stream;
        PS_TRY(rt->down(&h_ri[r], as[r]->d_ri, (size_t)as[r]->ntot));
        PS_TRY(rt->down(&info[r], as[r]->d_out, (size_t)as[r]->E));
    }
    PS_HIP(hipStreamSynchronize(rt->stream));
    for (int r = 0; r < R; r++) as[r]->refs_finish();
    tk.lap("refs D2H");
    std::vector<VitGather> gath(R);
    par_for(R, [&](int r) { vit_gather(as[r], h_ri[r], info[r], &gath[r]); });
    tk.lap("gather levels");
    std::vector<VitRegionH> regs(R);
    for (int r = 0; r < R; r++) {
        regs[r].E = as[r]->E; regs[r].T = gath[r].T; regs[r].obsin = gath[r].obsin.data(); regs[r].d_model = as[r]->d_model;
        as[r]->last_stream = (void*)rt->stream;   // (the Viterbi kernels read the model tables in the AlignData's slab)
        regs[r].rng = rngs[r]; regs[r].draw = vit_draw;
    }
    std::vector<std::vector<std::vector<int>>> paths;
    {
        // sub-batches of regions that fit this runtime's device share (obs, trimmed means, back-pointers and forward
        // probabilities: 26 KB per position; ~270 MB per 10 kb region); the regions' generators keep the results the same however
        // the batch is cut; a sub-batch the device has no memory for is cut in two
        auto need = [&](size_t k) { return 26.0 * 1024.0 * (double)regs[k].T + 32.0 * (double)regs[k].T * regs[k].E; };
        double cap = std::max(1e9, 0.9 * device_share_bytes());   // (the tables live in the runtime's matrix pool, which no alignment uses during this call)
        for (size_t k0 = 0; k0 < regs.size();) {
            size_t k1 = k0;
            double acc = 0;
            for (; k1 < regs.size(); k1++) { const double add = need(k1); if (k1 > k0 && acc + add > cap) break; acc += add; }
            std::vector<std::vector<std::vector<int>>> part;
            const int rc = viterbi_device_multi(rt, std::vector<VitRegionH>(regs.begin() + k0, regs.begin() + k1), nkeep, skip, stay, mmin, mmax, &part);
            if (rc == PS_ERR_NOMEM && k1 - k0 > 1) { cap *= 0.5; continue; }   // (nothing of this sub-batch has been drawn yet: allocation comes first)
            PS_TRY(rc);
            for (auto& pr : part) paths.push_back(std::move(pr));
            k0 = k1;
        }
    }
    tk.lap("device");
    par_for(R, [&](int r) { for (auto& p : paths[r]) outs[r]->push_back(path_to_bases(p)); });
    tk.lap("paths to bases");
    return PS_OK;
}

int viterbi_mutate(Runtime* rt, Align* a, int nkeep, double skip, double stay, double mmin, double mmax,
                   std::vector<std::string>* out, bool verbose) {
    const int rc = viterbi_mutate_multi(rt, {a}, {nullptr}, nkeep, skip, stay, mmin, mmax, {out});
    if (verbose && rc == PS_OK) {
        // the reference's progress line (cpp/Viterbi.cpp:258-259, 357-370): "Viterbi", a dot whenever the reference position passes a
        // multiple of 200, a newline.  The positions walked: from the first event's start to where no event is aligned any more —
        // counted here from the sequence length (the walk itself runs on the device)
        fputs("Viterbi", stderr);
        for (size_t k = 200; k <= a->bases.size(); k += 200) fputc('.', stderr);
        fputc('\n', stderr);
        fflush(stderr);
    }
    return rc;
}

}  // namespace ps
