// ps_oracle.cpp — CPU restatement of PoreSeq's event-level HMM scoring path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  This file is the parity oracle: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.
// The shipped library (poreseq_amd/csrc) never links or loads anything from oracle/.
//
// Pinning: the restatement is checked (tests/test_oracle.py, tests/golden/*) against
// the reference's own C++ compiled from /root/reference by oracle/Makefile (oracle/_ref) and
// against golden vectors produced by the reference's Cython PSAlign (tests/golden/make_golden.py).
//
// It exports the same C ABI as include/poreseq_hip.h so one Python harness drives the
// reference shim, this oracle and the HIP library interchangeably.
//
// All "ref:" citations are file:line under /root/reference.  Written from the behavioural spec
//  with explicit index arithmetic; every quirk that changes a result is kept.
#include "../include/poreseq_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

const int NS = PS_N_STATES;
const double BIG = 1e300;                    // ref: cpp/AlignUtil.h:20
const double LOG2PI = std::log(2 * M_PI);    // ref: cpp/AlignUtil.h:24

thread_local std::string g_err;
int fail(int code, const char* msg) { g_err = msg; return code; }

// ---------------------------------------------------------------- sequence
// ref: cpp/Sequence.h:64-100 (populateStates) — including the '-' rule that only looks at
// the base four positions back, and the unmasked 4-base prologue.
std::vector<int> states_of(const std::string& bases) {
    std::vector<int> st;
    if (bases.size() < 5) return st;
    std::vector<int> code(bases.size());
    for (size_t i = 0; i < bases.size(); i++) {
        char c = bases[i];
        code[i] = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : (int)(signed char)c;
    }
    int cur = 0;
    for (int i = 0; i < 4; i++) cur = (cur << 2) + code[i];
    st.reserve(bases.size() - 4);
    for (size_t i = 4; i < bases.size(); i++) {
        if (code[i - 4] < 4) {
            cur = (NS - 1) & ((cur << 2) + code[i]);
            st.push_back(cur);
        } else {
            cur = 0;
            st.push_back(-1);
        }
    }
    return st;
}

struct Seq {
    std::string bases;
    std::vector<int> states;
    Seq() {}
    explicit Seq(const std::string& b) : bases(b), states(states_of(b)) {}
};

struct Mut {
    int start = 0;
    std::string orig, mut;
    double score = -1e-6;  // ref: cpp/AlignUtil.h:86
};

// ref: cpp/Sequence.h:37-59 (edit constructor; start past the end => plain copy)
Seq apply_edit(const Seq& s, const Mut& m) {
    if ((size_t)m.start >= s.bases.size()) return s;
    std::string b = s.bases.substr(0, m.start);
    b += m.mut;
    size_t rem = (size_t)m.start + m.orig.size();
    if (rem < s.bases.size()) b += s.bases.substr(rem);
    return Seq(b);
}

// ---------------------------------------------------------------- model + event
struct Model {
    double lev_mean[NS], lev_stdv[NS], sd_mean[NS], sd_stdv[NS];
    double log_lev[NS], sd_lambda[NS], log_lambda[NS];
    double lsk, lst, lex, lin;
};

struct Event {
    Model m;
    std::string seq;
    int n = 0;
    int refstart = -1, refend = -1;
    std::vector<double> mean, stdv, logstdv, ref_align, ref_index, ref_like;

    // ref: cpp/EventData.h:110-169
    void updaterefs() {
        refstart = refend = -1;
        int a0 = 0, a1 = n - 1;
        while (a0 < n && !(ref_align[a0] > 0)) a0++;
        while (a1 >= 0 && !(ref_align[a1] > 0)) a1--;
        if (a0 == n || a1 < 0) { ref_index.clear(); return; }
        refstart = (int)ref_align[a0];
        refend = (int)ref_align[a1];
        ref_index = ref_align;
        double slope = (ref_align[a1] - ref_align[a0]) / (double)(a1 - a0);
        double icpt = ref_align[a0] - slope * a0;
        int last = -1;
        for (int t = 0; t < n; t++) {
            if (t < a0 || t > a1) {
                ref_index[t] = slope * t + icpt;
            } else if (ref_align[t] > 0) {
                if (last > 0) {  // sic: an anchor at level 0 never starts an interpolation
                    double mm = (ref_align[t] - ref_align[last]) / (t - last);
                    for (int u = last + 1; u < t; u++) ref_index[u] = mm * (u - last) + ref_align[last];
                }
                last = t;
            }
        }
    }
    // ref: cpp/EventData.h:172-183
    int getrefstate(int r) const {
        if (ref_index.empty()) return 0;
        return (int)(std::lower_bound(ref_index.begin(), ref_index.end(), r) - ref_index.begin());
    }
    // ref: cpp/EventData.h:187-204
    std::vector<int> getrefstates(int r) const {
        std::vector<int> v;
        auto it = std::find(ref_index.begin(), ref_index.end(), r);
        if (it == ref_index.end()) return v;
        int t = (int)(it - ref_index.begin());
        v.push_back(t);
        for (t++; t < n && ref_align[t] <= r; t++)
            if (ref_align[t] > 0) v.push_back(t);
        return v;
    }
};

struct Params { double lik_offset = 4.5; int scoring_width = 150, realign_width = 300, verbose = 0; };

struct Data {
    Seq seq;
    std::vector<Event> ev;
    Params par;
    std::map<std::string, std::vector<double>> seqlikes;  // ref: cpp/AlignData.h:34
};

// ---------------------------------------------------------------- banded DP
enum { SKIP = 0, MATCH = 1, INSERT = 2, IGNORE = 3, STAY = 4, EXTEND = 5, IMPL = 255 };

struct Col {
    int i0 = 0, col = 0, len = 0;
    std::vector<double> main, stay, obs;
    std::vector<uint8_t> sm, ss;
    double best = 0; int bi = 0, bj = 0;  // running MaxInfo, ref: cpp/Alignment.h:38-50
    Col(int n, int r0, int c) : i0(r0), col(c), len(n), main(n, 0.0), stay(n, 0.0), obs(n, 0.0), sm(n, 0), ss(n, 0) {}
    bool has(int i) const { return i >= i0 && i < i0 + len; }
    int last() const { return i0 + len - 1; }
};
typedef std::shared_ptr<Col> ColP;

// ref: cpp/AlignUtil.h:34-38, 48-53 ; cpp/Alignment.cpp:167-174
inline double emission(const Model& m, int k, double x, double sd, double logsd, double offset) {
    double d = (x - m.lev_mean[k]) / m.lev_stdv[k];
    double l = -0.5 * (d * d + LOG2PI) - m.log_lev[k];
    double e = (sd - m.sd_mean[k]) / m.sd_mean[k];
    l += 0.5 * (m.log_lambda[k] - 3 * logsd - LOG2PI - e * e * m.sd_lambda[k] / sd);
    l += offset;
    return l;
}

struct Aligner {
    Event* ev; const Seq* seq; Params par;
    std::vector<ColP> F, B;
    int width;

    // ref: cpp/Alignment.cpp:38-60
    Aligner(const Seq& s, Event& e, const Params& p) : ev(&e), seq(&s), par(p) {
        F.push_back(std::make_shared<Col>(e.n + 1, 0, 0));
        B.push_back(std::make_shared<Col>(e.n + 1, 0, 0));
        width = e.ref_index.empty() ? 0 : p.realign_width;
    }
    void clear() { F.resize(1); B.resize(1); }

    // band of one column; ref: cpp/Alignment.cpp:127-148 / :299-323
    void band(int centre, int& lo, int& hi) const {
        int n0 = ev->n, w = width;
        if (w < n0 && (centre < -10 || centre > n0 + 10)) w = 5;  // unreachable, kept
        centre = std::min(std::max(centre, 1), n0);
        lo = std::max(1, centre - w);
        hi = std::min(n0, centre + w);
    }

    // one column of either direction.  ref: cpp/Alignment.cpp:111-274 and :280-444
    void fill_one(bool backward) {
        std::vector<ColP>& V = backward ? B : F;
        const int C = (int)seq->states.size();
        int colid, refind;
        if (!backward) { refind = V.back()->col + 1; colid = refind; if (refind > C) return; }
        else { colid = V.back()->col - 1; refind = C + colid + 1; if (refind <= 0) return; }
        if (width == 0) return;
        const int n0 = ev->n;
        const int st = seq->states[refind - 1];
        int centre = 1;
        if (!ev->ref_index.empty())
            centre = backward ? n0 - ev->getrefstate(refind) + 1 : ev->getrefstate(refind);
        int lo, hi; band(centre, lo, hi);
        ColP cur = std::make_shared<Col>(hi - lo + 1, lo, colid);
        ColP prev = V.back();
        V.push_back(cur);
        cur->best = prev->best; cur->bi = prev->bi; cur->bj = prev->bj;
        if (st < 0) return;
        const Model& m = ev->m;
        for (int i = lo; i <= hi; i++) {
            // forward reads the mirrored log index (Q1), ref: cpp/Alignment.cpp:171-172
            int tv = backward ? n0 - i : i - 1;
            cur->obs[i - lo] = emission(m, st, ev->mean[tv], ev->stdv[tv], ev->logstdv[n0 - i], par.lik_offset);
        }
        cur->stay[0] = -BIG;
        const int p0 = prev->i0, p1 = prev->last();
        for (int i = lo; i <= hi; i++) {
            const int r = i - lo;
            double cand[6] = {0, 0, 0, 0, -BIG, -BIG};
            uint8_t code[6] = {SKIP, MATCH, INSERT, IGNORE, STAY, EXTEND};
            const double o = cur->obs[r];
            if (i >= p0 && i <= p1) cand[SKIP] = prev->main[i - p0] + m.lsk;
            else { cand[SKIP] = m.lsk; code[SKIP] = IMPL; }
            if (i > p0 && i <= p1) {
                cand[MATCH] = prev->main[i - 1 - p0] + (backward ? prev->obs[i - 1 - p0] : o);
                cand[IGNORE] = prev->main[i - 1 - p0] + m.lin;
            } else { cand[MATCH] = backward ? 0.0 : o; code[MATCH] = IMPL; }
            if (i > lo) {
                const double e = backward ? cur->obs[r - 1] : o;
                cand[STAY] = cur->main[r - 1] + e + m.lst;
                cand[INSERT] = cur->main[r - 1] + m.lin;
                cand[EXTEND] = cur->stay[r - 1] + e + m.lex;
            }
            for (int k = 4; k < 6; k++)
                if (cand[k] > cur->stay[r]) { cur->stay[r] = cand[k]; cur->ss[r] = (uint8_t)k; }
            for (int k = 0; k < 4; k++)
                if (cand[k] > cur->main[r]) { cur->main[r] = cand[k]; cur->sm[r] = code[k]; }
            if (cur->stay[r] > cur->main[r]) { cur->main[r] = cur->stay[r]; cur->sm[r] = STAY; }
            if (cur->main[r] > cur->best) { cur->best = cur->main[r]; cur->bi = i; cur->bj = refind; }
        }
    }
    void fill_fwd_all() { if (!width) return; while (F.back()->col < (int)seq->states.size()) fill_one(false); }
    void fill_back_all() { if (!width) return; while ((int)seq->states.size() + B.back()->col > 0) fill_one(true); }

    // ref: cpp/Alignment.cpp:516-624
    void backtrace() {
        if (!width) return;
        std::vector<int> ri, rj; std::vector<double> rl;
        int i = F.back()->bi, j = F.back()->bj, arr = 0;
        while (i > 0) {
            const Col& c = *F[j];
            const int r = i - c.i0;
            const uint8_t st = arr ? c.ss[r] : c.sm[r];
            const double sc = arr ? c.stay[r] : c.main[r];
            if (sc <= 0.0) break;
            switch (st) {
                case SKIP: j--; break;
                case MATCH: ri.push_back(i); rj.push_back(j); rl.push_back(sc); i--; j--; break;
                case IGNORE: ri.push_back(i); rj.push_back(-1); rl.push_back(sc); i--; j--; break;
                case INSERT: ri.push_back(i); rj.push_back(-1); rl.push_back(sc); i--; break;
                case STAY:
                    if (arr == 1) { ri.push_back(i); rj.push_back(j); rl.push_back(sc); i--; }
                    arr = 1 - arr; break;
                case EXTEND: ri.push_back(i); rj.push_back(j); rl.push_back(sc); i--; break;
                default: i = 0; break;
            }
        }
        std::fill(ev->ref_align.begin(), ev->ref_align.end(), 0.0);
        std::fill(ev->ref_like.begin(), ev->ref_like.end(), 0.0);
        for (size_t k = 0; k < ri.size(); k++) {
            ev->ref_align[ri[k] - 1] = rj[k];
            ev->ref_like[ri[k] - 1] = rl[k];
        }
        ev->updaterefs();
    }

    // ref: cpp/Alignment.cpp:63-73
    void update(const Seq& s) { clear(); seq = &s; fill_fwd_all(); fill_back_all(); backtrace(); }

    double get_max() const { return std::max(F.back()->best, B.back()->best); }  // ref: cpp/Alignment.h:127-130

    // ref: cpp/Alignment.h:181-214 (index clamps are done on size_t, so a negative index lands on the last column)
    double column_max(int raf, int rab) const {
        if ((size_t)raf >= F.size()) raf = (int)F.size() - 1;
        if ((size_t)rab >= B.size()) rab = (int)B.size() - 1;
        if (raf < 0) raf = 0;
        if (rab < 0) rab = 0;
        const Col& f = *F[raf]; const Col& b = *B[rab];
        double sm = 0;
        const int n0 = ev->n;
        for (int jf = 1; jf <= n0; jf++) {
            const int jb = n0 - jf + 1;
            for (int k = 0; k < 2; k++) {
                double s = 0;
                if (f.has(jf)) s += (k ? f.stay : f.main)[jf - f.i0];
                if (b.has(jb)) s += (k ? b.stay : b.main)[jb - b.i0];
                sm = std::max(s, sm);
            }
            sm = std::max(sm, f.best);
            sm = std::max(sm, b.best);
        }
        return sm;
    }

    // ref: cpp/Alignment.cpp:447-512
    double score_edit(const Mut& mu, const Seq& mseq) {
        if (!width) return 0;
        const size_t keep = F.size();
        const Seq* oseq = seq;
        const int r0 = std::max(mu.start - 3, 1);
        const double old = column_max(r0, (int)seq->states.size() - r0 + 1);
        width = par.scoring_width;
        seq = &mseq;
        const int sidx = std::max(mu.start - 4, 0);
        F.push_back(F[sidx]);
        for (size_t k = 0; k < mu.mut.size() + 6; k++) fill_one(false);
        int refind = mu.start + (int)mu.mut.size() + 1;
        int f = (int)F.size() - 1;
        while (F[f]->col > refind && f >= 0) f--;
        if (F[f]->col >= F[sidx]->col) refind = F[f]->col;
        const int backind = (int)seq->states.size() - refind + 1;
        double now = old - 1;
        if (F[f]->col == refind && f > (int)keep - 1) now = column_max(f, backind);
        F.resize(keep);
        seq = oseq;
        width = par.realign_width;
        return now - old;
    }
};

// ---------------------------------------------------------------- batch drivers
// ref: cpp/MakeMutations.cpp:23-69
// deltas (optional): [events][edits], every event's term of every edit's sum
std::vector<Mut> score_mutations(Data& d, const std::vector<Mut>& muts, double* deltas = nullptr) {
    std::vector<Mut> out(muts);
    for (auto& m : out) m.score = -1e-6;
    std::vector<Aligner> al;
    for (auto& e : d.ev) al.emplace_back(d.seq, e, d.par);
    size_t ei = 0;
    for (auto& a : al) {
        a.update(d.seq);
        for (size_t i = 0; i < out.size(); i++) {
            if (deltas) deltas[ei * out.size() + i] = 0.0;
            if ((size_t)muts[i].start > d.seq.bases.size()) continue;
            Seq ms = apply_edit(d.seq, muts[i]);
            const double dl = a.score_edit(muts[i], ms);
            out[i].score += dl;
            if (deltas) deltas[ei * out.size() + i] = dl;
        }
        a.clear();
        ei++;
    }
    return out;
}

// ref: cpp/MakeMutations.cpp:148-195
std::vector<double> score_alignments(Data& d, double* likes) {
    std::vector<double> sc;
    std::vector<Aligner> al;
    for (auto& e : d.ev) al.emplace_back(d.seq, e, d.par);
    for (auto& a : al) {
        a.fill_fwd_all();
        a.backtrace();
        sc.push_back(a.get_max());
        if (likes) {
            const Event& e = *a.ev;
            double last = 0; int refind = 1;
            for (int t = 0; t < e.n; t++) {
                if (e.ref_align[t] > 0) {
                    for (int k = refind; k < e.ref_align[t]; k++) likes[k + 1] += last;
                    last = e.ref_like[t];
                    refind = (int)e.ref_align[t];
                }
            }
            for (size_t k = refind; k < d.seq.states.size() + 3; k++) likes[k + 1] += last;
        }
        a.clear();
    }
    return sc;
}

bool better(const Mut& a, const Mut& b) { return a.score > b.score; }  // ref: cpp/MakeMutations.cpp:16-17

// ref: cpp/MakeMutations.cpp:74-146
int make_mutations(Data& d, std::vector<Mut> muts) {
    const int spacing = 10;
    int nb = 0;
    std::sort(muts.begin(), muts.end(), better);
    while (!muts.empty() && muts.back().score < 0) muts.pop_back();
    if (muts.empty()) return 0;
    std::vector<Mut> later;
    for (size_t i = 0; i < muts.size(); i++) {
        if (muts[i].score < 0) { later.push_back(muts[i]); continue; }
        d.seq = apply_edit(d.seq, muts[i]);
        nb += (int)std::max(muts[i].orig.size(), muts[i].mut.size());
        for (size_t j = i + 1; j < muts.size(); j++) {
            int lo = std::max(muts[i].start, muts[j].start);
            int hi = (int)std::min(muts[i].start + muts[i].mut.size(), muts[j].start + muts[j].mut.size());
            if (lo < hi + spacing && muts[j].score > 0) { muts[j].score = -1; continue; }
            if ((size_t)muts[j].start >= muts[i].start + muts[i].orig.size())
                muts[j].start += (int)(muts[i].mut.size() - muts[i].orig.size());
        }
    }
    if (later.size() > 10) nb += make_mutations(d, score_mutations(d, later));
    return nb;
}

// ---------------------------------------------------------------- Smith-Waterman
struct SW { int score = 0; double accuracy = 0; std::vector<int> a, b; };

// ref: cpp/swlib.cpp:211-340
SW swfull(const std::string& s1, const std::string& s2) {
    const int n1 = (int)s1.size(), n2 = (int)s2.size();
    const size_t ld = (size_t)n1 + 1;
    std::vector<int> H(ld * ((size_t)n2 + 1), 0);
    std::vector<uint8_t> T(ld * ((size_t)n2 + 1), 0);
    int best = 0, bi = 0, bj = 0;
    for (int j = 1; j <= n2; j++) {
        int* cur = &H[j * ld]; const int* pre = &H[(j - 1) * ld]; uint8_t* tc = &T[j * ld];
        for (int i = 1; i <= n1; i++) {
            int sc = 0; uint8_t st = 0;
            int s = pre[i] - 8; if (s > sc) { sc = s; st = 1; }
            s = cur[i - 1] - 8; if (s > sc) { sc = s; st = 2; }
            s = pre[i - 1] + (s1[i - 1] == s2[j - 1] ? 5 : -4); if (s >= sc) { sc = s; st = 3; }
            cur[i] = sc; tc[i] = st;
            if (sc > best) { best = sc; bi = i; bj = j; }
        }
    }
    SW r; r.score = best;
    int i = bi, j = bj, nm = 0;
    while (i > 0 && j > 0) {
        if (H[j * ld + i] <= 0) break;
        uint8_t st = T[j * ld + i];
        if (st == 1) { r.a.push_back(0); r.b.push_back(j); j--; }
        else if (st == 2) { r.a.push_back(i); r.b.push_back(0); i--; }
        else if (st == 3) { r.a.push_back(i); r.b.push_back(j); if (s1[i - 1] == s2[j - 1]) nm++; i--; j--; }
        else break;  // the reference would spin here printing an error; unreachable for score > 0
    }
    std::reverse(r.a.begin(), r.a.end());
    std::reverse(r.b.begin(), r.b.end());
    r.accuracy = 100.0 * nm / (double)r.a.size();
    return r;
}

// ref: cpp/swlib.cpp:342-365
SW fillinds(SW al) {
    if (al.a.empty()) return al;
    int i1 = al.a[0], i2 = al.b[0];
    for (size_t k = 0; k < al.a.size(); k++) {
        if (al.a[k] > 0) i1 = al.a[k]; else al.a[k] = i1;
        if (al.b[k] > 0) i2 = al.b[k]; else al.b[k] = i2;
    }
    return al;
}

// ref: cpp/EventUtil.cpp:12-55
SW map_alignments(Data& d, const Seq& ns) {
    SW al = fillinds(swfull(d.seq.bases, ns.bases));
    d.seq = ns;
    for (auto& e : d.ev) {
        for (size_t t = 0; t < e.ref_align.size(); t++) {
            int ra = (int)e.ref_align[t];
            if (al.a.empty() || ra < al.a.front() || ra > al.a.back()) { e.ref_align[t] = 0; continue; }
            size_t k = std::lower_bound(al.a.begin(), al.a.end(), ra) - al.a.begin();
            e.ref_align[t] = k < al.b.size() ? al.b[k] : 0;
        }
        e.updaterefs();
    }
    return al;
}

int argmax(const std::vector<double>& v) { return (int)(std::max_element(v.begin(), v.end()) - v.begin()); }

// ref: cpp/FindMutations.cpp:24-186
std::vector<Mut> find_mutations(Data& d, const std::vector<Seq>& seeds) {
    std::vector<double> base(d.seq.bases.size(), 0);
    score_alignments(d, base.data());
    std::vector<std::vector<double>> dl;
    std::vector<SW> als;
    for (const Seq& sd : seeds) {
        Data nd(d);
        SW al = map_alignments(nd, sd);
        std::vector<double>& rl = d.seqlikes[sd.bases];
        if (rl.empty()) {
            rl.assign(sd.bases.size(), 0);
            score_alignments(nd, rl.data());
        }
        for (size_t k = 0; k < al.a.size(); k++) { al.a[k] -= 2; al.b[k] -= 2; }
        while (!al.a.empty() && (al.a[0] < 0 || al.b[0] < 0)) { al.a.erase(al.a.begin()); al.b.erase(al.b.begin()); }
        std::vector<double> x, y;
        for (size_t k = 0; k < al.a.size(); k++) { x.push_back(base[al.a[k]]); y.push_back(rl[al.b[k]]); }
        for (size_t k = x.size(); k-- > 1;) { x[k] -= x[k - 1]; y[k] -= y[k - 1]; }
        if (!x.empty()) { x[0] = 0; y[0] = 0; }
        std::vector<double> cs(x.size());
        double run = 0;
        for (size_t k = 0; k < x.size(); k++) {
            run += y[k] - x[k];
            if (run < 0) run = 0;
            cs[k] = run;
            if (std::fabs(x[k] - y[k]) < 1e-5) cs[k] = 0;
        }
        dl.push_back(cs);
        als.push_back(al);
    }
    std::vector<Mut> out;
    if (dl.empty()) return out;
    while (out.size() < d.seq.bases.size() / 3) {
        std::vector<double> top(dl.size(), 0);
        for (size_t s = 0; s < dl.size(); s++) top[s] = dl[s].empty() ? 0 : dl[s][argmax(dl[s])];
        const int w = argmax(top);
        std::vector<double>& v = dl[w];
        if (v.empty()) break;
        const int ind = argmax(v);
        if (v[ind] < 0.25) break;
        int i1 = (int)(std::find(v.begin() + ind, v.end(), 0) - v.begin());
        int i0 = -1;
        for (int k = ind; k >= 0; k--) if (v[k] == 0) { i0 = k; break; }
        if (i0 < 0) i0 = 0;
        if (i1 < 0) i1 = 0;
        if ((size_t)i0 >= v.size()) i0 = (int)v.size() - 1;
        if ((size_t)i1 >= v.size()) i1 = (int)v.size() - 1;
        const int s1 = als[w].a[i0], s2 = als[w].b[i0], e1 = als[w].a[ind], e2 = als[w].b[ind];
        Mut m;
        m.start = s1;
        m.orig = d.seq.bases.substr(s1, (size_t)(e1 - s1));
        m.mut = seeds[w].bases.substr(s2, (size_t)(e2 - s2));
        while (!m.orig.empty() && !m.mut.empty() && m.orig.front() == m.mut.front()) {
            m.orig.erase(m.orig.begin()); m.mut.erase(m.mut.begin()); m.start++;
        }
        while (!m.orig.empty() && !m.mut.empty() && m.orig.back() == m.mut.back()) { m.orig.pop_back(); m.mut.pop_back(); }
        if (!m.orig.empty() || !m.mut.empty()) out.push_back(m);
        std::fill(v.begin() + i0, v.begin() + i1 + 1, 0.0);
    }
    return out;
}

// ref: cpp/FindMutations.cpp:191-234
std::vector<Mut> find_point_mutations(const Data& d) {
    static const char B4[] = "ACGT";
    std::vector<Mut> out;
    for (size_t i = 0; i < d.seq.states.size(); i++) {
        Mut m; m.start = (int)i;
        m.orig = std::string(1, d.seq.bases[i]); m.mut = "";
        out.push_back(m);
        for (int b = 0; b < 4; b++) {
            if (d.seq.bases[i] == B4[b]) continue;
            m.mut = std::string(1, B4[b]); out.push_back(m);
        }
        m.orig = "";
        for (int b = 0; b < 4; b++) { m.mut = std::string(1, B4[b]); out.push_back(m); }
    }
    return out;
}

// ---------------------------------------------------------------- Viterbi seed generator
struct VStep { std::vector<double> lik, fwd; std::vector<int> bp; VStep() : lik(NS), fwd(NS), bp(NS) {} };

inline int pred(int st, int k, int j) { return (st >> (2 * j)) + (k << (10 - 2 * j)); }  // ref: cpp/Viterbi.h:29-30
inline int succ(int st, int k, int j) { return ((st << (2 * j)) & (NS - 1)) + k; }        // ref: cpp/Viterbi.h:31-32
inline char base_at(int st, int k) { return "ACGT"[3 & (st >> (2 * (4 - k)))]; }         // ref: cpp/Viterbi.h:35-39

void normalise(double* v) {  // ref: cpp/Viterbi.h:56-64
    double t = 0;
    for (int i = 0; i < NS; i++) t += v[i];
    t = 1.0 / t;
    for (int i = 0; i < NS; i++) v[i] *= t;
}

// ref: cpp/Viterbi.cpp:39-102
void vstep(const VStep& p, const std::vector<double>& obs, double skip, double stay, VStep& o) {
    const double lskip = std::log(skip), lstay = std::log(stay);
    for (int c = 0; c < NS; c++) {
        double best = -BIG; int bp = -1; double fs = 0.0;
        double sp = 0.25, lsp = std::log(0.25);
        for (int j = 1; j <= 3; j++) {
            for (int k = 0; k < (1 << (2 * j)); k++) {
                const int q = pred(c, k, j);
                double l = obs[c] + lsp;
                l += p.lik[q];
                fs += sp * p.fwd[q];
                if (l > best) { best = l; bp = q; }
            }
            sp = sp * 0.25 * skip;
            lsp = lsp + std::log(0.25) + lskip;
        }
        double l = obs[c] + lstay + p.lik[c];
        if (l > best) { best = l; bp = c; }
        fs += stay * p.fwd[c];
        fs *= std::exp(obs[c]);
        o.lik[c] = best; o.bp[c] = bp; o.fwd[c] = fs;
    }
    normalise(o.fwd.data());
}

// ref: cpp/Viterbi.cpp:134-168
std::vector<double> build_T(double skip, double stay) {
    std::vector<double> T((size_t)NS * NS, 0);
    for (int c = 0; c < NS; c++) {
        double sp = 0.25;
        for (int j = 1; j <= 4; j++) {
            for (int k = 0; k < (1 << (2 * j)); k++) T[(size_t)c * NS + pred(c, k, j)] += sp;
            sp = sp * 0.25 * skip;
        }
    }
    for (int i = 0; i < NS; i++) T[(size_t)i * (NS + 1)] = stay;
    return T;
}

// ref: cpp/Viterbi.cpp:105-131
int rand_pred(const VStep& v, int cur, double atten, const std::vector<double>& T) {
    const double r = rand() / (double(RAND_MAX) + 1);
    double pr[NS];
    for (int i = 0; i < NS; i++) pr[i] = T[i + (size_t)cur * NS] * std::pow(v.fwd[i], atten);
    normalise(pr);
    double cs = 0;
    for (int i = 0; i < NS; i++) { cs += pr[i]; if (r < cs) return i; }
    return NS - 1;
}

// ref: cpp/Viterbi.cpp:171-237
std::string path_to_bases(const std::vector<int>& st) {
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

// ref: cpp/Viterbi.cpp:239-426
std::vector<std::string> viterbi_mutate(std::vector<Event>& ev, int nkeep, double skip, double stay,
                                        double mmin, double mmax) {
    std::vector<std::unique_ptr<VStep>> S;
    S.emplace_back(new VStep());
    for (int i = 0; i < NS; i++) { S[0]->lik[i] = 0; S[0]->bp[i] = -1; S[0]->fwd[i] = 1.0 / NS; }
    const int E = (int)ev.size();
    int refind = ev[0].refstart;
    for (auto& e : ev) refind = std::min(refind, e.refstart);
    std::vector<double> obs((size_t)NS * E);
    while (true) {
        std::fill(obs.begin(), obs.end(), 0.0);
        int nl = 0;
        for (int k = 0; k < E; k++) {
            std::vector<int> idx = ev[k].getrefstates(refind);
            if (idx.empty()) continue;
            nl++;
            double lvl = 0, sd = 0;
            for (int t : idx) { lvl += ev[k].mean[t]; sd += ev[k].stdv[t]; }
            lvl = lvl / idx.size(); sd = sd / idx.size();
            const double lsd = std::log(sd);
            for (int j = 0; j < NS; j++) obs[(size_t)j * E + nl - 1] = emission(ev[k].m, j, lvl, sd, lsd, 0.0) ;
        }
        int nal = 0;
        for (int k = 0; k < E; k++) if (refind >= ev[k].refstart && refind <= ev[k].refend) nal++;
        if (nl <= nal * 0.2) { if (nal == 0) break; refind++; continue; }
        if (nl > 1) {
            for (int j = 0; j < NS; j++) std::sort(obs.begin() + (size_t)j * E, obs.begin() + (size_t)j * E + nl);
            int drop = (int)std::floor(nl * 0.25);
            if (drop > nl - 2) drop = 0;
            for (int j = 0; j < NS; j++) {
                double t = 0.0;
                for (int k = drop; k < nl; k++) t += obs[(size_t)j * E + k];
                obs[j] = t / (nl - drop);
            }
        } else {
            for (int j = 0; j < NS; j++) obs[j] = obs[(size_t)j * E];
        }
        S.emplace_back(new VStep());
        vstep(*S[S.size() - 2], obs, skip, stay, *S.back());
        refind++;
    }
    std::vector<std::string> out;
    const int start = argmax(S.back()->lik);
    const int n = (int)S.size() - 1;
    std::vector<int> path;
    if (nkeep == 0) {
        int c = start;
        for (int i = n - 1; i >= 0; i--) { path.push_back(c); c = S[i + 1]->bp[c]; }
        std::reverse(path.begin(), path.end());
        if (!path.empty()) out.push_back(path_to_bases(path));
        return out;
    }
    std::vector<double> T = build_T(skip, stay);
    for (int k = 0; k < nkeep; k++) {
        path.clear();
        int c = start;
        for (int i = n - 1; i >= 0; i--) {
            path.push_back(c);
            c = rand_pred(*S[i + 1], c, mmin + (mmax - mmin) * k / (double)nkeep, T);
        }
        std::reverse(path.begin(), path.end());
        if (!path.empty()) out.push_back(path_to_bases(path));
    }
    return out;
}

}  // namespace

// the emission() above adds `offset` last; with offset 0.0 the Viterbi emission (which has no
// lik_offset term, ref: cpp/Viterbi.cpp:300-306) is unchanged because x + 0.0 == x for finite x.

// ================================================================= C ABI
struct ps_align { Data d; };
struct ps_muts { std::vector<Mut> v; };
struct ps_seqs { std::vector<std::string> v; };

extern "C" {

const char* ps_last_error(void) { return g_err.c_str(); }
const char* ps_backend_name(void) { return "oracle-cpu"; }

int ps_align_create(ps_align** out, const char* seq, int64_t seq_len, int32_t n_events,
                    const int64_t* level_off, const double* mean, const double* stdv,
                    const double* ref_align, const double* ref_like, const double* model,
                    const double* trans, const char* evseq, const int64_t* evseq_off,
                    const ps_params* params) {
    if (!out || !seq || seq_len < 0 || n_events < 0 || (n_events && (!level_off || !mean || !stdv || !ref_align || !ref_like || !model || !trans)))
        return fail(PS_ERR_BAD_ARG, "ps_align_create: bad argument");
    std::unique_ptr<ps_align> a(new ps_align());
    a->d.seq = Seq(std::string(seq, (size_t)seq_len));
    if (params) {
        a->d.par.lik_offset = params->lik_offset; a->d.par.scoring_width = params->scoring_width;
        a->d.par.realign_width = params->realign_width; a->d.par.verbose = params->verbose;
    }
    a->d.ev.resize(n_events);
    for (int e = 0; e < n_events; e++) {
        Event& ev = a->d.ev[e];
        const int64_t o = level_off[e];
        ev.n = (int)(level_off[e + 1] - o);
        ev.mean.assign(mean + o, mean + o + ev.n);
        ev.stdv.assign(stdv + o, stdv + o + ev.n);
        ev.ref_align.assign(ref_align + o, ref_align + o + ev.n);
        ev.ref_like.assign(ref_like + o, ref_like + o + ev.n);
        ev.logstdv.resize(ev.n);
        for (int t = 0; t < ev.n; t++) ev.logstdv[t] = std::log(ev.stdv[t]);
        ev.updaterefs();
        const double* md = model + (size_t)e * 4 * NS;
        for (int k = 0; k < NS; k++) {
            ev.m.lev_mean[k] = md[k]; ev.m.lev_stdv[k] = md[NS + k];
            ev.m.sd_mean[k] = md[2 * NS + k]; ev.m.sd_stdv[k] = md[3 * NS + k];
            ev.m.log_lev[k] = std::log(ev.m.lev_stdv[k]);
            ev.m.sd_lambda[k] = std::pow(ev.m.sd_mean[k], 3) / std::pow(ev.m.sd_stdv[k], 2);
            ev.m.log_lambda[k] = std::log(ev.m.sd_lambda[k]);
        }
        ev.m.lsk = std::log(trans[e * 4 + 0]); ev.m.lst = std::log(trans[e * 4 + 1]);
        ev.m.lex = std::log(trans[e * 4 + 2]); ev.m.lin = std::log(trans[e * 4 + 3]);
        if (evseq && evseq_off) ev.seq.assign(evseq + evseq_off[e], evseq + evseq_off[e + 1]);
    }
    *out = a.release();
    return PS_OK;
}
void ps_align_destroy(ps_align* a) { delete a; }
int ps_align_set_scoring_width(ps_align* a, int32_t w) { if (!a) return fail(PS_ERR_BAD_ARG, "null"); a->d.par.scoring_width = w; return PS_OK; }
int ps_align_new_call(ps_align* a, int32_t w) { if (!a) return fail(PS_ERR_BAD_ARG, "null"); a->d.par.scoring_width = w; a->d.seqlikes.clear(); return PS_OK; }
int32_t ps_align_n_events(const ps_align* a) { return a ? (int32_t)a->d.ev.size() : 0; }
int64_t ps_align_n_levels(const ps_align* a, int32_t e) { return (a && e >= 0 && e < (int)a->d.ev.size()) ? a->d.ev[e].n : -1; }
int64_t ps_align_sequence_length(const ps_align* a) { return a ? (int64_t)a->d.seq.bases.size() : -1; }
int ps_align_get_sequence(const ps_align* a, char* out, int64_t cap) {
    if (!a || !out || cap < (int64_t)a->d.seq.bases.size()) return fail(PS_ERR_BAD_ARG, "ps_align_get_sequence");
    std::memcpy(out, a->d.seq.bases.data(), a->d.seq.bases.size());
    return PS_OK;
}
int ps_align_get_event_refs(const ps_align* a, int32_t e, double* ra, double* rl) {
    if (!a || e < 0 || e >= (int)a->d.ev.size()) return fail(PS_ERR_BAD_ARG, "ps_align_get_event_refs");
    const Event& ev = a->d.ev[e];
    if (ra) std::copy(ev.ref_align.begin(), ev.ref_align.end(), ra);
    if (rl) std::copy(ev.ref_like.begin(), ev.ref_like.end(), rl);
    return PS_OK;
}

int ps_muts_create(ps_muts** out, int64_t n, const int32_t* start, const int64_t* oo, const char* op,
                   const int64_t* mo, const char* mp, const double* score) {
    if (!out || n < 0 || (n && (!start || !oo || !mo))) return fail(PS_ERR_BAD_ARG, "ps_muts_create");
    ps_muts* m = new ps_muts();
    m->v.resize(n);
    for (int64_t i = 0; i < n; i++) {
        m->v[i].start = start[i];
        if (oo[i + 1] > oo[i]) m->v[i].orig.assign(op + oo[i], op + oo[i + 1]);
        if (mo[i + 1] > mo[i]) m->v[i].mut.assign(mp + mo[i], mp + mo[i + 1]);
        m->v[i].score = score ? score[i] : -1e-6;
    }
    *out = m;
    return PS_OK;
}
void ps_muts_destroy(ps_muts* m) { delete m; }
int64_t ps_muts_count(const ps_muts* m) { return m ? (int64_t)m->v.size() : 0; }
int64_t ps_muts_orig_bytes(const ps_muts* m) { int64_t t = 0; if (m) for (auto& x : m->v) t += x.orig.size(); return t; }
int64_t ps_muts_mut_bytes(const ps_muts* m) { int64_t t = 0; if (m) for (auto& x : m->v) t += x.mut.size(); return t; }
int ps_muts_export(const ps_muts* m, int32_t* start, int64_t* oo, char* op, int64_t* mo, char* mp, double* score) {
    if (!m) return fail(PS_ERR_BAD_ARG, "ps_muts_export");
    int64_t a = 0, b = 0;
    for (size_t i = 0; i < m->v.size(); i++) {
        const Mut& x = m->v[i];
        if (start) start[i] = x.start;
        if (oo) oo[i] = a;
        if (mo) mo[i] = b;
        if (op) std::memcpy(op + a, x.orig.data(), x.orig.size());
        if (mp) std::memcpy(mp + b, x.mut.data(), x.mut.size());
        a += x.orig.size(); b += x.mut.size();
        if (score) score[i] = x.score;
    }
    if (oo) oo[m->v.size()] = a;
    if (mo) mo[m->v.size()] = b;
    return PS_OK;
}

int ps_seqs_create(ps_seqs** out, int64_t n, const int64_t* off, const char* pool) {
    if (!out || n < 0) return PS_ERR_BAD_ARG;
    ps_seqs* s = new ps_seqs();
    for (int64_t i = 0; i < n; i++) s->v.emplace_back(pool + off[i], pool + off[i + 1]);
    *out = s;
    return PS_OK;
}
void ps_seqs_destroy(ps_seqs* s) { delete s; }
int64_t ps_seqs_count(const ps_seqs* s) { return s ? (int64_t)s->v.size() : 0; }
int64_t ps_seqs_bytes(const ps_seqs* s) { int64_t t = 0; if (s) for (auto& x : s->v) t += x.size(); return t; }
int ps_seqs_export(const ps_seqs* s, int64_t* off, char* pool) {
    if (!s) return fail(PS_ERR_BAD_ARG, "ps_seqs_export");
    int64_t a = 0;
    for (size_t i = 0; i < s->v.size(); i++) {
        if (off) off[i] = a;
        if (pool) std::memcpy(pool + a, s->v[i].data(), s->v[i].size());
        a += s->v[i].size();
    }
    if (off) off[s->v.size()] = a;
    return PS_OK;
}

int ps_score_alignments(ps_align* a, double* scores, double* likes) {
    if (!a || !scores) return fail(PS_ERR_BAD_ARG, "ps_score_alignments");
    std::vector<double> s = score_alignments(a->d, likes);
    std::copy(s.begin(), s.end(), scores);
    return PS_OK;
}
int ps_find_point_mutations(ps_align* a, ps_muts** out) {
    if (!a || !out) return fail(PS_ERR_BAD_ARG, "ps_find_point_mutations");
    ps_muts* m = new ps_muts(); m->v = find_point_mutations(a->d); *out = m;
    return PS_OK;
}
int ps_find_mutations(ps_align* a, int32_t n, const int64_t* off, const char* pool, ps_muts** out) {
    if (!a || !out || n < 0 || (n && (!off || !pool))) return fail(PS_ERR_BAD_ARG, "ps_find_mutations");
    std::vector<Seq> seeds;
    for (int i = 0; i < n; i++) seeds.emplace_back(std::string(pool + off[i], pool + off[i + 1]));
    ps_muts* m = new ps_muts(); m->v = find_mutations(a->d, seeds); *out = m;
    return PS_OK;
}
int ps_score_mutations(ps_align* a, const ps_muts* in, ps_muts** out) {
    if (!a || !in || !out) return fail(PS_ERR_BAD_ARG, "ps_score_mutations");
    for (auto& x : in->v) if (x.start < 0) return fail(PS_ERR_BAD_ARG, "negative mutation start");
    ps_muts* m = new ps_muts(); m->v = score_mutations(a->d, in->v); *out = m;
    return PS_OK;
}
int ps_score_mutation_deltas(ps_align* a, const ps_muts* in, double* deltas) {
    if (!a || !in || !deltas) return fail(PS_ERR_BAD_ARG, "ps_score_mutation_deltas");
    for (auto& x : in->v) if (x.start < 0) return fail(PS_ERR_BAD_ARG, "negative mutation start");
    (void)score_mutations(a->d, in->v, deltas);
    return PS_OK;
}
int ps_make_mutations(ps_align* a, const ps_muts* in, int32_t* nb) {
    if (!a || !in || !nb) return fail(PS_ERR_BAD_ARG, "ps_make_mutations");
    *nb = make_mutations(a->d, in->v);
    return PS_OK;
}
int ps_viterbi_mutate(ps_align* a, int32_t nkeep, double skip, double stay, double mmin, double mmax,
                      int32_t, ps_seqs** out) {
    if (!a || !out || a->d.ev.empty()) return fail(PS_ERR_BAD_ARG, "ps_viterbi_mutate");
    ps_seqs* s = new ps_seqs(); s->v = viterbi_mutate(a->d.ev, nkeep, skip, stay, mmin, mmax); *out = s;
    return PS_OK;
}
int ps_swfull(const char* s1, int64_t n1, const char* s2, int64_t n2, int32_t* score, double* acc,
              int32_t* i1, int32_t* i2, int64_t cap, int64_t* np) {
    if (!s1 || !s2 || n1 < 0 || n2 < 0 || !np) return fail(PS_ERR_BAD_ARG, "ps_swfull");
    SW r = swfull(std::string(s1, n1), std::string(s2, n2));
    if ((int64_t)r.a.size() > cap) return fail(PS_ERR_BAD_ARG, "ps_swfull: capacity");
    if (score) *score = r.score;
    if (acc) *acc = r.accuracy;
    for (size_t k = 0; k < r.a.size(); k++) { if (i1) i1[k] = r.a[k]; if (i2) i2[k] = r.b[k]; }
    *np = (int64_t)r.a.size();
    return PS_OK;
}
int ps_seq_to_states(const char* seq, int64_t n, int32_t* st, int64_t* ns) {
    if (!seq || n < 0 || !ns) return fail(PS_ERR_BAD_ARG, "ps_seq_to_states");
    std::vector<int> v = states_of(std::string(seq, n));
    if (st) std::copy(v.begin(), v.end(), st);
    *ns = (int64_t)v.size();
    return PS_OK;
}

int ps_debug_fill(ps_align* a, int32_t e, int32_t dir, double* main, double* stay, uint8_t* sm, uint8_t* ss) {
    if (!a || e < 0 || e >= (int)a->d.ev.size() || !main) return fail(PS_ERR_BAD_ARG, "ps_debug_fill");
    Event& ev = a->d.ev[e];
    Aligner al(a->d.seq, ev, a->d.par);
    al.fill_fwd_all(); al.fill_back_all();
    const int C = (int)a->d.seq.states.size();
    const size_t ld = (size_t)C + 1;
    const size_t tot = ((size_t)ev.n + 1) * ld;
    const double nan = std::nan("");
    for (size_t k = 0; k < tot; k++) { main[k] = nan; if (stay) stay[k] = nan; if (sm) sm[k] = 0; if (ss) ss[k] = 0; }
    std::vector<ColP>& V = dir ? al.B : al.F;
    for (size_t c = 0; c < V.size(); c++) {
        const Col& col = *V[c];
        for (int r = 0; r < col.len; r++) {
            size_t at = (size_t)(col.i0 + r) * ld + c;
            main[at] = col.main[r];
            if (stay) stay[at] = col.stay[r];
            if (sm) sm[at] = col.sm[r];
            if (ss) ss[at] = col.ss[r];
        }
    }
    al.backtrace();
    return PS_OK;
}
int ps_srand(uint32_t seed) { srand(seed); return PS_OK; }
int ps_rand_draw(int64_t n, double* out) { for (int64_t k = 0; k < n; k++) out[k] = rand() / (double(RAND_MAX) + 1); return PS_OK; }
int ps_set_sweep_min(int32_t) { return PS_OK; }
int ps_set_sweep2_min(int32_t) { return PS_OK; }
int ps_set_sparse_min(int32_t) { return PS_OK; }
int ps_set_sweep_form(int32_t, int32_t) { return PS_OK; }
int ps_set_device_fraction(double) { return PS_OK; }
int ps_info(char* out, int64_t cap) { if (!out || cap <= 0) return PS_ERR_BAD_ARG; const char* s = "cpu checker"; size_t n = strlen(s) < (size_t)cap - 1 ? strlen(s) : (size_t)cap - 1; memcpy(out, s, n); out[n] = 0; return PS_OK; }
int ps_prof_enable(int32_t) { return PS_OK; }
int ps_prof_reset(void) { return PS_OK; }
int ps_prof_get(const char*, double* ms, int64_t* n, double* b) { if (ms) *ms = 0; if (n) *n = 0; if (b) *b = 0; return PS_OK; }
int ps_prof_units(const char*, double* u) { if (u) *u = 0; return PS_OK; }

}  // extern "C"

#include "ps_batch_loop.inc"   // lock-step batch entry points: loops over the calls above
