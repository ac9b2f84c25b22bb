// ref_shim.cpp — extern "C" adapter that exposes the REAL reference C++ core
// (/root/reference/cpp, compiled where it lies by oracle/Makefile into oracle/_ref/)
// through the ABI of include/poreseq_hip.h.
//
// *** TEST INFRASTRUCTURE ONLY *** — used to pin oracle/ps_oracle.cpp and as the
// "reference" CPU baseline in bench.py.  Contains no reference source text: it only
// calls the reference's public functions (cpp/Mutations.h:18-24, cpp/Viterbi.h:67-68,
// cpp/swlib.h:36-40) the same way poreseq/_poreseqcpp.pyx does.
#include "../include/poreseq_hip.h"

#include <algorithm>
#include <cmath>
#include <iostream>
#include <memory>
#include <stdint.h>
#include <string>
#include <vector>
// ps_debug_fill needs to read Alignment::scores / scores_back, which sit in the class's
// default-private section: open the class up for this translation unit only.
#define class struct
#define private public
#include "Alignment.h"
#undef private
#undef class
#include "Mutations.h"
#include "Viterbi.h"
#include "swlib.h"

#include <cstring>
#include <memory>
#include <string>
#include <vector>

struct ps_align { AlignData d; };
struct ps_muts { std::vector<MutScore> v; };
struct ps_seqs { std::vector<std::string> v; };

static thread_local std::string g_err;
static int fail(int c, const char* m) { g_err = m; return c; }

extern "C" {

const char* ps_last_error(void) { return g_err.c_str(); }
const char* ps_backend_name(void) { return "reference-cpp"; }

// mirrors PythonToAlignData / PythonToEvents (poreseq/_poreseqcpp.pyx:99-153)
int ps_align_create(ps_align** out, const char* seq, int64_t seq_len, int32_t n_events,
                    const int64_t* level_off, const double* mean, const double* stdv,
                    const double* ref_align, const double* ref_like, const double* model,
                    const double* trans, const char* evseq, const int64_t* evseq_off,
                    const ps_params* params) {
    if (!out || !seq) return fail(PS_ERR_BAD_ARG, "ps_align_create");
    std::unique_ptr<ps_align> a(new ps_align());
    a->d.sequence = Sequence(std::string(seq, (size_t)seq_len));
    if (params) {
        a->d.params.verbose = params->verbose;
        a->d.params.lik_offset = params->lik_offset;
        a->d.params.realign_width = params->realign_width;
        a->d.params.scoring_width = params->scoring_width;
    }
    for (int e = 0; e < n_events; e++) {
        EventData ev;
        const int64_t o = level_off[e];
        const int n = (int)(level_off[e + 1] - o);
        ev.setData(n, const_cast<double*>(mean + o), const_cast<double*>(stdv + o),
                   const_cast<double*>(ref_align + o), const_cast<double*>(ref_like + o));
        double* md = const_cast<double*>(model + (size_t)e * 4 * N_STATES);
        ev.model.setData(md, md + N_STATES, md + 2 * N_STATES, md + 3 * N_STATES, false);
        ev.model.setParams(trans[e * 4], trans[e * 4 + 1], trans[e * 4 + 2], trans[e * 4 + 3]);
        if (evseq && evseq_off) ev.sequence = Sequence(std::string(evseq + evseq_off[e], evseq + evseq_off[e + 1]));
        a->d.events.push_back(ev);
    }
    *out = a.release();
    return PS_OK;
}
void ps_align_destroy(ps_align* a) { delete a; }
int ps_align_set_scoring_width(ps_align* a, int32_t w) { a->d.params.scoring_width = w; return PS_OK; }
int ps_align_new_call(ps_align* a, int32_t w) { a->d.params.scoring_width = w; a->d.seqlikes.clear(); return PS_OK; }
int32_t ps_align_n_events(const ps_align* a) { return (int32_t)a->d.events.size(); }
int64_t ps_align_n_levels(const ps_align* a, int32_t e) { return a->d.events[e].length; }
int64_t ps_align_sequence_length(const ps_align* a) { return (int64_t)a->d.sequence.bases.size(); }
int ps_align_get_sequence(const ps_align* a, char* out, int64_t cap) {
    if ((int64_t)a->d.sequence.bases.size() > cap) return fail(PS_ERR_BAD_ARG, "cap");
    std::memcpy(out, a->d.sequence.bases.data(), a->d.sequence.bases.size());
    return PS_OK;
}
int ps_align_get_event_refs(const ps_align* a, int32_t e, double* ra, double* rl) {
    const EventData& ev = a->d.events[e];
    if (ra) std::copy(ev.ref_align.begin(), ev.ref_align.end(), ra);
    if (rl) std::copy(ev.ref_like.begin(), ev.ref_like.end(), rl);
    return PS_OK;
}

int ps_muts_create(ps_muts** out, int64_t n, const int32_t* start, const int64_t* oo, const char* op,
                   const int64_t* mo, const char* mp, const double* score) {
    ps_muts* m = new ps_muts();
    m->v.resize(n);
    for (int64_t i = 0; i < n; i++) {
        m->v[i].start = start[i];
        if (oo[i + 1] > oo[i]) m->v[i].orig.assign(op + oo[i], op + oo[i + 1]);
        if (mo[i + 1] > mo[i]) m->v[i].mut.assign(mp + mo[i], mp + mo[i + 1]);
        if (score) m->v[i].score = score[i];
    }
    *out = m;
    return PS_OK;
}
void ps_muts_destroy(ps_muts* m) { delete m; }
int64_t ps_muts_count(const ps_muts* m) { return (int64_t)m->v.size(); }
int64_t ps_muts_orig_bytes(const ps_muts* m) { int64_t t = 0; for (auto& x : m->v) t += x.orig.size(); return t; }
int64_t ps_muts_mut_bytes(const ps_muts* m) { int64_t t = 0; for (auto& x : m->v) t += x.mut.size(); return t; }
int ps_muts_export(const ps_muts* m, int32_t* start, int64_t* oo, char* op, int64_t* mo, char* mp, double* score) {
    int64_t a = 0, b = 0;
    for (size_t i = 0; i < m->v.size(); i++) {
        const MutScore& x = m->v[i];
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
int64_t ps_seqs_count(const ps_seqs* s) { return (int64_t)s->v.size(); }
int64_t ps_seqs_bytes(const ps_seqs* s) { int64_t t = 0; for (auto& x : s->v) t += x.size(); return t; }
int ps_seqs_export(const ps_seqs* s, int64_t* off, char* pool) {
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
    std::vector<double> s = ScoreAlignments(a->d, likes);
    std::copy(s.begin(), s.end(), scores);
    return PS_OK;
}
int ps_find_point_mutations(ps_align* a, ps_muts** out) {
    std::vector<MutInfo> mi = FindPointMutations(a->d);
    ps_muts* m = new ps_muts(); m->v.assign(mi.begin(), mi.end()); *out = m;
    return PS_OK;
}
int ps_find_mutations(ps_align* a, int32_t n, const int64_t* off, const char* pool, ps_muts** out) {
    std::vector<Sequence> seeds;
    for (int i = 0; i < n; i++) seeds.push_back(Sequence(std::string(pool + off[i], pool + off[i + 1])));
    std::vector<MutInfo> mi = FindMutations(a->d, seeds);
    ps_muts* m = new ps_muts(); m->v.assign(mi.begin(), mi.end()); *out = m;
    return PS_OK;
}
int ps_score_mutations(ps_align* a, const ps_muts* in, ps_muts** out) {
    std::vector<MutInfo> mi(in->v.begin(), in->v.end());
    ps_muts* m = new ps_muts(); m->v = ScoreMutations(a->d, mi); *out = m;
    return PS_OK;
}
// (the reference only returns the sums: not available from its public functions)
int ps_score_mutation_deltas(ps_align*, const ps_muts*, double*) { return PS_ERR_UNSUPPORTED; }
int ps_make_mutations(ps_align* a, const ps_muts* in, int32_t* nb) {
    *nb = MakeMutations(a->d, in->v);
    return PS_OK;
}
int ps_viterbi_mutate(ps_align* a, int32_t nkeep, double skip, double stay, double mmin, double mmax,
                      int32_t verbose, ps_seqs** out) {
    std::vector<Sequence> sq = ViterbiMutate(a->d.events, nkeep, skip, stay, mmin, mmax, verbose != 0);
    ps_seqs* s = new ps_seqs();
    for (auto& x : sq) s->v.push_back(x.bases);
    *out = s;
    return PS_OK;
}
int ps_swfull(const char* s1, int64_t n1, const char* s2, int64_t n2, int32_t* score, double* acc,
              int32_t* i1, int32_t* i2, int64_t cap, int64_t* np) {
    SWAlignment r = swfull(std::string(s1, n1), std::string(s2, n2));
    if ((int64_t)r.inds1.size() > cap) return fail(PS_ERR_BAD_ARG, "cap");
    if (score) *score = r.score;
    if (acc) *acc = r.accuracy;
    for (size_t k = 0; k < r.inds1.size(); k++) { if (i1) i1[k] = r.inds1[k]; if (i2) i2[k] = r.inds2[k]; }
    *np = (int64_t)r.inds1.size();
    return PS_OK;
}
int ps_seq_to_states(const char* seq, int64_t n, int32_t* st, int64_t* ns) {
    Sequence s(std::string(seq, n));
    if (st) std::copy(s.states.begin(), s.states.end(), st);
    *ns = (int64_t)s.states.size();
    return PS_OK;
}

int ps_debug_fill(ps_align* a, int32_t e, int32_t dir, double* main, double* stay, uint8_t* sm, uint8_t* ss) {
    EventData& ev = a->d.events[e];
    Alignment al(a->d.sequence, ev, a->d.params);
    al.fillColumns();
    al.fillColumnsBack();
    const size_t ld = a->d.sequence.states.size() + 1;
    const size_t tot = ((size_t)ev.length + 1) * ld;
    const double nan = std::nan("");
    for (size_t k = 0; k < tot; k++) { main[k] = nan; if (stay) stay[k] = nan; if (sm) sm[k] = 0; if (ss) ss[k] = 0; }
    std::vector<AlignPointer>& V = dir ? al.scores_back : al.scores;
    for (size_t c = 0; c < V.size(); c++) {
        AlignColumn& col = *V[c];
        for (int i = col.i0; i < col.i0 + col.length; i++) {
            size_t at = (size_t)i * ld + c;
            main[at] = *col.getPointer(i, 0);
            if (stay) stay[at] = *col.getPointer(i, 1);
            if (sm) sm[at] = *col.getStep(i, 0);
            if (ss) ss[at] = *col.getStep(i, 1);
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
