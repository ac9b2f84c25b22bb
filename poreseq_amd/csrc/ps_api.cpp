// ps_api.cpp — the C ABI of include/poreseq_hip.h over the HIP implementation.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include "ps_host.h"

using namespace ps;

struct ps_align { Align a; };
struct ps_muts { std::vector<Mut> v; };
struct ps_seqs { std::vector<std::string> v; };
struct ps_rng { RandState* s = nullptr; ~ps_rng() { if (s) rand_state_free(s); } };

#define NEED_RT()            \
    Runtime* rt = nullptr;   \
    PS_TRY(runtime(&rt))

extern "C" {

const char* ps_last_error(void) { return last_error(); }
const char* ps_backend_name(void) { return "hip-gfx950"; }
int ps_info(char* out, int64_t cap) {
    if (!out || cap <= 0) return fail(PS_ERR_BAD_ARG, "ps_info");
    const std::string s = info_string();
    const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
    memcpy(out, s.data(), n);
    out[n] = 0;
    return PS_OK;
}

int ps_align_create(ps_align** out, const char* seq, int64_t seq_len, int32_t n_events,
                    const int64_t* level_off, const double* mean, const double* stdv,
                    const double* ref_align, const double* ref_like, const double* model,
                    const double* trans, const char* evseq, const int64_t* evseq_off,
                    const ps_params* params) {
    if (!out || !seq || seq_len < 0 || n_events < 0 ||
        (n_events && (!level_off || !mean || !stdv || !ref_align || !ref_like || !model || !trans)))
        return fail(PS_ERR_BAD_ARG, "ps_align_create: bad argument");
    NEED_RT();
    std::unique_ptr<ps_align> h(new ps_align());
    PS_TRY(h->a.create(rt, seq, seq_len, n_events, level_off, mean, stdv, ref_align, ref_like, model, trans,
                       evseq, evseq_off, params));
    *out = h.release();
    return PS_OK;
}
void ps_align_destroy(ps_align* a) { delete a; }
int ps_align_set_scoring_width(ps_align* a, int32_t w) {
    if (!a) return fail(PS_ERR_BAD_ARG, "null handle");
    a->a.par.scoring_width = w;
    return PS_OK;
}
int ps_align_new_call(ps_align* a, int32_t w) {
    if (!a) return fail(PS_ERR_BAD_ARG, "null handle");
    a->a.par.scoring_width = w;
    a->a.seqlikes.clear();
    return PS_OK;
}
int32_t ps_align_n_events(const ps_align* a) { return a ? a->a.E : 0; }
int64_t ps_align_n_levels(const ps_align* a, int32_t e) { return (a && e >= 0 && e < a->a.E) ? a->a.n[e] : -1; }
int64_t ps_align_sequence_length(const ps_align* a) { return a ? (int64_t)a->a.bases.size() : -1; }
int ps_align_get_sequence(const ps_align* a, char* out, int64_t cap) {
    if (!a || !out || cap < (int64_t)a->a.bases.size()) return fail(PS_ERR_BAD_ARG, "ps_align_get_sequence");
    memcpy(out, a->a.bases.data(), a->a.bases.size());
    return PS_OK;
}
int ps_align_get_event_refs(const ps_align* ca, int32_t e, double* ra, double* rl) {
    ps_align* a = const_cast<ps_align*>(ca);
    if (!a || e < 0 || e >= a->a.E) return fail(PS_ERR_BAD_ARG, "ps_align_get_event_refs");
    NEED_RT();
    PS_TRY(a->a.refs_to_host(rt));
    const int64_t o = a->a.off[e];
    if (ra) memcpy(ra, a->a.h_ra.data() + o, a->a.n[e] * sizeof(double));
    if (rl) memcpy(rl, a->a.h_rl.data() + o, a->a.n[e] * sizeof(double));
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
        if (op) memcpy(op + a, x.orig.data(), x.orig.size());
        if (mp) memcpy(mp + b, x.mut.data(), x.mut.size());
        a += x.orig.size(); b += x.mut.size();
        if (score) score[i] = x.score;
    }
    if (oo) oo[m->v.size()] = a;
    if (mo) mo[m->v.size()] = b;
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
        if (pool) memcpy(pool + a, s->v[i].data(), s->v[i].size());
        a += s->v[i].size();
    }
    if (off) off[s->v.size()] = a;
    return PS_OK;
}

int ps_score_alignments(ps_align* a, double* scores, double* likes) {
    if (!a || (!scores && a->a.E)) return fail(PS_ERR_BAD_ARG, "ps_score_alignments");
    NEED_RT();
    return score_alignments(rt, &a->a, scores, likes);
}
int ps_find_point_mutations(ps_align* a, ps_muts** out) {
    if (!a || !out) return fail(PS_ERR_BAD_ARG, "ps_find_point_mutations");
    ps_muts* m = new ps_muts();
    find_point_mutations(&a->a, &m->v);
    *out = m;
    return PS_OK;
}
int ps_find_mutations(ps_align* a, int32_t n, const int64_t* off, const char* pool, ps_muts** out) {
    if (!a || !out || n < 0 || (n && (!off || !pool))) return fail(PS_ERR_BAD_ARG, "ps_find_mutations");
    NEED_RT();
    std::vector<std::string> seeds;
    for (int i = 0; i < n; i++) seeds.emplace_back(pool + off[i], pool + off[i + 1]);
    std::unique_ptr<ps_muts> m(new ps_muts());
    PS_TRY(find_mutations(rt, &a->a, seeds, &m->v));
    *out = m.release();
    return PS_OK;
}
int ps_score_mutations(ps_align* a, const ps_muts* in, ps_muts** out) {
    if (!a || !in || !out) return fail(PS_ERR_BAD_ARG, "ps_score_mutations");
    NEED_RT();
    std::unique_ptr<ps_muts> m(new ps_muts());
    PS_TRY(score_mutations(rt, &a->a, in->v, &m->v));
    *out = m.release();
    return PS_OK;
}
int ps_score_mutation_deltas(ps_align* a, const ps_muts* in, double* deltas) {
    if (!a || !in || !deltas) return fail(PS_ERR_BAD_ARG, "ps_score_mutation_deltas");
    NEED_RT();
    std::vector<Mut> scored;
    std::vector<double*> dout(1, deltas);
    return score_mutations_multi(rt, {&a->a}, {&in->v}, {&scored}, &dout);
}
int ps_make_mutations(ps_align* a, const ps_muts* in, int32_t* nb) {
    if (!a || !in || !nb) return fail(PS_ERR_BAD_ARG, "ps_make_mutations");
    NEED_RT();
    int n = 0;
    PS_TRY(make_mutations(rt, &a->a, in->v, &n));
    *nb = n;
    return PS_OK;
}
int ps_viterbi_mutate(ps_align* a, int32_t nkeep, double skip, double stay, double mmin, double mmax,
                      int32_t verbose, ps_seqs** out) {
    if (!a || !out || a->a.E == 0 || nkeep < 0) return fail(PS_ERR_BAD_ARG, "ps_viterbi_mutate");
    NEED_RT();
    std::unique_ptr<ps_seqs> s(new ps_seqs());
    PS_TRY(viterbi_mutate(rt, &a->a, nkeep, skip, stay, mmin, mmax, &s->v, verbose != 0));
    *out = s.release();
    return PS_OK;
}
// ---- lock-step batches (include/poreseq_hip.h) ----
int ps_rng_create(ps_rng** out, uint32_t seed) {
    if (!out) return fail(PS_ERR_BAD_ARG, "ps_rng_create");
    ps_rng* r = new ps_rng();
    r->s = rand_state_new(seed);
    *out = r;
    return PS_OK;
}
void ps_rng_destroy(ps_rng* r) { delete r; }
int ps_seqs_create(ps_seqs** out, int64_t n, const int64_t* off, const char* pool) {
    if (!out || n < 0 || (n && (!off || !pool))) return fail(PS_ERR_BAD_ARG, "ps_seqs_create");
    ps_seqs* s = new ps_seqs();
    for (int64_t i = 0; i < n; i++) s->v.emplace_back(pool + off[i], pool + off[i + 1]);
    *out = s;
    return PS_OK;
}
static int batch_handles(int32_t n, ps_align* const* a, std::vector<Align*>* as) {
    if (n < 0 || (n && !a)) return fail(PS_ERR_BAD_ARG, "batch: bad handle array");
    for (int i = 0; i < n; i++) {
        if (!a[i]) return fail(PS_ERR_BAD_ARG, "batch: null handle");
        for (int j = 0; j < i; j++) if (a[j] == a[i]) return fail(PS_ERR_BAD_ARG, "batch: the same handle twice");
        as->push_back(&a[i]->a);
    }
    return PS_OK;
}
int ps_batch_score_alignments(int32_t n, ps_align* const* a, double* const* scores, double* const* likes) {
    std::vector<Align*> as;
    PS_TRY(batch_handles(n, a, &as));
    if (n && !scores) return fail(PS_ERR_BAD_ARG, "ps_batch_score_alignments");
    NEED_RT();
    std::vector<double*> sc(n), lk(n);
    for (int i = 0; i < n; i++) {
        if (!scores[i] && as[i]->E) return fail(PS_ERR_BAD_ARG, "ps_batch_score_alignments: null scores");
        sc[i] = scores[i]; lk[i] = likes ? likes[i] : nullptr;
    }
    return score_alignments_multi(rt, as, sc, lk);
}
int ps_batch_find_mutations(int32_t n, ps_align* const* a, const ps_seqs* const* seeds, ps_muts** out) {
    std::vector<Align*> as;
    PS_TRY(batch_handles(n, a, &as));
    if (n && (!seeds || !out)) return fail(PS_ERR_BAD_ARG, "ps_batch_find_mutations");
    NEED_RT();
    std::vector<std::unique_ptr<ps_muts>> m(n);
    std::vector<const std::vector<std::string>*> sd(n);
    std::vector<std::vector<Mut>*> o(n);
    for (int i = 0; i < n; i++) {
        if (!seeds[i]) return fail(PS_ERR_BAD_ARG, "ps_batch_find_mutations: null seeds");
        m[i].reset(new ps_muts()); sd[i] = &seeds[i]->v; o[i] = &m[i]->v;
    }
    PS_TRY(find_mutations_multi(rt, as, sd, o));
    for (int i = 0; i < n; i++) out[i] = m[i].release();
    return PS_OK;
}
int ps_batch_score_mutations(int32_t n, ps_align* const* a, const ps_muts* const* muts, ps_muts** out) {
    std::vector<Align*> as;
    PS_TRY(batch_handles(n, a, &as));
    if (n && (!muts || !out)) return fail(PS_ERR_BAD_ARG, "ps_batch_score_mutations");
    NEED_RT();
    std::vector<std::unique_ptr<ps_muts>> m(n);
    std::vector<const std::vector<Mut>*> in(n);
    std::vector<std::vector<Mut>*> o(n);
    for (int i = 0; i < n; i++) {
        if (!muts[i]) return fail(PS_ERR_BAD_ARG, "ps_batch_score_mutations: null list");
        m[i].reset(new ps_muts()); in[i] = &muts[i]->v; o[i] = &m[i]->v;
    }
    PS_TRY(score_mutations_multi(rt, as, in, o));
    for (int i = 0; i < n; i++) out[i] = m[i].release();
    return PS_OK;
}
int ps_batch_make_mutations(int32_t n, ps_align* const* a, const ps_muts* const* scored, int32_t* n_bases) {
    std::vector<Align*> as;
    PS_TRY(batch_handles(n, a, &as));
    if (n && (!scored || !n_bases)) return fail(PS_ERR_BAD_ARG, "ps_batch_make_mutations");
    NEED_RT();
    std::vector<std::vector<Mut>> in(n);
    for (int i = 0; i < n; i++) if (!scored[i]) return fail(PS_ERR_BAD_ARG, "ps_batch_make_mutations: null list");
    par_for(n, [&](int i) { in[i] = scored[i]->v; });   // (a Refine list is 80 000 edits per region: the copies run side by side)
    std::vector<int> nb;
    PS_TRY(make_mutations_multi(rt, as, std::move(in), &nb));
    for (int i = 0; i < n; i++) n_bases[i] = nb[i];
    return PS_OK;
}
int ps_batch_viterbi_mutate(int32_t n, ps_align* const* a, ps_rng* const* rng, int32_t nkeep, double skip, double stay,
                            double mmin, double mmax, ps_seqs** out) {
    std::vector<Align*> as;
    PS_TRY(batch_handles(n, a, &as));
    if (n && !out) return fail(PS_ERR_BAD_ARG, "ps_batch_viterbi_mutate");
    if (nkeep < 0) return fail(PS_ERR_BAD_ARG, "ps_batch_viterbi_mutate: nkeep");
    for (Align* x : as) if (x->E == 0) return fail(PS_ERR_BAD_ARG, "ps_batch_viterbi_mutate: no events");
    NEED_RT();
    std::vector<std::unique_ptr<ps_seqs>> s(n);
    std::vector<RandState*> rs(n, nullptr);
    std::vector<std::vector<std::string>*> o(n);
    for (int i = 0; i < n; i++) { s[i].reset(new ps_seqs()); o[i] = &s[i]->v; rs[i] = (rng && rng[i]) ? rng[i]->s : nullptr; }
    PS_TRY(viterbi_mutate_multi(rt, as, rs, nkeep, skip, stay, mmin, mmax, o));
    for (int i = 0; i < n; i++) out[i] = s[i].release();
    return PS_OK;
}

int ps_swfull(const char* s1, int64_t n1, const char* s2, int64_t n2, int32_t* score, double* acc,
              int32_t* i1, int32_t* i2, int64_t cap, int64_t* np) {
    if (!s1 || !s2 || n1 < 0 || n2 < 0 || !np) return fail(PS_ERR_BAD_ARG, "ps_swfull");
    NEED_RT();
    int sc = 0; double ac = 0;
    std::vector<int> a, b;
    PS_TRY(sw_device(rt, std::string(s1, n1), std::string(s2, n2), &sc, &ac, &a, &b));
    if ((int64_t)a.size() > cap) return fail(PS_ERR_BAD_ARG, "ps_swfull: index capacity too small");
    if (score) *score = sc;
    if (acc) *acc = ac;
    for (size_t k = 0; k < a.size(); k++) { if (i1) i1[k] = a[k]; if (i2) i2[k] = b[k]; }
    *np = (int64_t)a.size();
    return PS_OK;
}
int ps_seq_to_states(const char* seq, int64_t n, int32_t* st, int64_t* ns) {
    if (!seq || n < 0 || !ns) return fail(PS_ERR_BAD_ARG, "ps_seq_to_states");
    std::vector<int> v = states_of(std::string(seq, n));  // pure index arithmetic, no DP: host
    if (st) std::copy(v.begin(), v.end(), st);
    *ns = (int64_t)v.size();
    return PS_OK;
}

int ps_debug_fill(ps_align* a, int32_t e, int32_t dir, double* main, double* stay, uint8_t* sm, uint8_t* ss) {
    if (!a || e < 0 || e >= a->a.E || !main || dir < 0 || dir > 1) return fail(PS_ERR_BAD_ARG, "ps_debug_fill");
    NEED_RT();
    return debug_fill(rt, &a->a, e, dir, main, stay, sm, ss);
}

int ps_srand(uint32_t seed) { rand_seed(seed); return PS_OK; }
int ps_rand_draw(int64_t n, double* out) {
    if (n < 0 || (n && !out)) return fail(PS_ERR_BAD_ARG, "ps_rand_draw");
    for (int64_t k = 0; k < n; k++) out[k] = rand_next() / (double(RAND_MAX) + 1);
    return PS_OK;
}

int ps_set_sweep_min(int32_t n) { sweep_min_set(n); return PS_OK; }
int ps_set_sweep2_min(int32_t n) { sweep2_min_set(n); return PS_OK; }
int ps_set_sparse_min(int32_t n) { sparse_min_set(n); return PS_OK; }
int ps_set_device_fraction(double f) {
    if (f > 1.0) return fail(PS_ERR_BAD_ARG, "ps_set_device_fraction: a fraction in (0, 1]; <= 0 restores the default");
    device_fraction_set(f);
    return PS_OK;
}
int ps_set_sweep_form(int32_t K, int32_t NW) {
    if (K <= 0 && NW > 0 && NW != 1 && NW != 2 && NW != 4) return fail(PS_ERR_BAD_ARG, "ps_set_sweep_form: 1, 2 or 4 wavefronts per sweep");
    if (K > 0 && !sweep_form_exists(K, NW)) return fail(PS_ERR_BAD_ARG, "ps_set_sweep_form: no such form (rows per lane, wavefronts)");
    sweep_form_set(K, NW);
    return PS_OK;
}

int ps_prof_enable(int32_t on) {
    NEED_RT();
    prof_flush(rt);
    rt->prof_on = on != 0;
    rt->prof_defer = on == 2;
    return PS_OK;
}
int ps_prof_reset(void) {
    NEED_RT();
    prof_flush(rt);
    rt->prof.clear();
    return PS_OK;
}
int ps_prof_get(const char* name, double* ms, int64_t* n, double* bytes) {
    NEED_RT();
    prof_flush(rt);
    Prof p;
    if (name) { auto it = rt->prof.find(name); if (it != rt->prof.end()) p = it->second; }
    if (ms) *ms = p.ms;
    if (n) *n = p.launches;
    if (bytes) *bytes = p.bytes;
    return PS_OK;
}

int ps_prof_units(const char* name, double* units) {
    NEED_RT();
    prof_flush(rt);
    Prof p;
    if (name) { auto it = rt->prof.find(name); if (it != rt->prof.end()) p = it->second; }
    if (units) *units = p.units;
    return PS_OK;
}

}  // extern "C"
