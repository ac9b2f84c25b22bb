/*
 * poreseq_hip.h — C ABI of libporeseq_hip.so, the MI355X (gfx950) implementation of
 * PoreSeq's event-level HMM scoring path.
 *
 * Every entry point replaces one function the reference's Cython binding
 * (poreseq/_poreseqcpp.pyx) calls in the reference C++ core; the reference
 * interface each one stands in for is cited as file:line relative to the
 * reference tree.  Only plain pointers and sizes cross this boundary.
 *
 * All functions return PS_OK (0) or a negative ps_status; ps_last_error()
 * gives the message of the last failure on the calling thread.  The library
 * never falls back to a CPU implementation: without a usable HIP device every
 * compute call fails with PS_ERR_NO_DEVICE.
 *
 * Threading: every host thread that calls into the library gets its own runtime (HIP streams and device
 * buffer pools), so independent ps_align handles may be driven concurrently from different threads — the
 * way to keep an MI355X busy with many independent regions.  One handle must not be shared between
 * threads.  The uniform deviates of ps_viterbi_mutate come from a per-thread generator (see ps_srand).
 *
 * Indices follow the reference: mutation `start` is a 0-based base index,
 * ref_align values are 1-based state indices (0 = unaligned, -1 = inserted
 * level), Smith-Waterman index lists are 1-based with 0 = gap.
 */
#ifndef PORESEQ_HIP_H_
#define PORESEQ_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_N_STATES 1024 /* cpp/AlignUtil.h:19 */

typedef enum ps_status {
    PS_OK = 0,
    PS_ERR_BAD_ARG = -1,     /* NULL pointer, negative size, negative mutation start ... */
    PS_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at init */
    PS_ERR_HIP = -3,         /* a HIP call failed */
    PS_ERR_UNSUPPORTED = -4, /* e.g. a band footprint beyond 2046 rows on one anti-diagonal (realign_width > 1022 in the worst case) */
    PS_ERR_NOMEM = -5
} ps_status;

const char* ps_last_error(void);
/* Name of the backend that serves this ABI ("hip-gfx950"); the test-only
 * oracle and reference shims answer "oracle-cpu" / "reference-cpp". */
const char* ps_backend_name(void);
/* One line about the process-wide state of the library, for logs and bench lines: how its streams get hardware queues (every
 * stream on one priority level with a queue each — GPU_MAX_HW_QUEUES >= 8 was in force when HIP started — or dealt over the
 * device's priority levels), runtimes (host threads inside the library), the per-runtime memory share and the slabs that hold
 * full score matrices.  A host that binds the C ABI directly and wants the first mode starts its process with GPU_MAX_HW_QUEUES=12
 * (or more) in the environment.  Launches nothing (it may start the HIP runtime to ask the device for its memory size). */
int ps_info(char* out, int64_t cap);

/* AlignParams — cpp/AlignUtil.h:57-66 (defaults 4.5 / 150 / 300 / 0). */
typedef struct ps_params {
    double lik_offset;
    int32_t scoring_width;
    int32_t realign_width;
    int32_t verbose;
} ps_params;

/* ---- AlignData (cpp/AlignData.h:26-35): sequence + events + params (+ seed-likelihood cache) ---- */
typedef struct ps_align ps_align;

/* Replaces PythonToAlignData / PythonToEvents (_poreseqcpp.pyx:99-153), i.e.
 * Sequence(seq) (cpp/Sequence.h:31-34), EventData::setData (cpp/EventData.h:208-224),
 * ModelData::setData / setParams (cpp/EventData.h:48-73).
 *   level_off[E+1]      CSR offsets of each event's levels in mean/stdv/ref_align/ref_like
 *   model[E][4][1024]   level_mean, level_stdv, sd_mean, sd_stdv
 *   trans[E][4]         prob_skip, prob_stay, prob_extend, prob_insert
 *   evseq/evseq_off     each event's own 2D base-called sequence (may be NULL)
 * Everything is copied; the caller keeps ownership of its buffers. */
int ps_align_create(ps_align** out, const char* seq, int64_t seq_len, int32_t n_events,
                    const int64_t* level_off, const double* mean, const double* stdv,
                    const double* ref_align, const double* ref_like, const double* model,
                    const double* trans, const char* evseq, const int64_t* evseq_off,
                    const ps_params* params);
void ps_align_destroy(ps_align* a);
/* `data.params.scoring_width = params['point_width']` (_poreseqcpp.pyx:293-294, 361-362, 465-466). */
int ps_align_set_scoring_width(ps_align* a, int32_t width);
/* A driver that keeps one AlignData alive across several PSAlign calls (events resident in device memory instead of
 * re-marshalled per call) announces each new call with this: it resets what PythonToAlignData would have rebuilt —
 * params.scoring_width and the seed-likelihood cache (cpp/AlignData.h:34), which the reference drops between PSAlign
 * calls (_poreseqcpp.pyx:139-153).  Sequence and events' ref_align / ref_like carry over, exactly as the Python
 * attributes do in the reference. */
int ps_align_new_call(ps_align* a, int32_t scoring_width);
int32_t ps_align_n_events(const ps_align* a);
int64_t ps_align_n_levels(const ps_align* a, int32_t ev);
int64_t ps_align_sequence_length(const ps_align* a);
/* data.sequence.bases (_poreseqcpp.pyx:375, 433, 470); `out` needs sequence_length bytes (no NUL). */
int ps_align_get_sequence(const ps_align* a, char* out, int64_t cap);
/* UpdatePythonEvents (_poreseqcpp.pyx:131-137): copy one event's ref_align / ref_like out. */
int ps_align_get_event_refs(const ps_align* a, int32_t ev, double* ref_align, double* ref_like);

/* ---- vector<MutInfo> / vector<MutScore> (cpp/AlignUtil.h:69-91) ---- */
typedef struct ps_muts ps_muts;
/* orig_off / mut_off are CSR offsets [n+1] into the two byte pools; score may be
 * NULL (then every score is the MutScore seed -1e-6, cpp/AlignUtil.h:86). */
int ps_muts_create(ps_muts** out, int64_t n, const int32_t* start, const int64_t* orig_off,
                   const char* orig_pool, const int64_t* mut_off, const char* mut_pool,
                   const double* score);
void ps_muts_destroy(ps_muts* m);
int64_t ps_muts_count(const ps_muts* m);
int64_t ps_muts_orig_bytes(const ps_muts* m);
int64_t ps_muts_mut_bytes(const ps_muts* m);
int ps_muts_export(const ps_muts* m, int32_t* start, int64_t* orig_off, char* orig_pool,
                   int64_t* mut_off, char* mut_pool, double* score);

/* ---- vector<Sequence> results (ViterbiMutate) ---- */
typedef struct ps_seqs ps_seqs;
void ps_seqs_destroy(ps_seqs* s);
int64_t ps_seqs_count(const ps_seqs* s);
int64_t ps_seqs_bytes(const ps_seqs* s);
int ps_seqs_export(const ps_seqs* s, int64_t* off /*[count+1]*/, char* pool);

/* ---- the five free functions of cpp/Mutations.h:18-24 ---- */

/* ScoreAlignments (cpp/MakeMutations.cpp:148-195): forward fill + backtrace per event;
 * scores[E]; likes (NULL or [sequence_length], accumulated into, as the reference does). */
int ps_score_alignments(ps_align* a, double* scores, double* likes);
/* FindPointMutations (cpp/FindMutations.cpp:191-234). */
int ps_find_point_mutations(ps_align* a, ps_muts** out);
/* FindMutations (cpp/FindMutations.cpp:24-186); seed sequences as a CSR string pool. */
int ps_find_mutations(ps_align* a, int32_t n_seqs, const int64_t* seq_off, const char* seq_pool,
                      ps_muts** out);
/* ScoreMutations (cpp/MakeMutations.cpp:23-69): same order as the input list.  Re-aligns
 * every event as a side effect, as the reference does. */
int ps_score_mutations(ps_align* a, const ps_muts* muts, ps_muts** out_scored);
/* The terms of that sum instead of the sum: deltas[e * n_muts + m] = what event e adds to the score of edit m
 * (cpp/MakeMutations.cpp:51: score[m] = -1e-6 + delta[0][m] + delta[1][m] + ... in event order).  For drivers that shard the
 * EVENTS of one region over several GPUs (second axis): every rank scores its events, the deltas are
 * gathered and every rank adds them up in the reference's order (poreseq_amd.dist.score_mutations_event_sharded).
 * Events are re-aligned as by ps_score_mutations.  deltas needs n_events * n_muts doubles. */
int ps_score_mutation_deltas(ps_align* a, const ps_muts* muts, double* deltas);
/* MakeMutations (cpp/MakeMutations.cpp:74-146): greedy application, returns mutated-base count. */
int ps_make_mutations(ps_align* a, const ps_muts* scored, int32_t* n_bases);

/* ViterbiMutate (cpp/Viterbi.h:67-68, cpp/Viterbi.cpp:239-426).  The nkeep > 0 stochastic back-traces draw
 * rand() / (RAND_MAX + 1.0) in the reference's call order (cpp/Viterbi.cpp:108).  The reference never seeds
 * libc rand() and runs one process per region, so every region sees the generator of a fresh process.  The
 * HIP library keeps that contract per host thread: each thread owns a glibc random_r() state (the TYPE_3
 * additive-feedback generator rand() itself uses), initially seeded with 1 like an unseeded process, without
 * the process-wide lock behind rand() that serialises concurrent regions.  ps_srand re-seeds the calling
 * thread's generator (srand(seed) in the oracle / reference builds of this ABI). */
int ps_srand(uint32_t seed);
/* The next n deviates rand() / (RAND_MAX + 1.0) of the calling thread's generator (consumes them). */
int ps_rand_draw(int64_t n, double* out);
int ps_viterbi_mutate(ps_align* a, int32_t nkeep, double skip_prob, double stay_prob,
                      double mut_min, double mut_max, int32_t verbose, ps_seqs** out);

/* ---- lock-step batches over independent AlignData ----------------------------------------------------------------
 * The reference refines one region per process (cmdline.py:182-195, split_fasta.py:50-133); regions are independent.
 * One region keeps a few percent of an MI355X busy, so a driver that has several regions in hand runs the SAME call
 * for all of them through these entry points: every phase (Smith-Waterman batch, banded fills, edit scoring, Viterbi)
 * becomes one launch chain over all regions' events, from one host thread, on the library's own stream.  Results are
 * those of the single-handle calls on each AlignData, bit for bit.  Arrays have n entries; handles must be distinct.
 *
 * ps_rng: the generator ViterbiMutate's stochastic back-traces draw from (rand() of cpp/Viterbi.cpp:108).  The
 * reference's process-per-region model gives every region the stream of a fresh process, continued across the
 * region's Viterbi calls; a lock-step driver therefore keeps one ps_rng per region (seed 1 = unseeded process). */
typedef struct ps_rng ps_rng;
int ps_rng_create(ps_rng** out, uint32_t seed);
void ps_rng_destroy(ps_rng* r);
/* vector<Sequence> from a CSR string pool (seed sequences for ps_batch_find_mutations). */
int ps_seqs_create(ps_seqs** out, int64_t n, const int64_t* off, const char* pool);
/* ScoreAlignments for n AlignData: scores[i] has n_events(i) entries, likes[i] is NULL or [sequence_length(i)]. */
int ps_batch_score_alignments(int32_t n, ps_align* const* a, double* const* scores, double* const* likes);
/* FindMutations: seeds[i] are the candidate sequences of AlignData i; out[i] receives a new ps_muts. */
int ps_batch_find_mutations(int32_t n, ps_align* const* a, const ps_seqs* const* seeds, ps_muts** out);
/* ScoreMutations: out[i] receives a new ps_muts with the scores of muts[i], same order. */
int ps_batch_score_mutations(int32_t n, ps_align* const* a, const ps_muts* const* muts, ps_muts** out);
/* MakeMutations: greedy application per AlignData; the re-scoring rounds of the recursion are batched. */
int ps_batch_make_mutations(int32_t n, ps_align* const* a, const ps_muts* const* scored, int32_t* n_bases);
/* ViterbiMutate: rng[i] may be NULL (the calling thread's generator, as ps_viterbi_mutate). */
int ps_batch_viterbi_mutate(int32_t n, ps_align* const* a, ps_rng* const* rng, int32_t nkeep, double skip_prob,
                            double stay_prob, double mut_min, double mut_max, ps_seqs** out);

/* swfull (cpp/swlib.h:36, cpp/swlib.cpp:211-340).  inds1/inds2 need room for n1+n2 entries. */
int ps_swfull(const char* seq1, int64_t n1, const char* seq2, int64_t n2, int32_t* score,
              double* accuracy, int32_t* inds1, int32_t* inds2, int64_t cap, int64_t* n_pairs);
/* Sequence::populateStates (cpp/Sequence.h:64-100); states needs max(n-4,0) entries. */
int ps_seq_to_states(const char* seq, int64_t n, int32_t* states, int64_t* n_states);

/* ---- kernel-level test hooks (K1/K2 matrices vs oracle dumps) ----
 * Runs Alignment::update (cpp/Alignment.cpp:63-73) for one event on the current sequence
 * and returns the dense (n_levels+1) x (n_states+1) forward or backward main matrix with
 * out-of-band cells as NaN; direction 0 = forward (column index = ref position), 1 = backward
 * (column index k = -col, cpp/Alignment.cpp:284-285).  stay (may be NULL) gets the stay matrix.
 * Step codes are returned for the forward matrix only — the only ones the reference ever reads
 * (backtrace, cpp/Alignment.cpp:516-624); for direction 1 they are zero. */
int ps_debug_fill(ps_align* a, int32_t ev, int32_t direction, double* main, double* stay,
                  uint8_t* step_main, uint8_t* step_stay);

/* Tuning knob (process-wide): forward-only alignment batches — ScoreAlignments, FindMutations' candidate sequences — of at
 * least `min_alignments` jobs run as strip sweeps (ps_sweep.hip / ps_sweepw.hip: one to four wavefronts per alignment, ps_set_sweep_form);
 * smaller ones a workgroup per alignment (k_fill),
 * which finishes a lone small batch sooner.  Results do not depend on it.  Negative: back to the default
 * (PORESEQ_SWEEP_MIN, else 400). */
int ps_set_sweep_min(int32_t min_alignments);
/* The same for Alignment::update batches (ScoreMutations: a forward and a backward sweep per alignment, with score matrices):
 * from `min_sweeps` sweeps on, strip sweeps with full records.  Negative: the default (PORESEQ_SWEEP2_MIN, else never:
 * their records cap a launch at a few hundred sweeps, where a workgroup per sweep is twice as fast — DESIGN.md section 4). */
int ps_set_sweep2_min(int32_t min_sweeps);
/* Alignment::update batches whose edit lists read at most a quarter of the matrix columns (every ScoreMutations call of a consensus
 * schedule except Refine's point edits at every position): from `min_sweeps` sweeps on, strip sweeps that store the
 * {main, stay} records of the columns scoreMutation / columnMax will read (cpp/Alignment.cpp:447-512, cpp/Alignment.h:181-214) and
 * nothing else of the score matrices.  Negative: the default (PORESEQ_SPARSE_MIN, else 160).  Results do not depend on it. */
int ps_set_sparse_min(int32_t min_sweeps);
/* The form every strip sweep tries first: `rows_per_lane` rows of the band per lane on `wavefronts` (1, 2 or 4) wavefronts per
 * (alignment, direction) — ps_sweep.hip / ps_sweepw.hip; a band too wide for it takes the next larger form.  rows_per_lane <= 0 with
 * wavefronts 1 / 2 / 4: that many wavefronts, the smallest strip height whose band fits; both <= 0: the library's own choice (by
 * launch size; PORESEQ_SWEEP_FORM=K,NW).  Returns PS_ERR_BAD_ARG for a form that is not built.
 * Results do not depend on it (the reference has one serial loop per alignment, cpp/Alignment.cpp:83-99). */
int ps_set_sweep_form(int32_t rows_per_lane, int32_t wavefronts);
/* The part of the device's memory THIS PROCESS plans for (process-wide; default 1, or PORESEQ_DEVICE_FRACTION): the slabs for full
 * score matrices, every runtime's share and the ceiling of the device pools are fractions of it.  One process per GPU leaves it
 * alone; ranks that share a GPU set 1 / (ranks on the device) before their first compute call (poreseq_amd.dist.init does).
 * fraction <= 0 restores the default; > 1 is PS_ERR_BAD_ARG.  Not part of the reference's interface (its processes share nothing). */
int ps_set_device_fraction(double fraction);

/* Hot-kernel instrumentation for bench.py: accumulated HIP-event time (ms), launches and
 * algorithmic bytes of the named kernel class ("fill" = k_fill, "sweep" = the strip sweeps k_sweep / k_sweeps / k_sweep2 and their _w builds,
 * "score", "viterbi", "sw") since reset; host-side launch counts by form under "sweep_w2", "sweep_w4", "sweep_kept", "sw_pk8", "slab". */
/* ps_prof_enable(1) makes every hot-kernel launch be bracketed by HIP events on the library's stream
 * (one extra synchronisation per launch: use it in a separate, untimed pass); ps_prof_enable(2) queues the
 * event pairs instead and reads them when the profile is asked for (no synchronisation per launch: usable
 * inside a timed region).  The profile belongs to the calling thread's runtime.  ps_prof_reset zeroes the sums. */
int ps_prof_enable(int32_t on);
int ps_prof_reset(void);
int ps_prof_get(const char* name, double* ms, int64_t* launches, double* alg_bytes);
/* work units of the class since reset: "fill" / "sweep" = sweeps (one alignment, one direction), "score" = (event, edit) items */
int ps_prof_units(const char* name, double* units);

#ifdef __cplusplus
}
#endif
#endif /* PORESEQ_HIP_H_ */
