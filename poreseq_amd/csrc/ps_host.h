// ps_host.h — host-side structures of libporeseq_hip.so
#ifndef PS_HOST_H_
#define PS_HOST_H_

#include <functional>

#include "ps_internal.h"

namespace ps {

const char* last_error();

// PORESEQ_TRACE=1: wall-clock phase timers on stderr (host-side tuning aid)
struct Tick {
    const char* what; double t0; bool on;
    explicit Tick(const char* w);
    void lap(const char* label);
};

struct Align;

void rand_seed(unsigned seed);   // per-thread generator of ViterbiMutate's deviates (ps_find.cpp)
int rand_next();
struct RandState;                // an explicit generator state (one per region of a lock-step driver)
RandState* rand_state_new(unsigned seed);
void rand_state_free(RandState* r);
void rand_state_seed(RandState* r, unsigned seed);
int rand_state_next(RandState* r);   // nullptr: the calling thread's generator

struct JobSpec {
    Align* a = nullptr;                                  // owner of the event data (one batch may mix several AlignData)
    int ev = 0;
    const std::vector<int>* states = nullptr;
    double *ra = nullptr, *rl = nullptr, *ri = nullptr;  // device arrays of the job
    JobOut* out = nullptr;                               // device result record of the job
    // column-sparse Alignment::update (ScoreMutations with a short edit list): per direction the device table of kept-column
    // indices (C + 2 entries, -1: not kept) and the number of kept columns
    const int* keep[2] = {nullptr, nullptr};
    int nkeep[2] = {0, 0};
};

// a set of alignment jobs with all their device workspaces carved out of the runtime pool
struct Batch {
    std::vector<JobD> jobs;
    BatchD d;
    int ndir = 1, P = 0, Pmax = 64, maxC = 0, maxn = 0, maxlbn = 0;
    int64_t maxS = 0, cells = 0, ncols = 0;
    // strip sweeps (forward-only batches, ps_sweep.hip)
    std::vector<SweepJob> sjobs;
    SweepD sd;
    int sweep_K = 0, sweep_NW = 1, sweep_maxT = 0;
    int64_t sweep_code_bytes = 0, sweep_sb = 0, sweep_recs = 0;
    bool sparse = false;          // ndir == 2: records of the kept columns only (JobSpec.keep)
    char* ext = nullptr;          // full matrices go here (a slab) instead of the runtime's own pools
    size_t ext_bytes = 0;
    std::vector<int> nkeep;       // [2 * job + direction]
    int build(Runtime* rt, const std::vector<JobSpec>& specs, int ndir, int lb_extra);
    int place(Runtime* rt, int P, bool can_split = false);
    double fill_alg_bytes() const;
};

// AlignData (cpp/AlignData.h:26-35) with the event data resident in HBM
struct Align {
    std::string bases;
    std::vector<int> states;
    ps_params par = {4.5, 150, 300, 0};  // cpp/AlignUtil.h:64
    int E = 0;
    std::vector<int> n;
    std::vector<int64_t> off;
    int64_t ntot = 0;
    std::vector<std::string> evseqs;
    std::vector<double> h_mean, h_stdv, h_ra, h_rl, h_model, h_trans;   // h_trans: log transition probabilities [E][4]
    bool host_refs_valid = true;
    void* slab = nullptr;
    size_t slab_cap = 0;
    void* last_stream = nullptr;   // the stream that last had work on the slab enqueued (hipStream_t; Batch::build, create, refs_to_host): ~Align
    double *d_mean = nullptr, *d_stdv = nullptr, *d_lsd = nullptr, *d_ra = nullptr, *d_rl = nullptr, *d_ri = nullptr;
    double *d_model = nullptr, *d_trans = nullptr;
    double *d_model8 = nullptr, *d_lev[2] = {nullptr, nullptr};   // k_fill's tables (ps_internal.h, JobD)
    bool fastdiv = true;                                          // tabulated reciprocals are usable (all divisors sane)
    JobOut* d_out = nullptr;
    std::map<std::string, std::vector<double>> seqlikes;  // cpp/AlignData.h:34

    ~Align();
    int create(Runtime* rt, const char* seq, int64_t seq_len, int32_t n_events, const int64_t* level_off,
               const double* mean, const double* stdv, const double* ref_align, const double* ref_like,
               const double* model, const double* trans, const char* evseq, const int64_t* evseq_off,
               const ps_params* params);
    int base_batch(Runtime* rt, Batch* b, int ndir, int lb_extra);
    int refs_to_host(Runtime* rt);
    int refs_to_host_async(Runtime* rt);   // enqueue; refs_finish() after the stream has been synchronised
    void refs_finish();
    double *pend_ra = nullptr, *pend_rl = nullptr;
};

std::vector<int> states_of(const std::string& bases);
std::string apply_edit(const std::string& b, const Mut& m);
void accumulate_likes(const double* ra, const double* rl, int n, int C, double* likes);

// strip sweeps (ps_sweep.hip, ps_sweepw.hip): K rows per lane on NW wavefronts per sweep
struct SweepForm { int K = 0, NW = 1; bool ok() const { return K > 0; } };
int sweep_guess_k(int W);                         // strip height of the one-wavefront form for realign_width W (0: too wide for a strip sweep)
SweepForm sweep_guess_form(int W, int NW);        // form to try first on NW wavefronts (K = 0: none)
SweepForm sweep_next_form(SweepForm f, int win);  // next larger one after a window of `win` strips did not fit (K = 0: none)
bool sweep_form_exists(int K, int NW);
int sweep_win_max(int NW);                        // widest window of strips a sweep on NW wavefronts supports
double sweep_job_bytes(int n0, int C, SweepForm f, bool full = false);   // bytes of one job: step codes (+ both directions' records)
int sweep_prepare(Runtime* rt, Batch& b, SweepForm f);  // band / qlo tables + the widest window (b.sd.maxwin, device)
void sweep_form_set(int K, int NW);               // tests / tuning: the form every strip sweep tries first (K <= 0: the library's choice)
int sweep_run(Runtime* rt, Batch& b);             // sweeps, maxima, backtrace, path scores (b.sd.codes placed by the caller)
void sweep_min_set(int n);                        // forward-only batches of at least n alignments take the strip sweep (< 0: default)
void sweep2_min_set(int n);                       // the same for Alignment::update batches (sweeps = 2 per alignment)
void sparse_min_set(int n);                       // Alignment::update batches whose edit list reads few columns: strip sweeps with kept columns from n sweeps on
int launch_likes(Runtime* rt, const BatchD& b, const LikeGroup* d_groups, int ngroups, double* d_out);   // per-base likelihood vectors on the device
int likes_max_states();                           // longest sequence (states) k_likes takes
bool sweep_enabled();                             // PORESEQ_NO_SWEEP unset
double fwd_job_bytes(const Align* a, int n0, int C);   // device bytes one forward-only alignment job will probably take

constexpr int PS_SPLIT = 1;   // realign(): the matrices would exceed `cap` bytes; nothing was launched, b.P holds the width they need
int realign(Runtime* rt, Batch& b, double cap = 0.0);
int score_alignments(Runtime* rt, Align* a, double* scores, double* likes);
// the *_multi forms run the same call for several AlignData (independent regions) in one launch chain
void par_for(int n, const std::function<void(int)>& fn);
int score_alignments_multi(Runtime* rt, const std::vector<Align*>& as, const std::vector<double*>& scores, const std::vector<double*>& likes);
int find_mutations_multi(Runtime* rt, const std::vector<Align*>& as, const std::vector<const std::vector<std::string>*>& seeds,
                         const std::vector<std::vector<Mut>*>& outs);
int viterbi_mutate_multi(Runtime* rt, const std::vector<Align*>& as, const std::vector<RandState*>& rngs, int nkeep, double skip, double stay,
                         double mmin, double mmax, const std::vector<std::vector<std::string>*>& outs);
std::string info_string();   // process-wide state in one line (ps_info)
int hwq_mode(std::string* why);   // 1: every stream on one priority level (a hardware queue each), 0: streams dealt over the levels
int peak_runtimes();   // most host threads that ever owned a runtime at the same time
int live_runtimes();   // host threads that currently own a runtime
int guess_slots(const Align* a);   // anti-diagonal footprint realign() will probably choose
void device_fraction_set(double f);   // the part of the device this process plans for (several ranks on one GPU); <= 0: PORESEQ_DEVICE_FRACTION, else 1
double device_fraction();
double device_share_bytes();   // this runtime's share of the device memory for its own pools (step codes, kept columns, small matrices)
// a process-wide slab for the full score matrices of one Refine-sized ScoreMutations call (ps_host.cpp)
struct SlabHold { void* s = nullptr; char* p = nullptr; size_t bytes = 0; hipStream_t drain = nullptr; void release(); ~SlabHold() { release(); } };   // release() drains `drain` first: nothing in flight may still use the slab
int slab_acquire(SlabHold* h);    // blocks while all slabs are taken
size_t slab_bytes();
double dense_cap_bytes();         // bytes of full matrices one call may place
int make_mutations_multi(Runtime* rt, const std::vector<Align*>& as, std::vector<std::vector<Mut>> muts, std::vector<int>* nbases);
// delta_out (optional): per AlignData a host array [E][M] receiving every event's term of every edit's score
int score_mutations_multi(Runtime* rt, const std::vector<Align*>& as, const std::vector<const std::vector<Mut>*>& muts,
                          const std::vector<std::vector<Mut>*>& outs, const std::vector<double*>* delta_out = nullptr);
int score_mutations(Runtime* rt, Align* a, const std::vector<Mut>& muts, std::vector<Mut>* out);
void find_point_mutations(const Align* a, std::vector<Mut>* out);
int make_mutations(Runtime* rt, Align* a, std::vector<Mut> muts, int* nbases);
int find_mutations(Runtime* rt, Align* a, const std::vector<std::string>& seeds, std::vector<Mut>* out);
int viterbi_mutate(Runtime* rt, Align* a, int nkeep, double skip, double stay, double mmin, double mmax,
                   std::vector<std::string>* out, bool verbose = false);
int debug_fill(Runtime* rt, Align* a, int ev, int dir, double* main, double* stay, uint8_t* sm, uint8_t* ss);

}  // namespace ps
#endif
