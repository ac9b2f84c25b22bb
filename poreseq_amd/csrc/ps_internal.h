// ps_internal.h — shared declarations of libporeseq_hip.so (host side + kernel launchers).
//
// Data layout in HBM:
//   levels      mean/stdv/logstdv f64, all events concatenated               (read-only per PSAlign call)
//   models      per event 6 x 1024 f64: lev_mean, lev_stdv, log_lev, sd_mean, sd_lambda, log_lambda
//   refs        per alignment job ref_align / ref_like / ref_index f64[n0]   (rewritten by the backtrace)
//   DP matrices per (job, direction): "skewed" storage  REC[s][slot], s = i + j (anti-diagonal),
//               slot = i mod P, one 16-byte record {main, stay} per cell, plus (forward) FLG[s][slot] u16
//               = {main step, stay step, score <= 0 bits}.  One anti-diagonal is one contiguous run of
//               P records: the fill kernel's stores are fully coalesced.
#ifndef PS_INTERNAL_H_
#define PS_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/poreseq_hip.h"

namespace ps {

constexpr int NS = PS_N_STATES;
constexpr double BIG = 1e300;  // "inf" of the reference, cpp/AlignUtil.h:20

// ---- error plumbing ------------------------------------------------------------------------
int fail(int code, const std::string& msg);
#define PS_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ps::fail(PS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)
#define PS_TRY(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc != PS_OK) return _rc; \
    } while (0)

// ---- runtime: device, stream, grow-only device buffers -------------------------------------
struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);  // grow-only; contents are NOT preserved across growth
    template <class T> T* as() const { return (T*)p; }
};

// grow-only pinned host buffer (hipHostMalloc): device->host copies into it are truly asynchronous
struct HBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    template <class T> T* as() const { return (T*)p; }
};

// Pinned staging arena: host<->device copies of per-call tables go through it so that hipMemcpyAsync is a
// plain enqueue (a pageable source makes the runtime stage the copy itself and block the calling thread, which
// also serialises concurrent host threads).  Memory handed out stays valid until reset(); the arena is reset at
// the next API entry of the owning thread, after its streams have drained.
struct Stage {
    struct Chunk { char* p; size_t cap; };
    std::vector<Chunk> chunks;
    size_t used = 0;          // bytes used in the last chunk
    bool dirty = false;
    void* alloc(size_t bytes);   // 256-byte aligned, nullptr when pinned memory cannot be had
    int reset();
    // scoped reuse inside one API call: everything allocated after mark() is handed back by release(); the caller
    // guarantees that no copy touching that memory is still in flight
    size_t mark() const { return chunks.size() <= 1 ? used : (size_t)-1; }
    void release(size_t m) { if (m != (size_t)-1 && chunks.size() <= 1) used = m; }
};

struct Prof { double ms = 0; int64_t launches = 0; double bytes = 0; double units = 0; };

struct Runtime {
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // Smith-Waterman batches run here, concurrently with the alignment fills
    hipEvent_t ev0 = nullptr, ev1 = nullptr, sw0 = nullptr, sw1 = nullptr;
    std::map<std::string, DBuf> pool;
    std::map<std::string, HBuf> hpool;
    HBuf& hbuf(const std::string& name) { return hpool[name]; }
    std::map<std::string, Prof> prof;
    bool prof_on = false;
    // prof_defer: event pairs are queued and read when the profile is asked for (ps_prof_get) instead of after every launch, so
    // that profiling does not serialise the host with the stream (bench.py profiles inside its timed region this way)
    bool prof_defer = false;
    struct ProfPend { hipEvent_t a, b; const char* name; double bytes; };
    std::vector<ProfPend> prof_pend;
    std::vector<hipEvent_t> prof_spare;
    DBuf& buf(const std::string& name) { return pool[name]; }
    Stage stage;
    // enqueue host -> device through the arena (the source may die as soon as this returns)
    int up(void* dst, const void* src, size_t bytes, hipStream_t st = nullptr);
    // enqueue device -> host into arena memory; *hptr is readable after the stream has been synchronised
    int down(void** hptr, const void* src, size_t bytes, hipStream_t st = nullptr);
    template <class T> int down(T** hptr, const void* src, size_t count) { return down((void**)hptr, src, count * sizeof(T)); }
};
int runtime(Runtime** out);  // PS_ERR_NO_DEVICE when no usable GPU; never falls back
// The stream Smith-Waterman batches should use: the runtime's second stream (created on first use), so that the batch overlaps with
// the base realign of FindMutations — while the calling thread is the only one inside the library.  With several threads (lock-step
// batches in flight) every runtime keeps to one stream: HIP maps streams onto 4 hardware queues by default (ps_host.cpp).
int second_stream(Runtime* rt, hipStream_t* out);
int live_runtimes();   // host threads that currently own a runtime (ps_host.cpp)

// ---- mutation list (vector<MutInfo>/vector<MutScore>, cpp/AlignUtil.h:69-91) ----------------
struct Mut {
    int start = 0;
    std::string orig, mut;
    double score = -1e-6;
};

// ---- device-side job descriptor: one (event, sequence) alignment ---------------------------
// Everything a kernel needs about a job hangs off the descriptor itself (device pointers into the owning
// AlignData's slab), so one batch can mix jobs of several AlignData handles (regions refined in lock-step).
struct JobOut;
struct JobD {
    const double* model8; // derived model of the job's event per 5-mer, rows of MODEL_ROW_BYTES: mean, 1/stdv, stdv, log stdv, sd mean, 1/sd mean, lambda, log lambda (k_fill)
    const double* lev[2]; // level records per direction, [n0][4]: forward row i -> {mean[i-1], stdv[i-1], 3 lsd[n0-i], 1/stdv[i-1]},
                          //                                      backward row i -> the same of level n0-i (cpp/Alignment.cpp:171-172, 345-349)
    const int* st;        // 5-mer states of the job's sequence [C] (4 ints of -1 padding on either side)
    double lsk, lst, lex, lin;   // log transition probabilities: skip, stay, extend, insert
    double lik_offset;
    int n0;          // levels in the event
    int C;           // states of the job's sequence
    int W;           // realign_width (band half-width of the fills)
    int force_inert; // realign_width == 0: every Alignment is a no-op (cpp/Alignment.cpp:85-86)
    int P;           // slots per anti-diagonal (multiple of 64, >= widest footprint + 9; k_fill_wide: multiple of 128, >= footprint + 2)
    int lbn;         // entries in each lb table (C + 2 + extra)
    int K;           // 0: skewed matrices (k_fill); > 0: strip matrices of K rows per lane (k_sweep2): REC[step][row of the strip][lane];
                     // < 0: column-sparse records (k_sweeps): only the columns an edit list reads, REC[kept column][row - band start]
    int pitch;       // column-sparse records: records per kept column (>= rows of the widest band)
    int NL;          // strip matrices: lanes of a sweep (64 per wavefront of its workgroup); strip q sits on lane q mod NL
    const int* keep[2];  // column-sparse records: per direction, kept-column index of column j (0 .. C + 1) or -1
    int64_t lb_off;      // lb table the fills were made with          (int32[lbn])
    int64_t lbn_off;     // lb table after the latest backtrace        (int32[lbn])
    int64_t mat_off[2];  // record offset of anti-diagonal 0 of the forward / backward matrix
    int64_t lo_off[2];   // into LO / HI (int32 per anti-diagonal)
    int64_t col_off[2];  // into per-column arrays (C+1 entries per direction)
    int64_t S;           // anti-diagonals: n0 + C + 1
    double* ra;          // ref_align  [n0]
    double* rl;          // ref_like   [n0]
    double* ri;          // ref_index  [n0]
    JobOut* out;         // per job results (persist in the AlignData between calls for its own events)
};

// per job results living in device memory
struct JobOut {
    double best;      // forward maxScore.score after the last column
    int bi, bj;       // its cell
    int has_index;    // ref_index non-empty after the latest updaterefs
    int refstart, refend;
    int inert;        // latched at the start of an API call: the reference's stripe_width == 0
    int term_i, term_w;   // where the latest backtrace stopped (bt_walk, ps_dev.h): row; column << 3 | matrix << 2 | kind
};

constexpr int MODEL_ROW_BYTES = 80;   // k_fill's model rows: 8 doubles + 16 bytes of padding (LDS bank spread)
constexpr int MAT_FRONT = 8;   // spare anti-diagonals in front of every matrix (the fill pipeline starts 8 steps early)
constexpr int MAT_BACK = 16;   // and behind it (the last loop body runs past S)
constexpr int LO_PAD = 160;    // LO / HI entries behind S, all -1 (k_fill prefetches them in chunks of 64)

struct SweepJob;
// device pointers of the pools a batch of jobs lives in (filled by Batch::build / place)
struct BatchD {
    const JobD* jobs;
    int njobs;
    int* lb;                              // pool of lb tables
    int* lo;                              // lowest in-band row per anti-diagonal (-1: none)
    int* hi;                              // highest in-band row per anti-diagonal
    double2* rec;                         // matrices {main, stay}, skewed
    unsigned short* flg;                  // forward matrices: step words {main step, stay step << 8, score <= 0 bits}
    double* cmax;                         // per column max of main
    double* pm;                           // prefix max over columns (MaxInfo.score per column)
    int* maxw;                            // widest band footprint of any job of the batch on one anti-diagonal (sizes P)
    int fastdiv;                          // every AlignData of the batch allows k_fill's tabulated reciprocals
    double log2pi;
    // strip matrices (JobD.K > 0, k_sweep2): the sweep jobs (2 per job: forward, backward) and their band / qlo / qhi tables
    const SweepJob* s_sj;
    const int2* s_band;
    const int* s_qlo;
    const int* s_qhi;
};

// ---- strip sweeps (ps_sweep.hip): forward-only alignments, one wave per job ------------------------
struct StripBest;
struct SweepJob {           // per job, parallel to BatchD.jobs
    int64_t codes_off;      // bytes into the code pool: step t of the job starts at codes_off + t * NL * K
    int64_t band_off;       // into band (int2 {i0, i1} per column, C + 2 entries)
    int64_t q_off;          // into qlo (int per step, T + 8 entries)
    int64_t sb_off;         // into the per-strip maxima (Q entries)
    int T;                  // steps: t = column + strip runs 1 .. T - 1
    int Q;                  // strips of K rows
};
struct SweepD {
    const SweepJob* sj;
    int2* band;
    int* qlo;
    int* qhi;               // lowest / highest strip in band per step (-1: none)
    unsigned char* codes;   // one byte of storage per cell, seven raw predicate bits of the fill (ps_codes.h: CB_*, code_fetch)
    StripBest* sb;
    int* maxwin;            // widest window of strips in band on one step, over the batch
    int K;
    int nl;                 // lanes of a sweep: 64 per wavefront (ps_sweepw.hip: two or four wavefronts per sweep)
    int ndir;               // 1: forward-only jobs; 2: sweep job jd = 2 * job + direction
    int sparse;             // ndir == 2: records of the kept columns only (JobD.keep) instead of every cell's
};

// one sequence's jobs (its events, in order) for the per-base likelihood vector of ScoreAlignments (k_likes, ps_sweep.hip)
struct LikeGroup {
    int job0, njobs;        // the sequence's jobs inside the batch
    int C, len;             // states of the sequence; doubles of its vector (bases)
    int64_t out_off;        // into the output pool
};

// ---- kernel launchers (ps_kernels.hip) ------------------------------------------------------
int launch_updaterefs(Runtime* rt, const BatchD& b);
int launch_lb(Runtime* rt, const BatchD& b, int which /*0: lb_off, 1: lbn_off*/, int maxlbn);
int launch_lo(Runtime* rt, const BatchD& b, int ndir, int64_t maxS);
int launch_fill(Runtime* rt, const BatchD& b, const std::vector<JobD>& jobs, int ndir, int64_t maxS, int P, int64_t ncols);
int launch_backtrace(Runtime* rt, const BatchD& b, int maxn);
int launch_prefix(Runtime* rt, const BatchD& b, int ndir);

struct ScoreArgs {
    int job0, njobs;           // the AlignData's jobs inside the batch
    int nitems_per_job;        // M
    int ncolmax;               // max new columns of any edit
    int ws;                    // scoring_width
    const int* m_start;        // [M]
    const int* m_mlen;         // [M] len(mut)
    const int* m_cm;           // [M] states of the edited sequence
    const int* m_ncol;         // [M] new columns actually produced
    const int* m_skip;         // [M] 1: edit skipped (start > len)
    const int* m_states;       // [M][ncolmax] states of the new columns
    const int* m_oldidx;       // [M] index into the unique-r0 list
    const int* r0;             // [nr0] unique max(start-3,1) values
    int nr0;
    double* old;               // [njobs][nr0]
    double* oldall;            // NULL, or [njobs][oldall_pitch]: column-pair maxima of every column (k_oldall), when the list touches most columns
    int64_t oldall_pitch, maxS;
    double* delta;             // [njobs][M]
    double* score;             // [M]
    // device lists of edit indices by new-column count: class k < 4 fits 8 << k lanes (k = 3: chunked, any size), class 4 fits 7 (point
    // edits, len(mut) <= 1: 6 or 7 new columns — nine edits to a wave instead of eight)
    const int* cls_items[5];
    int cls_count[5];
};
constexpr int SCORE_CLASSES = 5;
int launch_score(Runtime* rt, const BatchD& b, const ScoreArgs* d_sas, const std::vector<ScoreArgs>& h_sas);
int launch_begin(Runtime* rt, const BatchD& b);
int launch_gather_best(Runtime* rt, const BatchD& b, double* out);   // out[job] = the job's forward maxScore (JobOut.best)

// Smith-Waterman (ps_sw.hip)
int sw_device(Runtime* rt, const std::string& s1, const std::string& s2, int* score, double* accuracy,
              std::vector<int>* inds1, std::vector<int>* inds2);

// Viterbi (ps_viterbi.hip): one region of a batched ViterbiMutate call, host side
struct VitRegionH {
    int E = 0, T = 0;
    const double* obsin = nullptr;       // [T][E][4]: level, sd, log sd, present
    const double* d_model = nullptr;     // device, [E][6][1024]
    void* rng = nullptr;                 // the region's generator state, handed to draw()
    void (*draw)(void* rng, double* out /*[nkeep][T]*/, size_t n) = nullptr;
};
int viterbi_device_multi(Runtime* rt, const std::vector<VitRegionH>& regions, int nkeep, double skip, double stay, double mmin, double mmax,
                         std::vector<std::vector<std::vector<int>>>* paths);

void prof_flush(Runtime* rt);   // deferred mode: read the queued event pairs (drains the stream)
void prof_begin(Runtime* rt);
void prof_end(Runtime* rt, const char* name, double alg_bytes);

}  // namespace ps

#endif
