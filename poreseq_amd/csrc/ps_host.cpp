// ps_host.cpp — host side of libporeseq_hip.so: runtime, AlignData, batching, and the
// refinement-loop logic that the reference keeps in C++ above its Alignment class
// (cpp/MakeMutations.cpp, cpp/FindMutations.cpp, cpp/EventUtil.cpp, cpp/Sequence.h).
// All dynamic-programming arithmetic runs in the HIP kernels (ps_kernels.hip, ps_sw.hip,
// ps_viterbi.hip); there is no CPU implementation of it in this library.
#include "ps_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <array>
#include <functional>
#include <iterator>
#include <numeric>
#include <mutex>
#include <deque>
#include <memory>
#include <condition_variable>
#include <thread>
#include <cstdio>
#include <ctime>

namespace ps {

static thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
const char* last_error() { return g_err.c_str(); }

static double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
Tick::Tick(const char* w) : what(w), t0(now_s()) { static int e = getenv("PORESEQ_TRACE") ? 1 : 0; on = e; }
void Tick::lap(const char* label) {
    if (!on) return;
    const double t = now_s();
    fprintf(stderr, "[ps] %-18s %-22s %8.3f ms\n", what, label, 1e3 * (t - t0));
    t0 = t;
}

// ------------------------------------------------------------------------------------------ runtime
static size_t trim_idle_runtimes();   // below: hands back the device pools of runtimes no thread owns
static size_t device_total_bytes();
static size_t device_plan_bytes();   // this process's part of it (ps_set_device_fraction)
static std::atomic<long long> g_pool_bytes(0);   // device memory held by the pools of all runtimes

int DBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return PS_OK;
    static const bool trace = getenv("PORESEQ_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[ps] pool grow %zu -> %zu bytes\n", cap, bytes);
    if (p) { PS_HIP(hipFree(p)); p = nullptr; g_pool_bytes -= (long long)cap; cap = 0; }
    size_t want = std::max<size_t>(bytes + std::min<size_t>(bytes / 4, (size_t)1 << 30), 1 << 16);   // growth slack, at most 1 GB
    // never into the last 6 % of the device: a launch that finds no memory for the HSA runtime's own needs aborts the process
    // (HSA_STATUS_ERROR_OUT_OF_RESOURCES) — refuse here instead, callers that can cut their batch do so on PS_ERR_NOMEM
    if (const size_t tot = device_plan_bytes()) {
        auto over = [&](size_t w) { return (double)g_pool_bytes.load() + (double)w > 0.94 * (double)tot; };
        if (over(want)) want = std::max<size_t>(bytes, 1 << 16);
        if (over(want)) (void)trim_idle_runtimes();
        if (over(want))
            return fail(PS_ERR_NOMEM, "device pools of this process would reach " + std::to_string((size_t)((g_pool_bytes.load() + (long long)want) >> 20)) + " MB of " +
                                      std::to_string(tot >> 20) + " MB (a buffer of " + std::to_string(want >> 20) + " MB was asked for)");
    }
    if (hipMalloc(&p, want) != hipSuccess) {
        p = nullptr;
        (void)hipGetLastError();   // (sticky: the next launch check would report it)
        want = std::max<size_t>(bytes, 1 << 16);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            // runtimes on the free list (their threads are gone) keep their pools for the next thread: take them back first
            (void)hipGetLastError();
            const size_t got = trim_idle_runtimes();
            if (trace) fprintf(stderr, "[ps] out of device memory: %zu bytes taken back from idle runtimes\n", got);
            p = nullptr;
            e = got ? hipMalloc(&p, want) : e;
        }
        if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); 
            size_t fr = 0, tt = 0;
            (void)hipMemGetInfo(&fr, &tt);
            return fail(PS_ERR_NOMEM, std::string("hipMalloc of ") + std::to_string(want >> 20) + " MB: " + hipGetErrorString(e) + " (" + std::to_string(fr >> 20) + " of " +
                                      std::to_string(tt >> 20) + " MB free, " + std::to_string((long long)(g_pool_bytes.load() >> 20)) + " MB in this process's pools)");
        }
    }
    cap = want;
    g_pool_bytes += (long long)cap;
    return PS_OK;
}

int HBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return PS_OK;
    if (p) { PS_HIP(hipHostFree(p)); p = nullptr; cap = 0; }
    const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 16);
    PS_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return PS_OK;
}

void* Stage::alloc(size_t bytes) {
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    if (chunks.empty() || used + bytes > chunks.back().cap) {
        size_t want = std::max<size_t>(bytes, chunks.empty() ? (size_t)4 << 20 : 2 * chunks.back().cap);
        void* p = nullptr;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            want = bytes;
            if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
        }
        chunks.push_back({(char*)p, want});
        used = 0;
    }
    void* r = chunks.back().p + used;
    used += bytes;
    dirty = true;
    return r;
}

int Stage::reset() {
    if (chunks.size() > 1) {   // grew during the last call: one chunk of the combined size from now on
        size_t tot = 0;
        for (Chunk& c : chunks) { tot += c.cap; PS_HIP(hipHostFree(c.p)); }
        chunks.clear();
        void* p = nullptr;
        if (hipHostMalloc(&p, tot, hipHostMallocDefault) == hipSuccess) chunks.push_back({(char*)p, tot});
    }
    used = 0;
    dirty = false;
    return PS_OK;
}

int Runtime::up(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return PS_OK;
    void* h = stage.alloc(bytes);
    if (!h) return fail(PS_ERR_NOMEM, "hipHostMalloc (staging arena)");
    memcpy(h, src, bytes);
    PS_HIP(hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, st ? st : stream));
    return PS_OK;
}

int Runtime::down(void** hptr, const void* src, size_t bytes, hipStream_t st) {
    void* h = stage.alloc(bytes);
    if (!h) return fail(PS_ERR_NOMEM, "hipHostMalloc (staging arena)");
    *hptr = h;
    if (bytes) PS_HIP(hipMemcpyAsync(h, src, bytes, hipMemcpyDeviceToHost, st ? st : stream));
    return PS_OK;
}

// One runtime (HIP streams + grow-only device pools) per host thread that is inside the library: independent
// PSAlign pipelines driven from different threads run concurrently on the GPU — a single region keeps at most a
// few dozen of the 256 CUs busy, and regions are independent work-items.  Runtimes live in a process-wide
// free-list: a thread adopts one on its first call and hands it back when it exits, so short-lived worker
// threads reuse the pools instead of re-allocating (or leaking) them.
namespace {
struct RtSlot { Runtime R; int state = 0; std::string why; };   // state: 0 untried, 1 ok, -1 failed
std::mutex g_rt_mu;
std::vector<RtSlot*> g_rt_free;
struct RtHolder {
    RtSlot* s = nullptr;
    ~RtHolder();
};
thread_local RtHolder t_rt;
std::atomic<int> g_rt_live(0);
std::atomic<int> g_rt_peak(0);   // most threads that owned a runtime at the same time (forgotten a minute after the count was last that high)
std::atomic<double> g_rt_peak_at(0.0);
}  // namespace
int live_runtimes() { return g_rt_live.load(); }
int peak_runtimes() {
    // a burst of threads long ago must not shrink a later lone caller's share for good
    const double t = now_s();
    if (t - g_rt_peak_at.load() > 60.0) { g_rt_peak.store(std::max(g_rt_live.load(), 1)); g_rt_peak_at.store(t); }
    return g_rt_peak.load();
}
static size_t trim_idle_runtimes() {
    std::lock_guard<std::mutex> lk(g_rt_mu);
    size_t got = 0;
    for (RtSlot* s : g_rt_free)
        for (auto& kv : s->R.pool) {
            DBuf& b = kv.second;
            if (b.p && hipFree(b.p) == hipSuccess) got += b.cap;   // (the owning thread drained its streams before it left)
            if (b.p) g_pool_bytes -= (long long)b.cap;
            b.p = nullptr; b.cap = 0;
        }
    return got;
}
RtHolder::~RtHolder() {
    if (!s) return;
    if (getenv("PORESEQ_TRACE")) {   // what this thread's runtime holds, largest first
        std::vector<std::pair<size_t, std::string>> v;
        size_t tot = 0;
        for (auto& kv : s->R.pool) { v.push_back({kv.second.cap, kv.first}); tot += kv.second.cap; }
        std::sort(v.rbegin(), v.rend());
        std::string line = "[ps] runtime handed back: " + std::to_string(tot >> 20) + " MB of device pools:";
        for (size_t k = 0; k < v.size() && k < 12; k++) line += " " + v[k].second + " " + std::to_string(v[k].first >> 20);
        fprintf(stderr, "%s\n", line.c_str());
    }
    std::lock_guard<std::mutex> lk(g_rt_mu);
    g_rt_free.push_back(s);
    g_rt_live--;
}

// How the runtimes' streams get hardware queues (see make_stream below).  GPU_MAX_HW_QUEUES only counts when HIP read it, i.e. when
// it was in the environment before the HIP runtime started: either the process was started with it (/proc/self/environ is the
// environment at exec, later setenv calls do not show there), or the poreseq_amd package exported it at import after checking that
// nothing in the process had opened the GPU yet (it then sets PORESEQ_HWQ_SET_BY_PACKAGE=1).  A value that appeared any other way
// is not trusted: seven streams on four queues of ONE priority level would be the slowest arrangement of all (108 against 141 kb/s).
static int hwq_from_exec_env() {
    static const int v = [] {
        FILE* f = fopen("/proc/self/environ", "rb");
        if (!f) return 0;
        std::string all;
        char buf[4096];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) all.append(buf, n);
        fclose(f);
        const std::string key = "GPU_MAX_HW_QUEUES=";
        for (size_t at = 0; at < all.size();) {
            const size_t end = all.find('\0', at);
            const std::string kv = all.substr(at, end == std::string::npos ? std::string::npos : end - at);
            if (kv.compare(0, key.size(), key) == 0) return atoi(kv.c_str() + key.size());
            if (end == std::string::npos) break;
            at = end + 1;
        }
        return 0;
    }();
    return v;
}
int hwq_mode(std::string* why) {
    static int mode = -1;
    static std::string reason;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (mode < 0) {
        const char* e = getenv("GPU_MAX_HW_QUEUES");
        const int now = e ? atoi(e) : 0;
        const char* pk = getenv("PORESEQ_HWQ_SET_BY_PACKAGE");
        if (getenv("PORESEQ_ONE_PRIORITY")) { mode = 1; reason = "one priority level (PORESEQ_ONE_PRIORITY)"; }
        else if (getenv("PORESEQ_PRIORITY_LEVELS")) { mode = 0; reason = "streams dealt over the priority levels (PORESEQ_PRIORITY_LEVELS)"; }
        else if (hwq_from_exec_env() >= 8) { mode = 1; reason = "one priority level, a hardware queue per stream (GPU_MAX_HW_QUEUES=" + std::to_string(hwq_from_exec_env()) + " in the process's start-up environment)"; }
        else if (now >= 8 && pk && atoi(pk) == 1) { mode = 1; reason = "one priority level, a hardware queue per stream (GPU_MAX_HW_QUEUES=" + std::to_string(now) + " exported by the poreseq_amd package before HIP started)"; }
        else {
            mode = 0;
            reason = now >= 8 ? "streams dealt over the priority levels (GPU_MAX_HW_QUEUES=" + std::to_string(now) + " appeared after start-up without the package's guarantee that HIP had not started: not trusted)"
                              : "streams dealt over the priority levels (HIP's default of 4 hardware queues per level)";
        }
    }
    if (why) *why = reason;
    return mode;
}

int second_stream(Runtime* rt, hipStream_t* out) {
    // One stream per runtime as soon as several host threads drive the GPU (lock-step batches in flight): HIP maps streams onto
    // 4 hardware queues by default, and the 5th stream serialises behind another one — measured: 4 batches x 1 stream 101 kb/s,
    // 4 batches x 2 streams 69 kb/s; the overlap a second stream buys comes from the other batches anyway.
    // (PORESEQ_ONE_STREAM forces it for a lone thread too; read once: getenv races with setenv from other threads.)
    static const bool one = getenv("PORESEQ_ONE_STREAM") != nullptr;
    // PORESEQ_FORCE_STREAM2 (diagnostics only, tests/test_hip_variant.py and DESIGN.md section 9): second streams even with
    // several threads inside the library; "prio" puts them on the next stream priority level, as round 2's experiment did
    static const char* force = getenv("PORESEQ_FORCE_STREAM2");
    if (!force && (one || live_runtimes() > 1)) { *out = rt->stream; return PS_OK; }
    if (!rt->stream2 && force && !strcmp(force, "prio")) {
        static std::atomic<int> seq(1);
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo > hi)
            PS_HIP(hipStreamCreateWithPriority(&rt->stream2, hipStreamNonBlocking, hi + seq++ % (lo - hi + 1)));
    }
    if (!rt->stream2) PS_HIP(hipStreamCreateWithFlags(&rt->stream2, hipStreamNonBlocking));
    *out = rt->stream2;
    return PS_OK;
}

int runtime(Runtime** out) {
    if (!t_rt.s) {
        std::lock_guard<std::mutex> lk(g_rt_mu);
        if (!g_rt_free.empty()) { t_rt.s = g_rt_free.back(); g_rt_free.pop_back(); }
        else t_rt.s = new RtSlot();
        {
            const int n = ++g_rt_live;
            int pk = g_rt_peak.load();
            while (n > pk && !g_rt_peak.compare_exchange_weak(pk, n)) {}
            if (n >= g_rt_peak.load()) g_rt_peak_at.store(now_s());
        }
        if (t_rt.s->state == 1) (void)hipSetDevice(t_rt.s->R.device);   // the current device is per-thread state
    }
    Runtime& R = t_rt.s->R;
    int& state = t_rt.s->state;
    std::string& why = t_rt.s->why;
    if (state == 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            state = -1;
            why = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                  "); libporeseq_hip has no CPU fallback";
        } else {
            int dev = 0;
            if (const char* s = getenv("PORESEQ_DEVICE")) dev = atoi(s);
            else if (const char* s2 = getenv("LOCAL_RANK")) dev = atoi(s2) % n;
            if (dev < 0 || dev >= n) dev = 0;
            hipDeviceProp_t prop;
            // One non-blocking stream for the alignment pipeline; a second one for Smith-Waterman batches (they
            // overlap with the base realign inside FindMutations) is created on first use (second_stream()).
            // Partitioning the CUs between them (hipExtStreamCreateWithCUMask) was measured and made no
            // difference, so it is not used.
            // Streams and hardware queues.  HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues per stream
            // priority level (default 4), and streams that share a queue run their kernels one after the other: seven lock-step
            // batches on one level = seven streams on four queues, three kernels in flight on average, 108 kb/s.
            //  * GPU_MAX_HW_QUEUES >= 8 in force (hwq_mode() above: in the environment the process started with, or exported by the
            //    poreseq_amd package before HIP started): every runtime's stream on the default level, a queue each — 146-147 kb/s.
            //  * otherwise the streams are dealt round-robin to the device's three priority levels, not for the priorities' sake but
            //    for the 3 x 4 queues: 141 kb/s — the two or three batches on the lowest level finish ~0.8 s after the others
            //    (profiles/r03_d_sweep_forms.md).
            // (PORESEQ_ONE_PRIORITY=1 forces the default level, PORESEQ_PRIORITY_LEVELS=1 the dealing.)
            auto make_stream = [&](hipStream_t* st) {
                static std::atomic<int> seq(0);
                const bool one = hwq_mode(nullptr) == 1;
                int lo = 0, hi = 0;
                if (!one && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo > hi)
                    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi + seq++ % (lo - hi + 1));
                return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
            };
            if (hipSetDevice(dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
                make_stream(&R.stream) != hipSuccess ||
                hipEventCreate(&R.ev0) != hipSuccess || hipEventCreate(&R.ev1) != hipSuccess ||
                hipEventCreate(&R.sw0) != hipSuccess || hipEventCreate(&R.sw1) != hipSuccess) {
                state = -1; why = "HIP device initialisation failed";
            } else if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
                state = -1; why = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
            } else {
                R.device = dev; R.ready = true; state = 1;
            }
        }
    }
    if (state < 0) return fail(PS_ERR_NO_DEVICE, why);
    if (R.stage.dirty) {   // a new API call: nothing staged by the previous one may still be in flight
        PS_HIP(hipStreamSynchronize(R.stream));
        if (R.stream2) PS_HIP(hipStreamSynchronize(R.stream2));
        PS_TRY(R.stage.reset());
    }
    *out = &R;
    return PS_OK;
}

static hipEvent_t prof_event(Runtime* rt) {
    hipEvent_t e = nullptr;
    if (!rt->prof_spare.empty()) { e = rt->prof_spare.back(); rt->prof_spare.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
}
void prof_begin(Runtime* rt) {
    if (!rt->prof_on) return;
    if (rt->prof_defer) {
        Runtime::ProfPend p{prof_event(rt), nullptr, nullptr, 0.0};
        if (p.a) (void)hipEventRecord(p.a, rt->stream);
        rt->prof_pend.push_back(p);
        return;
    }
    (void)hipEventRecord(rt->ev0, rt->stream);
}
void prof_end(Runtime* rt, const char* name, double bytes) {
    if (!rt->prof_on) return;
    if (rt->prof_defer) {
        if (rt->prof_pend.empty() || rt->prof_pend.back().b) return;
        Runtime::ProfPend& p = rt->prof_pend.back();
        p.b = prof_event(rt); p.name = name; p.bytes = bytes;   // (names are string literals)
        if (p.b) (void)hipEventRecord(p.b, rt->stream);
        return;
    }
    (void)hipEventRecord(rt->ev1, rt->stream);
    (void)hipEventSynchronize(rt->ev1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, rt->ev0, rt->ev1);
    Prof& p = rt->prof[name];
    p.ms += ms; p.launches += 1; p.bytes += bytes;
}
// read the queued event pairs (deferred mode); the stream is drained first
void prof_flush(Runtime* rt) {
    if (rt->prof_pend.empty()) return;
    (void)hipStreamSynchronize(rt->stream);
    for (Runtime::ProfPend& q : rt->prof_pend) {
        float ms = 0;
        if (q.a && q.b && q.name && hipEventElapsedTime(&ms, q.a, q.b) == hipSuccess) {
            Prof& p = rt->prof[q.name];
            p.ms += ms; p.launches += 1; p.bytes += q.bytes;
        }
        if (q.a) rt->prof_spare.push_back(q.a);
        if (q.b) rt->prof_spare.push_back(q.b);
    }
    rt->prof_pend.clear();
}

// ------------------------------------------------------------------------------------------ sequences
// Sequence::populateStates, cpp/Sequence.h:64-100
std::vector<int> states_of(const std::string& bases) {
    std::vector<int> st;
    if (bases.size() < 5) return st;
    auto code = [](char c) -> int { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : (int)(signed char)c; };
    int cur = 0;
    for (int i = 0; i < 4; i++) cur = (cur << 2) + code(bases[i]);
    st.resize(bases.size() - 4);
    for (size_t i = 4; i < bases.size(); i++) {
        if (code(bases[i - 4]) < 4) { cur = (NS - 1) & ((cur << 2) + code(bases[i])); st[i - 4] = cur; }
        else { cur = 0; st[i - 4] = -1; }
    }
    return st;
}

// Sequence(original, mut), cpp/Sequence.h:37-59
std::string apply_edit(const std::string& b, const Mut& m) {
    if ((size_t)m.start >= b.size()) return b;
    std::string r = b.substr(0, m.start);
    r += m.mut;
    size_t rem = (size_t)m.start + m.orig.size();
    if (rem < b.size()) r += b.substr(rem);
    return r;
}

// ------------------------------------------------------------------------------------------ batch
int Batch::build(Runtime* rt, const std::vector<JobSpec>& specs, int ndir_, int lb_extra) {
    ndir = ndir_;
    P = 0; Pmax = 64;
    d.fastdiv = 1;
    jobs.clear();
    maxS = 0; maxC = 0; maxn = 0; maxlbn = 0;
    sparse = ndir == 2 && !specs.empty();
    nkeep.clear();
    constexpr int ST_PAD = 8;   // ints of -1 around every state list: k_fill fetches four states per 16-byte load
    int64_t st_tot = 0, lb_tot = 0, lo_tot = 0, col_tot = 0;
    std::vector<int> h_states;
    std::map<const std::vector<int>*, int64_t> st_seen;
    std::vector<int64_t> st_at(specs.size());
    for (size_t k = 0; k < specs.size(); k++) {
        const JobSpec& s = specs[k];
        const Align* a = s.a;
        s.a->last_stream = (void*)rt->stream;   // (every kernel over an AlignData's events goes through a Batch: ~Align, slab cache)
        const int W = a->par.realign_width;
        if (W < 0) return fail(PS_ERR_BAD_ARG, "realign_width < 0");
        // widest possible footprint 2W + 1, plus the idle slots k_fill wants between two rows of a lane; the slots actually
        // used follow the measured footprint (realign), typically (2W + 1) / (1 + levels per base) + 9
        const int pm = std::max(64, ((2 * W + 10 + 63) / 64) * 64);
        Pmax = std::max(Pmax, std::min(pm, 2048));
        JobD j;
        memset(&j, 0, sizeof(j));
        j.model8 = a->d_model8 + (size_t)s.ev * (MODEL_ROW_BYTES / 8) * NS;
        j.lev[0] = a->d_lev[0] + 4 * a->off[s.ev]; j.lev[1] = a->d_lev[1] + 4 * a->off[s.ev];
        if (!a->fastdiv) d.fastdiv = 0;
        j.lsk = a->h_trans[s.ev * 4 + 0]; j.lst = a->h_trans[s.ev * 4 + 1]; j.lex = a->h_trans[s.ev * 4 + 2]; j.lin = a->h_trans[s.ev * 4 + 3];
        j.lik_offset = a->par.lik_offset;
        j.n0 = a->n[s.ev]; j.C = (int)s.states->size(); j.W = W; j.P = 0;
        j.force_inert = (W == 0) ? 1 : 0;
        j.lbn = j.C + 2 + lb_extra;
        auto it = st_seen.find(s.states);
        if (it == st_seen.end()) {
            h_states.insert(h_states.end(), ST_PAD, -1);
            st_seen[s.states] = st_tot + ST_PAD; st_at[k] = st_tot + ST_PAD;
            h_states.insert(h_states.end(), s.states->begin(), s.states->end());
            h_states.insert(h_states.end(), ST_PAD, -1);
            st_tot += j.C + 2 * ST_PAD;
        } else st_at[k] = it->second;
        j.lb_off = lb_tot; lb_tot += j.lbn;
        j.lbn_off = lb_tot; lb_tot += j.lbn;
        j.S = (int64_t)j.n0 + j.C + 1;
        for (int d = 0; d < ndir; d++) {
            j.lo_off[d] = lo_tot; lo_tot += j.S + LO_PAD;
            j.col_off[d] = col_tot; col_tot += j.C + 1;
        }
        j.ra = s.ra; j.rl = s.rl; j.ri = s.ri; j.out = s.out;
        j.keep[0] = s.keep[0]; j.keep[1] = s.keep[1];
        if (!s.keep[0] || !s.keep[1]) sparse = false;
        nkeep.push_back(s.nkeep[0]); nkeep.push_back(s.nkeep[1]);
        maxS = std::max(maxS, j.S); maxC = std::max(maxC, j.C); maxn = std::max(maxn, j.n0); maxlbn = std::max(maxlbn, j.lbn);
        jobs.push_back(j);
    }
    ncols = col_tot;
    PS_TRY(rt->buf("jobs").ensure(std::max<size_t>(jobs.size(), 1) * sizeof(JobD)));
    PS_TRY(rt->buf("states").ensure(std::max<size_t>(h_states.size(), 1) * sizeof(int)));
    PS_TRY(rt->buf("lb").ensure(std::max<int64_t>(lb_tot, 1) * sizeof(int)));
    PS_TRY(rt->buf("lo").ensure(std::max<int64_t>(lo_tot, 1) * sizeof(int)));
    PS_TRY(rt->buf("hi").ensure(std::max<int64_t>(lo_tot, 1) * sizeof(int)));
    PS_TRY(rt->buf("cmax").ensure(std::max<int64_t>(col_tot, 1) * sizeof(double)));
    PS_TRY(rt->buf("pm").ensure(std::max<int64_t>(col_tot, 1) * sizeof(double)));
    PS_TRY(rt->buf("maxw").ensure(64));
    const int* d_states = rt->buf("states").as<int>();
    for (size_t k = 0; k < jobs.size(); k++) jobs[k].st = d_states + st_at[k];
    PS_TRY(rt->up(rt->buf("jobs").p, jobs.data(), jobs.size() * sizeof(JobD)));
    PS_TRY(rt->up(rt->buf("states").p, h_states.data(), h_states.size() * sizeof(int)));
    d.jobs = rt->buf("jobs").as<JobD>();
    d.njobs = (int)jobs.size();
    d.lb = rt->buf("lb").as<int>(); d.lo = rt->buf("lo").as<int>(); d.hi = rt->buf("hi").as<int>();
    d.rec = nullptr; d.flg = nullptr;
    d.s_sj = nullptr; d.s_band = nullptr; d.s_qlo = nullptr; d.s_qhi = nullptr;
    d.cmax = rt->buf("cmax").as<double>(); d.pm = rt->buf("pm").as<double>();
    d.maxw = rt->buf("maxw").as<int>();
    d.log2pi = std::log(2 * M_PI);  // cpp/AlignUtil.h:24
    return PS_OK;
}

// This runtime's share of the device memory for DP matrices (rec + flg = 18 bytes per slot): PORESEQ_MAX_BATCH_GB when set (read
// at every call), otherwise 65 % of the device divided by the most threads that owned a runtime at once (at least four; the maximum is
// forgotten a minute after the count was last that high).  Within a run of a multi-threaded driver the count only goes up, so
// shares only go down: pools sized under a larger share are given back at the owner's next Batch::place, and after the first
// step every pool fits its share (no regrowth, no thrash).
// 288 GB -> 47 GB per runtime up to four threads: a lone region's 170 candidate alignments (24 GB) stay one launch; a lock-step
// batch of 16 regions takes ~170 workgroups of two 10 kb sweeps per launch.  Callers size their batches on a guess of the band
// footprint (guess_slots) and split when realign() finds the matrices 20 % over the share, or the device short of memory.
static size_t device_total_bytes() {
    static std::atomic<size_t> dev_total(0);       // the device's memory size does not change: asked once
    size_t tot = dev_total.load();
    if (!tot) {
        size_t fr = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || !tot) return 0;
        dev_total.store(tot);
    }
    return tot;
}

// The part of the device this PROCESS plans for.  One process per GPU (the normal deployment) owns the device: 1.  Several ranks on one
// device (poreseq_amd.dist.init with more ranks than GPUs; the 8-rank test of the driver's command line on one GPU) each plan for
// their fraction — ps_set_device_fraction / PORESEQ_DEVICE_FRACTION — so that the slabs, the runtimes' shares and the 94 % guard of
// the pools add up to one device, not to one device per rank.
static std::atomic<double> g_dev_fraction(-1.0);
void device_fraction_set(double f) { g_dev_fraction.store(f > 0.0 && f <= 1.0 ? f : -1.0); }
double device_fraction() {
    const double f = g_dev_fraction.load();
    if (f > 0.0) return f;
    static const double env = [] { const char* e = getenv("PORESEQ_DEVICE_FRACTION"); const double v = e ? atof(e) : 1.0; return v > 0.0 && v <= 1.0 ? v : 1.0; }();
    return env;
}
static size_t device_plan_bytes() { return (size_t)(device_fraction() * (double)device_total_bytes()); }

// Memory plan of the device (309 GB on an MI355X): 27 % in three slabs for full score matrices (below), 13 % left alone (the HSA
// runtime aborts the process when a launch finds no memory for its own needs, and hipMalloc rounds: pools that sum to 302 GB left
// 5 GB free), 60 % to the runtimes.  What a runtime holds follows its share: the matrix pool grows 6 % past it, small forward batches
// (k_fill) add an eighth in step words, Smith-Waterman checkpoints a quarter, and ~2.5 GB do not depend on it (remapped alignments,
// band tables, edit tables): 1.4 x share + 2.5 GB, measured at 7, 10 and 14 batches in flight.  Share: 31 GB up to four threads,
// 17 GB with seven batches in flight, 7.7 GB with fourteen.  It sizes the chunks of FindMutations' candidate alignments (7 MB of
// step codes each), of Smith-Waterman batches and of the Viterbi tables.
double device_share_bytes() {
    if (const char* e = getenv("PORESEQ_MAX_BATCH_GB")) { const double g = atof(e); if (g > 0) return g * 1e9; }
    const size_t tot = device_plan_bytes();
    if (!tot) return 32e9;
    const int nrt = std::max(4, peak_runtimes());
    return std::max(2e9, (0.60 * (double)tot / nrt - 2.5e9) / 1.4);
}

// ---- slabs for full score matrices -----------------------------------------------------------------------------------------
// Only a ScoreMutations call whose edit list reads most columns (Refine / ScorePoints: point edits at every position, ~4 % of a
// consensus schedule's calls) keeps full forward + backward matrices: 265 MB per 10 kb event, 53 GB for a lock-step call of 20
// regions.  Sizing every runtime's pools for that (round 3: 65 % of the device divided by the batches in flight) made the number of
// batches in flight a memory question.  Instead the process keeps a few slabs (PORESEQ_SLABS, default 3, of PORESEQ_SLAB_GB,
// default 9 % of the device each: 28 GB on an MI355X = the matrices of 10 regions per launch), allocated on first use and never
// freed; a dense call takes one for its
// duration (fills, backtrace, edit scoring, read-back) and waits when all are taken.  Nothing is acquired while a slab is held.
// A single AlignData whose matrices exceed a slab (a 48 kb region with 30 events: 41 GB) takes the calling runtime's own pools.
namespace {
struct Slab { char* p = nullptr; size_t bytes = 0; bool busy = false; };
std::mutex g_slab_mu;
std::condition_variable g_slab_cv;
std::vector<Slab*> g_slabs;
}  // namespace
static int slab_count() { static const int n = getenv("PORESEQ_SLABS") ? std::max(1, atoi(getenv("PORESEQ_SLABS"))) : 3; return n; }
size_t slab_bytes() {
    static const double gb = getenv("PORESEQ_SLAB_GB") ? atof(getenv("PORESEQ_SLAB_GB")) : 0.0;
    if (gb > 0) return (size_t)(gb * 1e9);
    const size_t tot = device_plan_bytes();
    return tot ? (size_t)(0.09 * (double)tot) : (size_t)24e9;
}
// bytes of full matrices one dense call may place: the slab, or PORESEQ_MAX_BATCH_GB when set (tests: tiny budgets)
double dense_cap_bytes() {
    // (the smallest slab actually allocated, when one came out smaller than planned: sub-batches are cut to fit any of them)
    size_t cap = slab_bytes();
    { std::lock_guard<std::mutex> lk(g_slab_mu); for (const Slab* sl : g_slabs) cap = std::min(cap, sl->bytes); }
    if (const char* e = getenv("PORESEQ_MAX_BATCH_GB")) { const double g = atof(e); if (g > 0) return std::min(g * 1e9, (double)cap); }
    return (double)cap;
}
void SlabHold::release() {
    if (!s) return;
    if (drain) (void)hipStreamSynchronize(drain);   // (a no-op on the normal path: the call has read its results back)
    { std::lock_guard<std::mutex> lk(g_slab_mu); ((Slab*)s)->busy = false; }
    s = nullptr; p = nullptr; bytes = 0;
    g_slab_cv.notify_one();
}
int slab_acquire(SlabHold* h) {
    std::unique_lock<std::mutex> lk(g_slab_mu);
    for (;;) {
        for (Slab* sl : g_slabs) if (!sl->busy) { sl->busy = true; h->s = sl; h->p = sl->p; h->bytes = sl->bytes; return PS_OK; }
        if ((int)g_slabs.size() < slab_count()) {
            Slab* sl = new Slab();
            size_t want = slab_bytes();
            hipError_t e = hipMalloc((void**)&sl->p, want);
            if (e != hipSuccess) {   // the device is fuller than expected: idle runtimes' pools first, then a smaller slab
                (void)hipGetLastError();
                (void)trim_idle_runtimes();
                for (int k = 0; k < 3 && e != hipSuccess; k++) { if (k) want = want / 4 * 3; e = hipMalloc((void**)&sl->p, want); if (e != hipSuccess) (void)hipGetLastError(); }
            }
            if (e != hipSuccess) {
                delete sl;
                if (!g_slabs.empty()) { g_slab_cv.wait(lk); continue; }   // make do with the slabs there are
                return fail(PS_ERR_NOMEM, std::string("hipMalloc of a ") + std::to_string(want >> 20) + " MB slab for full score matrices: " + hipGetErrorString(e));
            }
            sl->bytes = want; sl->busy = true;
            g_pool_bytes += (long long)want;
            g_slabs.push_back(sl);
            h->s = sl; h->p = sl->p; h->bytes = sl->bytes;
            return PS_OK;
        }
        g_slab_cv.wait(lk);
    }
}

// The DP matrices ("rec": 16-byte records, or a strip sweep's step codes, which alias it; "flg": step words) are the only big pools,
// and several runtimes size theirs at different times (the share depends on how many threads are inside the library).  Two rules
// keep the sum below the device: a runtime whose pools were sized for a much larger share than today's gives them back before
// re-sizing, and no matrix pool grows into the last 8 % of the device (small buffers of every runtime live there) — PS_ERR_NOMEM
// instead, which callers that can split turn into smaller batches.
static int ensure_matrix_pools(Runtime* rt, const Batch& bt, size_t need_rec, size_t need_flg, bool can_split, void** rec_out, void** flg_out) {
    if (bt.ext) {   // full matrices of a dense ScoreMutations call: carved out of the slab the caller holds
        const size_t r = (need_rec + 255) & ~(size_t)255;
        if (r + need_flg > bt.ext_bytes)
            return fail(PS_ERR_NOMEM, "the score matrices of this call (" + std::to_string((r + need_flg) >> 20) + " MB) do not fit a slab of " + std::to_string(bt.ext_bytes >> 20) +
                                      " MB (PORESEQ_SLAB_GB)");
        *rec_out = bt.ext; *flg_out = bt.ext + r;
        return PS_OK;
    }
    DBuf& rec = rt->buf("rec");
    DBuf& flg = rt->buf("flg");
    size_t tot = device_plan_bytes();
    if (!getenv("PORESEQ_MAX_BATCH_GB") && rec.p && (double)rec.cap > 1.5 * device_share_bytes() + 2e9 && need_rec < rec.cap) {
        PS_HIP(hipStreamSynchronize(rt->stream));
        PS_HIP(hipFree(rec.p)); g_pool_bytes -= (long long)rec.cap; rec.p = nullptr; rec.cap = 0;
        if (flg.p) { PS_HIP(hipFree(flg.p)); g_pool_bytes -= (long long)flg.cap; flg.p = nullptr; flg.cap = 0; }
    }
    if (tot && (need_rec > rec.cap || need_flg > flg.cap)) {
        auto over = [&] { return (double)(g_pool_bytes.load() - (long long)rec.cap - (long long)flg.cap) + (double)need_rec + (double)need_flg > 0.92 * (double)tot; };
        if (over()) (void)trim_idle_runtimes();
        if (over() && can_split) return fail(PS_ERR_NOMEM, "the DP matrices of this batch do not fit beside the pools of the other threads' batches");
    }
    PS_TRY(rec.ensure(need_rec));
    if (need_flg) PS_TRY(flg.ensure(need_flg));
    *rec_out = rec.p; *flg_out = flg.p;
    return PS_OK;
}

// second phase: the anti-diagonal footprint of every band is known, size the skewed matrices
int Batch::place(Runtime* rt, int P_, bool can_split) {
    P = std::min(Pmax, std::max(64, ((P_ + 63) / 64) * 64));
    if (const char* fp = getenv("PORESEQ_DEBUG_MIN_P")) P = std::min(1024, std::max(P, (atoi(fp) + 63) / 64 * 64));  // tests / experiments: more slots than needed
    int64_t mat_tot = 0;
    for (JobD& j : jobs) {
        j.P = P;
        // spare anti-diagonals in front of and behind every matrix: k_fill's pipeline starts early and runs past S
        for (int dd = 0; dd < ndir; dd++) { j.mat_off[dd] = mat_tot + (int64_t)MAT_FRONT * P; mat_tot += (j.S + MAT_FRONT + MAT_BACK) * P; }
    }
    cells = mat_tot;
    void *prec = nullptr, *pflg = nullptr;
    PS_TRY(ensure_matrix_pools(rt, *this, (size_t)std::max<int64_t>(mat_tot, 1) * sizeof(double2), (size_t)std::max<int64_t>(mat_tot, 1) * sizeof(unsigned short), can_split, &prec, &pflg));
    PS_TRY(rt->up(rt->buf("jobs").p, jobs.data(), jobs.size() * sizeof(JobD)));
    d.rec = (double2*)prec; d.flg = (unsigned short*)pflg;
    return PS_OK;
}

// bytes the fills of this batch move by the SURVEY 8(d) accounting: 18 B (fwd) / 16 B (back) per band cell
double Batch::fill_alg_bytes() const {
    double t = 0;
    for (const JobD& j : jobs) {
        const double band = std::min<double>(2.0 * j.W + 1, j.n0);
        t += (double)j.C * band * (ndir == 2 ? 34.0 : 18.0) + 24.0 * j.n0 * ndir;
    }
    return t;
}

// ------------------------------------------------------------------------------------------ AlignData
// The slab of an AlignData (events, derived tables, results: ~13 MB for a 10 kb region at 10x) comes from a process-wide cache and goes
// back to it.  hipFree waits for EVERY stream of the process (43 ms per call with fourteen lock-step batches in flight: 280 regions per
// bench step were 12 s of blocked slot threads), hipMalloc takes ~2 ms; a cached slab costs an event: recorded on the stream that
// last had work on the slab when its AlignData goes, waited for (on the device, not the host) by the stream of the next owner.
namespace {
struct CachedSlab { void* p; size_t cap; hipEvent_t ev; bool pending; };
std::mutex g_aslab_mu;
std::vector<CachedSlab> g_aslabs;
size_t g_aslab_bytes = 0;
size_t aslab_cache_limit() {
    // 8 GB, at most 3 % of this process's part of the device (ranks that share a GPU: ps_set_device_fraction)
    static const double env = getenv("PORESEQ_ALIGN_CACHE_GB") ? atof(getenv("PORESEQ_ALIGN_CACHE_GB")) * 1e9 : -1.0;
    if (env >= 0) return (size_t)env;
    const size_t plan = device_plan_bytes();
    return plan ? std::min<size_t>((size_t)8e9, (size_t)(0.03 * (double)plan)) : (size_t)8e9;
}
}  // namespace

Align::~Align() {
    if (!slab) return;
    CachedSlab c{slab, slab_cap, nullptr, false};
    bool keep = slab_cap > 0 && hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) == hipSuccess;
    if (keep && last_stream) {
        if (hipEventRecord(c.ev, (hipStream_t)last_stream) == hipSuccess) c.pending = true;
        else { (void)hipGetLastError(); (void)hipEventDestroy(c.ev); keep = false; }
    }
    if (keep) {
        std::lock_guard<std::mutex> lk(g_aslab_mu);
        if (g_aslab_bytes + c.cap <= aslab_cache_limit()) { g_aslabs.push_back(c); g_aslab_bytes += c.cap; slab = nullptr; return; }
    }
    if (keep) (void)hipEventDestroy(c.ev);
    (void)hipFree(slab);
}

// a slab of at least `bytes` for an AlignData on `rt`'s stream: the smallest cached one that fits without wasting more than half of
// itself, else a fresh allocation (with an eighth of slack, so that regions of similar size find each other's slabs)
static int align_slab_take(Runtime* rt, size_t bytes, void** out, size_t* cap) {
    CachedSlab got{nullptr, 0, nullptr, false};
    {
        std::lock_guard<std::mutex> lk(g_aslab_mu);
        int best = -1;
        for (int k = 0; k < (int)g_aslabs.size(); k++)
            if (g_aslabs[k].cap >= bytes && g_aslabs[k].cap <= 2 * bytes + (1 << 20) && (best < 0 || g_aslabs[k].cap < g_aslabs[best].cap)) best = k;
        if (best >= 0) { got = g_aslabs[best]; g_aslabs[best] = g_aslabs.back(); g_aslabs.pop_back(); g_aslab_bytes -= got.cap; }
    }
    if (got.p) {
        if (got.pending) PS_HIP(hipStreamWaitEvent(rt->stream, got.ev, 0));   // (the previous owner's last work on it, if any is still queued)
        (void)hipEventDestroy(got.ev);   // (destruction is deferred by the runtime until the wait above has been honoured)
        *out = got.p; *cap = got.cap;
        return PS_OK;
    }
    const size_t want = (bytes + bytes / 8 + ((size_t)1 << 20) - 1) >> 20 << 20;
    if (hipMalloc(out, want) != hipSuccess) {
        (void)hipGetLastError();
        {   // hand the cache back and try once more
            std::lock_guard<std::mutex> lk(g_aslab_mu);
            for (CachedSlab& c : g_aslabs) { (void)hipEventDestroy(c.ev); (void)hipFree(c.p); }
            g_aslabs.clear(); g_aslab_bytes = 0;
        }
        PS_HIP(hipMalloc(out, want));
    }
    *cap = want;
    return PS_OK;
}

int Align::create(Runtime* rt, const char* seq, int64_t seq_len, int32_t n_events, const int64_t* level_off,
                  const double* mean, const double* stdv, const double* ref_align, const double* ref_like,
                  const double* model, const double* trans, const char* evseq, const int64_t* evseq_off,
                  const ps_params* params) {
    bases.assign(seq, (size_t)seq_len);
    states = states_of(bases);
    if (params) par = *params;
    E = n_events;
    n.resize(E); off.resize(E + 1);
    for (int e = 0; e <= E; e++) off[e] = E ? level_off[e] - level_off[0] : 0;
    for (int e = 0; e < E; e++) { n[e] = (int)(off[e + 1] - off[e]); if (n[e] < 0) return fail(PS_ERR_BAD_ARG, "level_off not monotone"); }
    ntot = E ? off[E] : 0;
    const int64_t base = E ? level_off[0] : 0;
    evseqs.resize(E);
    if (evseq && evseq_off) for (int e = 0; e < E; e++) evseqs[e].assign(evseq + evseq_off[e], evseq + evseq_off[e + 1]);
    h_mean.assign(mean + base, mean + base + ntot);
    h_stdv.assign(stdv + base, stdv + base + ntot);
    h_ra.assign(ref_align + base, ref_align + base + ntot);
    h_rl.assign(ref_like + base, ref_like + base + ntot);
    host_refs_valid = true;
    // derived model columns with the host libm, exactly as ModelData::setData / setParams (cpp/EventData.h:48-73)
    std::vector<double> lsd(ntot), mdl((size_t)E * 6 * NS), tr((size_t)E * 4);
    for (int64_t t = 0; t < ntot; t++) lsd[t] = std::log(h_stdv[t]);
    for (int e = 0; e < E; e++) {
        const double* src = model + (size_t)e * 4 * NS;
        double* dst = mdl.data() + (size_t)e * 6 * NS;
        for (int k = 0; k < NS; k++) {
            const double lm = src[k], ls = src[NS + k], sm = src[2 * NS + k], ss = src[3 * NS + k];
            const double lam = std::pow(sm, 3) / std::pow(ss, 2);
            dst[k] = lm; dst[NS + k] = ls; dst[2 * NS + k] = std::log(ls);
            dst[3 * NS + k] = sm; dst[4 * NS + k] = lam; dst[5 * NS + k] = std::log(lam);
        }
        for (int k = 0; k < 4; k++) tr[e * 4 + k] = std::log(trans[e * 4 + k]);
    }
    h_model = mdl;
    h_trans = tr;
    // k_fill's tables: model rows with the reciprocals of the two model divisors, level records per direction with the
    // reciprocal of the level stdv (correctly rounded: host IEEE division)
    std::vector<double> mdl8((size_t)E * (MODEL_ROW_BYTES / 8) * NS), lev((size_t)2 * 4 * std::max<int64_t>(ntot, 1));
    auto sane = [](double v) { return std::isfinite(v) && v > 1e-100 && v < 1e100; };
    fastdiv = true;
    for (int e = 0; e < E; e++) {
        const double* d6 = mdl.data() + (size_t)e * 6 * NS;
        double* d8 = mdl8.data() + (size_t)e * (MODEL_ROW_BYTES / 8) * NS;
        for (int k = 0; k < NS; k++) {
            const double lm = d6[k], ls = d6[NS + k], sm = d6[3 * NS + k], lam = d6[4 * NS + k];
            double* r8 = d8 + (size_t)k * (MODEL_ROW_BYTES / 8);
            r8[0] = lm; r8[1] = 1.0 / ls; r8[2] = ls; r8[3] = d6[2 * NS + k];
            r8[4] = sm; r8[5] = 1.0 / sm; r8[6] = lam; r8[7] = d6[5 * NS + k];
            if (!sane(ls) || !sane(sm) || !std::isfinite(lm) || !std::isfinite(lam) || std::fabs(lm) > 1e100 || std::fabs(lam) > 1e100) fastdiv = false;
        }
        const int64_t o = off[e];
        const int ne = n[e];
        for (int i = 1; i <= ne; i++) {
            double* f = lev.data() + 4 * (o + i - 1);
            double* bk = lev.data() + 4 * (ntot + o + i - 1);
            const double l3 = 3 * lsd[o + ne - i];
            f[0] = h_mean[o + i - 1]; f[1] = h_stdv[o + i - 1]; f[2] = l3; f[3] = 1.0 / h_stdv[o + i - 1];
            bk[0] = h_mean[o + ne - i]; bk[1] = h_stdv[o + ne - i]; bk[2] = l3; bk[3] = 1.0 / h_stdv[o + ne - i];
        }
    }
    for (int64_t t = 0; t < ntot; t++)
        if (!sane(h_stdv[t]) || !std::isfinite(h_mean[t]) || std::fabs(h_mean[t]) > 1e100) fastdiv = false;
    if (getenv("PORESEQ_EXACT_DIV")) fastdiv = false;   // tests: run the IEEE-division build of k_fill
    // one slab: mean, stdv, lsd, ra, rl, ri [ntot each] | model | trans | out
    const size_t nlev = (size_t)std::max<int64_t>(ntot, 1);
    const size_t bytes = (6 + 8) * nlev * sizeof(double) + (mdl.size() + mdl8.size() + tr.size() + 2) * sizeof(double) + 256 +
                         (size_t)std::max(E, 1) * sizeof(JobOut) + 64 * 16;
    PS_TRY(align_slab_take(rt, bytes, &slab, &slab_cap));
    last_stream = (void*)rt->stream;
    // the slab is filled by ONE host-to-device copy: its image is assembled in pinned staging memory first (ten copies and a memset per
    // region before: 3 000 of a bench step's copy commands)
    char* img = (char*)rt->stage.alloc(bytes);
    if (!img) return fail(PS_ERR_NOMEM, "hipHostMalloc (staging arena)");
    memset(img, 0, bytes);
    char* p = (char*)slab;
    auto carve = [&](size_t b) { char* r = p; p += (b + 63) / 64 * 64; return r; };
    auto put = [&](const void* dev, const void* src, size_t b) { if (b) memcpy(img + ((const char*)dev - (const char*)slab), src, b); };
    d_mean = (double*)carve(nlev * 8); d_stdv = (double*)carve(nlev * 8); d_lsd = (double*)carve(nlev * 8);
    d_ra = (double*)carve(nlev * 8); d_rl = (double*)carve(nlev * 8); d_ri = (double*)carve(nlev * 8);
    d_model = (double*)carve(std::max<size_t>(mdl.size(), 1) * 8); d_trans = (double*)carve(std::max<size_t>(tr.size(), 1) * 8);
    d_out = (JobOut*)carve((size_t)std::max(E, 1) * sizeof(JobOut));
    d_model8 = (double*)carve(std::max<size_t>(mdl8.size(), 1) * 8);
    d_lev[0] = (double*)carve(4 * nlev * 8); d_lev[1] = (double*)carve(4 * nlev * 8);
    if (ntot) {
        put(d_mean, h_mean.data(), ntot * 8); put(d_stdv, h_stdv.data(), ntot * 8); put(d_lsd, lsd.data(), ntot * 8);
        put(d_ra, h_ra.data(), ntot * 8); put(d_rl, h_rl.data(), ntot * 8);
        put(d_lev[0], lev.data(), 4 * ntot * 8); put(d_lev[1], lev.data() + 4 * ntot, 4 * ntot * 8);
    }
    if (E) { put(d_model, mdl.data(), mdl.size() * 8); put(d_trans, tr.data(), tr.size() * 8); put(d_model8, mdl8.data(), mdl8.size() * 8); }
    PS_HIP(hipMemcpyAsync(slab, img, (size_t)(p - (char*)slab), hipMemcpyHostToDevice, rt->stream));   // (d_out: zeros)
    PS_HIP(hipStreamSynchronize(rt->stream));
    // EventData::setData ends with updaterefs() (cpp/EventData.h:223)
    Batch b;
    PS_TRY(base_batch(rt, &b, 1, 0));
    PS_TRY(launch_updaterefs(rt, b.d));
    PS_HIP(hipStreamSynchronize(rt->stream));
    return PS_OK;
}

int Align::base_batch(Runtime* rt, Batch* b, int ndir, int lb_extra) {
    std::vector<JobSpec> specs(E);
    for (int e = 0; e < E; e++) {
        specs[e].a = this; specs[e].ev = e; specs[e].states = &states;
        specs[e].ra = d_ra + off[e]; specs[e].rl = d_rl + off[e]; specs[e].ri = d_ri + off[e];
        specs[e].out = d_out + e;
    }
    return b->build(rt, specs, ndir, lb_extra);
}

// device -> host mirror of ref_align / ref_like in two halves, so that several AlignData can share one synchronisation
int Align::refs_to_host_async(Runtime* rt) {
    pend_ra = nullptr; pend_rl = nullptr;
    if (host_refs_valid || !ntot) { host_refs_valid = true; return PS_OK; }
    last_stream = (void*)rt->stream;
    PS_TRY(rt->down(&pend_ra, d_ra, (size_t)ntot));
    PS_TRY(rt->down(&pend_rl, d_rl, (size_t)ntot));
    return PS_OK;
}
void Align::refs_finish() {   // after the stream has been synchronised
    if (pend_ra) { memcpy(h_ra.data(), pend_ra, ntot * 8); memcpy(h_rl.data(), pend_rl, ntot * 8); }
    pend_ra = nullptr; pend_rl = nullptr;
    host_refs_valid = true;
}
int Align::refs_to_host(Runtime* rt) {
    PS_TRY(refs_to_host_async(rt));
    if (pend_ra) PS_HIP(hipStreamSynchronize(rt->stream));
    refs_finish();
    return PS_OK;
}

// forward fill + backtrace + updaterefs of a batch (the body of ScoreAlignments per event,
// cpp/MakeMutations.cpp:148-195, and of Alignment::update with ndir == 2, cpp/Alignment.cpp:63-73)
static int sweep_min_default() { static const int v = getenv("PORESEQ_SWEEP_MIN") ? atoi(getenv("PORESEQ_SWEEP_MIN")) : 400; return v; }
static int sweep2_min_default() { static const int v = getenv("PORESEQ_SWEEP2_MIN") ? atoi(getenv("PORESEQ_SWEEP2_MIN")) : (1 << 30); return v; }
// column-sparse Alignment::update (ScoreMutations whose edit list reads few columns): from this many sweeps on.  A wave per sweep
// takes ~27 ms for a 10 kb event whatever the chip could do, a workgroup per sweep ~11-14 ms: a lone driver thread's small batches
// (one region: 20 sweeps) finish sooner on k_fill.  With several lock-step batches in flight the chip is shared and what counts is
// the SIMD time a sweep holds (1.6x less as a wave) and the CUs it leaves to the other batches: every size takes the sweep.
static int sparse_min_default() {
    static const int v = getenv("PORESEQ_SPARSE_MIN") ? atoi(getenv("PORESEQ_SPARSE_MIN")) : -1;
    return v >= 0 ? v : (live_runtimes() > 1 ? 0 : 160);
}
static std::atomic<int> g_sweep_min(-1), g_sweep2_min(-1), g_sparse_min(-1);
void sweep_min_set(int n) { g_sweep_min.store(n); }
void sweep2_min_set(int n) { g_sweep2_min.store(n); }
void sparse_min_set(int n) { g_sparse_min.store(n); }
static int sparse_min() { return g_sparse_min.load() >= 0 ? g_sparse_min.load() : sparse_min_default(); }
bool sweep_enabled() { static const bool off = getenv("PORESEQ_NO_SWEEP") != nullptr; return !off; }

// device bytes one forward-only job of AlignData a (n0 levels against C states) will probably take: step codes of a strip sweep,
// or the skewed {record, step word} matrix of k_fill
double fwd_job_bytes(const Align* a, int n0, int C) {
    SweepForm f;
    if (sweep_enabled()) f = sweep_guess_form(a->par.realign_width, 1);
    static const int dbg = getenv("PORESEQ_DEBUG_SWEEP_K") ? atoi(getenv("PORESEQ_DEBUG_SWEEP_K")) : 0;   // tests: a wrong guess
    if (f.ok() && dbg > 0) f.K = dbg;
    if (f.ok()) return 1.15 * sweep_job_bytes(n0, C, f);   // (the multi-wavefront forms take up to a tenth more: more steps, fewer rows per lane)
    return ((double)n0 + C + 1 + MAT_FRONT + MAT_BACK) * guess_slots(a) * 18.0;
}

// Which form a strip-sweep launch takes (ps_sweep.hip: K rows per lane on NW wavefronts per sweep).  Forced by
// ps_set_sweep_form / PORESEQ_SWEEP_FORM=K,NW (tests, tuning); else by the launch's size: a wavefront alone on its SIMD issues one
// vector instruction per ~5 cycles whatever the chip could do, so a launch that cannot fill the chip's SIMDs with one wavefront per
// sweep spreads every sweep over two or four.
static std::atomic<int> g_form_K(0), g_form_NW(0);
void sweep_form_set(int K, int NW) { g_form_K.store(K); g_form_NW.store(NW); }
static SweepForm pick_form(int W, int nsweeps, bool fastdiv) {
    SweepForm f;
    int fk = g_form_K.load(), fnw = g_form_NW.load();
    if (fk <= 0 && fnw <= 0) {                                    // (the API has precedence: the environment speaks only when neither was set)
        static const char* e = getenv("PORESEQ_SWEEP_FORM");
        if (e && sscanf(e, "%d,%d", &fk, &fnw) != 2) { fk = 0; fnw = 0; }
    }
    if (fk > 0 && sweep_form_exists(fk, fnw) && (fnw == 1 || fastdiv)) { f.K = fk; f.NW = fnw; return f; }
    if (fk <= 0 && (fnw == 1 || fnw == 2 || fnw == 4)) {         // only the wavefronts per sweep are given: the smallest strip height that fits
        for (int nw = fastdiv ? fnw : 1; nw >= 1; nw >>= 1) { f = sweep_guess_form(W, nw); if (f.ok()) return f; }
        return f;
    }
    // Two wavefronts per sweep by default: the SIMD time of one (K = 10) in half the time and three wavefronts per SIMD instead of two
    // (measured: 2 400 sweeps of 10 kb in 50 ms against 58; 20 in 13 ms against 23).  Four — a quarter more SIMD time, 11 ms — only for
    // a lone driver thread's small launches: with several lock-step batches in flight the chip is shared and SIMD time is what counts.
    static const int nw_env = getenv("PORESEQ_SWEEP_NW") ? atoi(getenv("PORESEQ_SWEEP_NW")) : 0;       // tuning: wavefronts per sweep
    static const int w4_max = getenv("PORESEQ_SWEEP_W4_MAX") ? atoi(getenv("PORESEQ_SWEEP_W4_MAX")) : 256;   // sweeps per launch up to which four wavefronts each pay
    int nw = nw_env > 0 ? nw_env : (live_runtimes() <= 1 && nsweeps <= w4_max ? 4 : 2);
    if (!fastdiv) nw = 1;                                         // (the multi-wavefront builds exist with tabulated reciprocals only)
    for (; nw >= 1; nw >>= 1) {
        f = sweep_guess_form(W, nw);
        // a narrow band on many wavefronts leaves most lanes without a strip: at least half of them busy, else fewer wavefronts
        if (f.ok() && (nw == 1 || ((2 * W + 1) / (f.K + 1) + 3) * 2 >= 64 * nw)) return f;
    }
    return sweep_guess_form(W, 1);
}

// Strip sweeps (ps_sweep.hip): one to four wavefronts per alignment and direction.  Forward-only batches (ScoreAlignments) keep one
// byte per cell; Alignment::update batches (ndir == 2, ScoreMutations) also the {main, stay} records of both directions, in strip
// order or of the kept columns only.  Returns -1 when the batch has to take the k_fill path instead (band too wide for any form).
static int realign_sweep(Runtime* rt, Batch& b, double cap) {
    int W = 0;
    for (const JobD& j : b.jobs) W = std::max(W, j.W);
    SweepForm f = pick_form(W, b.d.njobs * b.ndir, b.d.fastdiv != 0);
    if (const char* e = getenv("PORESEQ_DEBUG_SWEEP_K")) { f.K = atoi(e); f.NW = 1; }   // tests: a given strip height first
    if (!f.ok()) return -1;
    PS_TRY(launch_begin(rt, b.d));
    PS_TRY(launch_lb(rt, b.d, 0, b.maxlbn));
    for (;;) {
        PS_TRY(sweep_prepare(rt, b, f));
        int* w = nullptr;
        PS_TRY(rt->down(&w, b.sd.maxwin, (size_t)1));
        PS_HIP(hipStreamSynchronize(rt->stream));
        { static const bool trace = getenv("PORESEQ_TRACE") != nullptr; if (trace) fprintf(stderr, "[ps] realign (strip sweep%s): %d jobs x %d, K = %d on %d wavefronts, widest window %d strips\n", b.sparse ? ", kept columns" : "", b.d.njobs, b.ndir, f.K, f.NW, *w); }
        if (*w <= sweep_win_max(f.NW)) break;
        f = sweep_next_form(f, *w);
        if (!f.ok()) { for (JobD& j : b.jobs) j.K = 0; return -1; }
    }
    const int K = f.K;
    const double bytes = (double)b.sweep_code_bytes + 16.0 * (double)b.sweep_recs;
    if (cap > 0 && bytes > cap) {
        static const bool trace = getenv("PORESEQ_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[ps] realign (strip sweep): %.2f GB of step codes%s at K = %d, over the share: split\n", bytes * 1e-9, b.ndir == 2 ? (b.sparse ? " and kept columns" : " and records") : "", K);
        b.P = 0;
        return PS_SPLIT;
    }
    {
        // forward-only: the step codes live in the pool of the score matrices (a runtime runs one batch at a time: never both);
        // with both directions the records take that pool and the codes the step words'
        const size_t need_rec = b.ndir == 2 ? (size_t)std::max<int64_t>(b.sweep_recs, 1) * sizeof(double2) : (size_t)std::max<int64_t>(b.sweep_code_bytes, 1);
        const size_t need_flg = b.ndir == 2 ? (size_t)std::max<int64_t>(b.sweep_code_bytes, 1) : 0;
        void *prec = nullptr, *pflg = nullptr;
        Batch own;   // (kept columns and step codes are small: always the runtime's own pools)
        const int rc = ensure_matrix_pools(rt, b.sparse || b.ndir == 1 ? own : b, need_rec, need_flg, cap > 0, &prec, &pflg);
        if (rc == PS_ERR_NOMEM && cap > 0) { b.P = 0; return PS_SPLIT; }
        PS_TRY(rc);
        b.sd.codes = b.ndir == 2 ? (unsigned char*)pflg : (unsigned char*)prec;
        if (b.ndir == 2) {
            b.d.rec = (double2*)prec; b.d.flg = nullptr;
            b.d.s_sj = b.sd.sj; b.d.s_band = b.sd.band; b.d.s_qlo = b.sd.qlo; b.d.s_qhi = b.sd.qhi;
            PS_TRY(rt->up(rt->buf("jobs").p, b.jobs.data(), b.jobs.size() * sizeof(JobD)));   // JobD.K, JobD.mat_off
            PS_HIP(hipMemsetAsync(b.d.cmax, 0, b.ncols * sizeof(double), rt->stream));
        }
    }
    if (rt->prof_on) { rt->prof["sweep"].bytes += b.fill_alg_bytes(); rt->prof["sweep"].units += (double)b.d.njobs * b.ndir; }
    PS_TRY(sweep_run(rt, b));
    PS_TRY(launch_updaterefs(rt, b.d));
    return PS_OK;
}

int realign(Runtime* rt, Batch& b, double cap) {
    if (!b.d.njobs) return PS_OK;
    // forward-only batches take the strip sweep (one wave per alignment) from ps_set_sweep_min / PORESEQ_SWEEP_MIN alignments on
    // (default 400); a smaller batch alone on the chip finishes sooner with a workgroup per alignment (k_fill: ~11 ms against
    // ~25 ms for a 10 kb sweep; with several batches in flight the two take the same time)
    if (b.sparse && !(sweep_enabled() && b.d.njobs * 2 >= sparse_min())) b.sparse = false;
    const int sweep_min = b.ndir == 1 ? (g_sweep_min.load() >= 0 ? g_sweep_min.load() : sweep_min_default())
                                      : (b.sparse ? 0 : (g_sweep2_min.load() >= 0 ? g_sweep2_min.load() : sweep2_min_default()));
    // (a forward-only batch below the threshold whose skewed matrices would not fit this runtime's share — a third of FindMutations'
    //  candidate batches with many batches in flight — takes the sweep as well: 7 MB of step codes per alignment instead of 140 MB)
    bool too_big = false;
    if (b.ndir == 1 && sweep_enabled() && b.d.njobs < sweep_min) {
        double est = 0;
        for (const JobD& j : b.jobs) est += (double)(j.S + MAT_FRONT + MAT_BACK) * std::min(1024, std::max(64, (((2 * j.W + 1) * 10 / 19 + 9 + 63) / 64) * 64)) * 18.0;
        too_big = est > device_share_bytes();
    }
    if (sweep_enabled() && (b.d.njobs * b.ndir >= sweep_min || too_big)) {
        const int rc = realign_sweep(rt, b, cap);
        if (rc != -1) return rc;
    }
    b.sparse = false;   // (a band too wide for any strip height: skewed matrices)
    PS_TRY(launch_begin(rt, b.d));
    PS_TRY(launch_lb(rt, b.d, 0, b.maxlbn));
    PS_TRY(launch_lo(rt, b.d, b.ndir, b.maxS));
    int* w = nullptr;
    PS_TRY(rt->down(&w, b.d.maxw, (size_t)1));
    PS_HIP(hipStreamSynchronize(rt->stream));
    // nine slots more than the widest footprint: a lane idles at least nine anti-diagonals between two rows, so a prefetch
    // window of k_fill (fetched six steps ahead, four steps long) never spans two rows of a lane that has a cell
    // (a footprint beyond 1015 rows takes k_fill_wide: two slots per thread, up to 2048 slots, P a multiple of 128)
    if (std::max(*w, 1) + 2 > 2048)
        return fail(PS_ERR_UNSUPPORTED, "band footprint of " + std::to_string(*w) + " rows on one anti-diagonal: wider than two slots per lane of one "
                                        "workgroup (2046); realign_width up to 1022 fits for any input");
    { static const bool trace = getenv("PORESEQ_TRACE") != nullptr; if (trace) fprintf(stderr, "[ps] realign: %d jobs x %d, widest footprint %d\n", b.d.njobs, b.ndir, *w); }
    const int Pneed = std::max(*w, 1) + 9 <= 1024 ? std::max(*w, 1) + 9 : ((std::max(*w, 1) + 2 + 127) / 128) * 128;
    if (cap > 0) {   // the caller sized this batch on a guess of the footprint: let it split when the real one is much wider
        const int Pr = std::min(b.Pmax, std::max(64, ((Pneed + 63) / 64) * 64));
        double bytes = 0;
        for (const JobD& j : b.jobs) bytes += (double)(j.S + MAT_FRONT + MAT_BACK) * Pr * 18.0 * b.ndir;
        if (bytes > cap) {
            static const bool trace2 = getenv("PORESEQ_TRACE") != nullptr;
            if (trace2) fprintf(stderr, "[ps] realign: %.1f GB of matrices at %d slots per anti-diagonal, over the share: split\n", bytes * 1e-9, Pr);
            b.P = Pr;
            return PS_SPLIT;
        }
    }
    {
        const int rc = b.place(rt, Pneed, cap > 0);
        // several runtimes share the device and sized their pools at different times: when the matrices cannot be had even after the
        // idle pools were taken back (DBuf::ensure), a caller that can split does so instead of failing
        if (rc == PS_ERR_NOMEM && cap > 0) { b.P = std::min(b.Pmax, std::max(64, ((Pneed + 63) / 64) * 64)); return PS_SPLIT; }
        PS_TRY(rc);
    }
    if (rt->prof_on) { rt->prof["fill"].bytes += b.fill_alg_bytes(); rt->prof["fill"].units += (double)b.d.njobs * b.ndir; }
    PS_TRY(launch_fill(rt, b.d, b.jobs, b.ndir, b.maxS, b.P, b.ncols));
    PS_TRY(launch_backtrace(rt, b.d, b.maxn));
    PS_TRY(launch_updaterefs(rt, b.d));
    return PS_OK;
}

// run fn(k) for k in [0, n) on the calling thread plus helpers from a process-wide pool of host threads (disjoint outputs; the GPU
// work of a batched call is enqueued by the caller).  The pool's threads live for the process: a lock-step schedule makes ~500 such
// calls per batch, fourteen batches at once — creating up to 32 threads for each of them cost more than most of the loops.  Helpers
// per call: PORESEQ_HOST_THREADS (poreseq_amd.dist.init sets it to this rank's share of the node's cores when several ranks share a
// node), else up to 32; the pool holds twice that for callers that overlap.  A helper that is dequeued after the caller and the
// other helpers have taken every index finds nothing to do and never touches the caller's frame.
namespace {
struct ParJob {
    std::function<void(int)> fn;
    int n = 0;
    std::atomic<int> next{0}, done{0};
    std::mutex mu;
    std::condition_variable cv;
    void run() {
        int did = 0;
        for (int k = next++; k < n; k = next++) { fn(k); did++; }
        if (did && (done += did) >= n) { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
    }
};
struct ParPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<ParJob>> q;
    std::vector<std::thread> th;
    int idle = 0;
    size_t cap = 64;
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            idle++;
            cv.wait(lk, [&] { return !q.empty(); });
            idle--;
            std::shared_ptr<ParJob> j = q.front();
            q.pop_front();
            lk.unlock();
            j->run();
            j.reset();
            lk.lock();
        }
    }
    void submit(const std::shared_ptr<ParJob>& j, int helpers) {
        std::lock_guard<std::mutex> lk(mu);
        for (int k = 0; k < helpers; k++) q.push_back(j);
        int need = (int)q.size() - idle;   // queued tasks no waiting worker will take: new workers, up to the pool's size
        for (; need > 0 && th.size() < cap; need--) { th.emplace_back([this] { worker(); }); th.back().detach(); }
        cv.notify_all();
    }
};
static int par_cap() { static const int cap = [] { const char* e = getenv("PORESEQ_HOST_THREADS"); const int v = e ? atoi(e) : 32; return std::max(1, std::min(v, 64)); }(); return cap; }
ParPool* par_pool() {   // (never destroyed: its threads are detached and may outlive main; its size is set once, here)
    static ParPool* p = [] { ParPool* q = new ParPool(); q->cap = (size_t)std::max(2 * par_cap(), 8); return q; }();
    return p;
}
}  // namespace

void par_for(int n, const std::function<void(int)>& fn) {
    if (n <= 1) { if (n == 1) fn(0); return; }
    const int cap = par_cap();
    const int nth = std::min(n, cap);
    if (nth <= 1) { for (int k = 0; k < n; k++) fn(k); return; }
    std::shared_ptr<ParJob> j = std::make_shared<ParJob>();
    j->fn = fn; j->n = n;
    par_pool()->submit(j, nth - 1);
    j->run();
    std::unique_lock<std::mutex> lk(j->mu);
    j->cv.wait(lk, [&] { return j->done.load() >= n; });
}

// Slots per anti-diagonal realign() will probably need for this AlignData: footprint ~ (2W + 1) / 1.9 for about one level per base, + 9.
// Only a guess (ragged remapped alignments, few levels per base: up to 2W + 1): callers that size batches on it pass realign() a cap
// and split when it answers PS_SPLIT.  PORESEQ_DEBUG_GUESS_P overrides it (tests: a wrong guess).
int guess_slots(const Align* a) {
    static const int dbg = getenv("PORESEQ_DEBUG_GUESS_P") ? atoi(getenv("PORESEQ_DEBUG_GUESS_P")) : 0;
    const int g = dbg > 0 ? dbg : (((2 * a->par.realign_width + 1) * 10 / 19 + 9 + 63) / 64) * 64;
    return std::min(1024, std::max(64, g));
}

// Where the sub-batch of AlignData that starts at as[k0] ends when each event takes `ndir` sweeps: everything if it fits this
// runtime's device share; otherwise the batch is cut into the fewest sub-batches that fit, of about equal size (a remainder of
// two regions behind two full sub-batches would cost a whole launch's latency for a tenth of the work)
static size_t fit_share(const std::vector<Align*>& as, size_t k0, int ndir) {
    const double cap = ndir == 2 ? dense_cap_bytes() : device_share_bytes();   // (full forward + backward matrices live in a slab)
    auto need = [&](size_t k) {
        const Align* a = as[k];
        const int P = guess_slots(a);
        double add = 0;
        for (int e = 0; e < a->E; e++)
            add += ndir == 1 ? fwd_job_bytes(a, a->n[e], (int)a->states.size()) : ((double)a->n[e] + a->states.size() + 1 + MAT_FRONT + MAT_BACK) * P * 18.0 * ndir;
        return add;
    };
    double total = 0;
    for (size_t k = k0; k < as.size(); k++) total += need(k);
    if (total <= cap) return as.size();
    const double target = total / std::ceil(total / cap);      // bytes per sub-batch, all about equal
    double bytes = 0;
    size_t k = k0;
    for (; k < as.size(); k++) {
        const double add = need(k);
        if (k > k0 && (bytes + add > cap || bytes + 0.5 * add > target)) break;
        bytes += add;
    }
    return k;
}

// ScoreAlignments, cpp/MakeMutations.cpp:148-195, for several AlignData in one launch chain (independent regions in lock-step)
int score_alignments_multi(Runtime* rt, const std::vector<Align*>& as, const std::vector<double*>& scores, const std::vector<double*>& likes) {
    if (fit_share(as, 0, 1) < as.size()) {   // more matrices than this runtime's share of the device: sub-batches, one after the other
        for (size_t k0 = 0; k0 < as.size();) {
            const size_t k1 = fit_share(as, k0, 1);
            PS_TRY(score_alignments_multi(rt, std::vector<Align*>(as.begin() + k0, as.begin() + k1), std::vector<double*>(scores.begin() + k0, scores.begin() + k1),
                                          std::vector<double*>(likes.begin() + k0, likes.begin() + k1)));
            k0 = k1;
        }
        return PS_OK;
    }
    std::vector<JobSpec> specs;
    for (Align* a : as)
        for (int e = 0; e < a->E; e++) {
            JobSpec s;
            s.a = a; s.ev = e; s.states = &a->states;
            s.ra = a->d_ra + a->off[e]; s.rl = a->d_rl + a->off[e]; s.ri = a->d_ri + a->off[e]; s.out = a->d_out + e;
            specs.push_back(s);
        }
    if (specs.empty()) return PS_OK;
    Batch b;
    PS_TRY(b.build(rt, specs, 1, 0));
    {
        const int rc = realign(rt, b, as.size() > 1 ? 1.2 * device_share_bytes() : 0.0);
        if (rc == PS_SPLIT) {   // bands wider than fit_share guessed: two halves, one after the other
            const size_t h = as.size() / 2;
            PS_TRY(score_alignments_multi(rt, std::vector<Align*>(as.begin(), as.begin() + h), std::vector<double*>(scores.begin(), scores.begin() + h),
                                          std::vector<double*>(likes.begin(), likes.begin() + h)));
            return score_alignments_multi(rt, std::vector<Align*>(as.begin() + h, as.end()), std::vector<double*>(scores.begin() + h, scores.end()),
                                          std::vector<double*>(likes.begin() + h, likes.end()));
        }
        PS_TRY(rc);
    }
    // the jobs' scores, gathered into one array on the device: one copy back instead of one per AlignData
    for (Align* a : as) a->host_refs_valid = false;
    double* best = nullptr;
    {
        DBuf& gb = rt->buf("best");
        PS_TRY(gb.ensure(specs.size() * sizeof(double)));
        PS_TRY(launch_gather_best(rt, b.d, gb.as<double>()));
        PS_TRY(rt->down(&best, gb.p, specs.size()));
    }
    bool any_likes = false;
    for (size_t k = 0; k < as.size(); k++) if (likes[k]) { any_likes = true; PS_TRY(as[k]->refs_to_host_async(rt)); }
    PS_HIP(hipStreamSynchronize(rt->stream));
    {
        size_t j = 0;
        for (size_t k = 0; k < as.size(); k++)
            for (int e = 0; e < as[k]->E; e++) scores[k][e] = std::max(best[j++], 0.0);  // Alignment::getMax, cpp/Alignment.h:127-130
    }
    if (any_likes)
        par_for((int)as.size(), [&](int k) {
            Align* a = as[k];
            if (!likes[k]) return;
            a->refs_finish();
            for (int e = 0; e < a->E; e++) accumulate_likes(a->h_ra.data() + a->off[e], a->h_rl.data() + a->off[e], a->n[e], (int)a->states.size(), likes[k]);
        });
    return PS_OK;
}

int score_alignments(Runtime* rt, Align* a, double* scores, double* likes) {
    if (!a->E) return PS_OK;
    return score_alignments_multi(rt, {a}, {scores}, {likes});
}

// the `likes` loop of ScoreAlignments, cpp/MakeMutations.cpp:168-189 (one event)
void accumulate_likes(const double* ra, const double* rl, int n, int C, double* likes) {
    double last = 0;
    int refind = 1;
    for (int t = 0; t < n; t++) {
        if (ra[t] > 0) {
            for (int k = refind; k < ra[t]; k++) likes[k + 1] += last;
            last = rl[t];
            refind = (int)ra[t];
        }
    }
    for (int64_t k = refind; k < (int64_t)C + 3; k++) likes[k + 1] += last;
}

// states of the edited sequence at columns sidx+1 .. sidx+ncol, without building the whole sequence;
// equals Sequence(original, mut).states there (cpp/Sequence.h:37-100).  A 12-base look-back flushes
// every effect of earlier non-ACGT characters (they reach at most 8 states ahead).
static void edited_window(const std::string& b, const Mut& m, int sidx, int ncol, int* out) {
    const bool copy = (size_t)m.start >= b.size();
    const int64_t L = (int64_t)b.size();
    const int64_t cut = copy ? L : std::min<int64_t>(L, (int64_t)m.start + (int64_t)m.orig.size());
    const int64_t mlen = copy ? 0 : (int64_t)m.mut.size();
    const int64_t Lm = copy ? L : (int64_t)m.start + mlen + (L - cut);
    auto at = [&](int64_t p) -> char {
        if (copy || p < m.start) return b[p];
        if (p < m.start + mlen) return m.mut[p - m.start];
        return b[cut + (p - m.start - mlen)];
    };
    const int64_t lo = std::max<int64_t>(0, (int64_t)sidx - 12);
    const int64_t hi = std::min<int64_t>(Lm, (int64_t)sidx + ncol + 4);
    std::string w;
    w.reserve(hi - lo);
    for (int64_t p = lo; p < hi; p++) w.push_back(at(p));
    std::vector<int> st = states_of(w);
    for (int c = 0; c < ncol; c++) {
        const int64_t k = (int64_t)sidx + c - lo;
        out[c] = (k >= 0 && k < (int64_t)st.size()) ? st[k] : -1;
    }
}

namespace {
// host-side description of one AlignData's edit list for k_old / k_score
struct EditPlan {
    int M = 0, ncolmax = 1, extra = 0, nr0 = 0;
    std::vector<int> start, mlen, cm, ncol, skip, oldidx, states, r0s, cls[SCORE_CLASSES];
    std::vector<int> keep[2];   // per direction: kept-column index of column 0 .. C + 1, or -1 (plan_keep)
    int nkeep[2] = {0, 0};
    int rc = PS_OK;
};
}  // namespace

static void plan_edits(const Align* a, const std::vector<Mut>& muts, EditPlan* p) {
    const int M = (int)muts.size();
    p->M = M;
    for (const Mut& m : muts) if (m.start < 0) { p->rc = PS_ERR_BAD_ARG; return; }
    const int64_t L = (int64_t)a->bases.size();
    const int C = (int)a->states.size();
    const int WS = a->par.scoring_width;
    p->start.resize(M); p->mlen.resize(M); p->cm.resize(M); p->ncol.resize(M); p->skip.resize(M); p->oldidx.resize(M);
    int ncolmax = 1, extra = 0;
    for (int i = 0; i < M; i++) {
        const Mut& m = muts[i];
        p->start[i] = m.start; p->mlen[i] = (int)m.mut.size();
        p->skip[i] = (int64_t)m.start > L ? 1 : 0;  // "sanity check", cpp/MakeMutations.cpp:46-47
        const bool copy = (int64_t)m.start >= L;
        const int64_t cut = std::min<int64_t>(L, (int64_t)m.start + (int64_t)m.orig.size());
        const int64_t Lm = copy ? L : (int64_t)m.start + (int64_t)m.mut.size() + (L - cut);
        const int Cm = Lm >= 5 ? (int)(Lm - 4) : 0;
        p->cm[i] = Cm;
        const int sidx = std::max(m.start - 4, 0);
        int ncol = std::min<int64_t>((int64_t)m.mut.size() + 6, std::max<int64_t>(0, (int64_t)Cm - sidx));
        if (WS == 0 || p->skip[i]) ncol = 0;  // stripe_width 0 makes fillColumn a no-op, cpp/Alignment.cpp:118-119
        p->ncol[i] = ncol;
        ncolmax = std::max(ncolmax, ncol);
        if (!p->skip[i]) extra = std::max(extra, sidx + ncol + 1 - (C + 1));
    }
    p->ncolmax = ncolmax;
    p->extra = std::max(extra, 0) + 2;
}

// The matrix columns the scoring of this list will read (k_old / k_score, cpp/Alignment.cpp:447-512, cpp/Alignment.h:181-214):
// per edit the forward column it is spliced behind, max(start - 4, 0), the column pair of its old score, max(start - 3, 1) forward
// and C - that + 1 backward, and the backward column its target is combined with — with the kernels' own clamping.  Column 0 (the
// blank column) is never read from the records.
static void plan_keep(const Align* a, EditPlan* p) {
    const int C = (int)a->states.size();
    for (int d = 0; d < 2; d++) { p->keep[d].assign((size_t)C + 2, -1); p->nkeep[d] = 0; }
    auto clampc = [&](int c) { return (unsigned)c >= (unsigned)(C + 1) ? C : c; };
    for (int i = 0; i < p->M; i++) {
        if (p->skip[i]) continue;
        const int start = p->start[i];
        const int sidx = std::max(start - 4, 0);
        const int tcol = std::min(start + p->mlen[i] + 1, sidx + p->ncol[i]);
        const int backind = clampc(p->cm[i] - tcol + 1);
        const int r0 = std::max(start - 3, 1);
        const int raf = clampc(r0), rab = clampc(C - r0 + 1);
        const int sf = clampc(sidx);
        if (sf > 0) p->keep[0][sf] = 0;
        if (raf > 0) p->keep[0][raf] = 0;
        if (backind > 0) p->keep[1][backind] = 0;
        if (rab > 0) p->keep[1][rab] = 0;
    }
    for (int d = 0; d < 2; d++)
        for (int c = 0; c <= C + 1; c++) if (p->keep[d][c] == 0) p->keep[d][c] = p->nkeep[d]++;
}

// second half of the plan (runs on host threads while the GPU realigns): edited states, distinct r0, size classes
static void plan_tables(const Align* a, const std::vector<Mut>& muts, EditPlan* p, int nth) {
    const int M = p->M, ncolmax = p->ncolmax;
    const int64_t L = (int64_t)a->bases.size();
    p->states.assign((size_t)M * ncolmax, -1);
    {
        auto work = [&](int lo, int hi) {
            for (int i = lo; i < hi; i++)
                if (p->ncol[i] > 0) edited_window(a->bases, muts[i], std::max(muts[i].start - 4, 0), p->ncol[i], p->states.data() + (size_t)i * ncolmax);
        };
        if (M < 4096) nth = 1;   // Refine-sized lists: split over a few host threads (disjoint outputs)
        std::vector<std::thread> th;
        for (int t = 1; t < nth; t++) th.emplace_back(work, (int)((int64_t)M * t / nth), (int)((int64_t)M * (t + 1) / nth));
        work(0, (int)((int64_t)M / nth));
        for (std::thread& x : th) x.join();
    }
    std::vector<int> idx((size_t)std::max<int64_t>(L, 4) + 2, -1);   // r0 <= L - 3 for every edit that is not skipped
    for (int i = 0; i < M; i++) {
        if (p->skip[i]) { p->oldidx[i] = 0; continue; }
        const int r0 = std::max(muts[i].start - 3, 1);
        int& at = idx[r0];
        if (at < 0) { at = (int)p->r0s.size(); p->r0s.push_back(r0); }
        p->oldidx[i] = at;
    }
    p->nr0 = (int)p->r0s.size();
    static const int lim7 = getenv("PORESEQ_NO_SCORE7") ? -1 : 7;   // (tuning / A-B: point edits on the 8-lane class as before round 5)
    for (int i = 0; i < M; i++) {
        const int nc = p->ncol[i];
        p->cls[nc <= lim7 ? 4 : nc <= 8 ? 0 : nc <= 16 ? 1 : nc <= 32 ? 2 : 3].push_back(i);
    }
}

// ScoreMutations, cpp/MakeMutations.cpp:23-69, for several AlignData at once: one realign launch chain over all their
// events (forward + backward of one event share a workgroup), then the edit scoring of each
static int score_mutations_planned(Runtime* rt, const std::vector<Align*>& as, const std::vector<const std::vector<Mut>*>& muts,
                                   const std::vector<std::vector<Mut>*>& outs, const std::vector<double*>* delta_out, std::vector<EditPlan>& plan);

int score_mutations_multi(Runtime* rt, const std::vector<Align*>& as, const std::vector<const std::vector<Mut>*>& muts,
                          const std::vector<std::vector<Mut>*>& outs, const std::vector<double*>* delta_out) {
    const int R = (int)as.size();
    std::vector<EditPlan> plan(R);
    par_for(R, [&](int k) {   // (a Refine list is 80 000 edits per region: copied and sized side by side)
        *outs[k] = *muts[k];
        for (Mut& m : *outs[k]) m.score = -1e-6;
        plan_edits(as[k], *muts[k], &plan[k]);
    });
    for (int k = 0; k < R; k++) {
        if (as[k]->par.scoring_width < 0) return fail(PS_ERR_BAD_ARG, "scoring_width < 0");
        if (plan[k].rc != PS_OK) return fail(plan[k].rc, "negative mutation start");
        if (plan[k].ncolmax > 64 && as[k]->par.scoring_width > 511) return fail(PS_ERR_UNSUPPORTED, "edit longer than 58 bases with scoring_width > 511");
    }
    return score_mutations_planned(rt, as, muts, outs, delta_out, plan);
}

// `plan`: every list sized (plan_edits) and `outs` initialised — once per call: the sub-batches of a call that does not fit a slab or the
// runtime's share, and the halves of one whose bands came out wider than guessed, take their regions' plans with them (moved: the
// caller returns right behind them) instead of copying and sizing 80 000 edits per region again
static int score_mutations_planned(Runtime* rt, const std::vector<Align*>& as, const std::vector<const std::vector<Mut>*>& muts,
                                   const std::vector<std::vector<Mut>*>& outs, const std::vector<double*>* delta_out, std::vector<EditPlan>& plan) {
    Tick tk("score_mutations");
    const int R = (int)as.size();
    // the reference's progress line under `verbose` (cpp/MakeMutations.cpp:28-32, 55-66: "Scoring (<width>)", a dot per event, a newline);
    // a lock-step call over several AlignData has no single line to write: only the single-handle call speaks
    if (R == 1 && as[0]->par.verbose) {
        fprintf(stderr, "Scoring (%d)", (int)as[0]->par.scoring_width);
        for (int e = 0; e < as[0]->E; e++) fputc('.', stderr);
        fputc('\n', stderr);
        fflush(stderr);
    }
    {   // experiment: how much of a bench step is the latency of a chain?  PORESEQ_DEBUG_SLEEP_US of host sleep per ScoreMutations call
        static const int us = getenv("PORESEQ_DEBUG_SLEEP_US") ? atoi(getenv("PORESEQ_DEBUG_SLEEP_US")) : 0;
        if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us));
    }
    // Which columns of the score matrices will the edit lists read?  A short list (FindMutations' found edits, the rounds of
    // MakeMutations' recursion: tens to hundreds of edits per region) reads a few percent of them: the fills then run as strip
    // sweeps that keep those columns only (ps_sweep.hip, k_sweeps) — a few MB per alignment instead of 2 x 110 MB.  A list that
    // touches more than a quarter of the columns (Refine's point edits at every position) takes full matrices.
    static const double sparse_frac = getenv("PORESEQ_SPARSE_FRAC") ? atof(getenv("PORESEQ_SPARSE_FRAC")) : 0.25;
    bool sparse = sweep_enabled();
    int njobs_all = 0;
    double sparse_bytes = 0;
    for (int k = 0; k < R && sparse; k++) {
        const Align* a = as[k];
        const int C = (int)a->states.size();
        if (plan[k].keep[0].empty()) plan_keep(a, &plan[k]);   // (C + 2 entries once computed)
        if (std::max(plan[k].nkeep[0], plan[k].nkeep[1]) > sparse_frac * C) sparse = false;
        const int K = sweep_guess_k(a->par.realign_width);
        if (!K) sparse = false;
        njobs_all += a->E;
        for (int e = 0; e < a->E && sparse; e++)
            sparse_bytes += 1.15 * sweep_job_bytes(a->n[e], C, sweep_guess_form(a->par.realign_width, 1)) + 16.0 * (plan[k].nkeep[0] + plan[k].nkeep[1]) * (std::min(2 * a->par.realign_width + 1, a->n[e]) + 24);
    }
    if (sparse && 2 * njobs_all < sparse_min()) sparse = false;
    if (tk.on) {
        double fr = 0; size_t M = 0;
        for (int k = 0; k < R; k++) { M += plan[k].M; if (!plan[k].keep[0].empty()) fr = std::max(fr, (double)std::max(plan[k].nkeep[0], plan[k].nkeep[1]) / std::max<size_t>(as[k]->states.size(), 1)); }
        fprintf(stderr, "[ps] score_mutations: %d regions, %d events, %zu edits, kept columns <= %.3f of a matrix: %s\n", R, njobs_all, M, fr, sparse ? "kept columns" : "full matrices");
    }
    tk.lap("edit sizes");
    auto sub = [&](size_t k0, size_t k1) {   // regions k0 .. k1 - 1 as a call of their own, with their plans
        std::vector<double*> dsub;
        if (delta_out) dsub.assign(delta_out->begin() + k0, delta_out->begin() + k1);
        std::vector<EditPlan> psub(std::make_move_iterator(plan.begin() + k0), std::make_move_iterator(plan.begin() + k1));
        return score_mutations_planned(rt, std::vector<Align*>(as.begin() + k0, as.begin() + k1),
                                       std::vector<const std::vector<Mut>*>(muts.begin() + k0, muts.begin() + k1),
                                       std::vector<std::vector<Mut>*>(outs.begin() + k0, outs.begin() + k1), delta_out ? &dsub : nullptr, psub);
    };
    auto halves = [&]() {
        const size_t h = as.size() / 2;
        PS_TRY(sub(0, h));
        return sub(h, as.size());
    };
    if (sparse && R > 1 && sparse_bytes > device_share_bytes()) return halves();
    if (!sparse && fit_share(as, 0, 2) < as.size()) {   // sub-batches that fit this runtime's share of the device
        for (size_t k0 = 0; k0 < as.size();) {
            const size_t k1 = fit_share(as, k0, 2);
            PS_TRY(sub(k0, k1));
            k0 = k1;
        }
        return PS_OK;
    }
    // the kept-column tables of all AlignData in one block (before the jobs are built: their descriptors point into it)
    std::vector<const int*> d_keep(2 * (size_t)R, nullptr);
    if (sparse) {
        size_t tot = 0;
        for (int k = 0; k < R; k++) tot += plan[k].keep[0].size() + plan[k].keep[1].size();
        DBuf& kb = rt->buf("keep");
        PS_TRY(kb.ensure(std::max<size_t>(tot, 1) * sizeof(int)));
        std::vector<int> hk;
        hk.reserve(tot);
        for (int k = 0; k < R; k++)
            for (int d = 0; d < 2; d++) { d_keep[2 * k + d] = kb.as<int>() + hk.size(); hk.insert(hk.end(), plan[k].keep[d].begin(), plan[k].keep[d].end()); }
        PS_TRY(rt->up(kb.p, hk.data(), hk.size() * sizeof(int)));
    }
    // Alignment::update for every event of every AlignData: enqueued now, so that the fills run while the host prepares the edit tables
    std::vector<JobSpec> specs;
    std::vector<int> job0(R, 0);
    int extra = 0;
    for (int k = 0; k < R; k++) {
        Align* a = as[k];
        job0[k] = (int)specs.size();
        extra = std::max(extra, plan[k].extra);
        for (int e = 0; e < a->E; e++) {
            JobSpec s;
            s.a = a; s.ev = e; s.states = &a->states;
            s.ra = a->d_ra + a->off[e]; s.rl = a->d_rl + a->off[e]; s.ri = a->d_ri + a->off[e]; s.out = a->d_out + e;
            if (sparse) for (int d = 0; d < 2; d++) { s.keep[d] = d_keep[2 * k + d]; s.nkeep[d] = plan[k].nkeep[d]; }
            specs.push_back(s);
        }
    }
    if (specs.empty()) return PS_OK;
    Batch b;
    SlabHold slab;   // full matrices: one of the process's slabs for the duration of this call (released at every return)
    double lone_need = 0;   // a single AlignData cannot be split: matrices beyond a slab go to the runtime's own pools
    if (!sparse && R == 1) for (int e = 0; e < as[0]->E; e++) lone_need += ((double)as[0]->n[e] + as[0]->states.size() + 1 + MAT_FRONT + MAT_BACK) * std::min(1024, 2 * as[0]->par.realign_width + 74) * 36.0;
    if (!sparse && lone_need <= (double)slab_bytes()) {
        PS_TRY(slab_acquire(&slab));
        if (R == 1 && lone_need > (double)slab.bytes) {
            // the slab this call was handed is smaller than the plan (a device fuller than expected: slab_acquire's fallback sizes) and a
            // single AlignData cannot be split: its matrices go to the runtime's own pools, as those beyond a planned slab do
            slab.release();
        } else {
            if (rt->prof_on) rt->prof["slab"].launches++;   // (dense calls that took a slab: a host-side count)
            slab.drain = rt->stream;
            b.ext = slab.p; b.ext_bytes = R > 1 ? std::min(slab.bytes, (size_t)dense_cap_bytes()) : slab.bytes;
        }
        tk.lap("slab wait");
    }
    PS_TRY(b.build(rt, specs, 2, extra));
    {
        const int rc = realign(rt, b, R > 1 ? (sparse ? 1.2 * device_share_bytes() : (double)b.ext_bytes) : 0.0);
        if (rc == PS_SPLIT) { slab.release(); return halves(); }   // bands wider than guessed: two halves, one after the other
        PS_TRY(rc);
    }
    for (Align* a : as) a->host_refs_valid = false;
    tk.lap("realign enqueue");
    par_for(R, [&](int k) { plan_tables(as[k], *muts[k], &plan[k], R == 1 ? 8 : (R <= 4 ? 4 : 1)); });
    tk.lap("edit geometry");
    // upload the edit tables of all AlignData in one block
    size_t ints = 16, dbls = 1;
    for (int k = 0; k < R; k++) {
        const EditPlan& p = plan[k];
        ints += (size_t)p.M * 7 + (size_t)p.M * p.ncolmax + p.nr0 + 16;
        dbls += (size_t)as[k]->E * std::max(p.nr0, 1) + (size_t)as[k]->E * std::max(p.M, 1) + std::max(p.M, 1) + (size_t)as[k]->E * (as[k]->states.size() + 8);
    }
    DBuf& mb = rt->buf("mutint");
    PS_TRY(mb.ensure(ints * sizeof(int) + 64 + (size_t)R * sizeof(ScoreArgs)));
    DBuf& db = rt->buf("mutdbl");
    PS_TRY(db.ensure(dbls * sizeof(double)));
    int* dp = mb.as<int>();
    double* dd = db.as<double>();
    // every AlignData's score array first, back to back: ONE device-to-host copy returns them all (a copy per region before:
    // 20 000 of a bench step's 25 000 copy commands, ~0.2 ms each on a loaded stream)
    double* const score0 = dd;
    size_t score_tot = 0;
    std::vector<size_t> score_at(R, 0);
    for (int k = 0; k < R; k++) { score_at[k] = score_tot; score_tot += (size_t)std::max(plan[k].M, 1); }
    dd += score_tot;
    std::vector<int> stage;
    stage.reserve(ints);
    auto push = [&](const std::vector<int>& v) { int* r = dp + stage.size(); stage.insert(stage.end(), v.begin(), v.end()); return r; };
    std::vector<ScoreArgs> sas(R);
    for (int k = 0; k < R; k++) {
        const EditPlan& p = plan[k];
        ScoreArgs& sa = sas[k];
        memset(&sa, 0, sizeof(sa));
        sa.job0 = job0[k]; sa.njobs = as[k]->E;
        sa.nitems_per_job = p.M; sa.ncolmax = p.ncolmax; sa.ws = as[k]->par.scoring_width; sa.nr0 = p.nr0;
        sa.m_start = push(p.start); sa.m_mlen = push(p.mlen); sa.m_cm = push(p.cm); sa.m_ncol = push(p.ncol);
        sa.m_skip = push(p.skip); sa.m_oldidx = push(p.oldidx); sa.m_states = push(p.states); sa.r0 = push(p.r0s);
        for (int q = 0; q < SCORE_CLASSES; q++) { sa.cls_items[q] = push(p.cls[q]); sa.cls_count[q] = (int)p.cls[q].size(); }
        sa.old = dd; dd += (size_t)as[k]->E * std::max(p.nr0, 1);
        sa.delta = dd; dd += (size_t)as[k]->E * std::max(p.M, 1);
        sa.score = score0 + score_at[k];
        // edit positions on more than a quarter of the columns: column-pair maxima of ALL columns in one coalesced pass (k_oldall)
        sa.oldall_pitch = (int64_t)as[k]->states.size() + 8;
        sa.maxS = b.maxS;
        sa.oldall = !b.sparse && (size_t)p.nr0 * 4 > as[k]->states.size() && p.nr0 >= 256 ? dd : nullptr;   // (k_oldall walks full matrices)
        if (sa.oldall) PS_HIP(hipMemsetAsync(sa.oldall, 0, (size_t)as[k]->E * sa.oldall_pitch * sizeof(double), rt->stream));
        dd += (size_t)as[k]->E * sa.oldall_pitch;
        if (!p.M || !as[k]->E) { sa.njobs = 0; sa.nitems_per_job = 0; }   // nothing to score for this AlignData: its blocks leave at once
    }
    // the edit tables and their descriptors (ScoreArgs, behind the tables in the same buffer) in one copy
    const size_t sa_at = (stage.size() * sizeof(int) + 63) / 64 * 64;
    std::vector<char> blob(sa_at + (size_t)R * sizeof(ScoreArgs));
    memcpy(blob.data(), stage.data(), stage.size() * sizeof(int));
    memcpy(blob.data() + sa_at, sas.data(), (size_t)R * sizeof(ScoreArgs));
    PS_TRY(rt->up(dp, blob.data(), blob.size()));
    const ScoreArgs* d_sas = (const ScoreArgs*)((const char*)dp + sa_at);
    tk.lap("upload");
    if (tk.on) { PS_HIP(hipStreamSynchronize(rt->stream)); }
    tk.lap("realign fwd+back (rest)");
    PS_TRY(launch_lb(rt, b.d, 1, b.maxlbn));
    std::vector<double*> sc(R, nullptr);
    for (int k = 0; k < R; k++) {
        const EditPlan& p = plan[k];
        if (!p.M || !as[k]->E) continue;
        if (rt->prof_on) {
            // per (event, edit) item  16(Bs+1) + 16 Br + 24(Bs+c) + 32 Br / k + 8
            double t = 0;
            const double Bs = 2.0 * sas[k].ws + 1, Br = 2.0 * as[k]->par.realign_width + 1;
            const double kk = (double)p.M / std::max(p.nr0, 1);
            for (int i = 0; i < p.M; i++) t += 16 * (Bs + 1) + 16 * Br + 24 * (Bs + p.mlen[i] + 6) + 32 * Br / kk + 8;
            rt->prof["score"].bytes += t * as[k]->E;
            rt->prof["score"].units += (double)p.M * as[k]->E;
        }
    }
    PS_TRY(launch_score(rt, b.d, d_sas, sas));
    std::vector<double*> dl(R, nullptr);
    double* all_scores = nullptr;
    PS_TRY(rt->down(&all_scores, score0, score_tot));
    for (int k = 0; k < R; k++)
        if (plan[k].M && as[k]->E) {
            sc[k] = all_scores + score_at[k];
            if (delta_out && (*delta_out)[k]) PS_TRY(rt->down(&dl[k], sas[k].delta, (size_t)as[k]->E * plan[k].M));
        }
    PS_HIP(hipStreamSynchronize(rt->stream));
    for (int k = 0; k < R; k++) {
        if (sc[k]) for (int i = 0; i < plan[k].M; i++) (*outs[k])[i].score = sc[k][i];
        if (dl[k]) memcpy((*delta_out)[k], dl[k], (size_t)as[k]->E * plan[k].M * sizeof(double));
    }
    tk.lap("score edits");
    return PS_OK;
}

int score_mutations(Runtime* rt, Align* a, const std::vector<Mut>& muts, std::vector<Mut>* out) {
    if (!a->E) {
        *out = muts;
        for (Mut& m : *out) m.score = -1e-6;
        for (const Mut& m : muts) if (m.start < 0) return fail(PS_ERR_BAD_ARG, "negative mutation start");
        return PS_OK;
    }
    return score_mutations_multi(rt, {a}, {&muts}, {out});
}

// FindPointMutations, cpp/FindMutations.cpp:191-234
void find_point_mutations(const Align* a, std::vector<Mut>* out) {
    static const char B4[] = "ACGT";
    out->clear();
    out->reserve(a->states.size() * 8);
    for (size_t i = 0; i < a->states.size(); i++) {
        Mut m;
        m.start = (int)i;
        m.orig.assign(1, a->bases[i]);
        out->push_back(m);
        for (int k = 0; k < 4; k++) {
            if (a->bases[i] == B4[k]) continue;
            m.mut.assign(1, B4[k]);
            out->push_back(m);
        }
        m.orig.clear();
        for (int k = 0; k < 4; k++) { m.mut.assign(1, B4[k]); out->push_back(m); }
    }
    if (a->par.verbose) { fputs("Point ", stderr); fflush(stderr); }   // cpp/FindMutations.cpp:230-231
}

static bool by_score_desc(const Mut& x, const Mut& y) { return x.score > y.score; }  // cpp/MakeMutations.cpp:16-17

// MakeMutations, cpp/MakeMutations.cpp:74-146.  std::sort with the same comparator on the same
// libstdc++ gives the reference's (unstable) order for tied scores.
// One greedy pass (host only): sorts, applies the positive edits, returns the mutated-base count and the edits that
// were disabled on the way (the reference re-scores and recurses on those when there are more than ten).
static int greedy_apply(Align* a, std::vector<Mut>& muts, std::vector<Mut>* later, bool talk) {
    const int spacing = 10;
    int nb = 0;
    later->clear();
    {
        // The reference sorts the whole list by descending score and drops the negative tail (cpp/MakeMutations.cpp:80-86).  When
        // the scores that survive (>= 0) are pairwise different — the normal case: a Refine list is 80 000 edits of which a few
        // hundred are positive — their order does not depend on how the rest was permuted, so only they are sorted.  Equal scores
        // among them are ordered by std::sort's own permutation of the WHOLE list, which is then reproduced (index sort: the same
        // comparisons as on the structs, without moving two std::strings per swap).
        std::vector<int> order;
        for (int k = 0; k < (int)muts.size(); k++) if (!(muts[k].score < 0)) order.push_back(k);
        const std::vector<Mut>& mref = muts;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return by_score_desc(mref[x], mref[y]); });
        bool ties = false;
        for (size_t k = 1; k < order.size(); k++) if (muts[order[k - 1]].score == muts[order[k]].score) { ties = true; break; }
        if (ties || order.size() == muts.size()) {
            order.resize(muts.size());
            std::iota(order.begin(), order.end(), 0);
            std::sort(order.begin(), order.end(), [&](int x, int y) { return by_score_desc(mref[x], mref[y]); });
            while (!order.empty() && muts[order.back()].score < 0) order.pop_back();
        }
        std::vector<Mut> kept;
        kept.reserve(order.size());
        for (int k : order) kept.push_back(std::move(muts[k]));
        muts.swap(kept);
    }
    if (muts.empty()) return 0;
    if (talk) { fprintf(stderr, "Testing %zu mutations...\n", muts.size()); fflush(stderr); }   // cpp/MakeMutations.cpp:91-95
    bool changed = false;
    // cpp/MakeMutations.cpp:95-139 with the edits' numbers in flat arrays: the inner loop over all later edits (defer the ones
    // within `spacing` of the applied edit, shift the ones behind it) is then a branch-free integer loop the compiler vectorises —
    // a Mutate list has ~3 000 surviving edits, 4.5 million pair visits per region and call
    const size_t n = muts.size();
    std::vector<int> st(n), ml(n), ol(n), pos(n), dfr(n, 0);
    for (size_t k = 0; k < n; k++) {
        st[k] = muts[k].start; ml[k] = (int)muts[k].mut.size(); ol[k] = (int)muts[k].orig.size();
        pos[k] = muts[k].score > 0 ? 1 : 0;
    }
    for (size_t i = 0; i < n; i++) {
        muts[i].start = st[i];
        if (dfr[i] || muts[i].score < 0) { if (dfr[i]) muts[i].score = -1; later->push_back(muts[i]); continue; }
        a->bases = apply_edit(a->bases, muts[i]);
        changed = true;
        if (talk && a->par.verbose > 1) {   // cpp/MakeMutations.cpp:112-118 (operator<< of a double: six significant digits)
            fprintf(stderr, "Kept mutation %zu at %d of %zu to %zu with score %g\n", i, st[i], muts[i].orig.size(), muts[i].mut.size(), muts[i].score);
            fflush(stderr);
        }
        nb += (int)std::max(muts[i].orig.size(), muts[i].mut.size());
        const int si = st[i], ei = si + ml[i], oi = si + ol[i], d = ml[i] - ol[i];
        int* __restrict__ pst = st.data();
        int* __restrict__ ppos = pos.data();
        int* __restrict__ pdf = dfr.data();
        const int* __restrict__ pml = ml.data();
        for (size_t j = i + 1; j < n; j++) {
            const int sj = pst[j];
            const int lo = std::max(si, sj), hi = std::min(ei, sj + pml[j]);
            const int hit = (lo < hi + spacing) & ppos[j];          // overlaps the applied edit (with spacing) and still has a positive score: deferred
            ppos[j] &= ~hit;
            pdf[j] |= hit;
            pst[j] = sj + ((!hit & (sj >= oi)) ? d : 0);            // (a deferred edit keeps its start: the reference `continue`s before the shift)
        }
    }
    if (changed) a->states = states_of(a->bases);
    return nb;
}

// MakeMutations for several AlignData in lock-step: the greedy passes run on host threads, every round of re-scoring
// is one batched ScoreMutations over the AlignData that still have more than ten disabled edits
int make_mutations_multi(Runtime* rt, const std::vector<Align*>& as, std::vector<std::vector<Mut>> muts, std::vector<int>* nbases) {
    Tick tk("make_mutations");
    const int R = (int)as.size();
    nbases->assign(R, 0);
    std::vector<int> active(R);
    std::iota(active.begin(), active.end(), 0);
    std::vector<std::vector<Mut>> later(R);
    while (!active.empty()) {
        par_for((int)active.size(), [&](int q) {
            const int k = active[q];
            (*nbases)[k] += greedy_apply(as[k], muts[k], &later[k], R == 1 && as[k]->par.verbose);
        });
        tk.lap("greedy apply");
        std::vector<int> next;
        for (int k : active) if (later[k].size() > 10) next.push_back(k);
        if (next.empty()) break;
        std::vector<Align*> sa;
        std::vector<const std::vector<Mut>*> in;
        std::vector<std::vector<Mut>*> out;
        for (int k : next) { sa.push_back(as[k]); in.push_back(&later[k]); out.push_back(&muts[k]); }
        PS_TRY(score_mutations_multi(rt, sa, in, out));
        active.swap(next);
    }
    return PS_OK;
}

// one line about the process-wide state of the library (ps_info): stream / hardware-queue mode, runtimes, memory plan
std::string info_string() {
    std::string why;
    (void)hwq_mode(&why);
    size_t nslab = 0, slab_b = 0;
    { std::lock_guard<std::mutex> lk(g_slab_mu); nslab = g_slabs.size(); for (Slab* sl : g_slabs) slab_b += sl->bytes; }
    char buf[512];
    snprintf(buf, sizeof buf, "; device fraction of this process %.3f; runtimes: %d live, %d peak; share per runtime %.1f GB; slabs for full score matrices: %zu of %d allocated (%.1f GB, %.1f GB each by plan); device pools of this process %.1f GB",
             device_fraction(), live_runtimes(), peak_runtimes(), device_share_bytes() * 1e-9, nslab, slab_count(), slab_b * 1e-9, slab_bytes() * 1e-9, (double)g_pool_bytes.load() * 1e-9);
    return "hip-gfx950; streams: " + why + buf;
}

int make_mutations(Runtime* rt, Align* a, std::vector<Mut> muts, int* nbases) {
    std::vector<int> nb;
    std::vector<std::vector<Mut>> in(1);
    in[0] = std::move(muts);
    PS_TRY(make_mutations_multi(rt, {a}, std::move(in), &nb));
    *nbases = nb[0];
    return PS_OK;
}

}  // namespace ps
