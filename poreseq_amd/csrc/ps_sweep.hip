// ps_sweep.hip — the forward DP of one (event, sequence) alignment on ONE wavefront, for alignments that only need their
// backtrace: ScoreAlignments (cpp/MakeMutations.cpp:148-195), i.e. the base re-alignment and every candidate sequence of
// FindMutations (cpp/FindMutations.cpp:24-60) — a third of a consensus schedule's sweeps (bench: 52 900 of 158 800 per step).
//
// Reference behaviour reproduced (file:line under the reference tree):
//   fillColumn                    cpp/Alignment.cpp:111-274   (k_sweep: same cell arithmetic as k_fill, operation for operation)
//   running MaxInfo               cpp/Alignment.cpp:158, 270  (per-strip maxima + k_best)
//   backtrace                     cpp/Alignment.cpp:516-624   (k_backtrace_s: the walker of ps_dev.h on k_sweep's step codes;
//                                                              k_like_a / k_like_b: the scores along the path, recomputed)
//
// Geometry.  Rows (levels) are cut into strips of K consecutive rows; strip q = rows qK+1 .. qK+K.  Lane (q mod 64) of the wave
// (ps_sweepw.hip: lane q mod 128 / 256 of a workgroup of two / four waves, the cross-wave neighbour through LDS)
// owns strip q and walks it one column per step, K cells top to bottom; strip q works on column j at step t = j + q, so the
// lane one up finished the same column one step earlier: the cell above a strip's first row arrives by one wave-rotate DPP move
// of {main, stay}, the rest of a column's vertical dependencies are the lane's own registers.  One wavefront per sweep needs no
// barrier and no other wave.  The strips in band on one step form a contiguous window
// [qlo(t), qhi(t)] (both band ends are monotone in the column); K is chosen so that the widest window leaves one lane idle,
// and a lane's strip is qlo + ((lane - qlo) mod 64).
//   per step and lane (not per cell): band of its column and the last band row of the previous one (12-byte load, two steps ahead),
//   the column's 5-mer (two steps ahead) and its 64-byte model row, one step ahead, from a ring in LDS indexed by column that wave 0
//   fills — each column's row leaves L2 once per sweep (ps_sweep_body.h: "model rows through LDS"; round 6; the north_star's staging);
//   per strip: the K level records, kept in registers for the ~2W/slope columns the strip stays in band.
// What leaves the chip: SEVEN BITS per cell — the raw predicates of the fill (which candidate equals the maximum, EXTEND > STAY,
// the two "score > 0" tests the walker stops on; ps_sweep_body.h, CB_*), shifted into a register by add-with-carry and decoded by the
// reader (StripCodes below) — laid out [step][row group][lane] so every store is a full coalesced line; plus one {best, i, j} record
// per strip.  No score matrix: the scores along the backtrace path (ref_like) are recomputed afterwards.  A cell on the path
// equals its predecessor's score plus the move's terms (cpp/Alignment.cpp:196-237), with the very operations the fill used, so a
// serial pass along the path (k_like_b: one wave per job, ~3 dependent additions per level) reproduces them bit for bit.
#include "ps_sweep_body.h"

namespace ps {

// ------------------------------------------------------------------------------------------------
// band[j] = {i0(j), i1(j)} for j = 0 .. C + 1: column 0 is the blank column covering rows 0 .. n0 (cpp/Alignment.cpp:42),
// column C + 1 an empty sentinel.  grid (ceil((maxC + 2) / 256), njobs)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_band(BatchD b, SweepD sw) {
    chain_priority_wide();
    const int jd = blockIdx.y, dir = jd % sw.ndir;
    const JobD& J = b.jobs[jd / sw.ndir];
    const SweepJob& SJ = sw.sj[jd];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j > J.C + 1) return;
    int i0 = 0, i1 = J.n0;
    if (j > J.C) { i0 = J.n0 + 1; i1 = J.n0; }
    else if (j >= 1) band_of(b.lb + J.lb_off, dir, j, J.C, J.n0, J.W, i0, i1);
    sw.band[SJ.band_off + j] = make_int2(i0, i1);
}

// qlo[t] / qhi[t] = lowest / highest strip in band on step t (-1: none) and the widest window of the batch (sw.maxwin).
// Strip q works on column t - q; it is in band iff  i0(t-q) <= qK + K  (true from some q on: i0 falls as the column does)
// and  i1(t-q) >= qK + 1  (true up to some q).  grid (ceil((maxT + Q_PAD) / 256), njobs * ndir)
__global__ __launch_bounds__(256) void k_qlo(BatchD b, SweepD sw) {
    chain_priority_wide();
    const int jd = blockIdx.y;
    const JobD& J = b.jobs[jd / sw.ndir];
    const SweepJob& SJ = sw.sj[jd];
    const int t = blockIdx.x * 256 + threadIdx.x;
    int win = 0;
    if (t < SJ.T + Q_PAD) {
        const int K = sw.K, C = J.C;
        const int2* __restrict__ band = sw.band + SJ.band_off;
        int res = -1, top = -1;
        const int qa = max(0, t - C), qb = min(SJ.Q - 1, t - 1);
        if (t < SJ.T && qa <= qb && !J.out->inert) {
            int lo = qa, hi = qb + 1;                     // smallest q in [qa, qb] with i0(t - q) <= qK + K  (hi = qb + 1: none)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (band[t - mid].x <= mid * K + K) hi = mid; else lo = mid + 1; }
            if (lo <= qb && band[t - lo].y >= lo * K + 1) {
                res = lo;
                int l2 = lo, h2 = qb;                     // largest q with i1(t - q) >= qK + 1
                while (l2 < h2) { const int mid = (l2 + h2 + 1) >> 1; if (band[t - mid].y >= mid * K + 1) l2 = mid; else h2 = mid - 1; }
                top = l2;
                win = l2 - lo + 1;
            }
        }
        sw.qlo[SJ.q_off + t] = res;
        sw.qhi[SJ.q_off + t] = top;
    }
    for (int off = 32; off; off >>= 1) win = max(win, __shfl_xor(win, off));
    if ((threadIdx.x & 63) == 0 && win > 0) atomicMax(sw.maxwin, win);
}

// registers: two waves per SIMD up to K = 10 on one wave (the level records of ten rows are 80 registers), one beyond
#define PS_SWEEP_WPE(K) K <= 10 ? 2 : 1

// forward-only batches: one wave per job
template <int K, bool FD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PS_SWEEP_WPE(K), PS_SWEEP_WPE(K))))
void k_sweep(BatchD b, SweepD sw) {
    __shared__ __attribute__((aligned(16))) char mring[mring_bytes(1)];
    const JobD& J = b.jobs[blockIdx.x];
    if (J.out->inert) return;
    sweep_body<K, 1, 0, 0, FD>(b, sw, J, sw.sj[blockIdx.x], nullptr, nullptr, mring);
}

// Alignment::update batches: one wave per (job, direction), sweep job jd = 2 * job + direction
template <int K, bool FD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PS_SWEEP_WPE(K), PS_SWEEP_WPE(K))))
void k_sweep2(BatchD b, SweepD sw) {
    __shared__ unsigned long long ring[ring_cols(1)];
    __shared__ __attribute__((aligned(16))) char mring[mring_bytes(1)];
    const int jd = blockIdx.x;
    const JobD& J = b.jobs[jd >> 1];
    if (J.out->inert) return;
    if ((jd & 1) == 0) sweep_body<K, 1, 0, 1, FD>(b, sw, J, sw.sj[jd], ring, nullptr, mring);
    else sweep_body<K, 1, 1, 1, FD>(b, sw, J, sw.sj[jd], ring, nullptr, mring);
}

// Alignment::update batches whose edit list reads few columns: the same sweeps with column-sparse records
template <int K, bool FD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PS_SWEEP_WPE(K), PS_SWEEP_WPE(K))))
void k_sweeps(BatchD b, SweepD sw) {
    __shared__ unsigned long long ring[ring_cols(1)];
    __shared__ __attribute__((aligned(16))) char mring[mring_bytes(1)];
    const int jd = blockIdx.x;
    const JobD& J = b.jobs[jd >> 1];
    if (J.out->inert) return;
    if ((jd & 1) == 0) sweep_body<K, 1, 0, 2, FD>(b, sw, J, sw.sj[jd], ring, nullptr, mring);
    else sweep_body<K, 1, 1, 2, FD>(b, sw, J, sw.sj[jd], ring, nullptr, mring);
}

// the global maximum and its first cell — smallest column, then smallest row (cpp/Alignment.cpp:158, 270: strict '>' over columns
// in order, rows in order) — from the per-strip records; one wave per job
__global__ __launch_bounds__(64) void k_best(BatchD b, SweepD sw) {
    chain_priority();
    const JobD& J = b.jobs[blockIdx.x];
    JobOut* O = J.out;
    if (O->inert) return;
    const SweepJob& SJ = sw.sj[blockIdx.x * sw.ndir];
    const StripBest* SB = sw.sb + SJ.sb_off;
    double v = 0.0;
    int bi = 0x7fffffff, bj = 0x7fffffff;
    for (int q = threadIdx.x; q < SJ.Q; q += 64) {
        const StripBest s = SB[q];
        if (s.v > v || (s.v == v && s.v > 0.0 && (s.j < bj || (s.j == bj && s.i < bi)))) { v = s.v; bi = s.i; bj = s.j; }
    }
    for (int off = 32; off; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
        if (ov > v || (ov == v && ov > 0.0 && (oj < bj || (oj == bj && oi < bi)))) { v = ov; bi = oi; bj = oj; }
    }
    if (threadIdx.x == 0) {
        O->best = v;
        if (v > 0.0) { O->bi = bi; O->bj = bj; } else { O->bi = 0; O->bj = 0; }
    }
}

// ------------------------------------------------------------------------------------------------
// backtrace on k_sweep's codes
// ------------------------------------------------------------------------------------------------
template <int K>
struct StripCodes {
    const unsigned char* codes;
    const int2* band;                       // the sweep's band table of the job (forward): band[j] = rows of column j, j = 0 .. C + 1
    const int* st;                          // the sequence's 5-mers: a column without one has no cell
    int nl;                                 // lanes of a sweep (64 per wave): strip q sits on lane q mod nl
    int ci0, ci1, cp0, cp1, cst;            // lane l: band of column tj - l, band of the column before it, its 5-mer (prep)
    static constexpr bool ROWFAST = true;   // consecutive threads take consecutive rows: K contiguous bytes per strip
    static constexpr bool TWO_PASS = true;  // bt_load: every code word and the column tables in flight before the first decode
    // What the sweep's predicate bits leave to the reader — does the cell exist (its column has a 5-mer, the row is in the column's
    // band), is a MATCH real or implicit (cpp/Alignment.cpp:207-220: p0 < i <= p1 of the previous column's band) — hangs on the
    // COLUMN: lane l of every loading wave fetches the three table entries of column tj - l once per tile, together with the tile's
    // code words (one memory round trip, as before), and a cell picks its column's by v_readlane: the 64 threads that share a column
    // are one wave (bt_load: row offset = thread mod 64).
    __device__ __forceinline__ void prep(int, int tj, int t) {
        const int col = tj - (t & 63);
        ci0 = 1; ci1 = 0; cp0 = 0; cp1 = -1; cst = -1;
        if (col >= 1) {
            const int2 bc = band[col], bp = band[col - 1];
            ci0 = bc.x; ci1 = bc.y; cp0 = bp.x; cp1 = bp.y; cst = st[col - 1];
        }
    }
    // fetch: the cell's 32 / 16 / 8-bit code word (its address must not wait for the column tables: any row >= 1 of any column >= 1
    // has one, in band or not); decode: the walker's step word from the sweep's raw predicate bits (ps_codes.h, CB_*)
    __device__ __forceinline__ unsigned fetch(int ti, int tj, int a, int c) const {
        const int rs = max(ti - a, 1), col = max(tj - c, 1);
        const int q = (rs - 1) / K, rr = (rs - 1) - q * K;
        return code_fetch(codes + (size_t)(col + q) * (nl * K), K, q & (nl - 1), rr, nl);
    }
    __device__ __forceinline__ unsigned short decode(unsigned by, int ti, int tj, int a, int c) const {
        const int r = ti - a;
        const int cu = __builtin_amdgcn_readfirstlane(c);                      // (wave-uniform: see prep)
        const int i0 = __builtin_amdgcn_readlane(ci0, cu), i1 = __builtin_amdgcn_readlane(ci1, cu);
        const int p0 = __builtin_amdgcn_readlane(cp0, cu), p1 = __builtin_amdgcn_readlane(cp1, cu);
        const int sc = __builtin_amdgcn_readlane(cst, cu);
        const bool cell = r >= 1 && tj - cu >= 1 && sc >= 0 && r >= i0 && r <= i1;   // (outside the matrix / no cell here: score 0, the walk stops)
        const bool vd = r > p0 && r <= p1;
        const unsigned sm = code_main_step(by, vd), ss = code_stay_step(by);
        const unsigned w = sm | (ss << 8) | ((by & CB_POS) ? 0u : 0x4000u) | ((by & CB_SPOS) ? 0u : 0x8000u);
        return (unsigned short)(cell ? w : 0xC000u);
    }
};

template <int K>
__global__ __launch_bounds__(256) void k_backtrace_s(BatchD b, SweepD sw) {
    const JobD& J = b.jobs[blockIdx.x];
    const SweepJob& SJ = sw.sj[blockIdx.x * sw.ndir];
    StripCodes<K> src;
    src.codes = sw.codes + SJ.codes_off;
    src.band = sw.band + SJ.band_off;
    src.st = J.st;
    src.nl = sw.nl;
    bt_walk(J, src);
}

// ------------------------------------------------------------------------------------------------
// ref_like along the path.  k_like_a (parallel): the emission of every recorded cell whose move adds one, into the job's
// ref_index array (scratch here: updaterefs rewrites it next).  k_like_b (serial, one wave per job): from the cell the walk
// stopped on, level by level: skips since the previous record (`+ lik_skip` each), then the recorded move.
//   MATCH   prev.main[i-1] + obs            IGNORE  prev.main[i-1] + lik_insert        INSERT  cur.main[i-1] + lik_insert
//   STAY    cur.main[i-1] + obs + lik_stay  EXTEND  cur.stay[i-1] + obs + lik_extend   (cpp/Alignment.cpp:196-237)
// A switch from the main to the stay matrix on one cell leaves the score unchanged (the cell's main score IS its stay score).
// ------------------------------------------------------------------------------------------------
template <bool FD>
__device__ __forceinline__ double cell_emission(const BatchD& b, const JobD& J, int i, int j) {
    const int state = J.st[j - 1];
    const double2* row = (const double2*)((const char*)J.model8 + (size_t)max(state, 0) * MODEL_ROW_BYTES);
    const double2 q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
    const double m[8] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
    const double4 l4 = ((const double4*)J.lev[0])[i - 1];
    const double lv[4] = {l4.x, l4.y, l4.z, l4.w};
    return emission8<FD>(m, lv, b.log2pi, J.lik_offset);
}

template <bool FD>
__global__ __launch_bounds__(256) void k_like_a(BatchD b) {
    chain_priority_wide();
    const JobD& J = b.jobs[blockIdx.y];
    if (J.out->inert) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= J.n0) return;
    const long long w = ((const long long*)J.rl)[t];
    double o = 0.0;
    if (w != 0) {
        const int j = (int)(w >> 4), st = (int)(w >> 1) & 7;
        if (st == (int)M_MATCH || st == (int)M_STAY || st == (int)M_EXTEND) o = cell_emission<FD>(b, J, t + 1, j);
    }
    J.ri[t] = o;
}

template <bool FD>
__global__ __launch_bounds__(64) void k_like_b(BatchD b) {
    chain_priority();
    const JobD& J = b.jobs[blockIdx.x];
    const JobOut O = *J.out;
    if (O.inert || O.bi <= 0) return;
    const int lane = threadIdx.x;
    const int iz = O.term_i, jz = O.term_w >> 3, kind = O.term_w & 3;
    const double lsk = J.lsk, lst = J.lst, lex = J.lex, lin = J.lin;
    double v = 0.0;
    if ((kind & 1) && iz >= 1 && jz >= 1) v = cell_emission<FD>(b, J, iz, jz);
    int curj = jz;
    long long* rlw = (long long*)J.rl;
    for (int i0 = iz + 1; i0 <= O.bi; i0 += 64) {
        const int i = i0 + lane;
        long long w = 0; double o = 0.0;
        if (i <= O.bi) { w = rlw[i - 1]; o = J.ri[i - 1]; }
        const int wj = (int)(w >> 4), wst = (int)(w >> 1) & 7;
        double res = 0.0;
        const int n = min(64, O.bi - i0 + 1);
        for (int k = 0; k < n; k++) {
            const int j = __builtin_amdgcn_readlane(wj, k), st = __builtin_amdgcn_readlane(wst, k);
            const double ok = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(o), k), __builtin_amdgcn_readlane(__double2loint(o), k));
            const int jx = (st == (int)M_MATCH || st == (int)M_IGNORE) ? j - 1 : j;
            for (int s = curj; s < jx; s++) v = v + lsk;
            if (st == (int)M_MATCH) v = v + ok;
            else if (st == (int)M_IGNORE || st == (int)M_INSERT) v = v + lin;
            else if (st == (int)M_STAY) v = v + ok + lst;
            else v = v + ok + lex;
            curj = j;
            res = lane == k ? v : res;
        }
        if (i <= O.bi) J.rl[i - 1] = res;
    }
}

// ------------------------------------------------------------------------------------------------
// The `likes` loop of ScoreAlignments (cpp/MakeMutations.cpp:168-189) on the device: per (sequence, event) the per-base cumulative
// likelihood is piecewise constant — likes[k + 1] += ref_like of the LAST level aligned at or before base k (0 before the first) —
// and a sequence's vector is the sum of its events' in event order.  One 256-thread block per sequence (group of E jobs): for
// each event in order, every aligned level marks its base with its index (atomic max: the last level of equal bases wins), a
// prefix maximum over the bases turns the marks into "last aligned level so far", and the block adds ref_like of that level to the
// vector — the same additions in the same order as the host loop, so only L doubles per sequence cross PCIe instead of every
// job's ref_align / ref_like (517 MB per FindMutations call of a 20-region batch).
// ------------------------------------------------------------------------------------------------
constexpr int LK_MAXC = 12 * 1024;      // bases per sequence the LDS index table holds (longer sequences take the host loop)
__global__ __launch_bounds__(256) void k_likes(BatchD b, const LikeGroup* __restrict__ groups, double* __restrict__ out) {
    chain_priority_wide();
    __shared__ int s_idx[LK_MAXC];
    __shared__ int s_part[256];
    const LikeGroup G = groups[blockIdx.x];
    const int tid = threadIdx.x, nk = G.C + 4;                    // bases k = 0 .. C + 3
    double* __restrict__ lk = out + G.out_off;
    const int chunk = (nk + 255) / 256, k0 = min(nk, tid * chunk), k1 = min(nk, k0 + chunk);
    for (int k = tid; k < G.len; k += 256) lk[k] = 0.0;
    for (int e = 0; e < G.njobs; e++) {
        const JobD& J = b.jobs[G.job0 + e];
        const double* __restrict__ ra = J.ra;
        const double* __restrict__ rl = J.rl;
        for (int k = tid; k < nk; k += 256) s_idx[k] = -1;
        __syncthreads();
        for (int t = tid; t < J.n0; t += 256) {
            const double a = ra[t];
            if (a > 0) { const int k = (int)a; if (k < nk) atomicMax(&s_idx[k], t); }
        }
        __syncthreads();
        int run = -1;                                                 // prefix maximum: chunk-local, then across the chunks
        for (int k = k0; k < k1; k++) { run = max(run, s_idx[k]); s_idx[k] = run; }
        s_part[tid] = run;
        __syncthreads();
        int before = -1;
        for (int q = 0; q < tid; q++) before = max(before, s_part[q]);
        for (int k = k0; k < k1; k++) {
            const int t = max(s_idx[k], before);
            if (k >= 1 && k < G.C + 3 && k + 1 < G.len) lk[k + 1] += t >= 0 ? rl[t] : 0.0;
        }
        __syncthreads();
    }
}

int launch_likes(Runtime* rt, const BatchD& b, const LikeGroup* d_groups, int ngroups, double* d_out) {
    if (!ngroups) return PS_OK;
    hipLaunchKernelGGL(k_likes, dim3(ngroups), dim3(256), 0, rt->stream, b, d_groups, d_out);
    PS_HIP(hipGetLastError());
    return PS_OK;
}
int likes_max_states() { return LK_MAXC - 4; }

// =================================================================================================
// host side
// =================================================================================================
#define PS_LAUNCH_CHECK() PS_HIP(hipGetLastError())

// Forms of the strip sweep: K rows per lane on NW wavefronts per sweep (NL = 64 NW lanes).  The strips in band on one step form a
// window of about (2W + 1) / (K + levels per base) + 2 strips, which must fit NL - 1 lanes (sweep_win_max).
//   NW = 1   one wavefront per sweep: the fewest instructions per cell at K = 10 for the default width (57 of 64 lanes busy), ~27 ms
//            for a 10 kb sweep whatever the chip could do
//   NW = 2   K = 4: the same SIMD time per sweep (120 of 128 lanes busy; the per-step overhead of a lane is spread over 4 cells
//            instead of 10) in half the time, 80 registers fewer (the level records of six rows)
//   NW = 4   K = 2: a quarter more SIMD time, a third of the time: launches that leave most of the chip idle
static const int K_LIST1[] = {4, 6, 10, 16, 24, 32, 0};
static const int K_LIST2[] = {4, 5, 6, 10, 0};
static const int K_LIST4[] = {2, 3, 4, 6, 0};
static const int* k_list(int NW) { return NW == 1 ? K_LIST1 : NW == 2 ? K_LIST2 : NW == 4 ? K_LIST4 : nullptr; }
// The strips in band on one step must leave ONE lane of the sweep idle: the lane above the lowest strip in band then holds a strip
// that is out of band, so what the first row of a band reads as its upper neighbour is the absent-cell value (a band's top row has no
// neighbour above: cpp/Alignment.cpp:226-236; the kernels do not mask it, they rely on that value).  With a window of NL - 1 strips at
// most, the strip NL above the one a lane has just left cannot be in band yet.  (Rounds 3-4 kept two lanes idle; the bench's windows
// at K = 4 sit at 125-128 strips of 128 lanes, and every window of 127 took the next larger strip height at +16 % instructions.)
int sweep_win_max(int NW) { return 64 * NW - 1; }
static int guess_window(int W, int K) { return (2 * W + 1) / (K + 1) + 3; }

// smallest strip height on NW wavefronts whose window probably fits (K = 0: none)
SweepForm sweep_guess_form(int W, int NW) {
    SweepForm f;
    const int* l = k_list(NW);
    if (!l) return f;
    for (; *l; l++) if (guess_window(W, *l) <= sweep_win_max(NW)) { f.K = *l; f.NW = NW; return f; }
    return f;
}
// the next larger form after a window that did not fit: the next strip height on the same number of wavefronts, else the
// single-wavefront form of that capacity
SweepForm sweep_next_form(SweepForm f, int win) {
    SweepForm n;
    for (const int* l = k_list(f.NW); l && *l; l++) if (*l > f.K) { n.K = *l; n.NW = f.NW; return n; }
    if (f.NW > 1)
        for (const int* l = K_LIST1; *l; l++) if (*l * sweep_win_max(1) > f.K * win) { n.K = *l; n.NW = 1; return n; }
    return n;
}
bool sweep_form_exists(int K, int NW) {
    for (const int* l = k_list(NW); l && *l; l++) if (*l == K) return true;
    return false;
}
int sweep_guess_k(int W) { return sweep_guess_form(W, 1).K; }

template <int K>
static void sweep_launch_k(Runtime* rt, const BatchD& b, const SweepD& sw) {
    if (sw.ndir == 2 && sw.sparse) {
        if (b.fastdiv) hipLaunchKernelGGL((k_sweeps<K, true>), dim3(b.njobs * 2), dim3(64), 0, rt->stream, b, sw);
        else hipLaunchKernelGGL((k_sweeps<K, false>), dim3(b.njobs * 2), dim3(64), 0, rt->stream, b, sw);
        return;
    }
    if (sw.ndir == 2) {
        if (b.fastdiv) hipLaunchKernelGGL((k_sweep2<K, true>), dim3(b.njobs * 2), dim3(64), 0, rt->stream, b, sw);
        else hipLaunchKernelGGL((k_sweep2<K, false>), dim3(b.njobs * 2), dim3(64), 0, rt->stream, b, sw);
        return;
    }
    if (b.fastdiv) hipLaunchKernelGGL((k_sweep<K, true>), dim3(b.njobs), dim3(64), 0, rt->stream, b, sw);
    else hipLaunchKernelGGL((k_sweep<K, false>), dim3(b.njobs), dim3(64), 0, rt->stream, b, sw);
}
template <int K>
static void bt_launch_k(Runtime* rt, const BatchD& b, const SweepD& sw) {
    hipLaunchKernelGGL((k_backtrace_s<K>), dim3(b.njobs), dim3(256), 0, rt->stream, b, sw);
}

// bytes one job takes in form f: step codes (forward), and with full = true the records of both directions
double sweep_job_bytes(int n0, int C, SweepForm f, bool full) {
    const int K = std::max(f.K, 1);
    const double steps = (double)C + (n0 + K - 1) / K + 1;
    return steps * 64.0 * f.NW * K * (full ? 1.0 + 2 * 16.0 : 1.0);
}

// the strip tables of a batch in form f: sweep-job records (one per job and direction), band table, qlo / qhi + the
// widest window (read back by the caller).  bt.ndir == 2: also the record offsets of both directions (JobD.mat_off, JobD.K, JobD.NL).
int sweep_prepare(Runtime* rt, Batch& bt, SweepForm f) {
    const BatchD& b = bt.d;
    const int nd = bt.ndir, K = f.K, NL = 64 * f.NW;
    std::vector<SweepJob>& sj = bt.sjobs;
    sj.resize(bt.jobs.size() * nd);
    int64_t band_tot = 0, q_tot = 0, sb_tot = 0, code_tot = 0, rec_tot = 0;
    int maxT = 0;
    for (size_t k = 0; k < bt.jobs.size(); k++) {
        JobD& j = bt.jobs[k];
        j.K = nd == 2 ? (bt.sparse ? -1 : K) : 0;
        j.NL = NL;
        // kept columns: rows of the widest band + K - 1 records of padding on either side (a lane stores all K rows of its strip: those
        // outside the band land there), even: 32-byte aligned columns
        if (bt.sparse) j.pitch = (std::min(2 * j.W + 1, j.n0) + 2 * (K - 1) + 3) & ~1;
        for (int d = 0; d < nd; d++) {
            SweepJob s;
            s.Q = (j.n0 + K - 1) / K;
            s.T = j.C + s.Q;                         // steps t = column + strip run 1 .. C + Q - 1
            s.band_off = band_tot; band_tot += j.C + 2;
            s.q_off = q_tot; q_tot += s.T + Q_PAD;
            s.sb_off = sb_tot; if (d == 0) sb_tot += std::max(s.Q, 1);
            s.codes_off = code_tot; if (d == 0) code_tot += (int64_t)s.T * NL * K;
            if (nd == 2 && bt.sparse) { j.mat_off[d] = rec_tot + (K - 1); rec_tot += (int64_t)bt.nkeep[2 * k + d] * j.pitch + 2 * K; }
            else if (nd == 2) { j.mat_off[d] = rec_tot; rec_tot += (int64_t)s.T * NL * K; }
            maxT = std::max(maxT, s.T);
            sj[k * nd + d] = s;
        }
    }
    bt.sweep_K = K; bt.sweep_NW = f.NW; bt.sweep_maxT = maxT; bt.sweep_code_bytes = code_tot; bt.sweep_sb = sb_tot; bt.sweep_recs = rec_tot;
    PS_TRY(rt->buf("sw_jobs").ensure(std::max<size_t>(sj.size(), 1) * sizeof(SweepJob)));
    PS_TRY(rt->buf("sw_band").ensure(std::max<int64_t>(band_tot, 1) * sizeof(int2)));
    PS_TRY(rt->buf("sw_qlo").ensure(std::max<int64_t>(q_tot, 1) * sizeof(int)));
    PS_TRY(rt->buf("sw_qhi").ensure(std::max<int64_t>(q_tot, 1) * sizeof(int)));
    PS_TRY(rt->buf("sw_sb").ensure(std::max<int64_t>(sb_tot, 1) * sizeof(StripBest)));
    PS_TRY(rt->buf("sw_win").ensure(64));
    PS_TRY(rt->up(rt->buf("sw_jobs").p, sj.data(), sj.size() * sizeof(SweepJob)));
    SweepD& sw = bt.sd;
    sw.sj = rt->buf("sw_jobs").as<SweepJob>();
    sw.band = rt->buf("sw_band").as<int2>();
    sw.qlo = rt->buf("sw_qlo").as<int>();
    sw.qhi = rt->buf("sw_qhi").as<int>();
    sw.sb = (StripBest*)rt->buf("sw_sb").p;
    sw.maxwin = rt->buf("sw_win").as<int>();
    sw.codes = nullptr;
    sw.K = K;
    sw.nl = NL;
    sw.ndir = nd;
    sw.sparse = nd == 2 && bt.sparse ? 1 : 0;
    PS_HIP(hipMemsetAsync(sw.maxwin, 0, sizeof(int), rt->stream));
    hipLaunchKernelGGL(k_band, dim3((bt.maxC + 2 + 255) / 256, b.njobs * nd), dim3(256), 0, rt->stream, b, sw);
    hipLaunchKernelGGL(k_qlo, dim3((maxT + Q_PAD + 255) / 256, b.njobs * nd), dim3(256), 0, rt->stream, b, sw);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

// sweeps, maxima, backtrace and path scores of a prepared batch (the code / record pools are placed by the caller)
int sweep_run(Runtime* rt, Batch& bt) {
    const BatchD& b = bt.d;
    SweepD& sw = bt.sd;
    const int K = bt.sweep_K, NW = bt.sweep_NW;
    PS_HIP(hipMemsetAsync(sw.sb, 0, (size_t)std::max<int64_t>(bt.sweep_sb, 1) * sizeof(StripBest), rt->stream));
    prof_begin(rt);
    if (NW > 1) {
        if (!b.fastdiv || !sweepw_launch(rt, b, sw, K, NW)) return fail(PS_ERR_BAD_ARG, "sweep_run: no such multi-wavefront form");
    } else {
        switch (K) {
            case 4: sweep_launch_k<4>(rt, b, sw); break;
            case 6: sweep_launch_k<6>(rt, b, sw); break;
            case 10: sweep_launch_k<10>(rt, b, sw); break;
            case 16: sweep_launch_k<16>(rt, b, sw); break;
            case 24: sweep_launch_k<24>(rt, b, sw); break;
            case 32: sweep_launch_k<32>(rt, b, sw); break;
            default: return fail(PS_ERR_BAD_ARG, "sweep_run: strip height");
        }
    }
    PS_LAUNCH_CHECK();
    prof_end(rt, "sweep", 0.0);
    if (rt->prof_on) {   // (which form ran: host-side counts, no event pair)
        if (NW > 1) rt->prof[NW == 2 ? "sweep_w2" : "sweep_w4"].launches++;
        if (bt.ndir == 2 && bt.sparse) rt->prof["sweep_kept"].launches++;
    }
    hipLaunchKernelGGL(k_best, dim3(b.njobs), dim3(64), 0, rt->stream, b, sw);
    if (bt.ndir == 2) PS_TRY(launch_prefix(rt, b, 2));   // running MaxInfo per column of both directions (the strip jobs' best cell is k_best's)
    switch (K) {
        case 2: bt_launch_k<2>(rt, b, sw); break;
        case 3: bt_launch_k<3>(rt, b, sw); break;
        case 4: bt_launch_k<4>(rt, b, sw); break;
        case 5: bt_launch_k<5>(rt, b, sw); break;
        case 6: bt_launch_k<6>(rt, b, sw); break;
        case 10: bt_launch_k<10>(rt, b, sw); break;
        case 16: bt_launch_k<16>(rt, b, sw); break;
        case 24: bt_launch_k<24>(rt, b, sw); break;
        default: bt_launch_k<32>(rt, b, sw); break;
    }
    PS_LAUNCH_CHECK();
    if (bt.maxn > 0) {
        if (b.fastdiv) {
            hipLaunchKernelGGL(k_like_a<true>, dim3((bt.maxn + 255) / 256, b.njobs), dim3(256), 0, rt->stream, b);
            hipLaunchKernelGGL(k_like_b<true>, dim3(b.njobs), dim3(64), 0, rt->stream, b);
        } else {
            hipLaunchKernelGGL(k_like_a<false>, dim3((bt.maxn + 255) / 256, b.njobs), dim3(256), 0, rt->stream, b);
            hipLaunchKernelGGL(k_like_b<false>, dim3(b.njobs), dim3(64), 0, rt->stream, b);
        }
        PS_LAUNCH_CHECK();
    }
    return PS_OK;
}

}  // namespace ps
