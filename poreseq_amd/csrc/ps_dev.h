// ps_dev.h — device-side helpers shared by the kernel files (ps_kernels.hip, ps_sweep.hip): band arithmetic, the emission
// log-density with its exact quotients, address-space and wave-uniform helpers.  See ps_kernels.hip for the reference citations.
#ifndef PS_DEV_H_
#define PS_DEV_H_

#include "ps_internal.h"

namespace ps {

constexpr unsigned FLG_DEAD = 0xC000u;   // final step word of a cell in an invalid-5-mer column: both scores <= 0, no move
enum : unsigned { M_SKIP = 0, M_MATCH = 1, M_INSERT = 2, M_IGNORE = 3, M_STAY = 4, M_EXTEND = 5, M_IMPL = 255 };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Serial single-wave kernels of a launch chain (the backtrace walkers, the path-score pass, the per-job maxima: < 3 % of the vector
// work, a fifth of a ScoreMutations chain's latency under load) can raise their waves' issue priority at entry: beside three sweep
// waves on its SIMD a lone walker wave otherwise gets a quarter of the issue slots and runs 4x slower than alone, while its batch
// offers the chip nothing else.  PS_WALKER_PRIO: the serial walkers; PS_WIDE_PRIO: the short chip-wide table builders between them.
#ifndef PS_WALKER_PRIO
#define PS_WALKER_PRIO 3
#endif
#ifndef PS_WIDE_PRIO
#define PS_WIDE_PRIO 0
#endif
__device__ __forceinline__ void chain_priority() {
    if (PS_WALKER_PRIO > 0) __builtin_amdgcn_s_setprio(PS_WALKER_PRIO);
}
__device__ __forceinline__ void chain_priority_wide() {
    if (PS_WIDE_PRIO > 0) __builtin_amdgcn_s_setprio(PS_WIDE_PRIO);
}

// std::lower_bound(double*, int) exactly as libstdc++ walks it (cpp/EventData.h:178)
__device__ inline int lower_bound_d(const double* __restrict__ a, int n, int v) {
    int first = 0, len = n;
    const double dv = (double)v;
    while (len > 0) {
        int half = len >> 1;
        int mid = first + half;
        if (a[mid] < dv) { first = mid + 1; len = len - half - 1; } else { len = half; }
    }
    return first;
}

// band of column `col` (1-based) in direction dir; lb = raw lower_bound table (-1 = empty ref_index)
__device__ __forceinline__ void band_of(const int* __restrict__ lb, int dir, int col, int C, int n0, int W, int& i0, int& i1) {
    int c;
    if (dir == 0) { int v = lb[col]; c = v < 0 ? 1 : v; }
    else { int v = lb[C - col + 1]; c = v < 0 ? 1 : n0 - v + 1; }
    c = clampi(c, 1, n0);
    i0 = max(1, c - W);
    i1 = min(n0, c + W);
}


// index of cell (row i, column j) of a job's forward (dir 0) / backward (dir 1) matrix in the record pool: skewed storage
// REC[i + j][i mod P] (k_fill), strip storage REC[j + q][r][q mod NL] with q = (i - 1) / K, r = (i - 1) mod K (k_sweep2), or
// column-sparse storage REC[kept index of column j][i - i0] (k_sweeps; i0 = first row of the column's band, which every reader has
// at hand; only kept columns and rows inside their band exist)
__device__ __forceinline__ int64_t rec_index(const JobD& J, int dir, int i, int j, int i0) {
    if (J.K < 0) return J.mat_off[dir] + (int64_t)J.keep[dir][j] * J.pitch + (i - i0);
    if (J.K) {
        const int q = (i - 1) / J.K, r = (i - 1) - q * J.K;
        return J.mat_off[dir] + ((int64_t)(j + q) * J.K + r) * J.NL + (q & (J.NL - 1));
    }
    return J.mat_off[dir] + (int64_t)(i + j) * J.P + i % J.P;
}

// the value of the lane before (lane 0 of the wave reads zero: its callers give a group's first lane a value of its own).  With
// bound_ctrl the instruction needs no previous value of its destination — two register copies per use less than `old = v`.
__device__ __forceinline__ double wave_shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// `x` where keep, else a huge negative FINITE number: only the high word is selected (0xFFEFFFFF: -1.797e308 whatever the low word
// holds).  It stands for "no cell here" exactly as -infinity does — it loses every maximum against a real score (>= 0), against the
// floors 0 and -1e300 and in every `>` test, adding an emission or a transition term leaves it where it is, and two of them are never
// added — at one v_cndmask instead of two.
__device__ __forceinline__ double keep_or_absent(double x, bool keep) {
    const int hi = keep ? __double2hiint(x) : (int)0xFFEFFFFF;
    return __hiloint2double(hi, __double2loint(x));
}


// global-memory pointers the compiler cannot trace back to a kernel argument (they come out of the job table) are
// declared in address space 1 explicitly: otherwise every access through them is a FLAT access, which counts in both
// vmcnt and lgkmcnt and forces "s_waitcnt vmcnt(0) lgkmcnt(0)" — i.e. a full drain of the streaming stores — per step
#define PS_GLOBAL __attribute__((address_space(1)))
typedef const PS_GLOBAL double* gcdp;
typedef const PS_GLOBAL int* gcip;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef v4i __attribute__((aligned(4))) v4i_a4;   // four states at any 4-byte aligned address
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));

// a value every lane of the wave agrees on, moved to scalar registers (pointers and sizes read through the job table arrive in
// vector registers; addressing with them would cost 64-bit vector arithmetic per access)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* uni_ptr(T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (T*)(((uint64_t)hi << 32) | lo);
}


// a / b with y = RN(1 / b): Markstein's FMA sequence (see the header comment); exact IEEE quotient
__device__ __forceinline__ double mdiv(double a, double b, double y) {
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// emission log-density (cpp/AlignUtil.h:34-38, 48-53 + cpp/Alignment.cpp:169-173), operation for operation
// m = {mu, 1/sg, sg, log sg, sm, 1/sm, lambda, log lambda}; lev = {x, sd, 3 log sd, 1/sd}
template <bool FASTDIV>
__device__ __forceinline__ double emission8(const double (&m)[8], const double (&lev)[4], const double log2pi, const double off) {
    const double a1 = lev[0] - m[0], a2 = lev[1] - m[4];
    const double d = FASTDIV ? mdiv(a1, m[2], m[1]) : a1 / m[2];
    const double e = FASTDIV ? mdiv(a2, m[4], m[5]) : a2 / m[4];
    double l = -0.5 * (d * d + log2pi) - m[3];
    const double t = e * e * m[6];
    const double q = FASTDIV ? mdiv(t, lev[1], lev[3]) : t / lev[1];
    const double g = 0.5 * (m[7] - lev[2] - log2pi - q);
    l += g;
    l += off;
    return l;
}

// ------------------------------------------------------------------------------------------------
// backtrace (cpp/Alignment.cpp:516-624): one 256-thread block per job, 64 x 64 tiles of step words staged in LDS.
// The walker navigates on 16-bit step words alone — {main step, stay step << 8, bit 14 / 15: main / stay score <= 0} — which a
// code source hands out per cell: SkewCodes (k_fill's skewed u16 matrix) or StripCodes<K> (k_sweep's packed bytes, ps_sweep.hip).
// Wave 0 walks the current tile while waves 1-3 fetch the tile the path is expected to enter next (same diagonal, BTM cells of
// overlap to absorb drift) into the other LDS buffer; when the walk leaves the current tile inside the prefetched one no load is
// waited for.  For every recorded level the walker stores ref_align directly and, in ref_like's slot, the cell it was recorded
// from as an integer: column << 4 | step code << 1 | matrix.  Where the walk ended — the cell it stopped on without recording
// it — goes to the job's result record (term_i / term_w: column << 3 | matrix << 2 | kind, kind 1 = an implicit-match cell,
// whose score is its own emission; otherwise the score there is zero): k_like_path (ps_sweep.hip) recomputes the scores along
// the path from it, k_fill_like (ps_kernels.hip) reads them from the stored matrices.
// ------------------------------------------------------------------------------------------------
constexpr int BT = 64;   // tile edge of the backtrace
constexpr int BTM = 8;   // the tile prefetched during a walk overlaps the current one by this many rows / columns

struct SkewCodes {       // FLG[s][slot], s = i + j, slot = i mod P
    const unsigned short* flg; int P; int sti;
    static constexpr bool ROWFAST = false;
    static constexpr bool TWO_PASS = false;   // (words are stored as the walker reads them: nothing to decode)
    __device__ __forceinline__ void prep(int ti, int, int) { sti = __builtin_amdgcn_readfirstlane(ti > 0 ? ti % P : 0); }
    // fetch: the load of one cell's word; decode: what the walker reads (here: the word itself)
    __device__ __forceinline__ unsigned fetch(int ti, int tj, int a, int c) const {
        const int r = ti - a, col = tj - c;
        int slot = sti - a;          // (ti - a) mod P, a < BT <= P
        if (slot < 0) slot += P;
        return (r >= 1 && col >= 1) ? flg[(int64_t)(r + col) * P + slot] : 0xC000u;   // outside the matrix: score 0, the walk stops
    }
    __device__ __forceinline__ unsigned short decode(unsigned raw, int, int, int, int) const { return (unsigned short)raw; }
};

// cooperative load of the BT x BT step-word tile whose corner (largest row / column) is (ti, tj); NT threads, t in [0, NT).
// Two passes: every load of a thread is in flight (with the per-column tables a source fetches in prep) before the first word is
// decoded and stored — one memory round trip per tile.
template <int NT, class SRC>
__device__ __forceinline__ void bt_load(unsigned short (*__restrict__ dst)[BT + 2], SRC& src, const int ti, const int tj, const int t) {
    constexpr int NQ = (BT * BT + NT - 1) / NT;
    src.prep(ti, tj, t);
    if constexpr (!SRC::TWO_PASS) {
        // (a source whose words need no decoding: all loads of a thread in flight before the first LDS store, as since round 2)
        unsigned short tmp[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int idx = min(t + NT * q, BT * BT - 1);
            const int a = SRC::ROWFAST ? idx % BT : idx / BT, c = SRC::ROWFAST ? idx / BT : idx % BT;
            tmp[q] = src.decode(src.fetch(ti, tj, a, c), ti, tj, a, c);
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int idx = t + NT * q;
            if (idx < BT * BT) { const int a = SRC::ROWFAST ? idx % BT : idx / BT, c = SRC::ROWFAST ? idx / BT : idx % BT; dst[a][c] = tmp[q]; }
        }
        return;
    }
    unsigned raw[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int idx = min(t + NT * q, BT * BT - 1);
        const int a = SRC::ROWFAST ? idx % BT : idx / BT, c = SRC::ROWFAST ? idx / BT : idx % BT;
        raw[q] = src.fetch(ti, tj, a, c);
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int idx = t + NT * q;
        const int ic = min(idx, BT * BT - 1);
        const int a = SRC::ROWFAST ? ic % BT : ic / BT, c = SRC::ROWFAST ? ic / BT : ic % BT;
        const unsigned short w = src.decode(raw[q], ti, tj, a, c);
        if (idx < BT * BT) dst[a][c] = w;
    }
}

template <class SRC>
__device__ __forceinline__ void bt_walk(const JobD& J, SRC& src) {
    const JobOut O = *J.out;
    if (O.inert) return;  // stripe_width == 0: the event is left untouched
    const int tid = threadIdx.x, n0 = J.n0;
    // (wave priority for the WALKER wave only: with the tile loaders raised too, the kernel alone was 14 % slower than without any
    //  priority — 1.24 against 1.08 ms per 10 kb job — the loaders' decode work then competes with the walk it feeds)
    if (PS_WALKER_PRIO > 0 && tid < 64) __builtin_amdgcn_s_setprio(PS_WALKER_PRIO);
    double* __restrict__ ra = J.ra;
    long long* __restrict__ rlw = (long long*)J.rl;
    for (int t = tid; t < n0; t += 256) { ra[t] = 0.0; rlw[t] = 0ll; }
    __syncthreads();
    __shared__ unsigned short t_buf[2][BT][BT + 2];
    __shared__ int s_state[2][5];
    int i = O.bi, j = O.bj, arr = 0, term = 0;
    bool done = (i <= 0);
    int cur = 0, ti = 0, tj = 0, it = 0;
    bool need = true;
    while (!done) {
        if (need) {
            ti = i; tj = j;
            bt_load<256>(t_buf[cur], src, ti, tj, tid);
            __syncthreads();
        }
        const int pi = ti - (BT - BTM), pj = tj - (BT - BTM);
        unsigned short (*t_step)[BT + 2] = t_buf[cur];
        if (tid >= 64) bt_load<192>(t_buf[cur ^ 1], src, pi, pj, tid - 64);
        if (tid < 64) {
            // wave 0 walks; (i, j, arr) are wave-uniform.  Lane l looks l cells ahead on the diagonal, so a run of
            // MATCH steps (the common case) is emitted by one LDS read + one ballot with coalesced stores; the
            // first non-MATCH cell after the run is stepped from the word its lane already holds.
            const int l = tid;
            while (true) {
                i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j);
                arr = __builtin_amdgcn_readfirstlane(arr);
                if (i <= 0) { done = true; break; }
                const int a = ti - i, c = tj - j;
                if (a >= BT || c >= BT) break;  // left the tile: reload around (i, j)
                const int aa = a + l, cc = c + l;
                unsigned w = 0xC000u;
                if (aa < BT && cc < BT) w = t_step[aa][cc];
                int run = 0;
                if (arr == 0) {
                    const unsigned long long mm = __ballot((w & 0x40ffu) == M_MATCH);   // main cell, score > 0, MATCH
                    run = __builtin_amdgcn_readfirstlane(mm == ~0ull ? 64 : (int)__builtin_ctzll(~mm));
                    if (l < run) { ra[i - 1 - l] = (double)(j - l); rlw[i - 1 - l] = ((long long)(j - l) << 4) | (long long)(M_MATCH << 1); }
                    i -= run; j -= run;
                    if (run == 64 || a + run >= BT || c + run >= BT) continue;
                    if (i <= 0) { done = true; break; }
                }
                const unsigned wr = __builtin_amdgcn_readlane(w, run);
                const unsigned st = arr ? ((wr >> 8) & 7u) : (wr & 255u);
                if (wr & (arr ? 0x8000u : 0x4000u)) { done = true; break; }   // score <= 0
                const long long here = ((long long)j << 4) | (long long)((st & 7u) << 1) | arr;
                double rav = 0.0;
                int rec = 0, di = 0, dj = 0;
                if (st == M_SKIP) { dj = 1; }
                else if (st == M_MATCH) { rav = (double)j; rec = 1; di = 1; dj = 1; }
                else if (st == M_IGNORE) { rav = -1.0; rec = 1; di = 1; dj = 1; }
                else if (st == M_INSERT) { rav = -1.0; rec = 1; di = 1; }
                else if (st == M_STAY) {
                    if (arr == 1) { rav = (double)j; rec = 1; di = 1; }
                    arr = 1 - arr;
                }
                else if (st == M_EXTEND) { rav = (double)j; rec = 1; di = 1; }
                else { done = true; term = (st == M_IMPL) ? 1 : 0; break; }
                if (rec && l == 0) { ra[i - 1] = rav; rlw[i - 1] = here; }
                i -= di; j -= dj;
            }
            if (tid == 0) { int* ss = s_state[it & 1]; ss[0] = i; ss[1] = j; ss[2] = arr; ss[3] = done ? 1 : 0; ss[4] = term; }
        }
        __syncthreads();
        { const int* ss = s_state[it & 1]; i = ss[0]; j = ss[1]; arr = ss[2]; done = ss[3] != 0; term = ss[4]; }
        it++;
        // continue in the prefetched tile if the path left the current one inside it
        const int a = pi - i, c = pj - j;
        need = !(a >= 0 && c >= 0 && a < BT && c < BT);
        if (!need) { cur ^= 1; ti = pi; tj = pj; }
    }
    if (tid == 0) { J.out->term_i = i; J.out->term_w = (j << 3) | (arr << 2) | term; }
}

}  // namespace ps
#endif
