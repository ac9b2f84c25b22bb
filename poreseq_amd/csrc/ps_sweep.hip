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
// owns strip q and walks it one column per step, K cells top to bottom; strip q works on column j at step t = j + q, so the
// lane one up finished the same column one step earlier: the cell above a strip's first row arrives by one wave-rotate DPP move
// of {main, stay}, the rest of a column's vertical dependencies are the lane's own registers.  No LDS, no barrier, no other
// wave: 2048 sweeps are resident on the chip at two waves per SIMD.  The strips in band on one step form a contiguous window
// [qlo(t), qhi(t)] (both band ends are monotone in the column); K is chosen so that the widest window leaves two lanes idle,
// and a lane's strip is qlo + ((lane - qlo) mod 64).
//   per step and lane (not per cell): band of its column and the previous one (16-byte load, two steps ahead), the column's
//   5-mer (two steps ahead) and its 64-byte model row (one step ahead, straight from global memory / L2: one row serves K cells);
//   per strip: the K level records, kept in registers for the ~2W/slope columns the strip stays in band.
// What leaves the chip: ONE BYTE per cell — main step (3 bits, 7 = implicit), stay step (2 bits), the two "score <= 0" bits
// (32 / 64) the walker stops on — laid out [step][row group][lane] so every store is a full coalesced line; plus one {best, i, j} record
// per strip.  No score matrix: the scores along the backtrace path (ref_like) are recomputed afterwards.  A cell on the path
// equals its predecessor's score plus the move's terms (cpp/Alignment.cpp:196-237), with the very operations the fill used, so a
// serial pass along the path (k_like_b: one wave per job, ~3 dependent additions per level) reproduces them bit for bit.
#include "ps_dev.h"
#include "ps_host.h"

namespace ps {

// ---- byte layout of one step's codes: row groups ("planes") of 16 / 8 / 4 / 2 / 1 rows, each [64 lanes][rows of the group] ----
__host__ __device__ constexpr int plane_sz(int rem) { return rem >= 16 ? 16 : rem >= 8 ? 8 : rem >= 4 ? 4 : rem >= 2 ? 2 : 1; }
template <int K>
__device__ __forceinline__ int code_off(int lane, int r) {
    int r0 = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        const int sz = plane_sz(K - r0);
        if (r < r0 + sz) return 64 * r0 + lane * sz + (r - r0);
        r0 += sz;
        if (r0 >= K) break;
    }
    return 0;
}

struct StripBest { double v; int i, j; };
constexpr int Q_PAD = 8;          // qlo entries behind T (all -1): the sweep looks two steps ahead
constexpr int WIN_MAX = 62;       // strips in band on one step: two lanes stay idle (a lane is never handed its next strip in the step it leaves one)

// ------------------------------------------------------------------------------------------------
// band[j] = {i0(j), i1(j)} for j = 0 .. C + 1: column 0 is the blank column covering rows 0 .. n0 (cpp/Alignment.cpp:42),
// column C + 1 an empty sentinel.  grid (ceil((maxC + 2) / 256), njobs)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_band(BatchD b, SweepD sw) {
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

// ------------------------------------------------------------------------------------------------
// k_sweep: one wave per job
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_ror1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x13C /*wave_ror:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xf, 0xf, false);
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

// One sweep.  DIR 0 / 1: forward / backward fill (cpp/Alignment.cpp:111-274 / 280-444: the backward cell adds its emission when it
// is LEFT, so what a lane hands down and keeps for the diagonal is {main, stay + emission, main + emission}).  MODE 0
// (ScoreAlignments): only the forward step codes and the per-strip maxima leave the chip.  MODE 1 / 2: the sweep of an
// Alignment::update (ScoreMutations) — also the per-column maxima (MaxInfo, cpp/Alignment.cpp:158, 270) through a 256-column LDS
// ring to cmax, and {main, stay} records: of every cell, REC[step][row of the strip][lane] (MODE 1: one coalesced 1 KB store per
// row and step), or only of the columns the edit list will read (MODE 2: scoreMutation / columnMax, cpp/Alignment.cpp:447-512,
// cpp/Alignment.h:181-214, read the forward columns max(start-4, 0) and max(start-3, 1) and two backward columns per edit): the
// lane whose column is kept (JobD.keep, fetched with the band record) writes its in-band cells to REC[kept column][row - i0].
constexpr int RING = 256;       // columns of the maxima ring: the window's 62 + the 64 steps between two flushes, rounded up

template <int K, int DIR, int MODE, bool FD>
__device__ __forceinline__ void sweep_body(const BatchD& b, const SweepD& sw, const JobD& J, const SweepJob& SJ, unsigned long long* ring) {
    const int lane = threadIdx.x;
    const int C = uni(J.C), T = uni(SJ.T), n0 = uni(J.n0);
    typedef const __attribute__((address_space(4))) int* kcip;   // constant address space + uniform index = scalar load
    kcip QLO = (kcip)uni_ptr(sw.qlo + SJ.q_off);
    kcip QHI = (kcip)uni_ptr(sw.qhi + SJ.q_off);
    gcip band = (gcip)uni_ptr((const int*)(sw.band + SJ.band_off));
    gcip st = (gcip)uni_ptr(J.st);
    const PS_GLOBAL char* model = (const PS_GLOBAL char*)uni_ptr((const char*)J.model8);
    const PS_GLOBAL v4d* levs = (const PS_GLOBAL v4d*)uni_ptr(J.lev[DIR]);
    PS_GLOBAL unsigned char* codes = (PS_GLOBAL unsigned char*)uni_ptr(sw.codes + SJ.codes_off);
    PS_GLOBAL char* rec = (PS_GLOBAL char*)uni_ptr((char*)(b.rec + J.mat_off[DIR]));
    double* gcmax = uni_ptr(b.cmax + J.col_off[DIR]);
    StripBest* SB = sw.sb + SJ.sb_off;
    const double lsk = J.lsk, lst = J.lst, lex = J.lex, lin = J.lin, off = J.lik_offset, log2pi = b.log2pi;
    const double NINF = -__builtin_inf();

    // ---- what a lane fetches ahead of the step it is needed on
    struct Ahead { v4i bd; int sp, sc, kc; };   // bd = {p0, p1, i0, i1} of columns j - 1, j; sp / sc = 5-mer of column j - 1 / j; kc: kept-column index (MODE 2)
    gcip keep = (gcip)uni_ptr(J.keep[DIR]);
    const int pitch = uni(J.pitch);
    auto fetch = [&](int tt, int ql) -> Ahead {
        const int q = ql + ((lane - ql) & 63);
        const int j = clampi(tt - q, 1, max(C, 1));   // (a sequence without a 5-mer has no live step; its prefetches still need an address)
        Ahead a;
        a.bd = *(const PS_GLOBAL v4i_a4*)(band + 2 * (j - 1));
        typedef int v2i_a4 __attribute__((ext_vector_type(2), aligned(4)));
        if (DIR == 0) {
            const v2i_a4 s2 = *(const PS_GLOBAL v2i_a4*)(st + (j - 2));   // (ints of -1 around the list: column 0 reads as invalid)
            a.sp = s2.x; a.sc = s2.y;
        } else {                                                          // backward column j holds states[C - j]
            const v2i_a4 s2 = *(const PS_GLOBAL v2i_a4*)(st + (C - j));
            a.sc = s2.x; a.sp = s2.y;
        }
        a.kc = MODE == 2 ? keep[j] : -1;
        return a;
    };
    auto model_row = [&](int state, double (&m)[8]) {
        const PS_GLOBAL v2d* row = (const PS_GLOBAL v2d*)(model + (size_t)(unsigned)max(state, 0) * MODEL_ROW_BYTES);
        const v2d q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
        m[0] = q0.x; m[1] = q0.y; m[2] = q1.x; m[3] = q1.y; m[4] = q2.x; m[5] = q2.y; m[6] = q3.x; m[7] = q3.y;
    };

    // ---- lane state
    double lev[K][4];        // level records of the lane's strip: {mean, stdv, 3 log stdv, 1 / stdv}
    double pm[K];            // main scores of the previous column on the strip's rows; -infinity: no cell there
    double pe[DIR ? K : 1];  // backward: main + emission of the previous column
#pragma unroll
    for (int r = 0; r < K; r++) { pm[r] = NINF; if (DIR) pe[r] = NINF; lev[r][0] = 0.0; lev[r][1] = 1.0; lev[r][2] = 0.0; lev[r][3] = 1.0; }
    int qcur = -1;
    double bot_m = NINF, bot_s = NINF, bot_e = NINF;   // the lane's last cell of the step: what the lane one down reads as its upper neighbour
    double dm = NINF, de = NINF;                       // upper neighbour's main (backward: and main + emission) of the previous column
    double lbest = 0.0;                  // strictly greater wins: the first cell of the strip (column, then row) holding its maximum
    int lbt = 0, lbr = 0;
    int flushed = 1, next_flush = 64;    // (MODE > 0) columns below `flushed` have their maximum in memory
    if (MODE) {
        for (int k = lane; k < RING; k += 64) ring[k] = 0ull;
    }

    int ql0 = QLO[1], ql1 = QLO[2];          // qlo of step t, t + 1 (scalar registers; T + Q_PAD entries, -1 behind T)
    Ahead a0 = fetch(1, ql0), a1 = fetch(2, ql1);
    double mr[8];
    model_row(a0.sc, mr);

    for (int t = 1; t < T; t++) {
        const int ql2 = QLO[t + 2];
        const Ahead a2 = fetch(t + 2, ql2);
        const int ql = ql0;
        const bool live = ql >= 0;                                   // (uniform) some strip is in band on this step
        const int q = ql + ((lane - ql) & 63);
        const int j = t - q;
        if (live && q != qcur) {
            // the lane takes its next strip: hand in the old one's maximum, fetch the new level records
            if (DIR == 0 && lbest > 0.0) { StripBest sbv; sbv.v = lbest; sbv.i = qcur * K + 1 + lbr; sbv.j = lbt - qcur; SB[qcur] = sbv; }
            lbest = 0.0;
            qcur = q;
#pragma unroll
            for (int r = 0; r < K; r++) {
                const v4d v = levs[min(q * K + r, n0 - 1)];
                lev[r][0] = v.x; lev[r][1] = v.y; lev[r][2] = v.z; lev[r][3] = v.w;
                pm[r] = NINF;
                if (DIR) pe[r] = NINF;
            }
        }
        // emissions of the lane's K cells first: the model row is then free to receive the next column's (one step of lead)
        double ov[K];
#pragma unroll
        for (int r = 0; r < K; r++) {
            ov[r] = emission8<FD>(mr, lev[r], log2pi, off);
            if (r & 1) __builtin_amdgcn_sched_barrier(0);            // two emissions in flight: enough to fill the pipe, few enough to stay in registers
        }
        __builtin_amdgcn_sched_barrier(0);
        model_row(a1.sc, mr);
        if (live) {
            const int base = q * K + 1;
            const bool valid = j >= 1 && j <= C && a0.sc >= 0;     // (a column whose 5-mer is invalid is all zero: no cell takes part, cpp/Alignment.cpp:162-163)
            const int ra = valid ? a0.bd.z - base : K, rb = valid ? a0.bd.w - base : -1;   // band rows relative to the strip
            const int rc = a0.bd.x - base, rd_ = a0.bd.y - base;                          // previous column's band
            const bool pzero = a0.sp < 0;                            // previous column invalid (or column 0): its scores read as zero
            double um = wave_ror1(bot_m), us = wave_ror1(bot_s), ue = DIR ? wave_ror1(bot_e) : 0.0;
            double dprev = dm, deprev = de;
            dm = um; de = ue;
            const double lbefore = lbest;
            unsigned cw[(K + 3) / 4];                                // (forward) the step's codes, four to a register
            double crun = 0.0;                                       // (MODE > 0) the lane's share of its column's maximum
            PS_GLOBAL char* recp = rec + ((size_t)t * K * 64 + lane) * 16;
            const bool kept = MODE == 2 && a0.kc >= 0 && j >= 1 && j <= C;
            if (MODE == 2) {
                // records of the lane's column: row i of the band [i0, i1] at REC[kc][i - i0]; recp = the strip's first row
                recp = rec + ((int64_t)a0.kc * pitch + (base - a0.bd.z)) * 16;
                if (kept && !valid)   // a kept column without a 5-mer: its band reads as zeros (cpp/Alignment.cpp:162-163)
                    for (int r = max(0, a0.bd.z - base); r <= min(K - 1, a0.bd.w - base); r++) *(PS_GLOBAL v2d*)(recp + (size_t)r * 16) = (v2d){0.0, 0.0};
            }
#pragma unroll
            for (int r = 0; r < K; r++) {
                const double o = ov[r];
                const bool act = r >= ra && r <= rb;
                const bool top = r == ra;
                const bool vd = r > rc && r <= rd_;                  // cpp/Alignment.cpp:207: p0 < i <= p1
                const bool rd = vd && !pzero;
                const double pmr = pm[r];
                double L;
                asm("v_max_f64 %0, %1, 0" : "=v"(L) : "v"(pmr));
                const double D = rd ? dprev : 0.0;
                const double cSTAY = DIR == 0 ? um + o + lst : ue + lst;      // backward: (main + emission) of the cell above
                const double cEXT = DIR == 0 ? us + o + lex : us + lex;       // backward: `us` carries stay + emission
                const double cINS = um + lin;
                const double cSKIP = L + lsk;
                const double cMATCH = DIR == 0 ? D + o : (rd ? deprev : 0.0);
                const double cIGN = D + lin;
                // the stay floor of a band's first row is -1e300 (cpp/Alignment.cpp:230): a value that only ever loses.  Where the record of
                // the cell is kept for a bit-exact comparison (MODE 1) it is -BIG itself; elsewhere -1e300 with a zero low word does
                // the same at one select (the high word) instead of two
                const double floor_s = MODE == 1 ? (top ? -BIG : 0.0) : __hiloint2double(top ? __double2hiint(-BIG) : 0, 0);
                const double t1 = fmax(floor_s, cSTAY);
                const double ns = fmax(t1, cEXT);
                double nm = fmax(0.0, cSKIP);
                nm = fmax(nm, cMATCH);
                nm = fmax(nm, cINS);
                nm = fmax(nm, cIGN);
                nm = fmax(nm, ns);
                unsigned ss = 0u, sm = 4u;
                if (DIR == 0) {
                    // step codes: stay matrix STAY then EXTEND with strict '>', main matrix in the reference's order (first candidate equal to the maximum)
                    ss = cSTAY > floor_s ? 1u : 0u;
                    ss = cEXT > t1 ? 2u : ss;
                    sm = cIGN == nm ? 3u : sm;
                    sm = cINS == nm ? 2u : sm;
                    sm = cMATCH == nm ? (vd ? 1u : 7u) : sm;
                    sm = cSKIP == nm ? 0u : sm;
                    sm = nm > 0.0 ? sm : 0u;
                }
                const double nmx = keep_or_absent(nm, act), nsx = keep_or_absent(ns, act);
                if (DIR == 0) {
                    // (a row outside the band carries both "score <= 0" bits — the walker stops there before it reads the step — because
                    //  its scores are the absent-cell value; its step bits are whatever the selects left)
                    unsigned w = sm | (ss << 3) | (nmx > 0.0 ? 0u : 32u) | (nsx > 0.0 ? 0u : 64u);   // (every constant an inline operand)
                    asm volatile("" : "+v"(w));                      // keep the byte's shift out of the selects' constants (a VGPR per shifted literal otherwise)
                    if ((r & 3) == 0) cw[r >> 2] = w; else cw[r >> 2] |= w << (8 * (r & 3));
                }
                dprev = pmr;
                pm[r] = nmx;
                if (DIR) { deprev = pe[r]; pe[r] = nmx + o; }
                um = nmx;
                us = DIR == 0 ? nsx : nsx + o;
                if (DIR) ue = nmx + o;
                if (MODE == 1) {
                    double rx;   // the stored record: the cell; zeros where there is none (the stay value of a top row is -1e300 and stays so)
                    asm("v_max_f64 %0, %1, 0" : "=v"(rx) : "v"(nmx));
                    *(PS_GLOBAL v2d*)(recp + (size_t)r * 1024) = (v2d){rx, act ? ns : 0.0};
                    crun = fmax(crun, rx);
                }
                if (MODE == 2) {
                    if (act && kept) *(PS_GLOBAL v2d*)(recp + r * 16) = (v2d){nm, ns};   // (a cell in band: nm >= 0)
                    crun = fmax(crun, nmx);
                }
                if (DIR == 0) {
                    const bool gt = nmx > lbest;
                    asm("v_max_f64 %0, %1, %2" : "=v"(lbest) : "v"(lbest), "v"(nmx));   // (= gt ? nmx : lbest: one v_max instead of two selects; no canonicalisation of the operands)
                    lbr = gt ? r : lbr;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            bot_m = um; bot_s = us; bot_e = ue;
            if (DIR == 0) lbt = lbest > lbefore ? t : lbt;
            if (MODE) atomicMax(&ring[(unsigned)j & (RING - 1)], (unsigned long long)__double_as_longlong(crun));   // (scores >= 0 order like their bit patterns; 0 is a no-op)
            if (DIR == 0) {
                // ---- the step's codes: per row group one coalesced store
                PS_GLOBAL unsigned char* dst = codes + (size_t)t * (64 * K);
                int r0 = 0;
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const int sz = plane_sz(K - r0);
                    PS_GLOBAL unsigned char* p = dst + 64 * r0 + lane * sz;
                    if (sz == 16) {
                        v4i v;
                        v.x = cw[r0 / 4]; v.y = cw[r0 / 4 + 1]; v.z = cw[r0 / 4 + 2]; v.w = cw[r0 / 4 + 3];
                        *(PS_GLOBAL v4i*)p = v;
                    } else if (sz == 8) {
                        typedef int v2i __attribute__((ext_vector_type(2)));
                        v2i v;
                        v.x = cw[r0 / 4]; v.y = cw[r0 / 4 + 1];
                        *(PS_GLOBAL v2i*)p = v;
                    } else if (sz == 4) {
                        *(PS_GLOBAL unsigned*)p = cw[r0 / 4];
                    } else if (sz == 2) {
                        *(PS_GLOBAL unsigned short*)p = (unsigned short)(cw[r0 / 4] >> (8 * (r0 & 3)));
                    } else {
                        *p = (unsigned char)(cw[r0 / 4] >> (8 * (r0 & 3)));
                    }
                    r0 += sz;
                    if (r0 >= K) break;
                }
            }
            if (MODE && t >= next_flush) {
                // the maxima of completed columns: everything left of the column the highest strip in band is working on
                const int jdone = min(t - QHI[t], C + 1);
                for (int col = flushed + lane; col < jdone; col += 64) {
                    unsigned long long* e = &ring[(unsigned)col & (RING - 1)];
                    const unsigned long long v = *e;
                    *e = 0ull;
                    gcmax[col] = __longlong_as_double((long long)v);
                }
                flushed = max(flushed, jdone);
                next_flush = t + 64;
            }
        } else {
            bot_m = NINF; bot_s = NINF; bot_e = NINF;
        }
        a0 = a1; a1 = a2;
        ql0 = ql1; ql1 = ql2;
    }
    if (DIR == 0 && qcur >= 0 && lbest > 0.0) { StripBest sbv; sbv.v = lbest; sbv.i = qcur * K + 1 + lbr; sbv.j = lbt - qcur; SB[qcur] = sbv; }
    if (MODE)
        for (int col = flushed + lane; col <= C; col += 64) gcmax[col] = __longlong_as_double((long long)ring[(unsigned)col & (RING - 1)]);
}

// forward-only batches: one wave per job
template <int K, bool FD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K <= 10 ? 2 : 1, K <= 10 ? 2 : 1)))
void k_sweep(BatchD b, SweepD sw) {
    const JobD& J = b.jobs[blockIdx.x];
    if (J.out->inert) return;
    sweep_body<K, 0, 0, FD>(b, sw, J, sw.sj[blockIdx.x], nullptr);
}

// Alignment::update batches: one wave per (job, direction), sweep job jd = 2 * job + direction
template <int K, bool FD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K <= 10 ? 2 : 1, K <= 10 ? 2 : 1)))
void k_sweep2(BatchD b, SweepD sw) {
    __shared__ unsigned long long ring[RING];
    const int jd = blockIdx.x;
    const JobD& J = b.jobs[jd >> 1];
    if (J.out->inert) return;
    if ((jd & 1) == 0) sweep_body<K, 0, 1, FD>(b, sw, J, sw.sj[jd], ring);
    else sweep_body<K, 1, 1, FD>(b, sw, J, sw.sj[jd], ring);
}

// Alignment::update batches whose edit list reads few columns: the same sweeps with column-sparse records
template <int K, bool FD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K <= 10 ? 2 : 1, K <= 10 ? 2 : 1)))
void k_sweeps(BatchD b, SweepD sw) {
    __shared__ unsigned long long ring[RING];
    const int jd = blockIdx.x;
    const JobD& J = b.jobs[jd >> 1];
    if (J.out->inert) return;
    if ((jd & 1) == 0) sweep_body<K, 0, 2, FD>(b, sw, J, sw.sj[jd], ring);
    else sweep_body<K, 1, 2, FD>(b, sw, J, sw.sj[jd], ring);
}

// the global maximum and its first cell — smallest column, then smallest row (cpp/Alignment.cpp:158, 270: strict '>' over columns
// in order, rows in order) — from the per-strip records; one wave per job
__global__ __launch_bounds__(64) void k_best(BatchD b, SweepD sw) {
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
    static constexpr bool ROWFAST = true;   // consecutive threads take consecutive rows: K contiguous bytes per strip
    __device__ __forceinline__ void prep(int) {}
    __device__ __forceinline__ unsigned short word(int ti, int tj, int a, int c) const {
        const int r = ti - a, col = tj - c;
        if (r < 1 || col < 1) return (unsigned short)0xC000u;   // outside the matrix: score 0, the walk stops
        const int q = (r - 1) / K, rr = (r - 1) - q * K;
        const unsigned by = codes[(size_t)(col + q) * (64 * K) + code_off<K>(q & 63, rr)];
        const unsigned sm = by & 7u, ss = (by >> 3) & 3u;
        return (unsigned short)((sm == 7u ? (unsigned)M_IMPL : sm) | ((ss ? 3u + ss : 0u) << 8) | ((by & 0x60u) << 9));
    }
};

template <int K>
__global__ __launch_bounds__(256) void k_backtrace_s(BatchD b, SweepD sw) {
    const JobD& J = b.jobs[blockIdx.x];
    StripCodes<K> src;
    src.codes = sw.codes + sw.sj[blockIdx.x * sw.ndir].codes_off;
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

static const int K_LIST[] = {4, 6, 10, 16, 24, 32};
// smallest strip height whose window probably fits: band rows per step ~ (2W + 1) / (K + levels per base) + 2
int sweep_guess_k(int W) {
    for (int K : K_LIST) if ((2 * W + 1) / (K + 1) + 3 <= WIN_MAX) return K;
    return 0;
}
static int next_k(int K) { for (int k : K_LIST) if (k > K) return k; return 0; }

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

// bytes one job takes at strip height K: step codes (forward), and with full = true the records of both directions
double sweep_job_bytes(int n0, int C, int K, bool full) {
    const double steps = (double)C + (n0 + K - 1) / K + 1;
    return steps * 64.0 * K * (full ? 1.0 + 2 * 16.0 : 1.0);
}

// the strip tables of a batch at strip height K: sweep-job records (one per job and direction), band table, qlo / qhi + the
// widest window (read back by the caller).  bt.ndir == 2: also the record offsets of both directions (JobD.mat_off, JobD.K).
int sweep_prepare(Runtime* rt, Batch& bt, int K) {
    const BatchD& b = bt.d;
    const int nd = bt.ndir;
    std::vector<SweepJob>& sj = bt.sjobs;
    sj.resize(bt.jobs.size() * nd);
    int64_t band_tot = 0, q_tot = 0, sb_tot = 0, code_tot = 0, rec_tot = 0;
    int maxT = 0;
    for (size_t k = 0; k < bt.jobs.size(); k++) {
        JobD& j = bt.jobs[k];
        j.K = nd == 2 ? (bt.sparse ? -1 : K) : 0;
        if (bt.sparse) j.pitch = (std::min(2 * j.W + 1, j.n0) + 3) & ~1;   // rows of the widest band (+ slack), even: 32-byte aligned columns
        for (int d = 0; d < nd; d++) {
            SweepJob s;
            s.Q = (j.n0 + K - 1) / K;
            s.T = j.C + s.Q;                         // steps t = column + strip run 1 .. C + Q - 1
            s.band_off = band_tot; band_tot += j.C + 2;
            s.q_off = q_tot; q_tot += s.T + Q_PAD;
            s.sb_off = sb_tot; if (d == 0) sb_tot += std::max(s.Q, 1);
            s.codes_off = code_tot; if (d == 0) code_tot += (int64_t)s.T * 64 * K;
            if (nd == 2 && bt.sparse) { j.mat_off[d] = rec_tot; rec_tot += (int64_t)bt.nkeep[2 * k + d] * j.pitch; }
            else if (nd == 2) { j.mat_off[d] = rec_tot; rec_tot += (int64_t)s.T * 64 * K; }
            maxT = std::max(maxT, s.T);
            sj[k * nd + d] = s;
        }
    }
    bt.sweep_K = K; bt.sweep_maxT = maxT; bt.sweep_code_bytes = code_tot; bt.sweep_sb = sb_tot; bt.sweep_recs = rec_tot;
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
    const int K = bt.sweep_K;
    PS_HIP(hipMemsetAsync(sw.sb, 0, (size_t)std::max<int64_t>(bt.sweep_sb, 1) * sizeof(StripBest), rt->stream));
    prof_begin(rt);
    switch (K) {
        case 4: sweep_launch_k<4>(rt, b, sw); break;
        case 6: sweep_launch_k<6>(rt, b, sw); break;
        case 10: sweep_launch_k<10>(rt, b, sw); break;
        case 16: sweep_launch_k<16>(rt, b, sw); break;
        case 24: sweep_launch_k<24>(rt, b, sw); break;
        case 32: sweep_launch_k<32>(rt, b, sw); break;
        default: return fail(PS_ERR_BAD_ARG, "sweep_run: strip height");
    }
    PS_LAUNCH_CHECK();
    prof_end(rt, "sweep", 0.0);
    hipLaunchKernelGGL(k_best, dim3(b.njobs), dim3(64), 0, rt->stream, b, sw);
    if (bt.ndir == 2) PS_TRY(launch_prefix(rt, b, 2));   // running MaxInfo per column of both directions (the strip jobs' best cell is k_best's)
    switch (K) {
        case 4: bt_launch_k<4>(rt, b, sw); break;
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

int sweep_next_k(int K) { return next_k(K); }
int sweep_win_max() { return WIN_MAX; }

}  // namespace ps
