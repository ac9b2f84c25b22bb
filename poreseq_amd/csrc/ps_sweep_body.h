// ps_sweep_body.h — the strip sweep's device body, shared by ps_sweep.hip (one wavefront per sweep) and ps_sweepw.hip (two or
// four wavefronts per sweep).  See ps_sweep.hip for the geometry and the reference citations.
#ifndef PS_SWEEP_BODY_H_
#define PS_SWEEP_BODY_H_

#include "ps_dev.h"
#include "ps_host.h"
#include "ps_codes.h"

#ifndef PS_SWEEP_STRAIGHT
#define PS_SWEEP_STRAIGHT 1
#endif

namespace ps {

#ifdef __HIPCC__
// w = 2 w + (the lane's bit of a compare): the compare's lane mask is consumed where v_cmp left it (a scalar register pair)
__device__ __forceinline__ unsigned code_push(unsigned w, bool c) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(c);
    unsigned long long carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(w), "=s"(carry_out) : "v"(w), "s"(m));
    return w;
}
__device__ __forceinline__ unsigned code_first(bool c) {     // the first bit of a register
    const unsigned long long m = __builtin_amdgcn_ballot_w64(c);
    unsigned w;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(w) : "s"(m));
    return w;
}
#endif

struct StripBest { double v; int i, j; };
constexpr int Q_PAD = 8;          // qlo entries behind T (all -1): the sweep looks three steps ahead
// strips in band on one step: at most NL - 1, one lane stays idle (sweep_win_max, ps_sweep.hip: the lane above the lowest strip in band must hold
// an out-of-band strip, whose cells are the absent-cell value the band's top row reads as its upper neighbour)

__device__ __forceinline__ double wave_ror1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x13C /*wave_ror:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// ---- model rows through LDS (the north_star's staging, in the form that fits beside six sweeps per CU) ----
// A lane's column changes every step, so its 64-byte model row does too: straight from global memory that is four 16-byte gathers per
// lane and step with 64 different cache lines per instruction — 5 G L1 -> L2 requests and 43 GB of HBM / MALL fetch per 2 400-sweep launch
// (profiles/r05_c_sweep_pmc.md), a quarter of a CU's L1 tag bandwidth at six sweeps per CU.  But the columns in use on one step are a
// WINDOW (strip q works on column t - q: at most NL - 1 consecutive columns) that slides by at most one column per step: each column's
// row is fetched ONCE per sweep into a ring in LDS indexed by column (slot = column mod 2 NL, 80-byte pitch: eight consecutive slots
// cover all 32 banks, a 16-byte read per lane is conflict-free) by wave 0, sixteen columns per batch (64 lanes x 16 bytes), NL columns
// ahead of the newest column in use; the lanes read their rows from LDS.  640 KB of model rows per 10 kb sweep leave L2 instead of 100 MB.
#ifndef PS_MODEL_RING
#define PS_MODEL_RING 1
#endif
__host__ __device__ constexpr int mring_slots(int nw) { return 128 * nw; }          // 2 NL columns
__host__ __device__ constexpr int mring_bytes(int nw) { return PS_MODEL_RING ? (mring_slots(nw) + 1) * MODEL_ROW_BYTES : 16; }   // (+ a spare slot: f_write)

// columns of the LDS ring of per-column maxima: the widest window (NL - 1 strips) + the 64 steps between two flushes, rounded up
__host__ __device__ constexpr int ring_cols(int nw) { return nw <= 2 ? 256 : 512; }
constexpr int HAND_DOUBLES = 4;   // LDS hand-off record of a wave's last lane: {main, stay (+ emission), main + emission, -}

// One sweep on NW wavefronts (one workgroup of NL = 64 NW lanes; strip q sits on lane q mod NL).  DIR 0 / 1: forward / backward fill
// (cpp/Alignment.cpp:111-274 / 280-444: the backward cell adds its emission when it is LEFT, so what a lane hands down and keeps
// for the diagonal is {main, stay + emission, main + emission}).  MODE 0 (ScoreAlignments): only the forward step codes and the
// per-strip maxima leave the chip.  MODE 1 / 2: the sweep of an Alignment::update (ScoreMutations) — also the per-column maxima
// (MaxInfo, cpp/Alignment.cpp:158, 270) through an LDS ring to cmax, and {main, stay} records: of every cell,
// REC[step][row of the strip][lane] (MODE 1: one coalesced store per row and step), or only of the columns the edit list will read
// (MODE 2: scoreMutation / columnMax, cpp/Alignment.cpp:447-512, cpp/Alignment.h:181-214, read the forward columns max(start-4, 0)
// and max(start-3, 1) and two backward columns per edit): the lane whose column is kept (JobD.keep, fetched with the band record)
// writes its in-band cells to REC[kept column][row - i0].
//
// NW > 1: the cell above the first row of lane 0 of a wave is the last cell of lane 63 of the wave before it (cyclically).  Every
// wave's lane 63 leaves its {main, stay, main + emission} in LDS at the end of a step (`hand`, two slots alternating by step), the
// waves meet at one LDS-only barrier per step, and lane 0 picks the record up at the top of the next step — the read's latency is
// covered by the step's emissions, which do not depend on it.  Nothing else crosses waves: the ring of column maxima is shared
// (LDS atomics), every other table is per lane.
template <int K, int NW, int DIR, int MODE, bool FD>
__device__ __forceinline__ void sweep_body(const BatchD& b, const SweepD& sw, const JobD& J, const SweepJob& SJ, unsigned long long* ring, double* hand, char* mring) {
    constexpr int NL = 64 * NW;
    constexpr int RING = ring_cols(NW);
    constexpr int RM = mring_slots(NW);                            // slots of the model-row ring
    const int lane = threadIdx.x;                                  // lane of the sweep, 0 .. NL - 1
    const int wv = NW > 1 ? uni((int)threadIdx.x >> 6) : 0;       // its wave
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
    PS_GLOBAL double* gcmax = (PS_GLOBAL double*)uni_ptr(b.cmax + J.col_off[DIR]);
    StripBest* SB = sw.sb + SJ.sb_off;
    const double lsk = J.lsk, lst = J.lst, lex = J.lex, lin = J.lin, off = J.lik_offset, log2pi = b.log2pi;
    const double NINF = -__builtin_inf();

    // ---- what a lane fetches ahead of the step it is needed on
    struct Ahead { int p1, i0, i1, sc, kc, ro; };   // last band row of column j - 1, band rows and 5-mer of the lane's column j; kc: kept-column index (MODE 2); ro: byte offset of the column's model row in the LDS ring
    gcip keep = (gcip)uni_ptr(J.keep[DIR]);
    const int pitch = uni(J.pitch);
    // (every per-lane address below is a wave-uniform base in scalar registers + an unsigned 32-bit byte offset: the `saddr` form of
    //  the global instructions, one VALU instruction per address instead of four of 64-bit arithmetic)
    const PS_GLOBAL char* band_c = (const PS_GLOBAL char*)band;
    const PS_GLOBAL char* st_c = (const PS_GLOBAL char*)(DIR == 0 ? st - 1 : st);   // forward: entry j - 1 = the pair of columns j - 1, j
    const PS_GLOBAL char* keep_c = (const PS_GLOBAL char*)keep;
    const PS_GLOBAL char* levs_c = (const PS_GLOBAL char*)levs;
    auto fetch = [&](int tt, int ql) -> Ahead {
        const int q = ql + ((lane - ql) & (NL - 1));
        const int j = clampi(tt - q, 1, max(C, 1));   // (a sequence without a 5-mer has no live step; its prefetches still need an address)
        Ahead a;
        // {last row of column j - 1's band, first and last row of column j's}: three consecutive ints of the band table.  (The
        // 16-byte load of both columns' bands had a dead component, whose register the compiler reused while the load was in flight —
        // at the price of a full s_waitcnt vmcnt(0) in front of the reuse.)
        typedef int v3i_a4 __attribute__((ext_vector_type(3), aligned(4)));
        const v3i_a4 b3 = *(const PS_GLOBAL v3i_a4*)(band_c + (unsigned)(8 * j - 4));
        a.p1 = b3.x; a.i0 = b3.y; a.i1 = b3.z;
        // (forward column j holds states[j - 1], backward column j states[C - j]; ints of -1 around the list)
        a.sc = *(const PS_GLOBAL int*)(st_c + (unsigned)(4 * (DIR == 0 ? j : C - j)));
        a.kc = MODE == 2 ? *(const PS_GLOBAL int*)(keep_c + (unsigned)(4 * j)) : -1;
        a.ro = PS_MODEL_RING ? (j & (RM - 1)) * MODEL_ROW_BYTES : 0;
        return a;
    };
    auto model_row = [&](const Ahead& a, double (&m)[8]) {
        v2d q0, q1, q2, q3;
        if (PS_MODEL_RING) {
            const v2d* row = (const v2d*)(mring + a.ro);
            q0 = row[0]; q1 = row[1]; q2 = row[2]; q3 = row[3];
        } else {
            const PS_GLOBAL v2d* row = (const PS_GLOBAL v2d*)(model + (unsigned)max(a.sc, 0) * (unsigned)MODEL_ROW_BYTES);
            q0 = row[0]; q1 = row[1]; q2 = row[2]; q3 = row[3];
        }
        m[0] = q0.x; m[1] = q0.y; m[2] = q1.x; m[3] = q1.y; m[4] = q2.x; m[5] = q2.y; m[6] = q3.x; m[7] = q3.y;
    };

    // ---- the feeder of the model-row ring (wave 0; every decision below is wave-uniform).  Columns 1 .. jf are in the ring.
    // Steady state, per step: the first eight lanes load the row quarters of the next TWO columns at the top of the step (their 5-mer
    // states arrived as scalar loads issued the step before) and write them to LDS at the end of the same step, behind the cells —
    // nothing in flight is carried from step to step in vector registers (values that are would meet in register copies, i.e. in
    // full waits).  Two columns a step against at most one consumed: the ring runs ahead until the next column would overwrite its
    // slot's previous tenant (column - RM) before that one is older than the oldest column any strip can still be on — the newest
    // column in use, jn = t - qlo(t), minus the widest window, NL - 2 (both only grow while strips are in band: qlo never falls, and
    // a strip that has left the band does not come back) — i.e. NL columns ahead of jn.  At the start of a sweep and on the step
    // after a stretch without any strip in band, the columns the next step reads may not be there yet: all 64 lanes then load
    // sixteen columns at a time on the spot (two dependent round trips each: rare).
    const bool feeder = PS_MODEL_RING && wv == 0;
    typedef const __attribute__((address_space(4))) int* kst_t;
    kst_t st_k = (kst_t)uni_ptr(J.st);
    int jf = 0;                         // columns 1 .. jf are in the ring (visible behind the next barrier)
    int fs0 = -1, fs1 = -1, fsb = -1;   // 5-mer states of columns fsb + 1, fsb + 2 (scalar registers; fsb < 0: none loaded)
    auto st_of = [&](int c) -> int { return st_k[DIR == 0 ? c - 1 : C - c]; };   // (ints of -1 around the list)
    auto ring_at = [&](int c, int q16) -> v4i* { return (v4i*)(mring + (c <= C ? (c & (RM - 1)) : RM) * MODEL_ROW_BYTES + q16); };   // (behind the last column: a spare slot)
    auto feed_now = [&](int hi) {       // columns up to hi, at once
        const int fcol = (lane & 63) >> 2, fq16 = (lane & 3) * 16;
        const int lo = max(1, hi - (NL - 2));
        if (jf < lo - 1) jf = lo - 1;   // (nobody is on the columns in between any more)
        while (jf < hi) {
            const int c = jf + 1 + fcol, cc = min(c, max(C, 1));
            const int state = *(const PS_GLOBAL int*)(st_c + (unsigned)(4 * (DIR == 0 ? cc : C - cc)));
            const v4i row = *(const PS_GLOBAL v4i*)(model + (unsigned)max(state, 0) * (unsigned)MODEL_ROW_BYTES + (unsigned)fq16);
            *ring_at(c, fq16) = row;
            jf += 16;
        }
        fsb = -1;
    };
    v4i frow = {0, 0, 0, 0};
    bool fput = false;                  // (uniform) frow holds the row quarters of columns jf + 1, jf + 2: f_end writes them
    // top of step t.  hi_next: the newest column a lane reads a row of behind this step's barrier (0: none); jn_now: the newest
    // column in use on this step (0: none)
    auto f_top = [&](int hi_next, int jn_now) {
        if (!feeder) return;
        if (hi_next > 0 && jf < min(C, hi_next)) feed_now(min(C, hi_next));
        fput = false;
        if (jf < C && fsb == jf && jn_now > 0 && jf + 2 <= jn_now + NL + 1) {
            if ((lane & 63) < 8) {
                const int state = (lane & 4) ? fs1 : fs0;
                frow = *(const PS_GLOBAL v4i*)(model + (unsigned)max(state, 0) * (unsigned)MODEL_ROW_BYTES + (unsigned)((lane & 3) * 16));
            }
            fput = true;
        }
    };
    // end of step t (behind the cells: the loads of f_top have had the step to arrive)
    auto f_end = [&]() {
        if (!feeder) return;
        if (fput) {
            if ((lane & 63) < 8) *ring_at(jf + 1 + ((lane & 4) >> 2), (lane & 3) * 16) = frow;
            jf += 2;
        }
        if (jf < C && fsb != jf) { fsb = jf; fs0 = st_of(min(jf + 1, C)); fs1 = st_of(min(jf + 2, C)); }
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
        for (int k = lane; k < RING; k += NL) ring[k] = 0ull;
    }
    const int pw = NW > 1 ? (wv + NW - 1) % NW : 0;                // the wave whose lane 63 is the upper neighbour of this wave's lane 0
    const bool l0 = NW > 1 && (lane & 63) == 0, l63 = NW > 1 && (lane & 63) == 63;
    if (NW > 1) {
        if (l63) {
#pragma unroll
            for (int s = 0; s < 2; s++) { double* h = hand + (s * NW + wv) * HAND_DOUBLES; h[0] = NINF; h[1] = NINF; h[2] = NINF; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    int ql0 = QLO[1], ql1 = QLO[2], ql2 = QLO[3];   // qlo of step t, t + 1, t + 2 (scalar registers; T + Q_PAD entries, -1 behind T)
    Ahead aA = fetch(1, ql0), aB = fetch(2, ql1), aC = aB;   // three pipeline stages whose roles rotate with the step (the loop is unrolled by three)
    double mr[8];
    if (PS_MODEL_RING) {
        // the rows of step 1's columns (every lane's column is clamped into 1 .. C: column 1 where its strip has not started)
        if (feeder) { feed_now(min(max(C, 1), ql0 >= 0 ? max(1 - ql0, 1) : 1)); if (jf < C) { fsb = jf; fs0 = st_of(min(jf + 1, C)); fs1 = st_of(min(jf + 2, C)); } }
        if (NW > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    model_row(aA, mr);
    // The level records of the strip a lane works on at step t + 1 are fetched during step t, behind the step's emissions (the last
    // readers of the old ones): a strip's rows are read once per sweep, so the loads miss every cache, and issued at the top of
    // the step they are first used on they stalled the wave — and behind the step's barrier the whole workgroup — for the whole miss.
    int qlev = -1;                            // the strip whose level records are in `lev`
    auto load_levels = [&](int ql_next) {
        if (ql_next < 0) return;
        const int qn = ql_next + ((lane - ql_next) & (NL - 1));
        if (qn != qlev) {
            qlev = qn;
#pragma unroll
            for (int r = 0; r < K; r++) {
                const v4d v = *(const PS_GLOBAL v4d*)(levs_c + (unsigned)(32 * min(qn * K + r, n0 - 1)));
                lev[r][0] = v.x; lev[r][1] = v.y; lev[r][2] = v.z; lev[r][3] = v.w;
            }
        }
    };
    load_levels(ql0);

    // ---- stores of a step leave one step late, behind the next step's emissions.  A wave's vector-memory operations retire in
    // issue order, so a wait for a load (the model row of the next column) is also a wait for every store issued before it: with
    // the stores of a step issued in that step, every step ended in a drain of its own code bytes (~20 % of a lone wave's time).
    // Held back until the next step's loads have been waited for, a step's stores have a whole step to complete in.
    constexpr bool DEFER_REC = MODE == 2 && K <= 5;                 // (kept-column records: 4 K + 2 registers to hold them back)
    bool pend = false;                                               // (uniform) the previous step was live: it has stores waiting
    unsigned cwp[(K + 3) / 4];                                       // its codes, four to a register
    double pnm[DEFER_REC ? K : 1], pns[DEFER_REC ? K : 1];           // its kept-column records
    bool pkept = false;                                              // the lane's column of that step is kept and its strip meets the band
    unsigned poff = 0u;                                              // byte offset of the strip's first row in the record pool
#pragma unroll
    for (int w = 0; w < (K + 3) / 4; w++) cwp[w] = 0u;
    auto put_codes = [&](int tp, const unsigned (&cw)[(K + 3) / 4]) {   // the codes of step tp: per row group one coalesced store
        PS_GLOBAL unsigned char* dst = codes + (size_t)tp * (NL * K);      // (uniform)
        int r0 = 0;
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const int sz = plane_sz(K - r0);
            PS_GLOBAL unsigned char* p = dst + (unsigned)(NL * r0 + lane * sz);
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
#ifndef PS_CODES_NT
#define PS_CODES_NT 0
#endif
                // (PS_CODES_NT: the codes as non-temporal stores — 7 MB per sweep that only the backtrace reads, sparsely, a kernel later)
                if (PS_CODES_NT) __builtin_nontemporal_store(cw[r0 / 4], (PS_GLOBAL unsigned*)p);
                else *(PS_GLOBAL unsigned*)p = cw[r0 / 4];
            } else {
                // a register with nrow < 4 rows holds its first row in field nrow - 1: the plane's rows r0 .. r0 + sz - 1 are the sz
                // fields from field nrow - (r0 mod 4) - sz on, last row lowest
                const int nrow = K - (r0 & ~3) < 4 ? K - (r0 & ~3) : 4;
                const int sh = CODE_BITS * (nrow - (r0 & 3) - sz);
                // (the bits above the plane's fields are a neighbouring plane's or zero: the reader masks its field)
                if (sz == 2) *(PS_GLOBAL unsigned short*)p = (unsigned short)(cw[r0 / 4] >> sh);
                else *p = (unsigned char)(cw[r0 / 4] >> sh);
            }
            r0 += sz;
            if (r0 >= K) break;
        }
    };
    auto put_pending = [&](int tp) {
        if (!pend) return;
        if (DIR == 0) put_codes(tp, cwp);
        if (DEFER_REC) {
            if (pkept) {
                PS_GLOBAL char* recp = rec - 16 * (K - 1) + poff;
#pragma unroll
                for (int r = 0; r < K; r++) *(PS_GLOBAL v2d*)(recp + r * 16) = (v2d){pnm[r], pns[r]};
            }
        }
    };

    auto step = [&](const int t, const Ahead& a0, const Ahead& a1, Ahead& a2) __attribute__((always_inline)) {
        const int ql = ql0;
        const bool live = ql >= 0;                                   // (uniform over the workgroup) some strip is in band on this step
        const int q = ql + ((lane - ql) & (NL - 1));
        const int j = t - q;
        if (live && q != qcur) {
            // the lane takes its next strip (its level records are in place: load_levels of the step before): hand in the old one's maximum
            if (DIR == 0 && lbest > 0.0) { StripBest sbv; sbv.v = lbest; sbv.i = qcur * K + 1 + lbr; sbv.j = lbt - qcur; SB[qcur] = sbv; }
            lbest = 0.0;
            qcur = q;
#pragma unroll
            for (int r = 0; r < K; r++) { pm[r] = NINF; if (DIR) pe[r] = NINF; }
        }
        if (PS_MODEL_RING) f_top(ql1 >= 0 ? t + 1 - ql1 : 0, ql >= 0 ? t - ql : 0);   // (before the step's barrier: what it writes now is read behind it)
        // (qlo one step further ahead than it is needed: the scalar load's latency stays off the step.  With the model rows in LDS the
        //  emissions below wait for LDS reads — the same counter as scalar loads — so the load is issued behind them, further down)
        int ql3 = 0;
        if (!PS_MODEL_RING) ql3 = QLO[t + 3];
        a2 = fetch(t + 2, ql2);
        __builtin_amdgcn_sched_barrier(0);                           // (the step's loads are issued before its first emission waits for the model row)
        // emissions of the lane's K cells first: the model row is then free to receive the next column's (one step of lead).  NW > 1:
        // the waves meet between the emissions and the recurrences — the only part of a step that needs another wave's results — so
        // a wave that runs late by less than its emissions delays nobody; the hand-off record is read behind the barrier, under the
        // stores of the step before and the loads of the next column's model row
        double ov[K];
#pragma unroll
        for (int r = 0; r < K; r++) {
            ov[r] = emission8<FD>(mr, lev[r], log2pi, off);
            // (pinned here: left to itself the compiler sinks the emissions into the `live` block below, behind the next model row's
            //  loads — which then need registers of their own, a copy at the end of the step and a full `s_waitcnt vmcnt(0)` for it)
            asm volatile("" : "+v"(ov[r]));
            if (r & 1) __builtin_amdgcn_sched_barrier(0);            // two emissions in flight: enough to fill the pipe, few enough to stay in registers
        }
        __builtin_amdgcn_sched_barrier(0);
        double hm = NINF, hs = NINF, he = NINF;                      // (NW > 1) what the previous wave's lane 63 left at the end of step t - 1
        if (NW > 1) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const double* h = hand + (((t - 1) & 1) * NW + pw) * HAND_DOUBLES;
            const v2d hv = *(const v2d*)h;
            hm = hv.x; hs = hv.y;
            if (DIR) he = h[2];
        }
        put_pending(t - 1);
        load_levels(ql1);
        model_row(a1, mr);
        if (PS_MODEL_RING) ql3 = QLO[t + 3];
        pend = live;
        // (a step without a strip in band — the steps a last round of three adds behind T - 1 — runs the cells too, on no band: a uniform
        //  branch around them makes every loop-carried value a phi of two definitions, eight register copies per step)
        if ((PS_SWEEP_STRAIGHT && MODE != 1) || live) {   // (MODE 1 stores a record per cell and step: behind T - 1 there is no room for them)
            const int base = q * K + 1;
            const bool valid = live && j >= 1 && j <= C && a0.sc >= 0;     // (a column whose 5-mer is invalid is all zero: no cell takes part, cpp/Alignment.cpp:162-163)
            const int ra = valid ? a0.i0 - base : K, rb = valid ? a0.i1 - base : -1;   // band rows relative to the strip
            // (of the previous column's band only the last row is looked at: a cell tells "diagonal neighbour in the previous column's
            //  band" from the neighbour's value but for that row, below; the reader of the step codes looks the band up for MATCH
            //  against implicit MATCH)
            const int rp = a0.p1 - base;
            double um = wave_ror1(bot_m), us = wave_ror1(bot_s), ue = DIR ? wave_ror1(bot_e) : 0.0;
            if (NW > 1) { um = l0 ? hm : um; us = l0 ? hs : us; if (DIR) ue = l0 ? he : ue; }
            double dprev = dm, deprev = de;
            dm = um; de = ue;
            const double lbefore = lbest;
            double crun = 0.0;                                       // (MODE > 0) the lane's share of its column's maximum
            PS_GLOBAL char* recp = rec + ((size_t)t * K * NL + lane) * 16;
            const bool kept = MODE == 2 && live && a0.kc >= 0 && j >= 1 && j <= C;
            // (MODE 2) a lane whose column is kept stores all K rows of its strip when the strip meets the column's band: a row outside
            // the band lands in the K - 1 records of padding in front of / behind the column's band (sweep_prepare), which nobody reads
            const bool kstore = kept && ra < K && rb >= 0;
            if (MODE == 2) {
                // records of the lane's column: row i of the band [i0, i1] at REC[kc][i - i0]; recp = the strip's first row
                // (offsets from K - 1 records in front of the pool's first column, where its padding starts: never negative for a strip
                //  that meets the band; kept columns of one direction of one job: < 2^31 bytes)
                const unsigned ro = (unsigned)((a0.kc * pitch + (base - a0.i0) + (K - 1)) * 16);
                recp = rec - 16 * (K - 1) + ro;
                if (DEFER_REC) { poff = ro; pkept = kstore; }
                if (kept && !valid)   // a kept column without a 5-mer: its band reads as zeros (cpp/Alignment.cpp:162-163)
                    for (int r = max(0, a0.i0 - base); r <= min(K - 1, a0.i1 - base); r++) *(PS_GLOBAL v2d*)(recp + (size_t)r * 16) = (v2d){0.0, 0.0};
            }
            unsigned cw = 0u;                                        // code bytes of up to four rows, the first row's on top (code_push)
#pragma unroll
            for (int r = 0; r < K; r++) {
                const double o = ov[r];
                const bool act = r >= ra && r <= rb;
                const bool top = r == ra;
                // in band and not the band's first row (the compare of the row above serves: one per step more, none per cell)
                const bool below = r - 1 >= ra && r <= rb;
                const double pmr = pm[r];
                double L;
                asm("v_max_f64 %0, %1, 0" : "=v"(L) : "v"(pmr));
                // The diagonal neighbour (row i - 1 of the previous column) takes part when p0 < i <= p1, the previous column's band
                // (cpp/Alignment.cpp:207), else an implicit zero does.  A cell in band holds a main score >= 0 (every maximum starts from
                // 0); a row outside the band — or a whole column without a 5-mer, or the blank column — the absent-cell value, hugely
                // negative: that covers i - 1 < p0 and i - 1 > p1 by the neighbour's value alone.  Left is i = p1 + 1, whose neighbour
                // (the last row of the previous band) holds a score the reference does not use: one integer compare per cell (`blank`)
                // and one select of the high word make that neighbour absent too.
                //   forward   MATCH = diagonal + emission, implicit: the emission alone -> one maximum of the neighbour with 0 gives either
                //             (as for the left neighbour above); IGNORE = diagonal + lik_insert, implicit: the initial 0.0, which never wins
                //             the strict '>' against the floor 0 and never equals a positive maximum — the absent value + lik_insert does neither
                //   backward  (cpp/Alignment.cpp:384-394) MATCH = the neighbour's (main + emission), implicit: 0 — neutral beside the floor 0
                //             of the same maximum, as anything negative is: no step codes are made backward, only the values count
                const bool blank = rp == r - 1;
                const double dq = __hiloint2double(blank ? (int)0xFFEFFFFF : __double2hiint(dprev), __double2loint(dprev));
                double D = dq;
                if (DIR == 0) asm("v_max_f64 %0, %1, 0" : "=v"(D) : "v"(dq));
                const double eq_ = DIR ? __hiloint2double(blank ? (int)0xFFEFFFFF : __double2hiint(deprev), __double2loint(deprev)) : 0.0;
                const double cSTAY = DIR == 0 ? um + o + lst : ue + lst;      // backward: (main + emission) of the cell above
                const double cEXT = DIR == 0 ? us + o + lex : us + lex;       // backward: `us` carries stay + emission
                const double cINS = um + lin;
                const double cSKIP = L + lsk;
                const double cMATCH = DIR == 0 ? D + o : eq_;
                const double cIGN = dq + lin;
                // The stay floor of a band's first row is -1e300 (cpp/Alignment.cpp:230): a value that only ever loses; the other rows' is 0.
                // A first row has no upper neighbour in band: its STAY and EXTEND are built on the absent-cell value and lose to either
                // floor, so its stay score is the floor itself and its main score does not depend on which.  Where the records are kept
                // for a bit-exact comparison (MODE 1) the floor is -BIG as in the reference; elsewhere every row's floor is 0 and the first
                // row's stay score leaves the cell as the absent-cell value instead (`below`, next to the band mask): to the row below
                // and to columnMax (cpp/Alignment.h:181-214: stay + stay) it is the same "only ever loses" — one compare and one select
                // per cell less.
                const double floor_s = MODE == 1 ? (top ? -BIG : 0.0) : 0.0;
                const double t1 = fmax(floor_s, cSTAY);
                const double ns = fmax(t1, cEXT);
                double nm = fmax(0.0, cSKIP);
                nm = fmax(nm, cMATCH);
                nm = fmax(nm, cINS);
                nm = fmax(nm, cIGN);
                nm = fmax(nm, ns);
                if (DIR == 0) {
                    // the cell's code byte: stay matrix STAY then EXTEND with strict '>'; main matrix: which candidates equal the maximum (the
                    // reader takes the first in the reference's order); the two "score > 0" bits the walker stops on.  A row outside the band
                    // gets whatever its compares say: the reader knows the band.
                    cw = (r & 3) == 0 ? code_first(cEXT > t1) : code_push(cw, cEXT > t1);
                    cw = code_push(cw, ns > 0.0);
                    cw = code_push(cw, cSKIP == nm);
                    cw = code_push(cw, cMATCH == nm);
                    cw = code_push(cw, cINS == nm);
                    cw = code_push(cw, cIGN == nm);
                    cw = code_push(cw, nm > 0.0);
                    if ((r & 3) == 3 || r == K - 1) cwp[r >> 2] = cw;
                }
                // (a row outside the band carries the absent-cell value in both matrices)
                const double nmx = keep_or_absent(nm, act), nsx = keep_or_absent(ns, MODE == 1 ? act : below);
                dprev = pmr;
                pm[r] = nmx;
                if (DIR) { deprev = pe[r]; pe[r] = nmx + o; }
                um = nmx;
                us = DIR == 0 ? nsx : nsx + o;
                if (DIR) ue = nmx + o;
                if (MODE == 1) {
                    double rx;   // the stored record: the cell; zeros where there is none (the stay value of a top row is -1e300 and stays so)
                    asm("v_max_f64 %0, %1, 0" : "=v"(rx) : "v"(nmx));
                    *(PS_GLOBAL v2d*)(recp + (size_t)r * (16 * NL)) = (v2d){rx, act ? ns : 0.0};
                    crun = fmax(crun, rx);
                }
                if (MODE == 2) {
                    // (the masked values: a cell in band is stored as it is, a row outside the band lands in the column's padding)
                    if (DEFER_REC) { pnm[r] = nmx; pns[r] = nsx; }
                    else if (kstore) *(PS_GLOBAL v2d*)(recp + r * 16) = (v2d){nmx, nsx};
                    crun = fmax(crun, nmx);
                }
                if (DIR == 0) {
                    const bool gt = nmx > lbest;
                    asm("v_max_f64 %0, %1, %2" : "=v"(lbest) : "v"(lbest), "v"(nmx));   // (= gt ? nmx : lbest: one v_max instead of two selects; no canonicalisation of the operands)
                    lbr = gt ? r : lbr;
                }
#ifndef PS_SWEEP_ROW_FENCE
#define PS_SWEEP_ROW_FENCE 1
#endif
                if (PS_SWEEP_ROW_FENCE) __builtin_amdgcn_sched_barrier(0);
            }
            bot_m = um; bot_s = us; bot_e = ue;
            if (DIR == 0) lbt = lbest > lbefore ? t : lbt;
            if (MODE) atomicMax(&ring[(unsigned)j & (RING - 1)], (unsigned long long)__double_as_longlong(crun));   // (scores >= 0 order like their bit patterns; 0 is a no-op)
            if (MODE && live && t >= next_flush) {
                // the maxima of completed columns: everything left of the column the highest strip in band is working on (every wave's
                // contributions to them were made in earlier steps, i.e. before the barrier this step started behind)
                const int jdone = min(t - QHI[t], C + 1);
                for (int col = flushed + lane; col < jdone; col += NL) {
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
        if (NW > 1 && l63) {
            double* h = hand + ((t & 1) * NW + wv) * HAND_DOUBLES;
            *(v2d*)h = (v2d){bot_m, bot_s};
            if (DIR) h[2] = bot_e;
        }
        if (PS_MODEL_RING) f_end();
        ql0 = ql1; ql1 = ql2; ql2 = ql3;
    };

#ifndef PS_SWEEP_UNROLL_MAXK
#define PS_SWEEP_UNROLL_MAXK 5
#endif
    if (K <= PS_SWEEP_UNROLL_MAXK) {
        // steps 1 .. T - 1, three to a round so that the pipeline stages need no copies; the steps a last round adds behind T - 1 find
        // no strip in band (qlo = -1 there: Q_PAD) and do nothing.  (Taller strips keep one copy of the body: three of K = 10's
        // would be most of the instruction cache two CUs share.)
        for (int t = 1; t < T; t += 3) {
            step(t, aA, aB, aC);
            step(t + 1, aB, aC, aA);
            step(t + 2, aC, aA, aB);
        }
        put_pending(T - 1 + (3 - (T - 1) % 3) % 3);   // (the last step run: never live unless it is T - 1 itself)
    } else {
        for (int t = 1; t < T; t++) {
            step(t, aA, aB, aC);
            aA = aB; aB = aC;
        }
        put_pending(T - 1);
    }
    if (DIR == 0 && qcur >= 0 && lbest > 0.0) { StripBest sbv; sbv.v = lbest; sbv.i = qcur * K + 1 + lbr; sbv.j = lbt - qcur; SB[qcur] = sbv; }
    if (NW > 1 && MODE) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (every wave's last column maxima are in the ring)
    if (MODE)
        for (int col = flushed + lane; col <= C; col += NL) gcmax[col] = __longlong_as_double((long long)ring[(unsigned)col & (RING - 1)]);
}

// launchers of the multi-wave forms (ps_sweepw.hip); false: no such build
bool sweepw_launch(Runtime* rt, const BatchD& b, const SweepD& sw, int K, int NW);

}  // namespace ps
#endif
