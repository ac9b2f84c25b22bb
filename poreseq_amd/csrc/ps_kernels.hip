// ps_kernels.hip — gfx950 kernels for the banded event<->sequence DP and the edit scoring.
//
// Reference behaviour being reproduced (file:line under the reference tree):
//   updaterefs / getrefstate      cpp/EventData.h:110-183
//   fillColumn / fillColumnBack   cpp/Alignment.cpp:111-274 / :280-444
//   backtrace                     cpp/Alignment.cpp:516-624
//   scoreMutation / columnMax     cpp/Alignment.cpp:447-512, cpp/Alignment.h:181-214
//   ScoreMutations accumulation   cpp/MakeMutations.cpp:23-69
//
// Everything is FP64 and compiled with -ffp-contract=off: each cell is computed with the
// reference's operation order, so results are bit-identical to the CPU path (logs / pow are
// taken on the host with the same libm the reference uses).
//
// Execution model.  A DP cell (i, j) (level i, state column j) lives on anti-diagonal s = i + j.
// Storage is skewed: REC[s][i mod P].  One workgroup of P lanes owns one alignment and sweeps
// s = 2 .. n0 + C; lane `slot` owns the row i == slot (mod P) that is inside the band on that
// anti-diagonal (the band's footprint on one anti-diagonal is a contiguous run of at most
// 2W + 1 rows, because the band centre is a monotone function of the column; P is sized from the
// measured footprint, typically ~half the band).  Pipeline per batch of alignments:
//   k_lb / k_lo   band centres per column; lowest / highest in-band row + footprint per anti-diagonal
//   k_fill        emissions, band flags, recurrence, back-pointer codes and column maxima in one sweep
//                 (one 16-byte record + one 2-byte step word per cell leave the chip; nothing is re-read)
//   k_prefix      running MaxInfo per column, first cell of the global maximum
//   k_backtrace   LDS-tile walker with wave-wide look-ahead and tile prefetch, then k_fill_like, k_updaterefs /
//                 k_lb for the new band centres
//   k_old / k_score / k_reduce   edit scoring (scoreMutation + columnMax) and the per-edit sums
#include <algorithm>
#include <atomic>

#include "ps_internal.h"
#include "ps_slowmask.h"
#include "ps_dev.h"

namespace ps {

__device__ __forceinline__ int slot_of(int i, int P) { return i % P; }

// ------------------------------------------------------------------------------------------------
// updaterefs: ref_align -> ref_index, refstart, refend   (cpp/EventData.h:110-169)
// one 256-thread block per job; each thread owns a contiguous chunk of levels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_updaterefs(BatchD b) {
    chain_priority_wide();
    const JobD& J = b.jobs[blockIdx.x];
    JobOut* O = J.out;
    const int n = J.n0, tid = threadIdx.x;
    const double* __restrict__ ra = J.ra;
    double* __restrict__ ri = J.ri;
    __shared__ int s_first[256], s_last[256];
    const int chunk = (n + 255) / 256;
    const int t0 = min(n, tid * chunk), t1 = min(n, t0 + chunk);
    int first = 0x7fffffff, last = -1;
    for (int t = t0; t < t1; t++)
        if (ra[t] > 0) { if (first == 0x7fffffff) first = t; last = t; }
    s_first[tid] = first; s_last[tid] = last;
    __syncthreads();
    int a0 = 0x7fffffff, a1 = -1, prev = -1, next = 0x7fffffff;
    for (int k = 0; k < 256; k++) {
        a0 = min(a0, s_first[k]); a1 = max(a1, s_last[k]);
        if (k < tid) prev = max(prev, s_last[k]);
        if (k > tid) next = min(next, s_first[k]);
    }
    if (a1 < 0) {
        if (tid == 0) { O->has_index = 0; O->refstart = -1; O->refend = -1; }
        return;
    }
    if (tid == 0) { O->has_index = 1; O->refstart = (int)ra[a0]; O->refend = (int)ra[a1]; }
    const double slope = (ra[a1] - ra[a0]) / (double)(a1 - a0);
    const double icpt = ra[a0] - slope * (double)a0;
    int lastal = prev;
    for (int t = t0; t < t1; t++) {
        const double v = ra[t];
        if (t < a0 || t > a1) {
            ri[t] = slope * (double)t + icpt;
        } else if (v > 0) {
            ri[t] = v;
            lastal = t;
        } else {
            // inside [a0, a1], not aligned: interpolate between neighbours unless the left anchor is level 0 (sic)
            int nx = t + 1;
            while (nx < t1 && !(ra[nx] > 0)) nx++;
            if (nx >= t1) nx = next;
            if (lastal > 0) {
                const double mm = (ra[nx] - ra[lastal]) / (double)(nx - lastal);
                ri[t] = mm * (double)(t - lastal) + ra[lastal];
            } else {
                ri[t] = v;
            }
        }
    }
}

// lb[j] = getrefstate(j) for j = 0 .. lbn-1 (or -1 when ref_index is empty)
__global__ void k_lb(BatchD b, int which) {
    chain_priority_wide();
    const JobD& J = b.jobs[blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J.lbn) return;
    int* lb = b.lb + (which ? J.lbn_off : J.lb_off);
    lb[j] = J.out->has_index ? lower_bound_d(J.ri, J.n0, j) : -1;
}

// ------------------------------------------------------------------------------------------------
// lo[s] / hi[s]: lowest and highest in-band row on anti-diagonal s (-1: none).
// The columns present on s are those with i0(j) + j <= s <= i1(j) + j — a contiguous range
// [jlo, jhi] because both bounds are strictly increasing in j; rows are s - j, so lo = s - jhi,
// hi = s - jlo and the footprint hi - lo + 1 <= 2W + 1.  The per-job maximum footprint sizes P.
// Consecutive non-empty anti-diagonals move each end by at most one row (jhi and jlo grow by at most one per
// step), which is what lets k_fill derive every band flag from these two numbers and the lane's own history.
// Entries [S, S + LO_PAD) are -1: the fill pipeline looks a few anti-diagonals past the end.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lo(BatchD b, int ndir) {
    chain_priority_wide();
    const int jd = blockIdx.y, job = jd / ndir, dir = jd % ndir;
    const JobD& J = b.jobs[job];
    if (J.out->inert) return;
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int width = 0;
    if (s < J.S + LO_PAD) {
        const int* lb = b.lb + J.lb_off;
        int res = -1, top = -1;
        if (J.C >= 1 && s < J.S) {
            int i0, i1;
            band_of(lb, dir, 1, J.C, J.n0, J.W, i0, i1);
            if (i0 + 1 <= s) {
                int lo = 1, hi = J.C;  // largest j with i0(j) + j <= s
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    band_of(lb, dir, mid, J.C, J.n0, J.W, i0, i1);
                    if ((int64_t)i0 + mid <= s) lo = mid; else hi = mid - 1;
                }
                const int jhi = lo;
                band_of(lb, dir, J.C, J.C, J.n0, J.W, i0, i1);
                if ((int64_t)i1 + J.C >= s) {
                    lo = 1; hi = J.C;  // smallest j with i1(j) + j >= s
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        band_of(lb, dir, mid, J.C, J.n0, J.W, i0, i1);
                        if ((int64_t)i1 + mid >= s) hi = mid; else lo = mid + 1;
                    }
                    if (lo <= jhi) { res = (int)(s - jhi); top = (int)(s - lo); width = jhi - lo + 1; }
                }
            }
        }
        (b.lo + J.lo_off[dir])[s] = res;
        (b.hi + J.lo_off[dir])[s] = top;
    }
    for (int off = 32; off; off >>= 1) width = max(width, __shfl_xor(width, off));
    if ((threadIdx.x & 63) == 0 && width > 0) atomicMax(b.maxw, width);   // one P for the whole batch
}

// ------------------------------------------------------------------------------------------------
// k_fill: the whole of fillColumn / fillColumnBack (cpp/Alignment.cpp:111-274 / 280-444) for one alignment and
// direction in one pass — emission log-densities, band flags, the recurrence, the back-pointer codes (forward) and
// the per-column maxima — so a DP cell costs one 16-byte record (+ a 2-byte step word forward) of HBM traffic and
// nothing is re-read.
//
// One workgroup of P lanes per (job, direction) sweeps the anti-diagonals s = i + j.  Lane `slot` holds the row
// i == slot (mod P) that is inside [lo(s), lo(s) + P); its cell on anti-diagonal s is (i, s - i), so a lane walks
// along one row, one column per step, until lo passes the row, then takes row i + P.  What a cell needs:
//   (i, j-1)    the lane's own previous value (cm);
//   (i-1, j)    what the lane one slot up published on the previous anti-diagonal  -> two LDS exchange buffers,
//               written and read alternately, one LDS-only s_barrier per anti-diagonal;
//   (i-1, j-1)  what that lane published two anti-diagonals ago = what this lane read one step earlier (dm).
// A lane without a cell holds and publishes -infinity: read as the upper neighbour that is the top-row rule (no stay /
// extend / insert into a band's first row), read as the left neighbour max(., 0) turns it into the reference's
// "implicit zero" for neighbours outside the previous band (cpp/Alignment.cpp:201-225; real scores are >= 0).
// Band flags come from lo / hi and the lane's own history instead of per-cell band lookups:
//   in band        lo <= i <= hi
//   top row        i == lo on an anti-diagonal where a new column starts (lo did not move)
//   left valid     the lane's previous cell (same row, previous column) was in band, or this is column 1
//   diag valid     the reference tests ROW i against the previous band (p0 < i <= p1): previous cell in band and
//                  not that band's top row; the value is read as zero when the previous column is an invalid 5-mer
// Values are computed with v_max chains (the maximum does not depend on the reference's tie order); the forward
// step codes then follow from equality tests against the final value, which reproduce the ordered strict-'>'
// selection of cpp/Alignment.cpp:240-267 (the first candidate, in the reference's order, that equals the maximum).
//
// Divisions.  The three divisors of an emission (cpp/AlignUtil.h:34-53) are table values: the 5-mer's level stdv and sd
// mean, and the level's own stdv.  Their correctly rounded reciprocals y = RN(1/b) are tabulated (host IEEE division) and
// a/b is formed as  q0 = a*y; r = fma(-b,q0,a); q1 = fma(r,y,q0); r = fma(-b,q1,a); q = fma(r,y,q1)  — Markstein's
// sequence, whose last step is provably the correctly rounded quotient given y = RN(1/b) and a faithful q1 (Markstein 1990;
// Muller et al., Handbook of Floating-Point Arithmetic, section 4.7): bit-identical to the reference's IEEE division at 5
// instructions instead of ~13 (no v_rcp_f64 / v_div_*).  The host enables it per AlignData only when every divisor is a
// finite, normal, positive number of moderate magnitude; otherwise the kernel divides.  tests/test_fastdiv.py checks the
// sequence against IEEE division on 10^8 operand pairs, including all-ones and power-of-two significands.
//
// Software pipeline (everything that does not depend on a neighbour runs ahead of the recurrence):
//   windows, 4-step groups  5-mer states of the lane's next four columns (one 16-byte load) and the level record of its
//                           row, fetched at least three steps before first use.  A lane keeps its row for the whole window
//                           or idles: P exceeds the widest footprint by 9, so between two rows a lane idles 9 steps;
//   model row, 3 ahead      64-byte row {mean, 1/stdv, stdv, log stdv, sd mean, 1/sd mean, lambda, log lambda} of the
//                           column's 5-mer from LDS (64 KB per event);
//   emission, 2 ahead       ~30 FP64 instructions;
//   recurrence              exchange read, ~10 additions, max chain, codes, coalesced stores, exchange write.
// Bodies of 8 anti-diagonals near one on which the band resumes after an empty stretch (in practice: the start of the
// sweep) — rows may then jump by any amount and take a cell at once — run a SLOW variant that recomputes rows exactly
// and loads states / levels directly; every value is the same in both variants.
// Column maxima (MaxInfo per column) go through an LDS ring of ds_max and are flushed to memory as columns complete.
// ------------------------------------------------------------------------------------------------
// one bit per anti-diagonal of a 64-step chunk (lane k holds lo of step k, `before` = lo of the step before the chunk): the
// band resumes there after an empty stretch (lo(t-1) < 0 <= lo(t))
__device__ __forceinline__ unsigned long long resume_bits(int lov, int before) {
    const int prev = __builtin_amdgcn_update_dpp(before, lov, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return __ballot(lov >= 0 && prev < 0);
}
// resume_spread(): which 8-step bodies the resumes of one chunk make SLOW (ps_slowmask.h, host-tested)

constexpr int FB = 8;            // anti-diagonals per loop body
constexpr int FCH = 64;          // anti-diagonals per chunk: lo / hi prefetch unit, flush interval of the column maxima

template <int DIR>
struct FillState {
    double cm, cs;         // this lane's latest main / stay; -infinity while it has no cell
    double dm, de;         // upper neighbour's main (and, backward, main + emission) as read one step earlier
    double o1, o2;         // emission (+ lik_offset) of this lane's cell one / two anti-diagonals ahead
    bool pin, ptop, pdead; // the lane's previous anti-diagonal: had a cell position in band / was the band's top row / invalid 5-mer
    int row;               // the lane's row
    int stq[4];            // states of the lane's column on anti-diagonals s .. s + 3
    int stw[2][4];         // state windows (two groups in flight)
    double lv[2][4];       // level windows {level mean, level stdv, 3 log stdv, 1 / level stdv}
    double mr[8];          // model row of the lane's column on anti-diagonal s + 2 (read from LDS one step earlier)
};

struct FillCtx {
    PS_GLOBAL const v4d* lev;     // level records of this direction: [row - 1] = {mean, stdv, 3 log stdv, 1 / stdv}
    gcip st;
    gcip LO; gcip HI;
    PS_GLOBAL char* rec; PS_GLOBAL char* flg;   // anti-diagonal 0 of the job's matrix (uniform); lanes add their slot offset
    unsigned rec_off, flg_off;    // this lane's byte offset inside an anti-diagonal
    const char* mdl;              // LDS: model rows [NS][8]
    char* xch;                    // LDS: two exchange buffers of P records
    char* ring;                   // LDS: column maxima, ringmask + 1 bytes
    unsigned ringmask;
    int P, n0, C, slot, S;
    int wa[2], ra[2];             // byte offsets of this lane's / its upper neighbour's record in the two buffers
    double lsk, lst, lex, lin, off, log2pi;
};

template <bool FASTDIV>
__device__ __forceinline__ double fill_emission(const FillCtx& c, const double (&m)[8], const double (&lev)[4]) {
    return emission8<FASTDIV>(m, lev, c.log2pi, c.off);
}

// Model rows in LDS, two layouts:
//  * 80 bytes apart (the host's layout, copied as is).  With 64 the k-th quarter of every row falls on 4 of the 16 four-bank
//    groups (ds_read_b128 of 64 random rows: 71 % conflict cycles measured), with 80 on all 16.
//  * CMP (compact): 64 bytes apart, quarter k of row r stored in quarter k ^ ((r >> 2) & 3) — the same 16 groups in 64 KB
//    instead of 80, at 5 more VALU instructions per step.  With it a lone sweep of up to 384 lanes (256 backward) takes
//    at most half of a CU's 160 KB of LDS, so two workgroups share a CU.
template <bool CMP>
__device__ __forceinline__ void fill_model_row(const FillCtx& c, int state, double (&m)[8]) {
    const int st0 = state < 0 ? 0 : state;
    double2 a, bq, cq, dq;
    if (CMP) {
        const unsigned q0 = ((unsigned)st0 << 6) | (((unsigned)st0 << 2) & 0x30u);
        a = *(const double2*)(c.mdl + q0); bq = *(const double2*)(c.mdl + (q0 ^ 16u));
        cq = *(const double2*)(c.mdl + (q0 ^ 32u)); dq = *(const double2*)(c.mdl + (q0 ^ 48u));
    } else {
        const char* row = c.mdl + __umul24((unsigned)st0, (unsigned)MODEL_ROW_BYTES);   // v_mul_u32_u24: full rate (the compiler's v_mul_lo_u32 for (s << 6) + (s << 4) is not)
        a = *(const double2*)row; bq = *(const double2*)(row + 16); cq = *(const double2*)(row + 32); dq = *(const double2*)(row + 48);
    }
    m[0] = a.x; m[1] = a.y; m[2] = bq.x; m[3] = bq.y; m[4] = cq.x; m[5] = cq.y; m[6] = dq.x; m[7] = dq.y;
}

__device__ __forceinline__ void fill_levels(const FillCtx& c, int i, double (&lev)[4]) {
    const v4d v = c.lev[clampi(i, 1, c.n0) - 1];
    lev[0] = v.x; lev[1] = v.y; lev[2] = v.z; lev[3] = v.w;
}
// index of the state of the lane's column on anti-diagonal t, row i (out-of-range columns read the -1 padding;
// windows read [idx, idx + 3])
template <int DIR>
__device__ __forceinline__ int fill_state_index(const FillCtx& c, int t, int i) {
    const int j = t - i;
    const int idx = DIR == 0 ? j - 1 : c.C - j;
    return clampi(idx, -3, c.C - 1);
}
// row of residue class `slot` inside [lo, lo + P)
__device__ __forceinline__ int fill_row_of(const FillCtx& c, int lo) {
    int d = (c.slot - lo) % c.P;
    if (d < 0) d += c.P;
    return lo + d;
}

// One anti-diagonal.  PH = position in the loop body (compile time): exchange buffer, window set, state queue slot.
template <int DIR, int PH, bool SLOW, bool FASTDIV, bool CMP>
__device__ __forceinline__ void fill_step(FillState<DIR>& r, const FillCtx& c, const int s, const int lo_s, const int hi_s,
                                          const bool newcol /* uniform: a column starts on this anti-diagonal */) {
    const double NINF = -__builtin_inf();
    constexpr int G = PH >> 2, K = PH & 3;
    // ---- exchange read: what the lane one slot up published on the previous anti-diagonal
    double um, us, ue = 0.0;
    {
        const char* p = c.xch + c.ra[(PH & 1) ^ 1];
        if (DIR == 0) { const double2 u = *(const double2*)p; um = u.x; us = u.y; }
        else { um = *(const double*)p; us = *(const double*)(p + 8); ue = *(const double*)(p + 16); }
    }
    // ---- the lane's row and band flags on this anti-diagonal
    int i = r.row;
    {
        const bool behind = lo_s >= 0 && i < lo_s;
        if (!SLOW) i = behind ? i + c.P : i;     // lo moves one row per step at most: one slot's row falls out of [lo, lo + P)
        else if (behind) i = fill_row_of(c, lo_s);
        r.row = i;
    }
    const bool inb = lo_s >= 0 && i <= hi_s;
    const bool top = inb && newcol && i == lo_s;
    const bool first = s - i == 1;
    const bool dead = r.stq[K] < 0;
    // ---- windows: at a group start fetch the states / levels of anti-diagonals s + 6 .. s + 9 (a window = 4w .. 4w+3; s0 == 2 mod 8)
    if (K == 0) {
        const int t0 = s + 6;
        if (DIR == 0) {
            const int a = fill_state_index<0>(c, t0, i);           // ascending with the anti-diagonal
            const v4i v = *(const PS_GLOBAL v4i_a4*)(c.st + a);    // 4-byte aligned 16-byte load
            r.stw[G][0] = v.x; r.stw[G][1] = v.y; r.stw[G][2] = v.z; r.stw[G][3] = v.w;
        } else {
            const int a = fill_state_index<1>(c, t0 + 3, i);       // descending: element 3 - k belongs to step t0 + k
            const v4i v = *(const PS_GLOBAL v4i_a4*)(c.st + a);
            r.stw[G][0] = v.w; r.stw[G][1] = v.z; r.stw[G][2] = v.y; r.stw[G][3] = v.x;
        }
        fill_levels(c, i, r.lv[G]);
    }
    // ---- emission of anti-diagonal s + 2: its model row arrived during the previous step; levels from the window the
    //      previous group start fetched (set G ^ 1)
    double o_new;
    {
        if (!SLOW) o_new = fill_emission<FASTDIV>(c, r.mr, r.lv[G ^ 1]);
        else {
            const int lo2 = (s + 2 >= 0 && s + 2 < c.S) ? c.LO[s + 2] : -1;
            double lev[4];
            fill_levels(c, lo2 >= 0 ? fill_row_of(c, lo2) : i, lev);
            o_new = fill_emission<FASTDIV>(c, r.mr, lev);
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // stage fence: with three waves per SIMD the other waves hide latency; keeping the
                                         // stages apart keeps their temporaries from piling up (168 VGPRs for 12 waves per CU)
    // ---- model row of anti-diagonal s + 3: its state is element (PH + 1) & 3 of the window [4w, 4w+3] holding s + 3,
    //      fetched at body offset 4w - 6: the previous body's second group (set 1), this body's first (0) or second (1).
    //      (Handing the rows down the lanes with DPP instead of gathering them — lane L+1 needs on step s+1 the row lane L
    //      used on step s — removes the LDS bank conflicts of the gather, 71 % of the LDS cycles, but costs 16 more VALU
    //      instructions per step and the kernel is VALU-issue bound: measured 13.3 ms against 12.2 ms per 10 kb fill.)
    {
        constexpr int SET = (PH <= 2 || PH == 7) ? 1 : 0;
        int state = r.stw[SET][(PH + 1) & 3];
        if (SLOW) {
            const int lo3 = (s + 3 >= 0 && s + 3 < c.S) ? c.LO[s + 3] : -1;
            state = c.st[fill_state_index<DIR>(c, s + 3, lo3 >= 0 ? fill_row_of(c, lo3) : i)];
        }
        r.stq[(PH + 3) & 3] = state;
        fill_model_row<CMP>(c, state, r.mr);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- recurrence on anti-diagonal s
    {
        const bool vd = first || (r.pin && !r.ptop);
        const bool rd = !first && r.pin && !r.ptop && !r.pdead;
        const double o = r.o1;
        double L;   // max(cm, 0) in one instruction (fmax() would first canonicalise its operand)
        asm("v_max_f64 %0, %1, 0" : "=v"(L) : "v"(r.cm));
        const double D = rd ? r.dm : 0.0;
        const double cSTAY = DIR == 0 ? um + o + c.lst : ue + c.lst;       // backward: (main + emission) of the cell above
        const double cEXT = DIR == 0 ? us + o + c.lex : us + c.lex;         // backward: `us` carries stay + emission
        const double cINS = um + c.lin;
        const double cSKIP = L + c.lsk;
        const double cMATCH = DIR == 0 ? D + o : (rd ? r.de : 0.0);
        const double cIGN = D + c.lin;
        const double floor_s = top ? -BIG : 0.0;
        double ns = fmax(floor_s, cSTAY);
        ns = fmax(ns, cEXT);
        double nm = fmax(0.0, cSKIP);
        nm = fmax(nm, cMATCH);
        nm = fmax(nm, cINS);
        nm = fmax(nm, cIGN);
        nm = fmax(nm, ns);
        const bool act = inb && !dead;
        r.cm = act ? nm : NINF;
        r.cs = act ? ns : NINF;
        // stored record: the cell; zeros for a cell of an invalid-5-mer column (cpp/Alignment.cpp:162-163) and, harmlessly,
        // for slots outside the band (nm >= 0, and the stay value of a top row is -1e300 and must stay so)
        double rx;
        asm("v_max_f64 %0, %1, 0" : "=v"(rx) : "v"(r.cm));
        // (a half whose sweep is shorter than its partner's keeps stepping: its idle records go to the last padding row)
        const uint64_t sP = (uint64_t)(unsigned)(min(s, c.S + MAT_BACK - 1) + MAT_FRONT) * (unsigned)c.P;   // uniform
        *(PS_GLOBAL v2d*)(c.rec + sP * 16 + c.rec_off) = (v2d){rx, act ? ns : 0.0};
        if (DIR == 0) {
            // back-pointer codes: stay matrix STAY then EXTEND with strict '>', main matrix in the reference's order
            unsigned ss = cSTAY > floor_s ? M_STAY : 0u;
            ss = cEXT > fmax(floor_s, cSTAY) ? M_EXTEND : ss;
            unsigned sm = M_STAY;
            sm = cIGN == nm ? M_IGNORE : sm;
            sm = cINS == nm ? M_INSERT : sm;
            sm = cMATCH == nm ? (vd ? M_MATCH : M_IMPL) : sm;
            sm = cSKIP == nm ? M_SKIP : sm;
            sm = nm > 0.0 ? sm : 0u;
            // bits 14 / 15: main / stay score <= 0 (the backtrace stops there, cpp/Alignment.cpp:542)
            const unsigned w = sm | (ss << 8) | (nm > 0.0 ? 0u : 0x4000u) | (ns > 0.0 ? 0u : 0x8000u);
            *(PS_GLOBAL unsigned short*)(c.flg + sP * 2 + c.flg_off) = (unsigned short)(act ? w : FLG_DEAD);
        }
        // publish for the lane one slot down
        char* q = c.xch + c.wa[PH & 1];
        if (DIR == 0) {
            *(double2*)q = make_double2(r.cm, r.cs);
        } else {
            *(double*)q = r.cm; *(double*)(q + 8) = r.cs + o; *(double*)(q + 16) = r.cm + o;
        }
        // column maximum (scores are >= 0: their bit patterns order like unsigned integers; 0 is a no-op)
        atomicMax((unsigned long long*)(c.ring + (((unsigned)(s - i) << 3) & c.ringmask)), (unsigned long long)__double_as_longlong(rx));
        r.dm = um; r.de = ue;
        r.pin = inb; r.ptop = top; r.pdead = dead;
    }
    r.o1 = r.o2; r.o2 = o_new;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int DIR, bool SLOW, bool FASTDIV, bool CMP>
__device__ __forceinline__ void fill_group8(FillState<DIR>& r, const FillCtx& c, const int s0, const int (&lov)[FB], const int (&hiv)[FB], const int lo_p) {
#define PS_NEWCOL(k, prev) ((prev) < 0 || lov[k] == (prev))
    fill_step<DIR, 0, SLOW, FASTDIV, CMP>(r, c, s0 + 0, lov[0], hiv[0], PS_NEWCOL(0, lo_p));
    fill_step<DIR, 1, SLOW, FASTDIV, CMP>(r, c, s0 + 1, lov[1], hiv[1], PS_NEWCOL(1, lov[0]));
    fill_step<DIR, 2, SLOW, FASTDIV, CMP>(r, c, s0 + 2, lov[2], hiv[2], PS_NEWCOL(2, lov[1]));
    fill_step<DIR, 3, SLOW, FASTDIV, CMP>(r, c, s0 + 3, lov[3], hiv[3], PS_NEWCOL(3, lov[2]));
    fill_step<DIR, 4, SLOW, FASTDIV, CMP>(r, c, s0 + 4, lov[4], hiv[4], PS_NEWCOL(4, lov[3]));
    fill_step<DIR, 5, SLOW, FASTDIV, CMP>(r, c, s0 + 5, lov[5], hiv[5], PS_NEWCOL(5, lov[4]));
    fill_step<DIR, 6, SLOW, FASTDIV, CMP>(r, c, s0 + 6, lov[6], hiv[6], PS_NEWCOL(6, lov[5]));
    fill_step<DIR, 7, SLOW, FASTDIV, CMP>(r, c, s0 + 7, lov[7], hiv[7], PS_NEWCOL(7, lov[6]));
#undef PS_NEWCOL
}

constexpr int FILL_MODEL_BYTES = MODEL_ROW_BYTES * NS;

// One half of a k_fill workgroup: the sweep of one (job, direction).  `slot` is the thread's lane inside the half, `hsm` the
// half's private LDS (exchange buffers, ring, bitmap), `model` the workgroup's shared model rows, Smax the longer of the two
// sweeps of the workgroup (both halves execute the same number of barriers).
template <int DIR, bool FASTDIV, bool CMP>
__device__ __forceinline__ void fill_body(const BatchD& b, const JobD& J, const char* model, char* hsm, const int slot, const int Smax,
                                          const int rcols, const int slowwords) {
    const int P = uni(J.P);
    FillCtx c;
    c.lev = (PS_GLOBAL const v4d*)uni_ptr(J.lev[DIR]); c.st = (gcip)uni_ptr(J.st);
    c.LO = (gcip)uni_ptr(b.lo + J.lo_off[DIR]); c.HI = (gcip)uni_ptr(b.hi + J.lo_off[DIR]);
    c.rec = (PS_GLOBAL char*)uni_ptr(b.rec + J.mat_off[DIR] - (int64_t)MAT_FRONT * P); c.flg = (PS_GLOBAL char*)uni_ptr(b.flg + J.mat_off[DIR] - (int64_t)MAT_FRONT * P);
    c.rec_off = (unsigned)slot * 16u; c.flg_off = (unsigned)slot * 2u;
    c.P = P; c.n0 = uni(J.n0); c.C = uni(J.C); c.slot = slot;
    c.lsk = J.lsk; c.lst = J.lst; c.lex = J.lex; c.lin = J.lin; c.off = J.lik_offset; c.log2pi = b.log2pi;
    constexpr int RB = DIR ? 24 : 16;   // bytes per exchanged record: {main, stay} / {main, stay + em, main + em}
    c.mdl = model;
    c.xch = hsm;
    c.ring = c.xch + 2 * P * (CMP ? RB : 24);   // (a half of a pair may run either direction: sized for the larger record)
    c.ringmask = (unsigned)rcols * 8u - 1u;
    unsigned* slowmap = (unsigned*)(c.ring + rcols * 8);   // (!CMP) one bit per 8-step body: it runs the SLOW variant
    for (int k = slot; k < rcols; k += P) ((unsigned long long*)c.ring)[k] = 0ull;
    if (!CMP) for (int k = slot; k < slowwords; k += P) slowmap[k] = 0u;
    const int up_slot = slot == 0 ? P - 1 : slot - 1;
    const double NINF = -__builtin_inf();
#pragma unroll
    for (int k = 0; k < 2; k++) {
        c.wa[k] = (k * P + slot) * RB;
        c.ra[k] = (k * P + up_slot) * RB;
        *(double*)(c.xch + c.wa[k]) = NINF; *(double*)(c.xch + c.wa[k] + 8) = NINF;
        if (DIR) *(double*)(c.xch + c.wa[k] + 16) = NINF;
    }
    const int S = uni((int)J.S);                         // this half's sweep; the loops below run to Smax >= S (idle steps past S)
    const int s_first = 2 - FB;                          // the pipeline needs a few steps to fill; they fall into the front padding
    c.S = S;
    __syncthreads();
    // Bodies that must run the SLOW variant: the band resumes on anti-diagonal t (lo(t-1) < 0 <= lo(t)).  Windows fetched on
    // t - 9 .. t - 1 hold the states / levels of rows that may no longer be the lane's; they are consumed on t - 6 .. t + 6:
    // every body holding a step of [t - 6, t + 6] is slow.  The plain layout marks them in an LDS bitmap up front; the compact
    // one has no LDS to spare and derives them from the lo chunks as they stream by (resume_spread).  (One mechanism would do,
    // but the second costs the 168-VGPR pair build 70 spilled registers and 4 % — register allocation, not logic.)
    if (!CMP)
        for (int t = 2 + slot; t < S; t += P) {
            if (c.LO[t] >= 0 && c.LO[t - 1] < 0) {
                const int b0 = max(0, (t - 6 - s_first) >> 3), b1 = (t + 6 - s_first) >> 3;
                for (int bb = b0; bb <= b1; bb++) atomicOr(&slowmap[bb >> 5], 1u << (bb & 31));
            }
        }
    FillState<DIR> r;
    r.cm = NINF; r.cs = NINF; r.dm = NINF; r.de = NINF; r.o1 = 0.0; r.o2 = 0.0;
    r.pin = false; r.ptop = false; r.pdead = false;
    r.row = slot == 0 ? P : slot;   // rows start at 1: the smallest row of this slot's residue class
#pragma unroll
    for (int k = 0; k < 4; k++) r.stq[k] = 0;
#pragma unroll
    for (int g = 0; g < 2; g++) {
#pragma unroll
        for (int k = 0; k < 4; k++) r.stw[g][k] = 0;
        r.lv[g][0] = 0.0; r.lv[g][1] = 1.0; r.lv[g][2] = 0.0; r.lv[g][3] = 1.0;
    }
    r.mr[0] = 0.0; r.mr[1] = 1.0; r.mr[2] = 1.0; r.mr[3] = 0.0; r.mr[4] = 1.0; r.mr[5] = 1.0; r.mr[6] = 0.0; r.mr[7] = 0.0;
    __syncthreads();

    int flushed = 1;                                     // columns below this have their maximum in memory
    double* gcmax = b.cmax + J.col_off[DIR];
    int lo_p = -1;                                       // lo of the previous anti-diagonal
    const int lane = slot & 63;
    // lo / hi of 64 anti-diagonals per VGPR (lane k holds entry k), fetched one chunk ahead
    int loc = c.LO[max(s_first + lane, 0)], hic = c.HI[max(s_first + lane, 0)];
    if (s_first + lane < 0) { loc = -1; hic = -1; }
    // `slow` (CMP): bits 0-7 the bodies of the current chunk, above them what is known of the chunks after (one scalar register)
    unsigned slow = CMP ? __builtin_amdgcn_readfirstlane(resume_spread(resume_bits(loc, -1)) >> 8) : 0u;
    int nbody = 0;
    for (int cb = s_first; cb < Smax; cb += FCH) {
        const int lon = c.LO[min(cb + FCH + lane, S + LO_PAD - 1)], hin = c.HI[min(cb + FCH + lane, S + LO_PAD - 1)];
        if (CMP) slow |= __builtin_amdgcn_readfirstlane(resume_spread(resume_bits(lon, __builtin_amdgcn_readlane(loc, 63))));
#pragma unroll 1
        for (int o = 0; o < FCH && cb + o < Smax; o += FB, nbody++) {
            const int s0 = cb + o;
            int lov[FB], hiv[FB];
#pragma unroll
            for (int k = 0; k < FB; k++) { lov[k] = __builtin_amdgcn_readlane(loc, o + k); hiv[k] = __builtin_amdgcn_readlane(hic, o + k); }
            const unsigned sw = CMP ? slow >> (o >> 3) : __builtin_amdgcn_readfirstlane(slowmap[nbody >> 5]) >> (nbody & 31);
            if (sw & 1u) fill_group8<DIR, true, FASTDIV, CMP>(r, c, s0, lov, hiv, lo_p);
            else fill_group8<DIR, false, FASTDIV, CMP>(r, c, s0, lov, hiv, lo_p);
            lo_p = lov[FB - 1];
        }
        loc = lon; hic = hin;
        slow >>= 8;
        // flush the maxima of completed columns: every column left of the oldest one still present on the next anti-diagonal
        {
            const int sn = cb + FCH;
            const int lo_n = __builtin_amdgcn_readlane(loc, 0), hi_n = __builtin_amdgcn_readlane(hic, 0);
            if (sn < S && lo_n >= 0) {
                const int jdone = sn - hi_n;             // oldest column still present
                for (int col = flushed + slot; col < jdone; col += P) {
                    unsigned long long* e = (unsigned long long*)(c.ring + (((unsigned)col * 8u) & c.ringmask));
                    const unsigned long long v = *e;
                    *e = 0ull;
                    gcmax[col] = __longlong_as_double((long long)v);
                }
                flushed = max(flushed, jdone);
            }
        }
    }
    __syncthreads();
    for (int col = flushed + slot; col <= J.C; col += P)
        gcmax[col] = __longlong_as_double((long long)*(unsigned long long*)(c.ring + (((unsigned)col * 8u) & c.ringmask)));
}

// ------------------------------------------------------------------------------------------------
// k_fill_wide: the same sweep for bands whose footprint does not fit one lane per row (more than 1015 rows on an anti-diagonal:
// realign_width beyond ~500 in the worst case, ~980 for events with one level per base).  Every thread holds TWO slots (t and
// t + blockDim), so a workgroup of up to 1024 threads carries up to 2048 slots (realign_width <= 1022 for any input).  This is the
// plain form of k_fill — no prefetch windows, no software pipeline, model rows read from global memory (two 98 KB exchange buffers
// leave no room for the table in LDS), one sweep per workgroup — i.e. the width is supported, not tuned: ~4x the time per cell.
// Cell arithmetic, flags, codes and column maxima are the same expressions as in fill_step.
// ------------------------------------------------------------------------------------------------
template <int DIR, bool FD>
__device__ __forceinline__ void fill_wide_body(const BatchD& b, const JobD& J, char* smem, const int rcols) {
    constexpr int R = 2;
    const int T = blockDim.x, P = J.P, tid = threadIdx.x, C = J.C, S = (int)J.S;
    const int* __restrict__ LO = b.lo + J.lo_off[DIR];
    const int* __restrict__ HI = b.hi + J.lo_off[DIR];
    double2* __restrict__ rec = b.rec + J.mat_off[DIR];
    unsigned short* __restrict__ flg = b.flg + J.mat_off[DIR];
    const double4* __restrict__ levs = (const double4*)J.lev[DIR];
    const int* __restrict__ st = J.st;
    const char* __restrict__ model = (const char*)J.model8;
    const double lsk = J.lsk, lst = J.lst, lex = J.lex, lin = J.lin, off = J.lik_offset, log2pi = b.log2pi;
    const double NINF = -__builtin_inf();
    double* xch = (double*)smem;                                  // [2][P][3]
    unsigned long long* ring = (unsigned long long*)(smem + (size_t)2 * P * 24);
    const unsigned ringmask = (unsigned)rcols - 1u;
    for (int k = tid; k < 2 * P * 3; k += T) xch[k] = NINF;
    for (int k = tid; k < rcols; k += T) ring[k] = 0ull;
    double cm[R], cs[R], dm[R], de[R];
    int row[R];
    bool pin[R], ptop[R], pdead[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int slot = tid + r * T;
        cm[r] = NINF; cs[r] = NINF; dm[r] = NINF; de[r] = NINF;
        row[r] = slot == 0 ? P : slot;
        pin[r] = false; ptop[r] = false; pdead[r] = false;
    }
    __syncthreads();
    double* gcmax = b.cmax + J.col_off[DIR];
    int lo_p = -1, flushed = 1;
    for (int s = 2; s < S; s++) {
        const int lo = LO[s], hi = HI[s];
        const bool newcol = lo_p < 0 || lo == lo_p;
        const int par = s & 1;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int slot = tid + r * T;
            int i = row[r];
            if (lo >= 0 && i < lo) {
                i += P;
                if (i < lo) { int d = (slot - lo) % P; if (d < 0) d += P; i = lo + d; }
            }
            row[r] = i;
            const bool inb = lo >= 0 && i <= hi;
            const bool top = inb && newcol && i == lo;
            const int j = s - i;
            const bool first = j == 1;
            const double* up = xch + ((size_t)(par ^ 1) * P + (slot == 0 ? P - 1 : slot - 1)) * 3;
            const double um = up[0], us = up[1], ue = up[2];
            double o = 0.0;
            bool dead = false;
            if (inb) {
                const int state = st[DIR == 0 ? j - 1 : C - j];
                dead = state < 0;
                if (!dead) {
                    const double2* q = (const double2*)(model + (size_t)state * MODEL_ROW_BYTES);
                    const double2 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                    const double m[8] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
                    const double4 l4 = levs[i - 1];
                    const double lev[4] = {l4.x, l4.y, l4.z, l4.w};
                    o = emission8<FD>(m, lev, log2pi, off);
                }
            }
            const bool vd = first || (pin[r] && !ptop[r]);
            const bool rd = !first && pin[r] && !ptop[r] && !pdead[r];
            double L;
            asm("v_max_f64 %0, %1, 0" : "=v"(L) : "v"(cm[r]));
            const double D = rd ? dm[r] : 0.0;
            const double cSTAY = DIR == 0 ? um + o + lst : ue + lst;
            const double cEXT = DIR == 0 ? us + o + lex : us + lex;
            const double cINS = um + lin;
            const double cSKIP = L + lsk;
            const double cMATCH = DIR == 0 ? D + o : (rd ? de[r] : 0.0);
            const double cIGN = D + lin;
            const double floor_s = top ? -BIG : 0.0;
            double ns = fmax(floor_s, cSTAY);
            ns = fmax(ns, cEXT);
            double nm = fmax(0.0, cSKIP);
            nm = fmax(nm, cMATCH);
            nm = fmax(nm, cINS);
            nm = fmax(nm, cIGN);
            nm = fmax(nm, ns);
            const bool act = inb && !dead;
            cm[r] = act ? nm : NINF;
            cs[r] = act ? ns : NINF;
            double rx;
            asm("v_max_f64 %0, %1, 0" : "=v"(rx) : "v"(cm[r]));
            const int64_t cell = (int64_t)s * P + slot;
            rec[cell] = make_double2(rx, act ? ns : 0.0);
            if (DIR == 0) {
                unsigned ss = cSTAY > floor_s ? M_STAY : 0u;
                ss = cEXT > fmax(floor_s, cSTAY) ? M_EXTEND : ss;
                unsigned sm = M_STAY;
                sm = cIGN == nm ? M_IGNORE : sm;
                sm = cINS == nm ? M_INSERT : sm;
                sm = cMATCH == nm ? (vd ? M_MATCH : M_IMPL) : sm;
                sm = cSKIP == nm ? M_SKIP : sm;
                sm = nm > 0.0 ? sm : 0u;
                const unsigned w = sm | (ss << 8) | (nm > 0.0 ? 0u : 0x4000u) | (ns > 0.0 ? 0u : 0x8000u);
                flg[cell] = (unsigned short)(act ? w : FLG_DEAD);
            }
            double* mine = xch + ((size_t)par * P + slot) * 3;
            if (DIR == 0) { mine[0] = cm[r]; mine[1] = cs[r]; }
            else { mine[0] = cm[r]; mine[1] = cs[r] + o; mine[2] = cm[r] + o; }
            if (act && rx > 0.0) atomicMax(&ring[(unsigned)j & ringmask], (unsigned long long)__double_as_longlong(rx));
            dm[r] = um; de[r] = ue;
            pin[r] = inb; ptop[r] = top; pdead[r] = dead;
        }
        lo_p = lo;
        __syncthreads();
        if ((s & 63) == 63 && s + 1 < S) {   // flush the maxima of completed columns (left of the oldest column still present)
            const int lo_n = LO[s + 1];
            if (lo_n >= 0) {
                const int jdone = s + 1 - HI[s + 1];
                for (int col = flushed + tid; col < jdone; col += T) {
                    const unsigned long long v = ring[(unsigned)col & ringmask];
                    ring[(unsigned)col & ringmask] = 0ull;
                    gcmax[col] = __longlong_as_double((long long)v);
                }
                flushed = max(flushed, jdone);
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int col = flushed + tid; col <= C; col += T) gcmax[col] = __longlong_as_double((long long)ring[(unsigned)col & ringmask]);
}

template <bool FD>
__global__ __launch_bounds__(1024) void k_fill_wide(BatchD b, int ndir, int rcols) {
    extern __shared__ double2 fill_smem[];
    const int jd = blockIdx.x, job = jd / ndir, dir = jd % ndir;
    const JobD& J = b.jobs[job];
    if (J.out->inert) return;
    if (dir == 0) fill_wide_body<0, FD>(b, J, (char*)fill_smem, rcols); else fill_wide_body<1, FD>(b, J, (char*)fill_smem, rcols);
}

// the barrier sequence of fill_body for a half without a sweep (no partner, or an inert job)
__device__ __forceinline__ void fill_idle(const int Smax) {
    __syncthreads();
    __syncthreads();
    const int s_first = 2 - FB;
    for (int cb = s_first; cb < Smax; cb += FCH)
        for (int o = 0; o < FCH && cb + o < Smax; o += FB)
#pragma unroll
            for (int k = 0; k < FB; k++) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __syncthreads();
}

// One workgroup = up to two sweeps (halves of P lanes each) over the SAME event, so that one model table in LDS (80 KB) serves
// both: the forward and backward fill of an alignment, or two candidate sequences against one event.  With P = 320 / 384 that is
// 10 / 12 waves = 3 per SIMD: the four SIMDs carry equal instruction streams and one wave's exchange / barrier latency is covered
// by the other two (a lone sweep's 5-6 waves leave one SIMD with two waves and the others waiting for it).
// pairs[2 * blockIdx.x + h] = job * ndir + dir of half h, or -1.  PAIR = false: one sweep per workgroup (P > 384, launches of at
// most PAIR_MIN_SWEEPS sweeps, forward sweeps whose events all differ).  CMP: the compact LDS layout of fill_model_row, lone sweeps
// of up to 256 lanes (two workgroups per CU).
// FASTDIV: tabulated reciprocals (the normal case) or IEEE divisions (some divisor of the AlignData is not a sane number)
template <int MAXT, bool PAIR, bool FASTDIV, bool CMP>
__global__ __launch_bounds__(MAXT) void k_fill(BatchD b, const int* __restrict__ pairs, int ndir, int P, int rcols, int slowwords, int halfbytes) {
    static_assert(!(PAIR && CMP), "the compact layout is for lone sweeps");
    extern __shared__ double2 fill_smem[];
    char* smem = (char*)fill_smem;
    constexpr int MODEL_LDS = CMP ? 64 * NS : FILL_MODEL_BYTES;
    const int hw = PAIR ? (int)threadIdx.x / P : 0;
    const int slot = (int)threadIdx.x - hw * P;
    const int jdA = pairs[2 * blockIdx.x], jdB = PAIR ? pairs[2 * blockIdx.x + 1] : -1;
    const bool okA = jdA >= 0 && !b.jobs[jdA / ndir].out->inert, okB = jdB >= 0 && !b.jobs[jdB / ndir].out->inert;
    if (!okA && !okB) return;
    const int Smax = max(okA ? (int)b.jobs[jdA / ndir].S : 0, okB ? (int)b.jobs[jdB / ndir].S : 0);
    {   // the event's model rows (laid out by the host, 80 bytes apart): one copy for the workgroup
        const double2* src = (const double2*)b.jobs[(okA ? jdA : jdB) / ndir].model8;
        double2* dst = (double2*)smem;
        if (CMP) {
            for (int k = threadIdx.x; k < 4 * NS; k += blockDim.x) {
                const int row = k >> 2, q = k & 3;
                dst[(row << 2) | (q ^ ((row >> 2) & 3))] = src[row * 5 + q];
            }
        } else {
            for (int k = threadIdx.x; k < FILL_MODEL_BYTES / 16; k += blockDim.x) dst[k] = src[k];
        }
    }
    const int jd = hw == 0 ? jdA : jdB;
    if (!(hw == 0 ? okA : okB)) { fill_idle(Smax); return; }
    const JobD& J = b.jobs[jd / ndir];
    char* hsm = smem + MODEL_LDS + hw * halfbytes;
    if (jd % ndir == 0) fill_body<0, FASTDIV, CMP>(b, J, smem, hsm, slot, Smax, rcols, slowwords);
    else fill_body<1, FASTDIV, CMP>(b, J, smem, hsm, slot, Smax, rcols, slowwords);
}

// prefix max over columns + (fwd) the first cell achieving the global max ; grid njobs*ndir, block 64
__global__ __launch_bounds__(64) void k_prefix(BatchD b, int ndir) {
    chain_priority();
    const int jd = blockIdx.x, job = jd / ndir, dir = jd % ndir;
    const JobD& J = b.jobs[job];
    const double* cmax = b.cmax + J.col_off[dir];
    double* pm = b.pm + J.col_off[dir];
    const int lane = threadIdx.x;
    double carry = 0.0;
    if (lane == 0) pm[0] = 0.0;
    for (int c0 = 1; c0 <= J.C; c0 += 64) {
        const int c = c0 + lane;
        double v = c <= J.C ? cmax[c] : 0.0;
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(v, off);
            if (lane >= off && o > v) v = o;
        }
        if (carry > v) v = carry;
        if (c <= J.C) pm[c] = v;
        carry = __shfl(v, 63);
    }
    if (dir == 0 && J.K == 0) {   // (strip sweeps hand in their best cell themselves: k_best)
        const double best = carry;
        int bj = 0x7fffffff;
        if (best > 0.0)
            for (int c = 1 + lane; c <= J.C; c += 64)
                if (cmax[c] == best) { bj = c; break; }
        for (int off = 32; off; off >>= 1) bj = min(bj, __shfl_xor(bj, off));
        int bi = 0x7fffffff;
        if (best > 0.0) {  // first row of that column holding the maximum (row-ascending visit order)
            const int* lb = b.lb + J.lb_off;
            const double2* __restrict__ rec = b.rec + J.mat_off[0];
            int i0, i1;
            band_of(lb, 0, bj, J.C, J.n0, J.W, i0, i1);
            for (int i = i0 + lane; i <= i1; i += 64)
                if (rec[(int64_t)(i + bj) * J.P + slot_of(i, J.P)].x == best) { bi = i; break; }
            for (int off = 32; off; off >>= 1) bi = min(bi, __shfl_xor(bi, off));
        }
        if (lane == 0) {
            JobOut* O = J.out;
            O->best = best;
            if (best > 0.0) { O->bj = bj; O->bi = bi; }
            else { O->bj = 0; O->bi = 0; }
        }
    }
}

// (the backtrace walker is a template shared with ps_sweep.hip: ps_dev.h, bt_walk)
__global__ __launch_bounds__(256) void k_backtrace(BatchD b) {
    const JobD& J = b.jobs[blockIdx.x];
    SkewCodes src;
    src.flg = b.flg + J.mat_off[0]; src.P = J.P; src.sti = 0;
    bt_walk(J, src);
}

// ref_like[i-1] = score of the cell level i was recorded from (cpp/Alignment.cpp:610-618); grid (ceil(maxn/256), njobs)
__global__ __launch_bounds__(256) void k_fill_like(BatchD b) {
    const JobD& J = b.jobs[blockIdx.y];
    if (J.out->inert) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= J.n0) return;
    const long long w = ((const long long*)J.rl)[t];
    if (w == 0) return;   // level not on the path: ref_like stays 0
    const int i = t + 1, j = (int)(w >> 4);   // record = column << 4 | step code << 1 | matrix (bt_walk)
    const double2 v = (b.rec + J.mat_off[0])[(int64_t)(i + j) * J.P + slot_of(i, J.P)];
    J.rl[t] = (w & 1) ? v.y : v.x;
}

// ------------------------------------------------------------------------------------------------
// columnMax(raf, rab) on the filled matrices (cpp/Alignment.h:181-214); cooperative over `nl` lanes
// Only rows where both columns are in band can exceed the two running maxima (stay <= main <= max).
// ------------------------------------------------------------------------------------------------
// maximum over the NL lanes of a lane's group (groups of NL consecutive lanes from lane 0 of the wave on; wl: the lane in its wave).
// NL = 7 (nine groups to a wave, lane 63 idle): three rotations within the group — by 1, 2, 4: every lane then holds the maximum
// of 8 >= NL cyclically consecutive members.
template <int NL>
__device__ __forceinline__ double group_max(double v, int wl) {
    if ((NL & (NL - 1)) == 0) {
        for (int off = 1; off < NL; off <<= 1) v = fmax(v, __shfl_xor(v, off));
    } else {
        const int g0 = (wl / NL) * NL, c = wl - g0;
        for (int off = 1; off < NL; off <<= 1) v = fmax(v, __shfl(v, min(g0 + (c + off) % NL, 63)));
    }
    return v;
}

template <int NL>
__device__ double colmax_pair(const BatchD& b, const JobD& J, int raf, int rab, int lane, int wl) {
    constexpr int nl = NL;
    const int C = J.C, n0 = J.n0;
    if ((unsigned)raf >= (unsigned)(C + 1)) raf = C;
    if ((unsigned)rab >= (unsigned)(C + 1)) rab = C;
    const int* lb = b.lb + J.lb_off;
    int f0 = 0, f1 = n0, b0 = 0, b1 = n0;
    if (raf > 0) band_of(lb, 0, raf, C, n0, J.W, f0, f1);
    if (rab > 0) band_of(lb, 1, rab, C, n0, J.W, b0, b1);
    // jb = n0 - jf + 1 in [b0, b1]  <=>  jf in [n0 + 1 - b1, n0 + 1 - b0]
    const int lo = max(max(1, f0), n0 + 1 - b1), hi = min(min(n0, f1), n0 + 1 - b0);
    double sm = 0.0;
    for (int jf = lo + lane; jf <= hi; jf += nl) {
        const int jb = n0 - jf + 1;
        double2 fv = make_double2(0.0, 0.0), bv = make_double2(0.0, 0.0);
        if (raf > 0) fv = b.rec[rec_index(J, 0, jf, raf, f0)];
        if (rab > 0) bv = b.rec[rec_index(J, 1, jb, rab, b0)];
        sm = fmax(sm, fmax(fv.x + bv.x, fv.y + bv.y));
    }
    sm = group_max<NL>(sm, wl);
    sm = fmax(sm, b.pm[J.col_off[0] + raf]);
    sm = fmax(sm, b.pm[J.col_off[1] + rab]);
    return sm;
}

// old score for each distinct r0 = max(start - 3, 1) ; grid (nr0, njobs), block 64
__global__ __launch_bounds__(64) void k_old(BatchD b, const ScoreArgs* __restrict__ A) {
    chain_priority_wide();
    const ScoreArgs& a = A[blockIdx.z];
    if (a.oldall || (int)blockIdx.x >= a.nr0 || (int)blockIdx.y >= a.njobs) return;
    const JobD& J = b.jobs[a.job0 + blockIdx.y];
    if (J.out->inert) return;
    const int r0 = a.r0[blockIdx.x];
    const double v = colmax_pair<64>(b, J, r0, J.C - r0 + 1, threadIdx.x, threadIdx.x);
    if (threadIdx.x == 0) a.old[(size_t)blockIdx.y * a.nr0 + blockIdx.x] = v;
}

// ------------------------------------------------------------------------------------------------
// columnMax(j, C - j + 1) for EVERY column j at once (the `old` score of all edit positions of a Refine / ScorePoints list).
// Cell (i, j) of the forward matrix lies on anti-diagonal s = i + j; its partner (n0 - i + 1, C - j + 1) of the backward matrix lies
// on anti-diagonal n0 + C + 2 - s, in the mirrored slot order: one pass over the anti-diagonals reads both matrices fully coalesced
// (32 B per cell), where one wave per column (k_old) touches a 128-byte line per 16-byte record.  Sums of two non-negative scores
// are maximised per column through an LDS window and one global atomic per touched column; max is order-independent, so the result
// is the reference's.  grid (ceil(S / OA_SB), njobs), block 256.
// ------------------------------------------------------------------------------------------------
constexpr int OA_SB = 32;      // anti-diagonals per block
constexpr int OA_COLS = 2048;  // column window of a block in LDS
__global__ __launch_bounds__(256) void k_oldall(BatchD b, const ScoreArgs* __restrict__ A) {
    __shared__ unsigned long long s_max[OA_COLS];
    __shared__ int s_jbase, s_ok;
    const ScoreArgs& a = A[blockIdx.z];
    if (!a.oldall || !a.nr0 || (int)blockIdx.y >= a.njobs) return;
    const JobD& J = b.jobs[a.job0 + blockIdx.y];
    if (J.out->inert || J.K) return;
    const int s0 = blockIdx.x * OA_SB;
    const int S = (int)J.S, P = J.P, n0 = J.n0, C = J.C;
    if (s0 >= S) return;
    const int nst = min(OA_SB, S - s0);
    const int* __restrict__ LOf = b.lo + J.lo_off[0];
    const int* __restrict__ HIf = b.hi + J.lo_off[0];
    const int* __restrict__ LOb = b.lo + J.lo_off[1];
    const int* __restrict__ HIb = b.hi + J.lo_off[1];
    if (threadIdx.x == 0) {
        int jmin = 0x7fffffff, jmax = -1;
        for (int k = 0; k < nst; k++) {
            const int lo = LOf[s0 + k];
            if (lo < 0) continue;
            jmax = max(jmax, s0 + k - lo);
            jmin = min(jmin, s0 + k - HIf[s0 + k]);
        }
        s_jbase = jmin;
        s_ok = jmax < 0 ? -1 : (jmax - jmin + 1 <= OA_COLS ? 1 : 0);
    }
    for (int k = threadIdx.x; k < OA_COLS; k += 256) s_max[k] = 0ull;
    __syncthreads();
    if (s_ok < 0) return;
    const bool use_lds = s_ok == 1;
    const int jbase = s_jbase;
    const double2* __restrict__ rf = b.rec + J.mat_off[0];
    const double2* __restrict__ rb = b.rec + J.mat_off[1];
    unsigned long long* gmax = (unsigned long long*)(a.oldall + (size_t)blockIdx.y * a.oldall_pitch);
    const int nP = P >> 6, lane = threadIdx.x & 63;
    int k = 0, c = threadIdx.x >> 6;
    while (c >= nP) { c -= nP; k++; }
    for (; k < nst;) {
        const int s = s0 + k;
        const int slot = c * 64 + lane;
        const int lo = __builtin_amdgcn_readfirstlane(LOf[s]), hi = __builtin_amdgcn_readfirstlane(HIf[s]);
        c += 4;
        while (c >= nP) { c -= nP; k++; }
        if (lo < 0) continue;
        int d = slot - lo % P;
        if (d < 0) d += P;
        const int i = lo + d;
        if (i > hi) continue;
        const int j = s - i;
        const int jb = n0 - i + 1, cb = C - j + 1, sb = jb + cb;      // partner cell in backward coordinates
        if (cb < 1) continue;                                           // (column j = C + 1 does not exist; cb = 0 is the blank column: sums with zero, covered by pm)
        const int lob = LOb[sb];
        if (lob < 0 || jb < lob || jb > HIb[sb]) continue;              // partner outside the backward band
        const double2 fv = rf[(int64_t)s * P + slot];
        const double2 bv = rb[(int64_t)sb * P + jb % P];
        const double v = fmax(fv.x + bv.x, fv.y + bv.y);
        if (v > 0.0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            if (use_lds) atomicMax(&s_max[j - jbase], bits); else atomicMax(&gmax[j], bits);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int q = threadIdx.x; q < OA_COLS; q += 256) {
            const unsigned long long v = s_max[q];
            if (v) atomicMax(&gmax[jbase + q], v);
        }
    }
}

// The same pass over strip matrices (k_sweep2): the forward records of one step t and strip row r are NL consecutive records
// (one per lane = strip; NL = 64 per wavefront of a sweep); their partners in the backward matrix are 64 consecutive records too, in reverse order (row n0 - i + 1 of
// backward column C - j + 1: strip and step move by one per lane in opposite senses).  grid (ceil(maxT / OA_ST), njobs, regions), block 256.
constexpr int OA_ST = 16;      // steps per block
__global__ __launch_bounds__(256) void k_oldall_s(BatchD b, const ScoreArgs* __restrict__ A) {
    __shared__ unsigned long long s_max[OA_COLS];
    __shared__ int s_jbase, s_ok;
    const ScoreArgs& a = A[blockIdx.z];
    if (!a.oldall || !a.nr0 || (int)blockIdx.y >= a.njobs) return;
    const int job = a.job0 + blockIdx.y;
    const JobD& J = b.jobs[job];
    if (J.out->inert || !J.K) return;
    const SweepJob& SF = b.s_sj[2 * job];
    const SweepJob& SB = b.s_sj[2 * job + 1];
    const int K = J.K, T = SF.T, n0 = J.n0, C = J.C;
    const int t0 = blockIdx.x * OA_ST;
    if (t0 >= T) return;
    const int nst = min(OA_ST, T - t0);
    const int* __restrict__ QL = b.s_qlo + SF.q_off;
    const int* __restrict__ QH = b.s_qhi + SF.q_off;
    const int2* __restrict__ bandF = b.s_band + SF.band_off;
    const int2* __restrict__ bandB = b.s_band + SB.band_off;
    if (threadIdx.x == 0) {
        int jmin = 0x7fffffff, jmax = -1;
        for (int k = 0; k < nst; k++) {
            const int ql = QL[t0 + k];
            if (ql < 0) continue;
            jmax = max(jmax, t0 + k - ql);
            jmin = min(jmin, t0 + k - QH[t0 + k]);
        }
        s_jbase = jmin;
        s_ok = jmax < 0 ? -1 : (jmax - jmin + 1 <= OA_COLS ? 1 : 0);
    }
    for (int k = threadIdx.x; k < OA_COLS; k += 256) s_max[k] = 0ull;
    __syncthreads();
    if (s_ok < 0) return;
    const bool use_lds = s_ok == 1;
    const int jbase = s_jbase;
    const double2* __restrict__ rf = b.rec + J.mat_off[0];
    unsigned long long* gmax = (unsigned long long*)(a.oldall + (size_t)blockIdx.y * a.oldall_pitch);
    const int NL = J.NL, NWV = NL >> 6;                                  // lanes of a sweep, 64 per wavefront
    for (int p = threadIdx.x >> 6; p < nst * K * NWV; p += 4) {
        const int t = t0 + p / (K * NWV), r = (p / NWV) % K;
        const int lane = (p % NWV) * 64 + (threadIdx.x & 63);
        const int ql = __builtin_amdgcn_readfirstlane(QL[t]);
        if (ql < 0) continue;
        const int q = ql + ((lane - ql) & (NL - 1));
        const int i = q * K + 1 + r, j = t - q;
        if (j < 1 || j > C || i > n0) continue;
        const int2 bf = bandF[j];
        if (i < bf.x || i > bf.y) continue;
        const int ib = n0 - i + 1, cb = C - j + 1;                      // partner cell in backward coordinates
        const int2 bb = bandB[cb];
        if (ib < bb.x || ib > bb.y) continue;                           // partner outside the backward band
        const double2 fv = rf[((int64_t)t * K + r) * NL + lane];
        const double2 bv = b.rec[rec_index(J, 1, ib, cb, bb.x)];
        const double v = fmax(fv.x + bv.x, fv.y + bv.y);
        if (v > 0.0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            if (use_lds) atomicMax(&s_max[j - jbase], bits); else atomicMax(&gmax[j], bits);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int q = threadIdx.x; q < OA_COLS; q += 256) {
            const unsigned long long v = s_max[q];
            if (v) atomicMax(&gmax[jbase + q], v);
        }
    }
}

// old[r0] = max(0, column sums, forward MaxInfo up to r0, backward MaxInfo up to C - r0 + 1)  (cpp/Alignment.h:181-214); grid (ceil(nr0/256), njobs)
__global__ __launch_bounds__(256) void k_oldfin(BatchD b, const ScoreArgs* __restrict__ A) {
    const ScoreArgs& a = A[blockIdx.z];
    if (!a.oldall || (int)blockIdx.y >= a.njobs) return;
    const JobD& J = b.jobs[a.job0 + blockIdx.y];
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.nr0 || J.out->inert) return;
    const int C = J.C;
    int raf = a.r0[q], rab = C - raf + 1;
    double sm = 0.0;
    if ((unsigned)raf >= (unsigned)(C + 1)) raf = C;           // columnMax's own clamping (cpp/Alignment.h:186-189)
    else sm = (a.oldall + (size_t)blockIdx.y * a.oldall_pitch)[raf];
    if ((unsigned)rab >= (unsigned)(C + 1)) rab = C;
    sm = fmax(sm, b.pm[J.col_off[0] + raf]);
    sm = fmax(sm, b.pm[J.col_off[1] + rab]);
    a.old[(size_t)blockIdx.y * a.nr0 + q] = sm;
}

// ------------------------------------------------------------------------------------------------
// scoreMutation (cpp/Alignment.cpp:447-512): G lanes per (event, edit) item, lane = new column,
// rows stream through the group systolically (lane c works on row  base + t - c  at step t).
// ------------------------------------------------------------------------------------------------
// (at most 96 VGPRs — five waves per SIMD: with 100 a workgroup's wave did not fit beside the two 208-register waves a lone k_fill
//  sweep keeps on SIMD 0 and 1, so k_score ran only on CUs without a fill: 1.07 ms per launch in the bench against 0.25 ms alone)
template <int G, bool FD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_score(BatchD b, const ScoreArgs* __restrict__ A) {
    constexpr int IPW = 64 / G;   // (G = 7: nine items to a wave, lane 63 idle)
    constexpr int CLS = G == 7 ? 4 : G == 8 ? 0 : (G == 16 ? 1 : (G == 32 ? 2 : 3));
    const ScoreArgs& a = A[blockIdx.z];
    const int nitems = a.cls_count[CLS];
    if ((int)blockIdx.y >= a.njobs || (int)blockIdx.x * 4 * IPW >= nitems) return;
    const int* __restrict__ items = a.cls_items[CLS];
    __shared__ double s_carry[(G == 64) ? 4 * 1024 : 1];   // last column of a 64-column chunk, per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, c = lane % G;
    const int job = blockIdx.y;   // within the AlignData's own jobs, which start at a.job0 in the batch
    const JobD& J = b.jobs[a.job0 + job];
    const int it = (blockIdx.x * 4 + wave) * IPW + g;
    const bool have = it < nitems && g < IPW;
    const int m = have ? items[it] : 0;
    const int n0 = J.n0, C = J.C, WS = a.ws;
    const bool live = have && !J.out->inert && !a.m_skip[m];
    if (__ballot(live) == 0ull) {
        if (have && c == 0) a.delta[(size_t)job * a.nitems_per_job + m] = 0.0;
        return;
    }
    const int start = a.m_start[m], mlen = a.m_mlen[m], Cm = a.m_cm[m], ncol = live ? a.m_ncol[m] : 0;
    const int sidx = max(start - 4, 0);
    const int tcol = min(start + mlen + 1, sidx + ncol);   // target column (== sidx: the spliced copy itself)
    const int trel = tcol - sidx - 1;                      // 0-based new-column index of the target (-1: copy)
    int backind = Cm - tcol + 1;
    if ((unsigned)backind >= (unsigned)(C + 1)) backind = C;
    const int* __restrict__ lbf = b.lb + J.lb_off;    // tables the fills were made with
    const int* __restrict__ lbn = b.lb + J.lbn_off;   // after the backtrace: centres of the new columns
    // (the level records through a global-address-space pointer held in scalar registers: read through the job table the pointer is
    //  generic, and a FLAT load with 64-bit address arithmetic per row step costs nine vector instructions where this costs two)
    const PS_GLOBAL v4d* __restrict__ levf = (const PS_GLOBAL v4d*)uni_ptr(J.lev[0]);
    const double lsk = J.lsk, lst = J.lst, lex = J.lex, lin = J.lin;

    // band of the back column the target is combined with
    int bb0 = 0, bb1 = n0;
    if (backind > 0) band_of(lbf, 1, backind, C, n0, J.W, bb0, bb1);

    double fmaxv = live ? b.pm[J.col_off[0] + sidx] : 0.0;   // running MaxInfo score, carried from the spliced column
    double tm = 0.0;                                          // max over rows of fwd + back on the target column
    // previous column of the first chunk: the original forward column sidx (blank column 0 if sidx == 0)
    int pc0 = 0, pc1 = n0;
    if (live && sidx > 0) band_of(lbf, 0, sidx, C, n0, J.W, pc0, pc1);
    double* carry = s_carry + wave * 1024;

    const int nchunk = (ncol + G - 1) / G;
    int nchunk_w = nchunk;
    for (int off = 1; off < 64; off <<= 1) nchunk_w = max(nchunk_w, __shfl_xor(nchunk_w, off));
    for (int ch = 0; ch < nchunk_w; ch++) {
        const int cc = ch * G + c;                 // new-column index of this lane
        const bool mine = live && cc < ncol;
        const int jc = sidx + 1 + cc;              // column number in the edited sequence
        int i0 = 1, i1 = 0, state = -1;
        if (mine) {
            const int v = lbn[jc];
            const int ce = clampi(v < 0 ? 1 : v, 1, n0);
            i0 = max(1, ce - WS); i1 = min(n0, ce + WS);
            state = a.m_states[(size_t)m * a.ncolmax + cc];
        }
        // the column's model row as k_fill uses it: {mean, 1/stdv, stdv, log stdv, sd mean, 1/sd mean, lambda, log lambda}
        double mr[8] = {0, 1, 1, 0, 1, 1, 0, 0};
        if (state >= 0) {
            const double2* row = (const double2*)((const char*)J.model8 + (size_t)state * MODEL_ROW_BYTES);
            const double2 q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
            mr[0] = q0.x; mr[1] = q0.y; mr[2] = q1.x; mr[3] = q1.y; mr[4] = q2.x; mr[5] = q2.y; mr[6] = q3.x; mr[7] = q3.y;
        }
        // band of the column to the left
        int p0 = __shfl_up(i0, 1), p1 = __shfl_up(i1, 1);
        if (c == 0) { p0 = pc0; p1 = pc1; }
        const int base = __shfl(i0, g * G);                       // first row of the chunk's first column
        int lastcol = min(ncol - ch * G, G) - 1;                   // lanes 0..lastcol are in use
        int span = mine ? (i1 - base) + c : -1;
        for (int off = 1; off < 64; off <<= 1) span = max(span, __shfl_xor(span, off));
        double cm = 0.0, cs = 0.0, colmax = 0.0, lprev = 0.0;
        const bool is_t = mine && cc == trel;
        // The common case — skewed matrices, the item's columns in one chunk, no invalid 5-mer among them — runs a branch-free step:
        // every lane computes a cell on every step (on a clamped row outside its band) and what must not count is masked; the
        // matrix reads walk the skewed storage with cursors instead of index arithmetic per step.  Values are those of the general
        // loop below: candidates that are switched off there (no left / diagonal neighbour, top row) are -infinity, `lik_insert`
        // or `0 + x` here, which lose to the floors exactly where the reference's never-assigned candidates do.
        const bool use_f = mine && c == 0 && sidx > 0;          // lane 0 reads the spliced forward column
        const bool use_b = is_t && backind > 0;                  // the target lane reads the backward column
        // (a lane that reads both — an edit so close to the end of the sequence that its only new column is the target — takes the general loop)
        if (G < 64 && J.K <= 0 && __ballot((mine && state < 0) || (use_f && use_b)) == 0ull) {
            // (column-sparse records, J.K < 0: a column is one contiguous run of records, the cursors move by one; the wrap tests
            //  of the skewed storage never fire)
            const bool sp = J.K < 0;
            const int P = sp ? 0x40000000 : J.P;
            const unsigned stride = sp ? 1u : (unsigned)P + 1u;
            const double2* __restrict__ rf = b.rec + J.mat_off[0];
            const double2* __restrict__ rb = b.rec + J.mat_off[1];
            const double NINF = -__builtin_inf();
            int i = base - 1 - c;                                    // row of this lane on step t = -1
            int fslot = i >= 0 ? i % P : 0, bslot = (n0 - i + 1) % P;
            unsigned fidx = (unsigned)(max(i, 0) + sidx) * (unsigned)P + (unsigned)fslot;            // record of (i, sidx), forward matrix
            unsigned bidx = (unsigned)(n0 - i + 1 + backind) * (unsigned)P + (unsigned)bslot;        // record of (n0 - i + 1, backind), backward matrix
            if (sp) {
                fslot = 0; bslot = 0x7fffffff;
                fidx = use_f ? (unsigned)J.keep[0][sidx] * (unsigned)J.pitch + (unsigned)(i - pc0) : 0u;
                bidx = use_b ? (unsigned)J.keep[1][backind] * (unsigned)J.pitch + (unsigned)(n0 - i + 1 - bb0) : 0u;
            }
            // one cursor per lane: a lane reads the forward column (lane 0 of an item, `use_f`) or the backward one (the target lane,
            // `use_b`: new column len(mut) + 4 of the edit unless the sequence ends before) — not both (checked above)
            const bool bwd = use_b;
            unsigned idx = bwd ? bidx : fidx;
            int slot = bwd ? bslot : fslot;
            const int sgn = bwd ? -1 : 1, wrap_at = bwd ? 0 : P - 1, wrap_to = bwd ? P - 1 : 0;
            const unsigned dstride = bwd ? 0u - stride : stride;
            const double log2pi = b.log2pi, off = J.lik_offset;
            // forward level record of row i: {mean, stdv, 3 log stdv, 1 / stdv}; scalar base + unsigned 32-bit byte offset: one vector
            // instruction of address arithmetic.  (Fetched a step ahead it was 4 % slower, and the band masks as one high-word select
            // in front of a bare v_max_f64 9 % slower, than this loop: measured, `tools/gpu_scorebench.py`.)
            auto level_of = [&](int row) { return *(const PS_GLOBAL v4d*)((const PS_GLOBAL char*)levf + (unsigned)(32 * (clampi(row, 1, n0) - 1))); };
            for (int t = -1; t <= span; t++) {
                const v4d lv4 = level_of(i);
                double L = wave_shr1(cm);
                const bool vl = i >= p0 && i <= p1;
                if (c == 0) { L = 0.0; if (use_f && vl && i >= 1) L = rf[idx].x; }
                const double D = lprev;
                lprev = L;
                const bool inb = mine && i >= i0 && i <= i1;
                const double lev[4] = {lv4.x, lv4.y, lv4.z, lv4.w};
                const double o = emission8<FD>(mr, lev, log2pi, off);
                const bool vd = vl && i != p0, top = i == i0;
                const double Lz = vl ? L : 0.0, Dz = vd ? D : 0.0;
                const double um = top ? NINF : cm, us = top ? NINF : cs;
                const double cSKIP = Lz + lsk, cMATCH = Dz + o, cIGN = Dz + lin;
                const double cSTAY = um + o + lst, cEXT = us + o + lex, cINS = um + lin;
                const double ns = fmax(fmax(top ? -BIG : 0.0, cSTAY), cEXT);
                double nm = fmax(0.0, cSKIP);
                nm = fmax(nm, cMATCH);
                nm = fmax(nm, cINS);
                nm = fmax(nm, cIGN);
                nm = fmax(nm, ns);
                cm = nm; cs = ns;                                    // (rows outside the band leave garbage that nothing reads: the next in-band row is a top row)
                colmax = fmax(colmax, inb ? nm : 0.0);
                const int jb = n0 - i + 1;
                const bool hit = inb && is_t && jb >= bb0 && jb <= bb1;
                double2 bv = make_double2(0.0, 0.0);
                if (hit && use_b) bv = rb[idx];
                const double cand = fmax(nm + bv.x, ns + bv.y);
                tm = hit ? fmax(tm, cand) : tm;
                // next row: one anti-diagonal on, one slot on (forward); one back each (backward)
                i++;
                const bool wr = slot == wrap_at;
                idx += wr ? (unsigned)sgn : dstride; slot = wr ? wrap_to : slot + sgn;
            }
        } else {
            for (int t = -1; t <= span; t++) {
                const int i = base + t - c;
                // value of (i, jc-1): matrix / carried chunk column for lane 0, left neighbour otherwise
                double L = wave_shr1(cm);
                if (c == 0) {
                    L = 0.0;
                    if (mine && i >= p0 && i <= p1 && i >= 1) {
                        if (ch > 0) L = carry[i - p0];
                        else if (sidx > 0) L = b.rec[rec_index(J, 0, i, sidx, pc0)].x;
                    }
                }
                const double D = lprev;
                lprev = L;
                if (mine && i >= i0 && i <= i1) {
                    double nm = 0.0, ns = 0.0;
                    if (state >= 0) {
                        const v4d lv4 = levf[i - 1];       // forward level record of row i: {mean, stdv, 3 log stdv, 1 / stdv}
                        const double lev[4] = {lv4.x, lv4.y, lv4.z, lv4.w};
                        const double o = emission8<FD>(mr, lev, b.log2pi, J.lik_offset);
                        const bool vl = i >= p0 && i <= p1, vd = i > p0 && i <= p1;
                        const double cSKIP = vl ? L + lsk : lsk;
                        const double cMATCH = vd ? D + o : o;
                        const double cIGN = vd ? D + lin : 0.0;
                        double cSTAY = -BIG, cEXT = -BIG, cINS = 0.0;
                        if (i == i0) ns = -BIG;
                        else { cSTAY = cm + o + lst; cINS = cm + lin; cEXT = cs + o + lex; }
                        if (cSTAY > ns) ns = cSTAY;
                        if (cEXT > ns) ns = cEXT;
                        if (cSKIP > nm) nm = cSKIP;
                        if (cMATCH > nm) nm = cMATCH;
                        if (cINS > nm) nm = cINS;
                        if (cIGN > nm) nm = cIGN;
                        if (ns > nm) nm = ns;
                    }
                    cm = nm; cs = ns;
                    if (nm > colmax) colmax = nm;
                    if (is_t) {
                        const int jb = n0 - i + 1;
                        if (jb >= bb0 && jb <= bb1) {
                            double2 bv = make_double2(0.0, 0.0);
                            if (backind > 0) bv = b.rec[rec_index(J, 1, jb, backind, bb0)];
                            tm = fmax(tm, fmax(nm + bv.x, ns + bv.y));
                        }
                    }
                    if (G == 64 && c == lastcol && ch + 1 < nchunk) carry[i - i0] = nm;
                }
            }
        }
        // MaxInfo: max over the new columns up to and including the target
        double cmx = (mine && cc <= trel) ? colmax : 0.0;
        cmx = group_max<G>(cmx, lane);
        fmaxv = fmax(fmaxv, cmx);
        // next chunk continues from this chunk's last column
        pc0 = __shfl(i0, g * G + max(lastcol, 0)); pc1 = __shfl(i1, g * G + max(lastcol, 0));
        __builtin_amdgcn_wave_barrier();
    }
    // gather the target lane's row maximum
    const double tmx = group_max<G>(tm, lane);
    double now;
    if (trel < 0) {
        now = !live ? 0.0 : colmax_pair<G>(b, J, sidx, backind, c, lane);   // no new column: the spliced copy is the target
    } else {
        now = fmax(0.0, tmx);
        now = fmax(now, fmaxv);
        now = fmax(now, b.pm[J.col_off[1] + backind]);
    }
    if (have && c == 0) {
        double d = 0.0;
        if (live) {
            const double old = a.old[(size_t)job * a.nr0 + a.m_oldidx[m]];
            d = now - old;
        }
        a.delta[(size_t)job * a.nitems_per_job + m] = d;
    }
}

// score[m] = -1e-6 + sum over events in order (cpp/AlignUtil.h:86, cpp/MakeMutations.cpp:51)
__global__ void k_reduce(const ScoreArgs* __restrict__ A) {
    chain_priority_wide();
    const ScoreArgs& a = A[blockIdx.y];
    const int njobs = a.njobs;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.nitems_per_job) return;
    double s = -1e-6;
    for (int e = 0; e < njobs; e++) s += a.delta[(size_t)e * a.nitems_per_job + m];
    a.score[m] = s;
}

// latch the reference's "stripe_width == 0" decision (cpp/Alignment.cpp:51-59) for this API call
__global__ void k_begin(BatchD b) {
    chain_priority_wide();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) *b.maxw = 0;
    if (j >= b.njobs) return;
    JobOut* O = b.jobs[j].out;
    O->inert = (b.jobs[j].force_inert || !O->has_index) ? 1 : 0;
    O->best = 0.0; O->bi = 0; O->bj = 0;
}

// out[job] = JobOut.best of every job of the batch (the results live in their AlignData's own slabs: gathered here so that one copy
// returns them)
__global__ void k_gather_best(BatchD b, double* __restrict__ out) {
    chain_priority_wide();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < b.njobs) out[j] = b.jobs[j].out->best;
}

// =================================================================================================
// launchers
// =================================================================================================
#define PS_LAUNCH_CHECK() PS_HIP(hipGetLastError())

int launch_updaterefs(Runtime* rt, const BatchD& b) {
    if (!b.njobs) return PS_OK;
    hipLaunchKernelGGL(k_updaterefs, dim3(b.njobs), dim3(256), 0, rt->stream, b);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int launch_lb(Runtime* rt, const BatchD& b, int which, int maxlbn) {
    if (!b.njobs) return PS_OK;
    if (maxlbn <= 0) return fail(PS_ERR_BAD_ARG, "launch_lb: size");
    hipLaunchKernelGGL(k_lb, dim3((maxlbn + 255) / 256, b.njobs), dim3(256), 0, rt->stream, b, which);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int launch_lo(Runtime* rt, const BatchD& b, int ndir, int64_t maxS) {
    if (!b.njobs) return PS_OK;
    hipLaunchKernelGGL(k_lo, dim3((unsigned)((maxS + LO_PAD + 255) / 256), b.njobs * ndir), dim3(256), 0, rt->stream, b, ndir);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

constexpr int PAIR_MIN_SWEEPS = 160;
// LDS of one k_fill workgroup: the event's model rows + per half two exchange buffers, the column-maxima ring, the slow-body bitmap
static int fill_ring_cols(int P) { return P + 96 <= 512 ? 512 : (P + 96 <= 1024 ? 1024 : 2048); }
static int fill_slow_words(int64_t maxS) { return (int)((maxS + 2 * FB) / FB / 32 + 2); }
static int fill_half_bytes(int P, int64_t maxS) {
    return (int)((((size_t)2 * P * 24 + (size_t)fill_ring_cols(P) * 8 + (size_t)fill_slow_words(maxS) * 4) + 63) / 64 * 64);
}
// compact layout (lone sweeps): 64 KB of model rows, exchange records sized for the widest direction present
static int fill_compact_bytes(int P, int ndir) { return 64 * NS + 2 * P * (ndir == 2 ? 24 : 16) + fill_ring_cols(P) * 8; }
constexpr int LDS_HALF_CU = 80 * 1024;

template <int MAXT, bool PAIR, bool CMP>
static void fill_launch(Runtime* rt, const BatchD& b, const int* d_pairs, int nwg, int ndir, int P, int64_t maxS, size_t lds) {
    const dim3 grid(nwg), block(PAIR ? 2 * P : P);
    const int rc = fill_ring_cols(P), sw = fill_slow_words(maxS), hb = fill_half_bytes(P, maxS);
    if (b.fastdiv) hipLaunchKernelGGL((k_fill<MAXT, PAIR, true, CMP>), grid, block, lds, rt->stream, b, d_pairs, ndir, P, rc, sw, hb);
    else hipLaunchKernelGGL((k_fill<MAXT, PAIR, false, CMP>), grid, block, lds, rt->stream, b, d_pairs, ndir, P, rc, sw, hb);
}

int launch_fill(Runtime* rt, const BatchD& b, const std::vector<JobD>& jobs, int ndir, int64_t maxS, int P, int64_t ncols) {
    if (!b.njobs) return PS_OK;
    static std::atomic<bool> attr_set(false);   // more than the default 64 KB of dynamic LDS needs the attribute (idempotent: racing threads set the same value)
    if (!attr_set.load(std::memory_order_acquire)) {
#define PS_FILL_ATTR(...) PS_HIP(hipFuncSetAttribute((const void*)k_fill<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        PS_FILL_ATTR(768, true, true, false); PS_FILL_ATTR(768, true, false, false);
        PS_FILL_ATTR(512, false, true, false); PS_FILL_ATTR(512, false, false, false);
        PS_FILL_ATTR(512, false, true, true); PS_FILL_ATTR(512, false, false, true);
        PS_FILL_ATTR(1024, false, true, false); PS_FILL_ATTR(1024, false, false, false);
#undef PS_FILL_ATTR
        attr_set.store(true, std::memory_order_release);
    }
    PS_HIP(hipMemsetAsync(b.cmax, 0, ncols * sizeof(double), rt->stream));
    if (P > 1024) {   // footprint wider than one lane per row: the two-slots-per-thread sweep
        static std::atomic<bool> wide_attr(false);
        if (!wide_attr.load(std::memory_order_acquire)) {
            PS_HIP(hipFuncSetAttribute((const void*)k_fill_wide<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            PS_HIP(hipFuncSetAttribute((const void*)k_fill_wide<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            wide_attr.store(true, std::memory_order_release);
        }
        const int rc = 4096;
        const size_t lds = (size_t)2 * P * 24 + (size_t)rc * 8;
        prof_begin(rt);
        if (b.fastdiv) hipLaunchKernelGGL(k_fill_wide<true>, dim3(b.njobs * ndir), dim3(P / 2), lds, rt->stream, b, ndir, rc);
        else hipLaunchKernelGGL(k_fill_wide<false>, dim3(b.njobs * ndir), dim3(P / 2), lds, rt->stream, b, ndir, rc);
        PS_LAUNCH_CHECK();
        prof_end(rt, "fill", 0.0);
        hipLaunchKernelGGL(k_prefix, dim3(b.njobs * ndir), dim3(64), 0, rt->stream, b, ndir);
        PS_LAUNCH_CHECK();
        return PS_OK;
    }
    // workgroups: two sweeps over the same event share one (model table in LDS): forward + backward of a job, or two
    // forward-only jobs of one event (candidate sequences of FindMutations), longest with longest
    // (a launch that fits the chip with one sweep per workgroup keeps them apart: a lone sweep finishes ~20 % sooner than a pair,
    //  and a launch this small is on some region's critical path)
    // (a test hook that tests/test_hip_parity.py changes between calls of one process: read per launch; nothing in the library or
    //  its drivers calls setenv, so the read does not race)
    const char* pm_env = getenv("PORESEQ_DEBUG_PAIR_MIN");
    const int pair_min = pm_env ? atoi(pm_env) : PAIR_MIN_SWEEPS;
    bool pair = 2 * P <= 768 && b.njobs * ndir > pair_min;
    std::vector<int> pr;
    if (pair && ndir == 2) {
        for (int j = 0; j < b.njobs; j++) { pr.push_back(2 * j); pr.push_back(2 * j + 1); }
    } else if (pair) {
        std::map<const double*, std::vector<int>> by_event;
        for (int j = 0; j < b.njobs; j++) by_event[jobs[j].model8].push_back(j);
        bool any = false;
        for (auto& kv : by_event) any |= kv.second.size() > 1;
        if (!any) pair = false;   // every event appears once (forward sweeps of many regions): nothing to share a table with
        for (auto& kv : by_event) {
            if (!pair) break;
            std::vector<int>& v = kv.second;
            std::sort(v.begin(), v.end(), [&](int x, int y) { return jobs[x].S != jobs[y].S ? jobs[x].S > jobs[y].S : x < y; });
            for (size_t k = 0; k < v.size(); k += 2) { pr.push_back(v[k]); pr.push_back(k + 1 < v.size() ? v[k + 1] : -1); }
        }
    }
    if (!pair)
        for (int jd = 0; jd < b.njobs * ndir; jd++) { pr.push_back(jd); pr.push_back(-1); }
    const int nwg = (int)pr.size() / 2;
    PS_TRY(rt->buf("fill_pairs").ensure(pr.size() * sizeof(int)));
    int* d_pairs = rt->buf("fill_pairs").as<int>();
    PS_TRY(rt->up(d_pairs, pr.data(), pr.size() * sizeof(int)));
    // a lone sweep of up to four waves takes the compact layout: two workgroups per CU (one wave of each per SIMD; measured
    // 1.5x the sweeps per second of one workgroup per CU at P = 192).  Wider ones stay alone on their CU whatever the LDS says:
    // five or six waves put two on SIMD 0, and a second workgroup's two more do not fit there beside them at 200 VGPRs
    // (tried at 168 VGPRs and exactly 80 KB for P = 384: no co-residency, and 10 % slower for the spills).
    const bool compact = !pair && P <= 256 && fill_compact_bytes(P, ndir) <= LDS_HALF_CU;
    const size_t lds = compact ? fill_compact_bytes(P, ndir) : FILL_MODEL_BYTES + (size_t)(pair ? 2 : 1) * fill_half_bytes(P, maxS);
    if (lds > 160 * 1024) return fail(PS_ERR_UNSUPPORTED, "alignment too long for the fill kernel's LDS bitmap");
    prof_begin(rt);
    if (pair) fill_launch<768, true, false>(rt, b, d_pairs, nwg, ndir, P, maxS, lds);
    else if (compact) fill_launch<512, false, true>(rt, b, d_pairs, nwg, ndir, P, maxS, lds);
    else if (P <= 512) fill_launch<512, false, false>(rt, b, d_pairs, nwg, ndir, P, maxS, lds);
    else fill_launch<1024, false, false>(rt, b, d_pairs, nwg, ndir, P, maxS, lds);
    PS_LAUNCH_CHECK();
    prof_end(rt, "fill", 0.0);
    hipLaunchKernelGGL(k_prefix, dim3(b.njobs * ndir), dim3(64), 0, rt->stream, b, ndir);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int launch_prefix(Runtime* rt, const BatchD& b, int ndir) {
    if (!b.njobs) return PS_OK;
    hipLaunchKernelGGL(k_prefix, dim3(b.njobs * ndir), dim3(64), 0, rt->stream, b, ndir);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int launch_backtrace(Runtime* rt, const BatchD& b, int maxn) {
    if (!b.njobs) return PS_OK;
    hipLaunchKernelGGL(k_backtrace, dim3(b.njobs), dim3(256), 0, rt->stream, b);
    PS_LAUNCH_CHECK();
    if (maxn > 0) {
        hipLaunchKernelGGL(k_fill_like, dim3((maxn + 255) / 256, b.njobs), dim3(256), 0, rt->stream, b);
        PS_LAUNCH_CHECK();
    }
    return PS_OK;
}

int launch_gather_best(Runtime* rt, const BatchD& b, double* out) {
    if (!b.njobs) return PS_OK;
    hipLaunchKernelGGL(k_gather_best, dim3((b.njobs + 255) / 256), dim3(256), 0, rt->stream, b, out);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int launch_begin(Runtime* rt, const BatchD& b) {
    if (!b.njobs) return PS_OK;
    hipLaunchKernelGGL(k_begin, dim3((b.njobs + 63) / 64), dim3(64), 0, rt->stream, b);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

// edit scoring of several AlignData of one batch (d_sas: their ScoreArgs on the device, h_sas the same on the host): one launch
// per kernel over all of them — grid.z (k_reduce: grid.y) is the AlignData, blocks past an AlignData's own sizes leave at once
int launch_score(Runtime* rt, const BatchD& b, const ScoreArgs* d_sas, const std::vector<ScoreArgs>& h_sas) {
    const int R = (int)h_sas.size();
    int maxE = 0, maxnr0 = 0, maxnr0_all = 0, maxM = 0, cls_max[SCORE_CLASSES] = {0, 0, 0, 0, 0};
    int64_t maxS = 0;
    bool any_all = false, any_old = false;
    for (const ScoreArgs& a : h_sas) {
        if (!a.njobs || !a.nitems_per_job) continue;
        maxE = std::max(maxE, a.njobs); maxM = std::max(maxM, a.nitems_per_job);
        if (a.nr0 > 0 && a.oldall) { any_all = true; maxS = std::max(maxS, a.maxS); maxnr0_all = std::max(maxnr0_all, a.nr0); }
        else if (a.nr0 > 0) { any_old = true; maxnr0 = std::max(maxnr0, a.nr0); }
        for (int k = 0; k < SCORE_CLASSES; k++) cls_max[k] = std::max(cls_max[k], a.cls_count[k]);
    }
    if (!maxE || !maxM) return PS_OK;
    if (any_all) {
        // a list that touches most columns (Refine, ScorePoints): one coalesced pass over both matrices serves every position
        // (the caller has zeroed the oldall arrays)
        if (b.s_sj) hipLaunchKernelGGL(k_oldall_s, dim3((unsigned)((maxS + OA_ST - 1) / OA_ST), maxE, R), dim3(256), 0, rt->stream, b, d_sas);   // (maxS >= every T)
        else hipLaunchKernelGGL(k_oldall, dim3((unsigned)((maxS + OA_SB - 1) / OA_SB), maxE, R), dim3(256), 0, rt->stream, b, d_sas);
        hipLaunchKernelGGL(k_oldfin, dim3((maxnr0_all + 255) / 256, maxE, R), dim3(256), 0, rt->stream, b, d_sas);
        PS_LAUNCH_CHECK();
    }
    if (any_old) {
        hipLaunchKernelGGL(k_old, dim3(maxnr0, maxE, R), dim3(64), 0, rt->stream, b, d_sas);
        PS_LAUNCH_CHECK();
    }
    prof_begin(rt);
    for (int k = 0; k < SCORE_CLASSES; k++) {
        const int n = cls_max[k];
        if (!n) continue;
        const int G = k == 4 ? 7 : 8 << k, ipb = 4 * (64 / G);
        dim3 grid((n + ipb - 1) / ipb, maxE, R), block(256);
#define PS_SCORE(GG) do { if (b.fastdiv) hipLaunchKernelGGL((k_score<GG, true>), grid, block, 0, rt->stream, b, d_sas); \
                          else hipLaunchKernelGGL((k_score<GG, false>), grid, block, 0, rt->stream, b, d_sas); } while (0)
        if (k == 4) PS_SCORE(7); else if (k == 0) PS_SCORE(8); else if (k == 1) PS_SCORE(16); else if (k == 2) PS_SCORE(32); else PS_SCORE(64);
#undef PS_SCORE
        PS_LAUNCH_CHECK();
    }
    prof_end(rt, "score", 0.0);
    hipLaunchKernelGGL(k_reduce, dim3((maxM + 255) / 256, R), dim3(256), 0, rt->stream, d_sas);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // namespace ps
