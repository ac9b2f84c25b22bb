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
// measured footprint, typically ~half the band).  Cell (i, j) needs (i, j-1) = the lane's own
// previous value, (i-1, j) = the neighbour lane's previous value and (i-1, j-1) = what the neighbour
// published one step earlier; neighbour values travel through three rotating LDS buffers, one
// LDS-only s_barrier per anti-diagonal.  Pipeline per batch of alignments:
//   k_lb / k_lo   band centres per column, lowest in-band row + footprint per anti-diagonal
//   k_emis        chip-wide: emission log-densities (three FP64 divisions each) + band flags
//   k_recur       the serial part: values only (fmax); lanes without a cell hold -infinity, which doubles as the
//                 reference's implicit zeros / top-row rule, so only two band flags are consulted per cell
//   k_invfix      (only when a sequence has an invalid 5-mer) zero records of those columns
//   k_steps       chip-wide: back-pointer codes re-derived with the reference's ordered selection,
//                 per-column maxima (LDS-aggregated atomics)
//   k_prefix      running MaxInfo per column, first cell of the global maximum
//   k_backtrace   LDS-tile walker with wave-wide look-ahead and tile prefetch, then k_fill_like, k_updaterefs /
//                 k_lb for the new band centres
//   k_old / k_score / k_reduce   edit scoring (scoreMutation + columnMax) and the per-edit sums
#include "ps_internal.h"

namespace ps {

enum : unsigned { F_ACT = 1, F_VL = 2, F_VD = 4, F_TOP = 8, F_BLANK = 32,
                  F_RD = 128 /* the diagonal neighbour's value counts (see recur_step) */ };
constexpr unsigned FLG_DEAD = 0xC000u;   // final step word of a cell in an invalid-5-mer column: both scores <= 0, no move
enum : unsigned { M_SKIP = 0, M_MATCH = 1, M_INSERT = 2, M_IGNORE = 3, M_STAY = 4, M_EXTEND = 5, M_IMPL = 255 };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// std::lower_bound(double*, int) exactly as libstdc++ walks it (cpp/EventData.h:178)
__device__ int lower_bound_d(const double* __restrict__ a, int n, int v) {
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

__device__ __forceinline__ int slot_of(int i, int P) { return i % P; }

// emission log-density; cpp/AlignUtil.h:34-38,48-53 + cpp/Alignment.cpp:169-173
struct ModelRow { double mu, sg, lsg, sm, lam, llam; };
__device__ __forceinline__ double emission(const ModelRow& m, double x, double sd, double lsd, double log2pi, double off) {
    double d = (x - m.mu) / m.sg;
    double l = -0.5 * (d * d + log2pi) - m.lsg;
    double e = (sd - m.sm) / m.sm;
    double g = 0.5 * (m.llam - 3 * lsd - log2pi - e * e * m.lam / sd);
    l += g;
    l += off;
    return l;
}

// ------------------------------------------------------------------------------------------------
// updaterefs: ref_align -> ref_index, refstart, refend   (cpp/EventData.h:110-169)
// one 256-thread block per job; each thread owns a contiguous chunk of levels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_updaterefs(BatchD b) {
    const JobD& J = b.jobs[blockIdx.x];
    JobOut* O = b.out + blockIdx.x;
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
    const JobD& J = b.jobs[blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J.lbn) return;
    int* lb = b.lb + (which ? J.lbn_off : J.lb_off);
    lb[j] = b.out[blockIdx.y].has_index ? lower_bound_d(J.ri, J.n0, j) : -1;
}

// ------------------------------------------------------------------------------------------------
// lo[s]: lowest in-band row on anti-diagonal s (-1: none) and the footprint width of the band on s.
// The columns present on s are those with i0(j) + j <= s <= i1(j) + j — a contiguous range
// [jlo, jhi] because both bounds are strictly increasing in j; rows are s - j, so lo = s - jhi and
// width = jhi - jlo + 1 <= 2W + 1.  The per-job maximum width sizes P (slots per anti-diagonal).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lo(BatchD b, int ndir) {
    const int jd = blockIdx.y, job = jd / ndir, dir = jd % ndir;
    const JobD& J = b.jobs[job];
    if (b.out[job].inert) return;
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int width = 0;
    if (s < J.S) {
        const int* lb = b.lb + J.lb_off;
        int res = -1;
        if (J.C >= 1) {
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
                    if (lo <= jhi) { res = (int)(s - jhi); width = jhi - lo + 1; }
                }
            }
        }
        (b.lo + J.lo_off[dir])[s] = res;
    }
    for (int off = 32; off; off >>= 1) width = max(width, __shfl_xor(width, off));
    if ((threadIdx.x & 63) == 0 && width > 0) atomicMax(&b.out[job].maxw, width);
}

// ------------------------------------------------------------------------------------------------
// emission pass: for every (s, slot) write EM = emission (or 0) and FLG = band flags
// grid (nblk, njobs*ndir), block 1024 (16 waves share the event's 48 KB model in LDS: two blocks per CU give full
// occupancy).  The work unit is one wave-wide run of 64 slots of one anti-diagonal; every wave walks a contiguous
// range of units, so the anti-diagonal index and everything derived from it (LO[s], its residue mod P) is
// wave-uniform and no per-cell division is needed.
// ------------------------------------------------------------------------------------------------
constexpr int PF = 3;            // anti-diagonals per prefetch group of k_recur (two groups ping-pong in registers)
constexpr int REC_PAD = 2 * PF;  // spare anti-diagonals behind every matrix: the padded last groups of k_recur touch them
constexpr int EMIS_T = 1024;
__global__ __launch_bounds__(EMIS_T) void k_emis(BatchD b, int ndir) {
    __shared__ double s_model[6 * NS];
    const int jd = blockIdx.y, job = jd / ndir, dir = jd % ndir;
    const JobD& J = b.jobs[job];
    if (b.out[job].inert) return;
    const double* gm = b.model + (size_t)J.ev * 6 * NS;
    for (int k = threadIdx.x; k < 6 * NS; k += EMIS_T) s_model[k] = gm[k];
    __syncthreads();
    const int P = J.P, n0 = J.n0, C = J.C, W = J.W;
    const int* __restrict__ lb = b.lb + J.lb_off;
    const int* __restrict__ LO = b.lo + J.lo_off[dir];
    const int* __restrict__ st = b.states + J.st_off;
    const double* __restrict__ mean = b.mean + J.lev_off;
    const double* __restrict__ stdv = b.stdv + J.lev_off;
    const double* __restrict__ lsdv = b.logstdv + J.lev_off;
    double* __restrict__ em = b.em + J.mat_off[dir];
    unsigned short* __restrict__ flg = b.flg + J.mat_off[dir];
    const int lane = threadIdx.x & 63;
    const int nP = P >> 6;                                   // P is a multiple of 64
    const int64_t nunits = (J.S + REC_PAD) * nP;             // the spare anti-diagonals get zero flags (k_recur reads them)
    const int64_t nwaves = (int64_t)gridDim.x * (EMIS_T / 64);
    const int64_t per = (nunits + nwaves - 1) / nwaves;
    const int64_t u0 = min(nunits, ((int64_t)blockIdx.x * (EMIS_T / 64) + (threadIdx.x >> 6)) * per);
    const int64_t u1 = min(nunits, u0 + per);
    int s = __builtin_amdgcn_readfirstlane((int)(u0 / nP));
    int c = __builtin_amdgcn_readfirstlane((int)(u0 - (int64_t)s * nP));
    int lo = -1, lom = 0;
    bool fresh = true;
    for (int64_t u = u0; u < u1; u++) {
        if (fresh) {   // a new anti-diagonal: its first in-band row and that row's slot
            lo = s < J.S ? __builtin_amdgcn_readfirstlane(LO[s]) : -1;
            lom = lo >= 0 ? lo % P : 0;
            fresh = false;
        }
        const int slot = c * 64 + lane;
        unsigned f = 0;
        double e = 0.0;
        if (lo >= 0) {
            int d = slot - lom;
            if (d < 0) d += P;
            const int i = lo + d, j = s - i;
            if (i <= n0 && j >= 1 && j <= C) {
                int i0, i1;
                band_of(lb, dir, j, C, n0, W, i0, i1);
                if (i >= i0 && i <= i1) {
                    const int state = st[dir == 0 ? j - 1 : C - j];
                    if (state < 0) {
                        // invalid 5-mer: the whole column is zero (cpp/Alignment.cpp:162-163).  Its step word is
                        // final here; k_recur treats the cell like one outside the band (the same implicit zero for
                        // its right-hand neighbours) and k_invfix stores the zero record afterwards.
                        f = FLG_DEAD;
                    } else {
                        f = F_ACT;
                        int p0, p1;
                        if (j == 1) { p0 = 0; p1 = n0; f |= F_BLANK; }
                        else band_of(lb, dir, j - 1, C, n0, W, p0, p1);
                        if (i >= p0 && i <= p1) f |= F_VL;
                        // (a diagonal neighbour in an invalid-5-mer column is a zero as well: leave F_RD clear)
                        if (i > p0 && i <= p1) f |= F_VD | ((j == 1 || st[dir == 0 ? j - 2 : C - j + 1] < 0) ? 0u : F_RD);
                        if (i == i0) f |= F_TOP;
                        ModelRow m = {s_model[state], s_model[NS + state], s_model[2 * NS + state],
                                      s_model[3 * NS + state], s_model[4 * NS + state], s_model[5 * NS + state]};
                        // forward reads level i-1 but log_stdv[n0-i] (sic, cpp/Alignment.cpp:171-172); backward reads level n0-i
                        const int tv = dir == 0 ? i - 1 : n0 - i;
                        e = emission(m, mean[tv], stdv[tv], lsdv[n0 - i], b.log2pi, b.lik_offset);
                    }
                }
            }
        }
        const int64_t cell = (int64_t)s * P + slot;
        em[cell] = e;
        flg[cell] = (unsigned short)f;
        if (++c == nP) { c = 0; s++; fresh = true; }
    }
}

// ------------------------------------------------------------------------------------------------
// recurrence pass: one workgroup (P lanes) per (job, direction).  Values only: the maximum over the
// candidate moves equals the reference's ordered strict-'>' selection (cpp/Alignment.cpp:240-267)
// whatever the tie order, so plain fmax is exact; the step codes are derived afterwards by k_steps.
// A lane that has no in-band cell on an anti-diagonal holds zeros, which is exactly the reference's
// "implicit zero" for neighbours outside the previous band (cpp/Alignment.cpp:201-225); a missing
// upper neighbour (top row of a band) is -infinity for the stay / extend / insert moves.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}


template <int DIR>
struct RecurState {
    double cm, cs;             // this lane's latest main / stay; -infinity while it has no cell
    double pe = 0.0;           // backward pass: the emission this step's upper neighbour had one step earlier
    unsigned hist = 0;         // wave-uniform: bit k set = the wave had a cell k anti-diagonals ago
    int wa[3], ra[3];          // LDS byte offsets of this lane's slot / its upper neighbour's slot in the three buffers
};

// one anti-diagonal; PH = (anti-diagonal index) mod 3 selects the roles of the three exchange buffers.
//
// A lane without an in-band cell holds and publishes -infinity for main and stay.  Read as the upper neighbour
// (row i-1 one step ago) that is what the stay / extend / insert moves of a band's top row need; read as the left
// neighbour (the lane's own previous value) fmax(.., 0) turns it into the reference's "implicit zero" for
// neighbours outside the previous band (cpp/Alignment.cpp:201-225) and leaves real scores, which are >= 0,
// untouched.  This needs no flags because P exceeds the widest band footprint by two slots and the band edges
// move at most one row per anti-diagonal (the band centres are bisection results, hence non-decreasing in the
// column for any ref_index): a lane always idles for at least one step between two different rows.
// Cells of invalid-5-mer columns are all zero; k_emis finishes them and marks them idle.
// Two inputs keep a flag.  F_TOP: the stay matrix of a top row starts from -1e300 instead of 0.  F_RD: the
// diagonal neighbour is taken only when this cell's own row lies inside the previous column's band — the
// reference tests row i, not row i-1, against that band, so on the row just below the previous band's last row it
// reads zero although cell (i-1, j-1) exists.
// Neighbour values go through LDS for all lanes alike, in three rotating buffers: a cell reads what row i-1
// published one step ago (upper neighbour) and two steps ago (diagonal neighbour) while this step's results go
// to the third.  The backward recurrence adds the emission of the cell a move comes FROM (cpp/Alignment.cpp:
// 284-330); those are plain reads of the emission matrix one slot up, one and two anti-diagonals back (`o` is that
// stream in the backward pass, the cell's own emission in the forward pass), not part of the exchange.  A wave whose cells have all left the band runs three more all-idle steps, which put -infinity
// into its slots of all three buffers, and then only takes part in the barrier until a cell comes back.
template <int DIR, int PH>
__device__ __forceinline__ void recur_step(RecurState<DIR>& r, const double o, const unsigned f, double2* __restrict__ dst,
                                           char* __restrict__ xch, const double lsk, const double lst, const double lex,
                                           const double lin) {
    const double NINF = -__builtin_inf();
    // scalar bookkeeping: number of lanes with a cell (s_bcnt1) -> one history bit per anti-diagonal
    const unsigned nact = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(f & F_ACT));
    r.hist = ((r.hist << 1) & 14u) | min(nact, 1u);
    if (r.hist) {
        constexpr int W0 = PH, R1 = (PH + 2) % 3, R2 = (PH + 1) % 3;   // written now / one step ago / two steps ago
        // {main, stay} of row i-1 one step ago, main two steps ago
        const double2 u = *(const double2*)(xch + r.ra[R1]);
        const double dx = *(const double*)(xch + r.ra[R2]);
        const bool rd = f & F_RD;
        double L;   // max(cm, 0) in one instruction (fmax() would first canonicalise its operand)
        asm("v_max_f64 %0, %1, 0" : "=v"(L) : "v"(r.cm));
        const double D = rd ? dx : 0.0, po = rd ? r.pe : 0.0;
        const double eo = o;
        const double cSTAY = u.x + eo + lst;
        const double cEXT = u.y + eo + lex;
        const double cINS = u.x + lin;
        const double cSKIP = L + lsk;
        const double cMATCH = DIR == 0 ? D + o : D + po;
        const double cIGN = D + lin;
        double ns = (f & F_TOP) ? -BIG : 0.0;
        ns = fmax(ns, cSTAY);
        ns = fmax(ns, cEXT);
        double nm = fmax(0.0, cSKIP);
        nm = fmax(nm, cMATCH);
        nm = fmax(nm, cINS);
        nm = fmax(nm, cIGN);
        nm = fmax(nm, ns);
        const bool act = f & F_ACT;
        r.cm = act ? nm : NINF;
        r.cs = act ? ns : NINF;
        *dst = make_double2(r.cm, r.cs);
        *(double2*)(xch + r.wa[W0]) = make_double2(r.cm, r.cs);
    }
    if (DIR) r.pe = o;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int DIR>
__device__ __forceinline__ void recur_body(const BatchD& b, const JobD& J, char* __restrict__ xch) {
    const int P = J.P, slot = threadIdx.x;
    const int up_slot = slot == 0 ? P - 1 : slot - 1;
    const double lsk = b.trans[J.ev * 4 + 0], lst = b.trans[J.ev * 4 + 1], lex = b.trans[J.ev * 4 + 2], lin = b.trans[J.ev * 4 + 3];
    // forward: the cell's own emission; backward: the emission of the cell one slot up on the previous anti-diagonal
    const double* __restrict__ em = b.em + J.mat_off[DIR] + (DIR ? up_slot - P : slot);
    const unsigned short* __restrict__ flg = b.flg + J.mat_off[DIR] + slot;
    double2* __restrict__ rec = b.rec + J.mat_off[DIR] + slot;
    const int64_t S = J.S, SL = S - 1;
    constexpr int RB = 16;   // bytes per exchanged record {main, stay}
    const double NINF = -__builtin_inf();
    RecurState<DIR> r;
    r.cm = NINF; r.cs = NINF;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r.wa[k] = (k * P + slot) * RB;
        r.ra[k] = (k * P + up_slot) * RB;
        *(double2*)(xch + r.wa[k]) = make_double2(NINF, NINF);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    double eA[PF], eB[PF];
    unsigned fA[PF], fB[PF];
    // no tail handling: k_emis wrote zero flags on the REC_PAD spare anti-diagonals behind the matrix, which the
    // padded last groups read (and, as idle lanes, store to).  PF is a multiple of 3, so the buffer roles are
    // compile-time constants of the unrolled step index.
#define PS_LOAD(E, F, s0)                                                   \
    _Pragma("unroll") for (int u = 0; u < PF; u++) {                        \
        const int64_t sc = min((int64_t)(s0) + u, SL + REC_PAD) * P;        \
        E[u] = em[sc];                                                      \
        F[u] = flg[sc];                                                     \
    }
#define PS_STEP(E, F, s0, u) recur_step<DIR, (u) % 3>(r, E[u], F[u], rec + ((int64_t)(s0) + (u)) * P, xch, lsk, lst, lex, lin);
#define PS_RUN(E, F, s0) PS_STEP(E, F, s0, 0) PS_STEP(E, F, s0, 1) PS_STEP(E, F, s0, 2)
    static_assert(PF == 3, "PS_RUN is written out for three steps per group");
    PS_LOAD(eA, fA, 2)
    for (int64_t s0 = 2; s0 < S; s0 += 2 * PF) {   // every wave runs the same padded trip count
        PS_LOAD(eB, fB, s0 + PF)
        PS_RUN(eA, fA, s0)
        PS_LOAD(eA, fA, s0 + 2 * PF)
        PS_RUN(eB, fB, s0 + PF)
    }
#undef PS_LOAD
#undef PS_STEP
#undef PS_RUN
}

// zero records of the invalid-5-mer cells (k_recur leaves -infinity in idle slots); launched only for batches whose
// sequences contain an invalid 5-mer.  grid (nblk, njobs*ndir), block 256
__global__ __launch_bounds__(256) void k_invfix(BatchD b, int ndir) {
    const int jd = blockIdx.y, job = jd / ndir, dir = jd % ndir;
    const JobD& J = b.jobs[job];
    if (b.out[job].inert) return;
    const unsigned short* __restrict__ flg = b.flg + J.mat_off[dir];
    double2* __restrict__ rec = b.rec + J.mat_off[dir];
    const int64_t ncell = J.S * J.P;
    for (int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x; cell < ncell; cell += (int64_t)gridDim.x * 256)
        if (flg[cell] == FLG_DEAD) rec[cell] = make_double2(0.0, 0.0);
}

__global__ __launch_bounds__(1024) void k_recur(BatchD b, int ndir) {
    extern __shared__ double2 xch2[];   // three exchange buffers of P {main, stay} records
    char* xch = (char*)xch2;
    const int jd = blockIdx.x, job = jd / ndir, dir = jd % ndir;
    const JobD& J = b.jobs[job];
    if (b.out[job].inert) return;
    if (dir == 0) recur_body<0>(b, J, xch); else recur_body<1>(b, J, xch);
}

// ------------------------------------------------------------------------------------------------
// step / statistics pass, chip-wide and coalesced in the skewed layout:
//   * forward direction: re-derive each cell's back-pointer codes with the reference's ordered
//     strict-'>' selection from the stored neighbour values (cpp/Alignment.cpp:196-267)
//   * both directions: per-column maximum of the main matrix (-> MaxInfo per column), aggregated in
//     LDS per block of SB anti-diagonals, then one global atomic max per touched column
// grid (ceil(maxS / SB), njobs*ndir), block 256
// ------------------------------------------------------------------------------------------------
constexpr int SB = 32;
constexpr int SCOLS = 2048;

__global__ __launch_bounds__(256) void k_steps(BatchD b, int ndir) {
    __shared__ unsigned long long s_cmax[SCOLS];
    __shared__ int s_jbase, s_ok;
    const int jd = blockIdx.y, job = jd / ndir, dir = jd % ndir;
    const JobD& J = b.jobs[job];
    if (b.out[job].inert) return;
    const int64_t s0 = (int64_t)blockIdx.x * SB;
    if (s0 >= J.S) return;
    const int nst = (int)min((int64_t)SB, J.S - s0);
    const int P = J.P;
    const int* __restrict__ LO = b.lo + J.lo_off[dir];
    if (threadIdx.x == 0) {
        int jmin = 0x7fffffff, jmax = -1;
        for (int k = 0; k < nst; k++) {
            const int lo = LO[s0 + k];
            if (lo < 0) continue;
            const int s = (int)(s0 + k);
            jmax = max(jmax, s - lo);
            jmin = min(jmin, s - lo - P + 1);
        }
        s_jbase = jmin;
        s_ok = (jmax >= 0 && jmax - jmin + 1 <= SCOLS) ? 1 : (jmax < 0 ? -1 : 0);
    }
    for (int k = threadIdx.x; k < SCOLS; k += 256) s_cmax[k] = 0ull;
    __syncthreads();
    if (s_ok < 0) return;  // nothing in band on these anti-diagonals
    const bool use_lds = s_ok == 1;
    const int jbase = s_jbase;
    const double2* __restrict__ rec = b.rec + J.mat_off[dir];
    const double* __restrict__ em = b.em + J.mat_off[dir];
    unsigned short* __restrict__ flg = b.flg + J.mat_off[dir];
    unsigned long long* gcmax = (unsigned long long*)(b.cmax + J.col_off[dir]);
    const double lsk = b.trans[J.ev * 4 + 0], lst = b.trans[J.ev * 4 + 1], lex = b.trans[J.ev * 4 + 2], lin = b.trans[J.ev * 4 + 3];
    // work unit = one wave-wide run of 64 slots of one anti-diagonal (k, c): k and LO are wave-uniform
    const int nP = P >> 6, lane = threadIdx.x & 63;
    int k = 0, c = threadIdx.x >> 6;
    while (c >= nP) { c -= nP; k++; }
    for (; k < nst; ) {
        const int64_t s = s0 + k;
        const int slot = c * 64 + lane;
        const int64_t cell = s * P + slot;
        const int lo = __builtin_amdgcn_readfirstlane(LO[s]);
        const unsigned f = lo >= 0 ? flg[cell] : 0u;
        c += 4;
        while (c >= nP) { c -= nP; k++; }
        if (!(f & F_ACT)) continue;
        int d = slot - lo % P;
        if (d < 0) d += P;
        const int i = lo + d;
        const int j = (int)s - i;
        const double2 v = rec[cell];
        if (v.x > 0.0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v.x);
            if (use_lds) atomicMax(&s_cmax[j - jbase], bits); else atomicMax(&gcmax[j], bits);
        }
        if (dir == 0) {
            unsigned sm = 0, ss = 0;
            {
                const int sm1 = slot == 0 ? P - 1 : slot - 1;
                const double o = em[cell];
                double L = 0.0, D = 0.0, um = 0.0, us = 0.0;
                const bool blank = f & F_BLANK, vl = f & F_VL, vd = f & F_VD, top = f & F_TOP;
                if (vl && !blank) L = rec[cell - P].x;
                if (vd && !blank) D = rec[(s - 2) * P + sm1].x;
                if (!top) { const double2 u = rec[(s - 1) * P + sm1]; um = u.x; us = u.y; }
                const double cSKIP = vl ? L + lsk : lsk;
                const unsigned kSKIP = vl ? M_SKIP : M_IMPL;
                const double cMATCH = vd ? D + o : o;
                const unsigned kMATCH = vd ? M_MATCH : M_IMPL;
                const double cIGN = vd ? D + lin : 0.0;
                double cSTAY = -BIG, cEXT = -BIG, cINS = 0.0, ns = 0.0, nm = 0.0;
                if (top) ns = -BIG;
                else { cSTAY = um + o + lst; cINS = um + lin; cEXT = us + o + lex; }
                if (cSTAY > ns) { ns = cSTAY; ss = M_STAY; }
                if (cEXT > ns) { ns = cEXT; ss = M_EXTEND; }
                if (cSKIP > nm) { nm = cSKIP; sm = kSKIP; }
                if (cMATCH > nm) { nm = cMATCH; sm = kMATCH; }
                if (cINS > nm) { nm = cINS; sm = M_INSERT; }
                if (cIGN > nm) { nm = cIGN; sm = M_IGNORE; }
                if (ns > nm) { nm = ns; sm = M_STAY; }
            }
            // bits 14 / 15: main / stay score <= 0 (the backtrace stops there, cpp/Alignment.cpp:542) so that the
            // walker needs nothing but this word
            flg[cell] = (unsigned short)(sm | (ss << 8) | (v.x <= 0.0 ? 0x4000u : 0u) | (v.y <= 0.0 ? 0x8000u : 0u));
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int k = threadIdx.x; k < SCOLS; k += 256) {
            const unsigned long long v = s_cmax[k];
            if (v) atomicMax(&gcmax[jbase + k], v);
        }
    }
}

// prefix max over columns + (fwd) the first cell achieving the global max ; grid njobs*ndir, block 64
__global__ __launch_bounds__(64) void k_prefix(BatchD b, int ndir) {
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
    if (dir == 0) {
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
            JobOut* O = b.out + job;
            O->best = best;
            if (best > 0.0) { O->bj = bj; O->bi = bi; }
            else { O->bj = 0; O->bi = 0; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backtrace (cpp/Alignment.cpp:516-624): one wave per job, 16x16 tiles staged in LDS
// ------------------------------------------------------------------------------------------------
constexpr int BT = 64;   // tile edge of the backtrace (step words only: 2 bytes per cell)
constexpr int BTM = 8;   // the tile prefetched during a walk overlaps the current one by this many rows / columns

// cooperative load of the BT x BT step-word tile whose corner (largest row / column) is (ti, tj); NT threads, t in [0, NT)
template <int NT>
__device__ __forceinline__ void bt_load(unsigned short (*__restrict__ dst)[BT + 2], const unsigned short* __restrict__ flg,
                                        const int P, const int ti, const int tj, const int t) {
    constexpr int NQ = (BT * BT + NT - 1) / NT;
    const int sti = __builtin_amdgcn_readfirstlane(ti > 0 ? ti % P : 0);
    unsigned short tmp[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {   // all loads of a thread in flight before the first LDS store
        const int idx = min(t + NT * q, BT * BT - 1), a = idx / BT, c = idx % BT;
        const int r = ti - a, col = tj - c;
        int slot = sti - a;          // (ti - a) mod P, a < BT <= P
        if (slot < 0) slot += P;
        unsigned short st = 0x4000u | 0x8000u;   // outside the matrix: score 0, the walk stops
        if (r >= 1 && col >= 1) st = flg[(int64_t)(r + col) * P + slot];
        tmp[q] = st;
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) { const int idx = t + NT * q; if (idx < BT * BT) dst[idx / BT][idx % BT] = tmp[q]; }
}

// The walker navigates on the 16-bit step words alone (codes + "score <= 0" bits written by k_steps), so a
// tile is 64 x 64 cells = 8 KB of LDS.  Wave 0 walks the current tile while waves 1-3 fetch the tile the path
// is expected to enter next (same diagonal, BTM cells of overlap to absorb drift) into the other LDS buffer; when
// the walk leaves the current tile inside the prefetched one no load is waited for.  For every recorded level
// the walker stores ref_align directly and, in ref_like's slot, the cell it was recorded from as an integer
// (column << 1 | matrix); k_fill_like then replaces those by the stored scores in parallel.
__global__ __launch_bounds__(256) void k_backtrace(BatchD b) {
    const JobD& J = b.jobs[blockIdx.x];
    if (b.out[blockIdx.x].inert) return;  // stripe_width == 0: the event is left untouched
    const JobOut O = b.out[blockIdx.x];
    const int tid = threadIdx.x, P = J.P, n0 = J.n0;
    double* __restrict__ ra = J.ra;
    long long* __restrict__ rlw = (long long*)J.rl;
    for (int t = tid; t < n0; t += 256) { ra[t] = 0.0; rlw[t] = 0ll; }
    __syncthreads();
    const unsigned short* __restrict__ flg = b.flg + J.mat_off[0];
    __shared__ unsigned short t_buf[2][BT][BT + 2];
    __shared__ int s_state[2][4];
    int i = O.bi, j = O.bj, arr = 0;
    bool done = (i <= 0);
    int cur = 0, ti = 0, tj = 0, it = 0;
    bool need = true;
    while (!done) {
        if (need) {
            ti = i; tj = j;
            bt_load<256>(t_buf[cur], flg, P, ti, tj, tid);
            __syncthreads();
        }
        const int pi = ti - (BT - BTM), pj = tj - (BT - BTM);
        unsigned short (*t_step)[BT + 2] = t_buf[cur];
        if (tid >= 64) bt_load<192>(t_buf[cur ^ 1], flg, P, pi, pj, tid - 64);
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
                    if (l < run) { ra[i - 1 - l] = (double)(j - l); rlw[i - 1 - l] = (long long)(j - l) << 1; }
                    i -= run; j -= run;
                    if (run == 64 || a + run >= BT || c + run >= BT) continue;
                    if (i <= 0) { done = true; break; }
                }
                const unsigned wr = __builtin_amdgcn_readlane(w, run);
                const unsigned st = arr ? ((wr >> 8) & 7u) : (wr & 255u);
                if (wr & (arr ? 0x8000u : 0x4000u)) { done = true; break; }   // score <= 0
                const long long here = ((long long)j << 1) | arr;
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
                else { done = true; break; }
                if (rec && l == 0) { ra[i - 1] = rav; rlw[i - 1] = here; }
                i -= di; j -= dj;
            }
            if (tid == 0) { int* ss = s_state[it & 1]; ss[0] = i; ss[1] = j; ss[2] = arr; ss[3] = done ? 1 : 0; }
        }
        __syncthreads();
        { const int* ss = s_state[it & 1]; i = ss[0]; j = ss[1]; arr = ss[2]; done = ss[3] != 0; }
        it++;
        // continue in the prefetched tile if the path left the current one inside it
        const int a = pi - i, c = pj - j;
        need = !(a >= 0 && c >= 0 && a < BT && c < BT);
        if (!need) { cur ^= 1; ti = pi; tj = pj; }
    }
}

// ref_like[i-1] = score of the cell level i was recorded from (cpp/Alignment.cpp:610-618); grid (ceil(maxn/256), njobs)
__global__ __launch_bounds__(256) void k_fill_like(BatchD b) {
    const JobD& J = b.jobs[blockIdx.y];
    if (b.out[blockIdx.y].inert) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= J.n0) return;
    const long long w = ((const long long*)J.rl)[t];
    if (w == 0) return;   // level not on the path: ref_like stays 0
    const int i = t + 1, j = (int)(w >> 1);
    const double2 v = (b.rec + J.mat_off[0])[(int64_t)(i + j) * J.P + slot_of(i, J.P)];
    J.rl[t] = (w & 1) ? v.y : v.x;
}

// ------------------------------------------------------------------------------------------------
// columnMax(raf, rab) on the filled matrices (cpp/Alignment.h:181-214); cooperative over `nl` lanes
// Only rows where both columns are in band can exceed the two running maxima (stay <= main <= max).
// ------------------------------------------------------------------------------------------------
__device__ double colmax_pair(const BatchD& b, const JobD& J, int raf, int rab, int lane, int nl) {
    const int C = J.C, n0 = J.n0, P = J.P;
    if ((unsigned)raf >= (unsigned)(C + 1)) raf = C;
    if ((unsigned)rab >= (unsigned)(C + 1)) rab = C;
    const int* lb = b.lb + J.lb_off;
    int f0 = 0, f1 = n0, b0 = 0, b1 = n0;
    if (raf > 0) band_of(lb, 0, raf, C, n0, J.W, f0, f1);
    if (rab > 0) band_of(lb, 1, rab, C, n0, J.W, b0, b1);
    const double2* __restrict__ rf = b.rec + J.mat_off[0];
    const double2* __restrict__ rb = b.rec + J.mat_off[1];
    // jb = n0 - jf + 1 in [b0, b1]  <=>  jf in [n0 + 1 - b1, n0 + 1 - b0]
    const int lo = max(max(1, f0), n0 + 1 - b1), hi = min(min(n0, f1), n0 + 1 - b0);
    double sm = 0.0;
    for (int jf = lo + lane; jf <= hi; jf += nl) {
        const int jb = n0 - jf + 1;
        double2 fv = make_double2(0.0, 0.0), bv = make_double2(0.0, 0.0);
        if (raf > 0) fv = rf[(int64_t)(jf + raf) * P + slot_of(jf, P)];
        if (rab > 0) bv = rb[(int64_t)(jb + rab) * P + slot_of(jb, P)];
        sm = fmax(sm, fmax(fv.x + bv.x, fv.y + bv.y));
    }
    for (int off = 1; off < nl; off <<= 1) sm = fmax(sm, __shfl_xor(sm, off));
    sm = fmax(sm, b.pm[J.col_off[0] + raf]);
    sm = fmax(sm, b.pm[J.col_off[1] + rab]);
    return sm;
}

// old score for each distinct r0 = max(start - 3, 1) ; grid (nr0, njobs), block 64
__global__ __launch_bounds__(64) void k_old(BatchD b, ScoreArgs a) {
    const JobD& J = b.jobs[blockIdx.y];
    if (b.out[blockIdx.y].inert) return;
    const int r0 = a.r0[blockIdx.x];
    const double v = colmax_pair(b, J, r0, J.C - r0 + 1, threadIdx.x, 64);
    if (threadIdx.x == 0) a.old[(size_t)blockIdx.y * a.nr0 + blockIdx.x] = v;
}

// ------------------------------------------------------------------------------------------------
// scoreMutation (cpp/Alignment.cpp:447-512): G lanes per (event, edit) item, lane = new column,
// rows stream through the group systolically (lane c works on row  base + t - c  at step t).
// ------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void k_score(BatchD b, ScoreArgs a, const int* __restrict__ items, int nitems) {
    constexpr int IPW = 64 / G;
    __shared__ double s_carry[(G == 64) ? 4 * 1024 : 1];   // last column of a 64-column chunk, per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / G, c = lane % G;
    const int job = blockIdx.y;
    const JobD& J = b.jobs[job];
    const int it = (blockIdx.x * 4 + wave) * IPW + g;
    const bool have = it < nitems;
    const int m = have ? items[it] : 0;
    const int n0 = J.n0, C = J.C, P = J.P, WS = a.ws;
    const bool live = have && !b.out[job].inert && !a.m_skip[m];
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
    const double* __restrict__ mean = b.mean + J.lev_off;
    const double* __restrict__ stdv = b.stdv + J.lev_off;
    const double* __restrict__ lsdv = b.logstdv + J.lev_off;
    const double2* __restrict__ rf = b.rec + J.mat_off[0];
    const double2* __restrict__ rb = b.rec + J.mat_off[1];
    const double* gm = b.model + (size_t)J.ev * 6 * NS;
    const double lsk = b.trans[J.ev * 4 + 0], lst = b.trans[J.ev * 4 + 1], lex = b.trans[J.ev * 4 + 2], lin = b.trans[J.ev * 4 + 3];

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
        ModelRow mr = {0, 1, 0, 1, 0, 0};
        if (state >= 0) mr = {gm[state], gm[NS + state], gm[2 * NS + state], gm[3 * NS + state], gm[4 * NS + state], gm[5 * NS + state]};
        // band of the column to the left
        int p0 = __shfl_up(i0, 1), p1 = __shfl_up(i1, 1);
        if (c == 0) { p0 = pc0; p1 = pc1; }
        const int base = __shfl(i0, g * G);                       // first row of the chunk's first column
        int lastcol = min(ncol - ch * G, G) - 1;                   // lanes 0..lastcol are in use
        int span = mine ? (i1 - base) + c : -1;
        for (int off = 1; off < 64; off <<= 1) span = max(span, __shfl_xor(span, off));
        double cm = 0.0, cs = 0.0, colmax = 0.0, lprev = 0.0;
        const bool is_t = mine && cc == trel;
        for (int t = -1; t <= span; t++) {
            const int i = base + t - c;
            // value of (i, jc-1): matrix / carried chunk column for lane 0, left neighbour otherwise
            double L = wave_shr1(cm);
            if (c == 0) {
                L = 0.0;
                if (mine && i >= p0 && i <= p1 && i >= 1) {
                    if (ch > 0) L = carry[i - p0];
                    else if (sidx > 0) L = rf[(int64_t)(i + sidx) * P + slot_of(i, P)].x;
                }
            }
            const double D = lprev;
            lprev = L;
            if (mine && i >= i0 && i <= i1) {
                double nm = 0.0, ns = 0.0;
                if (state >= 0) {
                    const double o = emission(mr, mean[i - 1], stdv[i - 1], lsdv[n0 - i], b.log2pi, b.lik_offset);
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
                        if (backind > 0) bv = rb[(int64_t)(jb + backind) * P + slot_of(jb, P)];
                        tm = fmax(tm, fmax(nm + bv.x, ns + bv.y));
                    }
                }
                if (G == 64 && c == lastcol && ch + 1 < nchunk) carry[i - i0] = nm;
            }
        }
        // MaxInfo: max over the new columns up to and including the target
        double cmx = (mine && cc <= trel) ? colmax : 0.0;
        for (int off = 1; off < G; off <<= 1) cmx = fmax(cmx, __shfl_xor(cmx, off));
        fmaxv = fmax(fmaxv, cmx);
        // next chunk continues from this chunk's last column
        pc0 = __shfl(i0, g * G + max(lastcol, 0)); pc1 = __shfl(i1, g * G + max(lastcol, 0));
        __builtin_amdgcn_wave_barrier();
    }
    // gather the target lane's row maximum
    double tmx = tm;
    for (int off = 1; off < G; off <<= 1) tmx = fmax(tmx, __shfl_xor(tmx, off));
    double now;
    if (trel < 0) {
        now = !live ? 0.0 : colmax_pair(b, J, sidx, backind, c, G);   // no new column: the spliced copy is the target
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
__global__ void k_reduce(ScoreArgs a, int njobs) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.nitems_per_job) return;
    double s = -1e-6;
    for (int e = 0; e < njobs; e++) s += a.delta[(size_t)e * a.nitems_per_job + m];
    a.score[m] = s;
}

// latch the reference's "stripe_width == 0" decision (cpp/Alignment.cpp:51-59) for this API call
__global__ void k_begin(BatchD b) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b.njobs) return;
    b.out[j].inert = (b.jobs[j].force_inert || !b.out[j].has_index) ? 1 : 0;
    b.out[j].best = 0.0; b.out[j].bi = 0; b.out[j].bj = 0; b.out[j].maxw = 0;
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
    hipLaunchKernelGGL(k_lo, dim3((unsigned)((maxS + 255) / 256), b.njobs * ndir), dim3(256), 0, rt->stream, b, ndir);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int launch_fill(Runtime* rt, const BatchD& b, int ndir, int64_t maxS, int P, int64_t ncols, bool has_invalid) {
    if (!b.njobs) return PS_OK;
    // ~2048 workgroups of 16 waves over the chip, at least ~64 wave-units per wave
    int nblk = (int)std::min<int64_t>((maxS * (P / 64) + 16 * 64 - 1) / (16 * 64), std::max(4, 2048 / (b.njobs * ndir)));
    nblk = std::max(nblk, 1);
    hipLaunchKernelGGL(k_emis, dim3(nblk, b.njobs * ndir), dim3(EMIS_T), 0, rt->stream, b, ndir);
    PS_LAUNCH_CHECK();
    prof_begin(rt);
    hipLaunchKernelGGL(k_recur, dim3(b.njobs * ndir), dim3(P), 3 * P * sizeof(double2), rt->stream, b, ndir);
    PS_LAUNCH_CHECK();
    prof_end(rt, "fill", 0.0);
    if (has_invalid) {
        hipLaunchKernelGGL(k_invfix, dim3(std::max(1, 1024 / (b.njobs * ndir)), b.njobs * ndir), dim3(256), 0, rt->stream, b, ndir);
        PS_LAUNCH_CHECK();
    }
    PS_HIP(hipMemsetAsync(b.cmax, 0, ncols * sizeof(double), rt->stream));
    hipLaunchKernelGGL(k_steps, dim3((unsigned)((maxS + SB - 1) / SB), b.njobs * ndir), dim3(256), 0, rt->stream, b, ndir);
    PS_LAUNCH_CHECK();
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

int launch_begin(Runtime* rt, const BatchD& b) {
    if (!b.njobs) return PS_OK;
    hipLaunchKernelGGL(k_begin, dim3((b.njobs + 63) / 64), dim3(64), 0, rt->stream, b);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int launch_score(Runtime* rt, const BatchD& b, const ScoreArgs& a, const int* const cls_items[4], const int cls_count[4]) {
    if (!b.njobs || !a.nitems_per_job) return PS_OK;
    if (a.nr0 > 0) {
        hipLaunchKernelGGL(k_old, dim3(a.nr0, b.njobs), dim3(64), 0, rt->stream, b, a);
        PS_LAUNCH_CHECK();
    }
    prof_begin(rt);
    for (int k = 0; k < 4; k++) {
        const int n = cls_count[k];
        if (!n) continue;
        const int G = 8 << k, ipb = 4 * (64 / G);
        dim3 grid((n + ipb - 1) / ipb, b.njobs), block(256);
        if (k == 0) hipLaunchKernelGGL(k_score<8>, grid, block, 0, rt->stream, b, a, cls_items[k], n);
        else if (k == 1) hipLaunchKernelGGL(k_score<16>, grid, block, 0, rt->stream, b, a, cls_items[k], n);
        else if (k == 2) hipLaunchKernelGGL(k_score<32>, grid, block, 0, rt->stream, b, a, cls_items[k], n);
        else hipLaunchKernelGGL(k_score<64>, grid, block, 0, rt->stream, b, a, cls_items[k], n);
        PS_LAUNCH_CHECK();
    }
    prof_end(rt, "score", 0.0);
    hipLaunchKernelGGL(k_reduce, dim3((a.nitems_per_job + 255) / 256), dim3(256), 0, rt->stream, a, b.njobs);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // namespace ps
