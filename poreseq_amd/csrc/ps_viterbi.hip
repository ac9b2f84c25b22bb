// ps_viterbi.hip — the 1024-state consensus Viterbi of ViterbiMutate (cpp/Viterbi.cpp:239-426).
//
//   k_vit_obs    per reference position: emissions of all 1024 5-mers for every contributing
//                event, ascending sort across events, drop the lowest quarter, mean
//                (cpp/Viterbi.cpp:272-349) — chip-wide, one block per position.
//   k_vit_steps  the sequential part (V_LIK::V_LIK, cpp/Viterbi.cpp:39-102): one 1024-lane
//                workgroup, state vectors double-buffered in LDS.  The 4 + 16 + 64 predecessors
//                of a destination state depend only on its low bits, so the per-family maximum,
//                its first index and the forward-probability sum are computed once per family
//                (336 families) instead of once per state.  The first-strict-maximum rule of the
//                reference is kept exactly: a destination state falls back to the plain ordered scan in
//                the (sub-ulp) case where a value just below the family maximum would round to the same sum.
//   k_vit_trace  nkeep stochastic back-traces (randbp, cpp/Viterbi.cpp:105-131), one block each;
//                the uniform deviates are drawn on the host from libc rand() in the reference's
//                call order.
// Max-plus values (liks, back-pointers) are bit-exact; forward probabilities use device exp / log
// (fwd^atten as exp(atten * log fwd)) and tree sums, i.e. agree to a few ulp; they only weight the
// random back-traces.
#include <atomic>

#include "ps_internal.h"

namespace ps {

struct ModelRowV { double mu, sg, lsg, sm, lam, llam; };
__device__ __forceinline__ double emission_v(const ModelRowV& m, double x, double sd, double lsd, double log2pi) {
    double d = (x - m.mu) / m.sg;
    double l = -0.5 * (d * d + log2pi) - m.lsg;
    double e = (sd - m.sm) / m.sm;
    l += 0.5 * (m.llam - 3 * lsd - log2pi - e * e * m.lam / sd);
    return l;
}

// one region (AlignData) of a batched ViterbiMutate call
struct VitReg {
    const double* model;   // [E][6][1024]
    int E, T;              // events, reference positions kept
    int64_t in_off;        // into obsin (doubles)
    int64_t t_off;         // first position of the region in obs / eobs / bp / lfwd (rows of 1024)
};

// obsin[t][e][4] = {level mean, sd mean, log(sd mean), present}; obs[t][1024].  VMAXE = events one thread can sort in its
// private array: 64 covers the reference's max_coverage of 30 reads (60 events); 256 is the slow build for deeper stacks
template <int VMAXE>
__global__ __launch_bounds__(256) void k_vit_obs(const VitReg* __restrict__ regs, const int* __restrict__ pos_reg, const double* __restrict__ obsin,
                                                 double log2pi, double* __restrict__ obs, double* __restrict__ eobs) {
    const int t = blockIdx.x;
    const VitReg R = regs[pos_reg[t]];
    const int E = R.E;
    const double* in = obsin + R.in_off + (size_t)(t - R.t_off) * E * 4;
    for (int st = threadIdx.x; st < NS; st += 256) {
        double v[VMAXE];
        int nl = 0;
        for (int e = 0; e < E; e++) {
            if (in[e * 4 + 3] == 0.0) continue;
            const double* gm = R.model + (size_t)e * 6 * NS;
            ModelRowV m = {gm[st], gm[NS + st], gm[2 * NS + st], gm[3 * NS + st], gm[4 * NS + st], gm[5 * NS + st]};
            const double l = emission_v(m, in[e * 4 + 0], in[e * 4 + 1], in[e * 4 + 2], log2pi);
            // insertion into ascending order (std::sort result is unique for distinct/equal doubles)
            int k = nl++;
            while (k > 0 && v[k - 1] > l) { v[k] = v[k - 1]; k--; }
            v[k] = l;
        }
        double r;
        if (nl > 1) {
            int drop = (int)floor(nl * 0.25);
            if (drop > nl - 2) drop = 0;
            double s = 0.0;
            for (int k = drop; k < nl; k++) s += v[k];
            r = s / (double)(nl - drop);
        } else {
            r = nl == 1 ? v[0] : 0.0;
        }
        obs[(size_t)t * NS + st] = r;
        eobs[(size_t)t * NS + st] = exp(r);   // V_LIK multiplies the forward sum by exp(obs), cpp/Viterbi.cpp:91
    }
}

// The same with the per-state list of emissions in LDS instead of a private array (which the compiler keeps in scratch memory:
// 70 GB of HBM traffic per launch at 20 regions): thread t owns the column v[k * 256 + t] (one bank group per 32 threads:
// conflict-free), k < VE = events of the deepest region of the call (up to 72: 144 KB of the CU's 160 KB).
__global__ __launch_bounds__(256) void k_vit_obs_lds(const VitReg* __restrict__ regs, const int* __restrict__ pos_reg, const double* __restrict__ obsin,
                                                     double log2pi, double* __restrict__ obs, double* __restrict__ eobs) {
    extern __shared__ double vit_v[];
    const int t = blockIdx.x;
    const VitReg R = regs[pos_reg[t]];
    const int E = R.E;
    const double* in = obsin + R.in_off + (size_t)(t - R.t_off) * E * 4;
    double* v = vit_v + threadIdx.x;
    for (int st = threadIdx.x; st < NS; st += 256) {
        int nl = 0;
        for (int e = 0; e < E; e++) {
            if (in[e * 4 + 3] == 0.0) continue;
            const double* gm = R.model + (size_t)e * 6 * NS;
            ModelRowV m = {gm[st], gm[NS + st], gm[2 * NS + st], gm[3 * NS + st], gm[4 * NS + st], gm[5 * NS + st]};
            const double l = emission_v(m, in[e * 4 + 0], in[e * 4 + 1], in[e * 4 + 2], log2pi);
            // insertion into ascending order (std::sort result is unique for distinct/equal doubles)
            int k = nl++;
            while (k > 0) {
                const double w = v[(k - 1) * 256];
                if (!(w > l)) break;
                v[k * 256] = w;
                k--;
            }
            v[k * 256] = l;
        }
        double r;
        if (nl > 1) {
            int drop = (int)floor(nl * 0.25);
            if (drop > nl - 2) drop = 0;
            double s = 0.0;
            for (int k = drop; k < nl; k++) s += v[k * 256];
            r = s / (double)(nl - drop);
        } else {
            r = nl == 1 ? v[0] : 0.0;
        }
        obs[(size_t)t * NS + st] = r;
        eobs[(size_t)t * NS + st] = exp(r);   // V_LIK multiplies the forward sum by exp(obs), cpp/Viterbi.cpp:91
    }
}

// workgroup barrier that waits for LDS traffic only: __syncthreads() would also drain vmcnt(0), i.e.
// expose a full HBM round trip for the prefetched inputs and the streaming stores on every step
#define PS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Statistics of one predecessor family (the 4 / 16 / 64 states a destination can come from with a
// 1 / 2 / 3-base advance): maximum of the previous Viterbi scores, the SMALLEST state index among the
// maxima (= the first one the reference's ordered scan meets, since it scans in ascending state order)
// the largest value strictly below the maximum, and the sum of the previous forward probabilities.
// All four combine commutatively.
struct Fam { double mx, second, psum; int idx; int pad; };   // second: largest value strictly below mx

__device__ __forceinline__ Fam fam_join(const Fam& A, const Fam& B) {
    Fam r;
    r.mx = fmax(A.mx, B.mx);
    const int ia = A.mx == r.mx ? A.idx : 0x7fffffff, ib = B.mx == r.mx ? B.idx : 0x7fffffff;
    r.idx = min(ia, ib);
    const double lower = A.mx == B.mx ? -BIG * 10 : fmin(A.mx, B.mx);
    r.second = fmax(fmax(A.second, B.second), lower);
    r.psum = A.psum + B.psum;
    r.pad = 0;
    return r;
}

// DPP row shift: lane i receives lane i + N of its 16-lane row (zero past the row end)
template <int N>
__device__ __forceinline__ double row_shl(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ Fam fam_shl(const Fam& f) {
    Fam o;
    o.mx = row_shl<N>(f.mx); o.second = row_shl<N>(f.second); o.psum = row_shl<N>(f.psum);
    o.idx = __builtin_amdgcn_update_dpp(0, f.idx, 0x100 + N, 0xf, 0xf, true); o.pad = 0;
    return o;
}
// sum over the 64 lanes of a wave: DPP inside the rows, then the four row sums through SGPRs
__device__ __forceinline__ double wave_sum(double v) {
    v += row_shl<1>(v); v += row_shl<2>(v); v += row_shl<4>(v); v += row_shl<8>(v);
    auto lane_d = [&](int l) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
        return __hiloint2double(hi, lo);
    };
    return (lane_d(0) + lane_d(16)) + (lane_d(32) + lane_d(48));
}

// next double below x (x finite, non-zero)
__device__ __forceinline__ double below(double x) {
    long long b = __double_as_longlong(x);
    b += x < 0.0 ? 1 : -1;
    return __longlong_as_double(b);
}

#define VPAD(q) ((q) + ((q) >> 5))   // LDS padding: the 16/64/256-strided family reads stay conflict-free

// families: s_fam[0..255] (1-base advance, key = state mod 256), [256..319] (2-base, mod 64), [320..335] (3-base, mod 16)
// computed concurrently by waves 0-3, 4-7 and 8-11 straight from the previous score vector.
//
// Exactness of the back-pointers.  The reference takes, per destination c and advance length j, the
// first predecessor (ascending state order) whose fl(a + lik[p]) is largest, a = obs[c] + log-weight.
// fl(a + x) is monotone in x, so that is the smallest-index maximum of the family — unless a SMALLER
// value rounds to the same sum.  That can only happen if fl(a + second) == fl(a + mx), second being the
// family's largest value below its maximum; in that (sub-ulp) case the destination falls back to the
// reference's ordered scan.
//
// Forward probabilities: the reference renormalises the 1024-vector after every step
// (normvec, cpp/Viterbi.cpp:101).  Only ratios within one step's vector are ever used (randbp
// renormalises its own products), so here the vector is rescaled every 8 steps by an exact power of
// two: it stays in range, no rounding is added, and no reduce -> divide chain sits on the step path.
__global__ __launch_bounds__(1024) void k_vit_steps(const VitReg* __restrict__ regs, const double* __restrict__ obs_all, const double* __restrict__ eobs_all,
                                                    double skip, double stay, double lskip, double lstay, double l25,
                                                    short* __restrict__ bp_all, double* __restrict__ lfwd_all,
                                                    double* __restrict__ lik_final_all, int keep_fwd) {
    // one workgroup per region of the batch
    const int T = regs[blockIdx.x].T;
    const size_t roff = (size_t)regs[blockIdx.x].t_off * NS;
    const double* __restrict__ obs = obs_all + roff;
    const double* __restrict__ eobs = eobs_all + roff;
    short* __restrict__ bp = bp_all + roff;
    double* __restrict__ lfwd_out = lfwd_all + (keep_fwd ? roff : 0);
    double* __restrict__ lik_final = lik_final_all + (size_t)blockIdx.x * NS;
    if (T <= 0) return;
    __shared__ double s_lik[2][NS + NS / 32], s_fwd[2][NS + NS / 32];
    __shared__ Fam s_fam[336];
    __shared__ double s_red[16];
    __shared__ double s_scale;
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    s_lik[0][VPAD(c)] = 0.0;
    s_fwd[0][VPAD(c)] = 1.0 / NS;
    if (c == 0) s_scale = 1.0;
    __syncthreads();
    const double sp1 = 0.25, sp2 = sp1 * 0.25 * skip, sp3 = sp2 * 0.25 * skip;
    const double lsp1 = l25, lsp2 = lsp1 + l25 + lskip, lsp3 = lsp2 + l25 + lskip;
    // family-pass role of this thread: level fj (0 = none), family fg, 4 members q = fq0 + k * fqs
    int fj = 0, fg = 0, fq0 = 0, fqs = 0;
    if (c < 256) { fj = 1; fg = c; fq0 = c; fqs = 256; }
    else if (c < 512) { const int u = c - 256; fj = 2; fg = u >> 2; fq0 = fg + 256 * (u & 3); fqs = 64; }
    else if (c < 768) { const int u = c - 512; fj = 3; fg = u >> 4; fq0 = fg + 64 * (u & 15); fqs = 16; }
    int cur = 0;
    constexpr int VPF = 4;   // steps of obs / exp(obs) kept in flight in registers
    double oA[VPF], eA[VPF], oB[VPF], eB[VPF];
    const int TL = T - 1;
#define VIT_LOAD(O, E, t0)                                             \
    _Pragma("unroll") for (int u = 0; u < VPF; u++) {                   \
        const size_t at = (size_t)min((t0) + u, TL) * NS + c;           \
        O[u] = obs[at]; E[u] = eobs[at];                                \
    }
#define VIT_RUN(O, E, t0)                                               \
    _Pragma("unroll") for (int u = 0; u < VPF; u++) {                   \
        if ((t0) + u < T) vit_one((t0) + u, O[u], E[u]);                \
    }
    auto vit_one = [&](const int t, const double o, const double eo) {
        const double* pl = s_lik[cur];
        const double* pf = s_fwd[cur];
        const bool rescale = (t & 7) == 7;
        if (fj) {
            Fam m[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int q = fq0 + k * fqs;
                m[k].mx = pl[VPAD(q)]; m[k].idx = q; m[k].second = -BIG * 10; m[k].psum = pf[VPAD(q)]; m[k].pad = 0;
            }
            Fam f = fam_join(fam_join(m[0], m[1]), fam_join(m[2], m[3]));
            if (fj >= 2) { f = fam_join(f, fam_shl<1>(f)); f = fam_join(f, fam_shl<2>(f)); }
            if (fj == 3) { f = fam_join(f, fam_shl<4>(f)); f = fam_join(f, fam_shl<8>(f)); }
            const bool writer = fj == 1 || (fj == 2 && (c & 3) == 0) || (fj == 3 && (c & 15) == 0);
            if (writer) s_fam[(fj == 1 ? 0 : fj == 2 ? 256 : 320) + fg] = f;
        } else if (c == 1023 && rescale) {
            // the vector written at the end of this step is divided by 2^e, e from the previous total
            double ptot = 0.0;
            for (int w = 0; w < 16; w++) ptot += s_red[w];
            int ex;
            frexp(ptot, &ex);
            s_scale = ldexp(1.0, -ex);
        }
        PS_LDS_BARRIER();
        double best = -BIG; int bq = -1; double fsum = 0.0;
#pragma unroll
        for (int j = 1; j <= 3; j++) {
            const int g = j == 1 ? (c >> 2) : j == 2 ? (c >> 4) : (c >> 6);
            const Fam f = s_fam[(j == 1 ? 0 : j == 2 ? 256 : 320) + g];
            const double a = o + (j == 1 ? lsp1 : j == 2 ? lsp2 : lsp3);
            const double m = a + f.mx;
            fsum += (j == 1 ? sp1 : j == 2 ? sp2 : sp3) * f.psum;
            if (f.second > -BIG * 5 && a + f.second == m) {
                // a smaller member rounds to the same sum and may come first: ordered scan (reference order)
                const int cnt = 1 << (2 * j), sh = 10 - 2 * j;
                for (int k = 0; k < cnt; k++) {
                    const int q = g + (k << sh);
                    const double l = a + pl[VPAD(q)];
                    if (l > best) { best = l; bq = q; }
                }
            } else if (m > best) { best = m; bq = f.idx; }
        }
        {
            const double l = o + lstay + pl[VPAD(c)];
            if (l > best) { best = l; bq = c; }
            fsum += stay * pf[VPAD(c)];
        }
        double nf = fsum * eo;
        if (rescale) nf *= s_scale;
        if ((t & 7) == 6) {   // totals for the next rescale, one step ahead of their use
            const double wsum = wave_sum(nf);
            if (lane == 0) s_red[wave] = wsum;
        }
        s_lik[cur ^ 1][VPAD(c)] = best;
        s_fwd[cur ^ 1][VPAD(c)] = nf;
        bp[(size_t)t * NS + c] = (short)bq;
        if (keep_fwd) lfwd_out[(size_t)t * NS + c] = nf;
        PS_LDS_BARRIER();
        cur ^= 1;
    };
    VIT_LOAD(oA, eA, 0)
    for (int t0 = 0; t0 < T; t0 += 2 * VPF) {
        VIT_LOAD(oB, eB, t0 + VPF)
        VIT_RUN(oA, eA, t0)
        VIT_LOAD(oA, eA, t0 + 2 * VPF)
        VIT_RUN(oB, eB, t0 + VPF)
    }
#undef VIT_LOAD
#undef VIT_RUN
    lik_final[c] = s_lik[cur][VPAD(c)];
}

// T[cur][p] of buildT (cpp/Viterbi.cpp:134-168) in closed form: predecessor p is reached by a j-base
// advance iff its low 10-2j bits equal cur's high bits; contributions add in j order; the diagonal is
// overwritten with the stay probability.
__device__ __forceinline__ double trans_weight(int cur, int p, double skip, double stay) {
    double t = 0.0, sp = 0.25;
#pragma unroll
    for (int j = 1; j <= 4; j++) {
        if ((p & ((1 << (10 - 2 * j)) - 1)) == (cur >> (2 * j))) t += sp;
        sp = sp * 0.25 * skip;
    }
    if (p == cur) t = stay;
    return t;
}

// fwd -> log(fwd) in place, chip-wide: the back-traces need fwd^atten = exp(atten * log fwd)
__global__ void k_vit_log(double* __restrict__ v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = log(v[i]);
}

// DPP row shift the other way: lane i receives lane i - N of its 16-lane row (zero below the row start)
template <int N>
__device__ __forceinline__ double row_shr(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_get(double v, int l) {   // l wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// inclusive prefix sum over the 64 lanes of a wave (in lane order)
__device__ __forceinline__ double wave_scan(double v, int lane) {
    v += row_shr<1>(v); v += row_shr<2>(v); v += row_shr<4>(v); v += row_shr<8>(v);
    const double t0 = lane_get(v, 15), t1 = lane_get(v, 31), t2 = lane_get(v, 47);
    const int row = lane >> 4;
    const double add = row == 0 ? 0.0 : row == 1 ? t0 : row == 2 ? t0 + t1 : (t0 + t1) + t2;
    return v + add;
}

// nkeep stochastic back-traces (randbp, cpp/Viterbi.cpp:105-131): one 512-thread block per path,
// thread t owns states 2t, 2t+1 (so running sums are in state order).  The reference normalises the
// 1024 products and walks their running sum until it exceeds r; here the raw running sum is compared
// with r * total (same choice up to rounding of the forward weights).  grid nkeep.
constexpr int VT_THREADS = 512;                 // threads per back-trace
constexpr int VT_SPT = NS / VT_THREADS;         // consecutive states per thread (2)
constexpr int VT_WAVES = VT_THREADS / 64;       // 8 waves: two per SIMD, half the per-wave instruction stream of 4 x 4 states
__global__ __launch_bounds__(VT_THREADS) void k_vit_trace(const VitReg* __restrict__ regs, const double* __restrict__ lfwd_all, const int* __restrict__ starts,
                                                         double skip, double stay, const double* __restrict__ atten,
                                                         const double* __restrict__ rnd_all, short* __restrict__ path_all) {
    // grid (nkeep, regions); a region's deviates / paths are [nkeep][T] blocks at nkeep * t_off
    const int T = regs[blockIdx.y].T;
    if (T <= 0) return;
    const int start = starts[blockIdx.y];
    const double* __restrict__ lfwd = lfwd_all + (size_t)regs[blockIdx.y].t_off * NS;
    const double* __restrict__ rnd = rnd_all + (size_t)gridDim.x * regs[blockIdx.y].t_off;
    short* __restrict__ path = path_all + (size_t)gridDim.x * regs[blockIdx.y].t_off;
    __shared__ double s_wtot[2][VT_WAVES];
    __shared__ int s_pick[2][VT_WAVES];
    const int k = blockIdx.x, t = threadIdx.x, l = t & 63, w = t >> 6;
    const double at = atten[k];
    int cur = start;
    double2 lf = *(const double2*)(lfwd + (size_t)(T - 1) * NS + VT_SPT * t);
    // fwd^atten of this thread's states: independent of the state being traced, so the next row's powers are
    // computed while this row's scan and barriers are in flight
    double ex0 = exp(at * lf.x), ex1 = exp(at * lf.y);
    double rcur = rnd[(size_t)k * T];
    for (int i = T - 1; i >= 0; i--) {
        const int par = i & 1;
        if (t == 0) path[(size_t)k * T + i] = (short)cur;
        double2 nx = lf;
        double rnx = rcur;
        if (i > 0) { nx = *(const double2*)(lfwd + (size_t)(i - 1) * NS + VT_SPT * t); rnx = rnd[(size_t)k * T + (T - i)]; }
        const double tv0 = trans_weight(cur, VT_SPT * t, skip, stay), tv1 = trans_weight(cur, VT_SPT * t + 1, skip, stay);
        const double c0 = tv0 == 0.0 ? 0.0 : tv0 * ex0;
        const double c1 = c0 + (tv1 == 0.0 ? 0.0 : tv1 * ex1);
        const double nex0 = exp(at * nx.x), nex1 = exp(at * nx.y);
        const double incl = wave_scan(c1, l);
        if (l == 63) s_wtot[par][w] = incl;
        PS_LDS_BARRIER();
        double before = 0.0, total = 0.0;   // running sums over the waves in state order
#pragma unroll
        for (int q = 0; q < VT_WAVES; q++) {
            const double wq = s_wtot[par][q];
            if (q == w) before = total;
            total = q ? total + wq : wq;
        }
        const double base = before + (incl - c1);
        const double thr = rcur * total;
        int pick = 0x7fffffff;
        if (thr < base + c1) pick = thr < base + c0 ? VT_SPT * t : VT_SPT * t + 1;
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(pick != 0x7fffffff);
        int wpick = 0x7fffffff;
        if (hit) wpick = __builtin_amdgcn_readlane(pick, __builtin_ctzll(hit));
        if (l == 0) s_pick[par][w] = wpick;
        PS_LDS_BARRIER();
        int pk = 0x7fffffff;
#pragma unroll
        for (int q = VT_WAVES - 1; q >= 0; q--) { const int pq = s_pick[par][q]; pk = pq != 0x7fffffff ? pq : pk; }
        cur = pk == 0x7fffffff ? NS - 1 : pk;
        lf = nx; rcur = rnx; ex0 = nex0; ex1 = nex1;
    }
}

int viterbi_device_multi(Runtime* rt, const std::vector<VitRegionH>& regions, int nkeep, double skip, double stay, double mmin, double mmax,
                         std::vector<std::vector<std::vector<int>>>* paths) {
    const int R = (int)regions.size();
    paths->assign(R, {});
    std::vector<VitReg> regs(R);
    int64_t ttot = 0, intot = 0;
    int maxE = 0;
    for (int r = 0; r < R; r++) {
        regs[r].model = regions[r].d_model; regs[r].E = regions[r].E; regs[r].T = regions[r].T;
        regs[r].in_off = intot; regs[r].t_off = ttot;
        intot += (int64_t)regions[r].T * regions[r].E * 4; ttot += regions[r].T;
        maxE = std::max(maxE, regions[r].E);
    }
    if (ttot <= 0) return PS_OK;
    if (maxE > 256) return fail(PS_ERR_UNSUPPORTED, "ViterbiMutate: more than 256 events");
    std::vector<int> pos_reg((size_t)ttot);
    for (int r = 0; r < R; r++) std::fill(pos_reg.begin() + regs[r].t_off, pos_reg.begin() + regs[r].t_off + regs[r].T, r);
    PS_TRY(rt->buf("vit_regs").ensure(R * sizeof(VitReg)));
    PS_TRY(rt->buf("vit_posreg").ensure((size_t)ttot * sizeof(int)));
    PS_TRY(rt->buf("vit_in").ensure((size_t)std::max<int64_t>(intot, 1) * sizeof(double)));
    // The position x state tables (8 KB per position and table: 5.3 GB for 20 regions of 10 kb) live in the score-matrix pool: no
    // alignment is in flight during a ViterbiMutate call, and a runtime that kept them beside the matrices would hold ~2 % of an
    // MI355X for a phase that is 4 % of a schedule (seven runtimes: 38 GB that the matrices of the other phases can use).
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_obs = al((size_t)ttot * NS * sizeof(double)), b_fwd = al((size_t)(nkeep ? ttot : 1) * NS * sizeof(double)),
                 b_bp = al((size_t)ttot * NS * sizeof(short));
    DBuf& arena = rt->buf("rec");
    PS_TRY(arena.ensure(2 * b_obs + b_fwd + b_bp));
    PS_TRY(rt->buf("vit_lik").ensure((size_t)R * NS * sizeof(double)));
    if (nkeep > 0) {   // every allocation of the call before the first deviate is drawn: a caller may cut the batch again on PS_ERR_NOMEM
        PS_TRY(rt->buf("vit_att").ensure(nkeep * sizeof(double)));
        PS_TRY(rt->buf("vit_start").ensure(R * sizeof(int)));
        PS_TRY(rt->buf("vit_rnd").ensure((size_t)nkeep * ttot * sizeof(double)));
        PS_TRY(rt->buf("vit_path").ensure((size_t)nkeep * ttot * sizeof(short)));
    }
    VitReg* d_regs = rt->buf("vit_regs").as<VitReg>();
    int* d_posreg = rt->buf("vit_posreg").as<int>();
    double* d_in = rt->buf("vit_in").as<double>();
    char* abase = (char*)arena.p;
    double* d_obs = (double*)abase;
    double* d_eobs = (double*)(abase + b_obs);
    double* d_fwd = (double*)(abase + 2 * b_obs);
    short* d_bp = (short*)(abase + 2 * b_obs + b_fwd);
    double* d_lik = rt->buf("vit_lik").as<double>();
    PS_TRY(rt->up(d_regs, regs.data(), R * sizeof(VitReg)));
    PS_TRY(rt->up(d_posreg, pos_reg.data(), (size_t)ttot * sizeof(int)));
    for (int r = 0; r < R; r++)
        if (regions[r].T) PS_TRY(rt->up(d_in + regs[r].in_off, regions[r].obsin, (size_t)regions[r].T * regions[r].E * 4 * sizeof(double)));
    prof_begin(rt);
    bool lds_ok = false;
    if (maxE <= 72) {   // the LDS column sort needs 144 KB of dynamic LDS at 72 events: when the attribute cannot be had, the register / scratch kernels below serve
        static std::atomic<int> attr(0);   // 0: untried, 1: granted, -1: refused
        if (attr.load(std::memory_order_acquire) == 0) {
            const hipError_t e = hipFuncSetAttribute((const void*)k_vit_obs_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) (void)hipGetLastError();
            attr.store(e == hipSuccess ? 1 : -1, std::memory_order_release);
        }
        lds_ok = attr.load(std::memory_order_acquire) == 1;
    }
    if (lds_ok)
        hipLaunchKernelGGL(k_vit_obs_lds, dim3((unsigned)ttot), dim3(256), (size_t)256 * std::max(maxE, 1) * sizeof(double), rt->stream, d_regs, d_posreg, d_in,
                           std::log(2 * M_PI), d_obs, d_eobs);
    else if (maxE <= 64) hipLaunchKernelGGL(k_vit_obs<64>, dim3((unsigned)ttot), dim3(256), 0, rt->stream, d_regs, d_posreg, d_in, std::log(2 * M_PI), d_obs, d_eobs);
    else hipLaunchKernelGGL(k_vit_obs<256>, dim3((unsigned)ttot), dim3(256), 0, rt->stream, d_regs, d_posreg, d_in, std::log(2 * M_PI), d_obs, d_eobs);
    hipLaunchKernelGGL(k_vit_steps, dim3(R), dim3(1024), 0, rt->stream, d_regs, d_obs, d_eobs, skip, stay, std::log(skip), std::log(stay),
                       std::log(0.25), d_bp, d_fwd, d_lik, nkeep ? 1 : 0);
    PS_HIP(hipGetLastError());
    double* lik = nullptr;
    PS_TRY(rt->down(&lik, d_lik, (size_t)R * NS));
    // the uniform deviates of the stochastic back-traces are drawn on the host while the recursion runs: per region in the
    // reference's call order (for each kept path, one per back-step, cpp/Viterbi.cpp:108) from the region's own generator
    double* h_rand = nullptr;
    if (nkeep > 0) {
        h_rand = (double*)rt->stage.alloc((size_t)nkeep * ttot * sizeof(double));
        if (!h_rand) return fail(PS_ERR_NOMEM, "hipHostMalloc (staging arena)");
        for (int r = 0; r < R; r++) regions[r].draw(regions[r].rng, h_rand + (size_t)nkeep * regs[r].t_off, (size_t)nkeep * regions[r].T);
    }
    PS_HIP(hipStreamSynchronize(rt->stream));
    std::vector<int> starts(R, 0);
    for (int r = 0; r < R; r++) starts[r] = (int)(std::max_element(lik + (size_t)r * NS, lik + (size_t)(r + 1) * NS) - (lik + (size_t)r * NS));
    if (nkeep == 0) {
        short* bp = nullptr;
        PS_TRY(rt->down(&bp, d_bp, (size_t)ttot * NS));
        PS_HIP(hipStreamSynchronize(rt->stream));
        prof_end(rt, "viterbi", (double)ttot * NS * (8.0 * maxE + 8 + 2));
        for (int r = 0; r < R; r++) {
            if (!regions[r].T) continue;
            std::vector<int> p(regions[r].T);
            int c = starts[r];
            for (int i = regions[r].T - 1; i >= 0; i--) { p[i] = c; c = bp[((size_t)regs[r].t_off + i) * NS + c]; }
            (*paths)[r].push_back(p);
        }
        return PS_OK;
    }
    std::vector<double> att(nkeep);
    for (int k = 0; k < nkeep; k++) att[k] = mmin + (mmax - mmin) * k / (double)nkeep;
    PS_TRY(rt->up(rt->buf("vit_att").p, att.data(), nkeep * sizeof(double)));
    PS_TRY(rt->up(rt->buf("vit_start").p, starts.data(), R * sizeof(int)));
    PS_HIP(hipMemcpyAsync(rt->buf("vit_rnd").p, h_rand, (size_t)nkeep * ttot * sizeof(double), hipMemcpyHostToDevice, rt->stream));
    hipLaunchKernelGGL(k_vit_log, dim3((unsigned)(((size_t)ttot * NS + 255) / 256)), dim3(256), 0, rt->stream, d_fwd, (size_t)ttot * NS);
    hipLaunchKernelGGL(k_vit_trace, dim3(nkeep, R), dim3(VT_THREADS), 0, rt->stream, d_regs, d_fwd, rt->buf("vit_start").as<int>(), skip, stay,
                       rt->buf("vit_att").as<double>(), rt->buf("vit_rnd").as<double>(), rt->buf("vit_path").as<short>());
    PS_HIP(hipGetLastError());
    prof_end(rt, "viterbi", (double)ttot * NS * (8.0 * maxE + 8 + 2 + 8 + 8.0 * nkeep));
    short* hp = nullptr;
    PS_TRY(rt->down(&hp, rt->buf("vit_path").p, (size_t)nkeep * ttot));
    PS_HIP(hipStreamSynchronize(rt->stream));
    for (int r = 0; r < R; r++) {
        const int T = regions[r].T;
        if (!T) continue;
        const short* rp = hp + (size_t)nkeep * regs[r].t_off;
        for (int k = 0; k < nkeep; k++) {
            std::vector<int> p(T);
            for (int i = 0; i < T; i++) p[i] = rp[(size_t)k * T + i];
            (*paths)[r].push_back(p);
        }
    }
    return PS_OK;
}

}  // namespace ps
