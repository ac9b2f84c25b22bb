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
//                reference is kept exactly: a family remembers the largest value that precedes its
//                first maximum, and a destination state falls back to the plain ordered scan in the
//                (sub-ulp) case where that value would round to the same sum.
//   k_vit_trace  nkeep stochastic back-traces (randbp, cpp/Viterbi.cpp:105-131), one block each;
//                the uniform deviates are drawn on the host from libc rand() in the reference's
//                call order.
// Max-plus values (liks, back-pointers) are bit-exact; forward probabilities use device exp /
// pow and tree sums, i.e. agree to a few ulp (they only weight the random back-traces).
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

constexpr int VMAXE = 64;  // events handled per position without spilling to the slow path

// obsin[t][e][4] = {level mean, sd mean, log(sd mean), present}; obs[t][1024]
__global__ __launch_bounds__(256) void k_vit_obs(const double* __restrict__ obsin, const double* __restrict__ model,
                                                 int E, double log2pi, double* __restrict__ obs) {
    const int t = blockIdx.x;
    const double* in = obsin + (size_t)t * E * 4;
    for (int st = threadIdx.x; st < NS; st += 256) {
        double v[VMAXE];
        int nl = 0;
        for (int e = 0; e < E; e++) {
            if (in[e * 4 + 3] == 0.0) continue;
            const double* gm = model + (size_t)e * 6 * NS;
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
    }
}

struct Fam { double mx, prev, fsum; int idx; int pad; };

__global__ __launch_bounds__(1024) void k_vit_steps(const double* __restrict__ obs, int T, double skip, double stay,
                                                    double lskip, double lstay, double l25,
                                                    short* __restrict__ bp, double* __restrict__ fwd_out,
                                                    double* __restrict__ lik_final, int keep_fwd) {
    __shared__ double s_lik[2][NS], s_fwd[2][NS];
    __shared__ Fam s_fam[336];
    __shared__ double s_red[16];
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    s_lik[0][c] = 0.0;
    s_fwd[0][c] = 1.0 / NS;
    __syncthreads();
    double sp[4], lsp[4];
    sp[1] = 0.25; lsp[1] = l25;
    for (int j = 2; j <= 3; j++) { sp[j] = sp[j - 1] * 0.25 * skip; lsp[j] = lsp[j - 1] + l25 + lskip; }
    int cur = 0;
    for (int t = 0; t < T; t++) {
        const double* pl = s_lik[cur];
        const double* pf = s_fwd[cur];
        // ---- family pass: families 0..255 (j=1, 4 members), 256..319 (j=2, 16), 320..335 (j=3, 64)
        if (c < 336) {
            int j, g;
            if (c < 256) { j = 1; g = c; } else if (c < 320) { j = 2; g = c - 256; } else { j = 3; g = c - 320; }
            const int cnt = 1 << (2 * j), sh = 10 - 2 * j;
            double mx = -BIG * 10, prev = -BIG * 10, fs = 0.0;
            int idx = -1;
            for (int k = 0; k < cnt; k++) {
                const int q = g + (k << sh);
                const double x = pl[q];
                fs += sp[j] * pf[q];
                if (idx < 0 || x > mx) { mx = x; idx = q; }   // first strict maximum
            }
            for (int k = 0; k < cnt; k++) {                     // largest value ahead of it
                const int q = g + (k << sh);
                if (q == idx) break;
                const double x = pl[q];
                if (x > prev) prev = x;
            }
            s_fam[c].mx = mx; s_fam[c].prev = prev; s_fam[c].fsum = fs; s_fam[c].idx = idx;
        }
        __syncthreads();
        // ---- destination pass
        const double o = obs[(size_t)t * NS + c];
        double best = -BIG; int bq = -1; double fsum = 0.0;
#pragma unroll
        for (int j = 1; j <= 3; j++) {
            const int g = c >> (2 * j);
            const Fam& f = s_fam[(j == 1 ? 0 : j == 2 ? 256 : 320) + g];
            const double a = o + lsp[j];
            const double m = a + f.mx;
            fsum += f.fsum;
            if (f.prev > -BIG * 5 && a + f.prev == m) {
                // an earlier, smaller member rounds to the same sum: ordered scan (reference order)
                const int cnt = 1 << (2 * j), sh = 10 - 2 * j;
                for (int k = 0; k < cnt; k++) {
                    const int q = g + (k << sh);
                    const double l = a + pl[q];
                    if (l > best) { best = l; bq = q; }
                }
            } else if (m > best) { best = m; bq = f.idx; }
        }
        {
            const double l = o + lstay + pl[c];
            if (l > best) { best = l; bq = c; }
            fsum += stay * pf[c];
        }
        fsum *= exp(o);
        // ---- normalise forward probabilities (wave tree + 16-entry tree)
        double ssum = fsum;
        for (int off = 32; off; off >>= 1) ssum += __shfl_xor(ssum, off);
        if (lane == 0) s_red[wave] = ssum;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < 16; w++) tot += s_red[w];
        tot = 1.0 / tot;
        const double nf = fsum * tot;
        s_lik[cur ^ 1][c] = best;
        s_fwd[cur ^ 1][c] = nf;
        bp[(size_t)t * NS + c] = (short)bq;
        if (keep_fwd) fwd_out[(size_t)t * NS + c] = nf;
        __syncthreads();
        cur ^= 1;
    }
    lik_final[c] = s_lik[cur][c];
}

// nkeep stochastic back-traces; grid nkeep, block 1024.  path[k][i] for i = T-1 .. 0
__global__ __launch_bounds__(1024) void k_vit_trace(const double* __restrict__ fwd, const double* __restrict__ Tm, int T,
                                                    int start, const double* __restrict__ atten, const double* __restrict__ rnd,
                                                    short* __restrict__ path) {
    __shared__ double s_scan[16];
    __shared__ int s_pick;
    const int k = blockIdx.x, c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const double at = atten[k];
    int cur = start;
    for (int i = T - 1; i >= 0; i--) {
        if (c == 0) { path[(size_t)k * T + i] = (short)cur; s_pick = NS - 1; }
        // scores[i+1]->randbp(cur, ...): step index i+1 in the reference == stored step i here
        const double tv = Tm[(size_t)cur * NS + c];
        double p = tv == 0.0 ? 0.0 : tv * pow(fwd[(size_t)i * NS + c], at);
        // total
        double s = p;
        for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
        __syncthreads();
        if (lane == 0) s_scan[wave] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < 16; w++) tot += s_scan[w];
        tot = 1.0 / tot;
        p *= tot;
        // inclusive prefix sum over the 1024 states
        double x = p;
        for (int off = 1; off < 64; off <<= 1) { const double y = __shfl_up(x, off); if (lane >= off) x += y; }
        __syncthreads();
        if (lane == 63) s_scan[wave] = x;
        __syncthreads();
        double basev = 0.0;
        for (int w = 0; w < wave; w++) basev += s_scan[w];
        x += basev;
        const double r = rnd[(size_t)k * T + (T - 1 - i)];
        if (r < x) atomicMin(&s_pick, c);
        __syncthreads();
        cur = s_pick;
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------
static std::vector<double> build_T(double skip, double stay) {  // buildT, cpp/Viterbi.cpp:134-168
    std::vector<double> Tm((size_t)NS * NS, 0.0);
    for (int c = 0; c < NS; c++) {
        double sp = 0.25;
        for (int j = 1; j <= 4; j++) {
            for (int k = 0; k < (1 << (2 * j)); k++) Tm[(size_t)c * NS + ((c >> (2 * j)) + (k << (10 - 2 * j)))] += sp;
            sp = sp * 0.25 * skip;
        }
    }
    for (int i = 0; i < NS; i++) Tm[(size_t)i * (NS + 1)] = stay;
    return Tm;
}

int viterbi_device(Runtime* rt, int E, int T, const double* h_obsin, const double* d_model, int nkeep,
                   double skip, double stay, double mmin, double mmax, const double* h_rand,
                   std::vector<std::vector<int>>* paths) {
    paths->clear();
    if (T <= 0) return PS_OK;
    if (E > VMAXE) return fail(PS_ERR_UNSUPPORTED, "ViterbiMutate: more than 64 events");
    PS_TRY(rt->buf("vit_in").ensure((size_t)T * E * 4 * sizeof(double)));
    PS_TRY(rt->buf("vit_obs").ensure((size_t)T * NS * sizeof(double)));
    PS_TRY(rt->buf("vit_bp").ensure((size_t)T * NS * sizeof(short)));
    PS_TRY(rt->buf("vit_fwd").ensure((size_t)(nkeep ? T : 1) * NS * sizeof(double)));
    PS_TRY(rt->buf("vit_lik").ensure(NS * sizeof(double)));
    double* d_in = rt->buf("vit_in").as<double>();
    double* d_obs = rt->buf("vit_obs").as<double>();
    short* d_bp = rt->buf("vit_bp").as<short>();
    double* d_fwd = rt->buf("vit_fwd").as<double>();
    double* d_lik = rt->buf("vit_lik").as<double>();
    PS_HIP(hipMemcpyAsync(d_in, h_obsin, (size_t)T * E * 4 * sizeof(double), hipMemcpyHostToDevice, rt->stream));
    prof_begin(rt);
    hipLaunchKernelGGL(k_vit_obs, dim3(T), dim3(256), 0, rt->stream, d_in, d_model, E, std::log(2 * M_PI), d_obs);
    hipLaunchKernelGGL(k_vit_steps, dim3(1), dim3(1024), 0, rt->stream, d_obs, T, skip, stay, std::log(skip), std::log(stay),
                       std::log(0.25), d_bp, d_fwd, d_lik, nkeep ? 1 : 0);
    PS_HIP(hipGetLastError());
    std::vector<double> lik(NS);
    PS_HIP(hipMemcpyAsync(lik.data(), d_lik, NS * sizeof(double), hipMemcpyDeviceToHost, rt->stream));
    PS_HIP(hipStreamSynchronize(rt->stream));
    const int start = (int)(std::max_element(lik.begin(), lik.end()) - lik.begin());
    if (nkeep == 0) {
        std::vector<short> bp((size_t)T * NS);
        PS_HIP(hipMemcpyAsync(bp.data(), d_bp, bp.size() * sizeof(short), hipMemcpyDeviceToHost, rt->stream));
        PS_HIP(hipStreamSynchronize(rt->stream));
        prof_end(rt, "viterbi", (double)T * NS * (8.0 * E + 8 + 2));
        std::vector<int> p(T);
        int c = start;
        for (int i = T - 1; i >= 0; i--) { p[i] = c; c = bp[(size_t)i * NS + c]; }
        paths->push_back(p);
        return PS_OK;
    }
    // transition matrix for the back-steps, cached per (skip, stay)
    static double t_skip = -1, t_stay = -1;
    PS_TRY(rt->buf("vit_T").ensure((size_t)NS * NS * sizeof(double)));
    if (t_skip != skip || t_stay != stay) {
        std::vector<double> Tm = build_T(skip, stay);
        PS_HIP(hipMemcpyAsync(rt->buf("vit_T").p, Tm.data(), Tm.size() * sizeof(double), hipMemcpyHostToDevice, rt->stream));
        PS_HIP(hipStreamSynchronize(rt->stream));
        t_skip = skip; t_stay = stay;
    }
    std::vector<double> att(nkeep);
    for (int k = 0; k < nkeep; k++) att[k] = mmin + (mmax - mmin) * k / (double)nkeep;
    PS_TRY(rt->buf("vit_att").ensure(nkeep * sizeof(double)));
    PS_TRY(rt->buf("vit_rnd").ensure((size_t)nkeep * T * sizeof(double)));
    PS_TRY(rt->buf("vit_path").ensure((size_t)nkeep * T * sizeof(short)));
    PS_HIP(hipMemcpyAsync(rt->buf("vit_att").p, att.data(), nkeep * sizeof(double), hipMemcpyHostToDevice, rt->stream));
    PS_HIP(hipMemcpyAsync(rt->buf("vit_rnd").p, h_rand, (size_t)nkeep * T * sizeof(double), hipMemcpyHostToDevice, rt->stream));
    hipLaunchKernelGGL(k_vit_trace, dim3(nkeep), dim3(1024), 0, rt->stream, d_fwd, rt->buf("vit_T").as<double>(), T, start,
                       rt->buf("vit_att").as<double>(), rt->buf("vit_rnd").as<double>(), rt->buf("vit_path").as<short>());
    PS_HIP(hipGetLastError());
    prof_end(rt, "viterbi", (double)T * NS * (8.0 * E + 8 + 2 + 8 + 8.0 * nkeep));
    std::vector<short> hp((size_t)nkeep * T);
    PS_HIP(hipMemcpyAsync(hp.data(), rt->buf("vit_path").p, hp.size() * sizeof(short), hipMemcpyDeviceToHost, rt->stream));
    PS_HIP(hipStreamSynchronize(rt->stream));
    for (int k = 0; k < nkeep; k++) {
        std::vector<int> p(T);
        for (int i = 0; i < T; i++) p[i] = hp[(size_t)k * T + i];
        paths->push_back(p);
    }
    return PS_OK;
}

}  // namespace ps
