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
// Max-plus values (liks, back-pointers) are bit-exact; forward probabilities use device exp / log
// (fwd^atten as exp(atten * log fwd)) and tree sums, i.e. agree to a few ulp; they only weight the
// random back-traces.
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

// ordered combination of two family segments (A precedes B): first strict maximum, the largest
// value ahead of it, and the forward-probability partial sum
__device__ __forceinline__ Fam fam_join(const Fam& A, const Fam& B) {
    Fam r;
    if (B.mx > A.mx) { r.mx = B.mx; r.idx = B.idx; r.prev = A.mx > B.prev ? A.mx : B.prev; }
    else { r.mx = A.mx; r.idx = A.idx; r.prev = A.prev; }
    r.fsum = A.fsum + B.fsum;
    r.pad = 0;
    return r;
}

// families: 0..255 (j=1, 4 members), 256..319 (j=2, 16 members), 320..335 (j=3, 64 members);
// the j=3 families are scanned as 4 quarters (threads 320..383) and joined.
__global__ __launch_bounds__(1024) void k_vit_steps(const double* __restrict__ obs, int T, double skip, double stay,
                                                    double lskip, double lstay, double l25,
                                                    short* __restrict__ bp, double* __restrict__ lfwd_out,
                                                    double* __restrict__ lik_final, int keep_fwd) {
    __shared__ double s_lik[2][NS], s_fwd[2][NS];
    __shared__ Fam s_fam[336];
    __shared__ Fam s_q[64];
    __shared__ double s_red[16];
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    s_lik[0][c] = 0.0;
    s_fwd[0][c] = 1.0 / NS;
    __syncthreads();
    const double sp1 = 0.25, sp2 = sp1 * 0.25 * skip, sp3 = sp2 * 0.25 * skip;
    const double lsp1 = l25, lsp2 = lsp1 + l25 + lskip, lsp3 = lsp2 + l25 + lskip;
    // this thread's slice of the family pass
    int fj = 0, fg = 0, k0 = 0, kn = 0;
    if (c < 256) { fj = 1; fg = c; k0 = 0; kn = 4; }
    else if (c < 320) { fj = 2; fg = c - 256; k0 = 0; kn = 16; }
    else if (c < 384) { fj = 3; fg = (c - 320) & 15; k0 = ((c - 320) >> 4) * 16; kn = 16; }
    const int fsh = 10 - 2 * fj;
    const double fsp = fj == 1 ? sp1 : fj == 2 ? sp2 : sp3;
    int cur = 0;
    for (int t = 0; t < T; t++) {
        const double* pl = s_lik[cur];
        const double* pf = s_fwd[cur];
        const double o = obs[(size_t)t * NS + c];
        if (c < 384) {
            Fam f;
            f.mx = -BIG * 10; f.prev = -BIG * 10; f.fsum = 0.0; f.idx = -1; f.pad = 0;
            for (int k = k0; k < k0 + kn; k++) {
                const int q = fg + (k << fsh);
                const double x = pl[q];
                f.fsum += fsp * pf[q];
                if (f.idx < 0) { f.mx = x; f.idx = q; }
                else if (x > f.mx) { f.prev = f.mx; f.mx = x; f.idx = q; }   // all earlier values are <= the old maximum
            }
            if (c < 320) s_fam[c] = f; else s_q[c - 320] = f;
        }
        __syncthreads();
        if (c < 16) {
            Fam f = fam_join(fam_join(s_q[c], s_q[16 + c]), fam_join(s_q[32 + c], s_q[48 + c]));
            s_fam[320 + c] = f;
        }
        __syncthreads();
        // ---- destination pass
        double best = -BIG; int bq = -1; double fsum = 0.0;
#pragma unroll
        for (int j = 1; j <= 3; j++) {
            const int g = c >> (2 * j);
            const Fam& f = s_fam[(j == 1 ? 0 : j == 2 ? 256 : 320) + g];
            const double a = o + (j == 1 ? lsp1 : j == 2 ? lsp2 : lsp3);
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
        if (keep_fwd) lfwd_out[(size_t)t * NS + c] = log(nf);   // the back-traces need fwd^atten = exp(atten * log fwd)
        __syncthreads();
        cur ^= 1;
    }
    lik_final[c] = s_lik[cur][c];
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

// nkeep stochastic back-traces (randbp, cpp/Viterbi.cpp:105-131): one wave per path, lane l owns the
// 16 states 16l .. 16l+15 (so the running sum is in state order); the step's log-forward row is staged
// in LDS one step ahead.  grid nkeep, block 64.
__global__ __launch_bounds__(64) void k_vit_trace(const double* __restrict__ lfwd, int T, int start, double skip, double stay,
                                                  const double* __restrict__ atten, const double* __restrict__ rnd,
                                                  short* __restrict__ path) {
    const int k = blockIdx.x, l = threadIdx.x;
    const double at = atten[k];
    int cur = start;
    double lf[16], nx[16];
    {
        const double* row = lfwd + (size_t)(T - 1) * NS + 16 * l;
#pragma unroll
        for (int m = 0; m < 16; m++) lf[m] = row[m];
    }
    for (int i = T - 1; i >= 0; i--) {
        if (l == 0) path[(size_t)k * T + i] = (short)cur;
        if (i > 0) {
            const double* row = lfwd + (size_t)(i - 1) * NS + 16 * l;
#pragma unroll
            for (int m = 0; m < 16; m++) nx[m] = row[m];
        }
        double pr[16];
        double tot = 0.0;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const double tv = trans_weight(cur, 16 * l + m, skip, stay);
            const double x = tv == 0.0 ? 0.0 : tv * exp(at * lf[m]);
            pr[m] = x;
            tot += x;
        }
        for (int off = 32; off; off >>= 1) tot += __shfl_xor(tot, off);
        tot = 1.0 / tot;
        double run = 0.0;
#pragma unroll
        for (int m = 0; m < 16; m++) { pr[m] *= tot; run += pr[m]; }
        // exclusive prefix of the lane totals
        double x = run;
        for (int off = 1; off < 64; off <<= 1) { const double y = __shfl_up(x, off); if (l >= off) x += y; }
        double cs = x - run;
        const double r = rnd[(size_t)k * T + (T - 1 - i)];
        int pick = 0x7fffffff;
#pragma unroll
        for (int m = 0; m < 16; m++) { cs += pr[m]; if (pick == 0x7fffffff && r < cs) pick = 16 * l + m; }
        for (int off = 32; off; off >>= 1) pick = min(pick, __shfl_xor(pick, off));
        cur = pick == 0x7fffffff ? NS - 1 : pick;
#pragma unroll
        for (int m = 0; m < 16; m++) lf[m] = nx[m];
    }
}

// -------------------------------------------------------------------------------------------------
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
    std::vector<double> att(nkeep);
    for (int k = 0; k < nkeep; k++) att[k] = mmin + (mmax - mmin) * k / (double)nkeep;
    PS_TRY(rt->buf("vit_att").ensure(nkeep * sizeof(double)));
    PS_TRY(rt->buf("vit_rnd").ensure((size_t)nkeep * T * sizeof(double)));
    PS_TRY(rt->buf("vit_path").ensure((size_t)nkeep * T * sizeof(short)));
    PS_HIP(hipMemcpyAsync(rt->buf("vit_att").p, att.data(), nkeep * sizeof(double), hipMemcpyHostToDevice, rt->stream));
    PS_HIP(hipMemcpyAsync(rt->buf("vit_rnd").p, h_rand, (size_t)nkeep * T * sizeof(double), hipMemcpyHostToDevice, rt->stream));
    hipLaunchKernelGGL(k_vit_trace, dim3(nkeep), dim3(64), 0, rt->stream, d_fwd, T, start, skip, stay,
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
