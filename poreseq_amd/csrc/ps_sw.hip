// ps_sw.hip — full-matrix Smith-Waterman (swfull, cpp/swlib.cpp:211-340) on gfx950.
//
// Integer DP, +5 / -4 / -8, bit-exact with the reference including its tie rules
// (left, then up with strict >, then diagonal with >=; first strict maximum in column-major
// order starts the traceback; traceback stops at the first score <= 0).
//
// Tiling: 64 rows (one per lane) x TC columns per wave.  Lanes run systolically (lane l is l
// columns behind lane 0); the value above comes from the neighbour lane by DPP, the tile's top
// boundary row and left boundary column come from small global arrays written by the tiles
// above / to the left, which finished on the previous tile anti-diagonal (one launch per tile
// anti-diagonal, all sequence pairs of a batch in the same launch).
// Step codes are stored per tile in the order they are produced ([t][lane], 64-byte coalesced
// stores); the traceback pulls one whole tile into LDS and walks it there.
#include "ps_sw.h"

namespace ps {

constexpr int TC = 64;
constexpr int TSTEPS = ((TC + 63 + 3) / 4) * 4;   // steps per tile, padded to whole 4-step store groups

__device__ __forceinline__ int shr1_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

__global__ __launch_bounds__(64) void k_sw_tiles(const SwPair* pairs, const char* chars, unsigned char* steps,
                                                 int* hrow, int* hcol, int4* tiles, int d) {
    const SwPair p = pairs[blockIdx.y];   // by value: through a reference hipcc re-reads n2 from memory on every step
    const int rmin = max(0, d - (p.ntc - 1)), rmax = min(p.ntr - 1, d);
    const int r = rmin + blockIdx.x;
    if (r > rmax) return;
    const int c = d - r, l = threadIdx.x;
    const int i0 = r * 64, j0 = c * TC;
    const int i = i0 + 1 + l;
    const bool rowok = i <= p.n1;
    const char* s1 = chars + p.s1_off;
    const char* s2 = chars + p.s2_off;
    int* hr_prev = hrow + p.hrow_off + (int64_t)((r + 2) % 3) * (p.n2 + 1);
    int* hr_cur = hrow + p.hrow_off + (int64_t)(r % 3) * (p.n2 + 1);
    int* hc = hcol + p.hcol_off;
    __shared__ int s_top[TC + 1];
    __shared__ int s_bot[TC];
    __shared__ char s_c2[TC];
    for (int k = l; k <= TC; k += 64) { const int j = j0 + k; s_top[k] = (r > 0 && j <= p.n2) ? hr_prev[j] : 0; }
    for (int k = l; k < TC; k += 64) { const int j = j0 + 1 + k; s_c2[k] = j <= p.n2 ? s2[j - 1] : 0; s_bot[k] = 0; }
    __syncthreads();
    const char c1 = rowok ? s1[i - 1] : 1;
    int left = (c > 0 && rowok) ? hc[i] : 0;       // H(i, j0)
    int h = left;                                  // running H(i, j) of this lane
    int prevup = shr1_i(left);                     // H(i-1, j0)
    if (l == 0) prevup = s_top[0];
    int best = 0, bestt = 0;
    // columns this lane really has: jj in [0, ncl); it works on them at steps t = l + jj
    const int ncl = rowok ? min(max(p.n2 - j0, 0), TC) : 0;
    // step codes: 4 consecutive steps of a lane are packed into one 32-bit store, [t/4][lane][t%4]
    unsigned* st = (unsigned*)(steps + p.steps_off + (int64_t)(r * p.ntc + c) * TSTEPS * 64);
    for (int t0 = 0; t0 < TSTEPS; t0 += 4) {
        // LDS operands of the next four steps, issued together (one lgkmcnt wait per four steps)
        int tops[4];
        char ch[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + u;
            tops[u] = s_top[min(t + 1, TC)];
            ch[u] = s_c2[min(max(t - l, 0), TC - 1)];
        }
        unsigned packed = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + u;
            const bool act = (unsigned)(t - l) < (unsigned)ncl;
            int up = shr1_i(h);
            if (l == 0) up = tops[u];
            const int diag = prevup;
            prevup = up;
            // reference order (cpp/swlib.cpp:243-263): left with >, up with >, diagonal with >=, floor 0
            const bool eq = c1 == ch[u];
            const int sd = diag + (eq ? 5 : -4);
            const int sl = h - 8, su = up - 8;
            const int l0 = max(sl, 0);
            const int m = max(l0, su);
            const int score = max(m, sd);
            const unsigned step = sd >= m ? 3u : (su > l0 ? 2u : (sl > 0 ? 1u : 0u));
            const unsigned code = step | (score > 0 ? 4u : 0u) | (eq ? 8u : 0u);
            packed |= act ? code << (8 * u) : 0u;
            if (act && score > best) { best = score; bestt = t; }
            h = act ? score : h;
            if (l == 63 && t >= 63) s_bot[min(t - 63, TC - 1)] = h;
        }
        st[(t0 >> 2) * 64 + l] = packed;
    }
    const int bestj = j0 + 1 + (bestt - l);
    if (rowok) hc[i] = h;
    // tile maximum: largest score, then smallest column, then smallest row (column-major first hit)
    int bi = i, bj = best > 0 ? bestj : 0x7fffffff, bs = best;
    for (int off = 32; off; off >>= 1) {
        const int os = __shfl_xor(bs, off), oj = __shfl_xor(bj, off), oi = __shfl_xor(bi, off);
        if (os > bs || (os == bs && (oj < bj || (oj == bj && oi < bi)))) { bs = os; bj = oj; bi = oi; }
    }
    if (l == 0) tiles[p.tile_off + r * p.ntc + c] = make_int4(bs, bi, bj, 0);
    __syncthreads();
    if (i0 + 64 <= p.n1)
        for (int k = l; k < TC; k += 64) { const int j = j0 + 1 + k; if (j <= p.n2) hr_cur[j] = s_bot[k]; }
    if (l == 0 && c == 0) hr_cur[0] = 0;
}

__global__ __launch_bounds__(64) void k_sw_best(const SwPair* pairs, const int4* tiles, int* res) {
    const SwPair p = pairs[blockIdx.x];
    const int l = threadIdx.x, nt = p.ntr * p.ntc;
    int bs = 0, bi = 0, bj = 0x7fffffff;
    for (int k = l; k < nt; k += 64) {
        const int4 v = tiles[p.tile_off + k];
        if (v.x > bs || (v.x == bs && v.x > 0 && (v.z < bj || (v.z == bj && v.y < bi)))) { bs = v.x; bi = v.y; bj = v.z; }
    }
    for (int off = 32; off; off >>= 1) {
        const int os = __shfl_xor(bs, off), oj = __shfl_xor(bj, off), oi = __shfl_xor(bi, off);
        if (os > bs || (os == bs && os > 0 && (oj < bj || (oj == bj && oi < bi)))) { bs = os; bj = oj; bi = oi; }
    }
    if (l == 0) {
        int* o = res + p.res_off;
        o[0] = bs; o[1] = bs > 0 ? bi : 0; o[2] = bs > 0 ? bj : 0;
    }
}

__global__ __launch_bounds__(64) void k_sw_trace(const SwPair* pairs, const unsigned char* steps, int* out, int* res) {
    const SwPair p = pairs[blockIdx.x];
    const int l = threadIdx.x;
    __shared__ unsigned char s_t[TSTEPS * 64];
    int* o = res + p.res_off;
    int i = o[1], j = o[2];
    int np = 0, nm = 0;
    int* oi = out + p.out_off;
    int* oj = oi + (p.n1 + p.n2 + 2);
    bool done = !(i > 0 && j > 0);
    while (!done) {
        const int r = (i - 1) / 64, c = (j - 1) / TC;
        const int i0 = r * 64, j0 = c * TC;
        const uint4* src = (const uint4*)(steps + p.steps_off + (int64_t)(r * p.ntc + c) * TSTEPS * 64);
        uint4* dst = (uint4*)s_t;
        for (int k = l; k < TSTEPS * 4; k += 64) dst[k] = src[k];
        __syncthreads();
        if (l == 0) {
            auto code_at = [&](int ll, int jj) -> unsigned {   // clamped: callers check the real bounds
                ll = max(ll, 0); jj = max(jj, 0);
                const int tt = jj + ll;
                return s_t[((tt >> 2) * 64 + ll) * 4 + (tt & 3)];
            };
            int ll = i - i0 - 1, jj = j - j0 - 1;
            unsigned code = code_at(ll, jj);
            while (true) {
                if (!(i > 0 && j > 0)) { done = true; break; }
                if (ll < 0 || jj < 0) break;  // left this tile
                // the three possible successors, fetched while this cell is decoded
                const unsigned cL = code_at(ll, jj - 1), cD = code_at(ll - 1, jj - 1), cU = code_at(ll - 1, jj);
                if (!(code & 4)) { done = true; break; }   // score <= 0
                const unsigned stp = code & 3;
                if (stp == 1) { oi[np] = 0; oj[np] = j; np++; j--; jj--; code = cL; }
                else if (stp == 2) { oi[np] = i; oj[np] = 0; np++; i--; ll--; code = cU; }
                else if (stp == 3) { oi[np] = i; oj[np] = j; np++; if (code & 8) nm++; i--; j--; ll--; jj--; code = cD; }
                else { done = true; break; }
            }
        }
        i = __shfl(i, 0); j = __shfl(j, 0); np = __shfl(np, 0); nm = __shfl(nm, 0);
        done = __shfl((int)done, 0) != 0;
        __syncthreads();
    }
    if (l == 0) { o[3] = np; o[4] = nm; }
}

// -------------------------------------------------------------------------------------------------
// enqueue a batch of pairwise alignments on the runtime's second stream (asynchronous)
int sw_launch(Runtime* rt, const std::vector<std::pair<const std::string*, const std::string*>>& in, SwJob* job) {
    const int np = (int)in.size();
    job->np = np;
    if (!np) return PS_OK;
    std::vector<SwPair>& pairs = job->pairs;
    std::string& pool = job->pool;
    pairs.assign(np, SwPair());
    int64_t steps_tot = 0, hrow_tot = 0, hcol_tot = 0, tile_tot = 0, out_tot = 0;
    int maxdiag = 0, maxtiles = 0;
    for (int k = 0; k < np; k++) {
        SwPair& p = pairs[k];
        p.n1 = (int)in[k].first->size(); p.n2 = (int)in[k].second->size();
        p.ntr = std::max(1, (p.n1 + 63) / 64); p.ntc = std::max(1, (p.n2 + TC - 1) / TC);
        p.s1_off = (int64_t)pool.size(); pool += *in[k].first;
        p.s2_off = (int64_t)pool.size(); pool += *in[k].second;
        p.steps_off = steps_tot; steps_tot += (int64_t)p.ntr * p.ntc * TSTEPS * 64;
        p.hrow_off = hrow_tot; hrow_tot += 3 * ((int64_t)p.n2 + 1);
        p.hcol_off = hcol_tot; hcol_tot += (int64_t)p.n1 + 1;
        p.tile_off = tile_tot; tile_tot += (int64_t)p.ntr * p.ntc;
        p.out_off = out_tot; out_tot += 2 * ((int64_t)p.n1 + p.n2 + 2);
        p.res_off = (int64_t)k * 8;
        maxdiag = std::max(maxdiag, p.ntr + p.ntc - 1);
        maxtiles = std::max(maxtiles, std::min(p.ntr, p.ntc));
        job->cells += (double)p.n1 * p.n2;
    }
    pool.push_back(0);
    job->out_tot = out_tot;
    PS_TRY(rt->buf("sw_pairs").ensure(np * sizeof(SwPair)));
    PS_TRY(rt->buf("sw_chars").ensure(pool.size()));
    PS_TRY(rt->buf("sw_steps").ensure(steps_tot));
    PS_TRY(rt->buf("sw_hrow").ensure(hrow_tot * sizeof(int)));
    PS_TRY(rt->buf("sw_hcol").ensure(hcol_tot * sizeof(int)));
    PS_TRY(rt->buf("sw_tiles").ensure(tile_tot * sizeof(int4)));
    PS_TRY(rt->buf("sw_out").ensure(out_tot * sizeof(int)));
    PS_TRY(rt->buf("sw_res").ensure((size_t)np * 8 * sizeof(int)));
    SwPair* d_pairs = rt->buf("sw_pairs").as<SwPair>();
    char* d_chars = rt->buf("sw_chars").as<char>();
    unsigned char* d_steps = rt->buf("sw_steps").as<unsigned char>();
    int* d_hrow = rt->buf("sw_hrow").as<int>();
    int* d_hcol = rt->buf("sw_hcol").as<int>();
    int4* d_tiles = rt->buf("sw_tiles").as<int4>();
    int* d_out = rt->buf("sw_out").as<int>();
    int* d_res = rt->buf("sw_res").as<int>();
    hipStream_t st = rt->stream2;
    PS_TRY(rt->up(d_pairs, pairs.data(), np * sizeof(SwPair), st));
    PS_TRY(rt->up(d_chars, pool.data(), pool.size(), st));
    PS_HIP(hipMemsetAsync(d_res, 0, (size_t)np * 8 * sizeof(int), st));
    if (rt->prof_on) PS_HIP(hipEventRecord(rt->sw0, st));
    for (int d = 0; d < maxdiag; d++)
        hipLaunchKernelGGL(k_sw_tiles, dim3(maxtiles, np), dim3(64), 0, st, d_pairs, d_chars, d_steps, d_hrow, d_hcol, d_tiles, d);
    PS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sw_best, dim3(np), dim3(64), 0, st, d_pairs, d_tiles, d_res);
    hipLaunchKernelGGL(k_sw_trace, dim3(np), dim3(64), 0, st, d_pairs, d_steps, d_out, d_res);
    PS_HIP(hipGetLastError());
    if (rt->prof_on) PS_HIP(hipEventRecord(rt->sw1, st));
    PS_TRY(rt->hbuf("sw_res").ensure((size_t)np * 8 * sizeof(int)));
    PS_TRY(rt->hbuf("sw_out").ensure((size_t)out_tot * sizeof(int)));
    job->res = rt->hbuf("sw_res").as<int>();
    job->outbuf = rt->hbuf("sw_out").as<int>();
    PS_HIP(hipMemcpyAsync(job->res, d_res, (size_t)np * 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    PS_HIP(hipMemcpyAsync(job->outbuf, d_out, (size_t)out_tot * sizeof(int), hipMemcpyDeviceToHost, st));
    return PS_OK;
}

int sw_finish(Runtime* rt, SwJob* job, std::vector<SwResult>* out) {
    const int np = job->np;
    out->assign(np, SwResult());
    if (!np) return PS_OK;
    PS_HIP(hipStreamSynchronize(rt->stream2));
    if (rt->prof_on) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, rt->sw0, rt->sw1) == hipSuccess) {
            Prof& pr = rt->prof["sw"];
            pr.ms += ms; pr.launches += 1; pr.bytes += job->cells * 5.0;  // 4-byte score + 1-byte step per cell (the reference's footprint)
        }
    }
    for (int k = 0; k < np; k++) {
        const int n = job->res[k * 8 + 3], nm = job->res[k * 8 + 4];
        const SwPair& p = job->pairs[k];
        SwResult& r = (*out)[k];
        r.score = job->res[k * 8 + 0];
        const int* oi = job->outbuf + p.out_off;
        const int* oj = oi + (p.n1 + p.n2 + 2);
        r.a.assign(oi, oi + n); r.b.assign(oj, oj + n);
        std::reverse(r.a.begin(), r.a.end()); std::reverse(r.b.begin(), r.b.end());
        r.accuracy = 100.0 * nm / (double)n;  // NaN for an empty alignment, as the reference computes it
    }
    return PS_OK;
}

int sw_batch(Runtime* rt, const std::vector<std::pair<const std::string*, const std::string*>>& in, std::vector<SwResult>* out) {
    SwJob job;
    PS_TRY(sw_launch(rt, in, &job));
    return sw_finish(rt, &job, out);
}

int sw_device(Runtime* rt, const std::string& s1, const std::string& s2, int* score, double* accuracy,
              std::vector<int>* inds1, std::vector<int>* inds2) {
    std::vector<std::pair<const std::string*, const std::string*>> in(1, {&s1, &s2});
    std::vector<SwResult> out;
    PS_TRY(sw_batch(rt, in, &out));
    *score = out[0].score; *accuracy = out[0].accuracy;
    inds1->swap(out[0].a); inds2->swap(out[0].b);
    return PS_OK;
}

}  // namespace ps
