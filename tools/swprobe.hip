// swprobe: standalone timing harness for the Smith-Waterman fill kernel (tuning aid; kernel text is pasted from ps_sw.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>
struct SwPair { int n1, n2, nrb, ngw, pitch, pad; int64_t s1_off, s2_off, row_off, col_off, blk_off, out_off, res_off; };
#ifndef VARIANT
#define VARIANT 0
#endif
constexpr int SWB = 8;    // rows per pipeline step
constexpr int SWW = 16;   // waves per workgroup
constexpr bool getenv_nocs = false;

template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_max(int v) {   // lanes without a source keep their own value
    return max(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false));
}
// inclusive prefix maximum over the 64 lanes of a wave
__device__ __forceinline__ int wave_scan_max(int v) {
    v = dpp_max<0x111, 0xf>(v);   // row_shr:1
    v = dpp_max<0x112, 0xf>(v);   // row_shr:2
    v = dpp_max<0x114, 0xf>(v);   // row_shr:4
    v = dpp_max<0x118, 0xf>(v);   // row_shr:8
    v = dpp_max<0x142, 0xa>(v);   // row_bcast:15 -> rows 1 and 3
    v = dpp_max<0x143, 0xc>(v);   // row_bcast:31 -> rows 2 and 3
    return v;
}

// One row of one wave's strip, in shifted form: the lane's k-th column is carried as G[k] = H + 8k, which turns
// the in-lane part of the prefix maximum into a plain running max and folds the gap steps into constants
// (7 integer instructions per cell).  G[] holds row i-1 on entry and row i on return.
//   c1     character of row i (wave-uniform)          bl     H(i, first column - 1)   (left boundary, uniform)
//   bprev  H(i-1, first column - 1) (uniform)         lane   lane index, lane0 = lane ? -2^29 : 0
//   c2[] and c1 hold characters shifted left by 4
// MODE 0: values only.  MODE 1: also track the column-major first cell equal to `target` in (fc, fr).
// MODE 2 (K == 1): also return the cell's step code  step | 4*(score > 0) | 8*(characters equal).
template <int K, int MODE>
__device__ __forceinline__ unsigned sw_row(int (&G)[K], const int (&c2)[K], const int c1, const int bl, const int bprev,
                                           const int lane, const int lane0, const int target, const int i, const int jfirst, int& fc, int& fr) {
    // H(i-1, j-1) of this lane's first column: the previous row's last column of the lane to the left (+ 8(K-1))
    const int d0 = __builtin_amdgcn_update_dpp(bprev + 8 * (K - 1), G[K - 1], 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    int y[K];
    int sd0 = 0, up0 = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        // substitution penalty without a compare / select (VALU -> SGPR -> v_cndmask round trips are slow on
        // gfx950): characters are held shifted left by 4, so c2 ^ c1 is 0 when equal and >= 16 otherwise
        const int pen = min(c2[k] ^ c1, 9);
        // diagonal + substitution score, and value above - 8, both shifted by 8k
        const int sd = (k ? G[k - 1] + 13 : d0 + (5 - 8 * (K - 1))) - pen;
        const int up = G[k] - 8;
        if (k == 0) { sd0 = sd; up0 = up; }
        y[k] = max(max(sd, up), 8 * k);
    }
    // running maximum over the lane's columns as a two-level tree (groups of 4): depth 6 instead of K - 1
    constexpr int NG = (K + 3) / 4;
    int tg[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
#pragma unroll
        for (int k = 4 * g + 1; k < min(4 * g + 4, K); k++) y[k] = max(y[k], y[k - 1]);
        const int last = y[min(4 * g + 3, K - 1)];
        tg[g] = g ? max(last, tg[g - 1]) : last;
    }
    // scan of the lanes' last columns (as H + 8K*lane); the left boundary enters through lane 0
    int z = tg[NG - 1] + (8 * K * lane - 8 * (K - 1));
    z = max(z, (bl - 8 * K) + lane0);   // lane0 = 0 in lane 0, -2^29 elsewhere
    if (VARIANT != 3) z = wave_scan_max(z);
    // H(i, first column - 1) of this lane, minus one gap
    const int hl8 = __builtin_amdgcn_update_dpp(bl - 8 * K, z, 0x138, 0xf, 0xf, false) + (8 * K - 8 * K * lane - 8);
    unsigned code = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int gn = k >= 4 ? max(max(y[k], tg[k / 4 - 1]), hl8) : max(y[k], hl8);
        if (MODE == 2 && k == 0) {
            // reference order (cpp/swlib.cpp:243-263): left with >, up with >, diagonal with >=
            const int l0 = max(hl8, 0), m = max(l0, up0);
            const unsigned step = sd0 >= m ? 3u : (up0 > l0 ? 2u : (hl8 > 0 ? 1u : 0u));
            code = step | (gn > 0 ? 4u : 0u) | (c2[0] == c1 ? 8u : 0u);
        }
        if (MODE == 1) {
            const int col = jfirst + k + 1;
            if (gn == target + 8 * k && col < fc) { fc = col; fr = i; }
        }
        G[k] = gn;
    }
    return code;
}

// ---- fill: grid (pairs), block 1024; one launch per super-strip ------------------------------------------------
template <int K>
__global__ __launch_bounds__(1024) void k_sw_fill(const SwPair* pairs, const char* chars, int* rowsave, int* colsave,
                                                  int* blkmax, int ss) {
    const SwPair p = pairs[blockIdx.x];
    if (p.n1 <= 0 || p.n2 <= 0 || ss * SWW * 64 * K >= p.n2) return;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int gw = ss * SWW + w;
    const int wfirst = gw * 64 * K;            // 0-based first column of the wave
    const bool wave_on = wfirst < p.n2;
    const int jbase = wfirst + l * K;
    const char* s1 = chars + p.s1_off;
    const char* s2 = chars + p.s2_off;
    int c2[K], G[K], bmk[K];
#pragma unroll
    for (int k = 0; k < K; k++) { c2[k] = jbase + k < p.n2 ? (int)(unsigned char)s2[jbase + k] << 4 : 0; G[k] = 8 * k; bmk[k] = 0; }
    const int lane0 = l ? -(1 << 29) : 0;
    __shared__ int hand[SWW][2][SWB];
    // this lane's last column is column jbase + K (1-based); every 64th column is kept as a tile boundary
    const bool keeps = ((jbase + K) & 63) == 0 && jbase + K <= p.n2;
    int* csave = colsave + p.col_off + (int64_t)((jbase + K) >> 6) * (p.n1 + 1);
    const int* cprev = colsave + p.col_off + (int64_t)(wfirst >> 6) * (p.n1 + 1);   // H(*, wfirst): used by wave 0 when ss > 0
    if (wave_on && keeps) csave[0] = 0;
    int bprev = 0, dummy_c = 0, dummy_r = 0;
    const int nchunks = (p.n1 + SWB - 1) / SWB;
    // row characters (and, for the first wave of a later super-strip, the left boundary) are fetched one chunk
    // ahead; the workgroup barrier waits for LDS only, so global loads and stores stay in flight across it
    auto fetch = [&](int c, int& ch, int& bd) {
        ch = 1; bd = 0;
        const int i0 = c * SWB;
        if (wave_on && c >= 0 && c < nchunks && l < SWB && i0 + l < p.n1) {
            ch = (int)(unsigned char)s1[i0 + l] << 4;
            if (w == 0 && gw > 0) bd = cprev[i0 + 1 + l];
        }
    };
    int ch_nx, bd_nx;
    fetch(0 - w, ch_nx, bd_nx);
    for (int s = 0; s < nchunks + SWW - 1; s++) {
        const int c = s - w;
        const int ch1 = ch_nx, bd1 = bd_nx;
        fetch(c + 1, ch_nx, bd_nx);
        if (wave_on && c >= 0 && c < nchunks) {
            const int i0 = c * SWB;
            int bnd = bd1;
            if (w > 0 && l < SWB && i0 + l < p.n1) bnd = hand[w - 1][(s - 1) & 1][l];
#pragma unroll
            for (int r = 0; r < SWB; r++) {
                if (i0 + r < p.n1) {
                    const int bl = __builtin_amdgcn_readlane(bnd, r), c1 = __builtin_amdgcn_readlane(ch1, r);
                    sw_row<K, 0>(G, c2, c1, bl, bprev, l, lane0, 0, 0, 0, dummy_c, dummy_r);
                    bprev = bl;
#pragma unroll
                    for (int k = 0; k < K; k++) if (VARIANT != 6) bmk[k] = max(bmk[k], G[k]);
                    const int hlast = G[K - 1] - 8 * (K - 1);
                    if (VARIANT != 2 && VARIANT != 5 && l == 63) hand[w][s & 1][r] = hlast;
                    if (VARIANT != 1 && VARIANT != 5 && keeps) csave[i0 + r + 1] = hlast;
                }
            }
            const int iend = min(i0 + SWB, p.n1);
            if ((iend & 63) == 0 || iend == p.n1) {   // row block q complete
                const int q = (iend - 1) >> 6;
                int bm = 0;
#pragma unroll
                for (int k = 0; k < K; k++) { bm = max(bm, bmk[k] - 8 * k); bmk[k] = 0; }
                for (int o = 32; o; o >>= 1) bm = max(bm, __shfl_xor(bm, o));
                if (l == 0) blkmax[p.blk_off + (int64_t)q * p.ngw + gw] = bm;
                if (iend < p.n1) {                    // row 64(q+1) is the top boundary of block q+1
                    int* rs = rowsave + p.row_off + (int64_t)(q + 1) * p.pitch + jbase;
#pragma unroll
                    for (int k = 0; k < K; k++) if (jbase + k < p.n2) rs[k] = G[k] - 8 * k;
                }
            }
        }
        if (VARIANT != 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}


template <int K> void run(int L) {
    std::string a(L, 'A'), b(L, 'A');
    srand(1);
    for (int i = 0; i < L; i++) { a[i] = "ACGT"[rand() & 3]; b[i] = (rand() % 10) ? a[i] : "ACGT"[rand() & 3]; }
    std::string pool = a + b;
    SwPair p{}; p.n1 = L; p.n2 = L; p.nrb = (L + 63) / 64; const int ssw = 16 * 64 * K; p.ngw = 16 * ((L + ssw - 1) / ssw);
    p.pitch = ((L + 3) / 4) * 4 + 4; p.s1_off = 0; p.s2_off = L;
    char* dch; SwPair* dp; int *drow, *dcol, *dblk;
    hipMalloc(&dch, pool.size()); hipMemcpy(dch, pool.data(), pool.size(), hipMemcpyHostToDevice);
    hipMalloc(&dp, sizeof(p)); hipMemcpy(dp, &p, sizeof(p), hipMemcpyHostToDevice);
    hipMalloc(&drow, (size_t)p.nrb * p.pitch * 4); hipMalloc(&dcol, ((size_t)L / 64 + 1) * (L + 1) * 4); hipMalloc(&dblk, (size_t)p.nrb * p.ngw * 4);
    hipMemset(dblk, 0, (size_t)p.nrb * p.ngw * 4);
    const int nss = (L + ssw - 1) / ssw;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int ss = 0; ss < nss; ss++) hipLaunchKernelGGL(k_sw_fill<K>, dim3(1), dim3(1024), 0, 0, dp, dch, drow, dcol, dblk, ss);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            std::vector<int> blk((size_t)p.nrb * p.ngw); hipMemcpy(blk.data(), dblk, blk.size() * 4, hipMemcpyDeviceToHost);
            int best = 0; for (int v : blk) best = best > v ? best : v;
            printf("variant %d K=%d L=%d: %.3f ms  %.0f ns/row  (best %d)\n", VARIANT, K, L, ms, ms * 1e6 / L, best);
        }
    }
}
int main(int argc, char** argv) {
    run<4>(1000); run<4>(4000); run<8>(8000); run<16>(10000);
    return 0;
}
