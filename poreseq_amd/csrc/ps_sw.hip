// ps_sw.hip — full-matrix Smith-Waterman (swfull, cpp/swlib.cpp:211-340) on gfx950.
//
// Integer DP, +5 / -4 / -8 (linear gaps), bit-exact with the reference including its tie rules (left with >,
// then up with >, then diagonal with >=, floor 0; the first strict maximum in column-major order starts the
// traceback; the traceback stops at the first score <= 0).
//
// Formulation.  With a linear gap cost the row recurrence  H(i,j) = max(c(i,j), H(i,j-1) - 8),
// c(i,j) = max(0, H(i-1,j-1) + s, H(i-1,j) - 8)  is a prefix maximum:  H(i,j) + 8j = max_{j'<=j} (c(i,j') + 8j').
// So a whole row is computed at once — every c from the previous row, then one max-scan across the lanes (DPP) —
// with no systolic skew: all 64 lanes work on every row.
//
// Layout.  Lane l of wave w owns K consecutive columns (K = 4 or 8 chosen from the longest sequence); 8 waves x
// 64 lanes x K columns = one "super-strip" = one 512-thread workgroup, and a pair takes as many super-strips as
// its columns need, all in the same launch and pipelined over rows like the waves inside one (a 10 kb pair runs
// on three CUs).  The waves form a pipeline over rows: in
// pipeline step s wave w does rows 8(s-w)+1 .. 8(s-w)+8 and hands the H values of its last column to wave
// w+1 through LDS (one workgroup barrier per 8 rows).  Nothing per cell goes to memory: the fill keeps only
//   rowsave  H(64q, *)    every 64th row       (top boundaries of 64-row blocks)
//   colsave  H(*, 64c)    every 64th column    (left boundaries of 64-column blocks)
//   blkmax   max H per (row block, wave strip)
// and the traceback kernel recomputes the 64 x 64 tiles its path crosses (the same row routine with one column
// per lane, this time deriving the 4-bit step codes into LDS) — about 300 of the 25 000 tiles of a 10 kb pair.
// The starting cell is located by recomputing only the strips whose block maximum equals the global maximum.
#include <cstring>

#include "ps_sw.h"

namespace ps {

constexpr int SWB = 8;    // rows per pipeline step
constexpr int SWW = 8;    // waves per workgroup of the chained form: two per SIMD keeps one pair's strips issue-balanced over several CUs
constexpr int SWW1 = 16;  // waves per workgroup of the one-workgroup-per-pair form (below)

// inclusive prefix maximum over the 64 lanes of a wave: six v_max_i32 with a DPP-shifted first operand; a lane without a source
// (the first lanes of a row for row_shr, the rows a row_mask leaves out) is disabled for that instruction and keeps its value.
// Written out: through __builtin_amdgcn_update_dpp the compiler emits v_mov + v_mov_dpp + v_max per step (18 instead of 6 vector
// instructions per row: a sixth of the fill's).  The two wait states a DPP read needs after the write of its source are the s_nop 1.
__device__ __forceinline__ int wave_scan_max(int v) {
    asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v));
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
    z = wave_scan_max(z);
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

// ---- fill: grid (super-strips, pairs), block 64 * SWW ------------------------------------------------------------------
// The workgroups of one pair's super-strips run concurrently, as one more level of the row pipeline: the last wave
// of strip ss stores its boundary column (colsave) and, every 64 rows, publishes the row count with an agent-scope
// release; the first wave of strip ss+1 acquires it before it reads those rows.  A workgroup takes its strip index from a
// per-pair ticket (not from blockIdx, whose dispatch order HIP does not promise): whoever waits for strip ss-1 knows that a
// workgroup holding that ticket started before it, so the producer is always resident or finished.
//
// One workgroup per pair (WW = 16 waves, K = 8 / 16 columns per lane: up to 8 192 / 16 384 columns): the whole pair is
// ONE super-strip, its sixteen waves pipeline over rows through LDS and nobody waits on another workgroup — no boundary column in
// global memory, no tickets, no spinning waves holding slots and registers while their producer is scheduled.  Built for the round-3
// review's item 5, parity-tested, and slower than the chained strips where it was meant to win (see sw_launch): selected only by
// PORESEQ_SW_FORM=one.
constexpr int SW_SPIN_LIMIT = 1 << 22;
template <int K, int WW>
__global__ __launch_bounds__(64 * WW) void k_sw_fill(const SwPair* pairs, const char* chars, int* rowsave, int* colsave,
                                                  int* blkmax, int* prog, int* ticket, int* res) {
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ticket[blockIdx.y], 1);
    __syncthreads();
    const int ss = s_ticket;
    const SwPair p = pairs[blockIdx.y];
    if (p.n1 <= 0 || p.n2 <= 0 || ss * WW * 64 * K >= p.n2) return;
    int* prog_my = prog + (int64_t)blockIdx.y * gridDim.x + ss;
    const bool has_next = (ss + 1) * WW * 64 * K < p.n2;   // strip ss+1 exists: it consumes this strip's last column
    int seen = 0;                                           // rows of strip ss-1 known to be complete
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int gw = ss * WW + w;
    const int wfirst = gw * 64 * K;            // 0-based first column of the wave
    const bool wave_on = wfirst < p.n2;
    const int jbase = wfirst + l * K;
    const char* s1 = chars + p.s1_off;
    const char* s2 = chars + p.s2_off;
    int c2[K], G[K], bmk[K];
#pragma unroll
    for (int k = 0; k < K; k++) { c2[k] = jbase + k < p.n2 ? (int)(unsigned char)s2[jbase + k] << 4 : 0; G[k] = 8 * k; bmk[k] = 0; }
    const int lane0 = l ? -(1 << 29) : 0;
    __shared__ int hand[WW][2][SWB];
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
        if (w == 0 && ss > 0 && c >= 0 && c < nchunks) {   // wave-uniform: the rows of this chunk must have been published
            const int need = min(p.n1, (i0 + SWB + 63) & ~63);
            int spins = 0;
            for (; seen < need && spins < SW_SPIN_LIMIT; spins++) {
                seen = __hip_atomic_load(prog_my - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (seen < need) __builtin_amdgcn_s_sleep(8);
            }
            if (seen < need && l == 0) atomicOr(&res[p.res_off + 5], 1);   // gave up waiting: the pair's result is not valid
        }
        if (wave_on && c >= 0 && c < nchunks && l < SWB && i0 + l < p.n1) {
            ch = (int)(unsigned char)s1[i0 + l] << 4;
            if (w == 0 && gw > 0) bd = cprev[i0 + 1 + l];
        }
    };
    int ch_nx, bd_nx;
    fetch(0 - w, ch_nx, bd_nx);
    for (int s = 0; s < nchunks + WW - 1; s++) {
        const int c = s - w;
        const int ch1 = ch_nx, bd1 = bd_nx;
        fetch(c + 1, ch_nx, bd_nx);
        if (wave_on && c >= 0 && c < nchunks) {
            const int i0 = c * SWB;
            int bnd = bd1;
            if (w > 0 && l < SWB && i0 + l < p.n1) bnd = hand[w - 1][(s - 1) & 1][l];
            int hl[SWB];
#pragma unroll
            for (int r = 0; r < SWB; r++) hl[r] = 0;
#pragma unroll
            for (int r = 0; r < SWB; r++) {
                if (i0 + r < p.n1) {
                    const int bl = __builtin_amdgcn_readlane(bnd, r), c1 = __builtin_amdgcn_readlane(ch1, r);
                    sw_row<K, 0>(G, c2, c1, bl, bprev, l, lane0, 0, 0, 0, dummy_c, dummy_r);
                    bprev = bl;
#pragma unroll
                    for (int k = 0; k < K; k++) bmk[k] = max(bmk[k], G[k]);
                    const int hlast = G[K - 1] - 8 * (K - 1);
                    if (l == 63) hand[w][s & 1][r] = hlast;
                    hl[r] = hlast;
                }
            }
            // the kept boundary column's rows of this chunk in two 16-byte stores (round 6: one 4-byte store per row and kept column was
            // 101 G of a bench step's 225 G L1 -> L2 write requests, profiles/r06_mem.json)
            if (keeps) {
                if (i0 + SWB <= p.n1) {
                    typedef int v4i_a4 __attribute__((ext_vector_type(4), aligned(4)));
                    *(v4i_a4*)(csave + i0 + 1) = (v4i_a4){hl[0], hl[1], hl[2], hl[3]};
                    *(v4i_a4*)(csave + i0 + 5) = (v4i_a4){hl[4], hl[5], hl[6], hl[7]};
                } else {
#pragma unroll
                    for (int r = 0; r < SWB; r++) if (i0 + r < p.n1) csave[i0 + r + 1] = hl[r];
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
                // lane 63 of the strip's last wave stored the boundary column of these rows: release them to strip ss+1
                if (has_next && w == WW - 1 && l == 63) __hip_atomic_store(prog_my, iend, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (iend < p.n1) {                    // row 64(q+1) is the top boundary of block q+1
                    int* rs = rowsave + p.row_off + (int64_t)(q + 1) * p.pitch + jbase;
#pragma unroll
                    for (int k = 0; k < K; k++) if (jbase + k < p.n2) rs[k] = G[k] - 8 * k;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// ---- packed fill: two columns per 32-bit register --------------------------------------------------------------------
// Scores are at most 5 x min(n1, n2): for pairs up to 13 000 bases they fit 16 bits, and gfx950's packed 16-bit integer instructions
// (v_pk_add / sub / min / max_u16, with op_sel picking halves) work on two columns at once.  A lane owns 8 columns in four
// registers; column k is carried as G' = H + 8 (k + 1) (unsigned: the diagonal entering the lane's first column, H + 0, stays >= 0).
// Same recurrence, same saved rows / columns / block maxima as k_sw_fill<8, 8> — the traceback kernel reads either — at ~6.6 vector
// instructions per cell instead of ~10: the substitution term costs one LDS load per row when the row's base is A / C / G / T (per-lane
// tables of 13 / 4 for the four bases in LDS; any other byte takes xor / min / sub).
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 pk_of(unsigned v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ unsigned un_pk(us2 v) { return __builtin_bit_cast(unsigned, v); }
constexpr int PKK = 8;                      // columns per lane
constexpr int SW_PK_MAXLEN = 13000;         // 5 x 13 000 + 8 x 9 + 13 < 65 536

// one row; P[m] = G' of columns 2m (low half) and 2m + 1 (high half): row i - 1 on entry, row i on return.  sp[m]: 13 where the
// column's base equals the row's, 4 elsewhere (both halves).  bl / bprev: H(i, first column - 1) / H(i - 1, first column - 1) of the wave.
__device__ __forceinline__ void sw_row_pk(unsigned (&P)[4], const unsigned (&sp)[4], const int bl, const int bprev, const int lane8k, const int lane0) {
    // H(i-1, lane's first column - 1): the left lane's last column of the previous row (G' = H + 8 K there)
    const int plast = (int)(P[3] >> 16);
    const int d0 = __builtin_amdgcn_update_dpp(bprev + 8 * PKK, plast, 0x138 /*wave_shr:1*/, 0xf, 0xf, false) - 8 * PKK;
    us2 y[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        // G' of the column to the left, both halves: the diagonal neighbour in the same shift (its + 8 per column is the gap step's)
        const unsigned S = m ? __builtin_amdgcn_alignbit(P[m], P[m - 1], 16) : ((P[0] << 16) | (unsigned)d0);
        const us2 sd = pk_of(S) + pk_of(sp[m]);                                               // diagonal + substitution score (+ 8: the shift)
        const us2 up = __builtin_elementwise_sub_sat(pk_of(P[m]), (us2){8, 8});                // above - 8 (+ 8 - 8 of the shift: the same column)
        const us2 fl = {(unsigned short)(8 * (2 * m + 1)), (unsigned short)(8 * (2 * m + 2))};  // the floor H = 0
        y[m] = __builtin_elementwise_max(__builtin_elementwise_max(sd, up), fl);
    }
    // running maximum over the lane's eight columns (the shift absorbs the gap steps)
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const us2 lo = {y[m].x, y[m].x};
        y[m] = __builtin_elementwise_max(y[m], lo);
        if (m) { const us2 ph = {y[m - 1].y, y[m - 1].y}; y[m] = __builtin_elementwise_max(y[m], ph); }
    }
    // scan of the lanes' last columns (as H + 8 K (lane + 1)); the wave's left boundary enters through lane 0 (as a lane -1)
    int z = (int)y[3].y + lane8k;
    z = max(z, bl + lane0);                  // lane0 = 0 in lane 0, -2^29 elsewhere
    z = wave_scan_max(z);
    const int hl = __builtin_amdgcn_update_dpp(bl, z, 0x138, 0xf, 0xf, false) - lane8k;   // H(i, lane's first column - 1): reaches column k as hl in the G' shift
    const us2 hl2 = {(unsigned short)hl, (unsigned short)hl};
#pragma unroll
    for (int m = 0; m < 4; m++) P[m] = un_pk(__builtin_elementwise_max(y[m], hl2));
}

// (at most 64 registers: a wave then fits beside two 224-register sweep waves on a SIMD; the compiler takes 79 when left alone)
template <int WW>
__global__ __launch_bounds__(64 * WW) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_sw_fill_pk(const SwPair* pairs, const char* chars, int* rowsave, int* colsave,
                                                       int* blkmax, int* prog, int* ticket, int* res) {
    constexpr int K = PKK;
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ticket[blockIdx.y], 1);
    __syncthreads();
    const int ss = s_ticket;
    const SwPair p = pairs[blockIdx.y];
    if (p.n1 <= 0 || p.n2 <= 0 || ss * WW * 64 * K >= p.n2) return;
    int* prog_my = prog + (int64_t)blockIdx.y * gridDim.x + ss;
    const bool has_next = (ss + 1) * WW * 64 * K < p.n2;
    int seen = 0;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int gw = ss * WW + w;
    const int wfirst = gw * 64 * K;
    const bool wave_on = wfirst < p.n2;
    const int jbase = wfirst + l * K;
    const char* s1 = chars + p.s1_off;
    const char* s2 = chars + p.s2_off;
    // substitution scores of the lane's eight columns against each of the four bases, in LDS: a row reads its table with one 16-byte
    // LDS load (no vector instruction per column; tables in registers picked by a uniform branch cost 17 register moves per row)
    __shared__ uint4 s_sp[4][64 * WW];
    unsigned P[4], bmk[4], c2p[4];
    {
        unsigned tb[4][4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const unsigned a = jbase + 2 * m < p.n2 ? (unsigned)(unsigned char)s2[jbase + 2 * m] : 0u, b = jbase + 2 * m + 1 < p.n2 ? (unsigned)(unsigned char)s2[jbase + 2 * m + 1] : 0u;
            c2p[m] = (a << 4) | (b << 20);
            auto tab = [&](unsigned ch) { return (a == ch ? 13u : 4u) | ((b == ch ? 13u : 4u) << 16); };
            tb[0][m] = tab('A'); tb[1][m] = tab('C'); tb[2][m] = tab('G'); tb[3][m] = tab('T');
            P[m] = (unsigned)(8 * (2 * m + 1)) | ((unsigned)(8 * (2 * m + 2)) << 16);
            bmk[m] = 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) s_sp[q][threadIdx.x] = make_uint4(tb[q][0], tb[q][1], tb[q][2], tb[q][3]);
    }
    const int lane0 = l ? -(1 << 29) : 0, lane8k = 8 * K * l;
    __shared__ int hand[WW][2][SWB];
    const bool keeps = ((jbase + K) & 63) == 0 && jbase + K <= p.n2;
    int* csave = colsave + p.col_off + (int64_t)((jbase + K) >> 6) * (p.n1 + 1);
    const int* cprev = colsave + p.col_off + (int64_t)(wfirst >> 6) * (p.n1 + 1);
    if (wave_on && keeps) csave[0] = 0;
    int bprev = 0;
    const int nchunks = (p.n1 + SWB - 1) / SWB;
    auto fetch = [&](int c, int& ch, int& bd) {
        ch = 1; bd = 0;
        const int i0 = c * SWB;
        if (w == 0 && ss > 0 && c >= 0 && c < nchunks) {
            const int need = min(p.n1, (i0 + SWB + 63) & ~63);
            int spins = 0;
            for (; seen < need && spins < SW_SPIN_LIMIT; spins++) {
                seen = __hip_atomic_load(prog_my - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (seen < need) __builtin_amdgcn_s_sleep(8);
            }
            if (seen < need && l == 0) atomicOr(&res[p.res_off + 5], 1);
        }
        if (wave_on && c >= 0 && c < nchunks && l < SWB && i0 + l < p.n1) {
            ch = (int)(unsigned char)s1[i0 + l];
            if (w == 0 && gw > 0) bd = cprev[i0 + 1 + l];
        }
    };
    int ch_nx, bd_nx;
    fetch(0 - w, ch_nx, bd_nx);
    for (int s = 0; s < nchunks + WW - 1; s++) {
        const int c = s - w;
        const int ch1 = ch_nx, bd1 = bd_nx;
        fetch(c + 1, ch_nx, bd_nx);
        if (wave_on && c >= 0 && c < nchunks) {
            const int i0 = c * SWB;
            int bnd = bd1;
            if (w > 0 && l < SWB && i0 + l < p.n1) bnd = hand[w - 1][(s - 1) & 1][l];
            int hl[SWB];
#pragma unroll
            for (int r = 0; r < SWB; r++) hl[r] = 0;
#pragma unroll
            for (int r = 0; r < SWB; r++) {
                if (i0 + r < p.n1) {
                    const int bl = __builtin_amdgcn_readlane(bnd, r), c1 = __builtin_amdgcn_readlane(ch1, r);   // c1: the row's base (uniform)
                    const int bidx = c1 == 'A' ? 0 : c1 == 'C' ? 1 : c1 == 'G' ? 2 : c1 == 'T' ? 3 : -1;       // (scalar selects)
                    unsigned sp[4];
                    if (bidx >= 0) {
                        const uint4 v = s_sp[bidx][threadIdx.x];
                        sp[0] = v.x; sp[1] = v.y; sp[2] = v.z; sp[3] = v.w;
                    } else {
                        const unsigned c1p = ((unsigned)c1 << 4) | ((unsigned)c1 << 20);
#pragma unroll
                        for (int m = 0; m < 4; m++) sp[m] = un_pk((us2){13, 13} - __builtin_elementwise_min(pk_of(c2p[m] ^ c1p), (us2){9, 9}));
                    }
                    sw_row_pk(P, sp, bl, bprev, lane8k, lane0);
                    bprev = bl;
#pragma unroll
                    for (int m = 0; m < 4; m++) bmk[m] = un_pk(__builtin_elementwise_max(pk_of(bmk[m]), pk_of(P[m])));
                    const int hlast = (int)(P[3] >> 16) - 8 * K;
                    if (l == 63) hand[w][s & 1][r] = hlast;
                    hl[r] = hlast;
                }
            }
            // the kept boundary column's rows of this chunk in two 16-byte stores (round 6: one 4-byte store per row and kept column was
            // 101 G of a bench step's 225 G L1 -> L2 write requests, profiles/r06_mem.json)
            if (keeps) {
                if (i0 + SWB <= p.n1) {
                    typedef int v4i_a4 __attribute__((ext_vector_type(4), aligned(4)));
                    *(v4i_a4*)(csave + i0 + 1) = (v4i_a4){hl[0], hl[1], hl[2], hl[3]};
                    *(v4i_a4*)(csave + i0 + 5) = (v4i_a4){hl[4], hl[5], hl[6], hl[7]};
                } else {
#pragma unroll
                    for (int r = 0; r < SWB; r++) if (i0 + r < p.n1) csave[i0 + r + 1] = hl[r];
                }
            }
            const int iend = min(i0 + SWB, p.n1);
            if ((iend & 63) == 0 || iend == p.n1) {   // row block q complete
                const int q = (iend - 1) >> 6;
                int bm = 0;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    bm = max(bm, (int)(bmk[m] & 0xffffu) - 8 * (2 * m + 1));
                    bm = max(bm, (int)(bmk[m] >> 16) - 8 * (2 * m + 2));
                    bmk[m] = 0u;
                }
                for (int o = 32; o; o >>= 1) bm = max(bm, __shfl_xor(bm, o));
                if (l == 0) blkmax[p.blk_off + (int64_t)q * p.ngw + gw] = bm;
                if (has_next && w == WW - 1 && l == 63) __hip_atomic_store(prog_my, iend, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (iend < p.n1) {
                    int* rs = rowsave + p.row_off + (int64_t)(q + 1) * p.pitch + jbase;
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        if (jbase + 2 * m < p.n2) rs[2 * m] = (int)(P[m] & 0xffffu) - 8 * (2 * m + 1);
                        if (jbase + 2 * m + 1 < p.n2) rs[2 * m + 1] = (int)(P[m] >> 16) - 8 * (2 * m + 2);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// recompute rows 64q+1 .. 64q+nrows of the strip of 64K columns that starts at 0-based column c0 (a multiple of 64); one wave
template <int K, int MODE>
__device__ __forceinline__ void sw_tile(const SwPair& p, const char* s1, const char* s2, const int* rowsave, const int* colsave,
                                        const int q, const int c0, const int nrows, const int l, const int target, int& fc, int& fr,
                                        unsigned char (*codes)[64]) {
    const int jbase = c0 + l * K;
    int c2[K], G[K];
    const int* rs = rowsave + p.row_off + (int64_t)q * p.pitch + jbase;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const bool ok = jbase + k < p.n2;
        c2[k] = ok ? (int)(unsigned char)s2[jbase + k] << 4 : 0;
        G[k] = ((ok && q > 0) ? rs[k] : 0) + 8 * k;
    }
    const int* cprev = colsave + p.col_off + (int64_t)(c0 >> 6) * (p.n1 + 1);   // H(*, c0)
    int bprev = c0 > 0 ? cprev[64 * q] : 0;
    int bnd = 0, ch1 = 1;
    if (l < nrows) { ch1 = (int)(unsigned char)s1[64 * q + l] << 4; if (c0 > 0) bnd = cprev[64 * q + 1 + l]; }
    const int lane0 = l ? -(1 << 29) : 0;
    for (int r = 0; r < nrows; r++) {
        const int bl = __builtin_amdgcn_readlane(bnd, r), c1 = __builtin_amdgcn_readlane(ch1, r);
        const unsigned cd = sw_row<K, MODE>(G, c2, c1, bl, bprev, l, lane0, target, 64 * q + r + 1, jbase, fc, fr);
        if (MODE == 2) codes[r][l] = (unsigned char)cd;   // MODE 2 runs with K == 1
        bprev = bl;
    }
}

// ---- locate the starting cell, then trace back; grid (pairs), block 64 ------------------------------------------
template <int K>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K <= 8 ? 8 : 4, 8))) void k_sw_trace(const SwPair* pairs, const char* chars, const int* rowsave, const int* colsave,
                                                 const int* blkmax, int* out, int* res) {
    __shared__ unsigned char codes[64][64];
#ifndef PS_WALKER_PRIO
#define PS_WALKER_PRIO 3
#endif
    if (PS_WALKER_PRIO > 0) __builtin_amdgcn_s_setprio(PS_WALKER_PRIO);   // one serial wave per pair at the end of FindMutations' chain (ps_dev.h, chain_priority)
    const SwPair p = pairs[blockIdx.x];
    const int l = threadIdx.x;
    int* o = res + p.res_off;
    const char* s1 = chars + p.s1_off;
    const char* s2 = chars + p.s2_off;
    const int* bmx = blkmax + p.blk_off;
    int best = 0;
    if (p.n1 > 0 && p.n2 > 0)
        for (int k = l; k < p.nrb * p.ngw; k += 64) best = max(best, bmx[k]);
    for (int s = 32; s; s >>= 1) best = max(best, __shfl_xor(best, s));
    if (best <= 0) {
        if (l == 0) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; }
        return;
    }
    // first cell with the maximum in column-major order: smallest column, then smallest row
    int bi = 0, bj = 0;
    for (int gw = 0; gw < p.ngw && !bj; gw++) {
        int fc = 0x7fffffff, fr = 0x7fffffff;
        for (int q0 = 0; q0 < p.nrb; q0 += 64) {
            const int q = q0 + l;
            unsigned long long hit = __ballot(q < p.nrb && bmx[(int64_t)q * p.ngw + gw] == best);
            while (hit) {
                const int qq = q0 + (int)__builtin_ctzll(hit);
                hit &= hit - 1;
                sw_tile<K, 1>(p, s1, s2, rowsave, colsave, qq, gw * 64 * K, min(64, p.n1 - 64 * qq), l, best, fc, fr, codes);
            }
        }
        for (int s = 32; s; s >>= 1) {
            const int oc = __shfl_xor(fc, s), orow = __shfl_xor(fr, s);
            if (oc < fc || (oc == fc && orow < fr)) { fc = oc; fr = orow; }
        }
        if (fc != 0x7fffffff) { bj = fc; bi = fr; }
    }
    int* oi = out + p.out_off;
    int* oj = oi + (p.n1 + p.n2 + 2);
    int i = bi, j = bj, np = 0, nm = 0;
    bool done = false;
    while (!done && i > 0 && j > 0) {
        const int q = (i - 1) >> 6, cb = (j - 1) >> 6;   // 64 x 64 tile holding (i, j), one column per lane
        const int r0 = 64 * q + 1, cfirst = 64 * cb + 1;
        int dc = 0, dr = 0;
        __syncthreads();
        sw_tile<1, 2>(p, s1, s2, rowsave, colsave, q, 64 * cb, i - 64 * q, l, 0, dc, dr, codes);
        __syncthreads();
        // lane m looks m cells ahead on the diagonal: a run of diagonal steps is emitted at once
        while (true) {
            if (!(i > 0 && j > 0)) { done = true; break; }
            if (i < r0 || j < cfirst) break;   // left the tile
            const int ii = i - l, jj = j - l;
            unsigned code = 0;
            if (ii >= r0 && jj >= cfirst) code = codes[ii - r0][jj - cfirst];
            const unsigned long long dm = __ballot((code & 7u) == 7u);   // diagonal step from a cell with score > 0
            const int run = __builtin_amdgcn_readfirstlane(dm == ~0ull ? 64 : (int)__builtin_ctzll(~dm));
            if (l < run) { oi[np + l] = ii; oj[np + l] = jj; }
            nm += (int)__popcll(__ballot((code & 8u) && l < run));
            np += run; i -= run; j -= run;
            if (run == 64) continue;
            if (!(i > 0 && j > 0)) { done = true; break; }
            if (i < r0 || j < cfirst) break;
            const unsigned cr = __builtin_amdgcn_readlane(code, run);
            if (!(cr & 4u)) { done = true; break; }   // score <= 0
            const unsigned stp = cr & 3u;
            if (stp == 1u) { if (l == 0) { oi[np] = 0; oj[np] = j; } np++; j--; }
            else if (stp == 2u) { if (l == 0) { oi[np] = i; oj[np] = 0; } np++; i--; }
            else { done = true; break; }
        }
    }
    if (l == 0) { o[0] = best; o[1] = bi; o[2] = bj; o[3] = np; o[4] = nm; }
}

// the packed fill (8 columns per lane, 8 waves) with the traceback of the 8-column build
static int sw_run_pk(Runtime* rt, hipStream_t st, int np, int nss, const SwPair* d_pairs, const char* d_chars, int* d_row, int* d_col,
                     int* d_blk, int* d_prog, int* d_ticket, int* d_out, int* d_res) {
    // PORESEQ_SW_LDS_PAD_KB (tuning): dynamic LDS a strip's workgroup claims on top of its own, i.e. a cap on the strips resident per CU: a
    // batch of 340 pairs is 1 700 chained strips of eight waves that mostly wait for their left neighbours and would take every free wave slot
    static const size_t pad = getenv("PORESEQ_SW_LDS_PAD_KB") ? (size_t)atoi(getenv("PORESEQ_SW_LDS_PAD_KB")) * 1024 : 0;
    hipLaunchKernelGGL((k_sw_fill_pk<SWW>), dim3(nss, np), dim3(64 * SWW), pad, st, d_pairs, d_chars, d_row, d_col, d_blk, d_prog, d_ticket, d_res);
    PS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sw_trace<8>, dim3(np), dim3(64), 0, st, d_pairs, d_chars, d_row, d_col, d_blk, d_out, d_res);
    PS_HIP(hipGetLastError());
    return PS_OK;
}

template <int K, int WW>
static int sw_run(Runtime* rt, hipStream_t st, int np, int nss, const SwPair* d_pairs, const char* d_chars, int* d_row, int* d_col,
                  int* d_blk, int* d_prog, int* d_ticket, int* d_out, int* d_res) {
    hipLaunchKernelGGL((k_sw_fill<K, WW>), dim3(nss, np), dim3(64 * WW), 0, st, d_pairs, d_chars, d_row, d_col, d_blk, d_prog, d_ticket, d_res);
    PS_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sw_trace<K>, dim3(np), dim3(64), 0, st, d_pairs, d_chars, d_row, d_col, d_blk, d_out, d_res);
    PS_HIP(hipGetLastError());
    return PS_OK;
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
    int maxn2 = 1;
    for (int k = 0; k < np; k++) maxn2 = std::max(maxn2, (int)in[k].second->size());
    // 4 columns per lane = 2048 per workgroup: a 10 kb pair runs as five chained workgroups, all but the last full (8 columns per
    // lane: three, the last 44 % used, 7.8 instead of 6.6 ms per pair).  8 columns per lane spread the row scan over twice the cells:
    // 17 % fewer vector instructions per pair — with several lock-step batches in flight the chip is short of vector issue, not of
    // latency, and the wide build is the faster one (bench: 189.9 against 184.5 kb/s).  PORESEQ_SW_K forces either (tests).
    int K = live_runtimes() > 1 ? 8 : 4, WW = SWW;
    if (const char* e = getenv("PORESEQ_SW_K")) K = atoi(e) == 16 ? 16 : (atoi(e) == 8 ? 8 : 4);
    // PORESEQ_SW_FORM=one: one workgroup of 16 waves per pair when the longest sequence fits 16 x 64 x K columns — no chained workgroups,
    // nothing spins.  Measured and NOT the default: in the bench (14 lock-step batches in flight) 176.9 kb/s against 190.3 with the
    // chained strips — a 10 kb pair keeps 10 of the 16 waves busy, holds a whole CU, and meets at a 16-wave barrier every eight rows.
    {
        const char* form = getenv("PORESEQ_SW_FORM");   // (a test hook that the tests change between calls of one process: read per launch)
        const int k1 = maxn2 <= SWW1 * 64 * 8 ? 8 : (maxn2 <= SWW1 * 64 * 16 ? 16 : 0);   // (K divides 64: every 64th column is some lane's last)
        if (form && !strcmp(form, "one") && k1) { K = k1; WW = SWW1; }
    }
    const int sswidth = WW * 64 * K;
    const int nss = (maxn2 + sswidth - 1) / sswidth;
    int64_t row_tot = 0, col_tot = 0, blk_tot = 0, out_tot = 0;
    for (int k = 0; k < np; k++) {
        SwPair& p = pairs[k];
        p.n1 = (int)in[k].first->size(); p.n2 = (int)in[k].second->size();
        p.nrb = std::max(1, (p.n1 + 63) / 64);
        p.ngw = WW * std::max(1, (p.n2 + sswidth - 1) / sswidth);
        p.pitch = ((p.n2 + 3) / 4) * 4 + 4;
        p.s1_off = (int64_t)pool.size(); pool += *in[k].first;
        p.s2_off = (int64_t)pool.size(); pool += *in[k].second;
        p.row_off = row_tot; row_tot += (int64_t)p.nrb * p.pitch;
        p.col_off = col_tot; col_tot += ((int64_t)p.n2 / 64 + 1) * (p.n1 + 1);
        p.blk_off = blk_tot; blk_tot += (int64_t)p.nrb * p.ngw;
        p.out_off = out_tot; out_tot += 2 * ((int64_t)p.n1 + p.n2 + 2);
        p.res_off = (int64_t)k * 8;
        job->cells += (double)p.n1 * p.n2;
    }
    pool.push_back(0);
    job->out_tot = out_tot;
    PS_TRY(rt->buf("sw_pairs").ensure(np * sizeof(SwPair)));
    PS_TRY(rt->buf("sw_chars").ensure(pool.size()));
    PS_TRY(rt->buf("sw_row").ensure(row_tot * sizeof(int)));
    PS_TRY(rt->buf("sw_col").ensure(col_tot * sizeof(int)));
    PS_TRY(rt->buf("sw_blk").ensure(blk_tot * sizeof(int)));
    PS_TRY(rt->buf("sw_prog").ensure((size_t)np * (nss + 1) * sizeof(int)));   // progress per strip + one ticket counter per pair
    PS_TRY(rt->buf("sw_out").ensure(out_tot * sizeof(int)));
    PS_TRY(rt->buf("sw_res").ensure((size_t)np * 8 * sizeof(int)));
    SwPair* d_pairs = rt->buf("sw_pairs").as<SwPair>();
    char* d_chars = rt->buf("sw_chars").as<char>();
    int* d_row = rt->buf("sw_row").as<int>();
    int* d_col = rt->buf("sw_col").as<int>();
    int* d_blk = rt->buf("sw_blk").as<int>();
    int* d_prog = rt->buf("sw_prog").as<int>();
    int* d_ticket = d_prog + (size_t)np * nss;
    int* d_out = rt->buf("sw_out").as<int>();
    int* d_res = rt->buf("sw_res").as<int>();
    hipStream_t st = nullptr;
    PS_TRY(second_stream(rt, &st));
    job->stream = st;
    PS_TRY(rt->up(d_pairs, pairs.data(), np * sizeof(SwPair), st));
    PS_TRY(rt->up(d_chars, pool.data(), pool.size(), st));
    PS_HIP(hipMemsetAsync(d_res, 0, (size_t)np * 8 * sizeof(int), st));
    PS_HIP(hipMemsetAsync(d_blk, 0, blk_tot * sizeof(int), st));   // waves beyond a pair's last column never write theirs
    PS_HIP(hipMemsetAsync(d_prog, 0, (size_t)np * (nss + 1) * sizeof(int), st));
    if (rt->prof_on) PS_HIP(hipEventRecord(rt->sw0, st));
    // the packed 16-bit fill serves the 8-column build whenever every pair's scores fit 16 bits (PORESEQ_SW_PK=0: never; tests)
    bool packed = K == 8 && WW == SWW;
    for (int k = 0; k < np && packed; k++) if (std::min(pairs[k].n1, pairs[k].n2) > SW_PK_MAXLEN) packed = false;
    if (const char* e = getenv("PORESEQ_SW_PK")) if (atoi(e) == 0) packed = false;
    if (packed) {
        PS_TRY(sw_run_pk(rt, st, np, nss, d_pairs, d_chars, d_row, d_col, d_blk, d_prog, d_ticket, d_out, d_res));
        if (rt->prof_on) rt->prof["sw_pk8"].launches++;   // (which fill ran: a host-side count, no event pair)
    } else if (WW == SWW1) {
        switch (K) {
            case 8: PS_TRY((sw_run<8, SWW1>(rt, st, np, nss, d_pairs, d_chars, d_row, d_col, d_blk, d_prog, d_ticket, d_out, d_res))); break;
            default: PS_TRY((sw_run<16, SWW1>(rt, st, np, nss, d_pairs, d_chars, d_row, d_col, d_blk, d_prog, d_ticket, d_out, d_res))); break;
        }
    } else {
        switch (K) {
            case 4: PS_TRY((sw_run<4, SWW>(rt, st, np, nss, d_pairs, d_chars, d_row, d_col, d_blk, d_prog, d_ticket, d_out, d_res))); break;
            case 16: PS_TRY((sw_run<16, SWW>(rt, st, np, nss, d_pairs, d_chars, d_row, d_col, d_blk, d_prog, d_ticket, d_out, d_res))); break;
            default: PS_TRY((sw_run<8, SWW>(rt, st, np, nss, d_pairs, d_chars, d_row, d_col, d_blk, d_prog, d_ticket, d_out, d_res))); break;
        }
    }
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
    PS_HIP(hipStreamSynchronize(job->stream));
    if (rt->prof_on) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, rt->sw0, rt->sw1) == hipSuccess) {
            Prof& pr = rt->prof["sw"];
            pr.ms += ms; pr.launches += 1; pr.bytes += job->cells * 5.0;  // 4-byte score + 1-byte step per cell (the reference's footprint)
        }
    }
    for (int k = 0; k < np; k++)
        if (job->res[k * 8 + 5]) return fail(PS_ERR_HIP, "Smith-Waterman: a strip gave up waiting for its left neighbour (result discarded)");
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
