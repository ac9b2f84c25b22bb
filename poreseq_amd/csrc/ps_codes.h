// ps_codes.h — the strip sweeps' per-cell codes: layout in memory and meaning of the bits.  Pure integer logic shared by the kernels
// (ps_sweep_body.h writes them, ps_sweep.hip's StripCodes and ps_debug.cpp read them) and by a host test (tests/native/codes_check.cpp).
#ifndef PS_CODES_H_
#define PS_CODES_H_

#if defined(__HIPCC__)
#define PS_CODES_FN __host__ __device__
#else
#define PS_CODES_FN
#endif

namespace ps {

// ---- layout of one step's codes: row groups ("planes") of 16 / 8 / 4 / 2 / 1 rows, each [NL lanes][bytes of the group], one byte per
// row.  A lane builds the codes of up to four consecutive rows in ONE register by shifting predicate bits in at the bottom (code_push),
// CODE_BITS per row: the first row of a register sits in its highest field.  So a plane of four or more rows is 32-bit words of four
// 7-bit fields (row 4w + k of the plane in bits 7 (3 - k) .. of word w), a plane of two rows a 16-bit word of two, a plane of one a byte.
constexpr int CODE_BITS = 7;
PS_CODES_FN constexpr int plane_sz(int rem) { return rem >= 16 ? 16 : rem >= 8 ? 8 : rem >= 4 ? 4 : rem >= 2 ? 2 : 1; }
// the code of row r (0 .. K - 1) of a lane's strip in the bytes of one step (`step` = the step's first byte, NL K bytes)
PS_CODES_FN inline unsigned code_fetch(const unsigned char* step, int K, int lane, int r, int nl) {
    int r0 = 0;
    for (;;) {
        const int sz = plane_sz(K - r0);
        if (r < r0 + sz) {
            const int rr = r - r0;
            const unsigned char* p = step + nl * r0 + lane * sz;
            if (sz >= 4) return (*(const unsigned*)(p + (rr & ~3)) >> (CODE_BITS * (3 - (rr & 3)))) & 0x7Fu;
            if (sz == 2) return ((unsigned)*(const unsigned short*)p >> (CODE_BITS * (1 - rr))) & 0x7Fu;
            return (unsigned)*p & 0x7Fu;
        }
        r0 += sz;
    }
}

// ---- one cell's code: seven raw predicate bits of the fill (cpp/Alignment.cpp:196-267), decoded by the few readers ----
// The sweep does not work the reference's step codes out per cell (a chain of nine selects and four shifts / ors per cell, a fifth of
// a forward cell's vector instructions for a byte the backtrace reads on ~10 000 of 6 000 000 cells): every bit is ONE compare whose
// lane mask goes from its scalar register pair straight into the byte as the carry of an add-with-carry.  The reader combines them
// with what it can look up itself: whether the cell is in its column's band (act), whether the diagonal neighbour is in the
// previous column's band (vd: MATCH against implicit MATCH, cpp/Alignment.cpp:207-220) and whether the column has a 5-mer.
enum : unsigned { CB_POS = 1u,     // main score > 0
                  CB_IGN = 2u, CB_INS = 4u, CB_MATCH = 8u, CB_SKIP = 16u,   // candidate == the cell's main score; the first in the reference's order SKIP, MATCH, INSERT, IGNORE wins, none: STAY
                  CB_SPOS = 32u,   // stay score > 0
                  CB_EXT = 64u };  // EXTEND > max(floor, STAY)  (strict: the stay matrix takes EXTEND)
// ("STAY > floor" needs no bit: where the stay score is positive and EXTEND did not take it, STAY did — the floor of a row that has an
//  upper neighbour is 0; where it is not positive neither candidate beat the floor: a first row's -1e300 loses to nothing its candidates
//  can be, which are built on the absent-cell value — and the reference's step stays 0)
// main step (0 SKIP, 1 MATCH, 2 INSERT, 3 IGNORE, 4 STAY, 255 implicit MATCH; 0 when the score is not positive) and stay step (0, 4 STAY, 5 EXTEND)
PS_CODES_FN inline unsigned code_main_step(unsigned by, bool vd) {
    if (!(by & CB_POS)) return 0u;
    return (by & CB_SKIP) ? 0u : (by & CB_MATCH) ? (vd ? 1u : 255u) : (by & CB_INS) ? 2u : (by & CB_IGN) ? 3u : 4u;
}
PS_CODES_FN inline unsigned code_stay_step(unsigned by) { return (by & CB_EXT) ? 5u : (by & CB_SPOS) ? 4u : 0u; }

}  // namespace ps
#endif
