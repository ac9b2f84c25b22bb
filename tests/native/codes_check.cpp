// Host check of the strip sweeps' code layout (poreseq_amd/csrc/ps_codes.h): a lane shifts CODE_BITS predicate bits per row into a
// register, four rows to a register, first row on top (ps_sweep_body.h: code_first / code_push), and stores the registers of a step
// plane by plane (put_codes); the readers get a cell's bits back with code_fetch.  Here the packing and the plane stores are
// re-stated on the host for every strip height the kernels are built for and every lane count, random bits go in, code_fetch must
// hand the same bits back; then the decoders' truth tables (code_main_step / code_stay_step against the reference's selection order,
// cpp/Alignment.cpp:240-267).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../poreseq_amd/csrc/ps_codes.h"

using namespace ps;

// what put_codes does with the registers cw[] of one lane (ps_sweep_body.h), byte for byte
static void put_codes_host(unsigned char* step, int K, int nl, int lane, const std::vector<unsigned>& cw) {
    int r0 = 0;
    for (;;) {
        const int sz = plane_sz(K - r0);
        unsigned char* p = step + nl * r0 + lane * sz;
        if (sz >= 4) {
            for (int w = 0; w < sz / 4; w++) memcpy(p + 4 * w, &cw[r0 / 4 + w], 4);
        } else {
            const int nrow = K - (r0 & ~3) < 4 ? K - (r0 & ~3) : 4;
            const int sh = CODE_BITS * (nrow - (r0 & 3) - sz);
            if (sz == 2) { const unsigned short v = (unsigned short)(cw[r0 / 4] >> sh); memcpy(p, &v, 2); }
            else *p = (unsigned char)(cw[r0 / 4] >> sh);
        }
        r0 += sz;
        if (r0 >= K) break;
    }
}

int main() {
    long bad = 0, cells = 0;
    const int Ks[] = {2, 3, 4, 5, 6, 10, 16, 24, 32};
    const int NLs[] = {64, 128, 256};
    srand(7);
    for (int K : Ks)
        for (int nl : NLs) {
            std::vector<unsigned char> step((size_t)nl * K, 0xAA);
            std::vector<std::vector<unsigned>> want(nl, std::vector<unsigned>(K));
            for (int lane = 0; lane < nl; lane++) {
                std::vector<unsigned> cw((K + 3) / 4, 0u);
                unsigned acc = 0;
                for (int r = 0; r < K; r++) {
                    const unsigned bits = (unsigned)rand() & 0x7Fu;
                    want[lane][r] = bits;
                    // code_first + six code_push: the first bit pushed ends on top of the row's field
                    if ((r & 3) == 0) acc = 0;
                    for (int b = CODE_BITS - 1; b >= 0; b--) acc = acc + acc + ((bits >> b) & 1u);
                    if ((r & 3) == 3 || r == K - 1) cw[r >> 2] = acc;
                }
                put_codes_host(step.data(), K, nl, lane, cw);
            }
            for (int lane = 0; lane < nl; lane++)
                for (int r = 0; r < K; r++) { cells++; if (code_fetch(step.data(), K, lane, r, nl) != want[lane][r]) bad++; }
        }
    // decoders: every 7-bit pattern, both vd
    for (unsigned by = 0; by < 128; by++)
        for (int vd = 0; vd < 2; vd++) {
            unsigned sm;
            if (!(by & CB_POS)) sm = 0;                                   // a cell whose main score is not positive keeps step 0
            else if (by & CB_SKIP) sm = 0;                                // the reference's order: SKIP, MATCH, INSERT, IGNORE, else the stay matrix
            else if (by & CB_MATCH) sm = vd ? 1 : 255;
            else if (by & CB_INS) sm = 2;
            else if (by & CB_IGN) sm = 3;
            else sm = 4;
            const unsigned ss = (by & CB_EXT) ? 5 : (by & CB_SPOS) ? 4 : 0;
            cells++;
            if (code_main_step(by, vd != 0) != sm || code_stay_step(by) != ss) bad++;
        }
    printf("cells=%ld mismatches=%ld\n", cells, bad);
    return bad ? 1 : 0;
}
