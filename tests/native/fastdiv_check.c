// fastdiv_check.c — test program (tests/test_fastdiv.py): the division k_fill uses for its emissions,
//   q0 = a*y; r = fma(-b,q0,a); q1 = fma(r,y,q0); r = fma(-b,q1,a); q = fma(r,y,q1)   with y = RN(1/b)   (Markstein),
// against IEEE division, bit for bit, on operand pairs drawn from the three divisions of an emission (cpp/AlignUtil.h:34-53)
// and on random / all-ones / power-of-two significands over 80 binades.  Prints the mismatch count; exit status 1 on any.
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static inline double mdiv(double a, double b, double y) {
    double q = a * y;
    double r = fma(-b, q, a);
    q = fma(r, y, q);
    r = fma(-b, q, a);
    return fma(r, y, q);
}
static uint64_t rng = 88172645463325252ull;
static inline uint64_t xs() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
static inline double urand() { return (xs() >> 11) * (1.0 / 9007199254740992.0); }
int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 100000000; if (argc > 2) rng ^= strtoull(argv[2], 0, 10) * 0x9E3779B97F4A7C15ull;
    long bad = 0, bad1 = 0;
    for (long k = 0; k < n; k++) {
        double a, b;
        int mode = k & 3;
        if (mode == 0) { a = (urand() - 0.5) * 200; b = 0.3 + urand() * 3; }          // (x - mu) / sg
        else if (mode == 1) { a = (urand() - 0.5) * 4; b = 0.5 + urand() * 2; }        // (sd - sm) / sm
        else if (mode == 2) { a = urand() * 50; b = 0.2 + urand() * 4; }               // e*e*lam / sd
        else {   // random mantissas / wide exponents
            uint64_t ua = (xs() & 0x800fffffffffffffull) | ((uint64_t)(1023 - 40 + xs() % 80) << 52);
            uint64_t ub = (xs() & 0x000fffffffffffffull) | ((uint64_t)(1023 - 40 + xs() % 80) << 52);
            memcpy(&a, &ua, 8); memcpy(&b, &ub, 8);
            if ((k & 0xff) == 3) { ub |= 0x000fffffffffffffull; memcpy(&b, &ub, 8); }   // all-ones significand
            if ((k & 0xff) == 7) { ub &= ~0x000fffffffffffffull; memcpy(&b, &ub, 8); }  // power of two
        }
        const double y = 1.0 / b;
        const double q = mdiv(a, b, y), t = a / b;
        if (memcmp(&q, &t, 8)) { if (!(q == t)) bad++; else bad1++; if (bad + bad1 < 10) printf("mismatch a=%a b=%a q=%a t=%a\n", a, b, q, t); }
    }
    printf("n=%ld mismatches=%ld (sign-of-zero only: %ld)\n", n, bad, bad1);
    return (bad || bad1) ? 1 : 0;
}
