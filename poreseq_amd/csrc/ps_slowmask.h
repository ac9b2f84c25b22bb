// Which 8-step bodies of k_fill must run the SLOW variant — pure integer logic, shared by the kernel (ps_kernels.hip) and by a
// host test of it (tests/native/slowmask_check.cpp).
#pragma once
#if defined(__HIPCC__)
#define PS_SLOWMASK_FN __host__ __device__ __forceinline__
#else
#define PS_SLOWMASK_FN static inline
#endif

// `rm` has one bit per anti-diagonal of a 64-step chunk: the band resumes there after an empty stretch (lo(t-1) < 0 <= lo(t)).
// A resume on anti-diagonal t makes every body holding a step of [t - 6, t + 6] slow, i.e. a body starting at s0 is slow for a
// resume in [s0 - 6, s0 + 13].  Result — bit 7: the last body of the chunk before, bits 8-15: the chunk's own eight bodies,
// bit 16: the first body of the chunk after.
PS_SLOWMASK_FN unsigned resume_spread(unsigned long long rm) {
    if (!rm) return 0u;
    unsigned r = (rm & 0x3Full) ? 0x80u : 0u;
    r |= (rm >> 58) ? 0x10000u : 0u;
#pragma unroll
    for (int bd = 0; bd < 8; bd++) {
        const int a = 8 * bd - 6;
        const unsigned long long w = a < 0 ? (0xFFFFFull >> -a) : (0xFFFFFull << a);
        r |= (rm & w) ? (0x100u << bd) : 0u;
    }
    return r;
}
