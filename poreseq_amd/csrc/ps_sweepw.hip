// ps_sweepw.hip — the strip sweep of ps_sweep.hip on TWO or FOUR wavefronts per (event, sequence, direction): one workgroup of
// 128 / 256 lanes per sweep, K = 4 / 2 rows per lane for the default band (realign_width 300: ~120 / ~200 strips in band on a step).
//
// Why: one wavefront per alignment (the north_star's form, k_sweep / k_sweeps) takes ~27 ms for a 10 kb event whatever the chip
// could do — a wavefront alone on its SIMD issues one vector instruction per ~5 cycles — and needs ~220 registers at K = 10
// (two wavefronts per SIMD).  A consensus schedule's launches are 20-400 sweeps (MakeMutations' recursion rounds,
// cpp/MakeMutations.cpp:142-143: 1-7 regions of a lock-step batch still changing), far fewer than the chip's 1 024 SIMDs.  Spreading
// a sweep over NW wavefronts cuts the rows a lane carries (K = 4: the level records are 32 registers instead of 80 -> three or four
// wavefronts per SIMD) and the time of a sweep by ~NW, at the same SIMD time per sweep for NW = 2 (120 of 128 lanes busy against
// 57 of 64; the per-step overhead of a lane — band / 5-mer / model-row fetch, neighbour exchange, code store — is spread over 4 cells
// instead of 10, but there are C + n0 / 4 steps instead of C + n0 / 10) and ~1.3x for NW = 4.
//
// Reference behaviour reproduced: the same as ps_sweep.hip (cpp/Alignment.cpp:111-274 fillColumn, :280-444 fillColumnBack, :158 / :270
// MaxInfo, move order and strict '>' of :196-270), cell for cell — the body is shared (ps_sweep_body.h, NW is a template parameter);
// what differs is the one cross-wavefront hand-off per step (bottom cell of lane 63 -> top of the next wavefront's lane 0, through a
// two-slot LDS record behind one LDS-only barrier) and the lane count in every layout ([step][row group][NL lanes]).
// Built with tabulated reciprocals only (JobD fastdiv); batches that need IEEE division take the one-wavefront kernels.
#include "ps_sweep_body.h"

namespace ps {

// registers: K <= 5 rows per lane fit 168 (three wavefronts per SIMD), beyond that two
#define PS_SWEEPW_WPE(K) K <= 5 ? 3 : 2

template <int K, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(PS_SWEEPW_WPE(K), PS_SWEEPW_WPE(K))))
void k_sweep_w(BatchD b, SweepD sw) {
    __shared__ double hand[2 * NW * HAND_DOUBLES];
    __shared__ __attribute__((aligned(16))) char mring[mring_bytes(NW)];
    const JobD& J = b.jobs[blockIdx.x];
    if (J.out->inert) return;
    sweep_body<K, NW, 0, 0, true>(b, sw, J, sw.sj[blockIdx.x], nullptr, hand, mring);
}

template <int K, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(PS_SWEEPW_WPE(K), PS_SWEEPW_WPE(K))))
void k_sweep2_w(BatchD b, SweepD sw) {
    __shared__ unsigned long long ring[ring_cols(NW)];
    __shared__ double hand[2 * NW * HAND_DOUBLES];
    __shared__ __attribute__((aligned(16))) char mring[mring_bytes(NW)];
    const int jd = blockIdx.x;
    const JobD& J = b.jobs[jd >> 1];
    if (J.out->inert) return;
    if ((jd & 1) == 0) sweep_body<K, NW, 0, 1, true>(b, sw, J, sw.sj[jd], ring, hand, mring);
    else sweep_body<K, NW, 1, 1, true>(b, sw, J, sw.sj[jd], ring, hand, mring);
}

template <int K, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(PS_SWEEPW_WPE(K), PS_SWEEPW_WPE(K))))
void k_sweeps_w(BatchD b, SweepD sw) {
    __shared__ unsigned long long ring[ring_cols(NW)];
    __shared__ double hand[2 * NW * HAND_DOUBLES];
    __shared__ __attribute__((aligned(16))) char mring[mring_bytes(NW)];
    const int jd = blockIdx.x;
    const JobD& J = b.jobs[jd >> 1];
    if (J.out->inert) return;
    if ((jd & 1) == 0) sweep_body<K, NW, 0, 2, true>(b, sw, J, sw.sj[jd], ring, hand, mring);
    else sweep_body<K, NW, 1, 2, true>(b, sw, J, sw.sj[jd], ring, hand, mring);
}

template <int K, int NW>
static void launch_w(Runtime* rt, const BatchD& b, const SweepD& sw) {
    // PORESEQ_SWEEP_LDS_PAD_KB (tuning): dynamic LDS a sweep's workgroup claims on top of its own, i.e. a cap on the sweeps resident per CU
    // (160 KB of LDS per CU) that leaves registers and wave slots to the short kernels of the other batches' chains
    static const size_t pad = getenv("PORESEQ_SWEEP_LDS_PAD_KB") ? (size_t)atoi(getenv("PORESEQ_SWEEP_LDS_PAD_KB")) * 1024 : 0;
    if (sw.ndir == 2 && sw.sparse) hipLaunchKernelGGL((k_sweeps_w<K, NW>), dim3(b.njobs * 2), dim3(64 * NW), pad, rt->stream, b, sw);
    else if (sw.ndir == 2) hipLaunchKernelGGL((k_sweep2_w<K, NW>), dim3(b.njobs * 2), dim3(64 * NW), pad, rt->stream, b, sw);
    else hipLaunchKernelGGL((k_sweep_w<K, NW>), dim3(b.njobs), dim3(64 * NW), pad, rt->stream, b, sw);
}

bool sweepw_launch(Runtime* rt, const BatchD& b, const SweepD& sw, int K, int NW) {
    switch (NW * 100 + K) {
        case 204: launch_w<4, 2>(rt, b, sw); return true;
        case 205: launch_w<5, 2>(rt, b, sw); return true;
        case 206: launch_w<6, 2>(rt, b, sw); return true;
        case 210: launch_w<10, 2>(rt, b, sw); return true;
        case 402: launch_w<2, 4>(rt, b, sw); return true;
        case 403: launch_w<3, 4>(rt, b, sw); return true;
        case 404: launch_w<4, 4>(rt, b, sw); return true;
        case 406: launch_w<6, 4>(rt, b, sw); return true;
    }
    return false;
}

}  // namespace ps
