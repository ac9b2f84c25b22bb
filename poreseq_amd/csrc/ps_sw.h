// ps_sw.h — host-visible types of the Smith-Waterman module (ps_sw.hip)
#ifndef PS_SW_H_
#define PS_SW_H_
#include "ps_internal.h"

namespace ps {

struct SwPair {
    int n1, n2;                  // rows (first sequence), columns (second sequence)
    int nrb;                     // row blocks of 64
    int ngw;                     // wave strips (16 per super-strip)
    int pitch;                   // ints per saved row
    int pad;
    int64_t s1_off, s2_off;      // into the character pool
    int64_t row_off;             // rowsave: nrb x pitch ints, row q = H(64q, 1..n2) at [0..n2)
    int64_t col_off;             // colsave: (n2/64 + 1) x (n1 + 1) ints, entry c = H(0..n1, 64c)  (c = 0 unused: zeros)
    int64_t blk_off;             // blkmax: nrb x ngw ints
    int64_t out_off;             // 2 * (n1 + n2 + 2) ints: index pairs in walk order
    int64_t res_off;             // 8 ints: score, bi, bj, npairs, nmatch
};

struct SwResult { int score = 0; double accuracy = 0; std::vector<int> a, b; };

// an enqueued batch: host staging stays alive until sw_finish
struct SwJob {
    std::vector<SwPair> pairs;
    std::string pool;
    int* res = nullptr;        // pinned host staging (runtime-owned): results and index pairs
    int* outbuf = nullptr;
    int64_t out_tot = 0;
    double cells = 0;
    int np = 0;
    hipStream_t stream = nullptr;   // where the batch was enqueued
};

typedef std::vector<std::pair<const std::string*, const std::string*>> SwInput;
int sw_launch(Runtime* rt, const SwInput& in, SwJob* job);   // asynchronous, on the runtime's second stream (or its main one under load)
int sw_finish(Runtime* rt, SwJob* job, std::vector<SwResult>* out);
int sw_batch(Runtime* rt, const SwInput& in, std::vector<SwResult>* out);

}  // namespace ps
#endif
