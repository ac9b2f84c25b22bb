#include "ps_internal.h"
namespace ps {
int viterbi_device(Runtime*, int, int, const double*, const double*, int, double, double, double, double, const double*, std::vector<std::vector<int>>*) {
    return fail(PS_ERR_UNSUPPORTED, "viterbi: not built yet");
}
}
