// placeholder until the Smith-Waterman kernels land
#include "ps_internal.h"
namespace ps {
int sw_device(Runtime*, const std::string&, const std::string&, int*, double*, std::vector<int>*, std::vector<int>*) {
    return fail(PS_ERR_UNSUPPORTED, "swfull: not built yet");
}
}
