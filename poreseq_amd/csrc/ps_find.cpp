#include "ps_host.h"
namespace ps {
int find_mutations(Runtime*, Align*, const std::vector<std::string>&, std::vector<Mut>*) { return fail(PS_ERR_UNSUPPORTED, "find_mutations: not built yet"); }
int viterbi_mutate(Runtime*, Align*, int, double, double, double, double, std::vector<std::string>*) { return fail(PS_ERR_UNSUPPORTED, "viterbi_mutate: not built yet"); }
}
