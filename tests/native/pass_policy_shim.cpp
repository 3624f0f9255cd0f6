// C handle around bisbm::PassDepthPolicy (csrc/bisbm_pass_policy.hpp, host only) for tests/test_pass_policy.py: the selector
// that picks the pass depth of every sweep launch is driven here with made-up (speed, accepted fraction) sequences, no GPU.
#include "../../bipartitesbm-mcmc_amd/csrc/bisbm_pass_policy.hpp"

extern "C" {
void* pp_new() { return new bisbm::PassDepthPolicy(); }
void pp_free(void* p) { delete (bisbm::PassDepthPolicy*)p; }
void pp_reset(void* p) { ((bisbm::PassDepthPolicy*)p)->reset(); }
unsigned pp_choose(void* p, unsigned max_depth, int small_graph) { return ((bisbm::PassDepthPolicy*)p)->choose(max_depth, small_graph != 0); }
void pp_record(void* p, unsigned depth, double speed, double acc) { ((bisbm::PassDepthPolicy*)p)->record(depth, speed, acc); }
unsigned pp_current(void* p) { return ((bisbm::PassDepthPolicy*)p)->current(); }
unsigned pp_switches(void* p) { return ((bisbm::PassDepthPolicy*)p)->switches(); }
unsigned pp_looks(void* p) { return ((bisbm::PassDepthPolicy*)p)->looks(); }
int pp_settled(void* p) { return ((bisbm::PassDepthPolicy*)p)->settled() ? 1 : 0; }
double pp_figure(void* p, unsigned d) { return ((bisbm::PassDepthPolicy*)p)->figure(d); }
}
