// rt_fused_multi.hip -- the fused causal kernel's instantiations for calls with several outputs (H|P|R subsets).
// Same source as rt_fused.hip, a translation unit of its own because the two sets want different scheduler flags
// (zen_amd/build.py: the register-pressure trackers that take the one-output block build from 19 to 5 spilled
// registers add 10 to the three-output one).
#define ZEN_RT_FUSED_MULTI 1
#include "rt_fused.hip"
