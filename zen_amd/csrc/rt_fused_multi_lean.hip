// rt_fused_multi_lean.hip -- the fused causal kernel for several outputs with hard masks in the lean layout (nfft 4096,
// 47 taps, blocks of hops): masks kept as two bits per bin, one inverse transform per output from the spectrum
// registers.  Same source as rt_fused.hip; a translation unit of its own because this instantiation keeps all its
// values in registers under the default scheduler and spills three under the max-ILP strategy of rt_fused_multi.hip.
#define ZEN_RT_FUSED_MULTI_LEAN 1
#include "rt_fused.hip"
