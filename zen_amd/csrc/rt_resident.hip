// rt_resident.hip -- the fused causal kernel as a RESIDENT kernel for the per-hop realtime API (opt-in:
// zen_hip_hpr_set_resident): one workgroup stays on its CU between the hops of a stream and is handed each hop through a
// mailbox, instead of one launch per hop (HPRRealtime<GPU>::process_next_hop, libzen/hps.cu:334-339; the loop the
// reference times in zen/fakert.h:221-247).  Same source as rt_fused.hip (the body of a hop is rt_fused_body); a
// translation unit of its own so that the per-launch builds keep their register allocation.
#define ZEN_RT_RESIDENT 1
#include "rt_fused.hip"
