// W16A16 instantiations of the backward pair kernel: fmd_engine_edge_bwd.hip compiled with the packed pair type
// (see the note at the top of the forward kernel's file); entry points setup_edge_bwd_attrs_half / launch_edge_bwd_half.
#define FMD_HALF_TU 1
#include "fmd_engine_edge_bwd.hip"
