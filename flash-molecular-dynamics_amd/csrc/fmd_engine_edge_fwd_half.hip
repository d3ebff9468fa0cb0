// W16A16 instantiations of the forward pair kernel: fmd_engine_edge_fwd.hip compiled with the packed pair type
// (see the note at the top of that file); entry points setup_edge_fwd_attrs_half / launch_edge_fwd_half.
#define FMD_HALF_TU 1
#include "fmd_engine_edge_fwd.hip"
