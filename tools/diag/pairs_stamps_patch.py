"""Throw-away build of pairs_kernel with wall-clock stamps at its phase boundaries (tools/diag/pairs_stamps.py reads them):
  cp csrc/fmd_engine_pairs.hip /tmp/keep.hip; python tools/diag/pairs_stamps_patch.py; make -C flash-molecular-dynamics_amd/csrc;
  cp flash-molecular-dynamics_amd/flashmd_amd/libflashmd_hip.so gpurun_ab/lib_pairs_stamps.so; cp /tmp/keep.hip csrc/fmd_engine_pairs.hip; make ...
The patched source is never committed and the stamped library never replaces the product library on the GPU box."""
import sys
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'flash-molecular-dynamics_amd', 'csrc', 'fmd_engine_pairs.hip')
s=open(p).read()
s=s.replace("namespace {\n// ------","__device__ long long g_dbg[512 * 16];\nextern \"C\" int fmd_debug_pairs_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(long long) * 512 * 16); }\nnamespace {\n// ------",1)
s=s.replace("#pragma clang fp contract(off)\n  extern __shared__","#pragma clang fp contract(off)\n  int st_k = 0;\n#define ST() { if (threadIdx.x == 0) g_dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + st_k] = wall_clock64(); ++st_k; }\n  ST()\n  extern __shared__",1)
marks=[
 ("  __syncthreads();\n  auto dist2 = [&](int i, int j) {","  __syncthreads();\n  ST()\n  auto dist2 = [&](int i, int j) {"),
 ("  __syncthreads();\n  // ---- orientation:","  __syncthreads();\n  ST()\n  // ---- orientation:"),
 ("  // ---- which rows may share an 8-entry segment","  ST()\n  // ---- which rows may share an 8-entry segment"),
 ("  __syncthreads();\n  // ---- offsets:","  __syncthreads();\n  ST()\n  // ---- offsets:"),
 ("  __syncthreads();\n  const int total = tot_s[0];","  __syncthreads();\n  ST()\n  const int total = tot_s[0];"),
 ("  // ---- pass 2: fill from the masks (sorted by j)","  ST()\n  // ---- pass 2: fill from the masks (sorted by j)"),
 ("  __syncthreads();  // (workgroup-scope release/acquire of the entries written above)","  __syncthreads();  // (workgroup-scope release/acquire of the entries written above)\n  ST()"),
 ("  // fill the last 32-entry tile with null entries","  __syncthreads();\n  ST()\n  // fill the last 32-entry tile with null entries"),
]
for a,b in marks:
    assert a in s, a
    s=s.replace(a,b,1)
open(p,'w').write(s)
