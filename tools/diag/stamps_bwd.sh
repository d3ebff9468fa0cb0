# phase stamps of the backward kernels, consumers' C-A window: dbg 8 = no shared Gaussians, dbg 16 = no gW formation (timing only)
export FMD_STAMPS_LIB=gpurun_ab/stamps.so
for a in "--precision w16a16" "--precision w16a16 --dbg 8" "--precision w16a16 --dbg 16" "--precision w16a16 --dbg 24" "" "--dbg 8" "--dbg 16" "--dbg 24"; do echo "== $a"; timeout -k 10 200 python tools/phase_stamps.py $a 2>&1 | grep -v amdgpu.ids; done
