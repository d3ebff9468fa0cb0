export TMPDIR=/tmp
bash tools/profile_round.sh r05_a > gpurun_out/prof_r05_a.log 2>&1; echo "profile_round fp32 rc=$?"
bash tools/profile_round.sh r05_a_w16a16 "--precision w16a16" > gpurun_out/prof_r05_a_w16.log 2>&1; echo "profile_round w16a16 rc=$?"
bash tools/profile_sq.sh r05_a > gpurun_out/sq_r05_a.log 2>&1; echo "profile_sq fp32 rc=$?"
bash tools/profile_sq.sh r05_a_w16a16 --precision w16a16 > gpurun_out/sq_r05_a_w16.log 2>&1; echo "profile_sq w16a16 rc=$?"
for args in "" "--fwd" "--precision w16a16" "--precision w16a16 --fwd" "--block 0"; do
  echo "== $args"
  timeout -k 10 120 python tools/phase_stamps.py $args 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05_phase_stamps.txt 2>&1
echo stamps done
