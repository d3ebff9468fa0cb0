python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for args in "--batch 128 --beads 269" "--batch 128 --beads 269 --num-rbf 100" "--batch 128 --beads 269 --num-rbf 128" "--batch 128 --beads 269 --num-rbf 100 --precision w16a16" "--batch 1024 --beads 35" "--batch 2048 --beads 10" "--batch 512 --beads 76"; do
  echo "== slot_times $args"
  timeout -k 10 120 python tools/slot_times.py $args 2>&1 | grep -v amdgpu.ids
done
