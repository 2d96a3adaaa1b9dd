cd /root/repo
for c in "nat 5000 10000 0.01" "nat 333 160 0.3" "nat 1000 3000 0.01" "bal 30011 2000 3000000" "bal 80000 3000 4000000" "bald 80000 3000 4000000"; do
  echo "=== $c"; timeout 120 python tools/r05/dbg_tiled.py $c 2>&1 | grep -v "amdgpu.ids" | tail -6
done
for c in "onepass 5000 10000 0.01" "twopass 5000 10000 0.01" "sort 5000 10000 0.01" "onepass 700 9920 0.02" "onepass 700 9760 0.01"; do
  echo "=== $c"; timeout 120 python tools/r05/dbg2.py $c 2>&1 | grep -v "amdgpu.ids" | tail -7
done
