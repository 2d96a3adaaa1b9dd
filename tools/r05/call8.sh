cd /root/repo
for c in "onepass 5000 10000 0.01" "twopass 5000 10000 0.01" "sort 5000 10000 0.01" "onepass 700 9920 0.02" "onepass 700 9760 0.01"; do
  echo "=== $c"; timeout 120 python tools/r05/dbg2.py $c 2>&1 | grep -v "amdgpu.ids" | tail -7
done
