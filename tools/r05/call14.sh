cd /root/repo
for i in 1 2 3; do echo "== g32 run $i"; timeout 120 python tools/r05/dbg3.py /root/repo 40 2>&1 | grep -v amdgpu.ids | tail -2; done
for c in "bal 80000 3000 4000000" "bald 80000 3000 4000000" "bal 30011 2000 3000000"; do echo "=== $c"; for i in 1 2 3; do timeout 120 python tools/r05/dbg_tiled.py $c 2>&1 | grep -v "amdgpu.ids" | tail -1; done; done
