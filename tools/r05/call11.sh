cd /root/repo
for i in 1 2 3; do echo "== HEAD tree run $i"; timeout 120 python tools/r05/dbg3.py /root/repo/build_variant/head_tree 30 2>&1 | grep -v amdgpu.ids | tail -2; done
for i in 1 2 3; do echo "== working tree run $i"; timeout 120 python tools/r05/dbg3.py /root/repo 30 2>&1 | grep -v amdgpu.ids | tail -2; done
