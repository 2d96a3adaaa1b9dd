cd /root/repo
for i in 1 2 3; do echo "== nomap run $i"; SPAMD_LIB=/root/repo/sparse_amd/_lib/variants/libsparse_amd_nomap.so timeout 120 python tools/r05/dbg3.py /root/repo 30 2>&1 | grep -v amdgpu.ids | tail -2; done
