cd /root/repo
echo "=== default lib"; timeout 120 python tools/r05/dbg2.py onepass 5000 10000 0.01 2>&1 | grep -v "amdgpu.ids" | tail -3
echo "=== nomap variant"; SPAMD_LIB=/root/repo/sparse_amd/_lib/variants/libsparse_amd_nomap.so timeout 120 python tools/r05/dbg2.py onepass 5000 10000 0.01 2>&1 | grep -v "amdgpu.ids" | tail -3
