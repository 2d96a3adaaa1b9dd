cd /root/repo
tools/micro/read_pattern.bin | head -4
bash tools/gpu_job.sh variants tools/r06/stream_abl.py abl1 abl2 abl3 abl4 "" cl 2>&1 | grep -v amdgpu
tools/micro/read_pattern.bin | head -4
