timeout 600 python tools/lab/edges_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
