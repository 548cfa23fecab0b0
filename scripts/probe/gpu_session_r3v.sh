mkdir -p gpurun_out/r3v
timeout -k 10 300 python scripts/probe/syncbn_probe.py gpurun_out/r3v 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -70
