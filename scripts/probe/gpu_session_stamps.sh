export TMPDIR=/tmp
WHICH=wa B=64 K=6 WARM=3 CH=8 timeout -k 10 300 python scripts/diag_stamps.py 2>&1 | grep -v "warning\|offsetof\|\^\|amdgpu.ids" | tail -26
WHICH=wb B=64 K=6 WARM=3 timeout -k 10 300 python scripts/diag_stamps.py 2>&1 | grep -v "warning\|offsetof\|\^\|amdgpu.ids" | tail -7
