export TMPDIR=/tmp
python scripts/probe/h2d_ab.py 2>/dev/null | tail -1
HSA_ENABLE_SDMA=0 python scripts/probe/h2d_ab.py 2>/dev/null | tail -1
SPL=32 python scripts/probe/h2d_ab.py 2>/dev/null | tail -1
SPL=64 python scripts/probe/h2d_ab.py 2>/dev/null | tail -1
