export TMPDIR=/tmp
GRAPH=1 N=8 timeout -k 10 300 python scripts/probe/fm_time.py 2>&1 | grep "ms/step\|rror\|loss"
