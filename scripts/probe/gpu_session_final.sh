mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r2f/t_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2f/t_final.log
