mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r2h/t_final.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r2h/t_final.log
( time python bench.py > gpurun_out/r2h/bench_final.json 2> gpurun_out/r2h/bench_final.err ) 2> gpurun_out/r2h/bench_time.txt; echo "bench rc=$?"; grep real gpurun_out/r2h/bench_time.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2h/bench_driverlike.json 2> gpurun_out/r2h/bench_driverlike.err ) 2> gpurun_out/r2h/bench_time2.txt; echo "bench rc=$?"; grep real gpurun_out/r2h/bench_time2.txt
