export TMPDIR=/tmp
mkdir -p gpurun_out/fm
timeout -k 10 900 python -m pytest tests/test_gpu_train_gan.py -m gpu -q > gpurun_out/fm/t_tg.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/fm/t_tg.log
