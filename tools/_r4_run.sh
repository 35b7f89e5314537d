mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "swin or linear" > gpurun_out/r4/swin_tests.log 2>&1
echo "tests rc $?"; tail -n 5 gpurun_out/r4/swin_tests.log
