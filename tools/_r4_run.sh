mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "swin or linear" > gpurun_out/r4/swin_tests.log 2>&1
echo "tests rc $?"; tail -n 5 gpurun_out/r4/swin_tests.log
for v in 0 1 0 1; do
REID_SWIN_TWO_LINEAR=$v timeout -k 10 300 python bench.py --workload swin --crops 1024 --steps 3 --warmup 1 --no-cpu --single --precision f16x3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('two_linear=$v', d['value'], d['ms_per_step'])"
done
