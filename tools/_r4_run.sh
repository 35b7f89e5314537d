set -e
ROOT=$PWD
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/swt -o p -- python3 $ROOT/bench.py --workload swin --crops 1024 --steps 2 --warmup 1 --no-cpu --single --precision f16x3 > $ROOT/gpurun_out/r4/swin_tl.json 2> $ROOT/gpurun_out/r4/swin_tl.err
python3 $ROOT/tools/timeline.py /tmp/swt/p_results.db sfe_conv1 1 > $ROOT/gpurun_out/r4/swin_timeline3.txt
tail -n 1 $ROOT/gpurun_out/r4/swin_timeline3.txt
