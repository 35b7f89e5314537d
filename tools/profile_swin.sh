#!/bin/bash
# On the GPU box: per-kernel summary of the Swin-T workload (bench.py --workload swin), written to gpurun_out/<tag>_swin_<mode>_kernel_stats.csv
set -e
TAG=${1:-r02}
MODE=${2:-f16}
N=${3:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_swintrace_$MODE -o p -- python3 $ROOT/bench.py --workload swin --crops $N --steps 2 --warmup 1 --no-cpu --single --precision $MODE > $OUT/${TAG}_swin_$MODE.json 2> $OUT/${TAG}_swin_$MODE.err
python3 $ROOT/tools/rocprof_summary.py $OUT/${TAG}_swintrace_$MODE/p_results.db 30 > $OUT/${TAG}_swin_${MODE}_kernel_stats.csv
rm -rf $OUT/${TAG}_swintrace_$MODE
cat $OUT/${TAG}_swin_$MODE.json
cat $OUT/${TAG}_swin_${MODE}_kernel_stats.csv
