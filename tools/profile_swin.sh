#!/bin/bash
# On the GPU box: per-kernel summary + HBM-traffic PMC passes of the Swin-T workload (bench.py --workload swin), written to
# gpurun_out/<tag>_swin_<mode>_kernel_stats.csv and gpurun_out/<tag>_traffic_swin_<mode>.json   (bash tools/profile_swin.sh r03 f16x3)
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
# HBM-side traffic of the contraction class: counters in passes of their own (one step of the same pass size, so that a launch
# is the launch bench.py prices)
PCMD="python3 $ROOT/bench.py --workload swin --crops $N --steps 1 --warmup 1 --no-cpu --single --precision $MODE"
export REID_PROFILED_COMMAND="python3 bench.py --workload swin --crops $N --steps 1 --warmup 1 --no-cpu --single --precision $MODE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_swin_fetch_$MODE -o p -- $PCMD > /dev/null 2> $OUT/${TAG}_swin_fetch_$MODE.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_swin_write_$MODE -o p -- $PCMD > /dev/null 2> $OUT/${TAG}_swin_write_$MODE.err
python3 $ROOT/tools/pmc_traffic.py $OUT/${TAG}_swin_fetch_$MODE/p_results.db $OUT/${TAG}_swin_write_$MODE/p_results.db swin_$MODE > $OUT/${TAG}_traffic_swin_$MODE.json
rm -rf $OUT/${TAG}_swin_fetch_$MODE $OUT/${TAG}_swin_write_$MODE
cat $OUT/${TAG}_swin_$MODE.json
cat $OUT/${TAG}_swin_${MODE}_kernel_stats.csv
cat $OUT/${TAG}_traffic_swin_$MODE.json
