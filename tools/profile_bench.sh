#!/bin/bash
# On the GPU box: kernel-trace summary + HBM-traffic PMC passes of the default bench command, written to gpurun_out/<tag>_*.
#   bash tools/profile_bench.sh r02 f32     (then copy what should be judged into profiles/)
# rocprofv3 gets the program itself after "--" (python3 bench.py ...), counters in their own passes (no trace domains besides
# --kernel-trace), as the run instructions of this pool require.
set -e
TAG=${1:-r02}
MODE=${2:-f32}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace_$MODE -o p -- python3 $ROOT/bench.py --workload embed --steps 3 --warmup 1 --no-cpu --single --precision $MODE > $OUT/${TAG}_trace_$MODE.json 2> $OUT/${TAG}_trace_$MODE.err
python3 $ROOT/tools/rocprof_summary.py $OUT/${TAG}_trace_$MODE/p_results.db 40 > $OUT/${TAG}_bench_${MODE}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch_$MODE -o p -- python3 $ROOT/bench.py --workload embed --steps 1 --warmup 1 --no-cpu --single --precision $MODE > /dev/null 2> $OUT/${TAG}_pmc_fetch_$MODE.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write_$MODE -o p -- python3 $ROOT/bench.py --workload embed --steps 1 --warmup 1 --no-cpu --single --precision $MODE > /dev/null 2> $OUT/${TAG}_pmc_write_$MODE.err
export REID_PROFILED_COMMAND="python3 bench.py --workload embed --steps 1 --warmup 1 --no-cpu --single --precision $MODE"
python3 $ROOT/tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch_$MODE/p_results.db $OUT/${TAG}_pmc_write_$MODE/p_results.db $MODE > $OUT/${TAG}_traffic_conv_$MODE.json
# the raw databases are large: keep the summaries only
rm -rf $OUT/${TAG}_trace_$MODE $OUT/${TAG}_pmc_fetch_$MODE $OUT/${TAG}_pmc_write_$MODE
cat $OUT/${TAG}_bench_${MODE}_kernel_stats.csv
cat $OUT/${TAG}_traffic_conv_$MODE.json
