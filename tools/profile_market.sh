#!/bin/bash
# On the GPU box: kernel summary + HBM-traffic PMC passes of the Market-size workload (bench.py --workload market): the shard's
# distance matrix (gemm_f32_dma_kernel<E_DIST>) and the fused search (dist_select_kernel), written to gpurun_out/<tag>_*.
#   bash tools/profile_market.sh r03
set -e
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
CMD="python3 $ROOT/bench.py --workload market --steps 2 --warmup 1 --no-cpu"
export REID_PROFILED_COMMAND="python3 bench.py --workload market --steps 2 --warmup 1 --no-cpu"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_mk_trace -o p -- $CMD > $OUT/${TAG}_mk_trace.json 2> $OUT/${TAG}_mk_trace.err
python3 $ROOT/tools/rocprof_by_grid.py $OUT/${TAG}_mk_trace/p_results.db > $OUT/${TAG}_market_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_mk_fetch -o p -- $CMD > /dev/null 2> $OUT/${TAG}_mk_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_mk_write -o p -- $CMD > /dev/null 2> $OUT/${TAG}_mk_write.err
python3 $ROOT/tools/pmc_traffic.py $OUT/${TAG}_mk_fetch/p_results.db $OUT/${TAG}_mk_write/p_results.db market > $OUT/${TAG}_traffic_market.json
python3 $ROOT/tools/pmc_traffic.py $OUT/${TAG}_mk_fetch/p_results.db $OUT/${TAG}_mk_write/p_results.db select > $OUT/${TAG}_traffic_select.json
rm -rf $OUT/${TAG}_mk_trace $OUT/${TAG}_mk_fetch $OUT/${TAG}_mk_write
cat $OUT/${TAG}_market_kernel_stats.csv
cat $OUT/${TAG}_traffic_market.json $OUT/${TAG}_traffic_select.json
