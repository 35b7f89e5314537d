# per-kernel totals of the strict tracking stream (600 frames + warm-up + profiled repeat): bash tools/probes/tracking_stats.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trk -o p -- python3 $ROOT/bench.py --workload tracking --single --no-cpu --cameras 0 > $OUT/trk.json 2> $OUT/trk.err
python3 $ROOT/tools/rocprof_summary.py $OUT/trk/p_results.db 30 > $OUT/tracking_kernel_stats.csv
rm -rf $OUT/trk
cat $OUT/tracking_kernel_stats.csv
