# per-kernel times of a 1024-crop fp32-class pass, old and new halo convolution: bash tools/probes/x3_prof.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 2 3; do
  export REID_DEBUG_SWITCHES=split_x3=$v
  rocprofv3 --kernel-trace --stats -d $OUT/x3_trace_$v -o p -- python3 $ROOT/tools/time_pass.py 2 1024 > $OUT/x3_trace_$v.log 2>&1
  python3 $ROOT/tools/rocprof_summary.py $OUT/x3_trace_$v/p_results.db 16 > $OUT/x3_stats_$v.csv
  rm -rf $OUT/x3_trace_$v
  cat $OUT/x3_stats_$v.csv
done
