ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3 4 8 12 7 11 16 19 31 32 35; do
  export REID_DEBUG_SWITCHES=x3_ablate=$v
  rocprofv3 --kernel-trace --stats -d $OUT/x3_ab_$v -o p -- python3 $ROOT/tools/time_pass.py 2 1024 > $OUT/x3_ab_$v.log 2>&1
  echo "ablate $v: $(python3 $ROOT/tools/rocprof_summary.py $OUT/x3_ab_$v/p_results.db 16 | grep conv3x3_x3 | tr '\n' ' ')"
  rm -rf $OUT/x3_ab_$v $OUT/x3_ab_$v.log
done > $OUT/x3_ablate.txt 2>&1
cat $OUT/x3_ablate.txt
