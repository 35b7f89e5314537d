# per-kernel totals of the Swin pass with the dense x3 kernel on / off: bash tools/probes/swin_ab_prof.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rocprofv3 --kernel-trace --stats -d $OUT/sab$v -o p -- python3 $ROOT/tools/swin_ab.py 512 lin_x3 $v > $OUT/sab$v.log 2>&1
  python3 $ROOT/tools/rocprof_summary.py $OUT/sab$v/p_results.db 2>/dev/null | head -14 > $OUT/swin_ab_$v.txt || true
  rm -rf $OUT/sab$v
  echo "lin_x3=$v"; cat $OUT/swin_ab_$v.txt
done
