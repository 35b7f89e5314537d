# timeline of one 30-crop forward, product form and with the split-K forms of conv3x3_x3.hip: bash tools/probes/small.sh [n]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
N=${1:-30}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sw in "split_x3_small=0" "split_x3_small=1" "split_x3_small=1,split_x3=3"; do
  export REID_DEBUG_SWITCHES=$sw
  python3 $ROOT/tools/x3_small_check.py $N
  rocprofv3 --kernel-trace -d $OUT/sb -o p -- python3 $ROOT/tools/profile_small_batch.py $N f16x3 > $OUT/sb.log 2>&1
  tail -1 $OUT/sb.log
  python3 $ROOT/tools/timeline.py $OUT/sb/p_results.db stem_split 3 2>&1 | tail -40
  rm -rf $OUT/sb
done > $OUT/small_$N.txt 2>&1
cat $OUT/small_$N.txt
