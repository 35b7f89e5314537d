ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
N=${1:-30}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/sb -o p -- python3 $ROOT/tools/profile_small_batch.py $N f16x3 > $OUT/sb.log 2>&1
tail -1 $OUT/sb.log
python3 $ROOT/tools/timeline.py $OUT/sb/p_results.db stem_split 3 2>&1 | tail -40
rm -rf $OUT/sb
