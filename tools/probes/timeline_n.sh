# kernel timeline of one fp32-class forward of N crops (a batched frame time of K cameras is ~30 K): bash tools/probes/timeline_n.sh N [outdir]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${2:-r6}
N=${1:-120}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/tl -o p -- python3 $ROOT/tools/profile_small_batch.py $N f16x3 > $OUT/tl.log 2>&1
tail -1 $OUT/tl.log > $OUT/timeline_$N.txt
python3 $ROOT/tools/timeline.py $OUT/tl/p_results.db stem_split 3 >> $OUT/timeline_$N.txt 2>&1
rm -rf $OUT/tl
cat $OUT/timeline_$N.txt
