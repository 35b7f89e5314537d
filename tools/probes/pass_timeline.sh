# every launch of one 1024-crop fp32-class pass: bash tools/probes/pass_timeline.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/pt -o p -- python3 $ROOT/tools/time_pass.py 2 1024 > $OUT/pt.log 2>&1
python3 $ROOT/tools/timeline.py $OUT/pt/p_results.db stem_split 3 > $OUT/pass_timeline_1024.txt 2>&1
rm -rf $OUT/pt
cat $OUT/pass_timeline_1024.txt
