# per-kernel times of a fp32-class pass of N crops: bash tools/probes/prof_n.sh N
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
N=${1:-256}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/pn -o p -- python3 $ROOT/tools/time_pass.py 2 $N > $OUT/pn.log 2>&1
python3 $ROOT/tools/rocprof_summary.py $OUT/pn/p_results.db 14 > $OUT/prof_$N.csv
rm -rf $OUT/pn
tail -1 $OUT/pn.log
cat $OUT/prof_$N.csv
