# wait / issue / matrix-pipe counters per kernel of a 30-crop fp32-class pass (a tracking frame): bash tools/probes/pmc_waits_small.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pwsm -o p -- python3 $ROOT/tools/time_pass.py 2 30 > $OUT/pwsm.log 2>&1
python3 $ROOT/tools/pmc_waits.py $OUT/pwsm/p_results.db > $OUT/pmc_waits_small.txt 2>&1
rm -rf $OUT/pwsm
cat $OUT/pmc_waits_small.txt
