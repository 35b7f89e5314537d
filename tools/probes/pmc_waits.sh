# wait / issue / matrix-pipe counters per kernel of a 1024-crop fp32-class pass (PMC pass of its own): bash tools/probes/pmc_waits.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pw -o p -- python3 $ROOT/tools/time_pass.py 2 1024 > $OUT/pw.log 2>&1
python3 $ROOT/tools/pmc_waits.py $OUT/pw/p_results.db > $OUT/pmc_waits.txt 2>&1
rm -rf $OUT/pw
cat $OUT/pmc_waits.txt
