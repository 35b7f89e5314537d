# wait / issue / matrix-pipe counters per kernel of a 512-image Swin pass in the fp32-class mode (PMC pass of its own): bash tools/probes/pmc_waits_swin.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pws -o p -- python3 $ROOT/bench.py --workload swin --crops 512 --steps 1 --warmup 1 --no-cpu --single --precision f16x3 > $OUT/pws.log 2>&1
python3 $ROOT/tools/pmc_waits.py $OUT/pws/p_results.db > $OUT/pmc_waits_swin.txt 2>&1
rm -rf $OUT/pws
cat $OUT/pmc_waits_swin.txt
