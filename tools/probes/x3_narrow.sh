mkdir -p gpurun_out/r5
for sw in "x3_narrow=1" "x3_narrow=0" "x3_narrow=1" "x3_narrow=0" "x3_narrow=3"; do
  REID_DEBUG_SWITCHES=$sw timeout -k 5 120 python tools/time_pass.py 2 1024 2>&1 | tail -1
done > gpurun_out/r5/x3_narrow2.txt 2>&1
cat gpurun_out/r5/x3_narrow2.txt
