mkdir -p gpurun_out/r5
for sw in "" "f16_loader_waves=0" "f16_loader_prio=0" "f16_frag_ahead=0" "split_pair=1" "split_pair=2" ""; do
  REID_DEBUG_SWITCHES=$sw timeout -k 5 120 python tools/time_pass.py 2 1024 2>&1 | tail -1
done > gpurun_out/r5/x1.txt 2>&1
cat gpurun_out/r5/x1.txt
