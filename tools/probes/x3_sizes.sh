mkdir -p gpurun_out/r5
{
for n in 256 512; do
  for mb in 512 256 128; do
    REID_DEBUG_SWITCHES=split_x3_min_blocks=$mb timeout -k 5 120 python tools/time_pass.py 2 $n 2>&1 | tail -1
  done
done
for n in 2048 4096; do
  timeout -k 5 120 python tools/time_pass.py 2 $n 2>&1 | tail -1
done
} > gpurun_out/r5/x3_sizes.txt 2>&1
cat gpurun_out/r5/x3_sizes.txt
