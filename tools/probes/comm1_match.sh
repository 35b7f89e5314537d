# the strict tracking stream with a REAL 1-rank RCCL communicator, cost / update stages on the compute stream (0) or on the match stream (2): bash tools/probes/comm1_match.sh
for ms in 0 2; do
  REID_BENCH_COMM1=1 timeout -k 10 300 python bench.py --workload tracking --no-cpu --cameras 0 --match-stream $ms 2>/dev/null > gpurun_out/r6/c1_$ms.json
  python -c "import json; d=json.load(open('gpurun_out/r6/c1_$ms.json')); print('comm1 match_stream $ms', d['value'], d['ms_per_step'])"
done
