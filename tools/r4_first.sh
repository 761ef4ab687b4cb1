#!/bin/bash
# round 4, first GPU call: the segmented tail against the oracle, then the headline loop under a few segmentations
mkdir -p gpurun_out
O=gpurun_out/r4_first.txt
: > $O
timeout 900 python -m pytest tests/test_stream_segments.py tests/test_capture_stream.py -x -q -m gpu > gpurun_out/r4_first_pytest.txt 2>&1
echo "pytest rc=$?" >> $O
tail -30 gpurun_out/r4_first_pytest.txt >> $O
show='
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); c = d["config"]
        print("ms/step", d["ms_per_step"], "kernel", d["roofline"].get("kernel_ms"), "one capture", c.get("single_capture_incl_compact_d2h_ms"),
              "with pos", c.get("ms_per_step_with_device_positions"), "device only", c.get("device_only_ms_per_step"), "host", c.get("host_loop"), c.get("stream_stats"))'
F="--no-cpu-baseline --no-extra --no-reference-loop --steps 20 --warmup 3"
for cfg in "1 0" "8 0" "4 0" "12 0" "6 1" "8 1"; do
  set -- $cfg
  echo "== segments $1 shape $2" >> $O
  URH_STREAM_SEGMENTS=$1 URH_STREAM_SHAPE=$2 timeout 300 python bench.py $F 2>gpurun_out/r4_first_err_$1_$2.txt | python -c "$show" >> $O
  tail -3 gpurun_out/r4_first_err_$1_$2.txt >> $O
done
cat $O
