#!/bin/bash
# time split of the FAST passes: stage ms of the FAST kernel with the later passes cut off (ORBSLAM_AMD_FAST_STOP; results invalid)
for s in 9 1 2 3 0; do
  ORBSLAM_AMD_FAST_STOP=$s ORBSLAM_AMD_NO_OVERLAP=1 python bench.py --split 1 --steps 3 --warmup 1 --batches-per-step 2 --cpu-frames 0 --ba-windows 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stop_after=$s', 'fast ms', d['stage_ms']['fast'], 'fps', d['value'])"
done
