#!/bin/bash
mkdir -p gpurun_out
for cfg in "1 8 4" "4 8 4" "5 8 4" "4 8 2" "5 8 2" "4 8 3"; do
  set -- $cfg
  GITMI_PART_DBG=$1 timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --decode-cus $2 --decode-streams $3 2>/dev/null | tail -1 \
    | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('dbg=$1 decode_cus=%s streams=%s value=%.0f ms/batch=%.3f lat_med=%.1f' % (c.get('decode_cus'), c.get('decode_streams'), d['value'], d['ms_per_step'], d['batch_latency_ms']['median']))
" 2>&1 | tee -a gpurun_out/cupart.txt
done
