#!/bin/bash
# Socket power and shader clock while a command runs: samples `rocm-smi --showpower --showclocks` in the background
# (one call per ~0.2 s) -> gpurun_out/<tag>_power.txt.   tools/power_sample.sh TAG -- command ...
set -u; mkdir -p gpurun_out; TAG=$1; shift; [ "$1" = "--" ] && shift
OUT=gpurun_out/${TAG}_power.txt; : > $OUT
( while true; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ' >> $OUT; echo >> $OUT
  done ) &
S=$!
sleep 1.5
echo "## command starts" >> $OUT
"$@"
RC=$?
echo "## command ends rc=$RC" >> $OUT
sleep 1.0
kill $S 2>/dev/null; wait $S 2>/dev/null
exit $RC
