#!/bin/bash
# Samples GPU power and shader clock (rocm-smi, read-only) while a command runs: tools/power_trace.sh <outfile> <cmd...>
OUT=$1; shift
"$@" > /dev/null 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  rocm-smi -d 0 --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level" | tr '\n' ' ' >> $OUT
  echo >> $OUT
  sleep 0.15
done
wait $PID
