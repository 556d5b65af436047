#!/bin/bash
# The driver's 20-batch call under rocprofv3 --kernel-trace (+ memory copies are not traced: kernel dispatches only): every
# dispatch of the call with its duration and the gap in front of it (tools/trace_gaps.py).  GPU box.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$ROOT/gpurun_out/trace_call20; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $ROOT/tools/time_flow.py --steps 20 --flow -1 --reps 40 > $OUT/run.log 2>&1
cd $ROOT && python tools/trace_gaps.py $OUT
