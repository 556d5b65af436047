#!/bin/bash
# The driver's 20-batch call with the tables' way home inside the loss reduction launch (default) and as a launch of its own
# (option no_call_tail): tools/time_flow.py reads CREC_OPTS-free, so the option goes through the environment of this script.
cd "$(dirname "$0")/.." || exit 1
for v in 0 1 0 1; do echo "== no_call_tail=$v"; CREC_NO_CALL_TAIL=$v python tools/time_flow.py --steps 20 --flow -1 --reps 400 2>&1 | grep -v amdgpu.ids; done
