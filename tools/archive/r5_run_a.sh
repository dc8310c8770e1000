#!/bin/bash
# Round 5, GPU session A: parity first.  (1) the skip-path bisect of VERDICT r4 item 1a, (2) the whole GPU suite WITHOUT -x in the new
# collection order with the mask-frozen network parity and its per-tensor log, (3) the Winograd route's first hardware session.
o=gpurun_out/r5a; mkdir -p $o
python tools/bisect_skip.py > $o/bisect_skip.json 2> $o/bisect_skip.err; tail -3 $o/bisect_skip.err
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $o/gpu_tests.log; tail -15 $o/gpu_tests.log
cp gpurun_out/network_parity*.jsonl $o/ 2>/dev/null
timeout 900 bash tools/r5_first_session.sh > $o/winograd_session.txt 2>&1; tail -30 $o/winograd_session.txt
echo SESSION_A_DONE
