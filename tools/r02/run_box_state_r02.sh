#!/bin/bash
# Box-state diagnostic: clocks and copy time as the first GPU process, after a second process, after GPU tests.
set -e
mkdir -p gpurun_out/r02
O=gpurun_out/r02/box_state.txt
: > $O
(rocm-smi --showclocks --showperflevel --showpower --showmemuse 2>&1 || true) >> $O
PROBE_LABEL=first python3 tools/box_state_probe.py >> $O 2>&1
PROBE_LABEL=second python3 tools/box_state_probe.py >> $O 2>&1
python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q >> $O 2>&1
PROBE_LABEL=after_tests python3 tools/box_state_probe.py >> $O 2>&1
PROBE_LABEL=after_tests2 PROBE_ALLOCS=12 python3 tools/box_state_probe.py >> $O 2>&1
(rocm-smi --showclocks --showperflevel --showpower --showmemuse 2>&1 || true) >> $O
echo done
