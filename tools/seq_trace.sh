#!/bin/bash
# kernels of the last batch of a bench run in launch order with durations.  usage: tools/seq_trace.sh [bench args]
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python3 $ROOT/bench.py --no-cpu-baseline --no-extra-workloads --no-readback --no-parity-check "$@" > /tmp/seq.log 2>&1
python3 $ROOT/tools/trace_sequence.py "$(find /tmp/prof_seq -name '*kernel_trace.csv' | head -1)"
