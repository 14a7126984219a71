#!/bin/bash
# sample power / sclk while a kernel loop runs
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/smi_$1.log &
SMI=$!
shift
"$@"
wait $SMI
