#!/bin/bash
# Samples socket power and shader clock (rocm-smi, unprivileged) while a command runs on the GPU box.
#   tools/power_clock_sampler.sh <out.txt> <command...>
out=$1; shift
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E -i "power|sclk" | tr '\n' ' '; echo; sleep 0.2; done ) > "$out" &
spid=$!
"$@"
rc=$?
kill $spid 2>/dev/null
exit $rc
