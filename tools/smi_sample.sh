#!/bin/bash
# sample the GPU's clocks, power and busy percentage once a second while a command runs (tuning aid):
#   bash tools/smi_sample.sh <out.log> <command...>
OUT=$1; shift
( while true; do rocm-smi --showuse --showpower --showclocks --showmemuse 2>/dev/null | grep -E "GPU\[0\].*(sclk|mclk|Power|use|Busy|fclk)" | tr '\n' ';' ; echo; sleep 1; done ) > $OUT 2>&1 &
SP=$!
"$@"
kill $SP 2>/dev/null
