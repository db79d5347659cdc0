#!/bin/bash
# usage (GPU box, repo root): bash tools/probes/run_mfma_shape_power.sh <binary> <out.txt>
# samples rocm-smi twice a second while the probe runs; the probe prints which shape ran when
BIN=$1
OUT=$2
( for i in $(seq 1 60); do echo "t=$(date +%s.%N)"; rocm-smi --showpower --showclocks --csv 2>/dev/null | grep card0; sleep 0.4; done ) > $OUT.smi &
SMI=$!
echo "start=$(date +%s.%N)" > $OUT
$BIN 5 2>&1 | while read l; do echo "$(date +%s.%N) $l"; done >> $OUT
kill $SMI 2>/dev/null
wait $SMI 2>/dev/null
cat $OUT.smi >> $OUT
