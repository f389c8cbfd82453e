#!/bin/bash
# repeats one pytest selection N times and counts failures: tools/probes/flaky_loop.sh N "<pytest args>"
N=$1; shift
f=0
for i in $(seq 1 $N); do timeout 300 python -m pytest "$@" -q -m gpu 2>&1 | grep -q failed && f=$((f+1)); done
echo "failures: $f of $N"
