#!/bin/bash
# same-box A/B of two builds of the library on a probe: profiles/r6_ab.sh <libA.so> <libB.so> <reps> -- <command...>
A=$1; B=$2; R=$3; shift 4
for i in $(seq 1 $R); do
  echo "A: $(TDR_LIB_PATH=$PWD/$A "$@" 2>&1 | tail -1)"
  echo "B: $(TDR_LIB_PATH=$PWD/$B "$@" 2>&1 | tail -1)"
done
