#!/bin/bash
# tests/tools/repeat_fuzz.sh <seed> <cases> <repeats> [lib.so ...]: reruns the same fuzz cases to expose
# timing-dependent (nondeterministic) mismatches; prints failures per library.
seed=$1; cases=$2; reps=$3; shift 3
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=("")
for lib in "${libs[@]}"; do
  fails=0
  for i in $(seq 1 $reps); do
    if [ -n "$lib" ]; then export BS_LIB_PATH=$PWD/$lib; fi
    if python tests/tools/fuzz_parity.py --cases $cases --seed $seed 2>/dev/null | grep -q MISMATCH; then fails=$((fails+1)); fi
  done
  echo "lib=${lib:-default} seed=$seed cases=$cases: $fails / $reps runs with a mismatch"
done
