#!/bin/bash
# same-box A/B of two versions of one kernel source: tools/scratch/ab/run.sh <file.hip in csrc> <old> <new> <kbench filter>
cd "$(dirname "$0")/../../.."
for v in "$2" "$3" "$2" "$3"; do
  cp "$v" photoverse_amd/csrc/$1
  python -m photoverse_amd.build > /dev/null 2>&1
  echo "== $v"; python tools/kbench.py "$4" | tail -n +2
done
