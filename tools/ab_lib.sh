#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: tools/ab_lib.sh <variant .so> <command ...>
# (the command is run with QNN_LIB unset = the in-tree library, then with QNN_LIB=<variant>, twice each)
V=$1; shift
for i in 1 2; do
  echo "== in-tree"; "$@"
  echo "== $V"; QNN_LIB=$V "$@"
done
