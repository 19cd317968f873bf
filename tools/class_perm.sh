#!/bin/bash
OUT=gpurun_out/${1:-class_perm}; mkdir -p $OUT
for p in "1 2 2 3 3" "3 2 1 1 1" "3 2 2 1 1" "1 2 3 3 3" "1 2 1 3 3" "2 1 1 3 3" "3 1 1 2 2" "2 3 3 1 1" "1 2 2 3 3" "3 2 1 1 1"; do
  python tools/class_perm.py $p 2>/dev/null | grep roles
done | tee $OUT/class_perm.txt
