#!/bin/bash
# sample clocks/power while the SpMV soak runs (development tool)
OUT=gpurun_out/${1:-soak}; mkdir -p $OUT
( for i in $(seq 1 80); do echo "--- $(date +%s.%N)"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature" ; sleep 0.5; done ) > $OUT/smi.txt 2>&1 &
SMI=$!
sleep 2
echo "start $(date +%s.%N)" | tee $OUT/soak.txt
timeout 120 tools/place_lab 256 10 soak ${2:-60} 2>&1 | tee -a $OUT/soak.txt
echo "end $(date +%s.%N)" | tee -a $OUT/soak.txt
wait $SMI
grep -E "^---|sclk|Power|junction" $OUT/smi.txt | paste - - - - 2>/dev/null | head -90
