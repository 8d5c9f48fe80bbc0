# structured (band of +-1 K columns) and uniform matrices on shapes where AUTO takes an L2-level panel plan: custom_mm.naive_spmm (probe active)
# beside AUTO without a workspace (passes stay passes) and the one-pass plans
for a in "339200 115456 192 40" "607232 142080 192 36" "154368 32000 512 82" "118272 11776 384 46" "587776 28416 768 85" "271104 29952 96 275" "198400 499456 192 317" "120832 86528 320 540" "16384 16384 256 82" "65536 16384 256 49" "32768 32768 192 328" "8192 131072 256 1311"; do
  for pt in band1k uniform; do timeout -k 10 100 python tools/probes/adapt_probe.py $a $pt 2>&1 | grep "^M "; done
done
