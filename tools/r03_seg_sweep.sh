python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for seg in 4096 8192 16384; do
  for cfg in "1 8" "16 8" "1 1" "16 1"; do
    echo -n "SEG=$seg: "; EINCM_SEG=$seg python tools/dev_kernel_times.py $cfg 2>&1 | tail -1
  done
done
