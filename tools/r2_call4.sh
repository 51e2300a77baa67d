OUT=gpurun_out/c4; mkdir -p $OUT
for t in 60 66 63 14; do
  for cfg in "1 360 216 128 128 1" "1 360 216 128 128 3" "1 360 216 256 128 3" "1 360 216 512 128 3"; do
    echo "tile $t cfg $cfg: $(python tools/one_conv.py $cfg $t 5 2>&1 | tail -1)" >> $OUT/1_ksweep.txt
  done
done
for t in 67 66 63 22; do
  for cfg in "3 30 54 256 256 1" "3 30 54 256 256 3" "3 30 54 512 256 3" "3 30 54 1024 256 3"; do
    echo "tile $t cfg $cfg: $(python tools/one_conv.py $cfg $t 5 2>&1 | tail -1)" >> $OUT/1_ksweep.txt
  done
done
cat $OUT/1_ksweep.txt
