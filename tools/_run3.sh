export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
./build/micro/store_pattern 400000 20 > $O/store_pattern.txt 2>&1
./build/micro/store_pattern 400000 20 >> $O/store_pattern.txt 2>&1
cat $O/store_pattern.txt
