# the five thin / stride-2 3x3 layers, all three passes (isolated launches)
for sh in "32 32 64 640 640 3 2" "32 32 64 320 320 3 1" "32 64 128 320 320 3 2" "32 64 128 160 160 3 1"; do python tools/bench_conv.py $sh 10; done
