# 1x1 weight gradients: 8-phase kernel (256 x 256 tiles, default) against the 128 x 128 kernel
for sh in "32 256 128 80 80 1 1" "32 512 256 40 40 1 1" "32 1024 512 20 20 1 1" "32 768 256 40 40 1 1" "32 384 128 80 80 1 1" "32 128 64 160 160 1 1"; do
  echo "8-phase:   $(python tools/bench_conv.py $sh 10)"
  echo "128 x 128: $(FVA_WGRAD8=0 python tools/bench_conv.py $sh 10)"
done
