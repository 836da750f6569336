for sh in "32 32 64 320 320 3 1" "32 32 64 640 640 3 2" "32 64 32 320 320 1 1" "32 128 64 160 160 1 1"; do
  echo "square:"; FVA_WGRAD_THIN=0 python tools/bench_conv.py $sh 10
  echo "thin:  "; python tools/bench_conv.py $sh 10
done
