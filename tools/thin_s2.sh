for sh in "32 32 64 640 640 3 2" "32 64 128 320 320 3 2"; do
  echo "paired:"; FVA_PDGRAD2=0 python tools/bench_conv.py $sh 10
  echo "patch: "; python tools/bench_conv.py $sh 10
done
