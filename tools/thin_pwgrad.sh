# A/B of the all-taps weight-gradient kernel on its four layers: tap groups (FVA_PWGRAD=0) / all taps, default and small patches
for sh in "32 32 64 320 320 3 1" "32 32 64 640 640 3 2" "32 64 128 160 160 3 1" "32 64 128 320 320 3 2"; do
  echo "tap groups:       $(FVA_PWGRAD=0 python tools/bench_conv.py $sh 10)"
  echo "all taps 8/4:     $(python tools/bench_conv.py $sh 10)"
  echo "all taps 4/2:     $(FVA_PWGRAD_TH=4,2 python tools/bench_conv.py $sh 10)"
  echo "all taps 8/4 x3:  $(FVA_PWGRAD_SLOTS=3 python tools/bench_conv.py $sh 10)"
  echo "all taps 4/2 x3:  $(FVA_PWGRAD_TH=4,2 FVA_PWGRAD_SLOTS=3 python tools/bench_conv.py $sh 10)"
done
