# where the all-taps weight gradient's time goes (variants built with -DFVA_PWGRAD_ABL=n, see csrc/conv_wgrad.hip)
for sh in "32 32 64 320 320 3 1" "32 64 128 160 160 3 1"; do
  echo "full:                 $(python tools/bench_conv.py $sh 10)"
  echo "staging only:         $(FVA_LIB_PATH=fastvision_amd/csrc/variants/lib_pwabl1.so python tools/bench_conv.py $sh 10)"
  echo "k-steps only:         $(FVA_LIB_PATH=fastvision_amd/csrc/variants/lib_pwabl2.so python tools/bench_conv.py $sh 10)"
  echo "fragment reads only:  $(FVA_LIB_PATH=fastvision_amd/csrc/variants/lib_pwabl4.so python tools/bench_conv.py $sh 10)"
done
