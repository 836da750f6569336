# A/B of the patch kernel on the thin 3x3 stride-1 layers of YOLOv3 at B = 32 (bench_conv: fwd + stats, dgrad, wgrad)
for sh in "32 32 64 320 320 3 1" "32 64 128 160 160 3 1"; do
  echo "igemm:"; FVA_PCONV=0 python tools/bench_conv.py $sh 10
  for tpb in 1 2 4 8; do echo "pconv tpb=$tpb:"; FVA_PCONV_TPB=$tpb python tools/bench_conv.py $sh 10; done
done
