# every other first-round block of the 8-phase kernel starts late (FVA_IGEMM8_SKEW x 8128 cycles): do staggered epilogues pay?
for k in 0 1 2 0; do
  echo "skew $k"; FVA_IGEMM8_SKEW=$k python tools/tile_timing.py 2>&1 | grep -v amdgpu.ids | grep "stats=1" | cut -c1-200
done
