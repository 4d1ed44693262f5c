for cap in 0 57344; do
  for ab in 0 8 16 24 32; do
    echo "== cap $cap ablate $ab (8 = no DMA wait, 16 = no stage barrier, 32 = no DMA after the prologue)"
    timeout -k 10 60 tests/hip/sweep_gemm_ab$ab pmc $cap 2>&1 | tr '\n' ';'
    echo
  done
done
