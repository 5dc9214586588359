# Step time with the nt cache policy on the GEMMs' LDS-DMA loads (libraries from build_dma_policy_variants.sh), alternating.
C=speech-to-speech-translation_amd/csrc   # (the variant libraries of this run were built from a since-removed template parameter: see profiles/r06_dma_nontemporal_ab.txt)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  for v in default a b ab; do
    if [ $v = default ]; then L=$C/libs2st_hip.so; else L=$C/libs2st_hip_dma_$v.so; fi
    echo "== $v: $(S2ST_HIP_LIB=$L python bench.py --steps 20 --warmup 5 --no-other-configs 2>/dev/null | line)"
  done
done
for v in default a b ab; do
  if [ $v = default ]; then L=$C/libs2st_hip.so; else L=$C/libs2st_hip_dma_$v.so; fi
  echo "==== kernel table, $v"
  S2ST_HIP_LIB=$L S2ST_BENCH_VERBOSE=1 python bench.py --steps 20 --warmup 5 --no-other-configs 2>&1 | grep -E "launches .* avg" | head -24
done
echo "==== gradient-norm pass with nt loads (Adam nt in both)"
for rep in 1 2 3; do
  for s in 0 1; do
    echo "== adam nt, sumsq_nt $s: $(S2ST_ADAM_VARIANT=1 S2ST_SUMSQ_NT=$s python bench.py --steps 20 --warmup 5 --no-other-configs 2>/dev/null | line)"
  done
done
