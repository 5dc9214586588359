python -m pytest tests/test_gemm.py tests/test_engine.py tests/test_hubert_train.py -q -m gpu -x -k "persistent or group or base_golden or hubert or fused_backward or grouped" > gpurun_out/r02_t3.txt 2>&1; tail -5 gpurun_out/r02_t3.txt
for v in "" "S2ST_GEMM_PERSIST=0" "S2ST_NO_WGRAD_GROUP=1" "S2ST_WGRAD_MAIN_EVERY=7" "S2ST_GEMM_PERSIST=2" "S2ST_WGRAD_GROUP=8" "S2ST_TRANSPOSE_EACH=1"; do
  echo "== $v" >> gpurun_out/r02_ab1.txt
  env $v python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> gpurun_out/r02_ab1.txt
  env $v python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> gpurun_out/r02_ab1.txt
done
cat gpurun_out/r02_ab1.txt
S2ST_BENCH_VERBOSE=1 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r02_bench2.txt 2>&1
grep "launches/step\|single step\|GPU time" gpurun_out/r02_bench2.txt | cut -c19-
