python -m pytest tests/test_inference.py tests/test_inference_mtl.py tests/test_speaker.py tests/test_t2s.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r04_t_tests.txt
python tools/decode_attn_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > gpurun_out/r04_t_attn_bench.txt
for f in "S2ST_DECODE_ATTN_V1=1" "S2ST_DECODE_ATTN_NT=1024"; do echo "== $f" >> gpurun_out/r04_t_attn_bench.txt; env $f python tools/decode_attn_bench.py 2>&1 | grep "fp32\|bf16" >> gpurun_out/r04_t_attn_bench.txt; done
for f in "" "S2ST_DECODE_KV_BF16=1"; do echo "== $f"; env $f S2ST_BENCH_VERBOSE=1 python bench.py --config infer_base 2>gpurun_out/r04_t.err | tail -1 ; grep "decode_attn\|skinny\|gl_" gpurun_out/r04_t.err; done > gpurun_out/r04_t_infer.txt
cat gpurun_out/r04_t_tests.txt;  grep -o '"value": [0-9.]*\|"batch0_decode_ms": [0-9.]*\|"mcd_gpu_vs_cpu": [0-9.]*\|== .*\|.*launches.*' gpurun_out/r04_t_infer.txt
