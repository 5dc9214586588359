
for f in "" 24 32 48 96; do echo "== S2ST_GL_OLA_FRAMES=$f"; S2ST_GL_OLA_FRAMES=$f S2ST_BENCH_VERBOSE=1 python bench.py --config infer_base 2>gpurun_out/r04_p_R$f.err | tail -1 | grep -o '"value": [0-9.]*' | head -1; grep "gl_istft" gpurun_out/r04_p_R$f.err; done > gpurun_out/r04_p_infer.txt
cat gpurun_out/r04_p_infer.txt
