for f in "S2ST_HUBERT_AHEAD=1" "S2ST_HUBERT_AHEAD=0"; do echo "== $f"; env $f python bench.py --config base_recipe_hubert --steps 40 --cpu-seconds 0 2>gpurun_out/r04_w_$f.err | tail -1 | cut -c1-2000; done > gpurun_out/r04_w_hubert.txt
grep -o '== .*\|"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"host_fed_ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' gpurun_out/r04_w_hubert.txt; 
