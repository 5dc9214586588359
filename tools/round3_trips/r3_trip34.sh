#!/bin/bash
# round 3, trip 34: 100 different batches with a NaN-poisoned workspace before every forward (a kernel that relies on
# cleared memory would surface as a non-finite loss / skipped update), all three bench configurations
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
for cfg in base_recipe base_recipe_hubert infer_base; do
  S2ST_POISON_WORKSPACE=1 timeout 900 python bench.py --config $cfg --cpu-seconds 0 2>/dev/null > gpurun_out/t34_$cfg.txt
  python - "$cfg" <<'PY'
import json, sys
l = json.loads(open("gpurun_out/t34_%s.txt" % sys.argv[1]).read())
print(sys.argv[1], "value", l["value"], "ms", l["ms_per_step"], "final_loss", l["config"].get("final_loss"), "mcd", (l.get("mcd") or {}).get("mcd_gpu_vs_cpu"))
PY
done
echo DONE
