# HuBERT front end (config 3): nontemporal stores of the conv stack's big once-read outputs
#   C0 = the first convolution's apply pass (629 MB of bf16 per 24 x 8 s; switch S2ST_HUBERT_C0_NT of the experiment)
#   G  = GEMM epilogues store a bf16 result > 128 MB nontemporally (variant library, tools/experiments/build_gemm_nt_out_variant.sh)
C=speech-to-speech-translation_amd/csrc   # (C0 was a switch, G a variant library of the experiment: C0 became the default, G was removed -- profiles/r06_hubert_nontemporal_ab.txt)
B="python bench.py --config base_recipe_hubert --steps 20 --warmup 5 --cpu-seconds 0 --no-roofline --no-host-fed --no-other-configs"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
echo "==== front end alone, every dispatch (tools/hubert_timeline.py): default"
python tools/hubert_timeline.py 2>&1 | head -14; python tools/hubert_timeline.py 2>&1 | tail -2
echo "==== C0"
S2ST_HUBERT_C0_NT=1 python tools/hubert_timeline.py 2>&1 | head -14; S2ST_HUBERT_C0_NT=1 python tools/hubert_timeline.py 2>&1 | tail -2
echo "==== G"
S2ST_HIP_LIB=$C/libs2st_hip_ntout.so python tools/hubert_timeline.py 2>&1 | head -14; S2ST_HIP_LIB=$C/libs2st_hip_ntout.so python tools/hubert_timeline.py 2>&1 | tail -2
for rep in 1 2 3; do
  echo "== default: $($B 2>/dev/null | line)"
  echo "== C0: $(S2ST_HUBERT_C0_NT=1 $B 2>/dev/null | line)"
  echo "== G: $(S2ST_HIP_LIB=$C/libs2st_hip_ntout.so $B 2>/dev/null | line)"
  echo "== C0+G: $(S2ST_HUBERT_C0_NT=1 S2ST_HIP_LIB=$C/libs2st_hip_ntout.so $B 2>/dev/null | line)"
done
