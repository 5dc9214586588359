# Step time with the nontemporal policy on three more once-per-step streams, alternating on one box:
#   T = the batched weight transpose's stores (146 MB, read again only in the backward), S = the gradient-norm pass's loads (292 MB),
#   W = the weight-gradient epilogues' read-modify-write of the gradient arena (library built by build_wgrad_nt_variant.sh)
C=speech-to-speech-translation_amd/csrc   # (T / S / W were switches and a variant library of the experiment; removed after it: profiles/r06_nontemporal_other_streams_ab.txt)
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "== default: $($B 2>/dev/null | line)"
  echo "== T: $(S2ST_TRANSPOSE_NT=1 $B 2>/dev/null | line)"
  echo "== S: $(S2ST_SUMSQ_NT=1 $B 2>/dev/null | line)"
  echo "== W: $(S2ST_HIP_LIB=$C/libs2st_hip_wnt.so $B 2>/dev/null | line)"
  echo "== TSW: $(S2ST_TRANSPOSE_NT=1 S2ST_SUMSQ_NT=1 S2ST_HIP_LIB=$C/libs2st_hip_wnt.so $B 2>/dev/null | line)"
done
