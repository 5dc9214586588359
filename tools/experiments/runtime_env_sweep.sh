# HIP runtime switches (libamdhip64's own environment variables) against the step, and against the step with an EMPTY third stream
# (--exchange-proxy 64,1,600: waits only), one box, two alternations.  (ROC_SYSTEM_SCOPE_SIGNAL=0 HANGS the process on this image: the first run of
# this script spent its whole 30-minute limit there; every run is under `timeout` since)
B="python bench.py --steps 20 --warmup 5 --no-other-configs --cpu-seconds 0 --no-roofline --no-host-fed"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null || echo failed; }
for rep in 1 2; do
  for e in "X=0" "ROC_CPU_WAIT_FOR_SIGNAL=0" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "DEBUG_HIP_DYNAMIC_QUEUES=0" "DEBUG_HIP_DYNAMIC_QUEUES=1" "AMD_DIRECT_DISPATCH=0" "GPU_FLUSH_ON_EXECUTION=1" "DEBUG_CLR_MAX_BATCH_SIZE=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "ROC_USE_FGS_KERNARG=0" "ROC_AQL_QUEUE_SIZE=65536" "GPU_STREAMOPS_CP_WAIT=1"; do
    echo "== $e: step $(timeout 90 env $e $B 2>/dev/null | line)   with an empty third stream $(timeout 90 env $e $B --exchange-proxy 64,1,600 2>/dev/null | line)"
  done
done
