#!/bin/bash
# usage: tools/isa_loop.sh <file.hip> [n-th barrier-delimited region, default 2]: compile one kernel source to gfx950 ISA and
# summarise the instruction classes of the region between barriers n and n + 1 (the K-loop body of the GEMM kernels): spills
# (scratch_*), waits, MFMAs, LDS reads, LDS-DMA issues, in order
SRC=/root/repo/speech-to-speech-translation_amd/csrc
f=$1; n=${2:-2}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $SRC -I /root/repo/include -Wno-unused-value -x hip --cuda-device-only -S $SRC/$f -o /tmp/isa_loop.s 2>&1 | grep -v "hip-link"
grep "vgpr_spill_count\|\.vgpr_count\|sgpr_count" /tmp/isa_loop.s
awk -v n=$n '/s_barrier/{c++} c>=n && c<n+1' /tmp/isa_loop.s | grep -v "^\s*;" | grep -n "scratch_\|s_waitcnt\|v_mfma\|ds_read\|global_load_lds\|s_cbranch\|s_setprio" | awk '{print $2, $3, $4, $5, $6}' | head -${3:-120} | awk '{printf "%s ", $1; if ($1 ~ /scratch/ || $1 ~ /waitcnt/) print "   <<< " $0; else print ""}' | uniq -c
