#!/bin/bash
OUT=gpurun_out/r05_chains_dbg.txt
: > $OUT
for i in 1 2; do
S2ST_CHAINS=2 timeout 2400 python -m pytest tests/test_engine.py tests/test_full_size.py tests/test_fairseq_plugin.py tests/test_mtl.py tests/test_t2s.py tests/test_speaker.py tests/test_hubert_train.py tests/test_s2t.py tests/test_resume.py -q -m gpu 2>&1 | grep -E "^E  |^tests/|passed|failed|^FAILED|Error" | cut -c1-600 >> $OUT
done
