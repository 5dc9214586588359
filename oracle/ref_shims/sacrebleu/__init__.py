"""Stand-in for ``sacrebleu`` (absent from this image): only the tokenizer registry fairseq/scoring/tokenizer.py:36 looks
up.  TEST INFRASTRUCTURE (build container only)."""
