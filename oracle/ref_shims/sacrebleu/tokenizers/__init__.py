"""``TOKENIZERS["none"]``: the identity tokenizer -- the default of fairseq's WER scorer (wer.py:15-17, ``wer_tokenizer =
"none"``), the only one the mtl generator's scorer uses.  The others are not provided: asking for them fails loudly."""


class NoneTokenizer:
    def signature(self):
        return "none"

    def __call__(self, line):
        return line


TOKENIZERS = {"none": NoneTokenizer}
