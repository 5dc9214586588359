"""Symbol table with fairseq's file format and special-symbol layout.

Counterpart of ``fairseq/data/dictionary.py:17-329`` for what the s2s_translation data path uses: ``load`` of a
``<token> <count> [#fairseq:overwrite]`` text file on top of the four specials (<s>=0, <pad>=1, </s>=2, <unk>=3),
``index``, ``encode_line`` (whitespace tokenisation of ``fairseq/tokenizer.py``, EOS appended), ``string``.
"""
from __future__ import annotations

import re
from typing import Iterable, List

import torch

_SPACE = re.compile(r"\s+")


def tokenize_line(line: str) -> List[str]:
    return _SPACE.sub(" ", line).strip().split()


class Dictionary:
    def __init__(self, *, bos="<s>", pad="<pad>", eos="</s>", unk="<unk>", extra_special_symbols=None):
        self.bos_word, self.unk_word, self.pad_word, self.eos_word = bos, unk, pad, eos
        self.symbols: List[str] = []
        self.count: List[int] = []
        self.indices = {}
        self.bos_index = self.add_symbol(bos)
        self.pad_index = self.add_symbol(pad)
        self.eos_index = self.add_symbol(eos)
        self.unk_index = self.add_symbol(unk)
        for s in extra_special_symbols or []:
            self.add_symbol(s)
        self.nspecial = len(self.symbols)

    @classmethod
    def with_size(cls, n_symbols: int) -> "Dictionary":
        """Synthetic table of ``n_symbols`` entries (specials + s0, s1, ...), as the bench corpus uses."""
        d = cls()
        for i in range(n_symbols - d.nspecial):
            d.add_symbol(f"s{i}")
        return d

    def __len__(self):
        return len(self.symbols)

    def __contains__(self, sym):
        return sym in self.indices

    def __getitem__(self, idx):
        return self.symbols[idx] if idx < len(self.symbols) else self.unk_word

    def __eq__(self, other):
        return isinstance(other, Dictionary) and self.indices == other.indices

    def index(self, sym: str) -> int:
        assert isinstance(sym, str)
        return self.indices.get(sym, self.unk_index)

    def add_symbol(self, word: str, n: int = 1, overwrite: bool = False) -> int:
        if word in self.indices and not overwrite:
            idx = self.indices[word]
            self.count[idx] += n
            return idx
        idx = len(self.symbols)
        self.indices[word] = idx
        self.symbols.append(word)
        self.count.append(n)
        return idx

    def bos(self):
        return self.bos_index

    def pad(self):
        return self.pad_index

    def eos(self):
        return self.eos_index

    def unk(self):
        return self.unk_index

    @classmethod
    def load(cls, f) -> "Dictionary":
        d = cls()
        d.add_from_file(f)
        return d

    def add_from_file(self, f):
        if isinstance(f, str):
            with open(f, "r", encoding="utf-8") as fd:
                return self.add_from_file(fd)
        for raw in f.readlines():
            try:
                line, field = raw.rstrip().rsplit(" ", 1)
                overwrite = field == "#fairseq:overwrite"
                if overwrite:
                    line, field = line.rsplit(" ", 1)
                count = int(field)
            except ValueError:
                raise ValueError(f"Incorrect dictionary format, expected '<token> <cnt> [flags]': \"{raw}\"")
            if line in self and not overwrite:
                raise RuntimeError(f"Duplicate word found when loading Dictionary: '{line}'")
            self.add_symbol(line, n=count, overwrite=overwrite)

    def encode_line(self, line: str, line_tokenizer=tokenize_line, add_if_not_exist: bool = True, consumer=None,
                    append_eos: bool = True, reverse_order: bool = False) -> torch.Tensor:
        words = line_tokenizer(line)
        if reverse_order:
            words = list(reversed(words))
        ids = []  # (one tensor construction at the end: per-element tensor stores cost microseconds each)
        for w in words:
            idx = self.add_symbol(w) if add_if_not_exist else self.index(w)
            if consumer is not None:
                consumer(w, idx)
            ids.append(idx)
        if append_eos:
            ids.append(self.eos_index)
        return torch.tensor(ids, dtype=torch.int32)

    def string(self, tensor: Iterable, include_eos: bool = False, separator: str = " ") -> str:
        if torch.is_tensor(tensor) and tensor.dim() == 2:
            return "\n".join(self.string(t, include_eos=include_eos) for t in tensor)
        skip = {self.bos()} | (set() if include_eos else {self.eos()})
        return separator.join(self[int(i)] for i in tensor if int(i) not in skip)
