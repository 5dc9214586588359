"""Deterministic miniature of the recipe's on-disk layout (TSV manifests, uncompressed zip of .npy features, config
YAML, dictionaries, global-CMVN statistics), written into a scratch directory.  Shared by
oracle/gen_golden_data.py (which feeds it to the REFERENCE dataset classes) and tests/test_data.py (which feeds
the same files to ours): the committed golden holds only the expected outputs."""
import os
import zipfile

import numpy as np

N_UTTS = 9
FEAT = 80
SRC_WORDS = ["hola", "que", "tal", "bien", "gracias", "adios", "si", "no"]
TGT_WORDS = ["hello", "how", "are", "you", "fine", "thanks", "bye", "yes", "no", "well"]


def make_corpus(root: str) -> str:
    os.makedirs(root, exist_ok=True)
    rs = np.random.RandomState(20)
    src_len = rs.randint(23, 61, size=N_UTTS)
    src_len[3] = src_len[5]  # a tie, to pin the stable secondary order of ordered_indices
    tgt_len = rs.randint(17, 50, size=N_UTTS)
    zpath = os.path.join(root, "feats.zip")
    entries = {}
    with zipfile.ZipFile(zpath, "w", compression=zipfile.ZIP_STORED) as z:
        for i in range(N_UTTS):
            for side, n in (("src", src_len[i]), ("tgt", tgt_len[i])):
                arr = (rs.standard_normal((int(n), FEAT)) * 2.0 + 1.0).astype(np.float32)
                if i == N_UTTS - 1:  # the last utterance lives in plain .npy files
                    np.save(os.path.join(root, f"{side}_{i}.npy"), arr)
                    continue
                name = f"{side}_{i}.npy"
                with z.open(name, "w") as f:
                    np.save(f, arr)
                entries[name] = None
    with zipfile.ZipFile(zpath) as z:
        with open(zpath, "rb") as raw:
            for info in z.infolist():
                raw.seek(info.header_offset)
                hdr = raw.read(30)
                n_name, n_extra = int.from_bytes(hdr[26:28], "little"), int.from_bytes(hdr[28:30], "little")
                entries[info.filename] = (info.header_offset + 30 + n_name + n_extra, info.file_size)

    def ref(side, i):
        if i == N_UTTS - 1:
            return f"{side}_{i}.npy"
        off, size = entries[f"{side}_{i}.npy"]
        return f"feats.zip:{off}:{size}"

    def text(words, n):
        return " ".join(words[j] for j in rs.randint(0, len(words), size=n))

    cols = ["id", "src_audio", "src_n_frames", "tgt_audio", "tgt_n_frames", "src_text", "tgt_text", "speaker"]
    for split, ids in (("train_tiny", range(N_UTTS)), ("dev_tiny", range(2, 6))):
        with open(os.path.join(root, f"{split}.tsv"), "w") as f:
            f.write("\t".join(cols) + "\n")
            for i in ids:
                st = text(SRC_WORDS + ["oov_es"], int(rs.randint(2, 7)))
                tt = text(TGT_WORDS + ["oov_en"], int(rs.randint(2, 8)))
                f.write("\t".join([f"utt{i}", ref("src", i), str(src_len[i]), ref("tgt", i), str(tgt_len[i]), st, tt,
                                   f"spk{i % 2}"]) + "\n")
    for name, words in (("src_dict.txt", SRC_WORDS), ("tgt_dict.txt", TGT_WORDS)):
        with open(os.path.join(root, name), "w") as f:
            for k, w in enumerate(words):
                f.write(f"{w} {100 - k}\n")
    for name in ("src_gcmvn.npz", "tgt_gcmvn.npz"):
        np.savez(os.path.join(root, name), mean=rs.standard_normal(FEAT).astype(np.float32),
                 std=(0.5 + rs.rand(FEAT)).astype(np.float32))
    with open(os.path.join(root, "config.yaml"), "w") as f:
        f.write(f"""src_vocab_filename: src_dict.txt
tgt_vocab_filename: tgt_dict.txt
audio_root: {root}
shuffle: false
src_transforms:
  _train: [src_global_cmvn, specaugment]
  _eval: [src_global_cmvn]
tgt_transforms:
  '*': [utterance_cmvn, tgt_global_cmvn]
src_global_cmvn:
  stats_npz_path: {root}/src_gcmvn.npz
tgt_global_cmvn:
  stats_npz_path: {root}/tgt_gcmvn.npz
utterance_cmvn:
  norm_means: true
  norm_vars: false
specaugment:
  freq_mask_N: 2
  freq_mask_F: 27
  time_mask_N: 2
  time_mask_T: 10
  time_mask_p: 0.5
features:
  eps: 1.0e-05
  f_max: 8000
  f_min: 20
  hop_len_t: 0.004
  hop_length: 64
  n_fft: 256
  n_mels: 80
  n_stft: 129
  sample_rate: 16000
  type: spectrogram+melscale+log
  win_len_t: 0.0125
  win_length: 200
  window_fn: hann
sample_rate: 16000
""")
    return root


def make_mtl_extras(root: str) -> str:
    """A manifest with the FastSpeech-style columns the mtl task's dataset still reads (duration / pitch / energy,
    s2st_dataset_mtl.py:389-405): ``dev_fs.tsv`` over the dev utterances.  The reference appends a 0 "for EOS" to each
    of the three (:206-219) and asserts their collated width equals ``src_text``'s -- whose EOS it has REMOVED (:195) --
    so a manifest that passes has one value fewer than the source text has words."""
    import csv
    rs = np.random.RandomState(31)
    with open(os.path.join(root, "dev_tiny.tsv")) as f:
        rows = list(csv.DictReader(f, delimiter="\t", quotechar=None, doublequote=False, lineterminator="\n",
                                   quoting=csv.QUOTE_NONE))
    cols = list(rows[0].keys()) + ["duration", "pitch", "energy"]
    with open(os.path.join(root, "dev_fs.tsv"), "w") as f:
        f.write("\t".join(cols) + "\n")
        for k, r in enumerate(rows):
            n = len(r["src_text"].split(" ")) - 1
            r = dict(r)
            r["duration"] = " ".join(str(int(v)) for v in rs.randint(1, 9, size=n))
            for name in ("pitch", "energy"):
                np.save(os.path.join(root, f"{name}_{k}.npy"), rs.standard_normal(n).astype(np.float32))
                r[name] = f"{name}_{k}.npy"
            f.write("\t".join(r[c] for c in cols) + "\n")
    return root


def flatten_batch(b, prefix=""):
    """Collated sample -> {name: ndarray} (tensors only; strings / None recorded as such)."""
    import torch
    out = {}
    for k, v in b.items():
        if isinstance(v, dict):
            out.update(flatten_batch(v, prefix + k + "."))
        elif torch.is_tensor(v):
            out[prefix + k] = v.numpy()
        elif isinstance(v, (int, float)):
            out[prefix + k] = np.asarray(v)
        elif isinstance(v, list) and all(isinstance(s, str) for s in v):
            out[prefix + k] = np.asarray(v)
        elif v is None:
            out[prefix + k + ".is_none"] = np.asarray(1)
    return out
