"""CLIP byte-pair-encoding tokenizer for the text tower.

The reference tokenises prompts with ``open_clip.tokenize`` (lvdm/modules/encoders/condition.py:207-210; open_clip_torch 2.22,
``open_clip/tokenizer.py::SimpleTokenizer``), whose merge table ``bpe_simple_vocab_16e6.txt.gz`` ships inside the open_clip wheel —
not inside the reference repository and not in this image. This module restates the published algorithm and LOADS THE TABLE FROM A
USER-SUPPLIED PATH (``SimpleTokenizer(bpe_path)``, or the ``GEO4D_CLIP_BPE`` environment variable), so the prompt the shipped
scripts pass (``scripts/infer_geo4d.sh`` runs with ``--text_input``: test_geo4d.py:124-126 keeps the fixed prompt of :410) can be
encoded. Vocabulary layout (identical to open_clip's, so token ids index the checkpoint's ``token_embedding`` rows):

    [256 byte symbols] [256 byte symbols + '</w>'] [48 894 merges, in file order] <start_of_text> <end_of_text>   = 49 408 ids

Text cleaning = html.unescape twice + whitespace collapse + lower case. open_clip additionally runs ``ftfy.fix_text`` first (a
mojibake repair that is the identity on well-formed text); ftfy is not installed here, and that difference is stated, not hidden.
"""
import gzip
import html
import os
from functools import lru_cache

import regex as re
import torch

CONTEXT_LENGTH = 77
N_MERGES = 49152 - 256 - 2          # open_clip reads lines [1 : 49152 - 256 - 2 + 1] of the file (line 0 is a version header)


@lru_cache()
def bytes_to_unicode():
    """The GPT-2 byte <-> printable-unicode table: the 188 printable latin-1 code points map to themselves, the other 68 bytes to
    code points 256 + n in byte order."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def _pairs(word):
    return set(zip(word[:-1], word[1:]))


def _clean(text):
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class SimpleTokenizer:
    PATTERN = r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"

    def __init__(self, bpe_path=None, context_length=CONTEXT_LENGTH, merges=None):
        """``bpe_path``: a ``bpe_simple_vocab_16e6.txt[.gz]`` file (first line = header, then one merge "a b" per line);
        ``merges``: alternatively the merge list itself (tests build small synthetic tables)."""
        if merges is None:
            bpe_path = bpe_path or os.environ.get("GEO4D_CLIP_BPE")
            if not bpe_path or not os.path.exists(bpe_path):
                raise FileNotFoundError(
                    "CLIP BPE merge table not found: pass bpe_path= or set GEO4D_CLIP_BPE to open_clip's bpe_simple_vocab_16e6.txt.gz "
                    "(it ships in the open_clip_torch wheel, not in the Geo4D repository)")
            opener = gzip.open if str(bpe_path).endswith(".gz") else open
            with opener(bpe_path, "rb") as f:
                lines = f.read().decode("utf-8").split("\n")
            merges = [tuple(m.split()) for m in lines[1:N_MERGES + 1] if m.strip()]
        merges = [tuple(m) for m in merges]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<start_of_text>": "<start_of_text>", "<end_of_text>": "<end_of_text>"}
        self.pat = re.compile(self.PATTERN, re.IGNORECASE)
        self.context_length = context_length
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]

    def bpe(self, token):
        """Greedy lowest-rank-first merging of one pre-token; the last symbol carries the end-of-word marker."""
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            best = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if best not in self.bpe_ranks:
                break
            first, second = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    out.append(first + second)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        res = " ".join(word)
        self.cache[token] = res
        return res

    def encode(self, text):
        ids = []
        for tok in re.findall(self.pat, _clean(text).lower()):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def decode(self, ids):
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts, context_length=None):
        """open_clip.tokenize: [<start> ids... <end>] zero-padded to the context length; longer prompts are truncated and their last
        position overwritten with <end> -> int64 [B, context_length]."""
        if isinstance(texts, str):
            texts = [texts]
        n = context_length or self.context_length
        out = torch.zeros((len(texts), n), dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > n:
                ids = ids[:n]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out
