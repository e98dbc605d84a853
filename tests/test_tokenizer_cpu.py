"""CLIP BPE tokenizer (geo4d_amd/tokenizer.py, N3): open_clip's merge table is not available offline, so the algorithm is pinned on a
SYNTHETIC table (a small BPE trained here) against an independent implementation of the same published algorithm — HuggingFace
transformers' CLIPTokenizer built from the same vocabulary / merges — plus the open_clip.tokenize framing rules
(<start> ... <end>, zero padding to 77, truncation overwrites the last position with <end>)."""
import gzip
from collections import Counter

import pytest
import torch

from geo4d_amd.tokenizer import SimpleTokenizer, bytes_to_unicode

CORPUS = ("a photo of a drifting car turning on the road . the camera moves forward slowly , point map of the scene ; "
          "4d geometry it's don't").split()


def train_merges(n=60):
    b2u = bytes_to_unicode()

    def sym(w):
        u = "".join(b2u[b] for b in w.encode())
        return tuple(u[:-1]) + (u[-1] + "</w>",)
    words = Counter(sym(w) for w in CORPUS)
    merges = []
    for _ in range(n):
        pairs = Counter()
        for w, c in words.items():
            for p in zip(w[:-1], w[1:]):
                pairs[p] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        nw = Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1])
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            nw[tuple(out)] += c
        words = nw
    return merges


PROMPTS = ["a photo of a drifting car", "The camera moves forward slowly, it's 4d geometry; don't", "  road   scene . ",
           "unknownword zzz 123", "map&amp;scene", ""]


def test_matches_hf_clip_tokenizer_on_a_synthetic_table():
    transformers = pytest.importorskip("transformers")
    merges = train_merges()
    tok = SimpleTokenizer(merges=merges)
    ren = {"<start_of_text>": "<|startoftext|>", "<end_of_text>": "<|endoftext|>"}
    hf = transformers.CLIPTokenizer(vocab={ren.get(k, k): v for k, v in tok.encoder.items()}, merges=[tuple(m) for m in merges])
    for text in PROMPTS[:4]:
        mine = tok(text)[0]
        n = int((mine != 0).sum())
        assert mine[:n].tolist() == hf(text)["input_ids"], text
        assert mine[0] == tok.sot and mine[n - 1] == tok.eot and bool((mine[n:] == 0).all())
    assert tok.decode(tok.encode("the drifting car")).strip() == "the drifting car"
    assert tok.encode("map&amp;scene") == tok.encode("map&scene")                 # html.unescape in the cleaner


def test_table_file_loading_framing_and_truncation(tmp_path):
    merges = train_merges()
    path = tmp_path / "bpe_simple_vocab_16e6.txt.gz"
    with gzip.open(path, "wb") as f:
        f.write(("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n").encode("utf-8"))
    a, b = SimpleTokenizer(str(path)), SimpleTokenizer(merges=merges)
    assert a.encoder == b.encoder and a.bpe_ranks == b.bpe_ranks
    assert len(a.encoder) == 512 + len(merges) + 2 and a.sot == len(a.encoder) - 2 and a.eot == len(a.encoder) - 1
    t = a(PROMPTS)
    assert t.shape == (len(PROMPTS), 77) and t.dtype == torch.long
    assert t[-1, :3].tolist() == [a.sot, a.eot, 0]                                # the blank prompt: <start> <end> padding
    long = a("car " * 200)[0]
    assert long[0] == a.sot and long[-1] == a.eot and int((long == 0).sum()) == 0   # truncated, last position = <end>
    with pytest.raises(FileNotFoundError):
        SimpleTokenizer(str(tmp_path / "missing.txt.gz"))


def test_real_table_layout_gives_openclip_special_ids():
    """With a full-length (48 894-merge) table the specials land on open_clip's ids 49406 / 49407 - the constants the encoder uses for
    the blank prompt. Synthetic distinct merges stand in for the real lines (only the COUNT matters for this layout property)."""
    from geo4d_amd.encoders import EOT, SOT
    from geo4d_amd.tokenizer import N_MERGES
    tok = SimpleTokenizer(merges=[("x%d" % i, "y") for i in range(N_MERGES)])
    assert (tok.sot, tok.eot) == (SOT, EOT) == (49406, 49407) and len(tok.encoder) == 49408


def test_text_encoder_tokenize_uses_the_table(tmp_path, monkeypatch):
    from geo4d_amd.encoders import FrozenOpenCLIPEmbedder
    merges = train_merges()
    path = tmp_path / "bpe.txt"
    path.write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
    n_vocab = 512 + len(merges) + 2
    enc = FrozenOpenCLIPEmbedder(layer="penultimate", width=64, layers=1, heads=1, vocab_size=n_vocab, bpe_path=str(path))
    toks = enc.tokenize(["a photo of a drifting car", ""])
    ref = SimpleTokenizer(str(path))
    assert torch.equal(toks, ref(["a photo of a drifting car", ""]))
    blank = FrozenOpenCLIPEmbedder(layer="penultimate", width=64, layers=1, heads=1).tokenize([""])    # no table needed
    assert blank[0, :3].tolist() == [49406, 49407, 0]
    monkeypatch.delenv("GEO4D_CLIP_BPE", raising=False)
    with pytest.raises(FileNotFoundError, match="bpe_simple_vocab_16e6"):
        FrozenOpenCLIPEmbedder(layer="penultimate", width=64, layers=1, heads=1).tokenize(["a car"])
    small = FrozenOpenCLIPEmbedder(layer="penultimate", width=64, layers=1, heads=1, vocab_size=300, bpe_path=str(path))
    with pytest.raises(ValueError, match="outside the embedding table"):
        small.tokenize(["a car"])
