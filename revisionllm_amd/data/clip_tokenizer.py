"""CLIP byte-pair tokenizer (lower-cased, byte-level BPE with an end-of-word marker), host side.

Mirror of ``clip.tokenize`` / ``SimpleTokenizer`` as the reference's feature extractors use them
(data/feature_extraction/clip_extractor.py:18,47; clip/clip.py ``tokenize``; clip/simple_tokenizer.py:62-132).  The
algorithm is the published one: text -> whitespace-normalised lower case -> regex pre-tokens -> bytes mapped to
printable unicode symbols -> greedy lowest-rank pair merging (last symbol of a word carries ``</w>``) -> ids; a batch
is ``[<|startoftext|>] + ids + [<|endoftext|>]`` zero-padded to ``context_length`` (77).
The merge table is DATA that ships with CLIP (``bpe_simple_vocab_16e6.txt.gz``, one ``a b`` merge per line after a
header line); pass its path, or set ``CLIP_BPE_PATH``.  ``ftfy`` text repair is applied only if the package is present
(the reference imports it unconditionally; for already clean text it is the identity).
"""
import gzip
import html
import os

import regex

_PRETOKEN = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                          regex.IGNORECASE)
N_MERGES = 49152 - 256 - 2   # merges kept from the file (clip/simple_tokenizer.py:67)


def byte_symbols():
    """byte -> printable unicode symbol (the GPT-2 / CLIP reversible map): printable latin-1 bytes map to themselves,
    the other 68 bytes to code points 256.."""
    keep = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _clean(text):
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text))
    return regex.sub(r"\s+", " ", text.strip()).strip()


class ClipTokenizer:
    def __init__(self, bpe_path=None):
        bpe_path = bpe_path or os.environ.get("CLIP_BPE_PATH")
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError("CLIP merge table not found: pass bpe_path or set CLIP_BPE_PATH to bpe_simple_vocab_16e6.txt.gz")
        with gzip.open(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:1 + N_MERGES]]
        self.sym = byte_symbols()
        # vocabulary order: 256 byte symbols (in the map's own order: kept bytes first, then the remapped ones), the same
        # with </w>, one entry per merge, then the two specials
        order = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
        order += [b for b in range(256) if b not in order]
        base = [self.sym[b] for b in order]
        vocab = base + [s + "</w>" for s in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        self._cache = {"<|startoftext|>": ["<|startoftext|>"], "<|endoftext|>": ["<|endoftext|>"]}
        self._unsym = {v: k for k, v in self.sym.items()}

    def _merge_word(self, word):
        """Greedy BPE of one pre-token (a string of byte symbols) -> list of vocabulary symbols."""
        if word in self._cache:
            return self._cache[word]
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = None, None
            for pair in zip(parts, parts[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._cache[word] = parts
        return parts

    def encode(self, text):
        ids = []
        for tok in _PRETOKEN.findall(_clean(text).lower()):
            word = "".join(self.sym[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._merge_word(word))
        return ids

    def decode(self, ids):
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self._unsym[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def tokenize(self, texts, context_length=77):
        """-> int64 [n, context_length] (torch), zero padded; raises if a text does not fit (clip/clip.py ``tokenize``)."""
        import torch
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            out[i, :len(ids)] = torch.tensor(ids)
        return out
