"""Byte-level BPE tokenizer compatible with OpenAI CLIP's vocabulary (reference
clip/simple_tokenizer.py:62-127 implements the same published algorithm).

The merge table `bpe_simple_vocab_16e6.txt.gz` (OpenAI CLIP's published vocabulary — data, not code; the reference vendors
the same file, clip/simple_tokenizer.py:4-6) sits next to this module; `PCLIP_BPE_VOCAB` overrides the path.  A missing table
is a loud FileNotFoundError, never a silent fallback.  Token ids: 256 byte symbols, 256 end-of-word byte symbols, 48894
merges, then <|startoftext|>=49406 and <|endoftext|>=49407."""
import gzip
import html
import os
from functools import lru_cache

import regex as re

N_MERGES = 49152 - 256 - 2


@lru_cache()
def _byte_symbols():
    """Reversible byte -> printable unicode map (printable latin-1 bytes map to themselves, the rest
    are shifted past U+0100), so BPE never sees whitespace/control characters."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class SimpleTokenizer:
    def __init__(self, bpe_path: str):
        with gzip.open(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:N_MERGES + 1]]
        # OpenAI's vocabulary order: byte symbols in *sorted-keep-list* order, not byte order
        keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
        order = keep + [b for b in range(256) if b not in keep]
        sym = _byte_symbols()
        vocab = [sym[b] for b in order]
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.byte_sym = sym
        self.sym_byte = {v: k for k, v in sym.items()}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                              re.IGNORECASE)

    def _bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word[:-1], word[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and (word[i], word[i + 1]) == best:
                    merged.append(word[i] + word[i + 1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self.cache[token] = out
        return out

    @staticmethod
    def _clean(text: str) -> str:
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:      # identity on the ASCII prompts the datasets use
            pass
        text = html.unescape(html.unescape(text)).strip()
        return re.sub(r"\s+", " ", text).strip().lower()

    def encode(self, text: str):
        ids = []
        for piece in re.findall(self.pat, self._clean(text)):
            piece = "".join(self.byte_sym[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(piece).split(" "))
        return ids

    def decode(self, tokens) -> str:
        text = "".join(self.decoder[t] for t in tokens)
        return bytearray(self.sym_byte[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


_default = None


def default_bpe() -> str:
    """Path of the merge table: $PCLIP_BPE_VOCAB, else the copy shipped beside this module (clip/simple_tokenizer.py:9-11)."""
    return os.environ.get("PCLIP_BPE_VOCAB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "bpe_simple_vocab_16e6.txt.gz")


def default_tokenizer() -> SimpleTokenizer:
    global _default
    if _default is None:
        path = default_bpe()
        if not os.path.isfile(path):
            raise FileNotFoundError(
                f"CLIP BPE merge table not found at {path}: restore proto-clip_amd/clip/bpe_simple_vocab_16e6.txt.gz or set "
                "PCLIP_BPE_VOCAB=/path/to/bpe_simple_vocab_16e6.txt.gz (shipped with every CLIP installation)")
        _default = SimpleTokenizer(path)
    return _default
