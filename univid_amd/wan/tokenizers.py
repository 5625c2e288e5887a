"""Text -> token ids for the umT5 encoder: the host-side wrapper `T5EncoderModel` builds from `tokenizer_path`.

Mirrors the interface of models/wan/utils/modules/tokenizers.py:38-82 (`HuggingfaceTokenizer(name, seq_len, clean)`,
`tok(texts, return_mask=True, add_special_tokens=True) -> (ids [B, seq_len], mask [B, seq_len])`, attribute `vocab_size`) on top of
`transformers.AutoTokenizer`. Pure host code, nothing here touches the GPU.

Cleaning modes (reference :11-35): every mode first repairs the text (mojibake repair through `ftfy` when that package is importable -
it is not part of this image, and plain prompts are unaffected - then HTML entities unescaped twice, outer blanks stripped);
'whitespace' then collapses runs of blanks, 'lower' additionally lower-cases, 'canonicalize' replaces underscores by blanks, drops
ASCII punctuation, lower-cases and collapses blanks.
"""
import html
import re
import string

__all__ = ["HuggingfaceTokenizer", "clean_text"]

_BLANKS = re.compile(r"\s+")
_NO_PUNCT = str.maketrans("", "", string.punctuation)


def _repair(text: str) -> str:
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    return html.unescape(html.unescape(text)).strip()


def clean_text(text: str, mode) -> str:
    if mode is None:
        return text
    text = _repair(text)
    if mode == "canonicalize":
        text = text.replace("_", " ").translate(_NO_PUNCT).lower()
    elif mode == "lower":
        text = text.lower()
    elif mode != "whitespace":
        raise ValueError(f"unknown cleaning mode {mode!r}")
    return _BLANKS.sub(" ", text).strip()


class HuggingfaceTokenizer:
    def __init__(self, name, seq_len=None, clean=None, **kwargs):
        if clean not in (None, "whitespace", "lower", "canonicalize"):
            raise ValueError(f"unknown cleaning mode {clean!r}")
        from transformers import AutoTokenizer
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self.tokenizer = AutoTokenizer.from_pretrained(name, **kwargs)
        self.vocab_size = self.tokenizer.vocab_size

    def __call__(self, sequence, **kwargs):
        return_mask = kwargs.pop("return_mask", False)
        opts = {"return_tensors": "pt"}
        if self.seq_len is not None:
            opts.update(padding="max_length", truncation=True, max_length=self.seq_len)
        opts.update(kwargs)
        texts = [sequence] if isinstance(sequence, str) else list(sequence)
        enc = self.tokenizer([clean_text(t, self.clean) for t in texts], **opts)
        return (enc.input_ids, enc.attention_mask) if return_mask else enc.input_ids
