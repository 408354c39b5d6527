"""Whisper token-id layout and text decoding.

The reference gets its tokenizer from third-party mlx_whisper.tokenizer.get_tokenizer
(mlx_whisper_optimized_final.py:23,277-282); only the ids matter to the HIP path
(special tokens, timestamp_begin = 50365 for large-v3 as pinned by the token lists in
/root/reference/30m.json).  Text decoding uses a local HF `tokenizer.json` when a
checkpoint directory provides one; without it (random-weight runs) ids are rendered
as placeholders so the result dict keeps its shape.
"""
from dataclasses import dataclass, field
from typing import List, Optional

LANGUAGES = ["en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi",
             "fi", "vi", "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la",
             "mi", "ml", "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy",
             "ne", "mn", "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be",
             "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl",
             "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su", "yue"]

# language names of the Whisper tokenizer, in LANGUAGES order (for `--language English` style arguments)
LANGUAGE_NAMES = ['english', 'chinese', 'german', 'spanish', 'russian', 'korean', 'french', 'japanese', 'portuguese', 'turkish', 'polish', 'catalan', 'dutch', 'arabic', 'swedish', 'italian', 'indonesian', 'hindi', 'finnish', 'vietnamese', 'hebrew', 'ukrainian', 'greek', 'malay', 'czech', 'romanian', 'danish', 'hungarian', 'tamil', 'norwegian', 'thai', 'urdu', 'croatian', 'bulgarian', 'lithuanian', 'latin', 'maori', 'malayalam', 'welsh', 'slovak', 'telugu', 'persian', 'latvian', 'bengali', 'serbian', 'azerbaijani', 'slovenian', 'kannada', 'estonian', 'macedonian', 'breton', 'basque', 'icelandic', 'armenian', 'nepali', 'mongolian', 'bosnian', 'kazakh', 'albanian', 'swahili', 'galician', 'marathi', 'punjabi', 'sinhala', 'khmer', 'shona', 'yoruba', 'somali', 'afrikaans', 'occitan', 'georgian', 'belarusian', 'tajik', 'sindhi', 'gujarati', 'amharic', 'yiddish', 'lao', 'uzbek', 'faroese', 'haitian creole', 'pashto', 'turkmen', 'nynorsk', 'maltese', 'sanskrit', 'luxembourgish', 'myanmar', 'tibetan', 'tagalog', 'malagasy', 'assamese', 'tatar', 'hawaiian', 'lingala', 'hausa', 'bashkir', 'javanese', 'sundanese', 'cantonese']
TO_LANGUAGE_CODE = {**{n: c for c, n in zip(LANGUAGES, LANGUAGE_NAMES)},
                    "burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb",
                    "pushto": "ps", "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si",
                    "castilian": "es", "mandarin": "zh"}

# SuppressTokens("-1"): the non-speech symbol set of the multilingual vocabulary
# (openai-whisper tokenizer.non_speech_tokens; same list as HF generation_config
# suppress_tokens, minus the specials which are appended per vocabulary below).
NON_SPEECH_TOKENS = [1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503,
                     522, 542, 873, 893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268, 3536,
                     3846, 3961, 4183, 4667, 6585, 6647, 7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331, 12562,
                     13793, 14157, 14635, 15265, 15618, 16553, 16604, 18362, 18956, 20075, 21675, 22520, 26130, 26161,
                     26435, 28279, 29464, 31650, 32302, 32470, 36865, 42863, 47425, 49870, 50254]


# the same symbol set in the gpt2 vocabulary of the English-only models (*.en, n_vocab 51864): the ids differ because the
# vocabulary does (HF generation_config.suppress_tokens of whisper-*.en minus the specials).  From upstream knowledge,
# like the list above; a checkpoint's tokenizer.json (see non_speech_tokens_from) or generation_config.json overrides it.
NON_SPEECH_TOKENS_EN = [1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 357, 366,
                        438, 532, 685, 705, 796, 930, 1058, 1220, 1267, 1279, 1303, 1343, 1377, 1391, 1635, 1782, 1875,
                        2162, 2361, 2488, 3467, 4008, 4211, 4600, 4808, 5299, 5855, 6329, 7203, 9609, 9959, 10563, 10786,
                        11420, 11709, 11907, 13163, 13697, 13700, 14808, 15306, 16410, 16791, 17992, 19203, 19510, 20724,
                        22305, 22935, 27007, 30109, 30420, 33409, 34949, 40283, 40493, 40549, 47282, 49146]


def non_speech_tokens_from(encode):
    """The published construction of the SuppressTokens("-1") set from the vocabulary itself (openai-whisper
    `Tokenizer.non_speech_tokens`): speaker tags and non-speech annotations -- brackets, note symbols... -- are banned
    when they are a single token (note symbols: by their first token), with or without a leading space; " -" and " '"
    always.  `encode(text) -> ids` without special tokens."""
    symbols = list('"#()*+/:;<=>@[\\]^_`{|}~\u300c\u300d\u300e\u300f')
    symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} \u266a\u266a \u266a\u266a\u266a".split()
    misc = set("\u2669\u266a\u266b\u266c\u266d\u266e\u266f")
    out = {encode(" -")[0], encode(" '")[0]}
    for sym in symbols + sorted(misc):
        for toks in (encode(sym), encode(" " + sym)):
            if toks and (len(toks) == 1 or sym in misc):
                out.add(toks[0])
    return sorted(out)


@dataclass
class Tokenizer:
    n_vocab: int
    language: str = "en"
    task: str = "transcribe"
    hf: Optional[object] = None          # tokenizers.Tokenizer if available
    eot: int = 50257
    sot: int = 50258
    n_langs: int = 99
    translate: int = 0
    transcribe: int = 0
    sot_lm: int = 0
    sot_prev: int = 0
    no_speech: int = 0
    no_timestamps: int = 0
    timestamp_begin: int = 0
    blank_tokens: List[int] = field(default_factory=lambda: [220])   # encode(" ")
    multilingual: bool = True

    def __post_init__(self):
        # Published layout: the specials follow the ordinary tokens as <|endoftext|>, <|startoftranscript|>, the language
        # tokens, <|translate|>, <|transcribe|>, <|startoflm|>, <|startofprev|>, <|nospeech|>, <|notimestamps|> and the
        # 1501 timestamps.  The English-only (*.en, gpt2) vocabulary has one ordinary token fewer (EOT = 50256) and STILL
        # carries the 99 language tokens -- 50363 + 1501 = 51864 -- although its prompt is <|startoftranscript|> alone.
        if self.n_vocab == 51864:
            self.eot, self.sot, self.n_langs, self.multilingual = 50256, 50257, 99, False
        else:
            self.n_langs, self.multilingual = (100 if self.n_vocab >= 51866 else 99), True
        base = self.sot + 1 + self.n_langs
        self.translate, self.transcribe, self.sot_lm, self.sot_prev = base, base + 1, base + 2, base + 3
        self.no_speech, self.no_timestamps, self.timestamp_begin = base + 4, base + 5, base + 6
        assert self.n_vocab < 51864 or self.timestamp_begin + 1501 == self.n_vocab, "special-token layout does not fill the vocabulary"

    @property
    def is_multilingual(self):
        return self.multilingual

    def language_token(self, lang=None):
        lang = lang or self.language
        return self.sot + 1 + LANGUAGES.index(lang)

    def sot_sequence(self, lang=None, task=None):
        if not self.is_multilingual:
            return [self.sot]
        task = task or self.task
        return [self.sot, self.language_token(lang), self.transcribe if task == "transcribe" else self.translate]

    def suppress_tokens(self, extra=None):
        """ids banned at every step: non-speech symbols + transcribe/translate/sot/sot_prev/
        sot_lm/no_speech (published SuppressTokens construction)."""
        if extra is not None:
            base = extra
        elif self.hf is not None:           # the checkpoint's own vocabulary decides
            base = non_speech_tokens_from(lambda t: self.hf.encode(t, add_special_tokens=False).ids)
        else:
            base = NON_SPEECH_TOKENS if self.is_multilingual else NON_SPEECH_TOKENS_EN
        s = set(base)
        s.update([self.transcribe, self.translate, self.sot, self.sot_prev, self.sot_lm])
        if self.no_speech:
            s.add(self.no_speech)
        return sorted(t for t in s if 0 <= t < self.n_vocab)

    def decode(self, ids):
        ids = [t for t in ids if t < self.eot]
        if self.hf is not None:
            return self.hf.decode(ids, skip_special_tokens=True)
        return "".join(f" t{t}" for t in ids)

    def decode_token(self, t):
        """text of ONE token (how the reference groups tokens into words decodes them one by one); cached: a job decodes
        the same few thousand ids again and again, and with a real vocabulary every call goes through the HF decoder"""
        c = self.__dict__.setdefault("_tok_text", {})
        s = c.get(t)
        if s is None:
            s = c[t] = self.decode([t])
        return s

    def split_to_word_tokens(self, ids):
        """Groups text tokens into words: a token whose text starts with a space opens a
        new word (mlx_whisper_optimized_final.py:215-238)."""
        words, word_tokens = [], []
        for t in ids:
            s = self.decode_token(t)
            if not words or s.startswith(" "):
                words.append(s)
                word_tokens.append([t])
            else:
                words[-1] += s
                word_tokens[-1].append(t)
        return words, word_tokens


    def split_to_words(self, ids):
        """split_to_word_tokens for the hot host path: (word strings, token index bounds of the words [n_words + 1]) --
        the same grouping, with the per-token text and "opens a word" flag looked up in one dict each"""
        texts = self.__dict__.setdefault("_tok_text", {})
        opens = self.__dict__.setdefault("_tok_opens", {})
        words, bounds = [], []
        cur = None
        for k, t in enumerate(ids):
            s = texts.get(t)
            if s is None:
                s = self.decode_token(t)
            o = opens.get(t)
            if o is None:
                o = opens[t] = s.startswith(" ")
            if cur is None or o:
                if cur is not None:
                    words.append("".join(cur) if len(cur) > 1 else cur[0])
                bounds.append(k)
                cur = [s]
            else:
                cur.append(s)
        if cur is not None:
            words.append("".join(cur) if len(cur) > 1 else cur[0])
        bounds.append(len(ids))
        import numpy as np
        return words, np.asarray(bounds if words else [0], dtype=np.int64)


def get_tokenizer(n_vocab, language="en", task="transcribe", model_dir=None):
    hf = None
    if model_dir:
        import os
        p = os.path.join(model_dir, "tokenizer.json")
        if os.path.exists(p):
            from tokenizers import Tokenizer as HFTok
            hf = HFTok.from_file(p)
    return Tokenizer(n_vocab=n_vocab, language=language or "en", task=task or "transcribe", hf=hf)
