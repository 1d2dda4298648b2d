"""id <-> piece table for Reverb's `rev_bpe` tokenizer.

At inference the reference only maps ids to pieces through `tk.units.txt`
(asr/wenet/text/rev_bpe_tokenizer.py:10-82, char_tokenizer.py:71-76,
utils/file_utils.py:61-68); the sentencepiece model is needed only for text -> ids, which the
recognize_wav path never calls, so it is loaded lazily and only if asked for."""
from __future__ import annotations

from typing import Dict, List, Tuple


def read_symbol_table(path: str) -> Dict[str, int]:
    table: Dict[str, int] = {}
    with open(path, "r", encoding="utf8") as fin:
        for line in fin:
            parts = line.strip().split()
            assert len(parts) == 2, f"bad symbol table line: {line!r}"
            table[parts[0]] = int(parts[1])
    return table


class RevBpeTokenizer:
    def __init__(self, bpe_model, symbol_table, non_lang_syms=None, split_with_space=False,
                 connect_symbol: str = "", unk: str = "<unk>", full_config: dict | None = None):
        self._symbol_table = symbol_table if isinstance(symbol_table, dict) else read_symbol_table(symbol_table)
        self.char_dict = {v: k for k, v in self._symbol_table.items()}
        self.connect_symbol = connect_symbol
        self.unk = unk
        self._model = bpe_model
        self.bpe_model = None
        cfg = full_config or {}
        self.remove_sw = cfg.get("remove_sw", True)
        self.replace_unk_as_unknown = cfg.get("replace_unk_as_unknown", True)

    @property
    def symbol_table(self) -> Dict[str, int]:
        return self._symbol_table

    def vocab_size(self) -> int:
        return len(self.char_dict)

    def ids2tokens(self, ids: List[int]) -> List[str]:
        return [self.char_dict[w] for w in ids]

    def tokens2ids(self, tokens: List[str]) -> List[int]:
        out = []
        for t in tokens:
            if t in self._symbol_table:
                out.append(self._symbol_table[t])
            elif self.unk in self._symbol_table:
                out.append(self._symbol_table[self.unk])
        return out

    def tokens2text(self, tokens: List[str]) -> str:
        return self.connect_symbol.join(tokens).replace("▁", " ").strip()

    def detokenize(self, ids: List[int]) -> Tuple[str, List[str]]:
        tokens = self.ids2tokens(ids)
        return self.tokens2text(tokens), tokens

    def text2tokens(self, line: str) -> List[str]:
        if self.bpe_model is None:
            import sentencepiece as spm
            self.bpe_model = spm.SentencePieceProcessor()
            self.bpe_model.load(self._model)
        line = line.strip()
        if self.remove_sw:
            line = line.replace("<sw>", "").replace("  ", " ").strip()
        if self.replace_unk_as_unknown:
            line = line.replace("<unk>", "<unknown>")
        return self.bpe_model.encode(line, out_type=str)

    def tokenize(self, line: str):
        tokens = self.text2tokens(line)
        return tokens, self.tokens2ids(tokens)
