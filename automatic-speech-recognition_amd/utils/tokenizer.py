"""utils.tokenizer -- text <-> id tables (reference utils/tokenizer.py).

CharEncoder is pinned by golden G1.  SubwordEncoder / train_subword_tokenizer are restated on the
CURRENT `tokenizers` API: the reference's calls (`CharBPETokenizer(vocab_file=, merges_file=)`,
`tokenizer.save(path, "bpe")`, utils/tokenizer.py:41,53) fail under tokenizers>=0.8, so
subword tokenisation is parity-unpinned."""
import os
import string

SPECIAL_TOKENS = ['<PAD>', '<SOS>', '<EOS>', '<SPACE>']


def lookup_dicts(special_tokens):
    """token<->id tables: the four specials followed by A..Z (reference utils/tokenizer.py:6-24)."""
    tokens = list(special_tokens) + list(string.ascii_uppercase)
    return {t: i for i, t in enumerate(tokens)}, dict(enumerate(tokens))


class CharEncoder:
    """30-symbol character vocabulary (reference utils/tokenizer.py:87-117)."""

    def __init__(self):
        self.char2id, self.id2char = lookup_dicts(SPECIAL_TOKENS)
        self.token_to_id, self.id_to_token = self.char2id, self.id2char
        self.encode = self._encode_chars

    def get_vocab_size(self):
        return len(self.id2char)

    def _encode_chars(self, sentence, with_eos):
        space = self.char2id['<SPACE>']
        ids = [space if ch == ' ' else self.char2id[ch] for ch in sentence]
        return ids + [self.char2id['<EOS>']] if with_eos else ids


def train_subword_tokenizer(size, special_tokens, path):
    """Train a CharBPE vocabulary on <path>/corpus_all.txt and write bpe-vocab.json / bpe-merges.txt
    (reference utils/tokenizer.py:26-41)."""
    from tokenizers import CharBPETokenizer
    tok = CharBPETokenizer()
    tok.train([os.path.join(path, "corpus_all.txt")], vocab_size=size, min_frequency=2, show_progress=False,
              special_tokens=list(special_tokens[:3]) + ["<unk>"])
    tok.save_model(path, "bpe")
    return tok


class SubwordEncoder:
    """BPE subword vocabulary (reference utils/tokenizer.py:43-85); <EOS> has id 2."""

    def __init__(self, path='subword/'):
        from tokenizers import CharBPETokenizer
        self.subword_tokenizer = CharBPETokenizer(vocab=os.path.join(path, "bpe-vocab.json"),
                                                  merges=os.path.join(path, "bpe-merges.txt"))
        self.encode = self._encode_subwords
        n = self.get_vocab_size()
        self.id_to_token = {i: self.subword_tokenizer.id_to_token(i) for i in range(n)}
        self.token_to_id = {t: i for i, t in self.id_to_token.items()}

    def get_vocab_size(self):
        return self.subword_tokenizer.get_vocab_size()

    def _encode_subwords(self, sentence, with_eos):
        ids = list(self.subword_tokenizer.encode(sentence).ids)
        return ids + [2] if with_eos else ids
