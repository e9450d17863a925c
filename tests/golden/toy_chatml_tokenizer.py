"""A tiny character-level ChatML tokenizer (tokenizers WordLevel) shared by the golden generator and the tests.

"\\n" has id 198 like in the Qwen2 vocabulary (preprocess_qwen hard-codes 198, train/train.py:575); <|im_start|> and
<|im_end|> are the two additional special tokens.  ``Tok4`` restores the two transformers-4.x behaviours the reference's
preprocess_qwen relies on and transformers 5.x changed (additional_special_tokens_ids; apply_chat_template returning ids)."""
from tokenizers import Regex, Tokenizer, models, pre_tokenizers
from transformers import PreTrainedTokenizerFast


def build(cls=PreTrainedTokenizerFast):
    chars = [chr(i) for i in range(32, 127)]
    vocab = {"<unk>": 0}
    for i in range(1, 200):
        vocab[f"<fill{i}>"] = i
    for k, c in enumerate(chars):
        del vocab[f"<fill{k + 1}>"]
        vocab[c] = k + 1
    del vocab["<fill198>"]
    vocab["\n"] = 198
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split(Regex(r"[\s\S]"), behavior="isolated")
    return cls(tokenizer_object=tok, unk_token="<unk>", additional_special_tokens=["<|im_start|>", "<|im_end|>"])


class Tok4(PreTrainedTokenizerFast):
    @property
    def additional_special_tokens_ids(self):
        return self.convert_tokens_to_ids(["<|im_start|>", "<|im_end|>"])

    def apply_chat_template(self, conversation, **kw):
        kw.setdefault("tokenize", True)
        kw.setdefault("return_dict", False)
        return super().apply_chat_template(conversation, **kw)
