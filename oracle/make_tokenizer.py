"""Build the tiny Llama-style tokenizer fixture used by the prompt / splice / stopping tests.

The Mistral tokenizer is not available offline (no network), so parity of the integer/string
bookkeeping is pinned with a small seeded BPE that has the same *behaviour*
the reference's code depends on: `<s>` prepended to every encoded chunk, `</s>` as a single
special token, Metaspace word boundaries.  Output: tests/golden/tiny_tokenizer/tokenizer.json
(committed; regenerate with `python oracle/make_tokenizer.py`).
"""
import os
from tokenizers import Tokenizer, models, pre_tokenizers, decoders, trainers, processors

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "tiny_tokenizer")

CORPUS = [
    "[INST] <<SYS>>\nA chat between a curious user and an artificial intelligence assistant. "
    "The assistant gives helpful, detailed, and polite answers to the user's questions.\n<</SYS>>\n\n"
    "Please describe the video content in detail based on the provided information.\n [/INST]",
    "the person picks up a knife and cuts the onion on the board",
    "a player kicks the ball towards the goal and the keeper saves it",
    "the camera wearer opens the fridge door, takes the milk and closes it",
    "someone washes a plate in the sink then puts it on the rack",
    "0 1 2 3 4 5 6 7 8 9 . , ! ? ' \" : ; ( ) [ ] / < > _ - + = \n",
    "abcdefghijklmnopqrstuvwxyz ABCDEFGHIJKLMNOPQRSTUVWXYZ",
] * 4


def main():
    tok = Tokenizer(models.BPE(unk_token="<unk>", byte_fallback=False))
    tok.pre_tokenizer = pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="always")
    tok.decoder = decoders.Metaspace(replacement="▁", prepend_scheme="always")
    trainer = trainers.BpeTrainer(vocab_size=384, special_tokens=["<unk>", "<s>", "</s>"], show_progress=False)
    tok.train_from_iterator(CORPUS, trainer)
    tok.post_processor = processors.TemplateProcessing(
        single="<s> $A", pair="<s> $A $B", special_tokens=[("<s>", tok.token_to_id("<s>"))])
    os.makedirs(OUT, exist_ok=True)
    tok.save(os.path.join(OUT, "tokenizer.json"))
    print("vocab", tok.get_vocab_size())


if __name__ == "__main__":
    main()
