#!/usr/bin/env python3
"""Train the toy SentencePiece model the evaluation goldens use (TEST INFRASTRUCTURE, build container only).

The reference's evaluate() turns token ids into text with `spm.DecodePieces` before the edit distance
(src/monitor/metric.py:49-87); its own model (data/valid_train_en_unigram150.model) is a reference asset that never
travels.  This script trains a unigram model of the same vocabulary size (367 pieces incl. <unk>/<s>/</s>) on a seeded
synthetic corpus and writes

    tests/golden/toy_spm.model        (data fixture, ~10 KB)
    tests/golden/toy_spm_units.txt    (the `spm_mapping` file: "<piece> <id>" per line, 365 lines, layout of
                                       data/valid_train_en_unigram150_units.txt)

so that the reference (oracle/make_goldens.py) and the build compute CER/WER over identical piece strings.
"""
import io
import random
from pathlib import Path

import sentencepiece as spm

OUT = Path(__file__).resolve().parents[1] / "tests" / "golden"


def main():
    rng = random.Random(7)
    syll = [c + v for c in "bdfgklmnprstvz" for v in "aeiou"]
    words = ["".join(rng.choice(syll) for _ in range(rng.randint(1, 4))) for _ in range(600)]
    sents = [" ".join(rng.choice(words) for _ in range(rng.randint(3, 12))) for _ in range(4000)]
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(sents), model_writer=model, vocab_size=367, model_type="unigram",
                                   character_coverage=1.0, input_sentence_size=0, shuffle_input_sentence=False, num_threads=1,
                                   normalization_rule_name="identity", minloglevel=2)
    blob = model.getvalue()
    (OUT / "toy_spm.model").write_bytes(blob)
    sp = spm.SentencePieceProcessor(model_proto=blob)
    assert sp.GetPieceSize() == 367
    # id2units of the build/reference = [<s>] + units file + [</s>] (src/pretrain_interface.py:33-48): 365 lines, token id i
    # (1..365) -> line i-1.  <unk> first (as in the reference's file), then the learned pieces; <s>, </s> excluded.
    pieces = [sp.IdToPiece(i) for i in range(367) if sp.IdToPiece(i) not in ("<s>", "</s>")][:365]
    with open(OUT / "toy_spm_units.txt", "w") as f:
        for i, p in enumerate(pieces):
            f.write(f"{p} {i + 1}\n")
    print("toy_spm.model", len(blob), "bytes;", len(pieces), "units; e.g.", pieces[:6],
          "->", repr(sp.DecodePieces(pieces[1:6])))


if __name__ == "__main__":
    main()
