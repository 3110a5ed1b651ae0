"""Test infrastructure: golden corpus-BLEU values from NLTK itself (the function the reference calls at
trains/attention_scn.py:23,377 with its defaults).  NLTK is not importable by the project interpreter;
the image carries a second interpreter (/opt/conda/bin/python3.9, nltk 3.6.5) that has it:

    /opt/conda/bin/python3.9 oracle/gen_bleu_golden.py        # -> tests/golden/bleu_nltk.json

Cases mimic validate(): token-id hypotheses, several references per hypothesis, with <start>/<pad>
already stripped; they include short hypotheses, missing higher-order matches and empty overlap."""
import json
import os
import random
import warnings

from nltk.translate.bleu_score import corpus_bleu
import nltk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_case(rng, n_hyp, vocab, cpi, lo, hi, noise):
    refs_all, hyps = [], []
    for _ in range(n_hyp):
        base = [rng.randrange(1, vocab) for _ in range(rng.randint(lo, hi))]
        refs = []
        for _ in range(cpi):
            r = [w if rng.random() > 0.3 else rng.randrange(1, vocab) for w in base]
            if rng.random() < 0.5 and len(r) > 2:
                del r[rng.randrange(len(r))]
            refs.append(r)
        hyp = [w if rng.random() > noise else rng.randrange(1, vocab) for w in base]
        if rng.random() < 0.4 and len(hyp) > 1:
            hyp = hyp[:rng.randint(1, len(hyp))]
        refs_all.append(refs)
        hyps.append(hyp)
    return refs_all, hyps


def main():
    rng = random.Random(20240607)
    cases = []
    specs = [(1, 12, 1, 8, 8, 0.0), (1, 12, 1, 8, 8, 0.2), (5, 30, 5, 4, 14, 0.2), (20, 50, 5, 3, 20, 0.4),
             (50, 200, 5, 5, 25, 0.3), (8, 10, 3, 1, 5, 0.5), (3, 1000, 5, 6, 9, 1.0), (40, 60, 2, 2, 30, 0.1),
             (10, 20, 5, 1, 3, 0.3), (100, 500, 5, 8, 22, 0.25)]
    for spec in specs:
        refs, hyps = make_case(rng, *spec)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            score = corpus_bleu(refs, hyps)
            score2 = corpus_bleu(refs, hyps, weights=(0.5, 0.5))
        cases.append({"references": refs, "hypotheses": hyps, "bleu4": float(score), "bleu2": float(score2)})
    # hand-made corner cases
    corner = [([[[1, 2, 3, 4, 5, 6, 7, 8]]], [[1, 2, 3, 4, 5, 6, 7, 8]]),
              ([[[1, 2, 3, 4, 5, 6, 7, 8]]], [[9, 9, 9, 9]]),
              ([[[1, 2, 3, 4], [1, 2, 3, 4, 5, 6]]], [[1, 2, 3, 4, 5]]),        # closest-length tie -> shorter
              ([[[1, 2, 3, 4, 5, 6, 7]]], [[1, 2, 3]]),                           # no 4-gram possible
              ([[[1, 2, 2, 2, 3]], [[4, 5, 6, 7, 8, 9]]], [[2, 2, 2, 2, 2], [4, 5, 6, 7, 8, 9]])]
    for refs, hyps in corner:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cases.append({"references": refs, "hypotheses": hyps, "bleu4": float(corpus_bleu(refs, hyps)),
                          "bleu2": float(corpus_bleu(refs, hyps, weights=(0.5, 0.5)))})
    out = {"generator": "nltk %s corpus_bleu, defaults (no smoothing)" % nltk.__version__, "cases": cases}
    with open(os.path.join(ROOT, "tests", "golden", "bleu_nltk.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote %d cases" % len(cases), [round(c["bleu4"], 6) for c in cases])


if __name__ == "__main__":
    main()
