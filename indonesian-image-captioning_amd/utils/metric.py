"""Bookkeeping used around the hot path (reference: utils/metric.py:4-47) plus an NLTK-free corpus
BLEU for the validate() path (the reference calls nltk.translate.bleu_score.corpus_bleu,
trains/attention_scn.py:378; nltk is not installed in this image)."""
import math
from collections import Counter
from fractions import Fraction



class AverageMeter(object):
    """Most recent value, running sum, count and average of a metric."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def accuracy(scores, targets, k):
    """Top-k accuracy in percent: scores (N, V), targets (N,)."""
    batch_size = targets.size(0)
    _, ind = scores.topk(k, 1, True, True)
    correct = ind.eq(targets.view(-1, 1).expand_as(ind))
    return correct.view(-1).float().sum().item() * (100.0 / batch_size)


def binary_accuracy(score, targets):
    return (score >= 0.5).eq(targets >= 0.5).float().mean().cpu() * 100.0


def _ngrams(seq, n):
    return Counter(tuple(seq[i:i + n]) for i in range(len(seq) - n + 1))


def corpus_bleu(list_of_references, hypotheses, weights=(0.25, 0.25, 0.25, 0.25)):
    """Corpus-level BLEU as defined by Papineni et al. (2002) and computed by NLTK's ``corpus_bleu`` with
    its defaults (uniform 4-gram weights, no smoothing): clipped n-gram counts are summed over the corpus
    before the division, the brevity penalty uses the reference length closest to each hypothesis (ties
    -> the shorter one), and the score is 0 when there is no unigram overlap."""
    p_num, p_den = Counter(), Counter()
    hyp_len = ref_len = 0
    assert len(list_of_references) == len(hypotheses)
    for refs, hyp in zip(list_of_references, hypotheses):
        for n in range(1, len(weights) + 1):
            counts = _ngrams(hyp, n)
            max_ref = Counter()
            for ref in refs:
                for g, c in _ngrams(ref, n).items():
                    if c > max_ref[g]:
                        max_ref[g] = c
            p_num[n] += sum(min(c, max_ref[g]) for g, c in counts.items())
            p_den[n] += max(1, sum(counts.values()))
        hyp_len += len(hyp)
        ref_len += min((len(r) for r in refs), key=lambda rl: (abs(rl - len(hyp)), rl))
    if p_num[1] == 0:
        return 0.0
    if hyp_len == 0:
        return 0.0
    bp = 1.0 if hyp_len > ref_len else math.exp(1 - ref_len / hyp_len)
    s = 0.0
    for n, w in enumerate(weights, start=1):
        p = Fraction(p_num[n], p_den[n])
        if p == 0:   # NLTK (no smoothing) multiplies by a tiny epsilon-free 0 -> score 0 with a warning
            return 0.0
        s += w * math.log(p)
    return bp * math.exp(s)
