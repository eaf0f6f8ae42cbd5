"""CPU restatement of the reference's validation decode step (TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else).

Reference: training/torch/trainer.py:229-247 (``Trainer.decode``) --
    beams, _, _, beams_len = CTCBeamDecoder(vocab, beam_width=12, log_probs_input=True).decode(output, output_len)
    top beam -> PhonemeEncoder.fold_encoded(., 39) (training/torch/encoder.py:64-75) -> torch_edit_distance.compute_wer(
    top_beams, targets, top_beams_len, targets_len, blank=[0], sep=[]) -> mean.

Pinning status, per function:
* ``fold_table`` / ``fold_encoded``: PINNED -- the reference's PhonemeEncoder is pure Python and was run in the build
  container (tests/golden/make_decode_golden.py -> tests/golden/decode_fixtures.npz).  Note the reference's quirk, reproduced
  here: ``fold_encoded`` relabels IN PLACE, one source index after the other, so a label that was just moved to a higher
  index is moved again when the loop reaches that index ('cl' -> sil(31) -> 24 = 'ng'; 'el' -> l(21) -> 15 = 'hh';
  'en' -> n(23) -> 17 = 'ih'; 'epi' -> 31 -> 24).
* ``ctc_beam_search``: PARITY UNPINNED against the reference's dependency -- ``ctcdecode`` (parlance/ctcdecode @ 9a20e00,
  setup.py:49) is a C++ extension that is neither vendored nor installed.  This restates its published prefix beam search
  (ctc_beam_search_decoder.cpp: vocabulary pruned to the ``cutoff_top_n`` = 40 most probable classes per frame,
  cutoff_prob = 1.0, no language model, prefixes merged in a trie, per prefix log P(blank-ending) / log P(non-blank-ending),
  beams ordered by score then by last character) and is anchored on first principles instead: for small cases and a beam wide
  enough to be exhaustive its best beam must equal the labelling of maximum total CTC probability found by enumerating every
  alignment (``ctc_labelling_log_probs``).
* ``ctc_loss``: PINNED -- the reference's loss is a call into ATen (``F.ctc_loss``, trainer.py:36-42), and the same call on
  CPU tensors is the oracle, exactly as the reference would compute it on its CPU path.
* ``edit_distance`` / ``error_rate``: PARITY UNPINNED against ``torch_edit_distance`` (1ytic/pytorch-edit-distance, setup.py,
  not installed): token-level Levenshtein distance after removing blanks, divided by the reference length -- the standard
  phoneme error rate.
"""
import itertools
import math

import numpy as np

F32 = np.float32
NEG = -np.finfo(np.float32).max            # ctcdecode's "-infinity" (NUM_FLT_INF = numeric_limits<float>::max())


def log_sum_exp(x, y):
    """ctcdecode decoder_utils.h log_sum_exp, in float32."""
    x, y = F32(x), F32(y)
    if x <= NEG:
        return y
    if y <= NEG:
        return x
    m = max(x, y)
    return F32(F32(np.log(F32(np.exp(F32(x - m))) + F32(np.exp(F32(y - m))))) + m)


class _Prefix:
    __slots__ = ('char', 'parent', 'children', 'b_prev', 'nb_prev', 'b_cur', 'nb_cur', 'score', 'exists')

    def __init__(self, char, parent):
        self.char, self.parent, self.children = char, parent, {}
        self.b_prev = self.nb_prev = self.b_cur = self.nb_cur = self.score = NEG
        self.exists = True

    def child(self, c):
        node = self.children.get(c)
        if node is None:
            node = self.children[c] = _Prefix(c, self)
        elif not node.exists:
            node.exists = True
            node.b_prev = node.nb_prev = node.b_cur = node.nb_cur = NEG
        return node

    def tokens(self):
        out, node = [], self
        while node.parent is not None:
            out.append(node.char)
            node = node.parent
        return out[::-1]


def _collect(node, out):
    if node.exists:
        node.b_prev, node.nb_prev = node.b_cur, node.nb_cur
        node.b_cur = node.nb_cur = NEG
        node.score = log_sum_exp(node.b_prev, node.nb_prev)
        out.append(node)
    for ch in node.children.values():
        _collect(ch, out)


def _order(p):
    return (-float(p.score), p.char)       # prefix_compare: score descending, then last character ascending (root = -1)


def ctc_beam_search(log_probs, beam_width=12, blank=0, cutoff_top_n=40):
    """log_probs: (T, C) float32 log-probabilities of ONE utterance.  Returns [(tokens, -score)], best first, at most
    ``beam_width`` entries (ctcdecode returns the negated log score: lower is better)."""
    lp_all = np.asarray(log_probs, dtype=np.float32)
    n_cls = lp_all.shape[1] if lp_all.ndim == 2 else 0
    root = _Prefix(-1, None)
    root.score = root.b_prev = F32(0.0)
    prefixes = [root]
    for t in range(lp_all.shape[0]):
        lp = lp_all[t]
        classes = sorted(range(n_cls), key=lambda c: (-float(lp[c]), c))
        if cutoff_top_n < n_cls:
            classes = classes[:cutoff_top_n]
        for c in classes:
            for p in prefixes[:beam_width]:
                if c == blank:
                    p.b_cur = log_sum_exp(p.b_cur, F32(lp[c] + p.score))
                    continue
                if c == p.char:
                    p.nb_cur = log_sum_exp(p.nb_cur, F32(lp[c] + p.nb_prev))
                new = p.child(c)
                log_p = NEG
                if c == p.char and p.b_prev > NEG:
                    log_p = F32(lp[c] + p.b_prev)
                elif c != p.char:
                    log_p = F32(lp[c] + p.score)
                new.nb_cur = log_sum_exp(new.nb_cur, log_p)
        prefixes = []
        _collect(root, prefixes)
        prefixes.sort(key=_order)
        for p in prefixes[beam_width:]:
            p.exists = False
        prefixes = prefixes[:beam_width]
    prefixes.sort(key=_order)
    return [(p.tokens(), -float(p.score)) for p in prefixes[:beam_width]]


def ctc_labelling_log_probs(log_probs, blank=0):
    """Exhaustive CTC: {labelling (tuple): log of the summed probability of all its alignments}.  C**T alignments."""
    lp = np.asarray(log_probs, dtype=np.float64)
    n_frames, n_cls = lp.shape
    total = {}
    for path in itertools.product(range(n_cls), repeat=n_frames):
        score = sum(lp[t, c] for t, c in enumerate(path))
        lab = tuple(c for i, c in enumerate(path) if c != blank and (i == 0 or c != path[i - 1]))
        total[lab] = np.logaddexp(total.get(lab, -math.inf), score)
    return total


# ---- phoneme folding (reference training/timit_folding.txt via training/torch/encoder.py) ---------------------------------

def class_lists(folding_rows):
    """folding_rows: [(p61, p48, p39)] (empty string: dropped).  Sorted class lists of the three label sets (encoder.py:38-39)."""
    return [sorted({row[i] for row in folding_rows if row[i]}) for i in range(3)]


def index_mapping(folding_rows, src, dst):
    """encoder.py:41-49: {source index: destination index}, 0 = blank, classes numbered from 1 in sorted order."""
    lists = class_lists(folding_rows)
    fold = {row[src]: row[dst] for row in folding_rows}
    mapping = {0: 0}
    for i, ph in enumerate(lists[src]):
        to = fold[ph]
        mapping[i + 1] = lists[dst].index(to) + 1 if to else 0
    return mapping


def fold_encoded(tokens, mapping):
    """encoder.py:71-73, the in-place sequential relabelling (see the header for what it does to 'cl', 'el', 'en', 'epi')."""
    out = np.array(tokens, copy=True)
    for old, new in mapping.items():
        out[out == old] = new
    return out


def fold_table(mapping):
    """The lookup table equivalent to ``fold_encoded`` for labels 0 .. len(mapping) - 1."""
    return fold_encoded(np.arange(len(mapping)), mapping)


# ---- error rate -----------------------------------------------------------------------------------------------------------

def edit_distance(hyp, ref):
    """Levenshtein distance between two token lists (unit costs)."""
    prev = list(range(len(ref) + 1))
    for i, h in enumerate(hyp, 1):
        cur = [i] + [0] * len(ref)
        for j, r in enumerate(ref, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (h != r))
        prev = cur
    return prev[len(ref)]


def error_counts(hyp, hyp_len, ref, ref_len, blank=0, table=None):
    """Per utterance (distance, reference length) after the optional label table and the removal of blanks."""
    out = []
    for h, hn, r, rn in zip(hyp, hyp_len, ref, ref_len):
        h, r = [int(v) for v in h[: int(hn)]], [int(v) for v in r[: int(rn)]]
        if table is not None:
            h, r = [int(table[v]) for v in h], [int(table[v]) for v in r]
        h, r = [v for v in h if v != blank], [v for v in r if v != blank]
        out.append((edit_distance(h, r), len(r)))
    return out


# ---- loss -------------------------------------------------------------------------------------------------------------------

def ctc_loss(log_probs, output_len, targets, targets_len, reduce=True):
    """Reference training/torch/trainer.py:36-42 ``get_loss()``: log_probs (B, T', C) torch CPU tensor.  Returns the batch mean
    of nll / output_len (``reduce=False``: the per-utterance values before the mean)."""
    import torch
    import torch.nn.functional as F
    output_len = torch.as_tensor(output_len, dtype=torch.long)
    targets_len = torch.as_tensor(targets_len, dtype=torch.long)
    loss = F.ctc_loss(log_probs.permute(1, 0, 2), torch.as_tensor(targets, dtype=torch.long), output_len, targets_len,
                      reduction='none', zero_infinity=True)
    loss = loss / output_len
    return loss.mean() if reduce else loss


def ctc_loss_grad(logits, output_len, targets, targets_len):
    """(loss, d loss / d logits) of ``ctc_loss(log_softmax(logits), ...)`` by torch autograd on CPU: what the reference's
    ``loss.backward()`` (trainer.py:220-223, without the weight-norm term) hands to the model."""
    import torch
    logits = logits.detach().clone().requires_grad_(True)
    loss = ctc_loss(torch.log_softmax(logits, dim=2), output_len, targets, targets_len)
    loss.backward()
    return loss.detach(), logits.grad

