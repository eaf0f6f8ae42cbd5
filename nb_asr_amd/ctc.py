"""The step after the forward pass: log_softmax, length mapping, greedy CTC decoding.

Mirrors what the reference's ``Trainer.step`` / ``Trainer.decode`` do with the model output
(``training/torch/trainer.py:217-219, 229-247``): ``output = F.log_softmax(output, dim=2)``,
``output_len = audio_len // 4``, then decoding.  The reference decodes with ``ctcdecode``'s beam search (third-party C++,
not in its tree); the greedy decoder here is the width-1 case with the same blank (class 0) and collapse rule.
"""
import torch

from . import hip


def output_lengths(audio_len):
    """Frames of the model output that belong to each utterance: ``audio_len // 4`` (trainer.py:219).  Note this is the
    trainer's convention, not ceil(ceil(T/2)/2): for T not divisible by 4 the last partial output frame is ignored."""
    return torch.div(audio_len, 4, rounding_mode='floor')


def log_softmax(logits):
    """(B, T', C) float32 on a HIP device -> log-probabilities, same shape."""
    return hip.ctc_postprocess(logits, None, True, False)[0]


def greedy_decode(logits, audio_len=None, blank=0, return_log_probs=False):
    """Greedy CTC decoding of model logits.

    ``audio_len``: per-utterance input lengths in frames (tensor or sequence); outputs beyond ``audio_len // 4`` are ignored.
    Returns a list of 1-D int32 CPU tensors (one token sequence per utterance), plus the log-probabilities if requested."""
    lengths = None
    if audio_len is not None:
        lengths = output_lengths(torch.as_tensor(audio_len)).to(device=logits.device, dtype=torch.int32).contiguous()
    log_probs, tokens, counts = hip.ctc_postprocess(logits.contiguous(), lengths, return_log_probs, True, blank)
    tokens, counts = tokens.cpu(), counts.cpu()
    seqs = [tokens[i, : int(counts[i])] for i in range(tokens.shape[0])]
    return (seqs, log_probs) if return_log_probs else seqs
