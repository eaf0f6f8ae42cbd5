"""The step after the forward pass: log_softmax, length mapping, CTC decoding, phoneme error rate.

Mirrors what the reference's ``Trainer.step`` / ``Trainer.decode`` do with the model output
(``training/torch/trainer.py:217-219, 229-247``): ``output = F.log_softmax(output, dim=2)``,
``output_len = audio_len // 4``, then ``CTCBeamDecoder(vocab, beam_width=12, log_probs_input=True).decode`` (ctcdecode,
third-party C++, not in the reference's tree), ``PhonemeEncoder.fold_encoded(., 39)`` on the best beam and on the targets,
``torch_edit_distance.compute_wer`` and the mean.  All of it runs on the device: ``beam_decode`` (prefix beam search),
``fold_table`` (the 48 -> 39 label table, with the reference's relabelling order), ``error_rates`` (label table + blank
removal + Levenshtein distance) and ``decode_per`` = the whole of ``Trainer.decode``.  ``greedy_decode`` is the cheap
width-1 relative (per-frame argmax, collapse, drop blank).
"""
import torch

from . import hip
from .phonemes import fold_table  # noqa: F401  (part of this module's interface)


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


def _lengths(lengths, batch, device):
    if lengths is None:
        return None
    lengths = torch.as_tensor(lengths)
    if lengths.numel() != batch:
        raise ValueError(f'expected {batch} lengths, got {lengths.numel()}')
    return lengths.to(device=device, dtype=torch.int32).contiguous()


def beam_decode(log_probs, output_len=None, beam_width=12, blank=0, cutoff_top_n=40):
    """CTC prefix beam search over log-probabilities (B, T', C), the reference's
    ``CTCBeamDecoder(vocab, beam_width=12, log_probs_input=True).decode(output, output_len)`` (trainer.py:71,237).

    ``output_len``: valid output frames per utterance (``output_lengths(audio_len)``), tensor or sequence, or None.
    Returns ``(beams, scores, out_len)`` on the device like ctcdecode does (minus its per-token time steps): beams
    (B, beam_width, T') int32 best first (entries beyond ``out_len`` are 0), scores (B, beam_width) = -log P (lower is
    better), out_len (B, beam_width) int32."""
    if log_probs.dim() != 3:
        raise ValueError(f'log_probs must be (batch, frames, classes), got {tuple(log_probs.shape)}')
    return hip.ctc_beam_search(log_probs.contiguous(), _lengths(output_len, log_probs.shape[0], log_probs.device),
                               beam_width, blank, cutoff_top_n)


def error_rates(hyp, hyp_len, ref, ref_len, blank=0, table=None):
    """Per-utterance token error rate, ``torch_edit_distance.compute_wer(hyp, ref, hyp_len, ref_len, blank, sep=[])``
    (trainer.py:245): Levenshtein distance / reference length, after mapping both sides through ``table`` (optional int32
    label table, e.g. ``fold_table()``) and dropping ``blank``.  Returns a float32 device tensor (B); an empty reference
    gives inf (nan when the hypothesis is empty too), as the division does in the reference."""
    dev = hyp.device
    b = hyp.shape[0]
    counts = hip.token_error_counts(hyp.to(torch.int32).contiguous(), _lengths(hyp_len, b, dev), ref.to(device=dev, dtype=torch.int32).contiguous(),
                                    _lengths(ref_len, b, dev), None if table is None else table.to(device=dev, dtype=torch.int32).contiguous(),
                                    blank)
    if bool((counts[:, 0] < 0).any()):
        raise hip.HipError('error_rates: a sequence exceeds 2048 tokens or holds a label outside the table')
    return counts[:, 0].float() / counts[:, 1].float()


def decode_per(log_probs, output_len, targets, targets_len, beam_width=12, fold_to=39, num_classes=48):
    """``Trainer.decode`` (trainer.py:229-247): best beam and targets folded to ``fold_to`` phonemes, error rate per
    utterance, mean over the batch.  ``log_probs`` (B, T', num_classes + 1) on the device; targets (B, L) labels of the
    ``num_classes`` set (0 = blank / padding).  Returns a 0-dim float32 device tensor."""
    beams, _, beams_len = beam_decode(log_probs, output_len, beam_width=beam_width)
    table = fold_table(num_classes, fold_to).to(log_probs.device) if fold_to < num_classes else None
    per = error_rates(beams[:, 0].contiguous(), beams_len[:, 0].contiguous(), targets, targets_len, blank=0, table=table)
    return per.mean()


def ctc_loss(log_probs, output_len, targets, targets_len, blank=0):
    """The reference's loss value (``get_loss()``, trainer.py:36-42): ``F.ctc_loss(log_probs.permute(1, 0, 2), targets, output_len,
    targets_len, reduction='none', zero_infinity=True) / output_len`` averaged over the batch -- what ``Trainer.step`` reports
    for validation and test batches.  ``log_probs`` (B, T', C) stays batch-major on the device.  Forward value only (validation / test); the
    training step uses ``training_loss`` below."""
    dev = log_probs.device
    b = log_probs.shape[0]
    per = hip.ctc_loss(log_probs.contiguous(), _lengths(output_len, b, dev), targets.to(device=dev, dtype=torch.int32).contiguous(),
                       _lengths(targets_len, b, dev), blank, divide_by_length=True)
    return per.mean()


def evaluation_step(model, audio, audio_len, targets, targets_len, beam_width=12):
    """``Trainer.step(inputs, training=False)`` followed by ``Trainer.decode`` (trainer.py:207-247) for one batch:
    forward -> log_softmax -> ``output_len = audio_len // 4`` -> loss value and phoneme error rate.
    ``audio`` (B, 80, T) on the model's device.  Returns ``(loss, per)`` as 0-dim device tensors."""
    with torch.no_grad():
        log_probs = log_softmax(model(audio))
    output_len = output_lengths(torch.as_tensor(audio_len))
    loss = ctc_loss(log_probs, output_len, targets, targets_len)
    per = decode_per(log_probs, output_len, targets, targets_len, beam_width=beam_width)
    return loss, per


def evaluate(model, batches, beam_width=12):
    """The reference's validation / test loop (trainer.py:150-158, 190-200): running averages of the per-batch loss and
    phoneme error rate (its ``AvgMeter``: every batch weighs the same).  ``batches`` yields
    ``((audio, audio_len), (targets, targets_len))`` like the reference's data loaders.  Returns ``(loss, per)`` floats."""
    was_training = model.training
    model.eval()
    n, loss_avg, per_avg = 0, 0.0, 0.0
    try:
        for (audio, audio_len), (targets, targets_len) in batches:
            loss, per = evaluation_step(model, audio, audio_len, targets, targets_len, beam_width)
            loss, per = loss.item(), per.item()
            if n == 0:
                loss_avg, per_avg = loss, per
            else:
                loss_avg = loss_avg * (n / (n + 1)) + loss / (n + 1)
                per_avg = per_avg * (n / (n + 1)) + per / (n + 1)
            n += 1
    finally:
        model.train(was_training)
    return loss_avg, per_avg


def ctc_loss_and_grad(log_probs, output_len, targets, targets_len, blank=0):
    """``ctc_loss`` together with the gradient of that scalar with respect to the LOGITS (``log_probs = log_softmax(logits)``):
    what the reference's ``loss.backward()`` hands to the model (trainer.py:220-223, without its weight-norm term).  Returns
    ``(loss, grad_logits)``; ``training_loss`` wraps it as an autograd function."""
    dev = log_probs.device
    b = log_probs.shape[0]
    per, grad = hip.ctc_loss_grad(log_probs.contiguous(), _lengths(output_len, b, dev), targets.to(device=dev, dtype=torch.int32).contiguous(),
                                  _lengths(targets_len, b, dev), blank)
    return per.mean(), grad


class _TrainingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, output_len, targets, targets_len, blank):
        loss, grad = ctc_loss_and_grad(log_softmax(logits.detach().contiguous()), output_len, targets, targets_len, blank)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


def training_loss(logits, output_len, targets, targets_len, blank=0):
    """The trainer's loss on the LOGITS of a training-mode forward (trainer.py:215-223: ``log_softmax`` -> ``get_loss`` -> mean), attached
    to the autograd graph: ``training_loss(model.train()(x), ...).backward()`` reaches every parameter.  Loss and gradient with respect
    to the logits come from one HIP kernel pair (nbasr_ctc_loss_grad); add the reference's weight-norm term with torch if wanted."""
    return _TrainingLoss.apply(logits, output_len, targets, targets_len, blank)
