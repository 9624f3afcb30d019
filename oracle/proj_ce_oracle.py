"""CPU restatement (numpy, float64) of the text decoder's vocabulary projection,
token cross-entropy and arg-max.  TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product path.

Follows /root/reference/src/text_vae.py:
  :65     logits = self.proj(output.view(-1, output.size(-1)))      -> project()
  :77-79  F.cross_entropy(logits, targets, ignore_index=padidx,
          reduction='none').view(targets.size()); .sum(dim=0)        -> token_loss(), loss_rec()
  :81-101 accuracy: argmax(softmax(logits)) == targets over non-pad  -> accuracy()
and the gradients PyTorch's autograd derives for them              -> backward().

Pinned against the logits / loss_rec / accuracy the reference's own DAE
produced (tests/golden/textvae_golden.npz, tests/test_oracle_proj_ce.py).
"""
import numpy as np


def project(H, W, b):
    H, W = np.asarray(H, np.float64), np.asarray(W, np.float64)
    out = H @ W.T
    return out if b is None else out + np.asarray(b, np.float64)


def token_loss(logits, targets, padidx):
    """(rows,) cross-entropy per token; 0 where the target is the pad symbol."""
    logits = np.asarray(logits, np.float64)
    targets = np.asarray(targets).reshape(-1)
    m = logits.max(axis=1, keepdims=True)
    lse = np.log(np.exp(logits - m).sum(axis=1)) + m[:, 0]
    picked = logits[np.arange(len(targets)), np.clip(targets, 0, logits.shape[1] - 1)]
    return np.where(targets == padidx, 0.0, lse - picked)


def loss_rec(logits_tbv, targets_tb, padidx):
    """(B,) sum over time of the token losses (text_vae.py:79)."""
    T, B, V = logits_tbv.shape
    return token_loss(logits_tbv.reshape(T * B, V), targets_tb.reshape(-1), padidx).reshape(T, B).sum(axis=0)


def argmax_tokens(logits):
    return np.asarray(logits).argmax(axis=1)  # first maximum, as torch.argmax


def accuracy(logits_tbv, targets_tb, padidx):
    T, B, V = logits_tbv.shape
    pred = argmax_tokens(logits_tbv.reshape(T * B, V)).reshape(T, B)
    mask = targets_tb != padidx
    return float(((pred == targets_tb) & mask).sum() / mask.sum())


def backward(H, W, b, targets, padidx, gloss):
    """Gradients of sum_r gloss[r] * token_loss[r] w.r.t. H, W, b."""
    H, W = np.asarray(H, np.float64), np.asarray(W, np.float64)
    targets = np.asarray(targets).reshape(-1)
    logits = project(H, W, b)
    m = logits.max(axis=1, keepdims=True)
    p = np.exp(logits - m)
    p /= p.sum(axis=1, keepdims=True)
    keep = targets != padidx
    p[np.arange(len(targets))[keep], targets[keep]] -= 1.0
    p[~keep] = 0.0
    dlogits = p * np.asarray(gloss, np.float64).reshape(-1, 1)
    return dlogits @ W, dlogits.T @ H, dlogits.sum(axis=0)
