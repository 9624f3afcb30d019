"""Input noise for the denoising text auto-encoder.

Mirrors the live part of /root/reference/src/noise.py: ``noisy`` only ever
reaches ``word_drop`` (text_vae.py:70 calls it with drop_prob=0.3 and every
other probability 0; the other branches of noise.py:39-48 reference an
undefined ``vocab`` and cannot run).  ``word_drop`` here stays on the device:
on a GPU one library kernel (a wavefront per sentence: keep flags from a
counter-based generator, or from a mask drawn with a given torch generator,
and a ballot-rank compaction), instead of the reference's per-sentence
``.tolist()`` round trip (noise.py:12-22); tensors on the CPU take the
equivalent torch formulation (one stable sort).
"""
import torch


def word_drop(padidx: int, x: torch.Tensor, p: float, generator: torch.Generator = None) -> torch.Tensor:
    """x is (T, B) token ids; every token but the first of each column is kept
    with probability 1-p; kept tokens move up, the tail is filled with padidx."""
    T, B = x.shape
    if x.is_cuda:
        from . import functional as Fn

        keep = None
        if generator is not None:   # a caller-chosen random stream: draw the mask from it, compact in the kernel
            keep = torch.rand((T, B), device=generator.device, generator=generator) > p
        return Fn.word_drop(x, padidx, p, keep)
    keep = torch.rand((T, B), device=x.device, generator=generator) > p
    keep[0, :] = True
    # stable sort of the "dropped" flag moves kept tokens to the front, order preserved
    order = torch.sort((~keep).to(torch.int8), dim=0, stable=True).indices
    gathered = torch.gather(x, 0, order)
    kept_count = keep.sum(dim=0, keepdim=True)
    pos = torch.arange(T, device=x.device).unsqueeze(1)
    return torch.where(pos < kept_count, gathered, torch.full_like(gathered, padidx)).contiguous()


def noisy(padidx, x, drop_prob, blank_prob=0, sub_prob=0, shuffle_dist=0, generator=None):
    if blank_prob > 0 or sub_prob > 0 or shuffle_dist > 0:
        raise NotImplementedError("only word_drop is reachable in the reference (noise.py:39-48)")
    if drop_prob > 0:
        x = word_drop(padidx, x, drop_prob, generator)
    return x
