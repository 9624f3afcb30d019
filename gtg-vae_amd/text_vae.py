"""Text branch of the joint model (denoising LSTM auto-encoder).

Mirrors /root/reference/src/text_vae.py:22-132 (``TextModel``, ``DAE``): same
sub-module names and shapes, hence the same ``state_dict`` keys
(``embed, proj, E, G, hgen, z2emb``).  On a GPU the embedding lookup with its
dropout (and the decoder's added latent row), its scatter-add backward, the
output dropout and the word-drop noise are library kernels (``functional.
embed_dropout``, ``dropout``, ``word_drop``).  The LSTMs keep their parameters in ``nn.LSTM`` modules (same
``state_dict`` keys) but on a GPU their forward and backward run through the
library's own recurrence in every matrix mode (``functional.lstm``: one fused
product + cell launch per time step, the two directions of the encoder on two
streams); there is no vendor-LSTM path on the GPU.  The vocabulary projection with its cross-entropy and arg-max -- most
of the branch's arithmetic, 2*1024*vocab per token -- has a HIP path of its
own, ``decode_loss``: it skips the pad tokens' rows in all three GEMMs and
never holds more than one copy of the logits.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import functional as Fn
from .noise import noisy


class TextModel(nn.Module):
    def __init__(self, vocabsize, initrange=0.1):
        super().__init__()
        self.embed = nn.Embedding(vocabsize, 512)
        self.proj = nn.Linear(1024, vocabsize)
        with torch.no_grad():
            self.embed.weight.uniform_(-initrange, initrange)
            self.proj.bias.zero_()
            self.proj.weight.uniform_(-initrange, initrange)


class DAE(TextModel):
    def __init__(self, latent_dim, vocabsize, padidx):
        super().__init__(vocabsize)
        self.drop = nn.Dropout(0.5)
        self.E = nn.LSTM(512, 1024, 1, dropout=0.0, bidirectional=True)
        self.G = nn.LSTM(512, 1024, 1, dropout=0.0)
        self.hgen = nn.Linear(2048, latent_dim)
        self.z2emb = nn.Linear(latent_dim, 512)
        self.padidx = padidx
        self.word_drop = 0.3   # probability with which training drops a word of the encoder's input (text_vae.py:70)

    def flatten(self):
        self.E.flatten_parameters()
        self.G.flatten_parameters()

    @staticmethod
    def _side_stream(device):
        """A second stream on ``device`` for work that runs beside the current stream: the package's own
        ``lstm`` stream (one per device, created outside PyTorch's recycled stream pool and kept outside the
        module so that the module still copies and pickles)."""
        from .streams import owned_stream

        return owned_stream(device, "lstm")

    def run_lstm(self, mod, emb, hidden=None, need_output=True):
        """``mod(emb, hidden)`` for a one-layer ``nn.LSTM``.  On a GPU there is ONE path, the library's recurrence
        (functional.lstm, one call per direction) in every matrix mode -- the mode picks the pipe of the four
        products over all time steps, the per-step products run on split 16-bit planes at fp32 accuracy (two fp16
        planes in the default mode, three bf16 planes otherwise, the fp32 mode included) -- and sizes it does not
        take are an error, not a detour through the vendor's LSTM.  A module on the CPU (the tests' mirror of the
        reference's DAE) runs the stock module.  With ``need_output`` False the (T, B, dirs*H) output is not
        assembled (None is returned for it)."""
        if not emb.is_cuda:
            return mod(emb, hidden)
        T, B, I = emb.shape
        H = mod.hidden_size
        mode = _lib.matrix_mode(getattr(self, "matrix_mode", None))
        if emb.dtype != torch.float32 or mod.num_layers != 1 or not Fn.lstm_applies(T, B, I, H):
            raise _lib.GtgError("the LSTM recurrence takes fp32 input, one layer, input size a multiple of 4 and hidden "
                                "size a multiple of 64 (got %s, %d layers, I=%d, H=%d)" % (emb.dtype, mod.num_layers, I, H))

        def one_direction(d):
            sfx = "_l0_reverse" if d else "_l0"
            h0 = hidden[0][d] if hidden is not None else None
            c0 = hidden[1][d] if hidden is not None else None
            return Fn.lstm(emb, getattr(mod, "weight_ih" + sfx), getattr(mod, "weight_hh" + sfx),
                           getattr(mod, "bias_ih" + sfx), getattr(mod, "bias_hh" + sfx), h0, c0, bool(d), mode)

        if mod.bidirectional:
            # The two directions are independent and each step of a recurrence is one short launch with a block
            # per CU, far from filling it: the reverse direction runs on a second stream beside the forward one
            # (autograd runs each direction's backward on the stream of its forward, so the backward recurrences
            # overlap the same way).
            cur = torch.cuda.current_stream(emb.device)
            side = self._side_stream(emb.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                emb.record_stream(side)
                rev = one_direction(1)
                for t in rev:
                    t.record_stream(cur)
            fwd = one_direction(0)
            cur.wait_stream(side)
            results = [fwd, rev]
        else:
            results = [one_direction(0)]
        outs, hn, cn = [r[0] for r in results], [r[1] for r in results], [r[2] for r in results]
        output = None
        if need_output:
            output = outs[0] if len(outs) == 1 else torch.cat(outs, 2)
        return output, (torch.stack(hn), torch.stack(cn))

    def _embed(self, input, add=None):
        """self.drop(self.embed(input)) (+ add, a (B, 512) row per sentence broadcast over time): on a GPU one
        library kernel (lookup, dropout mask regenerated in the backward, the add) with an ordered scatter-add as
        backward; the stock modules for a model left on the CPU."""
        w = self.embed.weight
        if w.is_cuda:
            return Fn.embed_dropout(input, w, self.drop.p if self.training else 0.0, add)
        emb = self.drop(self.embed(input))
        return emb if add is None else emb + add

    def _drop(self, x):
        """self.drop(x): the library's dropout on a GPU (mask regenerated in the backward, never stored)."""
        return Fn.dropout(x, self.drop.p, self.training) if x.is_cuda else self.drop(x)

    @staticmethod
    def _linear(mod, x):
        """``mod(x)`` for an ``nn.Linear``: the library's GEMM on a GPU (functional.linear), the stock module for a
        model left on the CPU."""
        return Fn.linear(x, mod.weight, mod.bias) if x.is_cuda else mod(x)

    def encode(self, input):
        emb = self._embed(input)
        _, (ht, _) = self.run_lstm(self.E, emb, need_output=False)
        return self._linear(self.hgen, torch.cat([ht[-2], ht[-1]], 1))

    def decode(self, z, input, hidden=None):
        emb = self._embed(input, self._linear(self.z2emb, z))
        output, hidden = self.run_lstm(self.G, emb, hidden)
        output = self._drop(output)
        logits = self.proj(output.view(-1, output.size(-1)))
        return logits.view(output.size(0), output.size(1), -1), hidden

    def decode_loss(self, z, input, targets, hidden=None, matrix_mode=None):
        """decode + loss_rec + the arg-max of accuracy in one step (text_vae.py:61-66,
        77-79, 83-84): returns (per-token loss (T, B), predicted tokens (T, B) with -1
        at pad targets, hidden).  ``loss.sum(dim=0)`` is loss_rec's result.  The matrix
        pipe of the three GEMMs is ``matrix_mode``, else the module's ``matrix_mode``
        attribute if one was set, else the package default."""
        if matrix_mode is None:
            matrix_mode = getattr(self, "matrix_mode", None)
        emb = self._embed(input, self._linear(self.z2emb, z))
        output, hidden = self.run_lstm(self.G, emb, hidden)
        output = self._drop(output)
        loss, pred = Fn.proj_ce(output.reshape(-1, output.size(-1)), self.proj.weight, self.proj.bias,
                                targets.reshape(-1), self.padidx, matrix_mode)
        return loss.view(targets.size()), pred.view(targets.size()), hidden

    def accuracy_from_pred(self, pred, targets, as_float=True):
        mask = targets != self.padidx
        acc = ((pred == targets) & mask).sum() / mask.sum()
        return acc.item() if as_float else acc

    def forward(self, input):
        _input = noisy(self.padidx, input, self.word_drop, 0, 0, 0) if self.training and self.word_drop > 0 else input
        return self.encode(_input)

    def loss_rec(self, logits, targets):
        loss = F.cross_entropy(logits.view(-1, logits.size(-1)), targets.flatten(), ignore_index=self.padidx,
                               reduction="none").view(targets.size())
        return loss.sum(dim=0)

    def accuracy(self, logits, targets, is_dev=False):
        pred = torch.argmax(logits, 2)  # softmax is monotone: same arg-max as text_vae.py:83-84
        mask = targets != self.padidx
        return (((pred == targets) & mask).sum() / mask.sum()).item()

    def autoenc(self, inputs, targets):
        return self(inputs)

    def generate(self, z, max_len, alg):
        sents = []
        input = torch.full((1, len(z)), 2, dtype=torch.long, device=z.device)
        hidden = None
        for _ in range(max_len):
            sents.append(input)
            logits, hidden = self.decode(z, input, hidden)
            if alg == "greedy":
                input = logits.argmax(dim=-1)
            elif alg == "sample":
                input = torch.multinomial(logits.squeeze(dim=0).exp(), num_samples=1).t()
        return sents
