"""The question encoder's LSTM on the MI355X (the instruction path).

The reference's ``LSTMInstruction`` (``gnn/modules/question_encoding/lstm_encoder.py:27-36``) owns
``self.node_encoder = nn.LSTM(word_dim, entity_dim, batch_first=True)`` and calls it on ``[B, max_query_word, word_dim]``
with zero initial states - twice per forward (``base_encoder.py:74-80``: once from ``ReaRev.init_reason``, once from
the instruction module's own ``forward``).  ``torch.lstm`` on ROCm goes to MIOpen's RNN call, ~12 ms per call at these
shapes: two thirds of an evaluation batch's wall time at BASELINE config 2's hidden size.

``HipLSTM`` IS an ``nn.LSTM`` (same constructor, parameter names, ``state_dict``): without autograd its forward is one
``gnnrag_lstm_forward`` launch; with autograd enabled (training), or for a shape the kernel does not take, it is the parent
class.  ``swap_lstm(model)`` replaces every eligible ``nn.LSTM`` of a constructed model and shares - not copies - the
parameters, so a checkpoint loaded before or after lands in the same tensors.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import ops


def _eligible(m: nn.LSTM) -> bool:
    return (isinstance(m, nn.LSTM) and m.num_layers == 1 and not m.bidirectional and m.batch_first and
            getattr(m, "proj_size", 0) == 0 and 4 * m.hidden_size <= 1024)


class HipLSTM(nn.LSTM):
    """One-layer, one-direction, batch_first LSTM; inference forward through ``gnnrag_lstm_forward``."""

    def forward(self, input, hx=None):  # noqa: A002  (torch's own argument name)
        fast = (_eligible(self) and isinstance(input, torch.Tensor) and input.is_cuda and input.dim() == 3 and
                input.dtype == torch.float32 and input.shape[0] > 0 and input.shape[1] > 0 and
                all(p.dtype == torch.float32 and p.is_cuda for p in self.parameters()) and
                (hx is None or all(isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.is_cuda for t in hx)) and
                not (torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters()))))
        if not fast:
            return super().forward(input, hx)
        h0 = c0 = None
        if hx is not None:
            h0, c0 = hx
            B, H = input.shape[0], self.hidden_size
            if tuple(h0.shape) != (1, B, H) or tuple(c0.shape) != (1, B, H):
                return super().forward(input, hx)       # torch raises its own shape error
            h0, c0 = h0[0], c0[0]
        bias = self.bias
        ws = self.__dict__.setdefault("_gnnrag_ws", {})       # the module's own scratch (not a parameter / buffer)
        out, h_n, c_n = ops.lstm_forward(input, self.weight_ih_l0, self.weight_hh_l0,
                                         self.bias_ih_l0 if bias else None, self.bias_hh_l0 if bias else None, h0, c0,
                                         workspaces=ws)
        return out, (h_n.unsqueeze(0), c_n.unsqueeze(0))

    @classmethod
    def sharing(cls, old: nn.LSTM) -> "HipLSTM":
        """A HipLSTM that holds the SAME Parameter objects as ``old``."""
        new = cls(old.input_size, old.hidden_size, num_layers=1, bias=old.bias, batch_first=True, dropout=0.0,
                  bidirectional=False)
        for name, _ in list(new.named_parameters(recurse=False)):
            setattr(new, name, getattr(old, name))
        new.train(old.training)
        return new


def swap_lstm(model: nn.Module) -> int:
    """Replaces every eligible ``nn.LSTM`` child of ``model`` (at any depth) by a ``HipLSTM`` sharing its parameters.
    Returns how many were replaced."""
    n = 0
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            if type(child) is nn.LSTM and _eligible(child):
                setattr(parent, name, HipLSTM.sharing(child))
                n += 1
    return n
