"""Host side of the question-encoder LSTM swap (no GPU): HipLSTM is an nn.LSTM with the same state_dict, shares the
parameters of the module it replaces and - on the CPU, where there is no kernel - behaves as the parent class."""
import torch
import torch.nn as nn


def test_swap_lstm_shares_parameters_and_keeps_state_dict():
    from gnnrag_amd import install
    from gnnrag_amd.modules.question_encoding.lstm import HipLSTM

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.node_encoder = nn.LSTM(30, 20, batch_first=True)
            self.two_layers = nn.LSTM(30, 20, num_layers=2, batch_first=True)
            self.seq_first = nn.LSTM(30, 20)
            self.wide = nn.LSTM(30, 300, batch_first=True)               # 4 H > 1024: not taken

    torch.manual_seed(1)
    enc = Enc().eval()
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    x = torch.randn(4, 6, 30)
    with torch.no_grad():
        want, (wh, wc) = enc.node_encoder(x)
    assert install.swap_lstm(enc) == 1
    assert isinstance(enc.node_encoder, HipLSTM)
    assert all(type(getattr(enc, n)) is nn.LSTM for n in ("two_layers", "seq_first", "wide"))
    assert list(enc.state_dict()) == list(sd) and all(torch.equal(enc.state_dict()[k], sd[k]) for k in sd)
    with torch.no_grad():
        out, (h, c) = enc.node_encoder(x)                                 # CPU tensor: torch's own LSTM
    assert torch.equal(out, want) and torch.equal(h, wh) and torch.equal(c, wc)
    enc.load_state_dict(sd)                                               # checkpoints still load
    assert install.swap_lstm(enc) == 0                                    # idempotent


def test_lstm_forward_rejects_cpu_tensors():
    import pytest
    from gnnrag_amd import _lib, ops
    with pytest.raises(_lib.GnnragError):
        ops.lstm_forward(torch.zeros(1, 1, 4), torch.zeros(8, 4), torch.zeros(8, 2))
