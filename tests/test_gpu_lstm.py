"""gnnrag_lstm_forward / HipLSTM against torch.nn.LSTM in fp32 on the CPU (the reference's question encoder IS
nn.LSTM(word_dim, entity_dim, batch_first=True): gnn/modules/question_encoding/lstm_encoder.py:27-36).  Tolerance 2e-5
absolute on states in (-1, 1): both sides are fp32 with different summation orders over 300 + 200 terms per gate."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _ref_and_dev(E, H, seed, bias=True):
    torch.manual_seed(seed)
    ref = nn.LSTM(E, H, batch_first=True, bias=bias)
    return ref


@pytest.mark.parametrize("B,T,E,H", [(16, 9, 300, 200), (1, 1, 300, 50), (3, 13, 300, 50), (20, 7, 64, 256),
                                     (700, 6, 300, 200), (513, 5, 100, 52)])
def test_lstm_forward_matches_torch_cpu(B, T, E, H):
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import ops
    dev = torch.device("cuda", 0)
    ref = _ref_and_dev(E, H, seed=B + T)
    x = torch.randn(B, T, E)
    with torch.no_grad():
        want_out, (want_h, want_c) = ref(x)
        out, h_n, c_n = ops.lstm_forward(x.to(dev), ref.weight_ih_l0.to(dev), ref.weight_hh_l0.to(dev),
                                         ref.bias_ih_l0.to(dev), ref.bias_hh_l0.to(dev))
    assert (out.cpu() - want_out).abs().max().item() <= TOL
    assert (h_n.cpu() - want_h[0]).abs().max().item() <= TOL
    assert (c_n.cpu() - want_c[0]).abs().max().item() <= 4 * TOL          # |c| is not bounded by 1
    assert torch.equal(out[:, -1], h_n)
    out2, _, _ = ops.lstm_forward(x.to(dev), ref.weight_ih_l0.to(dev), ref.weight_hh_l0.to(dev),
                                  ref.bias_ih_l0.to(dev), ref.bias_hh_l0.to(dev))
    assert torch.equal(out, out2)                                          # one fixed summation order


def test_lstm_initial_states_and_no_bias():
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import ops
    dev = torch.device("cuda", 0)
    B, T, E, H = 5, 4, 32, 24
    ref = _ref_and_dev(E, H, seed=3, bias=False)
    x, h0, c0 = torch.randn(B, T, E), torch.randn(1, B, H), torch.randn(1, B, H)
    with torch.no_grad():
        want_out, (want_h, want_c) = ref(x, (h0, c0))
        out, h_n, c_n = ops.lstm_forward(x.to(dev), ref.weight_ih_l0.to(dev), ref.weight_hh_l0.to(dev), None, None,
                                         h0[0].to(dev), c0[0].to(dev))
    assert (out.cpu() - want_out).abs().max().item() <= TOL
    assert (c_n.cpu() - want_c[0]).abs().max().item() <= 4 * TOL


def test_hiplstm_is_a_drop_in_for_the_encoders_lstm():
    """The way the reference calls it: zero states as [1, B, H] tensors, output + (h_n, c_n) back; the swapped module
    shares the parameters and keeps the state_dict keys; with autograd on it is torch's own LSTM."""
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import install
    from gnnrag_amd.modules.question_encoding.lstm import HipLSTM
    dev = torch.device("cuda", 0)

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.node_encoder = nn.LSTM(300, 200, batch_first=True, bidirectional=False)
            self.other = nn.LSTM(300, 200, batch_first=True, bidirectional=True)       # not eligible: left alone

    torch.manual_seed(0)
    enc = Enc().to(dev).eval()
    keys = list(enc.state_dict())
    cpu = nn.LSTM(300, 200, batch_first=True)
    cpu.load_state_dict(enc.node_encoder.state_dict())
    w_before = enc.node_encoder.weight_ih_l0
    assert install.swap_lstm(enc) == 1
    assert isinstance(enc.node_encoder, HipLSTM) and type(enc.other) is nn.LSTM
    assert enc.node_encoder.weight_ih_l0 is w_before and list(enc.state_dict()) == keys
    x = torch.randn(16, 9, 300)
    zeros = torch.zeros(1, 16, 200, device=dev)
    with torch.no_grad():
        out, (h_n, c_n) = enc.node_encoder(x.to(dev), (zeros, zeros))
        want, (wh, wc) = cpu(x)
    assert out.shape == (16, 9, 200) and h_n.shape == (1, 16, 200) and c_n.shape == (1, 16, 200)
    assert (out.cpu() - want).abs().max().item() <= TOL and (h_n.cpu() - wh).abs().max().item() <= TOL
    # training: autograd's LSTM, gradients reach the shared parameters
    enc.train()
    out, _ = enc.node_encoder(x.to(dev))
    out.sum().backward()
    assert w_before.grad is not None and float(w_before.grad.abs().sum()) > 0
