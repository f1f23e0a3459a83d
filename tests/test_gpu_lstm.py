"""gnnrag_lstm_forward / HipLSTM against torch.nn.LSTM in fp32 on the CPU (the reference's question encoder IS
nn.LSTM(word_dim, entity_dim, batch_first=True): gnn/modules/question_encoding/lstm_encoder.py:27-36).  Tolerance 2e-5
absolute on states in (-1, 1): both sides are fp32 with different summation orders over 300 + 200 terms per gate."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden", "lstm_encoder.npz")

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


@pytest.mark.parametrize("tag", ["d50", "d128"])
def test_lstm_forward_against_the_live_reference_fixture_and_the_float64_oracle(tag):
    """The encoder's own inputs and outputs recorded from the LIVE LSTMInstruction.encode_question (tests/golden/
    lstm_encoder.npz): the HIP recurrence within 2e-6 of the float64 oracle (oracle/lstm_np64.py, pinned to the same
    fixture on the CPU) and within 4e-6 of the reference's fp32 states."""
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import ops
    sys.path.insert(0, REPO)
    import oracle.lstm_np64 as lstm_np64
    dev = torch.device("cuda", 0)
    g = np.load(GOLDEN)
    P = {k.split(".param.")[1]: g[k] for k in g.files if k.startswith(tag + ".param.")}
    w = [P["node_encoder." + n] for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    x = g[tag + ".word_emb"]
    want64 = lstm_np64.lstm_forward(x, *w)
    out, h_n, c_n = ops.lstm_forward(torch.from_numpy(x).to(dev), *[torch.from_numpy(a).to(dev) for a in w])
    for got, w64, ref in ((out, want64[0], g[tag + ".query_hidden_emb"]), (h_n, want64[1], g[tag + ".h_n"]),
                          (c_n, want64[2], g[tag + ".c_n"])):
        got = got.cpu().numpy().astype(np.float64)
        assert np.abs(got - w64).max() <= 2e-6, np.abs(got - w64).max()
        assert np.abs(got - ref).max() <= 4e-6, np.abs(got - ref).max()


@pytest.mark.parametrize("tag", ["d50", "d128"])
def test_live_instruction_module_with_the_swapped_encoder_reproduces_its_instructions(tag):
    """The reference's OWN LSTMInstruction (sources staged under oracle/_ref/gnn), parameters of the fixture, on the
    MI355X with install.swap_lstm: the three instructions and attention weights its forward derives
    (base_encoder.py:82-122) against what the same module produced on the CPU when the fixture was recorded."""
    import tempfile
    ref = os.path.join(REPO, "oracle", "_ref", "gnn")
    if not os.path.isfile(os.path.join(ref, "modules", "question_encoding", "lstm_encoder.py")):
        pytest.skip("oracle/_ref not staged")
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import install
    from gnnrag_amd.modules.question_encoding.lstm import HipLSTM
    sys.path.insert(0, ref)
    try:
        from modules.question_encoding import base_encoder, lstm_encoder
    finally:
        sys.path.remove(ref)
    dev = torch.device("cuda", 0)
    g = np.load(GOLDEN)
    P = {k.split(".param.")[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".param.")}
    text = g[tag + ".query_text"]
    vocab = int(text.max())                                  # the pad word = num_word (every fixture question is padded)
    word_dim, entity_dim = P["node_encoder.weight_ih_l0"].shape[1], P["node_encoder.weight_hh_l0"].shape[1]
    folder = tempfile.mkdtemp() + "/"
    with open(folder + "vocab.txt", "w") as f:
        f.write("\n".join("w%d" % i for i in range(vocab)) + "\n")
    args = dict(use_cuda=True, q_type="seq", num_step=3, lm_dropout=0.0, linear_dropout=0.0, lm_frozen=0, word_dim=word_dim,
                entity_dim=entity_dim, data_folder=folder, word2id="vocab.txt")
    init = base_encoder.BaseInstruction.__init__
    base_encoder.BaseInstruction.__init__ = lambda self, a, constraint=False: init(self, a, constraint)
    try:
        enc = lstm_encoder.LSTMInstruction(args, nn.Embedding(vocab + 1, word_dim, padding_idx=vocab), vocab)
    finally:
        base_encoder.BaseInstruction.__init__ = init
    enc.load_state_dict(P, strict=True)
    enc.to(dev).eval()
    assert install.swap_lstm(enc) == 1 and isinstance(enc.node_encoder, HipLSTM)
    with torch.no_grad():
        instructions, attn = enc(torch.from_numpy(text).long().to(dev))
    got_i = torch.stack(instructions).cpu().numpy()
    got_a = torch.stack(attn).cpu().numpy()
    assert np.abs(enc.query_hidden_emb.cpu().numpy() - g[tag + ".query_hidden_emb"]).max() <= 4e-6
    assert np.abs(got_i - g[tag + ".instructions"]).max() <= 2e-5
    assert np.abs(got_a - g[tag + ".attn"]).max() <= 2e-5
