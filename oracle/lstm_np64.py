"""ORACLE (test infrastructure, NOT product code) - float64 numpy restatement of the question encoder's recurrence:
``LSTMInstruction.encode_question`` (reference ``gnn/modules/question_encoding/lstm_encoder.py:32-36``) calls
``self.node_encoder = nn.LSTM(word_dim, entity_dim, batch_first=True, bidirectional=False)`` (``:27-30``) on the word
embeddings of the question with zero initial states.  torch.nn.LSTM's documented cell (gate order i, f, g, o in the rows
of ``weight_ih_l0`` / ``weight_hh_l0``):

    g_t = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh
    i, f, o = sigmoid(g_t[0:H]), sigmoid(g_t[H:2H]), sigmoid(g_t[3H:4H]);   g = tanh(g_t[2H:3H])
    c_t = f * c_{t-1} + i * g;   h_t = o * tanh(c_t)

Pinned by tests/test_oracle_golden.py against tests/golden/lstm_encoder.npz, recorded from the LIVE reference's
``LSTMInstruction`` (tests/golden/make_golden_lstm.py).  Only ``tests/`` may import this file."""
from __future__ import annotations

import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_forward(x, w_ih, w_hh, b_ih=None, b_hh=None, h0=None, c0=None):
    """x [B,T,E]; w_ih [4H,E]; w_hh [4H,H]; biases [4H] or None; h0 / c0 [B,H] or None.
    Returns (out [B,T,H], h_n [B,H], c_n [B,H]) in float64."""
    x = np.asarray(x, np.float64)
    w_ih, w_hh = np.asarray(w_ih, np.float64), np.asarray(w_hh, np.float64)
    B, T, _ = x.shape
    H = w_hh.shape[1]
    bias = np.zeros(4 * H)
    if b_ih is not None:
        bias = bias + np.asarray(b_ih, np.float64)
    if b_hh is not None:
        bias = bias + np.asarray(b_hh, np.float64)
    h = np.zeros((B, H)) if h0 is None else np.asarray(h0, np.float64).copy()
    c = np.zeros((B, H)) if c0 is None else np.asarray(c0, np.float64).copy()
    out = np.zeros((B, T, H))
    for t in range(T):                                           # lstm_encoder.py:34 (one nn.LSTM call over all tokens)
        g = x[:, t] @ w_ih.T + h @ w_hh.T + bias
        i, f, gg, o = _sigmoid(g[:, :H]), _sigmoid(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), _sigmoid(g[:, 3 * H:])
        c = f * c + i * gg
        h = o * np.tanh(c)
        out[:, t] = h
    return out, h, c
