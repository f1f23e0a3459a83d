#!/usr/bin/env python
"""Generates tests/golden/lstm_encoder.npz from the LIVE reference's ``LSTMInstruction``
(``gnn/modules/question_encoding/lstm_encoder.py``): ``encode_question`` on padded token ids - the word embeddings it
looks up, the LSTM's parameters, ``query_hidden_emb`` and the final states it stores - and the three instructions its
``forward`` derives from them (``base_encoder.py:82-122``).

    python tests/golden/make_golden_lstm.py          (build container only, CPU)
"""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/gnn")


def main():
    from modules.question_encoding import lstm_encoder, base_encoder
    # the reference's own start-up bug: LSTMInstruction calls BaseInstruction.__init__(args)
    # without the `constraint` argument - the same shim tools/run_reference.py applies
    _init = base_encoder.BaseInstruction.__init__
    base_encoder.BaseInstruction.__init__ = lambda self, args, constraint=False: _init(self, args, constraint)
    torch.manual_seed(77)
    rng = np.random.default_rng(77)
    out = {}
    for tag, (B, T, word_dim, entity_dim, vocab) in {"d50": (5, 7, 300, 50, 40), "d128": (16, 11, 64, 128, 60)}.items():
        folder = tempfile.mkdtemp() + "/"
        with open(folder + "vocab.txt", "w") as f:
            f.write("\n".join("w%d" % i for i in range(vocab)) + "\n")
        args = dict(use_cuda=False, q_type="seq", num_step=3, lm_dropout=0.0, linear_dropout=0.0, lm_frozen=0, word_dim=word_dim,
                    entity_dim=entity_dim, data_folder=folder, word2id="vocab.txt")
        word_embedding = nn.Embedding(vocab + 1, word_dim, padding_idx=vocab)
        enc = lstm_encoder.LSTMInstruction(args, word_embedding, vocab)
        enc.eval()
        text = rng.integers(0, vocab, (B, T))
        for b in range(B):                                       # ragged questions, padded with num_word (dataset_load.py)
            text[b, rng.integers(2, T + 1):] = vocab
        q = torch.from_numpy(text).long()
        with torch.no_grad():
            hidden, node = enc.encode_question(q)
            instructions, attn = enc(q)
        out.update({tag + ".query_text": text, tag + ".word_emb": word_embedding(q).detach().numpy(),
                    tag + ".query_hidden_emb": hidden.numpy(), tag + ".h_n": enc.instruction_hidden[0].numpy(),
                    tag + ".c_n": enc.instruction_mem[0].numpy(), tag + ".query_node_emb": node.numpy(),
                    tag + ".instructions": np.stack([i.numpy() for i in instructions]),
                    tag + ".attn": np.stack([a.numpy() for a in attn])})
        for k, v in enc.state_dict().items():
            out[tag + ".param." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "lstm_encoder.npz"), **out)
    print("wrote lstm_encoder.npz:", {k: v.shape for k, v in out.items() if "param" not in k})


if __name__ == "__main__":
    main()
