"""CPU oracle for the frozen text encoder of the GAN step.

TEST INFRASTRUCTURE ONLY: only tests/ may import this file; the product path (sba-gan_amd/) never does.

Plain numpy restatement of RNN_ENCODER.forward (reference AttnGAN2/code/model.py:127-159) for the
configuration the GAN step uses -- eval mode (dropout = identity, :137), one-layer bidirectional LSTM
(:104-108) over packed sequences (:139-147), words_emb = output.transpose(1, 2) (:151),
sent_emb = h_n.transpose(0, 1).view(-1, 2H) (:154-158).  The LSTM cell follows torch.nn.LSTM's published
equations (gate order i | f | g | o):
    i = sigmoid(W_ii x + b_ii + W_hi h + b_hi)   f = sigmoid(W_if x + b_if + W_hf h + b_hf)
    g = tanh   (W_ig x + b_ig + W_hg h + b_hg)   o = sigmoid(W_io x + b_io + W_ho h + b_ho)
    c' = f * c + i * g                            h' = o * tanh(c')
Packed-sequence semantics: the forward direction runs t = 0 .. len-1, the reverse direction t = len-1 .. 0,
outputs past len are zero, h_n is the state after each direction's last valid step.

Parity status: PINNED against tests/golden/text_encoder.npz, produced by the reference's own RNN_ENCODER
(tools/make_golden.py text) on closed-form parameters (oracle/fill.py), checked in tests/test_oracle_golden.py.
"""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def rnn_encoder_forward(P, captions, cap_lens, max_len=None):
    """P: state_dict-keyed float arrays ('encoder.weight', 'rnn.weight_ih_l0', 'rnn.weight_ih_l0_reverse', ...).
    captions [B][T] int, cap_lens [B] int.  Returns (words_emb [B][2H][L], sent_emb [B][2H]) in float64,
    L = max_len or max(cap_lens) (pad_packed_sequence pads to the longest caption, model.py:147)."""
    emb = np.asarray(P['encoder.weight'], dtype=np.float64)
    B, T = captions.shape
    H = np.asarray(P['rnn.weight_hh_l0']).shape[1]
    L = int(max(cap_lens)) if max_len is None else int(max_len)
    words = np.zeros((B, 2 * H, L))
    sent = np.zeros((B, 2 * H))
    for d, suf in enumerate(('', '_reverse')):
        w_ih = np.asarray(P['rnn.weight_ih_l0' + suf], dtype=np.float64)
        w_hh = np.asarray(P['rnn.weight_hh_l0' + suf], dtype=np.float64)
        bias = np.asarray(P['rnn.bias_ih_l0' + suf], dtype=np.float64) + np.asarray(P['rnn.bias_hh_l0' + suf],
                                                                                   dtype=np.float64)
        for b in range(B):
            n = int(cap_lens[b])
            h = np.zeros(H)
            c = np.zeros(H)
            steps = range(n) if d == 0 else range(n - 1, -1, -1)
            for t in steps:
                g = w_ih @ emb[int(captions[b, t])] + w_hh @ h + bias
                i, f = _sigmoid(g[:H]), _sigmoid(g[H:2 * H])
                gg, o = np.tanh(g[2 * H:3 * H]), _sigmoid(g[3 * H:])
                c = f * c + i * gg
                h = o * np.tanh(c)
                if t < L:
                    words[b, d * H:(d + 1) * H, t] = h
            sent[b, d * H:(d + 1) * H] = h
    return words, sent
