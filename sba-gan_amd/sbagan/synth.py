"""Synthetic CUB-shaped batches: what prepare_data (datasets.py:28-56) hands the
step, generated on the device.  Images U(-1,1) at 64/128/256 px, captions int64 B x 20 zero
padded with lengths sorted descending (max forced to 18 so Lmax is fixed), class_ids
arange(B); words/sentence embeddings N(0,1) stand in for the frozen text encoder's output
unless an encoder is supplied."""
import numpy as np
import torch

from .trainer import build_mask


def synthetic_batch(batch_size, branch_num=3, nef=256, words_num=20, lmax=18, vocab=5450, device='cuda',
                    seed=100, text_encoder=None):
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    imgs = [(torch.rand((batch_size, 3, 64 * 2 ** i, 64 * 2 ** i), generator=g) * 2 - 1).to(device)
            for i in range(branch_num)]
    lens = torch.randint(5, lmax + 1, (batch_size,), generator=g)
    lens[0] = lmax
    lens, _ = torch.sort(lens, 0, True)
    caps = torch.randint(1, vocab, (batch_size, words_num), generator=g)
    caps = torch.where(torch.arange(words_num)[None, :] < lens[:, None], caps, torch.zeros_like(caps))
    caps, lens = caps.to(device), lens.to(device)
    if text_encoder is not None:
        with torch.no_grad():
            hidden = text_encoder.init_hidden(batch_size)
            words_embs, sent_emb = text_encoder(caps, lens, hidden)
            words_embs, sent_emb = words_embs.detach().float().contiguous(), sent_emb.detach().float()
    else:
        words_embs = torch.randn((batch_size, nef, lmax), generator=g).to(device)
        sent_emb = torch.randn((batch_size, nef), generator=g).to(device)
    mask = build_mask(caps, words_embs.size(2))
    return dict(imgs=imgs, captions=caps, cap_lens=lens, class_ids=np.arange(batch_size),
                words_embs=words_embs, sent_emb=sent_emb, mask=mask)
