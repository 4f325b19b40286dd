"""DAMSM pre-training path (pretrain_DAMSM.py:49-130) on the GPU: the bi-LSTM training kernels (forward that keeps
the gates + back-propagation through time) against the reference module's own outputs and gradients
(tests/golden/text_encoder_train.npz), and one full DAMSM update against the CPU oracle."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import check, load_golden, rel_l2  # noqa: E402
from oracle import fill  # noqa: E402
from oracle import sbagan_oracle as O  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('name', ['small', 'bird'])
def test_text_encoder_training_path_vs_reference_golden(dev, golden_dir, name):
    import model
    T = load_golden(golden_dir, 'text_encoder_train.npz')
    ntoken, ninput, nhidden = (int(v) for v in T['%s/dims' % name])
    net = model.RNN_ENCODER(ntoken, ninput=ninput, drop_prob=0.0, nhidden=nhidden)
    net.load_state_dict(fill.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, salt=7))
    net.to(dev).train()
    cap = torch.from_numpy(T['%s/captions' % name]).to(dev)
    lens = torch.from_numpy(T['%s/cap_lens' % name]).to(dev)
    words, sent = net(cap, lens, net.init_hidden(cap.size(0)), max_len=int(lens.max()))
    assert words.grad_fn is not None and 'LstmBidirTrainFn' in type(words.grad_fn).__name__      # the HIP path ran
    np.testing.assert_allclose(words.detach().cpu().numpy(), T['%s/words_emb' % name], rtol=0, atol=1e-5)
    np.testing.assert_allclose(sent.detach().cpu().numpy(), T['%s/sent_emb' % name], rtol=0, atol=1e-5)
    gw, gs = fill.unit(tuple(words.shape), 801).to(dev), fill.unit(tuple(sent.shape), 802).to(dev)
    ((words * gw).sum() + (sent * gs).sum()).backward()
    torch.cuda.synchronize()
    for n, p in net.named_parameters():
        check(T, '%s/grad/%s' % (name, n), p.grad, rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_damsm_update_vs_oracle(dev, mode):
    """One pre-training update (both encoders' trainable parts) against the CPU: the image side is the nn.Module
    CNN_ENCODER itself evaluated by PyTorch -- in TRAINING mode (`train`: pretrain_DAMSM.py:51 cnn_model.train(): frozen
    Inception weights, BatchNorm with batch statistics, running statistics moved) and in eval mode (`eval`:
    pretrain_DAMSM.py:134) -- the text side and the losses the oracle.  Losses 2e-3, every gradient 1e-2 relative L2,
    parameters after the clipped Adam step within the Adam step size; in training mode the BatchNorm running statistics
    of the trunk after the step equal the module's."""
    import model
    from miscc.config import cfg, reset_cfg
    from sbagan import ops
    from sbagan.damsm import DAMSMStep
    reset_cfg()
    cfg.TEXT.EMBEDDING_DIM, cfg.TRAIN.RNN_GRAD_CLIP = 256, 0.25
    s = cfg.TRAIN.SMOOTH
    s.GAMMA1, s.GAMMA2, s.GAMMA3 = 4.0, 5.0, 10.0
    ops.set_compute_dtype(torch.float32)
    B, T = 6, 12
    torch.manual_seed(5)
    text = model.RNN_ENCODER(200, ninput=300, drop_prob=0.0, nhidden=256)
    enc = model.CNN_ENCODER(256)
    with torch.no_grad():       # non-trivial BatchNorm parameters / buffers
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.1, 0.1)
                m.running_var.uniform_(0.8, 1.2)
    text_ref, enc_ref = copy.deepcopy(text), copy.deepcopy(enc)
    text.to(dev).train()
    enc.to(dev).train(mode == 'train')
    enc_ref.train(mode == 'train')
    caps, lens = fill.synthetic_captions(B, words_num=T, lmax=T - 2, vocab=200, tag=31)
    class_ids = np.array([0, 1, 2, 0, 3, 4])
    img_cpu = fill.uniform((B, 3, 128, 128), 32)
    img = img_cpu.to(dev)
    lr = 2e-3
    st = DAMSMStep(text, enc, B, lr=lr)
    p0 = {n: p.detach().clone().cpu() for n, p in st.trainable.named_parameters()}
    out = st.step(img, caps.to(dev), lens.to(dev), class_ids)
    torch.cuda.synchronize()
    # ---- the same update on the CPU: nn.Module image encoder, oracle losses
    text_ref.train()
    for p in enc_ref.parameters():
        p.requires_grad_(False)
    params = list(text_ref.parameters()) + [enc_ref.emb_features.weight, enc_ref.emb_cnn_code.weight,
                                            enc_ref.emb_cnn_code.bias]
    for p in params:
        p.requires_grad_(True)
    wf, sc = enc_ref(img_cpu)
    we, se = text_ref(caps, lens, text_ref.init_hidden(B))
    labels = torch.arange(B)
    w0, w1 = O.words_loss(wf, we, labels, lens, class_ids, B, 4.0, 5.0, 10.0)
    s0, s1 = O.sent_loss(sc, se, labels, class_ids, B, 10.0)
    (w0 + w1 + s0 + s1).backward()
    for k, r in (('w_loss0', w0), ('w_loss1', w1), ('s_loss0', s0), ('s_loss1', s1)):
        assert abs(float(out[k]) - float(r)) <= 2e-3 * max(1.0, abs(float(r))), (k, float(out[k]), float(r))
    ref_grads = {}
    for n, p in text_ref.named_parameters():
        ref_grads['text_encoder.' + n] = p.grad
    ref_grads['emb_features.weight'] = enc_ref.emb_features.weight.grad
    ref_grads['emb_cnn_code.weight'] = enc_ref.emb_cnn_code.weight.grad
    ref_grads['emb_cnn_code.bias'] = enc_ref.emb_cnn_code.bias.grad
    total = torch.sqrt(sum((p.grad ** 2).sum() for p in text_ref.parameters()))
    assert abs(float(out['rnn_grad_norm']) - float(total)) <= 1e-2 * float(total)
    coef = min(1.0, 0.25 / (float(total) + 1e-6))
    for n, p in st.trainable.named_parameters():
        g_ref = ref_grads[n] * (coef if n.startswith('text_encoder.') else 1.0)
        assert rel_l2(p.grad, g_ref) <= 1e-2, (n, rel_l2(p.grad, g_ref))
        # Adam step 1 from zero moments: p - lr * g / (|g| + eps)
        want = p0[n].double() - lr * g_ref.double() / (g_ref.double().abs() + 1e-8)
        err = (p.detach().cpu().double() - want).abs()
        assert float((err > 0.05 * lr).double().mean()) <= 2e-2 and float(err.max()) <= 2.05 * lr, n
    # ---- BatchNorm buffers of the frozen trunk: moved by the training-mode forward exactly as the module moves them
    bufs, refb = dict(enc.named_buffers()), dict(enc_ref.named_buffers())
    worst = 0.0
    for n, b in refb.items():
        if n.endswith('num_batches_tracked'):
            assert int(bufs[n]) == int(b), n
        else:
            worst = max(worst, rel_l2(bufs[n], b))
    assert worst <= 2e-3, worst
    if mode == 'train':
        assert int(refb['Mixed_6e.branch_pool.bn.num_batches_tracked']) == 1
    reset_cfg()
