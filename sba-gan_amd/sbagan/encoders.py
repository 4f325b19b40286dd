"""Encoder modules around the hot path: module API and state_dict layout of the reference, so that
`image_encoder*.pth` / `text_encoder*.pth` checkpoints load.  Their arithmetic is third-party in the reference
(torchvision Inception-v3, pytorch_pretrained_bert; SURVEY.md 8c: parity unpinned).  On the GPU the frozen forwards
run on the hand-written kernels (sbagan.inception_hip.InceptionHIP, sbagan.bert_hip.BertHIP); the nn.Module
forwards below are the definitions those are tested against, and the training path of BertEncoder.

CNN_ENCODER carries its own Inception-v3 definition (torchvision is not installed on the
GPU box and the pretrained-weights URL of model.py:171 is unreachable offline); attribute
names follow torchvision's so the reference's checkpoints match key for key.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from miscc.config import cfg


class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, **kw):
        super(BasicConv2d, self).__init__()
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class InceptionA(nn.Module):
    def __init__(self, cin, pool_features):
        super(InceptionA, self).__init__()
        self.branch1x1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch5x5_1 = BasicConv2d(cin, 48, kernel_size=1)
        self.branch5x5_2 = BasicConv2d(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, padding=1)
        self.branch_pool = BasicConv2d(cin, pool_features, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b5 = self.branch5x5_2(self.branch5x5_1(x))
        b3 = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([b1, b5, b3, bp], 1)


class InceptionB(nn.Module):
    def __init__(self, cin):
        super(InceptionB, self).__init__()
        self.branch3x3 = BasicConv2d(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3(x)
        bd = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        return torch.cat([b3, bd, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


class InceptionC(nn.Module):
    def __init__(self, cin, c7):
        super(InceptionC, self).__init__()
        self.branch1x1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7_2 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = self.branch7x7dbl_1(x)
        for m in (self.branch7x7dbl_2, self.branch7x7dbl_3, self.branch7x7dbl_4, self.branch7x7dbl_5):
            bd = m(bd)
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([b1, b7, bd, bp], 1)


class InceptionD(nn.Module):
    def __init__(self, cin):
        super(InceptionD, self).__init__()
        self.branch3x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch3x3_2 = BasicConv2d(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3_2(self.branch3x3_1(x))
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        return torch.cat([b3, b7, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


class InceptionE(nn.Module):
    def __init__(self, cin):
        super(InceptionE, self).__init__()
        self.branch1x1 = BasicConv2d(cin, 320, kernel_size=1)
        self.branch3x3_1 = BasicConv2d(cin, 384, kernel_size=1)
        self.branch3x3_2a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b3 = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(b3), self.branch3x3_2b(b3)], 1)
        bd = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(bd), self.branch3x3dbl_3b(bd)], 1)
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([b1, b3, bd, bp], 1)


class CNN_ENCODER(nn.Module):
    """model.py:162-267: Inception-v3 trunk on a 299x299 bilinear resize; region features
    B x nef x 17 x 17 (1x1 conv on Mixed_6e) and a global code B x nef (Linear on the pooled
    Mixed_7c).  The trunk is frozen; weights come from a checkpoint (random otherwise)."""

    def __init__(self, nef):
        super(CNN_ENCODER, self).__init__()
        self.nef = nef if cfg.TRAIN.FLAG else 256
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.Mixed_5b = InceptionA(192, 32)
        self.Mixed_5c = InceptionA(256, 64)
        self.Mixed_5d = InceptionA(288, 64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, 128)
        self.Mixed_6c = InceptionC(768, 160)
        self.Mixed_6d = InceptionC(768, 160)
        self.Mixed_6e = InceptionC(768, 192)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        for p in self.parameters():
            p.requires_grad = False
        self.emb_features = nn.Conv2d(768, self.nef, kernel_size=1, stride=1, padding=0, bias=False)
        self.emb_cnn_code = nn.Linear(2048, self.nef)
        self.init_trainable_weights()

    def init_trainable_weights(self):
        self.emb_features.weight.data.uniform_(-0.1, 0.1)
        self.emb_cnn_code.weight.data.uniform_(-0.1, 0.1)

    def forward(self, x):
        x = F.interpolate(x, size=(299, 299), mode='bilinear', align_corners=True)
        x = self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x)))
        x = F.max_pool2d(x, kernel_size=3, stride=2)
        x = self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x))
        x = F.max_pool2d(x, kernel_size=3, stride=2)
        x = self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(x)))
        x = self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(self.Mixed_6a(x)))))
        features = x
        x = self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x)))
        x = F.avg_pool2d(x, kernel_size=8)
        x = x.view(x.size(0), -1)
        cnn_code = self.emb_cnn_code(x)
        features = self.emb_features(features)
        return features, cnn_code


class BertEncoder(nn.Module):
    """model_bert.py:161-189: BERT-base -> 1x1 conv 768->nef + tanh (words) and Linear 768->nef
    + tanh (sentence); no attention mask is passed (:181).  Uses HuggingFace `transformers`
    (the reference's pytorch_pretrained_bert is not installed); without network access the
    trunk is randomly initialised unless `bert_dir` points at local weights."""

    def __init__(self, embedding_dim=128, bert_dir=None):
        super(BertEncoder, self).__init__()
        self.max_length = cfg.TEXT.WORDS_NUM
        self.fc = nn.Linear(768, embedding_dim, bias=True)
        self.tanh = nn.Tanh()
        self.conv_text = nn.Conv2d(768, embedding_dim, kernel_size=1, stride=1, padding=0, bias=True)
        from transformers import BertConfig, BertModel
        self.model = BertModel.from_pretrained(bert_dir) if bert_dir else BertModel(BertConfig())
        for i, layer in enumerate(self.model.children()):
            if i == 2:
                break
            for p in layer.parameters():
                p.requires_grad = False

    use_hip = True      # frozen forward (eval / no_grad, CUDA) on the HIP kernels: sbagan.bert_hip.BertHIP

    def _hip_runner(self):
        from . import ops
        from .bert_hip import BertHIP
        key = (ops.compute_dtype(), tuple(p._version for p in self.parameters()))
        if getattr(self, '_hip_key', None) != key:
            self.__dict__['_hip'] = BertHIP(self)
            self.__dict__['_hip_key'] = key
        return self.__dict__['_hip']

    def forward(self, captions):
        if self.use_hip and captions.is_cuda and not torch.is_grad_enabled() and not self.training:
            return self._hip_runner()(captions)
        out = self.model(captions)
        words_embs = out.last_hidden_state.transpose(1, 2).contiguous().unsqueeze(3)
        words_embs = self.tanh(self.conv_text(words_embs).squeeze(3))
        sent_emb = self.tanh(self.fc(out.pooler_output))
        return words_embs, sent_emb
