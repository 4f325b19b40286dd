"""Text / image data path behind the training step, written from the BATCH CONTRACT the step consumes
rather than from the reference's loader code.  What a caller of AttnGAN2/code/datasets.py relies on, and what this module
keeps:

  * `TextDataset(data_dir, split, base_size, transform)`: a map-style dataset whose items are
        (images: list of BRANCH_NUM float tensors 3 x S_i x S_i in [-1, 1], S_i = base_size * 2**i,
         caption: int64 array WORDS_NUM x 1, zero padded,  caption length,  class id,  file key)
    with the attributes the scripts read: filenames, captions, ixtoword, wordtoix, n_words, class_id, imsize,
    embeddings_num, number_example.
  * the on-disk formats: <split>/filenames.pickle, <split>/class_info.pickle (latin1), and captions.pickle =
    [train captions, test captions, ixtoword, wordtoix] (protocol 2), built from text/<key>.txt on first use; word ids
    are handed out in order of first appearance (train split first), 0 = '<end>' -- a reference checkpoint's embedding
    rows line up with them.
  * `prepare_data(batch)`: rows sorted by caption length, descending (what the packed bi-LSTM needs), on the device.
  * the draw order on numpy's global generator per item -- image transform first, then the caption pick, then (only for
    captions longer than WORDS_NUM) the word subset -- so that a seeded run visits the same samples.
No nltk / torchvision / pandas: the tokenizer is the \\w+ pattern, transforms come from miscc.transforms.
"""
import os
import pickle
import re

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

from miscc import transforms
from miscc.config import cfg

END_TOKEN = '<end>'
_WORD = re.compile(r'\w+')
_CUB_SUBDIR = 'CUB_200_2011/CUB_200_2011'


def tokenize(text):
    """lower-cased \\w+ tokens with non-ASCII characters dropped (a token that becomes empty disappears)"""
    words = (w.encode('ascii', 'ignore').decode('ascii') for w in _WORD.findall(text.lower()))
    return [w for w in words if w]


def _unpickle(path, **kw):
    with open(path, 'rb') as f:
        return pickle.load(f, **kw)


def prepare_data(batch):
    """One collated batch -> [images per scale, captions B x T, caption lengths B (descending), class ids (numpy),
    keys], every tensor on the training device, all rows permuted together by the length sort."""
    images, captions, lengths, class_ids, keys = batch
    lengths, order = torch.sort(lengths, dim=0, descending=True)
    device = torch.device('cuda', cfg.GPU_ID) if cfg.CUDA else torch.device('cpu')
    images = [scale.index_select(0, order).to(device) for scale in images]
    captions = captions.index_select(0, order).reshape(captions.size(0), -1)      # B x T x 1 -> B x T
    keys = [keys[int(j)] for j in order]
    return [images, captions.to(device), lengths.to(device), class_ids.index_select(0, order).numpy(), keys]


def _crop_around(img, box):
    """square window of 1.5 x the longer side of box = (x, y, w, h), centred on the box, clipped to the image"""
    x, y, w, h = box
    half = int(max(w, h) * 0.75)
    cx, cy = int((2 * x + w) / 2), int((2 * y + h) / 2)
    W, H = img.size
    return img.crop([max(0, cx - half), max(0, cy - half), min(W, cx + half), min(H, cy + half)])


def get_imgs(img_path, imsize, bbox=None, transform=None, normalize=None):
    """The image pyramid of one sample: crop to the bounding box, apply the (random) transform once, then one
    normalised tensor per scale -- the transform's output is the LARGEST scale, the others are resized from it."""
    img = Image.open(img_path).convert('RGB')
    if bbox is not None:
        img = _crop_around(img, bbox)
    if transform is not None:
        img = transform(img)
    if cfg.GAN.B_DCGAN:
        return [normalize(img)]
    last = cfg.TREE.BRANCH_NUM - 1
    return [normalize(img if i == last else transforms.Resize(imsize[i])(img)) for i in range(cfg.TREE.BRANCH_NUM)]


class _CaptionStore(object):
    """Captions of both splits as word-id lists plus the vocabulary, cached in <data_dir>/captions.pickle."""

    def __init__(self, data_dir, per_image):
        self.data_dir, self.per_image = data_dir, per_image
        self.names = {s: self._names(s) for s in ('train', 'test')}
        cache = os.path.join(data_dir, 'captions.pickle')
        if os.path.isfile(cache):
            train, test, self.ixtoword, self.wordtoix = _unpickle(cache)[:4]
            print('Load from: ', cache)
        else:
            words = {s: self._read_split(self.names[s]) for s in ('train', 'test')}
            self.wordtoix = {END_TOKEN: 0}
            for sentence in words['train'] + words['test']:           # ids in order of first appearance
                for w in sentence:
                    self.wordtoix.setdefault(w, len(self.wordtoix))
            self.ixtoword = {i: w for w, i in self.wordtoix.items()}
            train, test = ([[self.wordtoix[w] for w in sentence] for sentence in words[s]] for s in ('train', 'test'))
            with open(cache, 'wb') as f:
                pickle.dump([train, test, self.ixtoword, self.wordtoix], f, protocol=2)
            print('Save to: ', cache)
        self.encoded = {'train': train, 'test': test}

    def _names(self, split):
        path = os.path.join(self.data_dir, split, 'filenames.pickle')
        if not os.path.isfile(path):
            return []
        names = _unpickle(path)
        print('Load filenames from: %s (%d)' % (path, len(names)))
        return names

    def _read_split(self, names):
        """the first `per_image` non-empty captions of every text/<key>.txt, tokenised"""
        out = []
        for key in names:
            with open(os.path.join(self.data_dir, 'text', key + '.txt'), 'r') as f:
                lines = [ln.replace('\ufffd\ufffd', ' ') for ln in f.read().split('\n') if ln]
            kept = [t for t in (tokenize(ln) for ln in lines) if t][:self.per_image]
            if len(kept) < self.per_image:
                print('ERROR: the captions for %s less than %d' % (key, len(kept)))
            out.extend(kept)
        return out


def _cub_boxes(data_dir):
    """{image key without extension: [x, y, w, h]} from the CUB-200-2011 annotation files"""
    base = os.path.join(data_dir, _CUB_SUBDIR)
    boxes = np.loadtxt(os.path.join(base, 'bounding_boxes.txt')).astype(int)[:, 1:]
    with open(os.path.join(base, 'images.txt')) as f:
        keys = [ln.split()[1][:-4] for ln in f if ln.strip()]
    print('Total filenames: ', len(keys), keys[0] + '.jpg')
    return dict(zip(keys, boxes.tolist()))


class TextDataset(data.Dataset):
    def __init__(self, data_dir, split='train', base_size=64, transform=None, target_transform=None):
        self.data_dir, self.transform, self.target_transform = data_dir, transform, target_transform
        self.norm = transforms.Compose([transforms.ToTensor(),
                                        transforms.Normalize((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))])
        self.embeddings_num = cfg.TEXT.CAPTIONS_PER_IMAGE
        self.imsize = [base_size * 2 ** i for i in range(cfg.TREE.BRANCH_NUM)]
        self.bbox = _cub_boxes(data_dir) if 'birds' in data_dir else None
        self.image_root = os.path.join(data_dir, _CUB_SUBDIR) if self.bbox is not None else data_dir
        store = _CaptionStore(data_dir, self.embeddings_num)
        split = 'train' if split == 'train' else 'test'
        self.filenames, self.captions = store.names[split], store.encoded[split]
        self.ixtoword, self.wordtoix, self.n_words = store.ixtoword, store.wordtoix, len(store.ixtoword)
        self.number_example = len(self.filenames)
        info = os.path.join(data_dir, split, 'class_info.pickle')
        self.class_id = _unpickle(info, encoding='latin1') if os.path.isfile(info) else np.arange(self.number_example)

    def get_caption(self, sent_ix):
        """caption `sent_ix` as a zero-padded WORDS_NUM x 1 int64 column and its (clipped) length; a longer caption
        keeps WORDS_NUM of its words, a random subset in the original order"""
        ids = np.asarray(self.captions[sent_ix], dtype='int64')
        if (ids == 0).any():
            print('ERROR: do not need END (0) token', ids)
        limit = cfg.TEXT.WORDS_NUM
        column = np.zeros((limit, 1), dtype='int64')
        if len(ids) > limit:
            pick = np.arange(len(ids))
            np.random.shuffle(pick)
            ids = ids[np.sort(pick[:limit])]
        column[:len(ids), 0] = ids
        return column, len(ids)

    def __getitem__(self, index):
        key = self.filenames[index]
        box = self.bbox[key] if self.bbox is not None else None
        imgs = get_imgs('%s/images/%s.jpg' % (self.image_root, key), self.imsize, box, self.transform,
                        normalize=self.norm)
        which = np.random.randint(0, self.embeddings_num)             # one of the image's captions
        caption, length = self.get_caption(index * self.embeddings_num + which)
        return imgs, caption, length, self.class_id[index], key

    def __len__(self):
        return self.number_example
