"""CUB / COCO text-image data path with the reference's surface (AttnGAN2/code/datasets.py):
`prepare_data`, `get_imgs`, `TextDataset` (same constructor, attributes, pickle file formats and
per-item tuple), without the nltk / torchvision dependencies: the tokenizer is the regular expression the
reference hands to nltk's RegexpTokenizer (r'\\w+', datasets.py:157), the transforms are miscc.transforms."""
import os
import pickle
import re
from collections import defaultdict

import numpy as np
import numpy.random as random
import torch
import torch.utils.data as data
from PIL import Image

from miscc import transforms
from miscc.config import cfg

_TOKEN = re.compile(r'\w+')


def tokenize(text):
    """RegexpTokenizer(r'\\w+').tokenize(text.lower()) followed by the ascii filter of datasets.py:164-168."""
    out = []
    for t in _TOKEN.findall(text.lower()):
        t = t.encode('ascii', 'ignore').decode('ascii')
        if len(t) > 0:
            out.append(t)
    return out


def prepare_data(data):
    """datasets.py:28-56: sort the batch by caption length (descending) and move it to the device.
    Returns [real_imgs (list per scale), captions B x T int64, sorted_cap_lens B int64, class_ids (numpy), keys]."""
    imgs, captions, captions_lens, class_ids, keys = data
    sorted_cap_lens, sorted_cap_indices = torch.sort(captions_lens, 0, True)
    dev = torch.device('cuda', cfg.GPU_ID) if cfg.CUDA else torch.device('cpu')
    real_imgs = []
    for i in range(len(imgs)):
        imgs[i] = imgs[i][sorted_cap_indices]
        real_imgs.append(imgs[i].to(dev))
    captions = captions[sorted_cap_indices].squeeze()
    if captions.dim() == 1:          # batch of one: squeeze() dropped the batch axis too
        captions = captions.unsqueeze(0)
    class_ids = class_ids[sorted_cap_indices].numpy()
    keys = [keys[i] for i in sorted_cap_indices.numpy()]
    return [real_imgs, captions.to(dev), sorted_cap_lens.to(dev), class_ids, keys]


def get_imgs(img_path, imsize, bbox=None, transform=None, normalize=None):
    """datasets.py:59-90: crop to 1.5x the bounding box, transform, one tensor per scale."""
    img = Image.open(img_path).convert('RGB')
    width, height = img.size
    if bbox is not None:
        r = int(np.maximum(bbox[2], bbox[3]) * 0.75)
        center_x = int((2 * bbox[0] + bbox[2]) / 2)
        center_y = int((2 * bbox[1] + bbox[3]) / 2)
        y1 = np.maximum(0, center_y - r)
        y2 = np.minimum(height, center_y + r)
        x1 = np.maximum(0, center_x - r)
        x2 = np.minimum(width, center_x + r)
        img = img.crop([x1, y1, x2, y2])
    if transform is not None:
        img = transform(img)
    ret = []
    if cfg.GAN.B_DCGAN:
        ret = [normalize(img)]
    else:
        for i in range(cfg.TREE.BRANCH_NUM):
            if i < (cfg.TREE.BRANCH_NUM - 1):
                re_img = transforms.Resize(imsize[i])(img)
            else:
                re_img = img
            ret.append(normalize(re_img))
    return ret


class TextDataset(data.Dataset):
    """datasets.py:93-318.  data_dir layout: {train,test}/filenames.pickle, captions.pickle (built from
    text/<name>.txt on first use), {train,test}/class_info.pickle, images under images/ (or, for 'birds',
    CUB_200_2011/CUB_200_2011/images with bounding_boxes.txt / images.txt)."""

    def __init__(self, data_dir, split='train', base_size=64, transform=None, target_transform=None):
        self.transform = transform
        self.norm = transforms.Compose([transforms.ToTensor(),
                                        transforms.Normalize((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))])
        self.target_transform = target_transform
        self.embeddings_num = cfg.TEXT.CAPTIONS_PER_IMAGE
        self.imsize = []
        for i in range(cfg.TREE.BRANCH_NUM):
            self.imsize.append(base_size)
            base_size = base_size * 2
        self.data = []
        self.data_dir = data_dir
        self.bbox = self.load_bbox() if data_dir.find('birds') != -1 else None
        split_dir = os.path.join(data_dir, split)
        self.filenames, self.captions, self.ixtoword, self.wordtoix, self.n_words = \
            self.load_text_data(data_dir, split)
        self.class_id = self.load_class_id(split_dir, len(self.filenames))
        self.number_example = len(self.filenames)

    def load_bbox(self):
        base = os.path.join(self.data_dir, 'CUB_200_2011/CUB_200_2011')
        boxes = np.loadtxt(os.path.join(base, 'bounding_boxes.txt')).astype(int)
        filenames = [line.split()[1] for line in open(os.path.join(base, 'images.txt')) if line.strip()]
        print('Total filenames: ', len(filenames), filenames[0])
        return {f[:-4]: boxes[i][1:].tolist() for i, f in enumerate(filenames)}

    def load_captions(self, data_dir, filenames):
        all_captions = []
        for i in range(len(filenames)):
            cap_path = '%s/text/%s.txt' % (data_dir, filenames[i])
            with open(cap_path, 'r') as f:
                captions = f.read().split('\n')
            cnt = 0
            for cap in captions:
                if len(cap) == 0:
                    continue
                cap = cap.replace('\ufffd\ufffd', ' ')
                tokens = tokenize(cap)
                if len(tokens) == 0:
                    print('cap', cap)
                    continue
                all_captions.append(tokens)
                cnt += 1
                if cnt == self.embeddings_num:
                    break
            if cnt < self.embeddings_num:
                print('ERROR: the captions for %s less than %d' % (filenames[i], cnt))
        return all_captions

    def build_dictionary(self, train_captions, test_captions):
        word_counts = defaultdict(float)
        for sent in train_captions + test_captions:
            for word in sent:
                word_counts[word] += 1
        vocab = [w for w in word_counts if word_counts[w] >= 0]
        ixtoword, wordtoix = {0: '<end>'}, {'<end>': 0}
        for ix, w in enumerate(vocab, 1):
            wordtoix[w] = ix
            ixtoword[ix] = w
        train_new = [[wordtoix[w] for w in t if w in wordtoix] for t in train_captions]
        test_new = [[wordtoix[w] for w in t if w in wordtoix] for t in test_captions]
        return [train_new, test_new, ixtoword, wordtoix, len(ixtoword)]

    def load_text_data(self, data_dir, split):
        filepath = os.path.join(data_dir, 'captions.pickle')
        train_names = self.load_filenames(data_dir, 'train')
        test_names = self.load_filenames(data_dir, 'test')
        if not os.path.isfile(filepath):
            train_captions = self.load_captions(data_dir, train_names)
            test_captions = self.load_captions(data_dir, test_names)
            train_captions, test_captions, ixtoword, wordtoix, n_words = \
                self.build_dictionary(train_captions, test_captions)
            with open(filepath, 'wb') as f:
                pickle.dump([train_captions, test_captions, ixtoword, wordtoix], f, protocol=2)
                print('Save to: ', filepath)
        else:
            with open(filepath, 'rb') as f:
                x = pickle.load(f)
            train_captions, test_captions, ixtoword, wordtoix = x[0], x[1], x[2], x[3]
            n_words = len(ixtoword)
            print('Load from: ', filepath)
        if split == 'train':
            return train_names, train_captions, ixtoword, wordtoix, n_words
        return test_names, test_captions, ixtoword, wordtoix, n_words

    def load_class_id(self, data_dir, total_num):
        path = data_dir + '/class_info.pickle'
        if os.path.isfile(path):
            with open(path, 'rb') as f:
                return pickle.load(f, encoding='latin1')
        return np.arange(total_num)

    def load_filenames(self, data_dir, split):
        filepath = '%s/%s/filenames.pickle' % (data_dir, split)
        if os.path.isfile(filepath):
            with open(filepath, 'rb') as f:
                filenames = pickle.load(f)
            print('Load filenames from: %s (%d)' % (filepath, len(filenames)))
            return filenames
        return []

    def get_caption(self, sent_ix):
        """datasets.py:279-298: zero-padded WORDS_NUM x 1 int64 column; longer captions keep a random, ordered
        subset of WORDS_NUM words."""
        sent_caption = np.asarray(self.captions[sent_ix]).astype('int64')
        if (sent_caption == 0).sum() > 0:
            print('ERROR: do not need END (0) token', sent_caption)
        num_words = len(sent_caption)
        x = np.zeros((cfg.TEXT.WORDS_NUM, 1), dtype='int64')
        x_len = num_words
        if num_words <= cfg.TEXT.WORDS_NUM:
            x[:num_words, 0] = sent_caption
        else:
            ix = list(np.arange(num_words))
            np.random.shuffle(ix)
            ix = np.sort(ix[:cfg.TEXT.WORDS_NUM])
            x[:, 0] = sent_caption[ix]
            x_len = cfg.TEXT.WORDS_NUM
        return x, x_len

    def __getitem__(self, index):
        key = self.filenames[index]
        cls_id = self.class_id[index]
        if self.bbox is not None:
            bbox = self.bbox[key]
            data_dir = '%s/CUB_200_2011/CUB_200_2011' % self.data_dir
        else:
            bbox = None
            data_dir = self.data_dir
        img_name = '%s/images/%s.jpg' % (data_dir, key)
        imgs = get_imgs(img_name, self.imsize, bbox, self.transform, normalize=self.norm)
        sent_ix = random.randint(0, self.embeddings_num)
        new_sent_ix = index * self.embeddings_num + sent_ix
        caps, cap_len = self.get_caption(new_sent_ix)
        return imgs, caps, cap_len, cls_id, key

    def __len__(self):
        return len(self.filenames)
