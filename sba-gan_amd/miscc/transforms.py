"""The handful of image transforms the reference takes from torchvision (main.py:112-115,
datasets.py:96-98): PIL in, PIL or float tensor out.  torchvision is not a dependency here."""
import random

import numpy as np
import torch
from PIL import Image


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class Resize(object):
    """int: the SHORTER side becomes `size` (aspect kept); (h, w): exact."""

    def __init__(self, size, interpolation=Image.BILINEAR):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img):
        if isinstance(self.size, int):
            w, h = img.size
            if (w <= h and w == self.size) or (h <= w and h == self.size):
                return img
            if w < h:
                ow, oh = self.size, int(self.size * h / w)
            else:
                oh, ow = self.size, int(self.size * w / h)
            return img.resize((ow, oh), self.interpolation)
        return img.resize(self.size[::-1], self.interpolation)


Scale = Resize          # the name pretrain_DAMSM.py:244 still uses


class RandomCrop(object):
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else size

    def __call__(self, img):
        w, h = img.size
        th, tw = self.size
        if w == tw and h == th:
            return img
        x1 = random.randint(0, w - tw)
        y1 = random.randint(0, h - th)
        return img.crop((x1, y1, x1 + tw, y1 + th))


class RandomHorizontalFlip(object):
    def __call__(self, img):
        if random.random() < 0.5:
            return img.transpose(Image.FLIP_LEFT_RIGHT)
        return img


class ToTensor(object):
    """HWC uint8 PIL image -> CHW float32 in [0, 1]."""

    def __call__(self, img):
        a = np.asarray(img, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div_(255.0)


class Normalize(object):
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std
