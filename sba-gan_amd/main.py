"""Entry point with the reference's command line (AttnGAN2/code/main.py:22-150):

    python main.py --cfg cfg/bird_style.yml --gpu 0 --data_dir ../data/birds [--manualSeed N]

cfg.TRAIN.FLAG: train; otherwise cfg.B_VALIDATION ? sampling(split) : gen_example(example_filenames.txt)."""
import os
import sys
import time

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from datasets import TextDataset, tokenize  # noqa: E402
from miscc import cli, transforms  # noqa: E402
from miscc.config import cfg  # noqa: E402


def parse_args(argv=None):
    return cli.options('Train a AttnGAN network', 'cfg/bird_attn2.yml', argv)


def _encode_sentences(path, wordtoix):
    """word-id lists of the non-empty lines of a text file (words outside the vocabulary are skipped)"""
    with open(path, 'r') as f:
        lines = [ln.replace('\ufffd\ufffd', ' ') for ln in f.read().split('\n') if ln]
    token_lists = [t for t in (tokenize(ln) for ln in lines) if t]
    return [[wordtoix[w] for w in tokens if w in wordtoix] for tokens in token_lists]


def build_example_dic(wordtoix, data_dir=None):
    """The caption batches of gen_example (reference main.py:34-83), from their contract: for every file listed in
    <data_dir>/example_filenames.txt, key = its base name and value = [captions (n x Lmax int64, zero padded, rows by
    DESCENDING length: what the packed bi-LSTM takes), the lengths in that order, the permutation that sorted them]."""
    data_dir = data_dir or cfg.DATA_DIR
    with open(os.path.join(data_dir, 'example_filenames.txt'), 'r') as f:
        listed = [ln for ln in f.read().split('\n') if ln]
    batches = {}
    for name in listed:
        print('Load from:', name)
        ids = _encode_sentences(os.path.join(data_dir, name + '.txt'), wordtoix)
        lengths = np.array([len(s) for s in ids])
        order = np.argsort(lengths)[::-1]
        padded = np.zeros((len(ids), int(lengths.max())), dtype='int64')
        for row, src in enumerate(order):
            padded[row, :lengths[src]] = ids[src]
        batches[os.path.basename(name)] = [padded, lengths[order], order]
    return batches


def gen_example(wordtoix, algo):
    algo.gen_example(build_example_dic(wordtoix))


def main(argv=None):
    args = parse_args(argv)
    cli.configure(args)
    output_dir = cli.output_dir()
    training = bool(cfg.TRAIN.FLAG)
    split_dir = 'train' if training else 'test'
    imsize = cli.image_size()
    image_transform = transforms.Compose([transforms.Resize(int(imsize * 76 / 64)), transforms.RandomCrop(imsize),
                                          transforms.RandomHorizontalFlip()])
    dataset = TextDataset(cfg.DATA_DIR, split_dir, base_size=cfg.TREE.BASE_SIZE, transform=image_transform)
    assert dataset
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=cfg.TRAIN.BATCH_SIZE, drop_last=True,
                                             shuffle=True, num_workers=int(cfg.WORKERS))
    print(len(dataloader))
    from trainer import condGANTrainer as trainer
    algo = trainer(output_dir, dataloader, dataset.n_words, dataset.ixtoword)
    start_t = time.time()
    if training:
        algo.train()
    elif cfg.B_VALIDATION:
        algo.sampling(split_dir)
    else:
        gen_example(dataset.wordtoix, algo)
    print('Total time for training:', time.time() - start_t)


if __name__ == '__main__':
    main()
