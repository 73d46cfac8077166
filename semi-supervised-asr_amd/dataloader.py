"""Collation = the hot path's input layout (dataloader.py:6-35 in the reference): batch sorted by frame
count descending, features zero-padded to the longest, `ilens` a host list, texts a list of int64 tensors."""
import numpy as np
import torch
from torch.utils.data import DataLoader


def _by_frames(batch):
    return sorted(batch, key=lambda item: item[0].shape[0], reverse=True)


def _pad_features(items):
    feats = [torch.from_numpy(np.asarray(f, dtype=np.float32)) for f, _ in items]
    return torch.nn.utils.rnn.pad_sequence(feats, batch_first=True, padding_value=0), [int(f.shape[0]) for f in feats]


def _collate_fn(batch):
    items = _by_frames(batch)
    padded, ilens = _pad_features(items)
    return padded, ilens, [torch.from_numpy(np.asarray(t, dtype=np.int64)) for _, t in items]


def _speech_collate_fn(batch):
    return _pad_features(_by_frames(batch))


def _text_collate_fn(batch):
    items = sorted(batch, key=lambda item: len(item[1]), reverse=True)
    return [torch.from_numpy(np.asarray(t, dtype=np.int64)) for _, t in items]


def get_data_loader(dataset, batch_size, shuffle, drop_last, speech_only=False, text_only=False, generator=None):
    fn = _speech_collate_fn if speech_only else (_text_collate_fn if text_only else _collate_fn)
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=fn, num_workers=0,
                      drop_last=drop_last, generator=generator)
