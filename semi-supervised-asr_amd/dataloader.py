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


class BucketBatchSampler(object):
    """Length-bucketed batches (BASELINE configs[4] "bucketed padding").  The datasets keep their keys
    sorted by frame count (dataset.py:54-71 in the reference), but its loader then draws uniformly (shuffle=True), so a
    batch is padded to the longest of `batch_size` random utterances.  Here a batch is `batch_size` NEIGHBOURS of the
    length-sorted order and shuffling permutes whole batches, which keeps the padded frames (compute of every layer,
    and - through the unmasked attention softmax, SURVEY F1/F2 - part of the result) close to the minimum.  The
    permutation comes from the loader's seeded generator, so every data-parallel rank draws the same global batches."""

    def __init__(self, n, batch_size, shuffle, drop_last, generator=None):
        self.n, self.batch_size, self.shuffle, self.drop_last, self.generator = n, batch_size, shuffle, drop_last, generator

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        starts = list(range(0, self.n, self.batch_size))
        if self.drop_last and starts and self.n - starts[-1] < self.batch_size:
            starts.pop()
        if self.shuffle:
            starts = [starts[i] for i in torch.randperm(len(starts), generator=self.generator).tolist()]
        for b0 in starts:
            yield list(range(b0, min(b0 + self.batch_size, self.n)))


def padded_fraction(lengths, batches):
    """Share of padded frames over the given batches (lists of dataset indices): 1 - sum(len) / sum(B * max len)."""
    real = sum(lengths[i] for b in batches for i in b)
    padded = sum(len(b) * max(lengths[i] for i in b) for b in batches)
    return 1.0 - real / float(padded)


def _raw_items(batch):
    """Collation deferred to the DeviceFeed: the utterances in the reference's batch order (frame count descending), as the
    (feature, token_ids) pairs the dataset holds - no padding, no copy."""
    return _by_frames(batch)


def _raw_texts(batch):
    return sorted(batch, key=lambda item: len(item[1]), reverse=True)


def get_data_loader(dataset, batch_size, shuffle, drop_last, speech_only=False, text_only=False, generator=None,
                    bucket=False, raw=False):
    """raw=True: batches are lists of (feature, token_ids) in the collated ORDER but not padded - the input of a DeviceFeed,
    which pads them straight into pinned memory (this rank's rows only under data parallelism)."""
    fn = _speech_collate_fn if speech_only else (_text_collate_fn if text_only else _collate_fn)
    if raw:
        fn = _raw_texts if text_only else _raw_items
    if bucket:
        sampler = BucketBatchSampler(len(dataset), batch_size, shuffle, drop_last, generator=generator)
        return DataLoader(dataset, batch_sampler=sampler, collate_fn=fn, num_workers=0)
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=fn, num_workers=0,
                      drop_last=drop_last, generator=generator)
