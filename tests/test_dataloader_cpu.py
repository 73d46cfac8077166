"""Host side of the path's input: collation layout (dataloader.py:6-35 in the reference) and the length-bucketed
batch sampler (BASELINE configs[4]).  CPU only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-supervised-asr_amd"))

from dataloader import BucketBatchSampler, get_data_loader, padded_fraction  # noqa: E402
from dataset import SyntheticDataset  # noqa: E402


def _lengths(ds):
    return [ds[i][0].shape[0] for i in range(len(ds))]


def test_collate_layout_sorted_desc_zero_padded():
    ds = SyntheticDataset(10, 8, 12, 30, seed=3)
    xs, ilens, ys = next(iter(get_data_loader(ds, 4, shuffle=True, drop_last=False, generator=torch.Generator().manual_seed(1))))
    assert ilens == sorted(ilens, reverse=True) and xs.shape == (4, ilens[0], 8)
    for b, l in enumerate(ilens):
        assert float(xs[b, l:].abs().sum()) == 0.0
    assert all(y.dtype == torch.int64 for y in ys)


def test_bucket_sampler_covers_every_utterance_once_and_cuts_padding():
    ds = SyntheticDataset(203, 4, 12, 400, seed=5)
    lens = _lengths(ds)
    assert lens == sorted(lens), "datasets keep their keys sorted by frame count"
    gen = torch.Generator().manual_seed(7)
    sampler = BucketBatchSampler(len(ds), 16, shuffle=True, drop_last=False, generator=gen)
    batches = list(sampler)
    assert len(batches) == len(sampler) == 13
    assert sorted(i for b in batches for i in b) == list(range(203))
    assert all(b == list(range(b[0], b[0] + len(b))) for b in batches), "a batch is a run of length-neighbours"
    assert [b[0] for b in batches] != sorted(b[0] for b in batches), "shuffle permutes whole batches"
    perm = torch.randperm(203, generator=torch.Generator().manual_seed(7)).tolist()
    uniform = [perm[i:i + 16] for i in range(0, 203, 16)]
    assert padded_fraction(lens, batches) < 0.03 < 0.15 < padded_fraction(lens, uniform)
    # drop_last and the loader path
    assert len(list(BucketBatchSampler(203, 16, False, True))) == 12
    loader = get_data_loader(ds, 16, shuffle=True, drop_last=False, generator=torch.Generator().manual_seed(7), bucket=True)
    seen = 0
    for xs, ilens, ys in loader:
        assert ilens == sorted(ilens, reverse=True) and (ilens[0] - ilens[-1]) <= 40
        seen += len(ilens)
    assert seen == 203


def test_bucket_sampler_is_rank_independent():
    a = list(BucketBatchSampler(100, 8, True, False, generator=torch.Generator().manual_seed(11)))
    b = list(BucketBatchSampler(100, 8, True, False, generator=torch.Generator().manual_seed(11)))
    assert a == b
