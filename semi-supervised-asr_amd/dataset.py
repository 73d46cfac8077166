"""Datasets feeding the hot path.  PickleDataset reads the reference's on-disk format
({utt_id: {'feature': float32[T, D], 'token_ids': list[int]}}, dataset.py:46-80): length filter from the
config keys max/min_feature_length, max/min_text_length, keys sorted by frame count.
SyntheticDataset generates the same structure in memory for benchmarks and tests."""
import pickle

import numpy as np
from torch.utils.data import Dataset


def _within(entry, config):
    frames = entry["feature"].shape[0]
    chars = len(entry["token_ids"])
    return (config["min_feature_length"] <= frames <= config["max_feature_length"]
            and config["min_text_length"] <= chars <= config["max_text_length"])


class DictDataset(Dataset):
    """Common behaviour over an {utt: {'feature', 'token_ids'}} dict."""

    def __init__(self, data_dict, config=None, sort=True):
        self.data_dict = data_dict
        keys = [k for k in data_dict if config is None or _within(data_dict[k], config)]
        if sort:
            keys.sort(key=lambda k: data_dict[k]["feature"].shape[0])
        self.keys = keys

    def __getitem__(self, index):
        item = self.data_dict[self.keys[index]]
        return item["feature"], item["token_ids"]

    def __len__(self):
        return len(self.keys)


class PickleDataset(DictDataset):
    def __init__(self, pickle_path, config=None, sort=True):
        with open(pickle_path, "rb") as f:
            data = pickle.load(f)
        super().__init__(data, config=config, sort=sort)


def synthetic_utterances(n, input_dim, vocab_size, t_max, seed, ragged=True, label_ratio=0.125):
    """N(0,1) features (CMVN-normalised fbank stand-in), lengths U[0.6 T, T] when ragged, labels
    uniform in [3, V), L = max(2, floor(label_ratio * T_i))."""
    rs = np.random.RandomState(seed)
    out = {}
    for i in range(n):
        t = int(rs.randint(int(0.6 * t_max), t_max + 1)) if ragged else int(t_max)
        out["utt%06d" % i] = dict(
            feature=rs.normal(0.0, 1.0, size=(t, input_dim)).astype(np.float32),
            token_ids=rs.randint(3, vocab_size, size=(max(2, int(label_ratio * t)),)).tolist())
    return out


class SyntheticDataset(DictDataset):
    def __init__(self, n, input_dim, vocab_size, t_max, seed=1234, ragged=True, config=None, sort=True):
        super().__init__(synthetic_utterances(n, input_dim, vocab_size, t_max, seed, ragged), config=config,
                         sort=sort)
