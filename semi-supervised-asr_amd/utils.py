"""Host-side helpers on the training path, same names/semantics as the reference's utils.py
(cc 150-152, to_gpu 154-158, pad_list 173-179, _seq_mask 181-190, adjust_learning_rate 134-139,
remove_pad_eos 192-201, to_sents/ind2character/char_list_to_str 160-163,212-235, calculate_cer
222-228, Logger 237-245, infinite_iter 247-254).  tensorboardX and editdistance are optional:
absent here, so logging degrades to a no-op and the edit distance is computed locally.
"""
import numpy as np
import torch

try:  # optional dependency (not installed in the ROCm image)
    from tensorboardX import SummaryWriter as _SummaryWriter
except Exception:  # pragma: no cover
    _SummaryWriter = None


def cc(net):
    """Move a module/tensor to the GPU when there is one ('cuda' is HIP on ROCm)."""
    return net.to(torch.device("cuda" if torch.cuda.is_available() else "cpu"))


def to_gpu(data):
    xs, ilens, ys = data
    return cc(xs), ilens, [cc(y) for y in ys]


def pad_list(xs, pad_value=0):
    """Ragged list of tensors -> [B, Lmax, ...] filled with pad_value."""
    longest = max(int(x.size(0)) for x in xs)
    out = xs[0].new_full((len(xs), longest) + tuple(xs[0].shape[1:]), pad_value)
    for i, x in enumerate(xs):
        out[i, : x.size(0)] = x
    return out


def _seq_mask(seq_len, max_len, is_list=True):
    """Float mask [B, max_len]: 1 where position < length."""
    lens = torch.as_tensor(np.asarray(seq_len)) if is_list else seq_len
    grid = torch.arange(0, max_len, device=lens.device).unsqueeze(0)
    return (grid < lens.unsqueeze(1)).float()


def adjust_learning_rate(optimizer, lr):
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr


def remove_pad_eos(sequences, eos=2):
    """Keep each sequence up to (excluding) its first <EOS>."""
    out = []
    for seq in sequences:
        seq = list(seq)
        out.append(seq[: seq.index(eos)] if eos in seq else seq)
    return out


def ind2character(sequences, non_lang_syms, vocab):
    inv = {v: k for k, v in vocab.items()}
    skip = set(vocab[s] for s in non_lang_syms)
    return [[inv[int(i)] for i in seq if int(i) not in skip] for seq in sequences]


def char_list_to_str(char_lists):
    return ["".join(" " if ch == "<space>" else ch for ch in chars) for chars in char_lists]


def to_sents(ind_seq, vocab, non_lang_syms):
    return char_list_to_str(ind2character(ind_seq, non_lang_syms, vocab))


def edit_distance(a, b):
    """Levenshtein distance (stands in for editdistance.eval)."""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def calculate_cer(hyps, refs):
    total_dis = sum(edit_distance(h, r) for h, r in zip(hyps, refs))
    total_len = sum(len(r) for r in refs)
    return float(total_dis) / float(total_len)


class Logger(object):
    """tensorboardX scalar/text logger; silently inert when tensorboardX is not installed."""

    def __init__(self, logdir="./log"):
        self.writer = _SummaryWriter(logdir) if _SummaryWriter is not None else None

    def scalar_summary(self, tag, value, step):
        if self.writer is not None:
            self.writer.add_scalar(tag, value, step)

    def text_summary(self, tag, value, step):
        if self.writer is not None:
            self.writer.add_text(tag, value, step)


def infinite_iter(iterable):
    while True:
        for item in iterable:
            yield item
