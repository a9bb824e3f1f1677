"""AverageMeter and a plain-text TermLogger with the reference's interface (logger.py).  The
reference draws curses progress bars with `blessings` + `progressbar2`; neither is installed here, and
a training job launched one-process-per-GPU logs to files anyway, so this version prints lines."""
from __future__ import division

import sys


class _Bar(object):
    def __init__(self, name, total):
        self.name, self.total = name, total

    def start(self):
        return self

    def update(self, i):
        pass

    def finish(self):
        pass


class _Writer(object):
    def __init__(self, prefix, stream=sys.stdout):
        self.prefix, self.stream = prefix, stream

    def write(self, string):
        self.stream.write('[{}] {}\n'.format(self.prefix, string))

    def flush(self):
        self.stream.flush()


class TermLogger(object):
    def __init__(self, n_epochs, train_size, valid_size):
        self.n_epochs, self.train_size, self.valid_size = n_epochs, train_size, valid_size
        self.epoch_bar = _Bar('epoch', n_epochs)
        self.train_writer, self.valid_writer = _Writer('train'), _Writer('valid')
        self.reset_train_bar()
        self.reset_valid_bar()

    def reset_train_bar(self):
        self.train_bar = _Bar('train', self.train_size)

    def reset_valid_bar(self):
        self.valid_bar = _Bar('valid', self.valid_size)


class AverageMeter(object):
    """Computes and stores the average and current value (logger.py:60-93)."""

    def __init__(self, i=1, precision=3):
        self.meters = i
        self.precision = precision
        self.reset(self.meters)

    def reset(self, i):
        self.val = [0] * i
        self.avg = [0] * i
        self.sum = [0] * i
        self.count = 0

    def update(self, val, n=1):
        if not isinstance(val, list):
            val = [val]
        assert (len(val) == self.meters)
        self.count += n
        for i, v in enumerate(val):
            self.val[i] = v
            self.sum[i] += v * n
            self.avg[i] = self.sum[i] / self.count

    def __repr__(self):
        val = ' '.join(['{:.{}f}'.format(v, self.precision) for v in self.val])
        avg = ' '.join(['{:.{}f}'.format(a, self.precision) for a in self.avg])
        return '{} ({})'.format(val, avg)
