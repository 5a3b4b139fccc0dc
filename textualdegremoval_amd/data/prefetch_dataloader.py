"""Prefetchers with the reference's names and `next()/reset()` protocol (data/prefetch_dataloader.py):
`CPUPrefetcher` walks a DataLoader, `CUDAPrefetcher` uploads the NEXT batch on its own HIP stream while the
current step runs (`prefetch_mode: cuda` of every shipped YAML), `PrefetchDataLoader` keeps a background
thread `num_prefetch_queue` batches ahead."""
import queue
import threading

import torch
from torch.utils.data import DataLoader

_END = object()


class _Ahead:
    """iterator that a daemon thread keeps `depth` items ahead of its consumer"""

    def __init__(self, it, depth):
        self.q = queue.Queue(maxsize=depth)
        self.t = threading.Thread(target=self._fill, args=(it,), daemon=True)
        self.t.start()

    def _fill(self, it):
        for item in it:
            self.q.put(item)
        self.q.put(_END)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is _END:
            raise StopIteration
        return item


class PrefetchDataLoader(DataLoader):
    def __init__(self, num_prefetch_queue, **kwargs):
        self.num_prefetch_queue = num_prefetch_queue
        super().__init__(**kwargs)

    def __iter__(self):
        return _Ahead(super().__iter__(), self.num_prefetch_queue)


class CPUPrefetcher:
    def __init__(self, loader):
        self.ori_loader = loader
        self.reset()

    def next(self):
        return next(self.loader, None)

    def reset(self):
        self.loader = iter(self.ori_loader)


class CUDAPrefetcher:
    """batch k+1 travels host -> HBM on a side stream while step k computes; `next()` makes the compute stream
    wait for that copy (pinned host memory required, as the trainer checks)."""

    def __init__(self, loader, opt):
        self.ori_loader = loader
        self.opt = opt
        self.device = torch.device('cuda' if opt['num_gpu'] != 0 else 'cpu')
        self.stream = torch.cuda.Stream() if self.device.type == 'cuda' else None
        self.reset()

    def _upload_next(self):
        self.batch = next(self.loader, None)
        if self.batch is None or self.stream is None:
            return
        with torch.cuda.stream(self.stream):
            for k, v in self.batch.items():
                if torch.is_tensor(v):
                    self.batch[k] = v.to(device=self.device, non_blocking=True)

    def next(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        batch = self.batch
        if batch is not None and self.stream is not None:
            for v in batch.values():              # the compute stream now owns these blocks
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(torch.cuda.current_stream())
        self._upload_next()
        return batch

    def reset(self):
        self.loader = iter(self.ori_loader)
        self._upload_next()
