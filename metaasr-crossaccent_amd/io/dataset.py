"""Data path of the reference (src/io/dataset.py) re-built for the MI355X path.

On-disk format is unchanged: `<dir>/feat.dat` is an NPY-format [sum T_i, idim] float
array opened with np.load(mmap_mode='r'); ilens.npy / olens.npy / label.npy.  Batching semantics
(BucketSampler's 1-frame buckets, half-batch rule, RNG consumption order; collate's sort + zero pad) are
the reference's, bit for bit (pinned by tests/golden/bucket_sampler.npz and fomaml_toy.npz).

What is different: no DataLoader worker processes.  A shard can be made HBM-resident
(`CommonVoiceDataset.to_device`): with 288 GB per GPU a whole accent shard fits, and collate becomes
one coalesced gather+pad kernel (masr_gather_pad) instead of a host memcpy + pinned upload.
"""
import os
import random
from pathlib import Path

import numpy as np
import torch

BUCKET_SIZE = 1
ILEN_MIN = 2
ILEN_MAX = 10000


def collate_fn(batch):
    """dataset.py:21-33: sort by ilen (desc, stable), zero-pad features, keep labels as a list."""
    batch.sort(key=lambda d: d['ilen'], reverse=True)
    tmax = int(batch[0]['ilen'])
    xs_pad = torch.zeros(len(batch), tmax, batch[0]['feat'].shape[1])
    for b, d in enumerate(batch):
        xs_pad[b, :d['feat'].shape[0]] = d['feat']
    ilens = torch.stack([d['ilen'] for d in batch])
    ys = [d['label'] for d in batch]
    olens = torch.stack([d['olen'] for d in batch])
    return xs_pad, ilens, ys, olens


def collate_rows(dset, idxs):
    """collate_fn for a batch given by its indices, written for the host -> HBM hand-over: the padded batch is assembled
    straight into PINNED host memory with plain memcpys (numpy; no torch CPU kernel, whose thread pool is oversubscribed on a
    cgroup-limited GPU box and which would make the later upload a staged pageable copy), so the engine's
    `.to(device, non_blocking=True)` is one asynchronous DMA.  Same sort / pad semantics and values as collate_fn."""
    idxs = sorted(idxs, key=lambda i: int(dset.ilens[i]), reverse=True)            # stable, like list.sort(reverse=True)
    lens = [int(dset.ilens[i]) for i in idxs]
    feat = dset.feat
    B, tmax, D = len(idxs), lens[0], feat.shape[1]
    xs_pad = torch.empty((B, tmax, D), dtype=torch.float32, pin_memory=torch.cuda.is_available())
    arr = xs_pad.numpy()
    for b, (i, n) in enumerate(zip(idxs, lens)):
        arr[b, :n] = feat[dset.iptr[i]:dset.iptr[i] + n]
        if n < tmax:
            arr[b, n:] = 0.0
    ilens = torch.from_numpy(np.asarray([dset.ilens[i] for i in idxs], dtype=dset.ilens.dtype))
    ys = [torch.from_numpy(np.asarray(dset.label[dset.optr[i]:dset.optr[i + 1]]).astype(np.int64)) for i in idxs]
    olens = torch.from_numpy(np.asarray([dset.olens[i] for i in idxs], dtype=dset.olens.dtype))
    return xs_pad, ilens, ys, olens


class BucketSampler:
    """dataset.py:35-110.  Buckets are 1 frame wide; python `random` orders the buckets once at construction,
    np.random shuffles inside a bucket lazily when iteration reaches it; batch size is halved for buckets
    beyond half_batch_ilen."""

    def __init__(self, ilens, min_ilen, max_ilen, half_batch_ilen, batch_size, bucket_size=BUCKET_SIZE,
                 bucket_reverse=False, drop_last=False):
        self.ilens = np.asarray(ilens)
        self.batch_size, self.drop_last, self.bucket_reverse = batch_size, drop_last, bucket_reverse
        lo = min(ILEN_MIN, bucket_size) if not min_ilen else min_ilen
        hi = max(ILEN_MAX, int(self.ilens.max())) if not max_ilen else max_ilen
        half = half_batch_ilen if half_batch_ilen else ILEN_MAX
        bins = np.arange(hi, lo, -bucket_size) if bucket_reverse else np.arange(lo, hi, bucket_size)
        which = np.digitize(self.ilens, bins, right=True)
        self.half_idx = np.digitize(half, bins, right=True)
        self.buckets = []
        for b in range(1, len(bins) - 1):                     # first and last bin are dropped, as in the reference
            members = np.where(which == b)[0]
            if len(members):
                self.buckets.append((b, members))
        random.shuffle(self.buckets)

    def _bs(self, bin_idx):
        halved = bin_idx < self.half_idx if self.bucket_reverse else bin_idx > self.half_idx
        return max(1, self.batch_size // 2) if halved else self.batch_size

    def __iter__(self):
        return _BucketIter(self)

    # ---- resumable state (extension: the reference restarts its data stream on --resume).  A bucket's members are shuffled IN
    # PLACE when an epoch reaches it, so the arrangement left by one epoch is the input of the next one's shuffle: the state is
    # the bucket order + every bucket's current arrangement (+ the cursor of a running iterator, _BucketIter.state)
    def state_dict(self):
        return {'buckets': [(int(b), np.array(m, copy=True)) for b, m in self.buckets]}

    def load_state_dict(self, st):
        self.buckets = [(int(b), np.array(m, copy=True)) for b, m in st['buckets']]

    def __len__(self):
        n = 0
        for bin_idx, members in self.buckets:
            bs = self._bs(bin_idx)
            n += len(members) // bs if self.drop_last else (len(members) + bs - 1) // bs
        return n


class _BucketIter:
    """one epoch of a BucketSampler (dataset.py:98-110 as an explicit cursor instead of a generator, so that a half-consumed
    epoch can be saved and resumed): entering a bucket shuffles it (np.random), batches are consecutive runs of its members"""

    def __init__(self, sampler, b=0, off=0):
        self.s, self.b, self.off = sampler, b, off

    def __iter__(self):
        return self

    def __next__(self):
        s = self.s
        while self.b < len(s.buckets):
            bin_idx, members = s.buckets[self.b]
            bs = s._bs(bin_idx)
            if self.off == 0:
                np.random.shuffle(members)
            if self.off < len(members):
                cur = [int(i) for i in members[self.off:self.off + bs]]
                self.off += bs
                if len(cur) == bs or not s.drop_last:
                    if self.off >= len(members):
                        self.b, self.off = self.b + 1, 0
                    return cur
            self.b, self.off = self.b + 1, 0
        raise StopIteration

    def state(self):
        return {'kind': 'bucket', 'b': self.b, 'off': self.off}


class _ListIter:
    """one epoch of a plain (sequential / RandomSampler) loader: a fixed order cut into batches"""

    def __init__(self, order, batch_size, drop_last, pos=0):
        self.order, self.batch_size, self.drop_last, self.pos = order, batch_size, drop_last, pos

    def __iter__(self):
        return self

    def __next__(self):
        while self.pos < len(self.order):
            chunk = self.order[self.pos:self.pos + self.batch_size]
            self.pos += self.batch_size
            if len(chunk) == self.batch_size or not self.drop_last:
                return chunk
        raise StopIteration

    def state(self):
        return {'kind': 'list', 'order': list(self.order), 'pos': self.pos}


class CommonVoiceDataset:
    """dataset.py:116-153: ragged rows of one NPY array + cumulative pointers."""

    def __init__(self, data_dir, is_memmap):
        data_dir = Path(data_dir)
        self._feat_path = data_dir / ('feat.dat' if is_memmap else 'feat.npy')
        self._is_memmap = is_memmap
        self._feat = None                      # opened on first use: a rank that never materialises a batch of this
        self._device = None                    # accent (multi-GPU task sharding) never maps / uploads its features
        self.ilens = np.load(data_dir / 'ilens.npy')
        self.olens = np.load(data_dir / 'olens.npy')
        self.label = np.load(data_dir / 'label.npy')
        assert len(self.ilens) == len(self.olens), "Number of samples should be the same in features and labels"
        self.iptr = np.concatenate([[0], np.cumsum(self.ilens)]).astype(np.int64)
        self.optr = np.concatenate([[0], np.cumsum(self.olens)]).astype(np.int64)
        self.dev_feat = None

    @property
    def feat(self):
        if self._feat is None:
            self._feat = np.load(self._feat_path, mmap_mode='r') if self._is_memmap else np.load(self._feat_path)
        return self._feat

    def __len__(self):
        return len(self.ilens)

    def __getitem__(self, idx):
        return {
            'feat': torch.from_numpy(np.array(self.feat[self.iptr[idx]:self.iptr[idx + 1], :], dtype=np.float32)),
            'ilen': torch.as_tensor(self.ilens[idx]),
            'label': torch.from_numpy(np.asarray(self.label[self.optr[idx]:self.optr[idx + 1]]).astype(np.int64)),
            'olen': torch.as_tensor(self.olens[idx]),
        }

    # ---- HBM-resident shard -------------------------------------------------------------
    def to_device(self, device, lazy=False):
        """Upload the whole shard once (coalesced HBM reads afterwards; SURVEY 8(d) 'ragged gather').
        lazy: upload when the first batch is gathered (a rank only pays for the accents it actually runs)."""
        self._device = device
        if not lazy:
            self._upload()
        return self

    def _upload(self):
        self.dev_feat = torch.from_numpy(np.ascontiguousarray(self.feat, dtype=np.float32)).to(self._device)

    @property
    def on_device(self):
        return self._device is not None

    def gather_batch(self, idxs):
        """collate on the GPU: same sort / pad semantics as collate_fn, features never leave HBM."""
        import ctypes as C
        from .._cabi import lib, check
        assert self._device is not None, "call to_device() first"
        if self.dev_feat is None:
            self._upload()
        idxs = sorted(idxs, key=lambda i: int(self.ilens[i]), reverse=True)
        dev = self.dev_feat.device
        lens = torch.tensor([int(self.ilens[i]) for i in idxs], dtype=torch.int32)
        B, tmax, D = len(idxs), int(lens.max()), self.dev_feat.shape[1]
        xs = torch.empty(B, tmax, D, device=dev)
        # row starts | lengths in ONE pinned block, uploaded asynchronously: a pageable `.to(device)` is a blocking copy that
        # first waits for everything already queued on the stream (the previous meta-step), which idles the GPU while the
        # host then enqueues the next one
        meta = torch.empty(2 * B, dtype=torch.int64, pin_memory=True)
        meta[:B] = torch.tensor([int(self.iptr[i]) for i in idxs], dtype=torch.int64)
        meta[B:] = lens.to(torch.int64)
        meta_d = meta.to(dev, non_blocking=True)
        rows_d, lens_d = meta_d[:B], meta_d[B:].to(torch.int32)
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        check(lib().masr_gather_pad(C.c_void_p(self.dev_feat.data_ptr()), C.c_void_p(rows_d.data_ptr()),
                                    C.c_void_p(lens_d.data_ptr()), C.c_void_p(xs.data_ptr()), B, tmax, D, s), "masr_gather_pad")
        ys = [torch.from_numpy(np.asarray(self.label[self.optr[i]:self.optr[i + 1]]).astype(np.int64)) for i in idxs]
        olens = torch.tensor([int(self.olens[i]) for i in idxs], dtype=torch.int64)
        return xs, lens.to(torch.int64), ys, olens


class _LazyListIter:
    """plain loader: like the generator it replaces, the order (and the RandomSampler's seed draw from the torch stream) is only
    made when the first batch is asked for"""

    def __init__(self, loader):
        self.loader, self.it = loader, None

    def __iter__(self):
        return self

    def __next__(self):
        if self.it is None:
            ld = self.loader
            order = list(ld.indices)
            if ld.shuffle:
                # torch.utils.data.RandomSampler: draw a seed from the default generator, permute with a private one
                g = torch.Generator()
                g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
                order = [order[i] for i in torch.randperm(len(order), generator=g).tolist()]
            self.it = _ListIter(order, ld.batch_size, ld.drop_last)
        return next(self.it)

    def state(self):
        return {'kind': 'lazy'} if self.it is None else self.it.state()


_PREFETCH_POOL = None


def _prefetch_pool():
    global _PREFETCH_POOL
    if _PREFETCH_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _PREFETCH_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="masr-prefetch")
    return _PREFETCH_POOL


_UPLOAD_STREAMS = {}


def _upload_stream(dev):
    st = _UPLOAD_STREAMS.get(dev)
    if st is None:
        st = _UPLOAD_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return st


class Loader:
    """In-process stand-in for the reference's DataLoader(batch_sampler=..., collate_fn=...)."""

    def __init__(self, dset, batch_sampler=None, batch_size=1, shuffle=False, drop_last=False, indices=None, prefetch=False, upload_device=None):
        self.dset, self.batch_sampler, self.batch_size = dset, batch_sampler, batch_size
        self.shuffle, self.drop_last, self.prefetch = shuffle, drop_last, prefetch
        # prefetching loaders of host-resident shards also start the batch's upload (a copy stream of their own, from the prefetch thread):
        # the 5 MB DMA of a 16 x 1000 x 80 batch then runs under the previous step instead of in front of this one (train.py: +5 %)
        self.upload_device = upload_device
        self.indices = list(range(len(dset))) if indices is None else list(indices)

    def _batches(self):
        if self.batch_sampler is not None:
            return iter(self.batch_sampler)
        return _LazyListIter(self)

    def restore_iter(self, st):
        """the index iterator a saved state (`.state()` of a running one) describes"""
        if st['kind'] == 'bucket':
            return _BucketIter(self.batch_sampler, st['b'], st['off'])
        if st['kind'] == 'lazy':
            return _LazyListIter(self)
        return _ListIter(list(st['order']), self.batch_size, self.drop_last, st['pos'])

    def iter_indices(self):
        """the batch INDEX stream alone: consumes the RNG streams exactly like __iter__ but touches no features.
        torch's DataLoader draws a base seed from the default generator every time an iterator is created
        (_BaseDataLoaderIter.__init__), BEFORE the sampler's own draws: replayed here, so that whatever consumes the torch
        stream next -- the model initialisation of the CLIs, which runs after load_data() -- sees the reference's values."""
        torch.empty((), dtype=torch.int64).random_()
        return self._batches()

    def materialize(self, idxs):
        if self.dset.on_device:
            return self.dset.gather_batch(idxs)
        return collate_rows(self.dset, idxs)

    def _materialize_ahead(self, idxs):
        batch = self.materialize(idxs)
        dev = self.upload_device
        xs = batch[0]
        if dev is None or xs.is_cuda or not xs.is_pinned():
            return batch
        st = _upload_stream(dev)
        with torch.cuda.device(dev), torch.cuda.stream(st):
            xd = xs.to(torch.device("cuda", dev), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        # the consumer's stream waits for `ready` before it reads (MasrEngine.run_batch); the pinned source lives as long as the copy's target
        xd._masr_ready, xd._masr_host = ev, xs
        return (xd,) + tuple(batch[1:])

    def __iter__(self):
        it = self.iter_indices()
        if not (self.prefetch and self.batch_sampler is not None and not self.dset.on_device):
            return (self.materialize(idxs) for idxs in it)
        return self._prefetched(it)

    def _prefetched(self, it):
        """bucketed host shards with num_workers > 0: the next batch is assembled (into pinned memory) by a background thread
        while the caller's step runs on the GPU -- what the reference's DataLoader workers do.  Its indices are drawn one batch
        early: the bucket sampler reads `np.random` only, which nothing else in the train loops touches between two batches
        (evaluate()'s dev loaders and the RandomSampler read the torch stream, hence bucketed loaders only)."""
        fut = None
        for idxs in it:
            nxt = _prefetch_pool().submit(self._materialize_ahead, idxs)
            if fut is not None:
                yield fut.result()
            fut = nxt
        if fut is not None:
            yield fut.result()

    def __len__(self):
        if self.batch_sampler is not None:
            return len(self.batch_sampler)
        n = len(self.indices)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size


def get_loader(data_dir, batch_size, is_memmap, is_bucket, num_workers=0, split_rate=1.0, split_seed=531,
               min_ilen=None, max_ilen=None, half_batch_ilen=None, bucket_reverse=False, shuffle=True,
               read_file=False, drop_last=False, pin_memory=True, device=None, lazy_upload=False):
    """dataset.py:156-198.  pin_memory is accepted for signature parity; num_workers > 0 = batches of a bucketed loader are assembled
    one ahead by a background thread (Loader._prefetched; the DataContainer has its own collate pool)."""
    assert not read_file, "Load from Kaldi ark haven't been implemented yet"
    dset = CommonVoiceDataset(data_dir, is_memmap)
    if device is not None:
        dset.to_device(device, lazy=lazy_upload)
    indices = None
    if split_rate < 1.0:
        n_tr = int(len(dset) * split_rate)
        perm = torch.randperm(len(dset), generator=torch.Generator().manual_seed(split_seed)).tolist()
        indices = perm[:n_tr]                                  # random_split's first subset
    if is_bucket and split_rate == 1.0:
        sampler = BucketSampler(dset.ilens, min_ilen=min_ilen, max_ilen=max_ilen, half_batch_ilen=half_batch_ilen,
                                batch_size=batch_size, bucket_size=BUCKET_SIZE, bucket_reverse=bucket_reverse,
                                drop_last=drop_last)
        ahead = (torch.cuda.current_device() if (num_workers and num_workers > 0 and device is None and torch.cuda.is_available()
                                                 and os.environ.get("MASR_UPLOAD_AHEAD", "1") != "0") else None)
        return Loader(dset, batch_sampler=sampler, prefetch=bool(num_workers and num_workers > 0), upload_device=ahead)
    return Loader(dset, batch_size=batch_size, shuffle=shuffle, drop_last=drop_last, indices=indices)


class DataContainer:
    """dataset.py:200-277: one endless train iterator per accent + dev loaders.

    Multi-GPU (tasks sharded over ranks): the sampler state of EVERY accent and the global `random` / `np.random`
    streams must advance identically on every rank, or an accent that moves to another rank would be served batches
    that were already consumed.  get_item(..., materialize=False) therefore draws the batch INDICES (all the RNG
    consumption there is) without reading a single feature row; only the owning rank materialises the batch.  Shards
    are opened / uploaded on first materialisation, so a rank only ever maps the accents it runs."""

    def __init__(self, data_dirs, batch_size, dev_batch_size, is_memmap, is_bucket, num_workers=0, min_ilen=None,
                 max_ilen=None, half_batch_ilen=None, bucket_reverse=False, shuffle=True, read_file=False,
                 drop_last=False, pin_memory=True, device=None, lazy_upload=False):
        self.data_dirs = [Path(d) for d in data_dirs]
        self.num_datasets = len(self.data_dirs)
        self.kw = dict(batch_size=batch_size, is_memmap=is_memmap, is_bucket=is_bucket, num_workers=num_workers,
                       min_ilen=min_ilen, max_ilen=max_ilen, half_batch_ilen=half_batch_ilen,
                       bucket_reverse=bucket_reverse, shuffle=shuffle, read_file=read_file, device=device, lazy_upload=lazy_upload)
        self.reload_cnt = 0
        # the reference's DataLoader worker PROCESSES (num_workers) become a small pool of threads that assemble batches into
        # pinned memory (memcpy releases the GIL) while the GPU works on the previous ones; index draws stay on the caller's
        # thread, in the reference's order
        self.pool = None
        if num_workers and num_workers > 0:
            from concurrent.futures import ThreadPoolExecutor
            # two threads are plenty (one 16 x 1000 x 80 batch is ~1.7 ms of memcpy against ~3 ms of GPU work, and batches are
            # drawn a meta-step ahead); more only fight the task threads for the interpreter: measured 4 880 utt/s end to
            # end with 1-2 threads, 4 440 with 8
            self.pool = ThreadPoolExecutor(max_workers=min(int(num_workers), 2), thread_name_prefix="masr-collate")
        self.loaders, self.loader_iters, self.dev_loaders = [], [], []
        for d in self.data_dirs:
            ld = get_loader(d / 'train', **self.kw)
            self.loaders.append(ld)
            self.loader_iters.append(ld.iter_indices())
            self.dev_loaders.append(get_loader(d / 'dev', batch_size=dev_batch_size, is_memmap=is_memmap, is_bucket=False,
                                               num_workers=num_workers, shuffle=False, device=device, lazy_upload=lazy_upload))

    def _reload(self, a):
        old = self.loaders[a]
        self.loaders[a] = get_loader(self.data_dirs[a] / 'train', **self.kw)
        ds, od = self.loaders[a].dset, old.dset                 # keep the mapped / uploaded shard of the old loader
        ds._feat, ds.dev_feat = od._feat, od.dev_feat
        self.loader_iters[a] = self.loaders[a].iter_indices()
        self.reload_cnt += 1

    def get_item(self, accent_idx=None, num=1, materialize=True):
        """-> [(accent, batch)] * num.  materialize=False: [(accent, None)] with the same RNG / iterator side effects;
        materialize='async': [(accent, future)] -- the batch is assembled by the collate pool, `.result()` yields it
        (HBM-resident shards gather on the caller's stream, so they are materialised at once)."""
        out = []
        ids = np.random.randint(self.num_datasets, size=num) if accent_idx is None else np.repeat(accent_idx, num)
        for a in ids:
            try:
                idxs = next(self.loader_iters[a])
            except StopIteration:
                self._reload(a)
                idxs = next(self.loader_iters[a])
            ld = self.loaders[a]
            if materialize == 'async' and self.pool is not None and not ld.dset.on_device:
                out.append((a, self.pool.submit(ld.materialize, idxs)))
            elif materialize == 'async':
                out.append((a, _Ready(ld.materialize(idxs))))
            else:
                out.append((a, ld.materialize(idxs) if materialize else None))
        return out


    # ---- resumable state of the train streams (extension; restored AFTER construction, which consumes the RNG streams itself)
    def state_dict(self):
        return {'reload_cnt': self.reload_cnt,
                'samplers': [ld.batch_sampler.state_dict() if ld.batch_sampler is not None else None for ld in self.loaders],
                'iters': [it.state() for it in self.loader_iters]}

    def load_state_dict(self, st):
        self.reload_cnt = st['reload_cnt']
        for a, (ld, ss, its) in enumerate(zip(self.loaders, st['samplers'], st['iters'])):
            if ss is not None:
                ld.batch_sampler.load_state_dict(ss)
            self.loader_iters[a] = ld.restore_iter(its)


def capture_rng():
    """the three global streams that drive batching and task order (python `random`, np.random, torch default generator)"""
    return {'random': random.getstate(), 'numpy': np.random.get_state(), 'torch': torch.get_rng_state()}


def restore_rng(st):
    random.setstate(st['random'])
    np.random.set_state(st['numpy'])
    torch.set_rng_state(st['torch'])


class _Ready:
    """future-like wrapper of a batch that is already there"""
    def __init__(self, v): self._v = v
    def result(self): return self._v
