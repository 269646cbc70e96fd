"""Image-caption data for the contrastive fine-tune (counterpart of /root/reference/src/datasets/finetune.py).

The reference reads CSV-listed image/caption pairs from hard-coded roots with PIL/torchvision (:11-142) — host-side I/O that
is outside the hot path and whose dependencies are absent from the build image.  What the hot path needs from it is the
batch contract: images float32 [B,3,S,S] in [0,1] WITHOUT mean/std normalisation (:17-24) and a list of caption strings,
`drop_last=True`.  `--synthetic` provides exactly that deterministically; real data can be supplied as a .pt file of
{"images": uint8/float [N,3,S,S], "texts": [str]*N} via --data_pt.
"""
import os

import torch
from torch.utils.data import DataLoader, Dataset

_WORDS = ("breast ultrasound image showing a benign malignant tumor lesion mass with irregular smooth margins hypoechoic "
          "shadowing cystic solid nodule thyroid liver kidney scan tissue region calcification").split()


class SyntheticPairs(Dataset):
    def __init__(self, n, img_size, seed):
        self.n, self.img_size, self.seed = n, img_size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        img = torch.rand(1, self.img_size, self.img_size, generator=g).repeat(3, 1, 1)      # grayscale repeated, U[0,1)
        k = int(torch.randint(6, 40, (1,), generator=g))
        words = [_WORDS[int(j)] for j in torch.randint(0, len(_WORDS), (k,), generator=g)]
        return img, " ".join(words)


class TensorPairs(Dataset):
    def __init__(self, blob):
        self.images, self.texts = blob["images"], blob["texts"]

    def __len__(self):
        return len(self.texts)

    def __getitem__(self, i):
        img = self.images[i]
        return (img.float() / 255.0 if img.dtype == torch.uint8 else img.float()), self.texts[i]


class RankShardSampler(torch.utils.data.Sampler):
    """torch.utils.data.DistributedSampler semantics without needing an initialised process group: every rank draws the SAME
    permutation of the dataset (seeded by seed + epoch), the index list is padded by wrap-around (or truncated with
    drop_last) to a multiple of `world`, and rank r takes indices r, r + world, r + 2·world, ...  Every rank therefore sees
    the same NUMBER of samples — the loops' collectives stay aligned — and the ranks' shards are disjoint: R ranks x 1
    micro-batch is the reference's accumulation over R different micro-batches (finetune.py:287-302)."""

    def __init__(self, n, rank=0, world=1, shuffle=True, seed=0, drop_last=False):
        assert 0 <= rank < world
        self.n, self.rank, self.world, self.shuffle, self.seed, self.drop_last = n, rank, world, shuffle, seed, drop_last
        self.epoch = 0
        self.per_rank = n // world if drop_last else (n + world - 1) // world

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def indices(self):
        if self.shuffle:
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist()
        else:
            order = list(range(self.n))
        total = self.per_rank * self.world
        if total > len(order):
            order = (order * (total // max(len(order), 1) + 1))[:total]
        return order[:total][self.rank::self.world]

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        return self.per_rank


class SharedBatchRing:
    """A ring of batch-sized slots in SHARED memory that the loader workers collate into and the training process copies to the GPU from.

    torch's DataLoader hands every batch over as a fresh shared-memory segment: at 256 x 3 x 224 x 224 fp32 that is a 154 MB mapping created by a worker,
    mapped, copied to a pinned staging buffer and unmapped again by the training process — and the unmap (37 K pages) runs in the tensor's destructor with the
    interpreter lock held, which stalled the thread that enqueues the GPU work for ~15 ms per batch (the fine-tune CLI took 46-49 ms per update against 40 for
    the same loop on a resident batch; a stack sample every 7 ms showed it).  With the ring nothing is mapped or unmapped per batch: the slots are allocated
    ONCE before the workers are forked, a worker's collate writes images and token ids straight into a free slot and returns only (slot, captions), the
    training process registers the ring as pinned host memory (hipHostRegister) once its GPU is up and issues the host-to-device copy from the slot itself —
    no staging copy either — and hands the slot back through a queue when the copy has run."""

    MARK = "__uia_ring__"

    def __init__(self, slots, batch, img_shape, ids_len, tokenizer, second_shape=None, second_dtype=torch.int64):
        import multiprocessing as mp
        self.images = torch.empty((slots, batch) + tuple(img_shape), dtype=torch.float32).share_memory_()
        self.ids = torch.zeros((slots, batch) + (tuple(second_shape) if second_shape is not None else (ids_len,)), dtype=second_dtype).share_memory_()      # token ids, or the segmentation loaders' masks
        self.free = mp.get_context("fork").Queue()
        for i in range(slots):
            self.free.put(i)
        self.slots, self.batch, self.tokenizer, self.pinned = slots, batch, tokenizer, None

    def __call__(self, samples):                               # the DataLoader's collate_fn: runs in a worker (or in-process with num_workers=0)
        texts = [s[1] for s in samples]
        if len(samples) != self.batch:                         # (drop_last=True never produces one; a ragged batch travels the ordinary way)
            images = torch.stack([s[0] for s in samples])
            return images, texts, self.tokenizer(texts)
        slot = self.free.get()
        torch.stack([s[0] for s in samples], out=self.images[slot])
        self.ids[slot].copy_(self.tokenizer(texts))
        return self.MARK, slot, texts

    def pin(self):
        """hipHostRegister the ring (once, from the process that owns the GPU): host-to-device copies from a slot are then asynchronous DMA.  False when the runtime
        refuses: the consumer stages through its own pinned buffers instead."""
        if self.pinned is None and os.environ.get("UIA_RING_NO_PIN"):      # measurement knob: host-to-device copies through the consumer's own pinned staging buffers
            self.pinned = False
        if self.pinned is None:
            ok = True
            try:
                for t in (self.images, self.ids):
                    rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)
                    ok = ok and int(rc) == 0
                ok = ok and bool(self.images.is_pinned())       # what Tensor.copy_(non_blocking=True) looks at
            except (AttributeError, RuntimeError):
                ok = False
            self.pinned = ok
        return self.pinned

    def unpin(self):
        """hipHostUnregister the ring (DataModule.shutdown): a registration outlives the mapping it names — once the ring's shared memory is unmapped, later host
        allocations land in the same address range, still 'registered', and the first host-to-device copy from them fails with hipErrorInvalidValue (round 6: the
        CLIPSeg entry point builds a second model for test() in the process that trained)."""
        if self.pinned:
            try:
                torch.cuda.synchronize()                           # every copy that reads a slot has run
                for t in (self.images, self.ids):
                    torch.cuda.cudart().cudaHostUnregister(t.data_ptr())
            except (AttributeError, RuntimeError):
                pass
        self.pinned = None

    def release(self, slot):
        self.free.put(slot)


def ring_slots(nw):
    """Slots a loader with nw worker processes needs: every batch a worker may have in flight (prefetch_factor 2 per worker) plus the consumer's pipeline."""
    return 2 * nw + 4


def plan_workers(nw, per_slot_bytes, owner=None):
    """(workers, ring slots) that FIT: the ring of ring_slots(nw) batch-sized slots is allocated up front in /dev/shm (pages are committed on first touch, and a
    write to a page that cannot be had raises SIGBUS, not an exception) — so the count is cut until the whole ring, plus the rings `owner` (a DataModule: it
    builds two or three loaders) planned before it, fits into half of what is free; no room for even a one-worker ring means in-process loading (0, 0)."""
    import shutil
    if nw <= 0:
        return 0, 0
    planned = getattr(owner, "_shm_planned", 0)
    try:
        room = shutil.disk_usage("/dev/shm").free // 2 - planned
    except OSError:
        room = 0
    while nw > 0 and ring_slots(nw) * per_slot_bytes > room:
        nw -= 1
    if nw == 0:
        return 0, 0
    if owner is not None:
        owner._shm_planned = planned + ring_slots(nw) * per_slot_bytes
    return nw, ring_slots(nw)


class _TokenisingCollate:
    """default_collate plus the caller's tokenizer applied to the caption list INSIDE the loader worker: batches arrive as (images, texts, token ids).  The reference
    tokenises in the training loop (finetune.py:275 `tokenizer(texts)`); done there it would hold the interpreter lock of the process that enqueues the GPU work for
    tens of milliseconds per batch."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def __call__(self, samples):
        images, texts = torch.utils.data.default_collate(samples)
        return images, texts, self.tokenizer(list(texts))


class DataModule:
    def __init__(self, args, rank=None, world=None, tokenizer=None):
        """rank / world default to the torch.distributed.run environment (RANK / WORLD_SIZE); world == 1 is the reference's
        single-process loader (shuffle=True, drop_last=True, datasets/finetune.py:124-142).  tokenizer (optional): applied to every batch's captions in the
        loader workers; batches then carry the token ids as a third element."""
        import os
        self.args = args
        self.tokenizer = tokenizer
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else world
        if getattr(args, "data_pt", None):
            blob = torch.load(args.data_pt)
            n_val = max(args.batch_size, len(blob["texts"]) // 10)
            self.train = TensorPairs({"images": blob["images"][n_val:], "texts": blob["texts"][n_val:]})
            self.val = TensorPairs({"images": blob["images"][:n_val], "texts": blob["texts"][:n_val]})
        elif getattr(args, "synthetic", False):
            self.train = SyntheticPairs(args.synthetic_train, args.img_size, args.seed)
            self.val = SyntheticPairs(args.synthetic_val, args.img_size, args.seed + 1)
        else:
            raise RuntimeError("no dataset: pass --synthetic or --data_pt (the reference's CSV/PIL loaders need torchvision, "
                               "which is outside this build)")
        self.train_sampler = self.val_sampler = None

    MAX_WORKERS = 6         # loader processes per rank (the reference's --num_workers default is 8; a one-GPU job's CPU share on the MI355X boxes is 16)

    def _loader(self, ds, shuffle):
        nw = max(0, min(int(getattr(self.args, "num_workers", 0) or 0), self.MAX_WORKERS if shuffle else 2))     # the validation loader: two workers
        if nw and torch.cuda.is_initialized():
            # worker processes are forked, and a child forked from a process with a live HIP runtime inherits its device handles: fork before the GPU is touched (the entry
            # points build the DataModule and call start_workers() first) or load in-process
            import logging
            logging.info("loader: the GPU is already initialised in this process; loading in-process (num_workers=0)")
            nw = 0
        slots = 0
        if nw:
            # the ring of shared slots is sized from what it really allocates: images + token ids per slot, ring_slots(nw) of them, both loaders counted
            ids_len = int(self.tokenizer(["x"]).shape[1]) if self.tokenizer is not None else 0
            nw, slots = plan_workers(nw, self.args.batch_size * (3 * self.args.img_size ** 2 * 4 + ids_len * 8), owner=self)
        kw = dict(num_workers=nw, drop_last=True)
        if self.tokenizer is not None and nw:
            # worker processes: batches travel through a ring of shared slots (SharedBatchRing)
            kw.update(collate_fn=SharedBatchRing(slots, self.args.batch_size, (3, self.args.img_size, self.args.img_size), ids_len, self.tokenizer))
        elif self.tokenizer is not None:
            kw.update(collate_fn=_TokenisingCollate(self.tokenizer))
        if nw:
            kw.update(persistent_workers=True, prefetch_factor=2)
        if self.world > 1:
            sampler = RankShardSampler(len(ds), self.rank, self.world, shuffle=shuffle, seed=getattr(self.args, "seed", 0))
            return DataLoader(ds, batch_size=self.args.batch_size, sampler=sampler, **kw), sampler
        return DataLoader(ds, batch_size=self.args.batch_size, shuffle=shuffle, **kw), None

    def start_workers(self):
        """Fork the loaders' worker processes NOW — the entry points call this before the process has touched the GPU, so that no child is forked from a process
        with a live HIP runtime.  (Persistent workers: every later iter(loader) re-uses them.)"""
        for loader in getattr(self, "_loaders", []):
            if loader.num_workers > 0 and len(loader) > 0:
                loader._uia_first_iter = iter(loader)         # the consumer's first epoch takes THIS iterator: a second iter() would reset it and drop the batches already in flight

    def shutdown(self):
        for loader in getattr(self, "_loaders", []):
            if hasattr(getattr(loader, "collate_fn", None), "unpin"):
                loader.collate_fn.unpin()
            it = getattr(loader, "_iterator", None)
            if it is not None and hasattr(it, "_shutdown_workers"):
                it._shutdown_workers()
            loader._iterator = None

    def train_dataloader(self):
        loader, self.train_sampler = self._loader(self.train, True)
        self._loaders = getattr(self, "_loaders", []) + [loader]
        return loader

    def val_dataloader(self):
        loader, self.val_sampler = self._loader(self.val, False)
        self._loaders = getattr(self, "_loaders", []) + [loader]
        return loader

    def set_epoch(self, epoch):
        """Reshuffle the rank shards for a new epoch (DistributedSampler.set_epoch); no-op in a single process."""
        if self.train_sampler is not None:
            self.train_sampler.set_epoch(epoch)
