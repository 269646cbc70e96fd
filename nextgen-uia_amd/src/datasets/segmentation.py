"""Image / mask data for the segmentation entry points (counterpart of /root/reference/src/datasets/segmentation.py).

The reference reads PNG pairs listed in ../data/NextGen-UIA/segmentation/<dataset>/{train,val,test}.txt with PIL, augments them with torchvision (:14-156) and
hands batches of (image float32 [B, 3, S, S] in [0, 1] — ONE grayscale channel repeated three times, :175, :199-200 — label float32 [B, 1, S, S] in {0, 1},
file names) to the loop, `shuffle=True, drop_last=True` for training and neither for validation / test (:223-251).  PIL / torchvision I/O and augmentation are
host-side work outside the hot path (and absent from the build image); what the hot path needs is that batch contract and the three splits.  `--synthetic`
supplies them deterministically (U[0,1) grayscale images with a random-ellipse mask); `--data_pt` takes real data as a .pt file of
{"images": [N, 1 or 3, S, S] uint8 / float, "labels": [N, 1, S, S], optional "names", optional "split": {"train": idx, "val": idx, "test": idx}}.

Because the three channels are copies of one, a batch travels host -> device as ONE channel (float32) plus a uint8 mask — 32 MB instead of 103 MB at 128 images —
through the same shared-memory slot ring the fine-tune loader uses; the towers' patch embedding takes the one channel with the channel-summed kernel
(uia_hip.functional.gray_conv_weight — the same convolution), `as_model_input(..., widen=True)` repeats the channel on the device for callers that want the reference's tensor.
"""
import torch
from torch.utils.data import DataLoader, Dataset

from src.datasets.finetune import RankShardSampler, SharedBatchRing, plan_workers


def synthetic_sample(size, seed):
    """One (image [1, S, S] float32, label [1, S, S] uint8) pair: U[0,1) noise and an axis-aligned ellipse."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(1, size, size, generator=g)
    c = torch.rand(2, generator=g) * size * 0.5 + size * 0.25
    r = torch.rand(2, generator=g) * size * 0.2 + size * 0.08
    yy = torch.arange(size, dtype=torch.float32)
    half = r[1] * torch.sqrt(torch.clamp(1 - ((yy - c[0]) / r[0]) ** 2, min=0))      # half-width of the ellipse on every row
    mask = ((yy[None, :] - c[1]).abs() <= half[:, None]) & ((yy - c[0]).abs() <= r[0])[:, None]
    return img, mask[None].to(torch.uint8)


def synthetic_batch(B, size, seed, device):
    """A resident batch (images [B, 3, S, S], labels [B, 1, S, S] float32) — bench.py's clipseg line and the biomedclip segmentation loop use it."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 1, size, size, generator=g).repeat(1, 3, 1, 1)
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    c = torch.rand(B, 2, generator=g) * size * 0.5 + size * 0.25
    r = torch.rand(B, 2, generator=g) * size * 0.2 + size * 0.08
    mask = (((yy[None] - c[:, 0, None, None]) / r[:, 0, None, None]) ** 2 + ((xx[None] - c[:, 1, None, None]) / r[:, 1, None, None]) ** 2) <= 1
    return img.to(device), mask[:, None].float().to(device)


def as_model_input(images, labels, in_channels=3, bufs=None, widen=True):
    """Device side of the batch contract: the grayscale channel repeated (reference :199-200), the mask as float32.
    widen=False: the one-channel batch goes to the towers as it is — their patch embedding takes it with the channel-summed kernel (UF.gray_conv_weight), which is
    the same convolution; the segmentation loops use it (no 77 MB copy written per iteration).
    bufs: a dict the caller keeps — a widened batch is written into the same two tensors every iteration (one copy kernel each, no allocation)."""
    if not widen and images.shape[1] == 1 and in_channels == 3:
        return images.float(), labels.float()
    if bufs is not None:
        key = (tuple(images.shape), in_channels, images.device)
        if key not in bufs:
            c = 3 if (images.shape[1] == 1 and in_channels == 3) else images.shape[1]
            bufs[key] = (torch.empty((images.shape[0], c) + tuple(images.shape[2:]), dtype=torch.float32, device=images.device),
                         torch.empty(labels.shape, dtype=torch.float32, device=labels.device))
        im, lab = bufs[key]
        im.copy_(images.expand_as(im) if images.shape[1] != im.shape[1] else images)
        lab.copy_(labels)
        return im, lab
    if images.shape[1] == 1 and in_channels == 3:
        images = images.expand(-1, 3, -1, -1).contiguous()
    return images.float(), labels.float()


class SyntheticSegmentation(Dataset):
    """n deterministic samples.  A sample costs ~0.4 ms of host time to generate (the noise draw above all) — 50 ms per 128-image batch per worker, more than the
    GPU step — so the dataset keeps what it generated in SHARED memory allocated before the workers fork (the usual in-RAM cache of a small medical dataset: BUSI has
    780 images): after the first epoch a sample is two views.  `cache_bytes` bounds it (default 4 GiB and a quarter of what /dev/shm has free); beyond it samples
    are generated on every access."""

    def __init__(self, n, img_size, seed, prefix, cache_bytes=4 << 30):
        self.n, self.img_size, self.seed, self.prefix = n, img_size, seed, prefix
        self._img = self._lab = self._have = None
        need = n * img_size * img_size * 5
        try:
            import shutil
            room = shutil.disk_usage("/dev/shm").free // 4
        except OSError:
            room = 0
        if 0 < need <= min(cache_bytes, room):
            self._img = torch.empty(n, 1, img_size, img_size).share_memory_()          # pages are committed when a worker first writes them
            self._lab = torch.empty(n, 1, img_size, img_size, dtype=torch.uint8).share_memory_()
            self._have = torch.zeros(n, dtype=torch.bool).share_memory_()

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        name = f"{self.prefix}_{i:05d}.png"
        if self._have is not None and bool(self._have[i]):
            return self._img[i], self._lab[i], name
        img, lab = synthetic_sample(self.img_size, self.seed * 1000003 + i)
        if self._have is not None:
            self._img[i].copy_(img)
            self._lab[i].copy_(lab)
            self._have[i] = True                                   # set after the data: a reader that sees the flag sees the sample (one writer per index at a time is
        return img, lab, name                                      # not guaranteed, but every writer writes the same bytes)


class TensorSegmentation(Dataset):
    def __init__(self, images, labels, names):
        self.images, self.labels, self.names = images, labels, names

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        img = self.images[i]
        img = img.float() / 255.0 if img.dtype == torch.uint8 else img.float()
        return img[:1], (self.labels[i][:1] > 0).to(torch.uint8), self.names[i]      # the reference converts every image to one grayscale channel (:175)


class SegBatchRing(SharedBatchRing):
    """SharedBatchRing whose second tensor is the uint8 mask: a worker's collate writes [B, 1, S, S] images and masks into a free shared slot and returns
    (mark, slot, names)."""

    def __init__(self, slots, batch, img_size):
        super().__init__(slots, batch, (1, img_size, img_size), 0, None, second_shape=(1, img_size, img_size), second_dtype=torch.uint8)

    def __call__(self, samples):
        names = [s[2] for s in samples]
        if len(samples) != self.batch:                         # the ragged last batch of a validation / test split travels the ordinary way
            return torch.stack([s[0] for s in samples]), torch.stack([s[1] for s in samples]), names
        slot = self.free.get()
        torch.stack([s[0] for s in samples], out=self.images[slot])
        torch.stack([s[1] for s in samples], out=self.ids[slot])
        return self.MARK, slot, names


def _plain_collate(samples):
    return torch.stack([s[0] for s in samples]), torch.stack([s[1] for s in samples]), [s[2] for s in samples]


def second_of(batch):
    """What engine.DevicePrefetcher copies beside the images: the masks."""
    return batch[1]


class DataModule:
    MAX_WORKERS = 6

    def __init__(self, args, rank=None, world=None):
        import os
        self.args = args
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else world
        if getattr(args, "data_pt", None):
            blob = torch.load(args.data_pt)
            n = len(blob["images"])
            names = list(blob.get("names") or [f"{i:05d}.png" for i in range(n)])
            split = blob.get("split")
            if split is None:                                   # 70 / 10 / 20 in file order (the reference's lists are pre-shuffled text files, :210-216)
                a, b = int(n * 0.7), int(n * 0.8)
                split = {"train": list(range(a)), "val": list(range(a, b)), "test": list(range(b, n))}
            mk = lambda idx: TensorSegmentation(blob["images"][idx], blob["labels"][idx], [names[i] for i in idx])
            self.train_dataset, self.val_dataset, self.test_dataset = mk(split["train"]), mk(split["val"]), mk(split["test"])
        elif getattr(args, "synthetic", False):
            self.train_dataset = SyntheticSegmentation(args.synthetic_train, args.img_size, args.seed, "train")
            self.val_dataset = SyntheticSegmentation(args.synthetic_val, args.img_size, args.seed + 1, "val")
            self.test_dataset = SyntheticSegmentation(args.synthetic_test, args.img_size, args.seed + 2, "test")
        else:
            raise RuntimeError("no dataset: pass --synthetic or --data_pt (the reference's PIL/torchvision loaders read ../data/NextGen-UIA and are outside this build)")
        self.train_sampler = None
        self._loaders = []

    def _loader(self, ds, train):
        a = self.args
        nw = max(0, min(int(getattr(a, "num_workers", 0) or 0), self.MAX_WORKERS if train else 2))
        if nw and torch.cuda.is_initialized():
            import logging
            logging.info("loader: the GPU is already initialised in this process; loading in-process (num_workers=0)")
            nw = 0
        per_slot = a.batch_size * a.img_size ** 2 * 5          # one float32 channel + one uint8 mask
        nw, slots = plan_workers(nw, per_slot, owner=self)
        kw = dict(num_workers=nw, drop_last=train)              # reference :223-251
        kw["collate_fn"] = SegBatchRing(slots, a.batch_size, a.img_size) if nw else _plain_collate
        if nw:
            kw.update(persistent_workers=True, prefetch_factor=2)
        if train and self.world > 1:                            # every rank trains on its own shard; validation / test run whole on every rank (identical weights ->
            sampler = RankShardSampler(len(ds), self.rank, self.world, shuffle=True, seed=getattr(a, "seed", 0), drop_last=True)    # identical decisions, no collective)
            self.train_sampler = sampler
            loader = DataLoader(ds, batch_size=a.batch_size, sampler=sampler, **kw)
        else:
            loader = DataLoader(ds, batch_size=a.batch_size, shuffle=train, **kw)
        self._loaders.append(loader)
        return loader

    def train_dataloader(self):
        return self._loader(self.train_dataset, True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, False)

    def test_dataloader(self):
        return self._loader(self.test_dataset, False)

    def start_workers(self):
        """Fork the worker processes before the process touches the GPU (datasets.finetune.DataModule.start_workers)."""
        for loader in self._loaders:
            if loader.num_workers > 0 and len(loader) > 0:
                loader._uia_first_iter = iter(loader)

    def set_epoch(self, epoch):
        if self.train_sampler is not None:
            self.train_sampler.set_epoch(epoch)

    def shutdown(self):
        for loader in self._loaders:
            if hasattr(loader.collate_fn, "unpin"):
                loader.collate_fn.unpin()                       # the ring's host registration must not outlive its mapping
            it = getattr(loader, "_iterator", None)
            if it is not None and hasattr(it, "_shutdown_workers"):
                it._shutdown_workers()
            loader._iterator = None
