"""Image-caption data for the contrastive fine-tune (counterpart of /root/reference/src/datasets/finetune.py).

The reference reads CSV-listed image/caption pairs from hard-coded roots with PIL/torchvision (:11-142) — host-side I/O that
is outside the hot path and whose dependencies are absent from the build image.  What the hot path needs from it is the
batch contract: images float32 [B,3,S,S] in [0,1] WITHOUT mean/std normalisation (:17-24) and a list of caption strings,
`drop_last=True`.  `--synthetic` provides exactly that deterministically; real data can be supplied as a .pt file of
{"images": uint8/float [N,3,S,S], "texts": [str]*N} via --data_pt.
"""
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


class DataModule:
    def __init__(self, args):
        self.args = args
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

    def train_dataloader(self):
        return DataLoader(self.train, batch_size=self.args.batch_size, shuffle=True, num_workers=0, drop_last=True)

    def val_dataloader(self):
        return DataLoader(self.val, batch_size=self.args.batch_size, shuffle=False, num_workers=0, drop_last=True)
