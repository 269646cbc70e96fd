"""Measurement probes of the CLIPSeg entry point's training loop (round 6; not used unless UIA_SEG_AB names them — tools/ab_clipseg_post.sh, ab_clipseg_attrib.sh).

Question they answer: the CLI's epochs run 3-10 % slower per iteration than bench.py's resident-batch loop on the same box although both are GPU-bound — is it the CLI's
PROCESS (loader workers, registered rings, allocator state) or its LOOP, and which ingredient of the loop?
  post    bench.py's loop (resident batch, 30 + 120 steps) inside the CLI's process, once with the loaders alive and once after they are shut down
  attrib  the resident loop plus ONE ingredient of the epoch loop at a time: host-to-device copies on a side stream, the wait on their event, the step reading
          rotating freshly-copied buffers, an event record per step; then the resident loop while a thread drains the training prefetcher (loaders working)
Findings: DESIGN.md §4 round 6 item 1."""
import threading
import time

import torch

from src.datasets.segmentation import synthetic_batch
from uia_hip.engine import segmentation_step


class Probe:
    def __init__(self, model, criterion, opt, prompt, cache, args, train_pf):
        self.model, self.criterion, self.opt, self.args, self.train_pf = model, criterion, opt, args, train_pf
        if args.batch_size not in cache:
            cache[args.batch_size] = prompt.repeat(args.batch_size, 1)
        self.bp = cache[args.batch_size]
        self.im, self.lab = synthetic_batch(args.batch_size, args.img_size, 1, args.device)
        self.results = {}

    def _step(self, x, y):
        segmentation_step(self.model, self.criterion, self.opt, x, y, input_ids=self.bp, lr=self.args.lr_min)

    def _timed(self, body, warm=30, n=120):
        for k in range(warm):
            body(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            body(k)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / n * 1e3, 3)

    def resident(self):
        self.model.train()
        return self._timed(lambda k: self._step(self.im, self.lab))

    def before_shutdown(self, ab):
        if "post" in ab:
            self.results["post_resident_ms"] = [self.resident()]           # loader workers alive, rings registered
        if "attrib" in ab:
            self.results["attrib_ms"] = self.attribute()

    def after_shutdown(self, ab):
        if "post" in ab:
            self.results["post_resident_ms"].append(self.resident())       # ... and gone

    def attribute(self):
        a, dev = self.args, self.args.device
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=dev)
        host_im = torch.empty(a.batch_size, 1, a.img_size, a.img_size, pin_memory=True)
        host_lab = torch.zeros(a.batch_size, 1, a.img_size, a.img_size, dtype=torch.uint8, pin_memory=True)
        dev_im = [torch.empty_like(host_im, device=dev) for _ in range(4)]
        dev_lab = [torch.empty_like(host_lab, device=dev) for _ in range(4)]
        gray = self.im[:, :1].contiguous()

        def variant(copy, wait, fresh, rec):
            def body(k):
                ev = None
                if copy:
                    with torch.cuda.stream(side):
                        dev_im[k % 4].copy_(host_im, non_blocking=True)
                        dev_lab[k % 4].copy_(host_lab, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                if wait and ev is not None:
                    cur.wait_event(ev)
                x = gray
                if fresh:                                        # the step reads a rotating buffer, as the epoch loop does
                    if not copy:
                        dev_im[k % 4].copy_(gray)
                    x = dev_im[k % 4]
                self._step(x, self.lab)
                if rec:
                    torch.cuda.Event().record(cur)
            return self._timed(body)
        out = {"resident gray": variant(False, False, False, False), "+H2D copies on a side stream": variant(True, False, False, False),
               "+copies +wait_event": variant(True, True, False, False), "+copies +wait +step reads the rotating buffers": variant(True, True, True, False),
               "rotating buffers filled by a kernel (no H2D)": variant(False, False, True, False), "+event record per step": variant(False, False, False, True)}
        # the resident loop while the loaders WORK: a thread drains the training prefetcher (workers collate, the ring fills and empties, copies run) beside it.
        # (The drain is unthrottled — it takes the interpreter lock as often as it can — so this figure is an upper bound of what working loaders cost, not the CLI's own.)
        stop = threading.Event()

        def drain():
            torch.cuda.set_device(torch.device(dev))
            while not stop.is_set():
                for _b in self.train_pf:
                    if stop.is_set():
                        break
        th = threading.Thread(target=drain, daemon=True)
        th.start()
        out["resident gray, loaders working beside it (unthrottled drain)"] = variant(False, False, False, False)
        stop.set()
        th.join(timeout=20)
        self.train_pf.close()
        out["resident gray again, loaders idle"] = variant(False, False, False, False)
        return out
