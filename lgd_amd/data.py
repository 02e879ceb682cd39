"""Synthetic COCO-shaped mini-batches in the `batched_inputs` format the meta-arch consumes
[ref: utils/dataset_mapper.py:136-355 emits dicts with "image" (3,H,W) uint8 BGR, "instances",
"height", "width"].  Real-image IO is out of scope."""
import torch

from . import synth
from .structures import Boxes, Instances


def multiscale_sizes(B, h=800, w=1333, min_sizes=(640, 672, 704, 736, 768, 800), max_size=1333, seed=0):
    """per-image (height, width) of detectron2's ResizeShortestEdge with MIN_SIZE_TRAIN_SAMPLING "choice"
    [ref: configs/Base-RetinaNet.yaml:26 MIN_SIZE_TRAIN (640,...,800); utils/dataset_mapper.py:257-355; d2-memory]: the short side of
    every image is drawn from `min_sizes`, the long side follows the aspect ratio of the (h x w) source image and is capped at
    `max_size` (then the short side shrinks with it).  Deterministic (hash of `seed`)."""
    u = synth.det_uniform((B,), 50021 + seed, 0.0, 1.0)
    out = []
    for b in range(B):
        s = min_sizes[min(int(u[b] * len(min_sizes)), len(min_sizes) - 1)]
        scale = s / min(h, w)
        if max(h, w) * scale > max_size:
            scale = max_size / max(h, w)
        out.append((int(h * scale + 0.5), int(w * scale + 0.5)))
    return out


def synthetic_batch(B, h=800, w=1333, n_boxes=10, seed=0, table=False, device="cpu", pin=False, sizes=None):
    """sizes: optional per-image (height, width) list (multi-scale training: the images of a batch differ in size and the meta-arch
    pads them to the batch maximum rounded up to 32, detectron2 ImageList.from_tensors); default: every image h x w."""
    sizes = list(sizes) if sizes is not None else [(h, w)] * B
    base_gt = base_img = None
    if any(tuple(sz) == (h, w) for sz in sizes):   # images at the source size: the fixed-size batch's arrays (bit-identical to it)
        base_gt = synth.synth_gt(B, h, w, n_boxes, seed=seed, table=table)
        base_img = synth.synth_images(B, h, w, seed=seed + 1)
    out = []
    for b in range(B):
        hb, wb = sizes[b]
        if (hb, wb) == (h, w):
            gt, img = base_gt[b], base_img[b]
        else:
            gt = synth.synth_gt(1, hb, wb, n_boxes, seed=seed * 131 + b, table=table)[0]
            img = synth.synth_images(1, hb, wb, seed=(seed + 1) * 977 + b)[0]
        img = torch.from_numpy(img)
        boxes = torch.from_numpy(gt[0].copy())
        cls = torch.from_numpy(gt[1].copy())
        if pin and torch.cuda.is_available():
            img, boxes, cls = img.pin_memory(), boxes.pin_memory(), cls.pin_memory()
        out.append({"image": img.to(device), "height": hb, "width": wb,
                    "instances": Instances((hb, wb), gt_boxes=Boxes(boxes.to(device)), gt_classes=cls.to(device))})
    return out


class DevicePrefetcher:
    """Hands the step DEVICE batches while the loader produces HOST batches: the host -> device copy of batch k + 1 is issued on a side
    stream while step k runs (pinned tensors, non-blocking copies; the step's stream waits for the copy's event, and the tensors are
    recorded on it so that the caching allocator does not recycle them under the step).  The reference copies inside the step
    (models/customized_detectors/retinanet.py:48, `x["image"].to(self.device)` in preprocess_image); here the same bytes move per step,
    off the step's critical path (bench.py `host_batch`: the rate with the batches handed over as host tensors)."""

    def __init__(self, batches, device):
        self.it = iter(batches)
        self.dev = torch.device(device)
        self.stream = torch.cuda.Stream(self.dev)
        self._next = self._ev = None
        self._preload()

    def _move(self, batch):
        out = []
        for x in batch:
            y = dict(x)
            y["image"] = x["image"].to(self.dev, non_blocking=True)
            if "instances" in x:
                y["instances"] = x["instances"].to(self.dev)
            out.append(y)
        return out

    def _preload(self):
        try:
            b = next(self.it)
        except StopIteration:
            self._next = None
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))   # (never ahead of the step that may still read the buffers being recycled)
        with torch.cuda.stream(self.stream):
            self._next = self._move(b)
            self._ev = torch.cuda.Event()
            self._ev.record(self.stream)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(self._ev)
        b = self._next
        for x in b:
            x["image"].record_stream(cur)
            inst = x.get("instances")
            if inst is not None:
                for v in getattr(inst, "_fields", {}).values():
                    t = getattr(v, "tensor", v)
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(cur)
        self._preload()
        return b
