"""Synthetic COCO-shaped mini-batches in the `batched_inputs` format the meta-arch consumes
[ref: utils/dataset_mapper.py:136-355 emits dicts with "image" (3,H,W) uint8 BGR, "instances",
"height", "width"].  Real-image IO is out of scope."""
import torch

from . import synth
from .structures import Boxes, Instances


def synthetic_batch(B, h=800, w=1333, n_boxes=10, seed=0, table=False, device="cpu", pin=False):
    gts = synth.synth_gt(B, h, w, n_boxes, seed=seed, table=table)
    imgs = synth.synth_images(B, h, w, seed=seed + 1)
    out = []
    for b in range(B):
        img = torch.from_numpy(imgs[b])
        boxes = torch.from_numpy(gts[b][0].copy())
        cls = torch.from_numpy(gts[b][1].copy())
        if pin and torch.cuda.is_available():
            img, boxes, cls = img.pin_memory(), boxes.pin_memory(), cls.pin_memory()
        out.append({"image": img.to(device), "height": h, "width": w,
                    "instances": Instances((h, w), gt_boxes=Boxes(boxes.to(device)), gt_classes=cls.to(device))})
    return out
