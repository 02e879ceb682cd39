"""Minimal detectron2-shaped data structures the LGD path touches
[ref: label_encoder.py:41-59,166-167 reads `instances.gt_boxes.tensor/.device`, `gt_classes`,
`len(instances)`, `images.tensor`; distillator.py:84 reads `images.image_sizes`]."""
import torch


class Boxes:
    def __init__(self, tensor):
        tensor = torch.as_tensor(tensor, dtype=torch.float32).reshape(-1, 4)
        self.tensor = tensor

    @property
    def device(self):
        return self.tensor.device

    def to(self, device):
        return Boxes(self.tensor.to(device, non_blocking=True))   # a no-wait copy when the loader pinned the annotation

    def __len__(self):
        return self.tensor.shape[0]


class Instances:
    """image_size=(h, w) plus arbitrary per-instance fields (gt_boxes: Boxes, gt_classes: int64)."""

    def __init__(self, image_size, **fields):
        object.__setattr__(self, "_image_size", tuple(image_size))
        object.__setattr__(self, "_fields", {})
        for k, v in fields.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def __setattr__(self, name, value):
        self.set(name, value)

    def __getattr__(self, name):
        f = object.__getattribute__(self, "_fields")
        if name in f:
            return f[name]
        raise AttributeError(name)

    def to(self, device):
        out = Instances(self._image_size)
        for k, v in self._fields.items():
            out.set(k, (v.to(device, non_blocking=True) if torch.is_tensor(v) else v.to(device)) if hasattr(v, "to") else v)
        return out

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0


class ImageList:
    """Batch of images padded bottom/right to a common size divisible by `size_divisibility`
    ([d2-memory] ImageList.from_tensors): .tensor (B,3,Hp,Wp), .image_sizes [(h,w)]."""

    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = [tuple(s) for s in image_sizes]

    def __len__(self):
        return len(self.image_sizes)

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        mh = max(s[0] for s in sizes)
        mw = max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            mh, mw = (mh + d - 1) // d * d, (mw + d - 1) // d * d
        if len(tensors) > 1 and all(s == (mh, mw) for s in sizes):
            return ImageList(torch.stack(tensors, 0), sizes)
        out = tensors[0].new_full((len(tensors), tensors[0].shape[0], mh, mw), pad_value)
        for i, t in enumerate(tensors):
            out[i, :, :t.shape[-2], :t.shape[-1]].copy_(t)
        return ImageList(out, sizes)
