"""Second HIP streams for chains of the step that do not depend on each other (the teacher's label encoder beside the backbone, the box tower of
the head beside the class tower): one process per GPU still, the streams overlap tails and small launches of one chain with the other's kernels.
Autograd runs every backward node on the stream of its forward, so the overlap carries over to the backward pass.
[ref: the reference issues everything on one stream -- train.py:182-215; the chains themselves: thirdparty_heads/fcos.py:520-546,
 detectron2 RetinaNetHead.forward]"""
import weakref

import torch

_SIDE = {}   # (device index, name) -> torch.cuda.Stream
_MAIN = {}   # device index -> the stream the step runs on (the one a fork was last taken from)


def side(device, name):
    """the side stream `name` of a device (created on first use)"""
    key = (device.index if device.index is not None else torch.cuda.current_device(), name)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = torch.cuda.Stream(device)
    return s


def fork(device, name, inputs=()):
    """-> (main, side): the side stream waits for everything issued on the current stream so far; `inputs` (tensors made on the current stream
    that the side stream is about to read) are recorded on it, so that the allocator does not hand their memory out again under it"""
    main = torch.cuda.current_stream(device)
    _MAIN[main.device.index] = main
    s = side(device, name)
    s.wait_stream(main)
    for t in inputs:
        t.record_stream(s)
    return main, s


def record_all(obj, stream):
    """record_stream on every device tensor reachable from obj (tensors, nested lists / tuples / dicts) AND on the magnitude-tag word a tensor
    carries (ops._amax_tag: a slice of the zero-word pool of the stream that PRODUCED the tensor): all of them were allocated on one stream and are
    about to be read on `stream` -- the caching allocator must not hand their memory out again under it"""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
            tag = getattr(obj, "_lgd_amax", None)
            if tag is not None and isinstance(tag[0], torch.Tensor) and tag[0].is_cuda:
                tag[0].record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            record_all(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            record_all(o, stream)


def join(main, s, outputs=()):
    """the current stream waits for the side stream; `outputs` (made on the side stream, read on the current one from here on: tensors or nested
    containers of them) are recorded on it together with their magnitude tags"""
    main.wait_stream(s)
    record_all(outputs, main)


_HOOKED = {}   # id(parameter) -> (weak reference to it, names of the streams it is joined to): kept OUTSIDE the Parameter, whose __dict__ a
               # whole-model pickle / deepcopy carries along while the hooks themselves stay behind


def _multi_rank():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def join_on_grad(params, name):
    """Data-parallel runs: DistributedDataParallel starts a bucket's all-reduce from the gradient hook of the bucket's LAST parameter and orders it
    behind the stream THAT hook runs on.  Parameters whose gradients are written on a side stream get a post-accumulate hook that joins the two
    streams (each waits for the other's work so far), so whichever hook of a bucket comes last, its stream has seen every gradient of the bucket.
    Registered once per (parameter object, stream name), and only once the process group has more than one rank (called on every forward pass: a
    single-process run never pays for the hooks, a model that was copied or unpickled gets them on its first multi-rank forward)."""
    if not _multi_rank():
        return

    def hook(p):
        s, m = _SIDE.get((p.device.index, name)), _MAIN.get(p.device.index)
        if s is not None and m is not None:
            s.wait_stream(m)
            m.wait_stream(s)
    for p in params:
        ent = _HOOKED.get(id(p))
        if ent is None or ent[0]() is not p:
            ent = _HOOKED[id(p)] = (weakref.ref(p, lambda _r, k=id(p): _HOOKED.pop(k, None)), set())
        if name not in ent[1]:
            p.register_post_accumulate_grad_hook(hook)
            ent[1].add(name)
