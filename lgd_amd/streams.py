"""Second HIP streams for chains of the step that do not depend on each other (the teacher's label encoder beside the backbone, the box tower of
the head beside the class tower): one process per GPU still, the streams overlap tails and small launches of one chain with the other's kernels.
Autograd runs every backward node on the stream of its forward, so the overlap carries over to the backward pass.
[ref: the reference issues everything on one stream -- train.py:182-215; the chains themselves: thirdparty_heads/fcos.py:520-546,
 detectron2 RetinaNetHead.forward]"""
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


def join(main, s, outputs=()):
    """the current stream waits for the side stream; `outputs` (made on the side stream, read on the current one from here on) are recorded on it"""
    main.wait_stream(s)
    for t in outputs:
        t.record_stream(main)


def join_on_grad(params, name):
    """Data-parallel runs: DistributedDataParallel starts a bucket's all-reduce from the gradient hook of the bucket's LAST parameter and orders it
    behind the stream THAT hook runs on.  Parameters whose gradients are written on a side stream get a post-accumulate hook that joins the two
    streams (each waits for the other's work so far) when the process group has more than one rank, so whichever hook of a bucket comes last,
    its stream has seen every gradient of the bucket.  Registered once per parameter and stream name."""
    def hook(p):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        s, m = _SIDE.get((p.device.index, name)), _MAIN.get(p.device.index)
        if s is not None and m is not None:
            s.wait_stream(m)
            m.wait_stream(s)
    for p in params:
        done = getattr(p, "_lgd_join_streams", None)
        if done is None:
            done = p._lgd_join_streams = set()
        if name not in done:
            p.register_post_accumulate_grad_hook(hook)
            done.add(name)
