"""ms/step of the training step with each SUBSET of its forks on (label encoder, class tower, adapter, FPN output convolution: lgd_amd/streams.py), one
process, same weights and batches -- run under different GPU_MAX_HW_QUEUES to see which fork gains or loses from a hardware queue of its own.
    GPU_MAX_HW_QUEUES=8 python tools/fork_subsets.py [--config configs/lgd_retinanet_r50.yaml --batch 8 --steps 20]
[ref: the step is train.py:182-215; the reference issues everything on one stream]"""
import argparse
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

os.environ.setdefault("LGD_SIDE_STREAMS", "force")   # (the probe exists to measure the forks under ANY queue count: ops.side_streams_ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="configs/lgd_retinanet_r50.yaml")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--all-subsets", action="store_true", help="all subsets instead of none / each alone / all / all but one")
    ap.add_argument("--sets", default="", help="explicit subsets, e.g. 'none;teacher+head+adapter;teacher+head+adapter+filters'")
    a = ap.parse_args()
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    from lgd_amd.student import fpn as _fpn
    from lgd_amd.student import retinanet as _rn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device("cuda:0")
    cfg = config.setup_cfg(os.path.join(root, a.config), ["MODEL.DEVICE", "cuda:0"])
    torch.manual_seed(0)
    model = build_model(cfg)
    tr = Trainer(cfg, model, distributed=False)
    data = [synthetic_batch(a.batch, 800, 1333, 10, seed=3 + j, device=dev) for j in range(2)]
    d = cfg.MODEL.DISTILLATOR
    it0 = max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
    names = ("teacher", "head", "adapter", "fpn")

    def setting(on):
        model.teacher.side_stream = "teacher" in on
        _rn._HEAD_STREAMS = "head" in on
        model.adapter_stream = "adapter" in on
        _fpn._FPN_STREAM = "fpn" in on

    def run(n):
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(n):
            tr.step(data[i % 2], it0 + i)
        torch.cuda.synchronize()
        return 1e3 * (time.time() - t0) / n
    if a.sets:
        subsets = [tuple(n for n in names if n in spec.split("+")) for spec in a.sets.split(";")]
    elif a.all_subsets:
        subsets = [c for r in range(len(names) + 1) for c in itertools.combinations(names, r)]
    else:
        subsets = [()] + [(n,) for n in names] + [tuple(m for m in names if m != n) for n in names] + [names]
    setting(names)
    run(5)
    print("GPU_MAX_HW_QUEUES=%s  %s  batch %d" % (os.environ.get("GPU_MAX_HW_QUEUES", "(default 4)"), a.config, a.batch), flush=True)
    for rep in range(2):
        for s in subsets:
            setting(s)
            run(2)
            print("forks on: %-32s %7.2f ms/step" % ("+".join(s) or "none", run(a.steps)), flush=True)


if __name__ == "__main__":
    main()
