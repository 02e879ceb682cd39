"""Which fork changes a bit of the step's FORWARD results?  BASELINE config 2 (8 images), one training-mode forward pass per setting from the same
weights and batch: every side stream off (the reference result), each fork alone, all of them; every setting twice (a race would show as run-to-run
noise, a different-but-deterministic result as the same difference twice).  Prints the five losses as exact hex floats and, for the FPN fork, compares
the pyramid itself.   python tools/fork_bisect.py [--batch 8]
[ref: the forward being forked is distillator.py:39-68]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--config", default="configs/lgd_retinanet_r50.yaml")
    a = ap.parse_args()
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.student import fpn as _fpn
    from lgd_amd.student import retinanet as _rn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device("cuda:0")
    cfg = config.setup_cfg(os.path.join(root, a.config), ["MODEL.DEVICE", "cuda:0"])
    torch.manual_seed(0)
    model = build_model(cfg).train()
    model.distill_flag = 1
    data = synthetic_batch(a.batch, 800, 1333, 10, seed=3, device=dev)

    def setting(teacher, head, adapter, fpn):
        model.teacher.side_stream = teacher
        _rn._HEAD_STREAMS = head
        model.adapter_stream = adapter
        _fpn._FPN_STREAM = fpn

    def forward():
        with torch.no_grad():
            pass
        losses = model(data)
        torch.cuda.synchronize()
        return {k: float(v.detach()) for k, v in losses.items()}

    def pyramid():
        s = model.student
        with torch.no_grad():
            _, feats, _, _ = s.backbone_features(data)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in feats.items()}
    names = {"none": (False, False, False, False), "teacher": (True, False, False, False), "head": (False, True, False, False),
             "adapter": (False, False, True, False), "fpn": (False, False, False, True), "all": (True, True, True, True)}
    ref = None
    for n, st in names.items():
        setting(*st)
        for rep in range(2):
            l = forward()
            if ref is None:
                ref = l
            print("%-8s run %d: " % (n, rep) + "  ".join("%s %s%s" % (k, float(v).hex(), "" if v == ref[k] else " (!= none: %+.1e)" % ((v - ref[k]) / abs(ref[k])))
                                                         for k, v in l.items()), flush=True)
    setting(False, False, False, False)
    p0 = pyramid()
    setting(False, False, False, True)
    p1 = pyramid()
    for k in p0:
        d = (p0[k] - p1[k]).abs()
        print("pyramid level %s: FPN fork vs none: max |diff| %.3e (max |x| %.3e), %d elements differ" % (k, float(d.max()), float(p0[k].abs().max()), int((d > 0).sum())))


if __name__ == "__main__":
    main()
