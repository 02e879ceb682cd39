"""Is ONE training step (forward + backward, no optimizer update) bit-reproducible -- with the step's forks off, with them on, and with a competing
copy stream beside them?  BASELINE config 2, the same weights and batch every time; losses and EVERY parameter gradient are compared bitwise with the
first one-stream run.  Which tensors move first tells where a remaining cross-stream effect sits (round 6: after the packed-fp32 fix the 300-step
stress run still drifted from its one-stream twin from step ~6 on while two one-stream runs stayed bit-identical).
    python tools/step_determinism.py [--runs 6] [--batch 8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--config", default="configs/lgd_retinanet_r50.yaml")
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
    d = cfg.MODEL.DISTILLATOR
    it0 = max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
    tr.set_phase(it0)
    model.train()
    data = synthetic_batch(a.batch, 800, 1333, 10, seed=3, device=dev)
    norm0 = model.student.loss_normalizer.clone() if hasattr(model.student, "loss_normalizer") else None

    def setting(teacher, head, adapter, fpn):
        model.teacher.side_stream, _rn._HEAD_STREAMS, model.adapter_stream, _fpn._FPN_STREAM = teacher, head, adapter, fpn

    def run(competitor=None):
        if norm0 is not None:
            model.student.loss_normalizer = norm0.clone()   # (the EMA advances in every forward pass: same start for every run)
        for p in model.parameters():
            p.grad = None
        if tr._step_folds is not None:
            tr._step_folds.prepare()
        if competitor is not None:
            with torch.cuda.stream(competitor[0]):
                for _ in range(24):
                    competitor[2].copy_(competitor[1], non_blocking=True)
        losses = model(data)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in losses.items()}, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    def report(tag, res, ref):
        dl = [k for k in ref[0] if not torch.equal(res[0][k], ref[0][k])]
        dg = [n for n in ref[1] if not torch.equal(res[1][n], ref[1][n])]
        worst = max([float((res[1][n] - ref[1][n]).abs().max() / (ref[1][n].abs().max() + 1e-30)) for n in dg] + [0.0])
        groups = {}
        for n in dg:
            key = ".".join(n.split(".")[:3])
            groups[key] = groups.get(key, 0) + 1
        print("%-34s losses differing: %-44s gradients differing: %3d of %d (worst %.1e) %s" % (tag, dl or "none", len(dg), len(ref[1]), worst,
                                                                                               dict(sorted(groups.items(), key=lambda kv: -kv[1])[:6])), flush=True)
    setting(False, False, False, False)
    ref = run()
    for i in range(2):
        report("one stream, run %d" % (i + 1), run(), ref)
    for name, st in (("teacher fork only", (True, False, False, False)), ("head fork only", (False, True, False, False)), ("adapter fork only", (False, False, True, False)),
                     ("fpn fork only", (False, False, False, True)), ("all forks", (True, True, True, True))):
        setting(*st)
        for i in range(a.runs if name == "all forks" else 2):
            report("%s, run %d" % (name, i), run(), ref)
    comp = (torch.cuda.Stream(dev), torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev))
    for i in range(a.runs):
        report("all forks + copy stream, run %d" % i, run(comp), ref)
    setting(False, False, False, False)
    for i in range(2):
        report("one stream + copy stream, run %d" % i, run(comp), ref)


if __name__ == "__main__":
    main()
