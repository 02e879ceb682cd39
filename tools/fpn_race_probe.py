"""Round 6 diagnosis: tools/fork_bisect.py found pyramid level p4 different in 240 elements with the FPN fork on.  Which elements, is it run-to-run
noise or deterministic, and does it follow the fork (a sync in front of it, the fork off, the ordering of library calls off)?
    python tools/fpn_race_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from lgd_amd import config, ops, streams
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.student import fpn as _fpn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device("cuda:0")
    cfg = config.setup_cfg(os.path.join(root, "configs/lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cuda:0"])
    torch.manual_seed(0)
    model = build_model(cfg).train()
    data = synthetic_batch(8, 800, 1333, 10, seed=3, device=dev)
    s = model.student
    model.teacher.side_stream = False

    def pyramid():
        with torch.no_grad():
            _, feats, _, _ = s.backbone_features(data)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in feats.items()}

    def report(tag, p, ref):
        out = []
        for k in ref:
            d = (p[k] - ref[k]).abs()
            n = int((d > 0).sum())
            out.append("%s:%d" % (k, n) + ("" if n == 0 else "(max %.2e)" % float(d.max())))
            if n and k == "p4":
                idx = torch.nonzero(d > 0)
                print("    p4 differing elements: images %s channels %s rows %s cols %s" % (sorted(set(idx[:, 0].tolist())), sorted(set(idx[:, 1].tolist()))[:12],
                                                                                             sorted(set(idx[:, 2].tolist())), sorted(set(idx[:, 3].tolist()))[:20]))
        print("%-44s %s" % (tag, "  ".join(out)), flush=True)
    _fpn._FPN_STREAM = False
    ref = pyramid()
    for i in range(2):
        report("fork off, run %d" % i, pyramid(), ref)
    _fpn._FPN_STREAM = True
    for i in range(3):
        report("fork on, run %d" % i, pyramid(), ref)
    real_fork = streams.fork

    def synced_fork(dv, name, inputs=()):
        torch.cuda.synchronize()
        return real_fork(dv, name, inputs)
    streams.fork = synced_fork
    for i in range(2):
        report("fork on, device sync in front of it, run %d" % i, pyramid(), ref)
    streams.fork = real_fork
    real_join = streams.join

    def early_join(main, side, outputs=()):
        return real_join(main, side, outputs)
    prev = ops.conv3x3_backend(min_tiles=2000)     # the small levels' output convolutions on the library's direct kernels (p4: 1008 tiles)
    ref2 = None
    _fpn._FPN_STREAM = False
    ref2 = pyramid()
    _fpn._FPN_STREAM = True
    for i in range(2):
        report("fork on, small levels on MIOpen, run %d (vs same, fork off)" % i, pyramid(), ref2)
    ops.conv3x3_backend(*prev)
    prev = ops.gemm3_backend(False)                # ... and with the Winograd products of the small levels on the library's GEMMs instead of gemm3
    _fpn._FPN_STREAM = False
    ref3 = pyramid()
    _fpn._FPN_STREAM = True
    for i in range(2):
        report("fork on, gemm3 off (library GEMMs), run %d (vs same, fork off)" % i, pyramid(), ref3)
    ops.gemm3_backend(*prev)


if __name__ == "__main__":
    main()
