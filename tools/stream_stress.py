"""Multi-stream stress run of the training step: N optimizer steps of the SHIPPED path with every fork of the step on
(label encoder, box tower, adapter, FPN small levels on side streams: lgd_amd/streams.py) WHILE a further stream keeps the memory system busy with
large device-to-device copies -- a stand-in for the RCCL kernels that run beside the backward in a data-parallel job -- and then the same N steps
from the same initial weights on ONE stream with nothing beside them -- twice: two identical one-stream runs drift apart (the vendor library's
small-level kernels are not bit-reproducible and 300 SGD steps amplify the last bit: measured 1e-7 after 5 steps, 5e-4 after 50, tens of per cent after
150), so a deviation of the forked run is judged against THAT.  Prints one JSON line: per-step losses of the three runs, ms/step, and how many competitor
copies were issued; the drift table goes to stderr.  tests/test_model_gpu.py::test_step_forks_under_a_competing_stream runs it as a subprocess with a deadline (a stall
is a failed test, not a hung suite); stand-alone:
    python tools/stream_stress.py --steps 300 [--config configs/lgd_retinanet_r50.yaml --batch 8]
[ref: the step being protected is train.py:182-215; the reference issues everything on one stream]"""
import argparse
import copy
import faulthandler
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--config", default="configs/lgd_retinanet_r50.yaml")
    ap.add_argument("--batch", type=int, default=8, help="images per step: at 8 (BASELINE config 2) every fork passes its per-call gate")
    ap.add_argument("--copies-per-step", type=int, default=24, help="competitor launches (256 MB device-to-device copies) issued per step")
    ap.add_argument("--deadline", type=float, default=900.0, help="seconds without a finished step after which all host stacks are dumped and the process exits 3")
    a = ap.parse_args()
    from lgd_amd import config, ops, streams
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    from lgd_amd.student import fpn as _fpn
    from lgd_amd.student import retinanet as _rn
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device("cuda:0")
    cfg = config.setup_cfg(os.path.join(root, a.config), ["MODEL.DEVICE", "cuda:0"])
    torch.manual_seed(0)
    base = build_model(cfg)
    twin = copy.deepcopy(base)
    data = [synthetic_batch(a.batch, 800, 1333, 10, seed=3 + j, device=dev) for j in range(2)]
    d = cfg.MODEL.DISTILLATOR
    it0 = max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
    _rn._HEAD_STREAMS = _fpn._FPN_STREAM = True   # (the FPN fork ships off since round 6 -- student/fpn.py -- the stress run keeps it on: every fork the code has)
    assert ops.side_streams_ok() and base.teacher.side_stream and base.adapter_stream
    forks = []
    real_fork = streams.fork
    streams.fork = lambda dv, name, inputs=(): (forks.append(name), real_fork(dv, name, inputs))[1]

    def run(model, competitor):
        tr = Trainer(cfg, model, distributed=False)
        src = torch.empty(64 << 20, dtype=torch.float32, device=dev).normal_() if competitor else None   # 256 MB
        dst = torch.empty_like(src) if competitor else None
        side = torch.cuda.Stream(dev) if competitor else None
        fences, issued, traj = [], 0, []
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(a.steps):
            faulthandler.dump_traceback_later(a.deadline, exit=True)      # (a C-level watchdog thread: fires also when the host sits inside the runtime)
            if competitor:
                if len(fences) >= 2:
                    fences.pop(0).synchronize()                           # at most two steps' worth of copies queued
                with torch.cuda.stream(side):
                    for _ in range(a.copies_per_step):
                        dst.copy_(src, non_blocking=True)
                    issued += a.copies_per_step
                    ev = torch.cuda.Event()
                    ev.record()
                    fences.append(ev)
            losses = tr.step(data[i % 2], it0 + i)
            traj.append({k: float(v) for k, v in losses.items()})          # (one host sync per step: progress is observed, not assumed)
            faulthandler.cancel_dump_traceback_later()
        torch.cuda.synchronize()
        return traj, 1e3 * (time.time() - t0) / a.steps, issued
    ta, ms_a, issued = run(base, True)
    nf = len(forks)
    base.teacher.side_stream = twin.teacher.side_stream = False
    _rn._HEAD_STREAMS = False
    twin.adapter_stream = False
    _fpn._FPN_STREAM = False
    twin2 = copy.deepcopy(twin)
    tb, ms_b, _ = run(twin, False)
    tc, _, _ = run(twin2, False)          # the SAME one-stream run once more: the run-to-run drift of this step, what a deviation is measured against
    assert len(forks) == nf, "a one-stream twin forked"

    def dev(x, y):
        return [max(abs(p[k] - q[k]) / max(abs(q[k]), 1e-6) for k in p) for p, q in zip(x, y)]
    dab, dbc = dev(ta, tb), dev(tb, tc)
    marks = [i for i in (1, 2, 5, 10, 20, 50, 100, 150, 200, 250, 300) if i <= a.steps]
    sys.stderr.write("step:                       " + " ".join("%8d" % i for i in marks) + "\n")
    sys.stderr.write("forked+load vs one stream:  " + " ".join("%8.1e" % max(dab[:i]) for i in marks) + "   (worst loss deviation up to that step)\n")
    sys.stderr.write("one stream vs one stream:   " + " ".join("%8.1e" % max(dbc[:i]) for i in marks) + "\n")
    print(json.dumps({"steps": a.steps, "forks_per_step": {n: forks.count(n) / a.steps for n in sorted(set(forks))}, "competitor_copies": issued,
                      "ms_per_step_forked_under_load": ms_a, "ms_per_step_one_stream": ms_b, "forked": ta, "one_stream": tb, "one_stream_again": tc}))


if __name__ == "__main__":
    main()
