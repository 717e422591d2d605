#!/usr/bin/env python3
"""Random sequential networks through the C++ Layer classes (architectures::Sequential::train_step, cnn_amd/host) in three settings --
default (fuse_layers + fuse_pool_block), fuse_pool_block off, fuse_layers off -- compared with each other BIT FOR BIT (losses, parameters,
gradients, the delta with respect to the input, every layer's get_output() after the last step: the fused passes do not write some of
those tensors and re-compute them on demand) and with the CPU oracle (oracle.pyoracle.SeqNet: logits and loss of the first step at 1e-4).
The fusion wiring depends on neighbours and shapes (Conv2D -> ReLU, BatchNorm2D -> ReLU -> MaxPool2D(2,2) with the pool inside the apply
pass and its backward from the pooled domain, ReLU' in the data gradients / the pool's backward): random layer lists reach the
combinations the BASELINE stacks do not.     usage: fuzz_nets.py [nets=12] [seed=1] [big] [dp]
big: inputs of 96 .. 160 pixels whose first block is Conv2D -> BatchNorm2D -> ReLU -> MaxPool2D(2,2) -- planes large enough for the
general (not channel-resident) BatchNorm2D kernels, where the pool runs inside the apply pass and the backward pass starts from the pooled domain
dp:  also with a one-rank RCCL communicator and the gradient exchange forced on (plain and bucketed): bit-identical to the plain step"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from cnn_amd import hostapi
from cnn_amd.stacks import he_init
from oracle import pyoracle as O

big = "big" in sys.argv[1:]
dp = "dp" in sys.argv[1:]
sys.argv = [a for a in sys.argv if a not in ("big", "dp")]
n_nets = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lib = hostapi.load()


bad = 0
for it in range(n_nets):
    # (shape bookkeeping is done by the oracle's own walk: build the spec first, then ask SeqNet for the layout)
    S = int(rs.randint(16, 49))
    Wd = S if rs.rand() < 0.6 else int(rs.randint(16, 49))
    if big:
        S, Wd = int(rs.randint(96, 161)), int(rs.randint(96, 161))
    in_shape = (3, S, Wd)
    H, W = S, Wd
    spec = []
    if big:
        k = int(rs.choice([3, 3, 5]))
        pad = int(rs.randint(0, k // 2 + 1))
        spec += [("conv", int(rs.choice([8, 16, 24])), k, 1, pad), ("bn",), ("relu",), ("pool", 2, 2)]
        H, W = (H + 2 * pad - k + 1 - 2) // 2 + 1, (W + 2 * pad - k + 1 - 2) // 2 + 1
    for blk in range(int(rs.randint(1 if big else 2, 4 if big else 5))):
        k = int(rs.choice([1, 3, 3, 3, 5]))
        s = int(rs.choice([1, 1, 2]))
        pad = int(rs.randint(0, k // 2 + 1))
        if (H + 2 * pad - k) // s + 1 < 2 or (W + 2 * pad - k) // s + 1 < 2:
            k, s, pad = 3, 1, 1
        spec.append(("conv", int(rs.choice([8, 16, 24, 32, 40])), k, s, pad))
        H, W = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        if rs.rand() < 0.5:
            spec.append(("bn",))
        spec.append(("relu",))
        if rs.rand() < 0.45 and H >= 4 and W >= 4:
            pk, ps = (2, 2) if rs.rand() < 0.8 else (3, 2)
            spec.append(("pool", pk, ps))
            H, W = (H - pk) // ps + 1, (W - pk) // ps + 1
        if rs.rand() < 0.2:
            spec.append(("dropout", float(rs.choice([0.2, 0.5]))))  # (dropout.cpp: every layer's own engine, seed 1314)
    spec.append(("linear", 3))
    B = int(rs.randint(2, 5))
    onet = O.SeqNet(spec, in_shape)
    # the bias gradient of a convolution that feeds a BatchNorm2D is exactly 0 (the batch mean is subtracted, batchnorm2d.cpp:46-61): what any
    # fp32 implementation holds there is rounding noise of the sum of its deltas -- kept out of the relative comparisons (tests/util.py
    # assert_noise_of_exact_zero bounds it in the suite)
    live = np.ones(onet.n_params, bool)
    for i, e in enumerate(onet.layers[:-1]):
        if e["kind"] == "conv" and onet.layers[i + 1]["kind"] == "bn":
            co = e["out"][0]
            live[e["off"] + e["n"] - co : e["off"] + e["n"]] = False
    p0 = he_init(onet.layers, 500 + it)
    onet.params[:] = p0
    x = rs.rand(B, *in_shape).astype(np.float32)
    labels = (np.arange(B) % 3).astype(np.int32)
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    names = hostapi._layer_names(spec)
    runs = {}
    arbitrated = []
    try:
        for mode, (fl, fp) in (("default", (1, 1)), ("no pool block", (1, 0)), ("no fusion", (0, 0))):
            lib.cnnh_set_fuse_layers(fl)
            lib.cnnh_set_fuse_pool_block(fp)
            net = hostapi.HostSequential(spec, in_shape)
            net.set_params(p0)
            trace = []
            for step in range(3):
                net.train_step(xd, ld, 1e-3)
                trace.append((net.last_loss(), net.get_params(), net.get_grads(), net.input_delta((B,) + in_shape)))
            # a SMALLER batch (the fused passes are sized for the first call's batch: the layers fall back), then the full one again
            net.train_step(xd[: B - 1], ld[: B - 1], 1e-3)
            trace.append((net.last_loss(), net.get_params(), net.get_grads(), net.input_delta((B,) + in_shape)))
            net.train_step(xd, ld, 1e-3)
            trace.append((net.last_loss(), net.get_params(), net.get_grads(), net.input_delta((B,) + in_shape)))
            outs = {names[i]: net.layer_output(names[i], (B,) + tuple(e["out"])) for i, e in enumerate(onet.layers)}
            if mode == "default":  # first step against the oracle
                net2 = hostapi.HostSequential(spec, in_shape)
                net2.set_params(p0)
                net2.train_step(xd, ld, 1e-3)
                logits = net2.layer_output("linear_1", (B, 3))
                ologits = onet.forward(x)
                oloss, _ = O.cross_entropy_backward(O.softmax(ologits), labels)
                e_log = float(np.abs(logits.reshape(ologits.shape) - ologits).max() / max(np.abs(ologits).max(), 1e-30))
                e_loss = abs(net2.last_loss() - oloss) / max(1.0, abs(oloss))
                net2.close()
                # ... and the partial-batch step (trace[3]) after three full ones: loss and gradients (whole-arena norm; ReLU / pool decisions
                # within rounding distance may flip over four steps: beyond 1e-4 / 1e-3 the fp64 restatement arbitrates, below)
                onet2 = O.SeqNet(spec, in_shape)  # (a fresh one: every Dropout layer's engine starts at its seed, like the net's)
                onet2.params[:] = p0
                for _ in range(3):
                    onet2.train_step(x, labels, 1e-3)
                pl, _ = onet2.train_step(x[: B - 1], labels[: B - 1], 1e-3)
                e_ploss = abs(trace[3][0] - pl) / max(1.0, abs(pl)) if np.isfinite(pl) else 0.0
                e_pgrad = float(np.abs(trace[3][2] - onet2.grads)[live].max() / max(np.abs(onet2.grads).max(), 1e-30)) if np.isfinite(pl) else 0.0
                if e_ploss > 1e-4 or e_pgrad > 1e-3:
                    # several steps deep the fp32 ORACLE itself has drifted from the exact sequence (its sequential sums, flipped ReLU / pool
                    # decisions: SURVEY.md H3): arbitrate with the fp64 restatement like tests/util.py -- the HIP path may be as far from
                    # it as the fp32 oracle is (x 3), no further
                    o64 = O.SeqNet(spec, in_shape, f64=True)
                    o64.params[:] = p0
                    for _ in range(3):
                        o64.train_step(x, labels, 1e-3)
                    pl64, _ = o64.train_step(x[: B - 1], labels[: B - 1], 1e-3)
                    gmax = max(np.abs(o64.grads).max(), 1e-30)
                    hip_g, ora_g = np.abs(trace[3][2] - o64.grads)[live].max() / gmax, np.abs(onet2.grads - o64.grads)[live].max() / gmax
                    # per layer: as far from fp64 as the fp32 oracle (x 3) -- and for the WEIGHTS of a convolution that feeds a BatchNorm2D an absolute
                    # 1e-1 of the layer's largest gradient (one- or two-sample batches on 100+ pixel planes reach 3e-2): they are sum(dx * x) over a dx whose sum is exactly 0 and an x that is not centred, i.e. the
                    # exact-zero noise above times mean(x); measured on isolated BatchNorm2D backward passes (126 x 126 planes): that sum is off by
                    # 1e-2 .. 3e-1 of its value on the HIP path and by 6e-2 .. 17 (!) in the fp32 oracle -- whichever is closer to fp64 is chance
                    layers_ok = True
                    for i_, e_ in enumerate(onet2.layers):
                        if e_.get("n", 0) > 0:
                            sl = slice(e_["off"], e_["off"] + e_["n"])
                            lv = live[sl]
                            m_ = max(np.abs(o64.grads[sl]).max(), 1e-300)
                            h_, o_ = np.abs(trace[3][2][sl] - o64.grads[sl])[lv].max() / m_, np.abs(onet2.grads[sl] - o64.grads[sl])[lv].max() / m_
                            feeds_bn = e_["kind"] == "conv" and i_ + 1 < len(onet2.layers) and onet2.layers[i_ + 1]["kind"] == "bn"
                            layers_ok = layers_ok and h_ <= max(3 * max(o_, 1e-4), 1e-1 if feeds_bn else 0.0)
                    if layers_ok:
                        hip_g = 0.0
                    hip_l, ora_l = abs(trace[3][0] - pl64) / max(1.0, abs(pl64)), abs(pl - pl64) / max(1.0, abs(pl64))
                    # (a saturated step -- loss ~ 1e-8, the true class' probability one ulp from 1: delta = p - y is rounding noise of expf's last
                    #  bit, and so is every gradient behind it; a relative error of noise says nothing)
                    degenerate = gmax < 1e-6 or abs(pl64) < 1e-3  # (|delta| ~ the loss: below 1e-3 one ulp of p is > 6e-5 of delta)
                    if (degenerate or hip_g <= 3 * max(ora_g, 1e-4)) and hip_l <= 3 * max(ora_l, 1e-5):
                        arbitrated.append(f"partial-batch step: HIP {hip_g:.1e} / fp32 oracle {ora_g:.1e} from fp64")
                        e_ploss, e_pgrad = 0.0, 0.0
                    else:
                        per = []
                        for e in onet2.layers:
                            if e.get("n", 0) > 0:
                                sl = slice(e["off"], e["off"] + e["n"])
                                m = max(np.abs(o64.grads[sl]).max(), 1e-300)
                                per.append(f"{e['kind']} max|g| {m:.1e} HIP {np.abs(trace[3][2][sl] - o64.grads[sl]).max() / m:.1e} ora {np.abs(onet2.grads[sl] - o64.grads[sl]).max() / m:.1e}")
                        if os.environ.get("FUZZ_DIAG"):  # the same sequence with other weight-gradient kernels: is it one kernel family?
                            from cnn_amd import capi as capi_

                            for opt in os.environ["FUZZ_DIAG"].split(","):
                                capi_.set_option(opt, "0")
                                n3 = hostapi.HostSequential(spec, in_shape)
                                n3.set_params(p0)
                                for _ in range(3):
                                    n3.train_step(xd, ld, 1e-3)
                                n3.train_step(xd[: B - 1], ld[: B - 1], 1e-3)
                                g3 = n3.get_grads()
                                n3.close()
                                capi_.set_option(opt, None)
                                e0 = onet2.layers[0]
                                sl0 = slice(e0["off"], e0["off"] + e0["n"])
                                per.append(f"[{opt}=0: first layer HIP {np.abs(g3[sl0] - o64.grads[sl0])[live[sl0]].max() / max(np.abs(o64.grads[sl0]).max(), 1e-300):.1e}]")
                        arbitrated.append(f"partial-batch step NOT within 3 x the oracle's own distance: gradients HIP {hip_g:.1e} / fp32 oracle {ora_g:.1e}, "
                                          f"loss HIP {hip_l:.1e} / oracle {ora_l:.1e} from fp64, B - 1 = {B - 1}; per layer: " + "; ".join(per))
            runs[mode] = (trace, outs)
            net.close()
        # inference (architectures::no_grad: BatchNorm2D on its moving statistics, Dropout off, nothing recorded; inference.cpp's use of the
        # classes) after two training steps: logits against the oracle in evaluation mode
        net = hostapi.HostSequential(spec, in_shape)
        net.set_params(p0)
        onet3 = O.SeqNet(spec, in_shape)
        onet3.params[:] = p0
        for _ in range(2):
            net.train_step(xd, ld, 1e-3)
            onet3.train_step(x, labels, 1e-3)
        lib.cnnh_set_no_grad(1)
        try:
            ev = net.forward_host(x)
        finally:
            lib.cnnh_set_no_grad(0)
        oev = onet3.forward(x, training=False)
        e_eval = float(np.abs(ev - oev.reshape(ev.shape)).max() / max(np.abs(oev).max(), 1e-30)) if np.all(np.isfinite(oev)) else 0.0
        net.close()
        if e_eval > 1e-4:
            o64 = O.SeqNet(spec, in_shape, f64=True)
            o64.params[:] = p0
            for _ in range(2):
                o64.train_step(x, labels, 1e-3)
            e64 = o64.forward(x, training=False)
            lmax = max(np.abs(e64).max(), 1e-30)
            hip_e, ora_e = np.abs(ev - e64.reshape(ev.shape)).max() / lmax, np.abs(oev - e64).max() / lmax
            if hip_e <= 3 * max(ora_e, 1e-5):
                arbitrated.append(f"inference: HIP {hip_e:.1e} / fp32 oracle {ora_e:.1e} from fp64")
                e_eval = 0.0
        # data parallelism with ONE rank (every collective the N-rank step issues is issued, every sum is an identity): a one-rank RCCL
        # communicator with the exchange forced on, plain and bucketed (sync-BN reductions on the same communicator), against the default run
        dp_diff = []
        if dp:
            from cnn_amd import capi
            from cnn_amd.dp import RcclComm

            comm = RcclComm(None, 1, 0)
            for opt in ("DP_FORCE_EXCHANGE", "DP_FORCE_BUCKETS"):
                capi.set_option(opt, "1")
                try:
                    net = hostapi.HostSequential(spec, in_shape)
                    net.set_params(p0)
                    net.set_comm(comm.handle, 1)
                    for step in range(3):
                        net.train_step(xd, ld, 1e-3)
                        la, pa, ga, _ = runs["default"][0][step]
                        for what, a, b in (("loss", np.float32(net.last_loss()), np.float32(la)), ("params", net.get_params(), pa), ("grads", net.get_grads(), ga)):
                            if not np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32)):
                                dp_diff.append(f"{opt}: step {step} {what}")
                    net.close()
                finally:
                    capi.set_option(opt, None)
            comm.destroy()
        # checkpoint round trip (alexnet.cpp:67-90: every layer's parameters in list order) and Grad-CAM (alexnet.cpp:95-142) on a random layer
        import tempfile

        net = hostapi.HostSequential(spec, in_shape)
        net.set_params(p0)
        net.train_step(xd, ld, 1e-3)
        net.train_step(xd, ld, 1e-3)
        misc_diff = []
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "net.model")
            net.save_checkpoint(path)
            twin = hostapi.HostSequential(spec, in_shape)
            twin.load_checkpoint(path)
            if not np.array_equal(twin.get_params().view(np.uint32), net.get_params().view(np.uint32)):
                misc_diff.append("checkpoint: parameters differ after save -> load")
            la, lb = net.forward_host(x), twin.forward_host(x)
            if not np.array_equal(la.view(np.uint32), lb.view(np.uint32)):
                misc_diff.append("checkpoint: logits of the loaded net differ")
            twin.close()
        cam_layers = [i for i, e in enumerate(onet.layers) if e["kind"] in ("conv", "relu", "pool")]
        ci = cam_layers[int(rs.randint(0, len(cam_layers)))]
        Cq, Hq, Wq = onet.layers[ci]["out"]
        net.forward_host(x)  # (a forward pass with gradients enabled precedes grad_cam, alexnet.cpp:95)
        fea = net.layer_output(names[ci], (B, Cq, Hq, Wq))
        img, cam = net.grad_cam(names[ci], (B, Hq, Wq))
        cam_ref, img_ref = O.grad_cam(fea)
        if not (np.array_equal(cam.view(np.uint32), cam_ref.view(np.uint32)) and np.array_equal(img, img_ref)):
            misc_diff.append(f"grad_cam({names[ci]}) differs from the oracle's on the same feature map")
        net.close()
        # the reference's own loop through the same classes (cnn.cpp:79-90: forward -> host softmax / cross_entroy_backward -> backward ->
        # update_gradients), from host tensors and from a device batch: parameters after every step against train_step's (host expf vs
        # device expf may differ in the last bit of the probabilities: 2e-5 of the arena's largest parameter)
        e_loop = 0.0
        for how in ("device batch", "host tensors"):
            net = hostapi.HostSequential(spec, in_shape)
            net.set_params(p0)
            for step in range(5):
                xs, ls = (x[: B - 1], labels[: B - 1]) if step == 3 else (x, labels)
                if how == "device batch":
                    net.train_step_device(torch.from_numpy(xs).cuda(), ls, 1e-3)
                else:
                    net.train_step_host(xs, ls, 1e-3)
                pr = runs["default"][0][step][1]
                if np.all(np.isfinite(pr)):
                    e_loop = max(e_loop, float(np.abs(net.get_params() - pr).max() / max(np.abs(pr).max(), 1e-30)))
            net.close()
    finally:
        lib.cnnh_set_fuse_layers(1)
        lib.cnnh_set_fuse_pool_block(1)
    diffs = []
    ref_trace, ref_outs = runs["no fusion"]
    for mode in ("default", "no pool block"):
        trace, outs = runs[mode]
        for step, ((la, pa, ga, da), (lb, pb, gb, db)) in enumerate(zip(trace, ref_trace)):
            for what, a, b in (("loss", np.float32(la), np.float32(lb)), ("params", pa, pb), ("grads", ga, gb), ("input delta", da, db)):
                if not np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32)):  # (bit patterns: a diverged step's NaN loss equals itself)
                    diffs.append(f"{mode}: step {step} {what} (max |d| {np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max():.2e})")
        for nm in outs:
            if not np.array_equal(outs[nm].view(np.uint32), ref_outs[nm].view(np.uint32)):
                diffs.append(f"{mode}: get_output({nm})")
    diffs += dp_diff + misc_diff
    ok = not diffs and e_log <= 1e-4 and e_loss <= 1e-4 and e_ploss <= 1e-4 and e_pgrad <= 1e-3 and e_loop <= 2e-5 and e_eval <= 1e-4
    bad += not ok
    print(f"net {it}: B{B} {in_shape} {spec}\n   vs oracle: logits {e_log:.2e} loss {e_loss:.2e}, partial-batch step loss {e_ploss:.2e} grads {e_pgrad:.2e}; reference loop vs train_step params {e_loop:.2e}; inference logits {e_eval:.2e}; fused vs unfused: {'bit-identical' if not diffs else diffs[:6]}{('; fp64-arbitrated: ' + '; '.join(arbitrated)) if arbitrated else ''}{'' if ok else '   <-- FAIL'}")
print(f"FUZZ NETS {'OK' if bad == 0 else 'FAILED'}: {n_nets} networks, {bad} with differences")
sys.exit(1 if bad else 0)
