#!/usr/bin/env python3
"""Random-shape sweep of the other loop nests of the path against the CPU oracle (companion of tests/sweeps/fuzz_conv.py):
  conv  -- 3x3 layers with WIDE planes (up to 230 columns: every width class of conv_rows_any, 28-column blocks of wgrad_sp_any) and the other
           filter sizes / strides the reference accepts (1x1, 5x5, 7x7; conv2d.cpp:41-42)
  pool  -- MaxPool2D forward (values + mask bit-exact, pool2d.cpp:40-83) and backward (+ the fused ReLU'), k in {2, 3}, step in {1, 2, 3}
  bn    -- BatchNorm2D training forward / backward, the fused BatchNorm -> ReLU -> MaxPool(2,2) forward and its pooled-domain backward
  lin   -- LinearLayer forward / backward and the fused loss head with dx
  misc  -- ReLU forward / backward and the SGD step at random (also unaligned) lengths, softmax + cross entropy, Dropout, Grad-CAM: bit-exact
           against the oracle where the arithmetic is the reference's own order; the uint8 batch stager (byte * 1.f / 255, planar) at random sizes
usage: fuzz_layers.py [conv|pool|bn|lin|misc|all] [cases=40] [seed=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from cnn_amd import capi
from oracle import pyoracle as O

what = sys.argv[1] if len(sys.argv) > 1 else "all"
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rs = np.random.RandomState(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
TOL = 1e-4
bad = 0
worst = 0.0


def rel(a, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


MIS = int(os.environ.get("FUZZ_MISALIGN", "0"))  # every input tensor MIS elements off its allocation's start (plain pointers are all the C ABI asks for)
GUARD, CANARY = 2048, -12345.5
_guards = []


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if MIS == 0 or t.numel() == 0:
        return t
    big = torch.empty((t.numel() + MIS,), dtype=t.dtype, device="cuda")
    big[MIS:] = t.reshape(-1)
    return big[MIS:].view(*t.shape)


def guarded(shape):
    """an fp32 output between two guard zones of a larger allocation: a store outside the tensor changes a canary"""
    n = int(np.prod(shape))
    big = torch.full((n + 2 * GUARD + MIS,), CANARY, dtype=torch.float32, device="cuda")
    _guards.append((big[MIS:], n))
    return big[GUARD + MIS:GUARD + MIS + n].view(*shape)


def overruns():
    torch.cuda.synchronize()
    bad_ = sum(int((b[:GUARD] != CANARY).sum().item()) + int((b[GUARD + n:] != CANARY).sum().item()) for b, n in _guards)
    _guards.clear()
    return {"STORES OUTSIDE AN OUTPUT TENSOR": float(bad_)} if bad_ else {}


def report(tag, errs, names=""):
    global bad, worst
    e = max(errs.values())
    worst = max(worst, e)
    fail = {k: f"{v:.2e}" for k, v in errs.items() if v > TOL}
    bad += bool(fail)
    print(f"{tag:46s} {e:.2e}  {names}{'   <-- FAIL ' + str(fail) if fail else ''}")


def conv_cases():
    for it in range(n_cases):
        mode = it % 3
        if mode == 0:  # wide 3x3 planes
            k, s, pad = 3, 1, int(rs.randint(0, 2))
            B, H, W = int(rs.randint(1, 3)), int(rs.randint(3, 9)), int(rs.randint(60, 231))
            Ci, Co = int(rs.choice([8, 16, 32, 40, 64])), int(rs.choice([16, 32, 48, 64]))
        elif mode == 1:  # other filter sizes
            k = int(rs.choice([1, 5, 7]))
            s, pad = int(rs.randint(1, 3)), int(rs.randint(0, k // 2 + 1))
            B, H, W = int(rs.randint(1, 4)), int(rs.randint(k + 1, 40)), int(rs.randint(k + 1, 40))
            Ci, Co = int(rs.choice([3, 8, 16, 32, 64])), int(rs.choice([8, 16, 32, 64, 72]))
        else:  # batch-heavy small planes (many units per workgroup)
            k, s, pad = 3, int(rs.choice([1, 1, 2])), int(rs.randint(0, 2))
            B, H = int(rs.randint(8, 40)), int(rs.randint(5, 20))
            W = H if rs.rand() < 0.5 else int(rs.randint(5, 20))
            Ci, Co = int(rs.choice([16, 32, 64, 128])), int(rs.choice([16, 32, 64, 128]))
        case = (B, Ci, H, W, Co, k, s, pad)
        x = rs.rand(B, Ci, H, W).astype(np.float32)
        w = (rs.standard_normal((Co, Ci, k, k)) * 0.1).astype(np.float32)
        b = (rs.standard_normal(Co) * 0.1).astype(np.float32)
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        dy = (rs.rand(B, Co, Ho, Wo) * 2 - 1).astype(np.float32)
        xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
        y_ref = O.conv2d_forward(xp, w, b, s)
        gw_ref, gb_ref, dxp = O.conv2d_backward(xp, dy, w, s)
        dx_ref = dxp[:, :, pad:pad + H, pad:pad + W] if pad else dxp
        conv = capi.Conv2d(*case)
        xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
        capi.kernel_timing(1)
        y = conv.forward(xd, wd, bd)
        dx = conv.backward_data(dyd, wd)
        gw, gb = conv.backward_weight(xd, dyd, float(B))
        gw2, gb2, dx2 = conv.backward(xd, dyd, wd, float(B))  # both gradients in one call (side stream)
        torch.cuda.synchronize()
        names = " ".join(sorted({k_.split("|")[0] for k_ in capi.kernel_timing_report() if not any(t in k_ for t in ("prep", "reduce", "pack"))}))
        capi.kernel_timing(0)
        report(str(case), {"y": rel(y.cpu().numpy(), y_ref), "dx": rel(dx.cpu().numpy(), dx_ref), "gw": rel(gw.cpu().numpy(), gw_ref),
                           "gb": rel(gb.cpu().numpy(), gb_ref), "dx(both)": rel(dx2.cpu().numpy(), dx_ref), "gw(both)": rel(gw2.cpu().numpy(), gw_ref),
                           "gb(both)": rel(gb2.cpu().numpy(), gb_ref)}, names)


def pool_cases():
    for it in range(n_cases):
        k = int(rs.choice([2, 2, 3]))
        step = int(rs.choice([2, 2, 1, 3])) if k == 3 else int(rs.choice([2, 2, 1]))
        if it % 4 == 3:  # any window / step (pool2d.cpp:14-15 accepts them)
            k, step = int(rs.randint(1, 6)), int(rs.randint(1, 5))
        B, C = int(rs.randint(1, 5)), int(rs.randint(1, 40))
        H, W = int(rs.randint(k, 70)), int(rs.randint(k, 70))
        x = (rs.rand(B, C, H, W) * 2 - 1).astype(np.float32)
        x[rs.rand(B, C, H, W) < 0.1] = 0.5  # ties: the first maximum wins (pool2d.cpp:66-75)
        y_ref, m_ref = O.maxpool_forward(x, k, step)
        dy = (rs.rand(*y_ref.shape) * 2 - 1).astype(np.float32)
        dx_ref = O.maxpool_backward(dy, m_ref, x.shape, k, step)
        xd = dev(x)
        y, m = capi.maxpool_forward(xd, k, step)
        dx = capi.maxpool_backward(dev(dy), m, x.shape, k, step, dx=guarded(x.shape))
        xr = np.maximum(x, 0)
        yr, mr = capi.maxpool_forward(dev(xr), k, step)
        yr_ref, mr_ref = O.maxpool_forward(xr, k, step)
        dxr = capi.maxpool_backward_relu(dev(dy), mr, yr, x.shape, k, step, dx=guarded(x.shape))
        dxr_ref = np.where(xr <= 0, np.float32(0), O.maxpool_backward(dy, mr_ref, x.shape, k, step))
        errs = {"y": 0.0 if np.array_equal(y.cpu().numpy().view(np.uint32), y_ref.view(np.uint32)) else 1.0,
                "mask": 0.0 if np.array_equal(m.cpu().numpy(), m_ref) else 1.0,
                "dx": 0.0 if np.array_equal(dx.cpu().numpy().view(np.uint32), dx_ref.view(np.uint32)) else 1.0,
                "dx+relu": 0.0 if np.array_equal(dxr.cpu().numpy().view(np.uint32), dxr_ref.view(np.uint32)) else 1.0}
        errs.update(overruns())
        report(f"pool B{B} C{C} {H}x{W} k{k} step{step}", errs)


def bn_cases():
    global MIS
    MIS = 0  # (include/cnn_amd.h: the BatchNorm2D entry points ask for 16-byte aligned tensors and refuse others with a status)
    for it in range(n_cases):
        B, C = int(rs.randint(1, 9)), int(rs.randint(1, 70))
        H, W = int(rs.randint(1, 40)), int(rs.randint(1, 40))
        if it % 2 == 0:
            H, W = 2 * max(1, H // 2), 4 * max(1, W // 4)  # the shapes the fused pool forms accept
        shape = (B, C, H, W)
        x = (rs.standard_normal(shape) * rs.uniform(0.5, 3) + rs.uniform(-2, 2)).astype(np.float32)
        gamma = (rs.rand(C) + 0.5).astype(np.float32)
        beta = (rs.rand(C) - 0.5).astype(np.float32)
        mm0, mv0 = (rs.rand(C) - 0.5).astype(np.float32), (rs.rand(C) + 0.5).astype(np.float32)
        dy = (rs.rand(*shape) * 2 - 1).astype(np.float32)
        y_o, _, sm_o, sv_o, mm_o, mv_o = O.batchnorm_forward(x, gamma, beta, mm0, mv0)
        dx_o, gg_o, gb_o = O.batchnorm_backward(x, dy, gamma, sm_o, sv_o)
        bn = capi.BatchNorm2d(B, C, H, W)
        xd, gd, bd, mmd, mvd = dev(x), dev(gamma), dev(beta), dev(mm0), dev(mv0)
        yd, rd = guarded(shape), guarded(shape)
        bn.forward(xd, gd, bd, mmd, mvd, yd, training=True, y_relu=rd)
        dyd = dev(dy)
        gg, gb = torch.full((C,), 7.0, device="cuda"), torch.full((C,), 7.0, device="cuda")
        bn.backward(xd, dyd, gd, gg, gb)
        errs = {"y": rel(yd.cpu().numpy(), y_o), "mean": rel(bn.saved_mean.cpu().numpy(), sm_o), "var": rel(bn.saved_var.cpu().numpy(), sv_o),
                "mm": rel(mmd.cpu().numpy(), mm_o), "mv": rel(mvd.cpu().numpy(), mv_o),
                "relu": 0.0 if np.array_equal(rd.cpu().numpy(), np.where(yd.cpu().numpy() >= 0, yd.cpu().numpy(), np.float32(0))) else 1.0}
        if B * H * W > 1:
            errs.update({"dx": rel(dyd.cpu().numpy(), dx_o), "ggamma": rel(gg.cpu().numpy(), gg_o), "gbeta": rel(gb.cpu().numpy(), gb_o)})
        tag = ""
        lib = capi.load()
        if H % 2 == 0 and W % 2 == 0 and lib.cnn_batchnorm2d_forward_relu_pool_supported(B, C, H, W):
            # fused forward: bit-identical to the separate calls
            p1, m1 = capi.maxpool_forward(rd, 2, 2)
            bn2 = capi.BatchNorm2d(B, C, H, W)
            p2, m2 = torch.full_like(p1, 7.0), torch.full_like(m1, -3)
            bn2.forward_relu_pool(xd, gd, bd, dev(mm0), dev(mv0), p2, m2, training=True)
            errs["fused fwd"] = 0.0 if (torch.equal(p1, p2) and torch.equal(m1, m2)) else 1.0
            tag += " +pool-fwd"
            if bn2.backward_pooled_supported():
                dpool = dev((rs.rand(*p1.shape) * 2 - 1).astype(np.float32))
                d_relu = capi.maxpool_backward_relu(dpool, m1, p1, shape, 2, 2)
                gg1, gb1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
                bn.backward(xd, d_relu, gd, gg1, gb1)  # (in place: d_relu becomes dx)
                gg2, gb2, dx2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.full_like(xd, 7.0)
                bn2.backward_pooled(xd, dpool, m2, p2, gd, gg2, gb2, dx2)
                errs["pooled bwd"] = 0.0 if (torch.equal(dx2, d_relu) and torch.equal(gg1, gg2) and torch.equal(gb1, gb2)) else 1.0
                tag += " +pooled-bwd"
        errs.update(overruns())
        report(f"bn {shape}{tag}", errs)


def lin_cases():
    lib = capi.load()
    for it in range(n_cases):
        B, n_in, n_out = int(rs.randint(1, 70)), int(rs.randint(1, 9000)), int(rs.choice([1, 2, 3, 3, 3, 5, 8, 10, 17]))
        if it % 4 == 0:
            n_in = 256 * int(rs.randint(1, 100))
        x = np.maximum(rs.rand(B, n_in) * 2 - 1, 0).astype(np.float32)
        w = (rs.standard_normal((n_in, n_out)) * 0.05).astype(np.float32)
        b = (rs.standard_normal(n_out) * 0.1).astype(np.float32)
        dy = (rs.rand(B, n_out) * 2 - 1).astype(np.float32)
        y_ref = O.linear_forward(x, w, b)
        gw_ref, gb_ref, dx_ref = O.linear_backward(x, dy, w)
        xd, wd = dev(x), dev(w)
        y = capi.linear_forward(xd, wd, dev(b), y=guarded((B, n_out)))
        gw, gb, dx = capi.linear_backward(xd, dev(dy), wd, float(B), gw=guarded((n_in, n_out)), gb=guarded((n_out,)), dx=guarded((B, n_in)))
        gw2, gb2, dxr = capi.linear_backward(xd, dev(dy), wd, float(B), relu_below=True)
        errs = {"y": rel(y.cpu().numpy(), y_ref), "gw": rel(gw.cpu().numpy(), gw_ref), "gb": rel(gb.cpu().numpy(), gb_ref),
                "dx": rel(dx.cpu().numpy(), dx_ref), "dx+relu": rel(dxr.cpu().numpy(), np.where(x <= 0, np.float32(0), dx_ref))}
        errs.update(overruns())
        if n_out > 8:  # (the fused head serves skinny layers: cnn_linear_forward_softmax_xent_dx requires out <= 8)
            report(f"linear B{B} {n_in}->{n_out}", errs)
            continue
        # the fused loss head with dx: against the oracle's softmax + cross entropy + linear backward
        labels = (np.arange(B) % n_out).astype(np.int32)
        probs_ref = O.softmax(y_ref)
        _, delta_ref = O.cross_entropy_backward(probs_ref, labels)
        _, _, dxh_ref = O.linear_backward(x, delta_ref, w)
        mk = lambda *shape: torch.full(shape, 7.0, device="cuda")
        logits, probs, delta, terms, dxh = mk(B, n_out), mk(B, n_out), mk(B, n_out), mk(B), mk(B, n_in)
        b_d, lab_d = dev(b), dev(labels)  # (kept alive: a temporary's memory would be handed to the next allocation before the launch)
        capi.check(lib.cnn_linear_forward_softmax_xent_dx(capi._ptr(xd), capi._ptr(wd), capi._ptr(b_d), capi._ptr(lab_d), capi._ptr(logits),
                                                          capi._ptr(probs), capi._ptr(delta), capi._ptr(terms), capi._ptr(dxh), 1, B, n_in, n_out,
                                                          capi._stream()), "head")
        gwh, gbh = mk(n_in, n_out), mk(n_out)
        capi.check(lib.cnn_linear_backward(capi._ptr(xd), capi._ptr(delta), capi._ptr(wd), capi._ptr(gwh), capi._ptr(gbh), None, B, n_in, n_out, float(B),
                                           capi._stream()), "wb")
        gwh_ref, gbh_ref, _ = O.linear_backward(x, delta_ref, w)
        errs.update({"head logits": rel(logits.cpu().numpy(), y_ref), "head delta": rel(delta.cpu().numpy(), delta_ref),
                     "head dx": rel(dxh.cpu().numpy(), np.where(x <= 0, np.float32(0), dxh_ref)), "head gw": rel(gwh.cpu().numpy(), gwh_ref),
                     "head gb": rel(gbh.cpu().numpy(), gbh_ref)})
        report(f"linear B{B} {n_in}->{n_out}", errs)


def misc_cases():
    import ctypes as C_

    bits = lambda a, b: 0.0 if np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32)) else 1.0
    for it in range(n_cases):
        n = int(rs.randint(1, 300000)) if it % 3 else int(rs.randint(1, 70))
        off = int(rs.randint(0, 4))  # (views that do not start on 16 bytes take the scalar paths)
        x = (rs.rand(n + off) * 2 - 1).astype(np.float32)
        x[rs.rand(n + off) < 0.05] = 0.0
        x[rs.rand(n + off) < 0.02] = -0.0
        d = (rs.rand(n + off) * 2 - 1).astype(np.float32)
        xd, dd = dev(x)[off:], dev(d)[off:]
        y = capi.relu_forward(xd.contiguous() if off == 0 else xd)
        y_ref = O.relu_forward(x[off:])
        dx = capi.relu_backward(y, dd.clone())
        errs = {"relu": bits(y.cpu().numpy(), y_ref), "relu'": bits(dx.cpu().numpy(), O.relu_backward(y_ref, d[off:]))}
        lr = float(rs.choice([1e-3, 0.05, 1.0]))
        p_new = capi.sgd_update(dev(x)[off:].clone(), dd, lr)
        errs["sgd"] = bits(p_new.cpu().numpy(), O.sgd_update(x[off:], d[off:], lr))
        report(f"relu / sgd n={n} offset {off}", errs)
        # softmax + cross entropy (func.cpp:16-73): clamped exp, sequential sums
        B, ncls = int(rs.randint(1, 400)), int(rs.randint(1, 12))
        logits = (rs.standard_normal((B, ncls)) * rs.choice([1, 10, 60])).astype(np.float32)
        labels = rs.randint(0, ncls, B).astype(np.int32)
        probs, delta, loss = capi.softmax_xent(dev(logits), dev(labels))
        p_ref = O.softmax(logits)
        l_ref, d_ref = O.cross_entropy_backward(p_ref, labels)
        lv = float(loss.item()) / B
        errs = {"probs": rel(probs.cpu().numpy(), p_ref), "delta": rel(delta.cpu().numpy(), d_ref),
                "loss": 0.0 if (not np.isfinite(l_ref) and not np.isfinite(lv)) else abs(lv - l_ref) / max(1.0, abs(l_ref))}
        report(f"softmax_xent B{B} classes {ncls}", errs)
        # Dropout (dropout.cpp:7-69) and Grad-CAM (alexnet.cpp:107-140)
        Bq, Cq, Hq, Wq = int(rs.randint(1, 5)), int(rs.randint(1, 100)), int(rs.randint(1, 30)), int(rs.randint(1, 30))
        pq = float(rs.choice([0.1, 0.2, 0.29, 0.5, 0.75]))
        xq = (rs.rand(Bq, Cq, Hq, Wq) * 2 - 1).astype(np.float32)
        yq = capi.dropout_forward(dev(xq), pq, training=True)
        yq_ref = O.dropout_forward(xq, pq, training=True)
        ye = capi.dropout_forward(dev(xq), pq, training=False)
        dq = capi.dropout_backward(dev(xq).clone(), pq)
        # (the oracle's engine state: one training forward above, so its backward sees the same dropped channels -- the reference drops the
        #  FIRST int(p * C) channels of a shuffled sequence seeded 1314: identical on every first call)
        errs = {"dropout train": bits(yq.cpu().numpy(), yq_ref), "dropout eval": bits(ye.cpu().numpy(), O.dropout_forward(xq, pq, training=False))}
        cam, img = capi.grad_cam(dev(np.abs(xq)))
        cam_ref, img_ref = O.grad_cam(np.abs(xq))
        errs["grad-cam map"] = bits(cam.cpu().numpy(), cam_ref)
        errs["grad-cam picture"] = 0.0 if np.array_equal(img.cpu().numpy(), img_ref) else 1.0
        report(f"dropout / grad-cam {(Bq, Cq, Hq, Wq)} p={pq}", errs)
        # the uint8 stager: interleaved bytes -> planar fp32 * (1 / 255) with the reference's expression (byte * 1.f / 255)
        if it % 4 == 0:
            Bs, Hs, Ws = int(rs.randint(1, 5)), int(rs.randint(1, 60)), int(rs.randint(1, 60))
            st = capi.BatchStager(u8_shape=(Bs, Hs, Ws), depth=2)
            img8 = rs.randint(0, 256, (Bs, Hs, Ws, 3)).astype(np.uint8)
            host, slot = st.acquire()
            host[:] = img8.reshape(-1)
            devp = st.submit(slot)
            st.wait(slot)
            out = torch.empty((Bs, 3, Hs, Ws), dtype=torch.float32, device="cuda")
            capi.check(capi.load().cnn_memcpy_d2d(C_.c_void_p(out.data_ptr()), C_.c_void_p(devp), Bs * 3 * Hs * Ws * 4, capi._stream()), "d2d") if hasattr(capi.load(), "cnn_memcpy_d2d") else None
            torch.cuda.synchronize()
            st.release(slot)
            ref = (img8.astype(np.float32) * np.float32(1.0) / np.float32(255)).transpose(0, 3, 1, 2)
            if hasattr(capi.load(), "cnn_memcpy_d2d"):
                report(f"u8 stager {(Bs, Hs, Ws)}", {"planar fp32": bits(out.cpu().numpy(), np.ascontiguousarray(ref))})
            st.close()


for name, fn in (("conv", conv_cases), ("pool", pool_cases), ("bn", bn_cases), ("lin", lin_cases), ("misc", misc_cases)):
    if what in (name, "all"):
        fn()
print(f"FUZZ {'OK' if bad == 0 else 'FAILED'} ({what}): worst error {worst:.2e}, {bad} case(s) above {TOL:.0e}")
sys.exit(1 if bad else 0)
