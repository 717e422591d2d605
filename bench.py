#!/usr/bin/env python3
"""bench.py -- images/sec of one full train step of a BASELINE workload on the HIP path (MI355X).

A "step" = one iteration of cpu/src/cnn.cpp:79-90 on one batch of synthetic 224x224x3 fp32 images already resident
in HBM: forward, softmax + cross-entropy, backward, [RCCL all-reduce of the flat gradient arena], SGD.
    python bench.py --gpus 1 --steps 20 --warmup 5                      # the reference net, BASELINE configs[1] (the metric's config)
    python bench.py --config vgg11 | resnet18                           # configs[3] / [4] alone, with their own CPU legs
`value` is measured through the boundary north_star names: the C++ Layer classes (architectures::Sequential::train_step,
cnn_amd/host) with their DEFAULT settings; the default command adds the Python driver, the every-tensor-written variant, the
north-star convolution and a few steps of configs[3] / [4] as extra keys.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
Timing protocol (round 4): after the warm-up steps and >= 0.6 s of untimed steps (clock ramp), REPEATS = 7 regions of EXACTLY K steps are
timed, each between two barriers + device synchronisations; `value` / `ms_per_step` are the MEDIAN region (max over ranks per region),
`spread` holds the slowest / fastest region.  `reference_loop` is the same workload through the reference's own loop on the unchanged
Layer API (cpu/src/cnn.cpp:79-90, host-side loss glue).
Prints ONE JSON line on rank 0 (contract in the task brief) with two extra objects:
  roofline     -- the dominant kernel of the step: algorithmic bytes (or FLOPs) per launch / its average duration,
                  measured with HIP events on the launch stream DURING the timed region (only that kernel is
                  bracketed, so the perturbation is two event records per step);
  cpu_baseline -- the CPU oracle (oracle/cnn_oracle.c, a line-by-line port of the reference loop nests) timed on the
                  host, 1 thread, bounded sample.
plus conv_ns: the north-star Conv2d forward shape (3x3, 64->128, 112x112, batch 256) against the fp32-MFMA peak.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
PEAK_MFMA_F32_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak (256 CU x 2.4 GHz x 256 FLOP/clk)


def algorithmic_work(key):
    """(bytes, flops) of ONE launch of the kernel identified by "<kernel>|<geometry>" (SURVEY.md 8(d) figures)."""
    kernel, geo = key.split("|", 1)
    m = re.match(r"B(\d+) Ci(\d+) (\d+)x(\d+) Co(\d+) k(\d+) s(\d+) p(\d+)", geo)
    if m:
        B, Ci, H, W, Co, k, s, p = map(int, m.groups())
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        flops = 2.0 * B * Co * Ho * Wo * Ci * k * k
        x_b, y_b, w_b = 4.0 * B * Ci * H * W, 4.0 * B * Co * Ho * Wo, 4.0 * Co * Ci * k * k
        if kernel.endswith(("+pool", "+poolm", "+poolm8")) or kernel.startswith("conv_fwd_pool_pk"):
            # first block in the pooled domain: the Co*Ho*Wo tensors are never touched.  pooled-domain tensor = B*Co*(Ho/2)*(Wo/2)
            pd_b = 4.0 * B * Co * (Ho // 2) * (Wo // 2)
            # the pool mask: an int32 per window, or ("...8": CNN_CONV2D_POOL_MASK_PACKED) one byte per window in 4-byte-aligned rows
            mk_b = 1.0 * B * Co * (Ho // 2) * (((Wo // 2) + 3) & ~3) if kernel.endswith("8") else pd_b
            fl = 2.0 * B * Co * (2 * (Ho // 2)) * (2 * (Wo // 2)) * Ci * k * k  # conv pixels inside a pooling window
            if kernel.startswith("conv_fwd_pool_pk"):
                return x_b + pd_b + mk_b + w_b, fl       # x, w -> pooled + mask
            nt = 1 if kernel.endswith(("+poolm", "+poolm8")) else 2    # "+poolm": the ReLU mask rides in the pool mask, the pooled tensor is not read
            return x_b + nt * pd_b + mk_b + w_b, fl       # wgrad: x + (dpool, mask[, pooled]) -> gw ; dgrad: those -> dx
        if kernel.startswith(("igemm_kernel", "igemm_dma_kernel", "wgrad_kernel", "wgrad_rd", "wgrad_sp", "conv_rows", "conv_direct", "conv_dgrad_pk", "conv_fwd_pk",
                              "conv_wgrad_pk", "conv_wgrad_win", "conv_wgrad_os", "conv_stem", "conv_dgrad_thin", "conv_fwd_rd", "conv_dgrad_rd", "conv_1x1")):
            fused = y_b if (kernel.endswith("/fwd+relu") or ",relu" in kernel or kernel.endswith(">+relu")) else 0.0  # second output tensor
            if "dgrad" in kernel and kernel.endswith("+relu"):
                fused = x_b  # fused ReLU::backward: the mask tensor (shape of dx) is read
            return x_b + y_b + w_b + fused, flops  # fwd: x,w -> y ; dgrad: dy,w -> dx ; wgrad: x,dy -> gw
        if kernel.startswith("bias_grad_partial"):
            return y_b, B * Co * Ho * Wo
        return w_b * 2, 0.0  # prep / reduce kernels: weight-sized
    m = re.match(r"n=(\d+)", geo)
    if m:
        n = int(m.group(1))
        per = {"relu_fwd_vec": 8, "relu_bwd_vec": 12, "sgd_vec": 12}.get(kernel, 8)
        return float(per * n), float(n)
    m = re.match(r"B(\d+) C(\d+) (\d+)x(\d+) k(\d+) step(\d+)", geo)
    if m:
        B, C, H, W, k, st = map(int, m.groups())
        Ho, Wo = (H - k) // st + 1, (W - k) // st + 1
        if "fwd" in kernel:
            return 4.0 * B * C * (H * W + 2 * Ho * Wo), float(B * C * Ho * Wo * k * k)
        fused = 1 if kernel.endswith("+relu") else 0  # fused ReLU backward also reads the pooled output
        return 4.0 * B * C * ((2 + fused) * Ho * Wo + H * W), 0.0
    m = re.match(r"B(\d+) in(\d+) out(\d+)", geo)
    if m:
        B, n_in, n_out = map(int, m.groups())
        return 4.0 * (B * n_in + n_in * n_out + B * n_out), 2.0 * B * n_in * n_out
    return 0.0, 0.0


def pmc_traffic(key):
    """HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (tools/profile_bench.sh ->
    tools/pmc_traffic.py -> profiles/rNN/hbm_traffic.json), or None when that kernel was not profiled / is ambiguous."""
    import glob

    name = key.split("|")[0].split("/")[0].replace(" ", "").replace("+relu", "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic.json")), reverse=True):
        try:
            kernels = json.load(open(path))["kernels"]
        except Exception:
            continue
        hits = [v for k, v in kernels.items() if k.split("|")[0].replace(" ", "") == name]
        if len(hits) == 1:
            return hits[0]["hbm_bytes"], os.path.relpath(path, ROOT)
    return None, None


def roofline_entry(key, launches, total_ms, with_traffic=False):
    nbytes, flops = algorithmic_work(key)
    avg_s = total_ms / 1e3 / max(launches, 1)
    ridge = PEAK_MFMA_F32_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    if nbytes > 0 and flops / nbytes >= ridge:
        ach = flops / avg_s / 1e12
        return {"bound": "mfma", "kernel": key, "achieved": round(ach, 3), "peak": PEAK_MFMA_F32_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F32_TFLOPS, 4), "traffic": None,
                "avg_us": round(avg_s * 1e6, 2), "launches": launches}
    ach = nbytes / avg_s / 1e9
    traffic, src = pmc_traffic(key) if with_traffic else (None, None)
    out = {"bound": "hbm", "kernel": key, "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic, "avg_us": round(avg_s * 1e6, 2),
           "launches": launches, "algorithmic_bytes": nbytes}
    if src:
        out["traffic_source"] = src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 on gfx950)"
    return out


def cpu_baseline(batch=16, budget_s=12.0):
    """the reference's own loop nests (oracle port) on 1 host thread: full train steps at batch 16 (BASELINE config 1)"""
    import numpy as np

    from oracle import pyoracle as O

    rs = np.random.RandomState(0)
    x = rs.rand(batch, 3, 224, 224).astype(np.float32)
    labels = (np.arange(batch) % 3).astype(np.int32)
    net = O.Net(batch, 3)
    net.params[:] = (rs.standard_normal(net.n_params) * 0.1).astype(np.float32)
    net.train_step(x, labels, 1e-3)  # warm-up
    steps, t0 = 0, time.perf_counter()
    while True:
        net.train_step(x, labels, 1e-3)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 64:
            break
    return {"value": round(steps * batch / el, 2), "unit": "images/sec", "cores": 1, "kind": "port",
            "host_cores": os.cpu_count(),
            "sample": f"{steps} full train steps of the reference net at batch {batch}, 224x224x3 (after 1 warm-up); "
                      f"oracle/cnn_oracle.c built -O2 without FMA like cpu/CMakeLists.txt:5"}


def conv_ns_bench(torch, capi, reps=5):
    """Conv2d forward, 3x3, 64->128, 112x112 (pad 0 -> 110x110), batch 256: FLOPs / kernel time vs fp32-MFMA peak"""
    case = (256, 64, 112, 112, 128, 3, 1, 0)
    conv = capi.Conv2d(*case)
    conv.autotune()  # (what Conv2D::forward does on its first call: the library measures which of its kernels / tiles runs this shape)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand((256, 64, 112, 112), generator=g, device="cuda")
    w = torch.randn((128, 64, 3, 3), generator=g, device="cuda") * 0.1
    b = torch.randn((128,), generator=g, device="cuda") * 0.1
    y = torch.empty(conv.out_shape(), device="cuda")
    dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    dx = torch.empty_like(x)
    for _ in range(2):
        conv.forward(x, w, b, y)
        conv.backward_data(dy, w, dx)
        conv.backward_weight(x, dy, 256.0)
    torch.cuda.synchronize()
    capi.kernel_timing(1)
    for _ in range(reps):
        conv.forward(x, w, b, y)
        conv.backward_data(dy, w, dx)
        conv.backward_weight(x, dy, 256.0)
    rep = capi.kernel_timing_report()
    capi.kernel_timing(0)
    flops = 2.0 * 256 * 128 * 110 * 110 * 64 * 9
    out = {"shape": "B256 64->128 k3 s1 112x112->110x110", "gflop": round(flops / 1e9, 2), "peak_tflops": PEAK_MFMA_F32_TFLOPS}
    for key, (cnt, ms) in rep.items():
        name = key.split("|")[0]
        if name.startswith(("igemm_kernel", "igemm_dma_kernel", "wgrad_kernel", "wgrad_rd", "wgrad_sp", "conv_rows", "conv_fwd_rd", "conv_dgrad_rd")):
            tf = flops / (ms / 1e3 / cnt) / 1e12
            tag = "fwd" if name.endswith(("/fwd", "/fwd+relu")) else ("dgrad" if name.endswith("/dgrad") else "wgrad")
            out[tag] = {"kernel": name, "avg_ms": round(ms / cnt, 4), "tflops": round(tf, 2),
                        "frac_of_mfma_peak": round(tf / PEAK_MFMA_F32_TFLOPS, 4)}
    # HBM-side traffic per launch from the committed PMC passes of the same three launches (tools/pmc_conv_ns.sh), next to the algorithmic bytes
    import glob

    alg = 4.0 * (256 * 64 * 112 * 112 + 256 * 128 * 110 * 110 + 128 * 64 * 9)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic_conv_ns.json")), reverse=True)[:1]:
        try:
            kernels = json.load(open(path))["kernels"]
        except Exception:
            break
        for tag in ("fwd", "dgrad", "wgrad"):
            if tag not in out:
                continue
            m = re.match(r"(conv_rows|wgrad_sp)<(\d+,\d+),", out[tag]["kernel"])
            if not m:
                continue
            hits = [v for k, v in kernels.items() if k.replace(" ", "").startswith(f"{m.group(1)}_kernel<{m.group(2)},")]
            if len(hits) == 1:
                out[tag]["algorithmic_bytes"] = alg
                out[tag]["traffic"] = hits[0]["hbm_bytes"]
                out[tag]["traffic_source"] = os.path.relpath(path, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH x2 on gfx950; L2 misses: served by the Infinity Cache or HBM)"
    del x, y, dy, dx
    torch.cuda.empty_cache()
    return out


def cpu_baseline_stack(name, budget_s=25.0):
    """the oracle's layer functions composed into the VGG-11 / ResNet-18-shaped list (oracle.pyoracle.SeqNet), 1 thread"""
    import numpy as np

    from cnn_amd import stacks
    from oracle import pyoracle as O

    O.set_threads(1)
    batch = 1 if name == "vgg11" else 2  # (BatchNorm2D needs more than one sample to be meaningful)
    spec = stacks.STACKS[name]()
    net = O.SeqNet(spec)
    net.params[:] = stacks.he_init(stacks.walk(spec), 1234)
    rs = np.random.RandomState(0)
    x = rs.rand(batch, 3, 224, 224).astype(np.float32)
    labels = (np.arange(batch) % 3).astype(np.int32)
    steps, t0 = 0, time.perf_counter()
    while True:
        net.train_step(x, labels, 1e-3)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 8:
            break
    return {"value": round(steps * batch / el, 3), "unit": "images/sec", "cores": 1, "kind": "port", "host_cores": os.cpu_count(),
            "sample": f"{steps} full train step(s) of the {name}-shaped stack at batch {batch}, 224x224x3 (no warm-up: one step is "
                      f"~{el / steps:.0f} s); oracle/cnn_oracle.c layer functions composed by oracle/pyoracle.py SeqNet, built -O2 "
                      f"without FMA like cpu/CMakeLists.txt:5"}


def staged_input_bench(torch, capi, args):
    """the host-buffer variant of the boundary (the reference's DataLoader hands over host tensors, pipeline.cpp:129-140): every
    step's batch is uploaded from a pinned slot by the double-buffered stager while the previous step computes"""
    import numpy as np

    from cnn_amd import hostapi

    B = args.batch or 256
    net = hostapi.HostAlexNet(3)
    rs = np.random.RandomState(1234)
    net.set_params((rs.standard_normal(net.n_params) * 0.1).astype(np.float32))
    labels = (torch.arange(B, device="cuda") % 3).to(torch.int32)
    nbytes = B * 3 * 224 * 224 * 4
    stager = capi.BatchStager(nbytes, depth=2)
    for _ in range(2):  # synthetic images in both pinned slots (a real loader would decode into them)
        host, slot = stager.acquire()
        host[:] = rs.rand(host.size).astype(np.float32)
        stager.submit(slot)

    def step():
        host, slot = stager.acquire()   # (the producer would fill `host` here)
        dev = stager.submit(slot)       # H2D on the copy stream
        stager.wait(slot)               # the compute stream waits for it
        net.train_step_ptr(dev, labels, B, 224, 224, 1e-3)
        stager.release(slot)

    el = median(timed_regions(step, torch.cuda.synchronize, args.steps, args.warmup, repeats=3))
    loss = net.last_loss()
    stager.close()
    out = {"value": round(B * args.steps / el, 1), "unit": "images/sec", "ms_per_step": round(el / args.steps * 1e3, 4),
           "h2d_GBps": round(nbytes * args.steps / el / 1e9, 1), "final_loss": round(loss, 5),
           "note": "every batch uploaded from pinned host memory (cnn_batch_stager_*, 2 slots, copy stream overlapped with compute)"}
    # the same with what a real input is in front of Tensor3D::read_from_opencv_mat (data_format.cpp:13-23): interleaved BYTES, a quarter
    # of the PCIe traffic; the stager's kernel writes the fp32 planar batch (bit-identical conversion) on the copy stream
    stager = capi.BatchStager(u8_shape=(B, 224, 224), depth=2)
    for _ in range(2):
        host, slot = stager.acquire()
        host[:] = rs.randint(0, 256, size=host.size, dtype=np.uint8)
        stager.submit(slot)
    el8 = median(timed_regions(step, torch.cuda.synchronize, args.steps, args.warmup, repeats=3))
    loss8 = net.last_loss()
    stager.close()
    net.close()
    out["u8"] = {"value": round(B * args.steps / el8, 1), "unit": "images/sec", "ms_per_step": round(el8 / args.steps * 1e3, 4),
                 "h2d_GBps": round(B * 224 * 224 * 3 * args.steps / el8 / 1e9, 1), "final_loss": round(loss8, 5),
                 "device_conversion_bytes_per_batch": B * 224 * 224 * 3 * 5,
                 "note": "cnn_batch_stager_create_u8: 150 KB of bytes per image over PCIe instead of 602 KB of floats; byte * 1.f / 255 through a "
                         "host-filled 256-entry table on the copy stream (reads B*150 KB, writes B*602 KB of HBM per batch); the first-layer "
                         "kernels still read the fp32 batch"}
    return out


def reference_loop_leg(torch, args):
    """The SAME workload through the reference's own training loop on the UNCHANGED Layer API -- cpu/src/cnn.cpp:79-90:
    network.forward(batch) -> softmax(output) -> cross_entroy_backward(probs, one_hot(labels)) -> network.backward(delta) ->
    network.update_gradients(lr) -- with the host-side func.cpp loss glue (the logits are read back and the loss delta uploaded
    every step, as a maintainer who drops the library into cpu/src gets it; cnnh_net_train_step_device in host/src/host_capi.cpp).
    `value` above is architectures::Sequential::train_step, an ADDITION to that API (device-side loss, fused step tail)."""
    import numpy as np

    from cnn_amd import hostapi

    B = args.batch or 256
    net = hostapi.HostAlexNet(3)
    rs = np.random.RandomState(1234)
    net.set_params((rs.standard_normal(net.n_params) * 0.1).astype(np.float32))
    g = torch.Generator(device="cuda").manual_seed(100)
    x = torch.rand((B, 3, 224, 224), generator=g, device="cuda")
    labels = (np.arange(B) % 3).astype(np.int32)
    last = [0.0]

    def step():
        last[0] = net.train_step_device(x, labels, 1e-3)

    regions = timed_regions(step, torch.cuda.synchronize, args.steps, args.warmup, repeats=3)
    out = dict(leg_result(B, args.steps, regions), final_loss=round(last[0], 5),
               driver="C++ Layer API, the reference's loop verbatim: AlexNet::forward -> softmax -> cross_entroy_backward -> "
                      "AlexNet::backward -> AlexNet::update_gradients (cpu/src/cnn.cpp:79-90; host-side loss glue, one D2H + one H2D "
                      "per step), default settings")
    net.close()
    del x
    torch.cuda.empty_cache()
    return out


def rccl_log_excerpt(path, limit=14):
    """what RCCL says it chose (NCCL_DEBUG=INFO lines of THIS process: version, channels, rings / trees, algorithm / protocol) -- the first
    8-GPU run then records the transport it actually used, not just images/s"""
    import glob
    import re

    keep = re.compile(r"(version|Init COMPLETE|[Cc]hannel|Ring|Tree|nranks|Algo|Proto|xgmi|XGMI|P2P|via)")
    seen, out = set(), []
    for f in sorted(glob.glob(path.replace("%p", "*").replace("%h", "*"))):
        try:
            for line in open(f, errors="replace"):
                line = re.sub(r"^\S+:\d+:\d+ \[\d+\] ", "", line.strip())
                key = re.sub(r"0x[0-9a-f]+|\d+", "#", line)
                if keep.search(line) and key not in seen:
                    seen.add(key)
                    out.append(line[:200])
        except OSError:
            pass
    return out[:limit]


def init_comm(capi, torch, dist, world, rank):
    """the data path's exchange (cnn_amd.dp.RcclComm: C ABI -> RCCL), checked once against torch.distributed's own all-reduce"""
    from cnn_amd.dp import RcclComm

    comm = RcclComm(dist, world, rank)
    probe = torch.arange(1024, device="cuda", dtype=torch.float32) * (rank + 1)
    ref = probe.clone()
    comm.all_reduce(probe)
    dist.all_reduce(ref)
    torch.cuda.synchronize()
    same = bool(torch.equal(probe, ref))  # (integers below 2^24 times a small rank factor: any summation order gives the same floats)
    if not same:
        print("bench.py: cnn_allreduce_grads disagrees with torch.distributed.all_reduce", file=sys.stderr)
    return comm, {"ranks": world, "rccl_version": comm.version, "entry": "cnn_allreduce_grads (include/cnn_amd.h)",
                  "allreduce_matches_torch_distributed": same,
                  "NCCL_ALGO": os.environ.get("NCCL_ALGO", "default"), "NCCL_PROTO": os.environ.get("NCCL_PROTO", "default")}


def make_runner(config, api, batch, torch, capi, world, rank, comm, pool_block=True):
    """-> dict(step, flush, loss, ...): one train step (cnn.cpp:79-90) of a BASELINE workload on a device-resident synthetic
    batch.  api "layer" (default, what `value` is measured through): the C++ Layer API -- architectures::Sequential::train_step
    (cnn_amd/host) -> C ABI -- with its default settings; api "pynet": the Python driver -> C ABI (reference net only)."""
    import numpy as np

    from cnn_amd import stacks

    B = batch or stacks.DEFAULT_BATCH[config]
    g = torch.Generator(device="cuda").manual_seed(100 + rank)  # each rank has its own shard of the global batch
    x = torch.rand((B, 3, 224, 224), generator=g, device="cuda")
    labels = (torch.arange(B, device="cuda") % 3).to(torch.int32)
    lr = 1e-3
    rs = np.random.RandomState(1234)  # identical init on every rank: replicas stay in lock-step without a broadcast
    if config == "alexnet" and api == "pynet":
        from cnn_amd.pynet import AlexNetHip

        # defer_input_grad: conv_layer_1's data gradient (no consumer) is launched one forward pass later on a second stream;
        # flush() launches a pending one, so the timed region contains exactly K of them (cnn_amd/pynet.py)
        net = AlexNetHip(B, 3, defer_input_grad=True, fuse_pool=pool_block)
        net.load_params((rs.standard_normal(net.n_params) * 0.1).astype(np.float32))
        handle = comm if world > 1 else None
        return dict(step=lambda: net.train_step(x, labels, lr, handle, world), flush=net.flush, B=B, n_params=net.n_params,
                    loss=lambda: float(net.loss_sum.item()) / B, keep=(net, x, labels), close=lambda: None, params=None,
                    api="python driver (cnn_amd/pynet.py) -> C ABI; first block pool-fused, conv_layer_1 data gradient deferred")
    from cnn_amd import hostapi

    lib = hostapi.load()
    lib.cnnh_set_fuse_pool_block(1 if pool_block else 0)  # (1 = the library's default)
    if config == "alexnet":
        net = hostapi.HostAlexNet(3)
        net.set_params((rs.standard_normal(net.n_params) * 0.1).astype(np.float32))
    else:
        spec = stacks.STACKS[config]()
        net = hostapi.HostSequential(spec)
        net.set_params(stacks.he_init(net.layout, 1234))
    if comm is not None and api != "pynet":
        net.set_comm(comm.handle, world)  # train_step exchanges the gradient arena (two buckets for the reference net); BatchNorm2D runs as sync-BN

    def close():
        net.close()
        lib.cnnh_set_fuse_pool_block(1)

    return dict(step=lambda: net.train_step(x, labels, lr), flush=net.flush, B=B, n_params=net.n_params, loss=net.last_loss,
                keep=(net, x, labels), close=close, params=net.get_params,
                api="C++ Layer API (architectures::Sequential::train_step, cnn_amd/host) -> C ABI, default settings"
                    + ("" if pool_block else " except architectures::fuse_pool_block = false (every tensor written by the pass itself)"))


REPEATS = 7          # timed regions of exactly K steps each; `value` is their MEDIAN, `spread` their min / max
CLOCK_WARMUP_S = 0.6  # untimed steps in front of the first region until the clocks have ramped (the driver's box was 7 % slower)


def measure(run, steps, warmup, sample_every, capi, barrier, time_mod=time, repeats=1, clock_warmup_s=0.0, agree=lambda v: v):
    """the measurement protocol of one workload: 3 plain + 3 instrumented steps (which kernel dominates?), `warmup` steps [+ steps
    until `clock_warmup_s` have passed], then `repeats` timed regions of exactly `steps` steps, each between two barriers, with
    only the dominant kernel event-bracketed -> (list of elapsed_s per region, key, launches, total_ms, table).
    agree(v): rank 0's value of v on every rank (identity on one rank) -- every decision that changes HOW MANY steps a rank runs goes
    through it: a step is a collective at N > 1, and ranks that disagree about the count would hang each other"""
    step = run["step"]
    for _ in range(3):
        step()
    barrier()
    capi.kernel_timing(1)
    for _ in range(3):
        step()
    barrier()
    table = capi.kernel_timing_report()
    capi.kernel_timing(0)
    ranked = sorted(((k, v) for k, v in table.items() if not k.startswith("span:")), key=lambda kv: -kv[1][1])  # (spans are waits, not kernels)
    dominant = ranked[0][0]
    if agree(1 if (len(ranked) > 1 and ranked[1][1][1] >= 0.8 * ranked[0][1][1]) else 0):
        # two kernels within 20 % of each other in the fully instrumented steps (every launch event-bracketed: that changes what
        # overlaps what): decide between them the way the timed region measures -- plain steps, ONE kernel sampled -- so that the
        # roofline entry does not flip between runs (reference net: the first layer's weight gradient 78 vs its data gradient 72 us
        # instrumented, 78 vs 65 us in a plain step)
        in_situ = {}
        for key, _ in (ranked[:2] if len(ranked) > 1 else ranked[:1] * 2):
            capi.kernel_timing(2, key, every=sample_every)
            for _ in range(max(steps, 8)):
                step()
            barrier()
            rep = capi.kernel_timing_report()
            capi.kernel_timing(0)
            cnt, ms = rep.get(key, (0, 0.0))
            in_situ[key] = ms / cnt if cnt else 0.0
        dominant = max(in_situ.items(), key=lambda kv: kv[1])[0]
    for _ in range(warmup):
        step()
    if clock_warmup_s > 0:
        barrier()
        t0 = time_mod.perf_counter()
        while agree(1 if time_mod.perf_counter() - t0 < clock_warmup_s else 0):
            for _ in range(max(steps, 1)):
                step()
            barrier()
    capi.kernel_timing(2, dominant, every=sample_every)
    regions = []
    for _ in range(repeats):
        barrier()
        t0 = time_mod.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        regions.append(time_mod.perf_counter() - t0)
    dom = capi.kernel_timing_report()
    capi.kernel_timing(0)
    cnt, ms = dom[dominant]
    return regions, dominant, cnt, ms, table


def median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def timed_regions(step, barrier, steps, warmup, repeats=REPEATS, clock_warmup_s=CLOCK_WARMUP_S):
    """extra legs: `warmup` steps, clock warm-up, then `repeats` regions of `steps` steps -> list of elapsed_s"""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < clock_warmup_s:
        for _ in range(max(steps, 1)):
            step()
        barrier()
    out = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        out.append(time.perf_counter() - t0)
    return out


def leg_result(B, steps, regions):
    el = median(regions)
    return {"value": round(B * steps / el, 1), "unit": "images/sec", "ms_per_step": round(el / steps * 1e3, 4),
            "spread": {"repeats": len(regions), "min": round(B * steps / max(regions), 1), "max": round(B * steps / min(regions), 1)}}


def conv_kernel_alone(key, torch, capi, reps=5):
    """The stacks run a layer's weight gradient BESIDE its data gradient (two streams), so the in-situ duration of either is the pair's.
    This times the dominant kernel's pass ALONE -- same geometry, same kernel, nothing else on the chip -- next to the in-situ figure
    `roofline` is computed from (never instead of it).  None when the key is not a convolution pass."""
    m = re.match(r"B(\d+) Ci(\d+) (\d+)x(\d+) Co(\d+) k(\d+) s(\d+) p(\d+)", key.split("|", 1)[1] if "|" in key else "")
    if not m or algorithmic_work(key)[1] <= 0:
        return None
    case = tuple(int(v) for v in m.groups())
    B, Ci, H, W, Co, k, s_, pad = case
    name = key.split("|")[0]
    conv = capi.Conv2d(*case)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand((B, Ci, H, W), generator=g, device="cuda")
    w = torch.randn((Co, Ci, k, k), generator=g, device="cuda") * 0.05
    b = torch.zeros((Co,), device="cuda")
    dy = torch.rand(conv.out_shape(), generator=g, device="cuda") * 2 - 1
    op = "fwd" if "/fwd" in name or "_fwd" in name else ("dgrad" if "dgrad" in name else "wgrad")

    def run():
        if op == "fwd":
            conv.forward(x, w, b)
        elif op == "dgrad":
            conv.backward_data(dy, w)
        else:
            conv.backward_weight(x, dy, float(B))

    run(); run()
    torch.cuda.synchronize()
    capi.kernel_timing(1)
    for _ in range(reps):
        run()
    rep = capi.kernel_timing_report()
    capi.kernel_timing(0)
    del x, w, dy
    torch.cuda.empty_cache()
    # the pass's main kernel (the same family as the in-situ key: the fused-epilogue variants of the step share its main loop)
    fam = name.split("<")[0]
    hits = [(kk, v) for kk, v in rep.items() if kk.split("<")[0] == fam]
    if not hits:
        return None
    kk, (cnt, ms) = max(hits, key=lambda kv: kv[1][1])
    us = ms / cnt * 1e3
    tf = algorithmic_work(key)[1] / (us * 1e-6) / 1e12
    return {"kernel": kk.split("|")[0], "avg_us": round(us, 2), "achieved": round(tf, 3), "unit": "TFLOP/s", "frac": round(tf / PEAK_MFMA_F32_TFLOPS, 4),
            "note": "the same pass alone on the chip (in the step it shares the chip with the layer's other gradient on a second stream)"}


def stack_leg(config, torch, capi, steps=5, warmup=2):
    """BASELINE configs[3] / [4] inside the default command: a few timed steps of the stack through the C++ Layer API (no CPU leg)"""
    from cnn_amd import stacks

    run = make_runner(config, "layer", None, torch, capi, 1, 0, None)

    def barrier():
        run["flush"]()
        torch.cuda.synchronize()

    regions, dominant, cnt, ms, _ = measure(run, steps, warmup, 1, capi, barrier, repeats=3)
    elapsed = median(regions)
    B = run["B"]
    flops_img = stacks.train_flops_per_image(stacks.STACKS[config]())
    tf = flops_img * B * steps / elapsed / 1e12
    out = {"workload": WORKLOADS[config], "value": round(B * steps / elapsed, 1), "unit": "images/sec", "per_gpu_batch": B, "steps": steps,
           "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
           "spread": {"repeats": len(regions), "min": round(B * steps / max(regions), 1), "max": round(B * steps / min(regions), 1)},
           "step_tflops": round(tf, 2),
           "step_frac_of_mfma_peak": round(tf / PEAK_MFMA_F32_TFLOPS, 4), "roofline": roofline_entry(dominant, cnt, ms),
           "final_loss": round(run["loss"](), 5), "driver": run["api"]}
    run["close"]()
    del run
    torch.cuda.empty_cache()
    alone = conv_kernel_alone(dominant, torch, capi)
    if alone:
        out["roofline"]["alone"] = alone
    return out


WORKLOADS = {
    "alexnet": "reference AlexNet-style net (cpu/src/alexnet.cpp:10-33: 4x Conv3x3/s2 + ReLU, one MaxPool2x2, Linear 4608->3), "
               "full train step (fwd + softmax/CE + bwd + SGD), 224x224x3 fp32, BASELINE configs[1]",
    "vgg11": "VGG-11-shaped stack of the reference's layer types (8x Conv3x3/s1/p1 3->64->128->256->256->512->512->512->512 + ReLU, "
             "MaxPool2x2 after convs 1,2,4,6,8, Linear 25088->3), full train step, 224x224x3 fp32, BASELINE configs[3]",
    "resnet18": "ResNet-18-shaped sequential stack (7x7/s2 stem, MaxPool2x2, 16 convs in 4 stages: 3x3/s1/p1 + stage entries 3x3/s2, "
                "1x1/s2, 3x3/s2; BatchNorm2D + ReLU after every conv; Linear 25088->3), full train step, 224x224x3 fp32, "
                "BASELINE configs[4] (batch 512 over 8 GPUs = 64 per GPU)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["alexnet", "vgg11", "resnet18"], default="alexnet",
                    help="alexnet = BASELINE configs[1] (the metric's configuration), vgg11 = configs[3], resnet18 = configs[4]")
    ap.add_argument("--api", choices=["pynet", "layer"], default=None,
                    help="layer (default): C++ Layer API (architectures::Sequential::train_step) -> C ABI; pynet: Python driver -> C ABI "
                         "(reference net only)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 256 / 128 / 64 for alexnet / vgg11 / resnet18)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-conv-ns", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the extra legs of the default config (unfused C++ path, Python driver)")
    ap.add_argument("--no-stacks", action="store_true", help="skip the VGG-11 / ResNet-18-shaped legs of the default config")
    ap.add_argument("--one-rank-comm", action="store_true",
                    help="(profiling) single process with a ONE-rank RCCL communicator and the gradient exchange forced on: the step then "
                         "contains the all-reduce kernels of the N > 1 path (sums are identities)")
    ap.add_argument("--main-priority", type=int, default=0, help="(A/B) run the step on a torch stream of this priority (-1 = high)")
    ap.add_argument("--no-pool-fusion", action="store_true", help="architectures::fuse_pool_block = false for the main leg (A/B)")
    ap.add_argument("--staged-input", action="store_true",
                    help="also time the reference net with every batch coming from (pinned) HOST memory through cnn_batch_stager_* "
                         "-- the PCIe-inclusive rate, reported beside `value`, never as it")
    ap.add_argument("--repeats", type=int, default=None,
                    help=f"timed regions of --steps steps each; value = their median (default {REPEATS}; 3 for the stacks)")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-kernel table to stderr")
    args = ap.parse_args()
    small = args.config == "alexnet"
    if args.steps is None:
        args.steps = 100 if small else 10
    if args.warmup is None:
        args.warmup = 20 if small else 3
    api = args.api or "layer"
    if api == "pynet" and not small:
        raise SystemExit("--api pynet drives the reference net only; the stacks run through the C++ Layer API")

    rccl_log = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.one_rank_comm:
        # RCCL's own account of what it set up (version, channels, rings / trees): INFO-level init lines of this process into a file, an
        # excerpt of rank 0's goes into the JSON line.  Set BEFORE torch (and with it librccl) is loaded; init-time logging only, the
        # steady state is not touched.
        rccl_log = f"/tmp/cnn_amd_rccl_{os.getpid()}_%p.log"
        os.environ["NCCL_DEBUG"] = "INFO"  # (assigned, not defaulted: the image may export NCCL_DEBUG=WARN; the file keeps stdout = one JSON line)
        os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,ENV"
        os.environ["NCCL_DEBUG_FILE"] = rccl_log
    import numpy as np
    import torch

    from cnn_amd import capi, stacks

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    arch = capi.load().cnn_amd_device_arch().decode()
    assert arch == "gfx950", f"libcnn_amd.so targets gfx950, device reports {arch}"
    dist, comm, comm_info = None, None, None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # plumbing: rendezvous, barrier, max-over-ranks
        comm, comm_info = init_comm(capi, torch, dist, world, rank)             # the data path's exchange: C ABI -> RCCL
    elif args.one_rank_comm:
        from cnn_amd.dp import RcclComm

        comm = RcclComm(None, 1, 0)
        capi.set_option("DP_FORCE_EXCHANGE", "1")
        comm_info = {"ranks": 1, "rccl_version": comm.version, "entry": "cnn_allreduce_grads (include/cnn_amd.h)",
                     "note": "one-rank communicator, exchange forced on (profiling aid, not a scaling number)"}

    if args.main_priority != 0:
        from cnn_amd import hostapi

        main_stream = torch.cuda.Stream(priority=args.main_priority)
        torch.cuda.set_stream(main_stream)
        hostapi.load().cnnh_set_stream(main_stream.cuda_stream)
    run = make_runner(args.config, api, args.batch, torch, capi, world, rank, comm, pool_block=not args.no_pool_fusion)
    B = run["B"]

    def barrier():
        run["flush"]()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: three plain + three instrumented steps to find the dominant kernel, then the warm-up; timed region: exactly K
    # steps, only the dominant kernel event-bracketed (every 4th launch of it for the 0.45 ms net: the event pair around a
    # kernel costs its stream two ~6 us bubbles -- measured 13 us per step)
    every = 4 if small else 1
    repeats = args.repeats or (REPEATS if small else 3)
    def agree(v):  # rank 0's decision on every rank (see measure())
        if world == 1:
            return v
        t = torch.tensor([int(v)], device="cuda", dtype=torch.int64)
        dist.broadcast(t, src=0)
        return int(t.item())

    regions, dominant, cnt, ms, table = measure(run, args.steps, args.warmup, every, capi, barrier, repeats=repeats,
                                                clock_warmup_s=CLOCK_WARMUP_S if small else 0.0, agree=agree)
    if world > 1:  # every region's time is the MAX over the ranks
        t = torch.tensor(regions, device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions = [float(v) for v in t.tolist()]
    elapsed = median(regions)  # `value` / `ms_per_step`: the median region of exactly K steps
    if comm_info is not None and api != "pynet":
        # how much of the gradient exchange is EXPOSED: the compute stream's wait for the communication stream at the end of the backward
        # walk (Sequential::backward / allreduce_gradients bracket it as "span:exchange_wait"), sampled in K more (untimed) steps on every
        # rank; the step's all-reduce kernels themselves run on the communication stream under the backward pass
        capi.kernel_timing(2, "span:exchange_wait")
        for _ in range(args.steps):
            run["step"]()
        barrier()
        rep = capi.kernel_timing_report()
        capi.kernel_timing(0)
        cnt_x, ms_x = next(iter(rep.values()), (0, 0.0))
        exposed = torch.tensor([ms_x / cnt_x * 1e3 if cnt_x else -1.0], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(exposed, op=dist.ReduceOp.MAX)
        # (-1: this workload's step has no such wait -- the reference net's fused tail runs its second bucket ON the compute stream)
        comm_info["exchange_exposed_us"] = round(float(exposed.item()), 1) if float(exposed.item()) >= 0 else None
        comm_info["exchange_exposed_note"] = "max over ranks of the mean compute-stream wait for the exchange per step (HIP events around the wait)"
    loss = run["loss"]()
    assert np.isfinite(loss), "training diverged: loss is not finite"
    if world > 1 and run["params"] is not None:
        # replicas start identical and apply identical reduced gradients: after ALL the steps above their parameters must be the same
        # bits on every rank; a digest per rank, gathered and compared
        import hashlib

        digest = int.from_bytes(hashlib.sha256(run["params"]().tobytes()).digest()[:7], "little")
        mine = torch.tensor([digest], device="cuda", dtype=torch.int64)
        every_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every_rank, mine)
        digests = [int(t.item()) for t in every_rank]
        # (reported, not asserted: a run that has never been observed with more than one rank must leave its evidence in the JSON line
        #  either way -- a False here means the number above is NOT a valid data-parallel step)
        comm_info["param_digests_equal_on_all_ranks"] = len(set(digests)) == 1
        comm_info["ranks_match_gpus"] = comm_info["ranks"] == args.gpus == world
        comm_info["param_digest"] = f"{digest:014x}"
        if len(set(digests)) != 1:
            comm_info["param_digests_per_rank"] = [f"{d:014x}" for d in digests]
            print(f"bench.py: replicas diverged: parameter digests per rank {digests}", file=sys.stderr)
    if comm_info is not None and not capi.load().cnn_amd_measure_build():
        # (the step without its all-reduce calls computes wrong results: that switch exists only in the measurement build of the library,
        # `make -C cnn_amd/csrc measure` + CNN_AMD_LIB=cnn_amd/lib/libcnn_amd_measure.so; the product library refuses it)
        comm_info["exchange_cost_us_per_step"] = None
        comm_info["exchange_cost_note"] = "needs the measurement build of the library (DP_SKIP_EXCHANGE is not in libcnn_amd.so); exchange_exposed_us is measured by the product library"
    elif comm_info is not None:
        # what the exchange COSTS the step: K more steps with and K without the all-reduce calls (DP_SKIP_EXCHANGE on every rank; everything
        # else -- buckets, events, the 1/N scale -- unchanged), back to back, max over ranks.  Behind the digest check: the replicas diverge here.
        pair = []
        for skip in ("0", "1"):
            capi.set_option("DP_SKIP_EXCHANGE", skip)
            for _ in range(2):
                run["step"]()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run["step"]()
            barrier()
            t = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pair.append(float(t.item()) / args.steps * 1e6)
        capi.set_option("DP_SKIP_EXCHANGE", None)
        comm_info["step_us_with_exchange"], comm_info["step_us_without_exchange"] = round(pair[0], 1), round(pair[1], 1)
        comm_info["exchange_cost_us_per_step"] = round(pair[0] - pair[1], 1)

    out = None
    if rank == 0:
        spec = stacks.STACKS[args.config]()
        flops_img = stacks.train_flops_per_image(spec)
        out = {
            "metric": "images/sec (train step, 224x224x3)",
            "value": round(world * B * args.steps / elapsed, 1),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "spread": {"repeats": len(regions), "regions_of_steps": args.steps, "statistic": "median",
                       "min": round(world * B * args.steps / max(regions), 1), "max": round(world * B * args.steps / min(regions), 1),
                       "clock_warmup_s": CLOCK_WARMUP_S if small else 0.0},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": WORKLOADS[args.config],
                "per_gpu_batch": B,
                "global_batch": world * B,
                "parallelism": f"dp{world}" + (f" (RCCL all-reduce of the {run['n_params']}-float gradient arena per step)" if world > 1 else ""),
                "driver": run["api"],
            },
            "roofline": dict(roofline_entry(dominant, cnt, ms, with_traffic=True),
                             sampled=("every 4th launch" if every == 4 else "every launch") + " in the timed region"),
            "step_tflops": round(flops_img * B * args.steps / elapsed / 1e12, 2),
            "step_frac_of_mfma_peak": round(flops_img * B * args.steps / elapsed / 1e12 / PEAK_MFMA_F32_TFLOPS, 4),
            "final_loss": round(loss, 5),
        }
        if comm_info:
            if rccl_log and os.environ.get("NCCL_DEBUG_FILE") == rccl_log:
                comm_info["rccl_init_log_excerpt"] = rccl_log_excerpt(rccl_log)
            out["exchange"] = comm_info
        if args.breakdown:
            tot = sum(v[1] for v in table.values())
            for k, (c, m) in sorted(table.items(), key=lambda kv: -kv[1][1]):
                r = roofline_entry(k, c, m)
                print(f"{m / c:9.4f} ms x{c / 3:3.0f} {100 * m / tot:5.1f}%  {r['achieved']:>9} {r['unit']:8} {k}", file=sys.stderr)
            print(f"{tot / 3:9.4f} ms kernel total for one step (mean of 3 instrumented steps; side-stream kernels overlap in real steps)", file=sys.stderr)
    run["close"]()
    del run
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not small and out is not None:
        alone = conv_kernel_alone(dominant, torch, capi)  # (the stacks: the dominant kernel's pass alone, beside the in-situ roofline)
        if alone:
            out["roofline"]["alone"] = alone

    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1:
        if small and api == "layer" and not args.no_extra_legs:
            # the same workload (a) with every tensor written by the pass itself (architectures::fuse_pool_block = false: no lazy
            # re-materialisation, plain backward -> SGD sequence) and (b) through the Python driver over the same C ABI
            for key, kw in (("layer_api_all_tensors_written", dict(api="layer", pool_block=False)), ("python_driver", dict(api="pynet"))):
                r2 = make_runner("alexnet", kw["api"], args.batch, torch, capi, 1, 0, None, pool_block=kw.get("pool_block", True))

                def sync():
                    r2["flush"]()
                    torch.cuda.synchronize()

                out[key] = dict(leg_result(r2["B"], args.steps, timed_regions(r2["step"], sync, args.steps, args.warmup, repeats=3)),
                                driver=r2["api"], final_loss=round(r2["loss"](), 5))
                r2["close"]()
                del r2
                torch.cuda.empty_cache()
            out["reference_loop"] = reference_loop_leg(torch, args)
            # NOT the headline: the same train_step with architectures::input_gradient = false -- the delta with respect to the input
            # image (conv2d.cpp:168-199 for conv_layer_1; nothing consumes it, alexnet.cpp:53-55) is not computed, as in every training
            # framework for an input that needs no gradient.  `value` above always computes it, like the reference.
            from cnn_amd import hostapi as _h

            _h.load().cnnh_set_input_gradient(0)
            try:
                r3 = make_runner("alexnet", "layer", args.batch, torch, capi, 1, 0, None)

                def sync3():
                    r3["flush"]()
                    torch.cuda.synchronize()

                out["without_input_gradient"] = dict(leg_result(r3["B"], args.steps, timed_regions(r3["step"], sync3, args.steps, args.warmup, repeats=3)),
                                                     driver=r3["api"] + ", architectures::input_gradient = false (the unobserved d(loss)/d(input image) "
                                                     "of the first layer is skipped; extra leg, never `value`)", final_loss=round(r3["loss"](), 5))
                r3["close"]()
                del r3
            finally:
                _h.load().cnnh_set_input_gradient(1)
            torch.cuda.empty_cache()
        if small and args.staged_input:
            out["pcie_inclusive"] = staged_input_bench(torch, capi, args)
        if small and not args.no_conv_ns:
            out["conv_ns"] = conv_ns_bench(torch, capi)
        if small and not args.no_stacks and args.batch is None:
            # BASELINE configs[3] / [4] in the same driver-run command (a few steps each; their own CPU legs: --config vgg11|resnet18)
            out["stacks"] = {name: stack_leg(name, torch, capi) for name in ("vgg11", "resnet18")}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline() if small else cpu_baseline_stack(args.config)
    if rank == 0:
        print(json.dumps(out))
    if comm is not None:
        comm.destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
