"""Layer lists of the BASELINE.json workloads, in the vocabulary of the reference's container: a strictly sequential
`std::list<std::shared_ptr<Layer>>` (cpu/include/architectures.h:200, iterated at cpu/src/alexnet.cpp:41) of
Conv2D / BatchNorm2D / ReLU / MaxPool2D / LinearLayer.  Pure host data: no device code, no torch.

A spec is a list of tuples
    ("conv", Co, k, stride, pad)   Conv2D(name, Ci, Co, k, stride[, pad])      architectures.h:69 (+ the padding extension)
    ("bn",)                        BatchNorm2D(name, C)                        architectures.h:167
    ("relu",)                      ReLU(name)                                  architectures.h:109
    ("pool", k, step)              MaxPool2D(name, k, step)                    architectures.h:96
    ("dropout", p)                 Dropout(name, p)                            architectures.h:188
    ("linear", out)                LinearLayer(name, C*H*W, out)               architectures.h:131
Channel / spatial sizes follow from the input shape (walk()).
"""


def alexnet(classes=3, batch_norm=False):
    """the reference's own network, alexnet.cpp:10-33"""
    spec = []
    for i, co in enumerate((16, 32, 64, 128)):
        spec.append(("conv", co, 3, 2, 0))
        if batch_norm:
            spec.append(("bn",))
        spec.append(("relu",))
        if i == 0:
            spec.append(("pool", 2, 2))
    spec.append(("linear", classes))
    return spec


def vgg11(classes=3, batch_norm=False):
    """BASELINE configs[3] / SURVEY.md 8(d) config 4: eight 3x3 stride-1 pad-1 convolutions 3->64->128->256->256->512->512->
    512->512, ReLU after each, MaxPool(2,2) after convolutions 1, 2, 4, 6, 8 (224 -> 7), Linear(512*7*7 -> classes)."""
    spec = []
    for i, co in enumerate((64, 128, 256, 256, 512, 512, 512, 512), start=1):
        spec.append(("conv", co, 3, 1, 1))
        if batch_norm:
            spec.append(("bn",))
        spec.append(("relu",))
        if i in (1, 2, 4, 6, 8):
            spec.append(("pool", 2, 2))
    spec.append(("linear", classes))
    return spec


def resnet18(classes=3, batch_norm=True):
    """BASELINE configs[4] / SURVEY.md 8(d) config 5: the convolution SHAPES of ResNet-18 as a strictly sequential list (the
    reference's container has no residual adds, alexnet.cpp:41): 7x7 stride-2 pad-3 stem 3->64 (224 -> 112), MaxPool(2,2)
    (-> 56), then four stages of four convolutions each (64, 128, 256, 512 channels; 3x3 stride-1 pad-1 inside a stage).
    The stage entries carry the down-sampling shapes of the real network: 3x3 stride-2 pad-1 (64->128, 56 -> 28), the 1x1
    stride-2 projection shape (128->256, 28 -> 14) and 3x3 stride-2 pad-1 again (256->512, 14 -> 7).  BatchNorm2D after
    every convolution, ReLU after every BatchNorm2D; Linear(512*7*7 -> classes).  17 convolutions + 1 linear layer."""
    spec = []

    def conv(co, k, s, p):
        spec.append(("conv", co, k, s, p))
        if batch_norm:
            spec.append(("bn",))
        spec.append(("relu",))

    conv(64, 7, 2, 3)
    spec.append(("pool", 2, 2))
    for _ in range(4):
        conv(64, 3, 1, 1)
    conv(128, 3, 2, 1)
    for _ in range(3):
        conv(128, 3, 1, 1)
    conv(256, 1, 2, 0)
    for _ in range(3):
        conv(256, 3, 1, 1)
    conv(512, 3, 2, 1)
    for _ in range(3):
        conv(512, 3, 1, 1)
    spec.append(("linear", classes))
    return spec


STACKS = {"alexnet": alexnet, "vgg11": vgg11, "resnet18": resnet18}
# per-GPU batch of the BASELINE configuration each stack is quoted on (configs[1], [3], [4] = 512 over 8 GPUs)
DEFAULT_BATCH = {"alexnet": 256, "vgg11": 128, "resnet18": 64}


def walk(spec, C=3, H=224, W=224):
    """-> list of dicts, one per layer: kind, input (C, H, W), output (C, H, W), parameter count (checkpoint order:
    conv2d.cpp:220-226 weights then bias; batchnorm2d.cpp:168-173 gamma, beta, moving_mean, moving_var; linear.cpp:105-108)"""
    out = []
    for item in spec:
        kind = item[0]
        ent = {"kind": kind, "in": (C, H, W)}
        if kind == "conv":
            _, co, k, s, p = item
            ent.update(Co=co, k=k, s=s, pad=p, params=co * C * k * k + co)
            C, H, W = co, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        elif kind == "bn":
            ent.update(params=4 * C)
        elif kind == "relu":
            ent.update(params=0)
        elif kind == "dropout":
            ent.update(p=item[1], params=0)
        elif kind == "pool":
            _, k, st = item
            ent.update(k=k, step=st, params=0)
            H, W = (H - k) // st + 1, (W - k) // st + 1
        elif kind == "linear":
            _, n_out = item
            ent.update(n_in=C * H * W, n_out=n_out, params=C * H * W * n_out + n_out)
            C, H, W = n_out, 1, 1
        else:
            raise ValueError(f"unknown layer kind {kind!r}")
        assert H > 0 and W > 0, f"layer {item} has an empty output"
        ent["out"] = (C, H, W)
        out.append(ent)
    return out


def conv_geometries(name, H=224, W=224):
    """(Ci, H, W, Co, k, s, pad) of every convolution of a named stack, in layer order"""
    return [(e["in"][0], e["in"][1], e["in"][2], e["Co"], e["k"], e["s"], e["pad"])
            for e in walk(STACKS[name](), 3, H, W) if e["kind"] == "conv"]


def train_flops_per_image(spec, H=224, W=224):
    """algorithmic FLOPs of one train step per image: 3 x the forward count of SURVEY.md 8(d) (conv 2*Co*Ho*Wo*Ci*k^2, linear
    2*in*out; fwd + dgrad + wgrad)"""
    fl = 0.0
    for e in walk(spec, 3, H, W):
        if e["kind"] == "conv":
            co, ho, wo = e["out"]
            fl += 2.0 * co * ho * wo * e["in"][0] * e["k"] ** 2
        elif e["kind"] == "linear":
            fl += 2.0 * e["n_in"] * e["n_out"]
    return 3.0 * fl


def he_init(layout, seed):
    """seeded synthetic parameters for a walk()-style layout, flat in checkpoint order: N(0, 2/fan_in) filters and matrices
    (keeps activations O(1) through 8-17 layers; the reference's own N(0,1)/10 -- conv2d.cpp:24-29 -- overflows the deep
    stacks), small biases, BatchNorm2D gamma ~ 1, beta ~ 0, moving statistics 0 (batchnorm2d.cpp:18-20).  There is no network
    for checkpoints of these shapes; benches and parity tests inject these values into every implementation alike."""
    import numpy as np

    rs = np.random.RandomState(seed)
    parts = []
    for e in layout:
        kind = e["kind"]
        if kind == "conv":
            ci, co, k = e["in"][0], e["Co"], e["k"]
            parts.append(rs.standard_normal(co * ci * k * k) * np.sqrt(2.0 / (ci * k * k)))
            parts.append(rs.standard_normal(co) * 0.05)
        elif kind == "bn":
            c = e["in"][0]
            parts += [1.0 + 0.1 * rs.standard_normal(c), 0.1 * rs.standard_normal(c), np.zeros(c), np.zeros(c)]
        elif kind == "linear":
            parts.append(rs.standard_normal(e["n_in"] * e["n_out"]) * np.sqrt(1.0 / e["n_in"]))
            parts.append(rs.standard_normal(e["n_out"]) * 0.05)
    return np.concatenate(parts).astype(np.float32) if parts else np.zeros(0, np.float32)
