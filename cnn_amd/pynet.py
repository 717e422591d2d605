"""Python driver of the reference network on the HIP path (plumbing for tests/ and bench.py).

Same layer sequence as the reference's AlexNet container (cpu/src/alexnet.cpp:10-33, batch_norm=false):
    Conv(3->16,k3,s2) ReLU MaxPool(2,2) Conv(16->32) ReLU Conv(32->64) ReLU Conv(64->128) ReLU Linear(4608->classes)
and the same step order as cpu/src/cnn.cpp:79-90 (forward, softmax, cross-entropy, backward, SGD).  Every layer
call goes through the C ABI (cnn_amd.capi); torch only owns the device buffers and the stream.  Parameters and
gradients live in ONE flat arena each, in checkpoint order (alexnet.cpp:69-77), so that a .model file loads with one
copy, the SGD step is one kernel and the data-parallel exchange is one all-reduce.
"""
import numpy as np

from . import capi


class AlexNetHip:
    CHANS = [3, 16, 32, 64, 128]

    def __init__(self, batch, classes=3, H=224, W=224, device="cuda", fuse=True):
        import torch

        self.torch = torch
        # fuse: Conv2D+ReLU forward and MaxPool2D+ReLU backward run as one kernel each (bit-identical results, every
        # layer's output tensor is still written); fuse=False issues the reference's one-call-per-layer sequence
        self.fuse = fuse
        # prepared filters: with fuse=True the per-call filter re-layout kernels (8 per step) are replaced by
        # cnn_conv2d_prepare_filters (2 launches) whenever the parameters changed
        self._prep_valid = False
        self.B, self.classes, self.H, self.W, self.dev = batch, classes, H, W, device
        f32 = dict(dtype=torch.float32, device=device)
        self.convs, self.conv_in_hw, self.conv_out_hw = [], [], []
        h, w = H, W
        off = 0
        self.w_off, self.b_off = [], []
        for l in range(4):
            ci, co = self.CHANS[l], self.CHANS[l + 1]
            self.convs.append(capi.Conv2d(batch, ci, h, w, co, 3, 2, 0, device=device))
            self.conv_in_hw.append((h, w))
            h, w = capi.conv_out_dim(h, 3, 2), capi.conv_out_dim(w, 3, 2)
            self.conv_out_hw.append((h, w))
            self.w_off.append(off); off += co * ci * 9
            self.b_off.append(off); off += co
            if l == 0:
                h, w = capi.pool_out_dim(h, 2, 2), capi.pool_out_dim(w, 2, 2)
        self.pool_hw = self.conv_in_hw[1]
        self.lin_in = 128 * h * w
        self.lw_off = off; off += self.lin_in * classes
        self.lb_off = off; off += classes
        self.n_params = off
        self.params = torch.zeros(off, **f32)
        self.grads = torch.zeros(off, **f32)
        # layer-owned buffers, allocated once (the reference is shape-static too: conv2d.cpp:47-52)
        self.conv_out = [torch.empty((batch, self.CHANS[l + 1]) + self.conv_out_hw[l], **f32) for l in range(4)]
        self.relu_out = [torch.empty_like(t) for t in self.conv_out]
        self.pool_out = torch.empty((batch, 16) + self.pool_hw, **f32)
        self.pool_mask = torch.empty((batch, 16) + self.pool_hw, dtype=torch.int32, device=device)
        self.logits = torch.empty((batch, classes), **f32)
        self.probs = torch.empty((batch, classes), **f32)
        self.delta = torch.empty((batch, classes), **f32)
        self.loss_sum = torch.zeros(1, **f32)
        self.d_lin = torch.empty((batch, self.lin_in), **f32)
        self.d_conv = [torch.empty((batch, self.CHANS[l]) + self.conv_in_hw[l], **f32) for l in range(4)]
        self.d_pool = torch.empty((batch, 16) + self.conv_out_hw[0], **f32)
        self.x = None
        import os

        self.use_prep = fuse and not os.environ.get("CNN_AMD_NO_PREPARED")  # (A/B switch for measurements)
        self.prep = [c.prepared_buffers(device) for c in self.convs] if self.use_prep else None

    # ---- parameter views (reference layouts) ----
    def conv_w(self, l, arena=None):
        a = self.params if arena is None else arena
        ci, co = self.CHANS[l], self.CHANS[l + 1]
        return a[self.w_off[l] : self.w_off[l] + co * ci * 9].view(co, ci, 3, 3)

    def conv_b(self, l, arena=None):
        a = self.params if arena is None else arena
        return a[self.b_off[l] : self.b_off[l] + self.CHANS[l + 1]]

    def lin_w(self, arena=None):
        a = self.params if arena is None else arena
        return a[self.lw_off : self.lw_off + self.lin_in * self.classes].view(self.lin_in, self.classes)

    def lin_b(self, arena=None):
        a = self.params if arena is None else arena
        return a[self.lb_off : self.lb_off + self.classes]

    def load_params(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        assert flat.size == self.n_params, (flat.size, self.n_params)
        self.params.copy_(self.torch.from_numpy(flat))
        self._prep_valid = False

    def _prepare(self):
        if not self._prep_valid:
            capi.prepare_filters(self.convs, [self.conv_w(l) for l in range(4)], [self.conv_b(l) for l in range(4)],
                                 [p[0] for p in self.prep], [p[1] for p in self.prep])
            self._prep_valid = True

    def load_checkpoint(self, path):
        self.load_params(np.fromfile(path, dtype=np.float32))

    # ---- alexnet.cpp:35-46 ----
    def forward(self, x, record=True):
        self.x = x
        cur = x
        if self.use_prep:
            self._prepare()
        for l in range(4):
            if self.fuse and not self.use_prep:
                self.convs[l].forward_relu(cur, self.conv_w(l), self.conv_b(l), self.conv_out[l], self.relu_out[l])
            elif self.fuse:
                self.convs[l].forward_prepared(cur, self.prep[l][0], self.conv_b(l), self.conv_out[l], self.relu_out[l])
            else:
                self.convs[l].forward(cur, self.conv_w(l), self.conv_b(l), self.conv_out[l])
                capi.check(capi.load().cnn_relu_forward(capi._ptr(self.conv_out[l]), capi._ptr(self.relu_out[l]),
                                                        self.conv_out[l].numel(), capi._stream()), "cnn_relu_forward")
            cur = self.relu_out[l]
            if l == 0:
                hh, ww = self.conv_out_hw[0]
                capi.check(capi.load().cnn_maxpool2d_forward(capi._ptr(cur), capi._ptr(self.pool_out),
                                                             capi._ptr(self.pool_mask) if record else None, self.B, 16, hh,
                                                             ww, 2, 2, capi._stream()), "cnn_maxpool2d_forward")
                cur = self.pool_out
        capi.linear_forward(cur.view(self.B, self.lin_in), self.lin_w(), self.lin_b(), self.logits)
        return self.logits

    # ---- func.cpp:16-73 on the device ----
    def loss_backward_seed(self, labels):
        capi.check(capi.load().cnn_softmax_xent(capi._ptr(self.logits), capi._ptr(labels), capi._ptr(self.probs),
                                                capi._ptr(self.delta), capi._ptr(self.loss_sum), self.B, self.classes,
                                                capi._stream()), "cnn_softmax_xent")
        return self.delta

    # ---- alexnet.cpp:49-59; `divisor` is the batch size the gradients are averaged over on THIS rank ----
    def backward(self, delta, divisor=None):
        div = float(self.B if divisor is None else divisor)
        g = self.grads
        capi.linear_backward(self.relu_out[3].view(self.B, self.lin_in), delta, self.lin_w(), div, self.lin_w(g),
                             self.lin_b(g), self.d_lin)
        cur = self.d_lin.view(self.relu_out[3].shape)
        for l in (3, 2, 1, 0):
            if l == 0:
                hh, ww = self.conv_out_hw[0]
                if self.fuse:  # pool backward + relu_layer_1 backward in one pass
                    capi.maxpool_backward_relu(cur, self.pool_mask, self.pool_out, (self.B, 16, hh, ww), 2, 2, self.d_pool)
                else:
                    capi.maxpool_backward(cur, self.pool_mask, (self.B, 16, hh, ww), 2, 2, self.d_pool)
                cur = self.d_pool
            if not (self.fuse and l == 0):
                capi.relu_backward(self.relu_out[l], cur)  # in place on the upstream delta (relu.cpp:37-39)
            lin = self.x if l == 0 else (self.pool_out if l == 1 else self.relu_out[l - 1])
            # Conv2D::backward in one call: weight/bias gradient on the library's side stream, concurrently with dgrad
            if self.use_prep:
                self.convs[l].backward_prepared(lin, cur, self.prep[l][1], div, self.conv_w(l, g), self.conv_b(l, g),
                                                self.d_conv[l], defer_join=True)
            else:
                self.convs[l].backward(lin, cur, self.conv_w(l), div, self.conv_w(l, g), self.conv_b(l, g), self.d_conv[l],
                                       defer_join=True)
            cur = self.d_conv[l]
        capi.side_stream_join()  # all weight gradients are in the arena before SGD / all-reduce read it

    # ---- alexnet.cpp:62-65 (+ the data-parallel mean) ----
    def update(self, lr, grad_scale=1.0):
        capi.sgd_update(self.params, self.grads, lr, grad_scale)
        self._prep_valid = False

    def train_step(self, x, labels, lr, dist=None, world=1):
        """one iteration of cnn.cpp:79-90.  Under data parallelism every rank holds B local samples: local grads are
        (1/B)*sum_local, all-reduce(sum) over `world` ranks, then x(1/world) folded into the SGD kernel."""
        self.forward(x)
        self.loss_backward_seed(labels)
        self.backward(self.delta)
        from .dp import allreduce_grads

        self.update(lr, allreduce_grads(self.grads, dist, world))
