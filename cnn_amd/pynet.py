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

    def __init__(self, batch, classes=3, H=224, W=224, device="cuda", fuse=True, defer_input_grad=False, fuse_pool=False):
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
        self._loss_sum = torch.zeros(1, **f32)
        self.loss_terms = torch.zeros(batch, **f32)  # fused head: per-sample log(p[label]); summed on demand (loss_sum)
        self._loss_from_terms = False
        self.d_lin = torch.empty((batch, self.lin_in), **f32)
        self.d_conv = [torch.empty((batch, self.CHANS[l]) + self.conv_in_hw[l], **f32) for l in range(4)]
        self.d_pool = torch.empty((batch, 16) + self.conv_out_hw[0], **f32)
        self.x = None
        import os

        self.use_prep = fuse and not os.environ.get("CNN_AMD_NO_PREPARED")  # (A/B switch for measurements)
        self._no_fbr = bool(os.environ.get("CNN_AMD_NO_BWD_RELU_FUSION"))  # (likewise)
        self._no_head_fusion = bool(os.environ.get("CNN_AMD_NO_HEAD_FUSION"))
        self._dx0_release = int(os.environ.get("CNN_AMD_DX0_RELEASE", "2"))  # conv layer after whose forward the deferred dgrad starts
        # conv_layer_1 -> relu_layer_1 -> max_pool_1 as one kernel: conv_out[0] / relu_out[0] are then NOT written (nothing in
        # the step reads them: the backward pass of that block works from pool_out + pool_mask)
        self.fuse_pool = bool(fuse_pool) and self.use_prep and self.convs[0].relu_maxpool2_supported()
        # ... and its backward pass rebuilds the convolution delta from (d pool_out, pool_mask, pool_out) inside the conv1
        # weight / data gradient kernels: no MaxPool2D::backward / ReLU::backward kernel, no d_pool tensor.  The deferred
        # data gradient (below) then still needs pool_out / pool_mask of ITS step while the next forward pass is already
        # writing new ones: two sets, alternating.
        self._relu_only = [c.relu_only_supported() for c in self.convs] if self.fuse_pool else [False] * 4
        # the fused block's mask as one byte per window (include/cnn_amd.h, CNN_CONV2D_POOL_MASK_PACKED) where the library's kernels
        # read it: only the block's own calls touch pool_mask in this mode; pool_mask_int32() unpacks it for everybody else
        self.mask_packed = self.fuse_pool and self.convs[0].pool_mask_packed_supported()
        if self.mask_packed:
            self.convs[0].set_pool_mask_packed()
            self.pool_mask = torch.empty(self.convs[0].pool_mask_bytes(), dtype=torch.uint8, device=device)
        self.pool_sets = [(self.pool_out, self.pool_mask)]
        if self.fuse_pool and defer_input_grad:
            self.pool_sets.append((torch.empty_like(self.pool_out), torch.empty_like(self.pool_mask)))
        self.pool_cur = 0
        self.prep = [c.prepared_buffers(device) for c in self.convs] if self.use_prep else None
        # The data gradient of conv_layer_1 (the delta w.r.t. the input image, conv2d.cpp:168-199) has no consumer: nothing
        # waits for it.  It is still computed every step, but as a DEFERRED launch on a second stream that is released
        # in the next forward pass once the HBM-bound layers are through (after conv_layer_2), so that this HBM-bound
        # kernel overlaps the latency/compute-bound layers 3-4 instead of fighting conv1's weight gradient for bandwidth.
        # flush() launches a still-pending one (end of training / before timing ends).  Needs its own copy of the
        # prepared filters of the step it belongs to (two buffers, alternating), because SGD runs in between.
        # Opt-in (defer_input_grad=True, what bench.py uses): callers that read d_conv[0] must flush() first.
        self.defer_dx0 = bool(defer_input_grad) and self.use_prep and not os.environ.get("CNN_AMD_NO_DEFER_DX0")
        if self.defer_dx0:
            self.side_b = torch.cuda.Stream(device=device)
            self.ev_release, self.ev_b_done = torch.cuda.Event(), torch.cuda.Event()
            self.prep0_dgrad = [self.prep[0][1], self.convs[0].prepared_buffers(device)[1]]
            self.parity = 0
            self.pending_dx0 = None   # prepared-filter buffer of the step whose conv1 dgrad has not been launched yet
            self.b_in_flight = False
        # Single-rank steps (train_step without a process group) end without a tail of small launches: the slab reductions, the
        # SGD step and the re-prepared filters of conv_layer_2..4 and the linear layer are queued on the library's side stream as
        # soon as their gradients are complete and run UNDER conv_layer_1's weight-gradient kernel; that kernel is followed by
        # one small launch (reduction + SGD + filter images of conv_layer_1) and the next forward pass starts right behind it.
        # Before: slab_reduce, sgd_vec, pack_batch, rd_prepare one after the other, ~35 us of a 490 us step with the chip idle.
        self.early_update = self.defer_dx0 and self.fuse_pool and not os.environ.get("CNN_AMD_NO_EARLY_UPDATE")
        self._side = None
        self._updated = False
        self._publish = not os.environ.get("CNN_AMD_NO_PUBLISH")  # (A/B switch: plain event records at the forks)
        if self.early_update:
            self.ev_early = torch.cuda.Event()

    @property
    def loss_sum(self):
        """-sum_b log(p[b][label_b]) of the last step (1-element device tensor).  With the fused head the ordered sum over the
        per-sample terms is computed here, i.e. only when somebody asks for the loss value."""
        if self._loss_from_terms:
            capi.check(capi.load().cnn_loss_from_terms(capi._ptr(self.loss_terms), capi._ptr(self._loss_sum), self.B, capi._stream()),
                       "cnn_loss_from_terms")
            self._loss_from_terms = False
        return self._loss_sum

    # ---- parameter views (reference layouts) ----
    def pool_mask_int32(self):
        """max_pool_1's mask in cnn_maxpool2d_forward's form (+ bit 31 of the fused kernel), whatever form the block keeps it in"""
        import torch

        if not self.mask_packed:
            return self.pool_mask
        out = torch.empty((self.B, 16) + self.pool_hw, dtype=torch.int32, device=self.pool_mask.device)
        return self.convs[0].pool_mask_unpack(self.pool_mask, out)

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
            dg = [p[1] for p in self.prep]
            if self.defer_dx0:
                self.parity ^= 1
                dg[0] = self.prep0_dgrad[self.parity]
            capi.prepare_filters(self.convs, [self.conv_w(l) for l in range(4)], [self.conv_b(l) for l in range(4)],
                                 [p[0] for p in self.prep], dg)
            self._prep_valid = True

    def _launch_pending_dx0(self, gated, published=False):
        """conv_layer_1's data gradient of the PREVIOUS backward pass, on the second stream"""
        if not self.defer_dx0 or self.pending_dx0 is None:
            return
        torch = self.torch
        main = torch.cuda.current_stream()
        if published:  # the kernel just launched carries the event (capi.publish_next_kernel): no marker packet on the main stream
            capi.wait_published(self.side_b)
        else:
            self.ev_release.record(main)
        with torch.cuda.stream(self.side_b):
            if not published:
                self.side_b.wait_event(self.ev_release)  # gated: not before this point of the main stream
            if self.fuse_pool:
                prep_dg, dpool, mask, pooled = self.pending_dx0
                self.convs[0].backward_data_pooled2(dpool, mask, pooled, None, self.d_conv[0], prepared_dgrad=prep_dg)
            else:
                self.convs[0].backward_data_prepared(self.d_pool, self.pending_dx0, self.d_conv[0])
            self.ev_b_done.record(self.side_b)
        self.pending_dx0 = None
        self.b_in_flight = True
        if not gated:
            main.wait_event(self.ev_b_done)
            self.b_in_flight = False

    def flush(self):
        """make sure every deferred kernel has been launched and is ordered before later work on the current stream"""
        self._launch_pending_dx0(gated=False)
        if self.defer_dx0 and self.b_in_flight:
            self.torch.cuda.current_stream().wait_event(self.ev_b_done)
            self.b_in_flight = False

    def load_checkpoint(self, path):
        self.load_params(np.fromfile(path, dtype=np.float32))

    # ---- alexnet.cpp:35-46 ----
    def forward(self, x, record=True, labels=None):
        """labels (device int32 [B], optional): with fused kernels the loss head runs inside the last layer's forward kernel
        and loss_backward_seed() becomes a no-op for this pass"""
        self.x = x
        self._head_done = False
        cur = x
        if self.use_prep:
            self._prepare()
        for l in range(4):
            release_here = self.defer_dx0 and l == self._dx0_release and self.pending_dx0 is not None and self._publish
            if release_here:
                capi.publish_next_kernel()  # this layer's forward kernel releases the deferred data gradient
            if l == 0 and self.fuse_pool:
                self.pool_cur = (self.pool_cur + 1) % len(self.pool_sets)
                self.pool_out, self.pool_mask = self.pool_sets[self.pool_cur]
                self.convs[0].relu_maxpool2_forward(cur, None, None, self.pool_out, self.pool_mask if record else None,
                                                    prepared_fwd=self.prep[0][0])
                cur = self.pool_out
                continue
            if self.fuse and not self.use_prep:
                self.convs[l].forward_relu(cur, self.conv_w(l), self.conv_b(l), self.conv_out[l], self.relu_out[l])
            elif self.fuse:
                # (opt-in with fuse_pool: pre-activations nobody reads are not written either)
                y = None if (self.fuse_pool and self._relu_only[l]) else self.conv_out[l]
                self.convs[l].forward_prepared(cur, self.prep[l][0], self.conv_b(l), y, self.relu_out[l])
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
            if l == self._dx0_release:
                # release point of the deferred conv1 data gradient, measured (images/s at batch 256, same box): no
                # deferral 261.0k | before conv1 ~249k | after max_pool_1 ~257k | after conv_layer_2 269.3k | after
                # conv_layer_3 264.5k -- it then overlaps the latency-bound layers 3-4, the linear layer and the loss
                self._launch_pending_dx0(gated=True, published=release_here)
        if labels is not None and self.use_prep and self.classes <= 8 and not self._no_head_fusion:
            capi.check(capi.load().cnn_linear_forward_softmax_xent(capi._ptr(cur), capi._ptr(self.lin_w()), capi._ptr(self.lin_b()),
                                                                   capi._ptr(labels), capi._ptr(self.logits), capi._ptr(self.probs),
                                                                   capi._ptr(self.delta), capi._ptr(self.loss_terms), self.B,
                                                                   self.lin_in, self.classes, capi._stream()),
                       "cnn_linear_forward_softmax_xent")
            self._head_done = True
            self._loss_from_terms = True
            return self.logits
        capi.linear_forward(cur.view(self.B, self.lin_in), self.lin_w(), self.lin_b(), self.logits)
        return self.logits

    # ---- func.cpp:16-73 on the device ----
    def loss_backward_seed(self, labels):
        if getattr(self, "_head_done", False):  # done by forward(labels=...)
            return self.delta
        self._loss_from_terms = False
        capi.check(capi.load().cnn_softmax_xent(capi._ptr(self.logits), capi._ptr(labels), capi._ptr(self.probs),
                                                capi._ptr(self.delta), capi._ptr(self._loss_sum), self.B, self.classes,
                                                capi._stream()), "cnn_softmax_xent")
        return self.delta

    # ---- alexnet.cpp:49-59; `divisor` is the batch size the gradients are averaged over on THIS rank ----
    def _early_update(self, lr, scale):
        """side stream: reductions, SGD and filter images of everything behind conv_layer_1 (see __init__)"""
        torch = self.torch
        if self._side is None:
            self._side = capi.side_stream()
        # behind their readers (the data gradients of layers 2-4): conv_layer_2's data gradient is the published kernel
        if self._publish:
            capi.wait_published(self._side)
        else:
            self.ev_early.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            if not self._publish:
                self._side.wait_event(self.ev_early)
            capi.flush_reduces()
            lo = self.w_off[1]
            capi.sgd_update(self.params[lo:], self.grads[lo:], lr, scale)
            capi.prepare_filters(self.convs[1:], [self.conv_w(l) for l in (1, 2, 3)], [self.conv_b(l) for l in (1, 2, 3)],
                                 [p[0] for p in self.prep[1:]], [p[1] for p in self.prep[1:]])

    def backward(self, delta, divisor=None, sgd=None):
        """sgd = (lr, grad_scale): the single-rank train step -- parameters are updated and filters re-prepared inside the pass"""
        div = float(self.B if divisor is None else divisor)
        g = self.grads
        # fuse_bwd_relu: every ReLU::backward runs inside the kernel that PRODUCES its delta (the linear backward for
        # relu_layer_4, the data gradient of conv_layer_{l+1} for relu_layer_l)
        fbr = self.use_prep and not self._no_fbr
        pub = self._publish and self.use_prep
        if pub:
            capi.publish_next_kernel()  # (each data-gradient kernel is the fork point of the next layer's weight gradient)
        capi.linear_backward(self.relu_out[3].view(self.B, self.lin_in), delta, self.lin_w(), div, self.lin_w(g),
                             self.lin_b(g), self.d_lin, relu_below=fbr)
        cur = self.d_lin.view(self.relu_out[3].shape)
        if self._dx0_release >= 4:  # (tuning) release the deferred conv1 data gradient here: it overlaps conv 4-3 data gradients
            self._launch_pending_dx0(gated=True)
        for l in (3, 2, 1, 0):
            if l == 1 and self.fuse_pool and self.defer_dx0:
                self.flush()  # the previous step's deferred dgrad reads d_conv[1] (= d pool_out), which is rewritten next
            if l == 0 and self.fuse_pool:
                # conv_layer_1 from the pooled domain: cur = d pool_out
                # (the fused forward kernel marked the windows with pooled <= 0 in the mask itself, bit 31: relu_layer_1's backward
                # pass needs no tensor of its own -- pooled = None, and conv_layer_2's data gradient stays unmasked)
                pooled = None
                if self.defer_dx0 and sgd is not None and self.early_update:
                    self._early_update(*sgd)
                    nxt = self.parity ^ 1  # (the deferred data gradient of THIS step still reads the filters of this step)
                    self.convs[0].backward_weight_pooled2_sgd(self.x, cur, self.pool_mask, pooled, div, self.conv_w(0, g), self.conv_b(0, g),
                                                              self.conv_w(0), self.conv_b(0), sgd[0], sgd[1], self.prep[0][0],
                                                              self.prep0_dgrad[nxt])
                    self.pending_dx0 = (self.prep0_dgrad[self.parity], cur, self.pool_mask, pooled)
                    self.parity = nxt
                    self._updated = True
                    self._prep_valid = True
                elif self.defer_dx0:
                    self.convs[0].backward_weight_pooled2(self.x, cur, self.pool_mask, pooled, div, self.conv_w(0, g), self.conv_b(0, g))
                    self.pending_dx0 = (self.prep0_dgrad[self.parity], cur, self.pool_mask, pooled)
                else:  # both gradients now, concurrently (weight gradient on the library's side stream)
                    self.convs[0].backward_pooled2_prepared(self.x, cur, self.pool_mask, pooled, self.prep[0][1], div,
                                                            self.conv_w(0, g), self.conv_b(0, g), self.d_conv[0], defer_join=True)
                break
            if l == 0:
                hh, ww = self.conv_out_hw[0]
                if self.defer_dx0:
                    self.flush()  # the previous step's deferred dgrad reads d_pool, which is rewritten next
                if self.fuse:  # pool backward + relu_layer_1 backward in one pass
                    capi.maxpool_backward_relu(cur, self.pool_mask, self.pool_out, (self.B, 16, hh, ww), 2, 2, self.d_pool)
                else:
                    capi.maxpool_backward(cur, self.pool_mask, (self.B, 16, hh, ww), 2, 2, self.d_pool)
                cur = self.d_pool
            if not (self.fuse and l == 0) and not fbr:
                capi.relu_backward(self.relu_out[l], cur)  # in place on the upstream delta (relu.cpp:37-39)
            lin = self.x if l == 0 else (self.pool_out if l == 1 else self.relu_out[l - 1])
            # Conv2D::backward in one call: weight/bias gradient on the library's side stream, concurrently with dgrad
            if self.defer_dx0 and l == 0:
                # weight / bias gradient now (SGD needs it); the data gradient is launched in the next forward pass
                self.convs[0].backward_weight(lin, cur, div, self.conv_w(0, g), self.conv_b(0, g))
                self.pending_dx0 = self.prep0_dgrad[self.parity]
            elif self.use_prep:
                if pub:
                    capi.publish_next_kernel()
                # lin is relu_out[l-1] for l >= 2: its ReLU::backward is fused into this data gradient; for l == 1 in a pool-fused
                # net lin = pool_out, and masking d(pool_out) by (pool_out <= 0) IS relu_layer_1's backward pass in the pooled
                # domain (at an argmax position the ReLU output equals the pooled value)
                self.convs[l].backward_prepared(lin, cur, self.prep[l][1], div, self.conv_w(l, g), self.conv_b(l, g),
                                                self.d_conv[l], defer_join=True,
                                                relu_below=lin if (fbr and l >= 2) else None)
            else:
                self.convs[l].backward(lin, cur, self.conv_w(l), div, self.conv_w(l, g), self.conv_b(l, g), self.d_conv[l],
                                       defer_join=True)
            cur = self.d_conv[l]
        capi.side_stream_join()  # all weight gradients are in the arena before SGD / all-reduce read it

    # ---- alexnet.cpp:62-65 (+ the data-parallel mean) ----
    def update(self, lr, grad_scale=1.0):
        if self._updated:  # backward(sgd=...) has done it
            self._updated = False
            return
        capi.sgd_update(self.params, self.grads, lr, grad_scale)
        self._prep_valid = False

    def train_step(self, x, labels, lr, dist=None, world=1):
        """one iteration of cnn.cpp:79-90.  Under data parallelism every rank holds B local samples: local grads are
        (1/B)*sum_local, all-reduce(sum) over `world` ranks, then x(1/world) folded into the SGD kernel."""
        self.forward(x, labels=labels)
        self.loss_backward_seed(labels)
        from .dp import allreduce_grads

        if dist is None and world == 1 and self.early_update and self._prep_valid:
            self.backward(self.delta, sgd=(lr, 1.0))
        else:
            self.backward(self.delta)
        self.update(lr, allreduce_grads(self.grads, dist, world))
