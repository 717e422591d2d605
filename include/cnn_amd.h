/*
 * cnn_amd.h -- C ABI of libcnn_amd.so: the MI355X (gfx950) replacement for the arithmetic inside
 * hermosayhl/CNN's Layer::forward / Layer::backward / update_gradients (cpu/include/architectures.h:34-138).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (int32 for the pool mask) unless it says "host";
 *     a batch is contiguous NCHW: sample b of the reference's std::vector<tensor> (data_format.h:53) lives
 *     at base + b*C*H*W, each sample in the reference's own CHW order (data_format.h:11-18);
 *   - weight layouts are the reference's: conv [Co][Ci][k][k] then bias [Co] (conv2d.cpp:18-21,220-226),
 *     linear [in][out] then bias [out] (linear.cpp:40,105-108) -- a .model checkpoint streams straight in;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call only ENQUEUES work;
 *   - no allocation inside: scratch comes from the caller (`ws`, sized by the *_workspace_bytes query);
 *   - return value: 0 = ok, non-zero = CNN_AMD_E_* (or 1000 + hipError_t); cnn_amd_last_error() gives text.
 *     The reference has no error channel (assert only), so the host layer classes abort on non-zero.
 */
#ifndef CNN_AMD_H
#define CNN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNN_AMD_ABI_VERSION 2

enum {
    CNN_AMD_OK = 0,
    CNN_AMD_E_BADARG = 1,    /* null pointer / non-positive dimension / unsupported geometry */
    CNN_AMD_E_WORKSPACE = 2, /* ws too small for this call */
    CNN_AMD_E_HIP = 1000,    /* 1000 + hipError_t */
    CNN_AMD_E_COMM = 2000    /* 2000 + ncclResult_t (RCCL); 2000 itself = librccl could not be loaded */
};

int cnn_amd_abi_version(void);
const char* cnn_amd_last_error(void);
/* "gfx950" when a device is present and matches, otherwise an explanatory string; never throws */
const char* cnn_amd_device_arch(void);

/* ---- measurement switches: the A/B switches of DESIGN.md section 10 (kernel-family choices, tile overrides, debug ablations).
 * The library reads the CNN_AMD_* variables of the environment ONCE, when it is first used; afterwards this is the only way to
 * change one (value == NULL removes it).  `name` with or without the "CNN_AMD_" prefix.  Not needed for normal use: the defaults
 * are the product path.  cnn_amd_get_option returns 0 and copies the value when the switch is set, 1 when it is not. */
int cnn_amd_set_option(const char* name, const char* value);
int cnn_amd_get_option(const char* name, char* value_out, size_t cap);
/* Switches that change RESULTS or exist for timing experiments only (a kernel without its stores / DMAs / MFMAs, per-phase cycle
 * printers, an LDS-request override, a data-parallel step without its exchange: DBG, *_DBG, ROWS_LDS, DP_SKIP_EXCHANGE) are compiled
 * out of libcnn_amd.so: cnn_amd_set_option() returns CNN_AMD_E_BADARG for them and the environment is not consulted.  They exist in
 * the measurement build only (make -C cnn_amd/csrc measure -> libcnn_amd_measure.so, loaded by tools/ through CNN_AMD_LIB).
 * cnn_amd_measure_build(): 1 in that build, 0 in the product library. */
int cnn_amd_measure_build(void);

/* ---- measurement: per-kernel durations from HIP events recorded on the launch stream --------------------- */
/* mode 0 = off, 1 = time every kernel launch, 2 = only launches whose "<kernel>|<geometry>" key contains filter */
int cnn_amd_kernel_timing_enable(int mode, const char* filter);
/* mode 2 only: bracket every `every`-th matching launch instead of each one (the two event records around a kernel leave
 * ~6 us bubbles on its stream); the report then counts the sampled launches. */
int cnn_amd_kernel_timing_sampling(int every);
/* device-synchronises; writes "<kernel>|<geometry>\t<launches>\t<total_ms>\n" lines (host buffer) and clears
 * the records; returns bytes needed (records are kept when cap is too small), -1 on error */
long long cnn_amd_kernel_timing_report(char* buf, size_t cap);
/* brackets a span of `stream` that holds NO library kernel -- e.g. the wait for another stream -- with the kernel timer's events: when
 * timing is on it appears in the report under "<name>|span" (give such names the prefix "span:" so that consumers can tell them from
 * kernels); no-ops otherwise.  The host container uses it for "span:exchange_wait": how long the compute stream waits for the
 * gradient exchange at the end of a data-parallel step (the EXPOSED part of the all-reduce). */
int cnn_amd_timing_span_begin(void* stream, const char* name);
int cnn_amd_timing_span_end(void* stream);

/* ---- geometry helpers (host side, pure) -------------------------------------------------------------- */
/* conv2d.cpp:41-42:  out = (H + 2*pad - k)/s + 1, integer division (the reference has pad == 0) */
int cnn_conv2d_out_dim(int in, int k, int s, int pad);
/* pool2d.cpp:14-15:  out = (H - k)/step + 1 */
int cnn_maxpool2d_out_dim(int in, int k, int step);

/* ---- Conv2D : conv2d.cpp --------------------------------------------------------------------------- */
typedef struct {
    int B, Ci, H, W; /* input  batch / channels / height / width */
    int Co, k, s;    /* filters, kernel edge (any k >= 1; the reference asserts odd >= 3), stride */
    int pad;         /* zero padding on each side (extension; the reference is fixed at 0, architectures.h:59) */
    int flags;       /* 0, or CNN_CONV2D_POOL_MASK_PACKED (ABI version 2; only the fused Conv2D -> ReLU -> MaxPool2D family reads it) */
} cnn_conv2d_desc;
/* flags bit: the pool mask the fused first-block entry points (cnn_conv2d_relu_maxpool2_forward*, cnn_conv2d_backward_*pooled2*)
 * write and read is ONE BYTE per window instead of an int32 flat index -- see cnn_conv2d_pool_mask_packed_supported below */
#define CNN_CONV2D_POOL_MASK_PACKED 1

/* scratch needed by ANY of the three MFMA entry points below for this geometry */
size_t cnn_conv2d_workspace_bytes(const cnn_conv2d_desc* d);

/* Optional, once per geometry and process: times a short list of implicit-GEMM tile shapes for the forward and the data-gradient
 * kernels of this geometry on the device (scratch buffers of its own, zeros; SYNCHRONISES) and pins the fastest for every
 * later call with the same desc -- the built-in rules were tuned on the reference net's layers, other geometries gain up to
 * 2.6x (tools/sweep_igemm.py).  Call it BEFORE cnn_conv2d_prepare_filters / the first forward of the layer (the re-arranged
 * filter image depends on the tile); geometries served by the specialised kernels are left alone.  CNN_AMD_IGEMM_AUTOTUNE=0
 * turns it into a no-op.  The host Conv2D layer calls it when it first sees its input shape. */
int cnn_conv2d_autotune(const cnn_conv2d_desc* d, void* stream);
/* The same measurement with the CALLER's scratch (the header's "no allocation inside" convention; cnn_conv2d_autotune above is the
 * older form that borrows device memory of its own for the duration of the call): scratch of cnn_conv2d_autotune_workspace_bytes(d)
 * bytes (the geometry's input + output + filters + the kernels' workspace).  The host Conv2D layer uses this form. */
size_t cnn_conv2d_autotune_workspace_bytes(const cnn_conv2d_desc* d);
int cnn_conv2d_autotune_ws(const cnn_conv2d_desc* d, void* scratch, size_t scratch_bytes, void* stream);
/* Measured choices are per process; replicas of a data-parallel job should run the SAME kernels (their reduced gradients are identical
 * either way, their local activations then are too).  export: what this process pinned for the geometry, four integers --
 * [0] forward tile, [1] data-gradient tile (-1 = the rule-based default measured best), [2] / [3] = 1 when the implicit GEMM replaces
 * the register-direct forward / data-gradient kernel; CNN_TUNE_NONE = never measured.  import: pins them in another process (entries
 * equal to CNN_TUNE_NONE leave its table alone).  Ship them with cnn_comm_broadcast; the host container does (Sequential::set_comm). */
#define CNN_TUNE_NONE (-2147483647 - 1)
int cnn_conv2d_tune_export(const cnn_conv2d_desc* d, int32_t out[4]);
int cnn_conv2d_tune_import(const cnn_conv2d_desc* d, const int32_t in[4]);

/* replaces Conv2D::forward's loop nest (conv2d.cpp:69-92): y = bias + valid cross-correlation.
 * Implicit GEMM on v_mfma_f32_32x32x2_f32 / 16x16x4_f32, input rows + filter slabs staged in LDS. */
int cnn_conv2d_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                       void* ws, size_t ws_bytes, void* stream);
/* Conv2D::forward immediately followed by ReLU::forward (relu.cpp:21-26) in ONE kernel: y as above AND
 * y_relu = (y >= 0 ? y : 0), both NCHW tensors are written (the convolution output stays observable, alexnet.cpp:97);
 * saves the ReLU kernel's re-read of y.  Bit-identical to cnn_conv2d_forward + cnn_relu_forward. */
int cnn_conv2d_forward_relu(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                            float* y_relu, void* workspace, size_t workspace_bytes, void* stream);
/* != 0: cnn_conv2d_forward_relu / cnn_conv2d_forward_prepared accept y == NULL for this layer and then write ONLY y_relu
 * (a training step never reads the pre-activation tensor: ReLU::backward masks by its own output, relu.cpp:35-40). */
int cnn_conv2d_relu_only_supported(const cnn_conv2d_desc* d);

/* The reference's first block Conv2D -> ReLU -> MaxPool2D(2, 2) (alexnet.cpp:12-15; conv2d.cpp:69-92, relu.cpp:21-26,
 * pool2d.cpp:53-87) in ONE kernel: pooled [B][Co][Ho/2][Wo/2] and mask (same meaning as cnn_maxpool2d_forward: flat index
 * into the sample's Co*Ho*Wo of the window's first maximum; may be NULL) are bit-identical to cnn_conv2d_forward_relu +
 * cnn_maxpool2d_forward, but the convolution / ReLU outputs are NOT materialised -- nothing downstream needs them: the
 * backward pass takes pooled + mask (cnn_maxpool2d_backward_relu, cnn_conv2d_backward_*_pooled2).
 * Bit 31 of a mask entry is additionally SET when that window's pooled value is <= 0, i.e. when the block's ReLU::backward
 * (relu.cpp:37) blocks the window's delta: the pooled-domain gradient entry points below compare the entry with the flat index they
 * expect and therefore skip such windows by themselves (pass pooled = NULL: no tensor is read for the ReLU mask anywhere in the
 * backward pass); cnn_maxpool2d_backward(_relu) ignore the bit.  (mask & 0x7fffffff) is cnn_maxpool2d_forward's mask, bit for bit.
 * Covered geometry: cnn_conv2d_relu_maxpool2_supported(d) != 0 (the thin 3 -> 16 channel 3x3 stride-2 layer). */
int cnn_conv2d_relu_maxpool2_supported(const cnn_conv2d_desc* d);
/* The packed pool mask (d->flags & CNN_CONV2D_POOL_MASK_PACKED).  A 2x2 window's first maximum is one of FOUR pixels, so the int32
 * flat index carries two bits of information in four bytes -- and the three kernels of the block (forward, weight gradient, data
 * gradient) move it through HBM once each: 3 x 49.6 MB per step of the reference net at batch 256.  With the flag set the mask
 * argument of the family is an opaque buffer of cnn_conv2d_pool_mask_bytes(d) bytes (device, 4-byte aligned) holding
 *   byte[((b * Co + co) * (Ho/2) + ph) * pitch + pw],  pitch = ((Wo/2) + 3) & ~3,
 *   bits 0..1 = 2 * (row of the maximum inside the window) + (its column), bit 7 = the int32 form's bit 31 (pooled value <= 0),
 * written by cnn_conv2d_relu_maxpool2_forward(_prepared) and read by the cnn_conv2d_backward_*pooled2* calls with the SAME desc
 * (pooled must then be NULL: bit 7 carries the block's ReLU mask).  cnn_conv2d_pool_mask_unpack converts it to the int32 form
 * (bit 31 included) for anybody else -- cnn_maxpool2d_backward(_relu) read only that one.  Results of every call of the family are
 * bit-identical between the two forms.  supported(d) != 0: the geometry is covered AND the kernels that read the packed form are
 * the ones selected (measurement switches that select the older kernels turn it off); a call with the flag set otherwise fails
 * with CNN_AMD_E_BADARG. */
int cnn_conv2d_pool_mask_packed_supported(const cnn_conv2d_desc* d);
size_t cnn_conv2d_pool_mask_bytes(const cnn_conv2d_desc* d); /* flag clear: B*Co*(Ho/2)*(Wo/2)*4; set: B*Co*(Ho/2)*pitch + 64 */
int cnn_conv2d_pool_mask_unpack(const cnn_conv2d_desc* d, const void* packed, int32_t* mask, void* stream);
int cnn_conv2d_relu_maxpool2_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias,
                                     float* pooled, int32_t* mask, void* workspace, size_t workspace_bytes, void* stream);

/* Conv2D::backward of that block when the delta of the convolution output exists only in the pooled domain:
 *   dy = ReLU::backward(MaxPool2D::backward(dpool))   (relu.cpp:35-40, pool2d.cpp:96-107)
 * is rebuilt on the fly from dpool (delta of the pool output), mask and pooled (both as written by the forward call), so
 * the Co*Ho*Wo delta tensor is never written or read.  Results are bit-identical to cnn_maxpool2d_backward_relu followed by
 * cnn_conv2d_backward_weight / cnn_conv2d_backward_data.  pooled == NULL: dpool already carries the ReLU mask
 * ((pooled <= 0) ? 0 : dpool, e.g. from cnn_conv2d_backward_data_relu(..., relu_below = pooled, ...) of the next layer). */
int cnn_conv2d_backward_weight_pooled2(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask,
                                       const float* pooled, float* gw, float* gb, float divisor, void* workspace,
                                       size_t workspace_bytes, void* stream);
int cnn_conv2d_backward_data_pooled2(const cnn_conv2d_desc* d, const float* dpool, const int32_t* mask, const float* pooled,
                                     const float* w, float* dx, void* workspace, size_t workspace_bytes, void* stream);
/* cnn_conv2d_backward_weight_pooled2 followed -- in ONE more launch instead of three -- by this layer's share of the end of a
 * single-rank train step: gw/gb as above, then w/bias -= lr * (grad_scale * gw/gb) (cnn_sgd_update's arithmetic, alexnet.cpp:62-65
 * -> conv2d.cpp update_gradients) and the layer's prepared filters (cnn_conv2d_prepare_filters images of the UPDATED w/bias;
 * either may be NULL).  Bit-identical to the three separate calls.  workspace as for cnn_conv2d_backward_weight_pooled2. */
int cnn_conv2d_backward_weight_pooled2_sgd(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask,
                                           const float* pooled, float* gw, float* gb, float divisor, float* w, float* bias, float lr,
                                           float grad_scale, void* fwd_prepared, void* dgrad_prepared, void* workspace,
                                           size_t workspace_bytes, void* stream);

/* ... and additionally stores the filters / biases as they were BEFORE the step into w_previous [Co][Ci*k*k] / bias_previous [Co]
 * (either may be NULL).  The host container keeps that snapshot so that Layer::get_output() of a layer whose output tensor was fused
 * away can still be re-computed with the parameters the last forward pass used (alexnet.cpp:97,105), after the step has moved them. */
int cnn_conv2d_backward_weight_pooled2_sgd_keep(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask,
                                                const float* pooled, float* gw, float* gb, float divisor, float* w, float* bias, float lr,
                                                float grad_scale, void* fwd_prepared, void* dgrad_prepared, float* w_previous,
                                                float* bias_previous, void* workspace, size_t workspace_bytes, void* stream);

/* replaces conv2d.cpp:117-159: gw = (sum_b sum_pq dy*x)/divisor, gb = (sum_b sum_pq dy)/divisor.
 * The reference divides by the batch size per sample and accumulates (:148,:157); pass divisor = B of the
 * WHOLE batch (per-rank shard size under data parallelism, see cnn_sgd_update).  gb may be NULL.
 * Split-K over pixels with a deterministic second-stage reduction (no atomics). */
int cnn_conv2d_backward_weight(const cnn_conv2d_desc* d, const float* x, const float* dy, float* gw, float* gb,
                               float divisor, void* ws, size_t ws_bytes, void* stream);

/* replaces conv2d.cpp:168-199 (zero fill + scatter-add) by the equivalent gather: transposed implicit GEMM,
 * stride handled by output-parity classes; input rows/cols no window covers come out 0 like the reference. */
int cnn_conv2d_backward_data(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws,
                             size_t ws_bytes, void* stream);
/* Conv2D data gradient followed by the ReLU::backward (relu.cpp:35-40) of the ReLU layer in front of this convolution:
 * relu_below = that layer's forward output (same shape as dx); dx = (relu_below <= 0 ? 0 : dx), applied in the kernel's
 * store epilogue.  Bit-identical to cnn_conv2d_backward_data + cnn_relu_backward(relu_below, dx). */
int cnn_conv2d_backward_data_relu(const cnn_conv2d_desc* d, const float* dy, const float* w, const float* relu_below,
                                  float* dx, void* workspace, size_t workspace_bytes, void* stream);

/* Conv2D::backward (conv2d.cpp:97-202) in one call: the weight/bias-gradient kernels run on an internal side stream
 * CONCURRENTLY with the data-gradient kernels (fork/join by events on `stream`; hipGraph-capturable).  Same results as
 * the two separate calls.  gb may be NULL. */
size_t cnn_conv2d_backward_workspace_bytes(const cnn_conv2d_desc* d);
/* defer_join = 0: gw/gb/dx are all ordered on `stream` when the call returns.  defer_join = 1: only dx is; gw/gb (and this
 * call's half of ws) belong to the side stream until cnn_amd_side_stream_join(stream) -- lets the weight gradients of
 * layer L overlap the whole backward of layers L-1.. (the caller joins once, before it reads the gradient arena).  With
 * defer_join = 1 the final reduction of the layer's partial gradient slabs (held in ws) is only recorded; the join launches
 * the recorded reductions of the calling thread in one batch (join from the thread that made the calls): ws must not be
 * reused, and gw/gb are undefined, until then. */
int cnn_conv2d_backward(const cnn_conv2d_desc* d, const float* x, const float* dy, const float* w, float* gw, float* gb,
                        float* dx, float divisor, void* ws, size_t ws_bytes, void* stream, int defer_join);
/* Scope of the hidden state behind defer_join (VERDICT r03 asked): the side stream, its fork / join events and the list of recorded
 * slab reductions exist once per (host thread, device), NOT per caller stream.  A thread that forks weight gradients off two different
 * streams of one device gets them serialised on the one side stream; cnn_amd_side_stream_join(S) launches EVERY recorded reduction of the
 * thread (also those forked off another stream) and makes S wait for the side stream as a whole -- conservative, never too weak: a later
 * join from the other stream finds nothing left to launch and waits for the same work.  Two host threads never share any of it. */
int cnn_amd_side_stream_join(void* stream);
/* The weight / bias-gradient half of cnn_conv2d_backward(defer_join = 1) alone: the kernels run on the side stream, forked off `stream`
 * where the call is made (x and dy must be ordered on `stream` by then); gw / gb / ws belong to the side stream until
 * cnn_amd_side_stream_join(stream).  For callers that obtain the data gradients of several layers another way
 * (e.g. one fused kernel of their own).  workspace: cnn_conv2d_workspace_bytes(d). */
int cnn_conv2d_backward_weight_side(const cnn_conv2d_desc* d, const float* x, const float* dy, float* gw, float* gb, float divisor,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* For callers that schedule more work behind the deferred weight gradients (e.g. the SGD step and the re-prepared filters of the
 * layers whose gradients are complete, while the first layers' backward kernels still run on `stream`):
 *   cnn_amd_side_stream_get   the side stream (hipStream_t) of the calling thread on the current device;
 *   cnn_amd_flush_reduces     launches the recorded slab reductions of the calling thread on `stream` NOW -- normally the side
 *                             stream -- instead of at the join (gw/gb of those layers are then ordered on that stream). */
int cnn_amd_side_stream_get(void** side_stream);
int cnn_amd_flush_reduces(void* stream);
/* Cross-stream dependencies without a marker packet on the producing stream (an event recorded BETWEEN two kernels costs that
 * stream ~4.5 us on this hardware; an event carried by the producing kernel's own dispatch ~1.7 us):
 *   cnn_amd_publish_next_kernel(stream)  the next kernel this thread launches on `stream` through the library publishes its
 *                                        completion (forward, data-gradient and linear-backward kernels of the reference net
 *                                        carry it in their dispatch packet; any other kernel is followed by a plain event record);
 *   cnn_amd_wait_published(other)        `other` waits for that kernel = cnn_event_record + cnn_stream_wait_event.
 * cnn_conv2d_backward*() use the published kernel as their fork point when it is the last thing the library launched on
 * `stream`: the caller asserts that nothing else the weight gradient depends on was queued on that stream behind it. */
int cnn_amd_publish_next_kernel(void* stream);
int cnn_amd_wait_published(void* stream);
/* != 0: a published kernel exists on `stream` and it is the LAST thing this thread queued on that stream through the library
 * (kernels, copies, waits, collectives) -- i.e. cnn_amd_wait_published(other) is equivalent to record + wait right now. */
int cnn_amd_published_is_last(void* stream);

/* im2col + plain tiled GEMM: functional fallback kept ONLY for parity checks of the three calls above */
size_t cnn_conv2d_im2col_workspace_bytes(const cnn_conv2d_desc* d);
int cnn_conv2d_forward_im2col(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias,
                              float* y, void* ws, size_t ws_bytes, void* stream);
int cnn_conv2d_backward_weight_im2col(const cnn_conv2d_desc* d, const float* x, const float* dy, float* gw,
                                      float* gb, float divisor, void* ws, size_t ws_bytes, void* stream);
int cnn_conv2d_backward_data_im2col(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx,
                                    void* ws, size_t ws_bytes, void* stream);

/* ---- MaxPool2D : pool2d.cpp ------------------------------------------------------------------------ */
/* replaces pool2d.cpp:53-87. mask (may be NULL = the no_grad path, :41,:61,:79) receives the int32 flat index
 * into the sample's C*H*W of each window's FIRST maximum (strict '<', :71).  Bit-exact vs the reference. */
int cnn_maxpool2d_forward(const float* x, float* y, int32_t* mask, int B, int C, int H, int W, int k, int step,
                          void* stream);
/* replaces pool2d.cpp:96-107: dx = 0; dx[mask[i]] = dy[i] (assignment; with overlapping windows the highest
 * output index wins, reproduced deterministically). */
int cnn_maxpool2d_backward(const float* dy, const int32_t* mask, float* dx, int B, int C, int H, int W, int k,
                           int step, void* stream);
/* MaxPool2D::backward immediately followed by the ReLU::backward (relu.cpp:35-40) of the ReLU layer whose output was
 * this pool's input, in ONE kernel.  pooled = the pool's forward output y: at an argmax position the ReLU output equals
 * pooled[window], everywhere else the delta is 0 either way, so dx[argmax] = (pooled <= 0 ? 0 : dy) reproduces
 * cnn_maxpool2d_backward + cnn_relu_backward bit for bit without reading the ReLU output tensor. */
int cnn_maxpool2d_backward_relu(const float* dy, const int32_t* mask, const float* pooled, float* dx, int B, int C, int H,
                                int W, int k, int step, void* stream);

/* ---- ReLU : relu.cpp ------------------------------------------------------------------------------- */
/* relu.cpp:21-26: y = x >= 0 ? x : 0   (keeps -0.0, NaN -> 0) */
int cnn_relu_forward(const float* x, float* y, size_t n, void* stream);
/* relu.cpp:35-40: dy = (y <= 0) ? 0 : dy, IN PLACE on the caller's delta */
int cnn_relu_backward(const float* y, float* dy_inout, size_t n, void* stream);

/* ---- Dropout : dropout.cpp (row n4 of SURVEY.md 8f; the reference keeps it out of its net, alexnet.cpp:28) -------------
 * Channel dropout as the reference implements it: training (dropout.cpp:34-41) zeroes channels 0 .. dropped_channels-1 of every
 * sample (dropped_channels = int(p * C), :18; the shuffled sequence never reaches the data path) and copies the rest; under
 * no_grad (:44-53) y = x * keep with keep = 1 - p.  backward (:57-69) zeroes the same channels of the delta IN PLACE. */
int cnn_dropout_forward(const float* x, float* y, int B, int C, int H, int W, int dropped_channels, int training, float keep,
                        void* stream);
int cnn_dropout_backward(float* dy_inout, int B, int C, int H, int W, int dropped_channels, void* stream);

/* ---- LinearLayer : linear.cpp ---------------------------------------------------------------------- */
/* linear.cpp:33-43: y[b][j] = (sum_i x[b][i]*W[i*out+j]) + bias[j] */
int cnn_linear_forward(const float* x, const float* w, const float* bias, float* y, int B, int in, int out,
                       void* stream);
/* linear.cpp:56-90: gW = (x^T dy)/divisor (assigned), gb = (sum_b dy)/divisor, dx = dy W^T; any output may be NULL */
int cnn_linear_backward(const float* x, const float* dy, const float* w, float* gw, float* gb, float* dx, int B,
                        int in, int out, float divisor, void* stream);
/* Linear::backward followed by the ReLU::backward (relu.cpp:35-40) of the ReLU layer whose OUTPUT is this layer's input x:
 * dx = (x <= 0 ? 0 : dx).  Bit-identical to cnn_linear_backward + cnn_relu_backward(x, dx). */
int cnn_linear_backward_relu(const float* x, const float* dy, const float* w, float* gw, float* gb, float* dx, int B,
                             int in, int out, float divisor, void* stream);

/* ---- optional: filter preparation hoisted out of the per-layer calls ----------------------------------
 * cnn_conv2d_forward / cnn_conv2d_backward_data re-arrange the filters for their kernels in a small launch of their own
 * on every call, because the filters may have changed (Conv2D::update_gradients, conv2d.cpp:205-217).  A caller that
 * knows WHEN they change -- once per SGD step -- can re-arrange the filters of all layers with one or two launches and
 * pass the results in; outputs are bit-identical to the plain calls.
 *   fwd[i] / dgrad[i]: device buffers of cnn_conv2d_prepared_bytes(&descs[i]) bytes each (NULL entries / arrays skip that
 *   mode); they must be re-prepared after every change of w[i] / bias[i].  At most 6 layers per call. */
size_t cnn_conv2d_prepared_bytes(const cnn_conv2d_desc* d);
int cnn_conv2d_prepare_filters(int n, const cnn_conv2d_desc* descs, const float* const* w, const float* const* bias,
                               void* const* fwd, void* const* dgrad, void* stream);
/* cnn_conv2d_forward / _forward_relu (y_relu nullable) from prepared filters */
int cnn_conv2d_forward_prepared(const cnn_conv2d_desc* d, const float* x, const void* prepared_fwd, const float* bias,
                                float* y, float* y_relu, void* stream);
int cnn_conv2d_backward_data_prepared(const cnn_conv2d_desc* d, const float* dy, const void* prepared_dgrad, float* dx,
                                      void* stream);
/* cnn_conv2d_relu_maxpool2_forward from prepared filters (the bias is part of the prepared buffer) */
int cnn_conv2d_relu_maxpool2_forward_prepared(const cnn_conv2d_desc* d, const float* x, const void* prepared_fwd,
                                              float* pooled, int32_t* mask, void* stream);
int cnn_conv2d_backward_data_pooled2_prepared(const cnn_conv2d_desc* d, const float* dpool, const int32_t* mask,
                                              const float* pooled, const void* prepared_dgrad, float* dx, void* stream);
int cnn_conv2d_backward_data_relu_prepared(const cnn_conv2d_desc* d, const float* dy, const void* prepared_dgrad,
                                           const float* relu_below, float* dx, void* stream);
/* cnn_conv2d_backward_weight_pooled2 (side stream) + cnn_conv2d_backward_data_pooled2_prepared in one call, like
 * cnn_conv2d_backward_prepared */
int cnn_conv2d_backward_pooled2_prepared(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask,
                                         const float* pooled, const void* prepared_dgrad, float* gw, float* gb, float* dx,
                                         float divisor, void* workspace, size_t workspace_bytes, void* stream, int defer_join);
/* cnn_conv2d_backward with the data gradient from prepared filters; workspace: cnn_conv2d_workspace_bytes(d) */
int cnn_conv2d_backward_prepared(const cnn_conv2d_desc* d, const float* x, const float* dy, const void* prepared_dgrad,
                                 float* gw, float* gb, float* dx, float divisor, void* workspace, size_t workspace_bytes,
                                 void* stream, int defer_join);
/* ... and with the ReLU::backward of the layer in front fused into the data gradient (relu_below nullable) */
int cnn_conv2d_backward_prepared_relu(const cnn_conv2d_desc* d, const float* x, const float* dy, const void* prepared_dgrad,
                                      const float* relu_below, float* gw, float* gb, float* dx, float divisor,
                                      void* workspace, size_t workspace_bytes, void* stream, int defer_join);

/* ---- BatchNorm2D : batchnorm2d.cpp:24-95 (forward), :98-158 (backward) ------------------------------- */
/* Per-channel statistics over (B,H,W) of an NCHW tensor; gamma/beta/moving_mean/moving_var/saved_* are [C].
 * training != 0 (the reference's !no_grad branch, :46-80): two-pass batch mean and BIASED variance are written to
 *   saved_mean / saved_var (the reference's buffer_mean / buffer_var), y = gamma*((x-mean)/sqrt(var+eps)) + beta, and
 *   moving = (1-momentum)*moving + momentum*stat (moving statistics start at 0/0 in the reference, :20).
 * training == 0 (:82-93): the moving statistics are used; saved_* / workspace may be NULL.
 * The reference's normed_input buffer is not produced: backward recomputes it from x and saved_*.
 * x / y / dy must be 16-byte aligned.  workspace: cnn_batchnorm2d_workspace_bytes(B,C,H,W) bytes. */
size_t cnn_batchnorm2d_workspace_bytes(int B, int C, int H, int W);
int cnn_batchnorm2d_forward(const float* x, float* y, const float* gamma, const float* beta, float* moving_mean,
                            float* moving_var, float* saved_mean, float* saved_var, int B, int C, int H, int W,
                            float eps, float momentum, int training, void* workspace, size_t workspace_bytes,
                            void* stream);
/* BatchNorm2D followed by a ReLU layer (every BN site of the ResNet-shaped stack): y as above AND y_relu = relu(y)
 * (relu.cpp:25) from the same pass -- the ReLU layer's own kernel would read y again.  y_relu: 16-byte aligned, same shape as y.
 * y may be NULL (round 4): then only y_relu is written -- a train step never reads the normalised tensor itself (ReLU::backward masks by
 * its own output, BatchNorm2D::backward recomputes from x and saved_*); cnn_batchnorm2d_forward(training = 0) with saved_mean / saved_var
 * in the place of the moving statistics re-computes y bit for bit when somebody asks for it (the host layer's get_output() does). */
int cnn_batchnorm2d_forward_relu(const float* x, float* y, float* y_relu, const float* gamma, const float* beta,
                                 float* moving_mean, float* moving_var, float* saved_mean, float* saved_var, int B, int C, int H,
                                 int W, float eps, float momentum, int training, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* BatchNorm2D -> ReLU -> MaxPool2D(2, 2) (round 6): the apply pass also pools -- pooled [B][C][H/2][W/2] and pool_mask (int32 flat index into
 * the sample, pool2d.cpp:81; NULL: not recorded) are what cnn_maxpool2d_forward would make of y_relu, bit for bit (the pool's scan order and
 * strict '<').  y and y_relu are each written only when non-NULL: a train step needs neither (cnn_batchnorm2d_backward_pooled takes the
 * ReLU's test from `pooled`).  H, W even, channels beyond the one-workgroup limit (_supported answers 1); tensors 8-byte aligned. */
int cnn_batchnorm2d_forward_relu_pool_supported(int B, int C, int H, int W);
int cnn_batchnorm2d_forward_relu_pool(const float* x, float* y, float* y_relu, float* pooled, int32_t* pool_mask, const float* gamma,
                                      const float* beta, float* moving_mean, float* moving_var, float* saved_mean, float* saved_var, int B,
                                      int C, int H, int W, float eps, float momentum, int training, void* workspace, size_t workspace_bytes,
                                      void* stream);
/* dy is overwritten with dx IN PLACE, like the reference (:149-155).  ggamma[c] = sum dy*norm, gbeta[c] = sum dy:
 * plain sums over (B,H,W), NOT divided by the batch (:123-124).  x is the forward input, saved_* the batch
 * statistics of that forward call. */
int cnn_batchnorm2d_backward(const float* x, float* dy, const float* gamma, const float* saved_mean,
                             const float* saved_var, float* ggamma, float* gbeta, int B, int C, int H, int W, float eps,
                             void* workspace, size_t workspace_bytes, void* stream);
/* BatchNorm2D <- ReLU <- MaxPool2D(2, 2) (round 6): cnn_maxpool2d_backward_relu + cnn_batchnorm2d_backward without the tensor in between.
 * The delta at the normalisation's output is REBUILT from the pooled domain -- dpool [B][C][H/2][W/2] (the pool's incoming delta), mask (the
 * pool's int32 flat index into the sample, pool2d.cpp:81) and pooled (the pool's output = the ReLU output at the maximum: relu.cpp:37's test)
 * -- inside the two backward kernels, which walk the same elements in the same order with the same arithmetic as cnn_batchnorm2d_backward:
 * ggamma, gbeta and dx are BIT-IDENTICAL to the three-call sequence.  dx [B][C][H][W] is written (nothing is read there).  H even, W a
 * multiple of 4 (_supported answers 1); x / dx 16-byte, dpool / mask / pooled 8-byte aligned; workspace as cnn_batchnorm2d_backward's. */
int cnn_batchnorm2d_backward_pooled_supported(int B, int C, int H, int W);
int cnn_batchnorm2d_backward_pooled(const float* x, const float* dpool, const int32_t* mask, const float* pooled, float* dx,
                                    const float* gamma, const float* saved_mean, const float* saved_var, float* ggamma, float* gbeta,
                                    int B, int C, int H, int W, float eps, void* workspace, size_t workspace_bytes, void* stream);

/* BatchNorm2D with the batch SHARDED over data-parallel processes ("sync-BN"): the reference normalises over the whole
 * batch (batchnorm2d.cpp:46-63), so each rank computes per-channel partial sums, the caller all-reduces (sum) the small
 * vectors between the calls, and the apply kernels use the GLOBAL element count `count` = B_global*H*W.  Training step:
 *   cnn_batchnorm2d_partial_sums(x, NULL, 0, s1)              all-reduce s1[C]     (sum of x)
 *   cnn_batchnorm2d_partial_sums(x, s1, count, s2)            all-reduce s2[C]     (sum of (x - s1/count)^2: two-pass)
 *   cnn_batchnorm2d_forward_from_sums(..., s1, s2, count)     mean / biased variance -> saved_*, moving_*, y
 *   cnn_batchnorm2d_backward_sums(x, dy, ..., sums4)          all-reduce sums4[C][4]
 *   cnn_batchnorm2d_backward_from_sums(..., sums4, count)     dy -> dx in place; ggamma / gbeta = the FULL-batch sums
 *                                                             (identical on every rank: exclude them from the gradient
 *                                                             all-reduce or divide by the world size afterwards)
 * With one process this is the arithmetic of cnn_batchnorm2d_forward / _backward on their general path (layers whose channels
 * fit LDS take a one-workgroup-per-channel kernel there, which sums a channel in a different order: equal within rounding).
 * Evaluation needs no exchange. */
int cnn_batchnorm2d_partial_sums(const float* x, const float* sum_x, float count, float* out, int B, int C, int H, int W,
                                 void* workspace, size_t workspace_bytes, void* stream);
int cnn_batchnorm2d_forward_from_sums(const float* x, float* y, const float* gamma, const float* beta, float* moving_mean,
                                      float* moving_var, float* saved_mean, float* saved_var, const float* sum_x,
                                      const float* sum_sq, float count, int B, int C, int H, int W, float eps,
                                      float momentum, void* stream);
/* ... with the ReLU output of the layer behind it from the same pass (see cnn_batchnorm2d_forward_relu) */
int cnn_batchnorm2d_forward_from_sums_relu(const float* x, float* y, float* y_relu, const float* gamma, const float* beta,
                                           float* moving_mean, float* moving_var, float* saved_mean, float* saved_var,
                                           const float* sum_x, const float* sum_sq, float count, int B, int C, int H, int W,
                                           float eps, float momentum, void* stream);
int cnn_batchnorm2d_backward_sums(const float* x, const float* dy, const float* gamma, const float* saved_mean,
                                  const float* saved_var, float* sums4, int B, int C, int H, int W, float eps,
                                  void* workspace, size_t workspace_bytes, void* stream);
int cnn_batchnorm2d_backward_from_sums(const float* x, float* dy, const float* gamma, const float* saved_mean,
                                       const float* saved_var, const float* sums4, float count, float* ggamma,
                                       float* gbeta, int B, int C, int H, int W, float eps, void* stream);

/* ---- SGD step : conv2d.cpp:205-217, linear.cpp:95-102 ---------------------------------------------- */
/* p -= lr * (g * grad_scale) over one flat parameter arena.  grad_scale = 1 reproduces the reference exactly
 * (two roundings, no FMA); grad_scale = 1/G folds the data-parallel mean after an all-reduce(sum) over G ranks
 * whose kernels each divided by their local batch. */
int cnn_sgd_update(float* params, const float* grads, size_t n, float lr, float grad_scale, void* stream);
/* the same step; previous[i] (nullable) receives params[i] as it was before (see cnn_conv2d_backward_weight_pooled2_sgd_keep) */
int cnn_sgd_update_keep(float* params, const float* grads, size_t n, float lr, float grad_scale, float* previous, void* stream);

/* AlexNet::grad_cam (alexnet.cpp:107-140) from the feature map of the chosen layer, [B][C][H][W] on the device:
 *   weights[b][o] = mean_i feature[b][o][i];  cam[b] = ReLU(sum_o weights[b][o] * feature[b][o]);  cam = (cam - min) / (max - min)
 * with min / max over the whole [B][H][W] tensor (:136-139).  cam: [B][H][W] floats (output).  image (nullable): H*W bytes, the 8-bit
 * picture of the FIRST plane -- what the reference returns as cv::Mat through Tensor3D::opecv_mat(1) (data_format.cpp:98-103). */
int cnn_grad_cam(const float* feature, int B, int C, int H, int W, float* cam, unsigned char* image, void* stream);

/* ---- data-parallel gradient exchange (new: the reference is single-process) -------------
 * The batch is sharded over G replicas (one per GPU); the only cross-sample coupling of the path is the batch mean inside
 * the weight / bias gradients (conv2d.cpp:148,157; linear.cpp:62,70).  Every replica's kernels divide by their LOCAL batch,
 * cnn_allreduce_grads sums the flat gradient arena (or one bucket of it) in place over the replicas -- RCCL ncclAllReduce,
 * fp32 sum, over xGMI -- and cnn_sgd_update(..., grad_scale = 1/G) folds the rest: exactly the reference's (1/B) * sum over the
 * whole batch up to summation order.  BatchNorm2D gamma / beta gradients computed by the sync-BN entry points are already
 * full-batch sums on every replica; summing them over G and scaling by 1/G leaves them unchanged, so they ride in the same
 * arena.  librccl is bound at run time (dlopen("librccl.so.1")): libcnn_amd.so has no link-time dependency on it.
 *   one process per GPU : rank 0 calls cnn_comm_unique_id, ships the 128 bytes to the other ranks by any means (the test
 *                         harness uses torch.distributed's store), every rank calls cnn_comm_init_rank on its device;
 *   one process, n GPUs : cnn_comm_init_all (ncclCommInitAll), one host thread per device or cnn_comm_group_start/_end
 *                         around the per-device calls. */
#define CNN_COMM_ID_BYTES 128
int cnn_comm_available(void); /* != 0: librccl was found and bound */
int cnn_comm_version(void);   /* ncclGetVersion code, 0 when unavailable */
int cnn_comm_unique_id(void* id_out /* host, CNN_COMM_ID_BYTES */);
int cnn_comm_init_rank(void** comm, int world, int rank, const void* id /* host, CNN_COMM_ID_BYTES */);
int cnn_comm_init_all(void** comms /* [ndev] */, int ndev, const int* devices /* NULL = 0..ndev-1 */);
int cnn_comm_info(void* comm, int* world, int* rank);
int cnn_comm_destroy(void* comm);
int cnn_comm_group_start(void);
int cnn_comm_group_end(void);
/* in-place sum of n floats over all ranks of comm, enqueued on `stream` */
int cnn_allreduce_grads(void* comm, float* grads, size_t n, void* stream);
/* a second communicator over the same ranks (ncclCommSplit; every rank calls it with the same color, key orders the ranks): collectives
 * on different communicators are independent queues, so BatchNorm2D's small sync-BN reductions (batchnorm2d.cpp:46-61, 129-147) need not
 * queue behind the bucketed gradient exchange.  Fails with CNN_AMD_E_COMM when the bound librccl has no ncclCommSplit. */
int cnn_comm_split(void* comm, int color, int key, void** new_comm);
/* `bytes` bytes at buf (device) of rank `root` to every rank of comm, in place, enqueued on `stream` */
int cnn_comm_broadcast(void* comm, void* buf, size_t bytes, int root, void* stream);

/* ---- loss glue : func.cpp:16-73 (caller side of the path; keeps the step on the device) -------------- */
/* probs = softmax(logits) with the reference's clamped exp and NaN->0; delta = probs - onehot(labels);
 * loss_sum[0] = -sum_b log(probs[b][label_b])  (the caller divides by the global batch, func.cpp:67,71).
 * probs / loss_sum may be NULL. */
int cnn_softmax_xent(const float* logits, const int32_t* labels, float* probs, float* delta, float* loss_sum,
                     int B, int classes, void* stream);
/* LinearLayer::forward of the last layer + cnn_softmax_xent in ONE kernel (out <= 8): logits, probs (nullable), delta as
 * above, and the per-sample loss term log(probs[b][label_b]) into loss_terms[b]; cnn_loss_from_terms adds the terms in
 * ascending sample order (the reference's order) whenever the loss value is actually wanted.  Bit-identical to
 * cnn_linear_forward + cnn_softmax_xent. */
int cnn_linear_forward_softmax_xent(const float* x, const float* w, const float* bias, const int32_t* labels, float* logits,
                                    float* probs, float* delta, float* loss_terms, int B, int in, int out, void* stream);
/* ... and the data gradient of that layer in the same kernel: dx[b][i] = sum_j delta[b][j] * W[i][j] (linear.cpp:73-90; a sample's row
 * needs only that sample's delta), relu_below != 0: followed by the ReLU::backward of the layer whose output is x (x <= 0 ? 0 : dx).
 * Bit-identical to cnn_linear_forward_softmax_xent + the dx of cnn_linear_backward(_relu).  The layer's weight / bias gradient (which
 * needs every sample's delta) is cnn_linear_backward(x, delta, w, gw, gb, NULL, ...): with dx == NULL it runs the same summation
 * order as the three-output call, on any stream -- a train step then has ONE kernel between its last forward convolution and its
 * first data gradient. */
int cnn_linear_forward_softmax_xent_dx(const float* x, const float* w, const float* bias, const int32_t* labels, float* logits,
                                       float* probs, float* delta, float* loss_terms, float* dx, int relu_below, int B, int in, int out,
                                       void* stream);
int cnn_loss_from_terms(const float* loss_terms, float* loss_sum, int B, void* stream);

/* ---- device memory / transfer helpers (the host layer classes use only these) ------------------------ */
int cnn_device_alloc(void** ptr, size_t bytes);
int cnn_device_free(void* ptr);
int cnn_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream); /* async on stream */
int cnn_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream); /* async on stream */
int cnn_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int cnn_memset_zero(void* dst_dev, size_t bytes, void* stream);
int cnn_stream_synchronize(void* stream);
/* streams / events for callers that overlap independent work themselves (the communication stream of the data-parallel
 * exchange, the staging stream of cnn_batch_upload_async); streams are non-blocking w.r.t. the default stream, events carry
 * no timing.  The host layer classes use only these, never the HIP runtime directly. */
int cnn_stream_create(void** stream);
/* level < 0: the device's highest stream priority, 0: default, > 0: lowest (hipStreamCreateWithPriority; dispatch arbitration between
 * queues, running waves are not preempted) */
int cnn_stream_create_priority(void** stream, int level);
int cnn_stream_destroy(void* stream);
int cnn_event_create(void** event);
int cnn_event_destroy(void* event);
int cnn_event_record(void* event, void* stream);
int cnn_stream_wait_event(void* stream, void* event);
/* like cnn_stream_wait_event, for a dependency that only concerns work queued on `stream` ITSELF: the caller asserts that nothing the
 * library forks off `stream` afterwards (the weight-gradient side stream of cnn_conv2d_backward*) depends on the event, so a fork
 * point published by the last kernel (cnn_amd_publish_next_kernel) stays valid.  cnn_stream_wait_event invalidates it. */
int cnn_stream_wait_event_local(void* stream, void* event);
/* ---- input staging (row n4: what DataLoader::generate_batch's host buffers -- pipeline.cpp:112-140 -- need in front of a device
 * consumer).  A stager owns `depth` (>= 2) page-locked host slots and as many device buffers of batch_bytes each, a copy stream
 * and the events that order producer -> H2D -> consumer -> producer:
 *   acquire(&host, &slot)   the pinned slot to fill next; BLOCKS (host) until the consumer released it from its previous use
 *   submit(slot, &dev)      enqueue the H2D copy of that slot on the stager's own stream; dev = its device buffer
 *   wait(slot, stream)      make `stream` wait for that copy (enqueue only)
 *   release(slot, stream)   `stream` is done reading the device buffer: the slot may be overwritten once it gets there
 * so that the upload of batch i+1 overlaps the kernels of batch i.  A batch of 256 x 3 x 224 x 224 floats is 154 MB: ~2.4 ms
 * over PCIe Gen5 x16, i.e. the PCIe-inclusive ceiling of the reference net is ~10^5 images/s per GPU (bench.py --staged-input). */
int cnn_batch_stager_create(void** stager, size_t batch_bytes, int depth);
/* The same for what a real input is BEFORE Tensor3D::read_from_opencv_mat (data_format.cpp:13-23): B images of H x W x 3 interleaved
 * bytes (cv::Mat CV_8UC3, channel order as it lies -- BGR for cv::imread).  The pinned slots and the H2D copies hold BYTES (a quarter
 * of the fp32 batch: 150 KB instead of 602 KB per 224 x 224 image); behind the copy, still on the stager's own stream, one kernel
 * writes the batch the layers consume -- fp32, planar [B][3][H][W], data[c*H*W + i] = byte * 1.f / 255 through a 256-entry table that
 * the HOST fills with that very expression, so the result is bit-identical to the reference's conversion.  Any H, W (pixel counts that
 * are multiples of four take the 16-byte-store kernel).  submit() returns the fp32 batch; acquire / wait / release / destroy as above. */
int cnn_batch_stager_create_u8(void** stager, int B, int H, int W, int depth);
int cnn_batch_stager_destroy(void* stager);
int cnn_batch_stager_acquire(void* stager, void** pinned_host, int* slot);
int cnn_batch_stager_submit(void* stager, int slot, void** device_ptr);
int cnn_batch_stager_wait(void* stager, int slot, void* stream);
int cnn_batch_stager_release(void* stager, int slot, void* stream);

/* page-locked host memory (async H2D / D2H copies are only asynchronous from / to pinned buffers) */
int cnn_host_alloc_pinned(void** ptr, size_t bytes);
int cnn_host_free_pinned(void* ptr);

#ifdef __cplusplus
}
#endif
#endif /* CNN_AMD_H */
