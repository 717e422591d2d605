/*
 * cnn_oracle.c -- CPU restatement of the hermosayhl/CNN layer arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cnn_amd/ may include, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PIN STATUS: the reference ships no tests and cannot be built here (every
 * translation unit pulls <opencv2/core.hpp> through cpu/include/data_format.h:7 and
 * OpenCV is not in the image), so there is no oracle/_ref.  The FORWARD path of this
 * file is pinned against the one known answer the reference publishes: README.md:92 /
 * imgs/image-20230208213627060.png (inference.exe on datasets/images/{dog,panda,bird}.jpg
 * with cpu/checkpoints/AlexNet_aug_1e-3/iter_395000_train_0.918_valid_0.913.model gives
 * 0.850634 / 0.999978 / 0.999998), see tests/golden/make_readme_kat.py and
 * tests/test_oracle_golden.py.  The BACKWARD path has no reference-held vector:
 * "parity unpinned" for backward beyond (a) being the exact derivative of the pinned
 * forward (checked in fp64 by finite differences, tests/test_oracle_gradcheck.py) and
 * (b) following the reference loop nests line by line as cited below.
 *
 * All tensors are contiguous NCHW; the reference holds a batch as B separate CHW
 * buffers (cpu/include/data_format.h:11-53), which is the same element order per sample.
 *
 * Build: see oracle/Makefile (-O2, no -march, -ffp-contract=off: the reference builds with
 * "-std=c++17 -O2", cpu/CMakeLists.txt:5, so sums stay sequential and un-fused).
 * Compile with -DORACLE_REAL=double for the fp64 arbiter (symbols get a _f64 suffix).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORACLE_REAL
#define ORACLE_REAL float
#define SUF(name) name
#define REAL_SQRT(v) sqrtf(v) /* std::sqrt(data_type) is the float overload */
#define REAL_EXP(v) expf(v)   /* std::exp(data_type), func.cpp:10: the float overload */
#define REAL_LOG(v) logf(v)   /* std::log(data_type), func.cpp:65 */
#else
#define SUF(name) name##_f64
#define REAL_SQRT(v) sqrt(v)
#define REAL_EXP(v) exp(v)
#define REAL_LOG(v) log(v)
#endif
typedef ORACLE_REAL real;

/* ------------------------------------------------------------------------------------------
 * Conv2D  (cpu/src/conv2d.cpp)
 * ---------------------------------------------------------------------------------------- */

/* out_H = (H - k - 2*0)/s + 1 with integer division: conv2d.cpp:41-42 */
static int conv_out_dim(int H, int k, int s) { return (H - k) / s + 1; }

/*
 * conv2d.cpp:69-92.  Window centre walks x = r; x < H - r; x += s (:76-77); accumulation
 * order i -> kx -> ky, sequential in `real`, then + bias (:78-87).
 * w layout [Co][Ci][k][k] (conv2d.cpp:18-21), bias [Co].
 */
void SUF(oracle_conv2d_forward)(const real* x, const real* w, const real* bias, real* y,
                                int B, int Ci, int H, int W, int Co, int k, int s) {
    const int r = (k - 1) / 2;
    const int Ho = conv_out_dim(H, k, s), Wo = conv_out_dim(W, k, s);
    const int plane = H * W, oplane = Ho * Wo, win = k * k;
    /* (threads split the independent (b, o) planes; every output element stays one sequential sum) */
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; ++b) {
        for (int o = 0; o < Co; ++o) {
            const real* xb = x + (size_t)b * Ci * plane;
            real* yo = y + ((size_t)b * Co + o) * oplane;
            const real* wo = w + (size_t)o * Ci * win;
            int cnt = 0;
            for (int cx = r; cx < H - r; cx += s) {
                for (int cy = r; cy < W - r; cy += s) {
                    real sum = 0;
                    for (int i = 0; i < Ci; ++i) {
                        const real* xi = xb + (size_t)i * plane + cx * W + cy;
                        const real* wi = wo + i * win;
                        int t = 0;
                        for (int dx = -r; dx <= r; ++dx)
                            for (int dy = -r; dy <= r; ++dy, ++t)
                                sum += xi[dx * W + dy] * wi[t];
                    }
                    sum += bias[o];
                    yo[cnt] = sum;
                    ++cnt;
                }
            }
        }
    }
}

/*
 * conv2d.cpp:117-159 (weight / bias gradients) and :168-199 (data gradient).
 *  gw[o][i][kx][ky] += (sum_{x,y} dy[b,o,x,y] * in[b,i,x*s+kx,y*s+ky]) / B   per sample b (:135-148)
 *  gb[o]            += (sum_d dy[b,o,d]) / B                                   per sample b (:153-157)
 *  dx zeroed (:168) then scatter-add dx[b,i,win] += w[o,i,t] * dy[b,o,cnt] over the forward's
 *  centre walk (:175-199).
 * Any of gw/gb/dx may be NULL to skip that part.
 */
void SUF(oracle_conv2d_backward)(const real* x, const real* dy, const real* w, real* gw, real* gb,
                                 real* dx, int B, int Ci, int H, int W, int Co, int k, int s) {
    const int r = (k - 1) / 2;
    const int Ho = conv_out_dim(H, k, s), Wo = conv_out_dim(W, k, s);
    const int plane = H * W, oplane = Ho * Wo, win = k * k;
    if (gw) memset(gw, 0, sizeof(real) * (size_t)Co * Ci * win);
    if (gb) memset(gb, 0, sizeof(real) * (size_t)Co);
    if (gw || gb) {
        /* (threads split the output channels: each gw[o] / gb[o] still accumulates its samples in ascending b) */
#pragma omp parallel for schedule(dynamic)
        for (int o = 0; o < Co; ++o) {
            for (int b = 0; b < B; ++b) {
                const real* od = dy + ((size_t)b * Co + o) * oplane;
                if (gw) {
                    for (int i = 0; i < Ci; ++i) {
                        const real* in = x + ((size_t)b * Ci + i) * plane;
                        real* g = gw + ((size_t)o * Ci + i) * win;
                        for (int kx = 0; kx < k; ++kx) {
                            for (int ky = 0; ky < k; ++ky) {
                                real sum = 0;
                                for (int px = 0; px < Ho; ++px) {
                                    const real* drow = od + px * Wo;
                                    const real* irow = in + (px * s + kx) * W;
                                    for (int py = 0; py < Wo; ++py) sum += drow[py] * irow[py * s + ky];
                                }
                                g[kx * k + ky] += sum / B;
                            }
                        }
                    }
                }
                if (gb) {
                    real sum = 0;
                    for (int d = 0; d < oplane; ++d) sum += od[d];
                    gb[o] += sum / B;
                }
            }
        }
    }
    if (dx) {
        memset(dx, 0, sizeof(real) * (size_t)B * Ci * plane);
        /* (threads split (sample, input channel): each dx element still receives its contributions in ascending o, then
         * in window order -- the reference's o -> window -> i nest visits a fixed (i, element) in exactly that order) */
#pragma omp parallel for collapse(2) schedule(dynamic)
        for (int b = 0; b < B; ++b) {
          for (int i = 0; i < Ci; ++i) {
            real* dxb = dx + (size_t)b * Ci * plane;
            for (int o = 0; o < Co; ++o) {
                const real* od = dy + ((size_t)b * Co + o) * oplane;
                const real* wo = w + (size_t)o * Ci * win;
                int cnt = 0;
                for (int cx = r; cx < H - r; cx += s) {
                    for (int cy = r; cy < W - r; cy += s) {
                        {
                            real* xi = dxb + (size_t)i * plane + cx * W + cy;
                            const real* wi = wo + i * win;
                            int t = 0;
                            for (int ddx = -r; ddx <= r; ++ddx)
                                for (int ddy = -r; ddy <= r; ++ddy, ++t)
                                    xi[ddx * W + ddy] += wi[t] * od[cnt];
                        }
                        ++cnt;
                    }
                }
            }
          }
        }
    }
}

/* Threads used by the convolution loop nests above (test infrastructure only: big parity cases finish in seconds on the GPU
 * box's host cores).  The split never changes the order in which any single result element is accumulated, so the output is
 * bit-identical for every thread count (tests/test_oracle_golden.py checks that); bench.py's cpu_baseline pins it to 1, the
 * reference being single-threaded. */
#ifdef _OPENMP
#include <omp.h>
void SUF(oracle_set_threads)(int n) { omp_set_num_threads(n > 0 ? n : omp_get_num_procs()); }
int SUF(oracle_get_threads)(void) { return omp_get_max_threads(); }
#else
void SUF(oracle_set_threads)(int n) { (void)n; }
int SUF(oracle_get_threads)(void) { return 1; }
#endif

/* conv2d.cpp:205-217 and linear.cpp:95-102:  p -= lr * g */
void SUF(oracle_sgd_update)(real* p, const real* g, size_t n, real lr) {
    for (size_t i = 0; i < n; ++i) p[i] -= lr * g[i];
}

/* ------------------------------------------------------------------------------------------
 * MaxPool2D  (cpu/src/pool2d.cpp)
 * ---------------------------------------------------------------------------------------- */

/*
 * pool2d.cpp:53-87.  out = (H - k)/step + 1 (:14-15); window top-left walks x = 0; x <= H-k;
 * x += step (:63-65); max starts at window[0], replaced only when max < comp in row-major window
 * order (:67-75): first maximum wins, a NaN never replaces a non-NaN, -0/+0 tie.
 * mask[b][c*Ho*Wo + cnt] = c*H*W + (x+di)*W + (y+dj): int32 index into the sample's C*H*W (:79-82).
 * mask may be NULL (the no_grad path, :41,61,79).
 */
void SUF(oracle_maxpool_forward)(const real* x, real* y, int* mask, int B, int C, int H, int W, int k,
                                 int step) {
    const int Ho = (H - k) / step + 1, Wo = (W - k) / step + 1;
    const int plane = H * W, oplane = Ho * Wo;
    for (int b = 0; b < B; ++b) {
        for (int c = 0; c < C; ++c) {
            const real* xc = x + ((size_t)b * C + c) * plane;
            real* yc = y + ((size_t)b * C + c) * oplane;
            int* mc = mask ? mask + ((size_t)b * C + c) * oplane : NULL;
            int cnt = 0;
            for (int px = 0; px <= H - k; px += step) {
                const real* row = xc + px * W;
                for (int py = 0; py <= W - k; py += step) {
                    real best = row[py];
                    int best_off = 0;
                    for (int t = 1; t < k * k; ++t) {
                        const int off = (t / k) * W + (t % k);
                        const real comp = row[py + off];
                        if (best < comp) {
                            best = comp;
                            best_off = off;
                        }
                    }
                    yc[cnt] = best;
                    if (mc) mc[cnt] = c * plane + px * W + py + best_off;
                    ++cnt;
                }
            }
        }
    }
}

/* pool2d.cpp:96-107: dx zeroed, then dx[mask[i]] = dy[i] (assignment, ascending i). */
void SUF(oracle_maxpool_backward)(const real* dy, const int* mask, real* dx, int B, int C, int H, int W,
                                  int k, int step) {
    const int Ho = (H - k) / step + 1, Wo = (W - k) / step + 1;
    const size_t in_len = (size_t)C * H * W, out_len = (size_t)C * Ho * Wo;
    memset(dx, 0, sizeof(real) * B * in_len);
    for (int b = 0; b < B; ++b) {
        const real* src = dy + b * out_len;
        const int* m = mask + b * out_len;
        real* dst = dx + b * in_len;
        for (size_t i = 0; i < out_len; ++i) dst[m[i]] = src[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * ReLU  (cpu/src/relu.cpp)
 * ---------------------------------------------------------------------------------------- */

/* relu.cpp:21-26: y = x >= 0 ? x : 0  (keeps -0.0, NaN -> 0) */
void SUF(oracle_relu_forward)(const real* x, real* y, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] = x[i] >= 0 ? x[i] : 0;
}

/* relu.cpp:35-40: dy = (y <= 0) ? 0 : dy, in place on the caller's delta */
void SUF(oracle_relu_backward)(const real* y, real* dy, size_t n) {
    for (size_t i = 0; i < n; ++i) dy[i] = y[i] <= 0 ? 0 : dy[i];
}

/* ------------------------------------------------------------------------------------------
 * Dropout  (cpu/src/dropout.cpp) -- channel dropout; commented out of the reference net (alexnet.cpp:28)
 * ---------------------------------------------------------------------------------------- */

/* dropout.cpp:7-55.  selected_num = int(p * C) (:18).  Training (:30-41): the loop tests the channel INDEX o against
 * selected_num -- channels o >= selected_num are copied, the others zeroed; the shuffled `sequence` (:26) only fills `mask`
 * (:33) and never selects data, so the dropped set is always channels 0 .. selected_num-1.  no_grad (:44-53): y = x * (1 - p). */
void SUF(oracle_dropout_forward)(const real* x, real* y, int B, int C, int area, real p, int training) {
    const int selected = (int)(p * C);
    const real prob = 1 - p;
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < C; ++o) {
            const real* src = x + ((size_t)b * C + o) * area;
            real* dst = y + ((size_t)b * C + o) * area;
            for (int i = 0; i < area; ++i) dst[i] = training ? (o >= selected ? src[i] : 0) : src[i] * prob;
        }
}

/* dropout.cpp:57-69: channels with mask[o] == -1 (o < selected_num, :33) get a zero delta, in place */
void SUF(oracle_dropout_backward)(real* dy, int B, int C, int area, real p) {
    const int selected = (int)(p * C);
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < selected; ++o) memset(dy + ((size_t)b * C + o) * area, 0, sizeof(real) * (size_t)area);
}

/* ------------------------------------------------------------------------------------------
 * AlexNet::grad_cam  (cpu/src/alexnet.cpp:95-142) -- the arithmetic behind the returned picture.  PARITY UNPINNED: the function
 * returns cv::Mat and cannot be built here, and the reference holds no vector for it; this follows the loops line by line.
 * ---------------------------------------------------------------------------------------- */

/* alexnet.cpp:107-140 on the feature map [B][C][H][W] of the chosen layer.  :111-119 weights[b][o] = (sum_i fea)/area -- the
 * channel mean of the FEATURE MAP (the delta computed in :97-102 is not used); :121-131 cam[b][i] = sum_o w * fea, in o order;
 * :133-134 ReLU `if (v < 0) v = 0`; :136-139 min-max normalisation over the whole [B][H][W] tensor with Tensor3D::min / max
 * (data_format.cpp:37-62: first extremum, strict comparisons).  image (nullable): opecv_mat(1) of the result = the first plane as
 * saturate_cast<uchar>(255 * v) (data_format.cpp:98-103; cvRound = lrint, clamped; NaN -> 0). */
void SUF(oracle_grad_cam)(const real* fea, int B, int C, int H, int W, real* cam, unsigned char* image) {
    const int area = H * W;
    real* weights = (real*)malloc(sizeof(real) * (size_t)B * C);
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < C; ++o) {
            const real* f = fea + ((size_t)b * C + o) * area;
            real mean_value = 0;
            for (int i = 0; i < area; ++i) mean_value += f[i];
            weights[b * C + o] = mean_value / area;
        }
    const size_t length = (size_t)B * area;
    memset(cam, 0, sizeof(real) * length);
    for (int b = 0; b < B; ++b) {
        real* c = cam + (size_t)b * area;
        for (int o = 0; o < C; ++o) {
            const real w = weights[b * C + o];
            const real* f = fea + ((size_t)b * C + o) * area;
            for (int i = 0; i < area; ++i) c[i] += w * f[i];
        }
    }
    free(weights);
    for (size_t i = 0; i < length; ++i)
        if (cam[i] < 0) cam[i] = 0;
    size_t lo = 0, hi = 0;
    for (size_t i = 1; i < length; ++i) {
        if (cam[i] < cam[lo]) lo = i;
        if (cam[i] > cam[hi]) hi = i;
    }
    const real min_value = cam[lo], max_value = cam[hi];
    const real res_value = max_value - min_value;
    for (size_t i = 0; i < length; ++i) cam[i] = (cam[i] - min_value) / res_value;
    if (image)
        for (int i = 0; i < area; ++i) {
            const real sv = 255 * cam[i];
            long r = (sv != sv) ? 0 : lrint((double)sv);
            image[i] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
}

/* ------------------------------------------------------------------------------------------
 * LinearLayer  (cpu/src/linear.cpp); W stored [in][out] row-major (linear.cpp:40)
 * ---------------------------------------------------------------------------------------- */

/* linear.cpp:33-43: y[b][i] = (sum_j x[b][j] * W[j*out + i]) + bias[i] */
void SUF(oracle_linear_forward)(const real* x, const real* w, const real* bias, real* y, int B, int in,
                                int out) {
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < out; ++i) {
            real sum = 0;
            for (int j = 0; j < in; ++j) sum += x[(size_t)b * in + j] * w[(size_t)j * out + i];
            y[(size_t)b * out + i] = sum + bias[i];
        }
}

/*
 * linear.cpp:56-71: gW[i*out+j] = (sum_b x[b][i]*dy[b][j]) / B (assigned); gb[j] = (sum_b dy[b][j]) / B
 * linear.cpp:80-90: dx[b][i] = sum_j dy[b][j] * W[i*out + j]
 */
void SUF(oracle_linear_backward)(const real* x, const real* dy, const real* w, real* gw, real* gb, real* dx,
                                 int B, int in, int out) {
    if (gw)
        for (int i = 0; i < in; ++i)
            for (int j = 0; j < out; ++j) {
                real sum = 0;
                for (int b = 0; b < B; ++b) sum += x[(size_t)b * in + i] * dy[(size_t)b * out + j];
                gw[(size_t)i * out + j] = sum / B;
            }
    if (gb)
        for (int j = 0; j < out; ++j) {
            real sum = 0;
            for (int b = 0; b < B; ++b) sum += dy[(size_t)b * out + j];
            gb[j] = sum / B;
        }
    if (dx)
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < in; ++i) {
                real sum = 0;
                for (int j = 0; j < out; ++j) sum += dy[(size_t)b * out + j] * w[(size_t)i * out + j];
                dx[(size_t)b * in + i] = sum;
            }
}

/* ------------------------------------------------------------------------------------------
 * BatchNorm2D  (cpu/src/batchnorm2d.cpp) -- row n1 of SURVEY.md 8(f)
 * ---------------------------------------------------------------------------------------- */

/*
 * batchnorm2d.cpp:24-95.  training != 0 (the !no_grad branch, :46-80): per channel, two-pass batch statistics over
 * (B,H,W) -- mean (:48-55), then BIASED variance of (x-u)^2 (:57-63) --, saved into saved_mean/saved_var (:64-67),
 * var_inv = 1/sqrt(var+eps) (:69), norm = (x-u)*var_inv, y = gamma*norm + beta (:70-77), and the moving statistics
 * m = (1-momentum)*m + momentum*stat (:79-80; they start at 0/0, :20).  training == 0 (:82-93): the moving statistics
 * are used instead.  norm_out may be NULL (the reference keeps it in normed_input for the backward pass).
 */
void SUF(oracle_batchnorm_forward)(const real* x, real* y, real* norm_out, const real* gamma, const real* beta,
                                   real* moving_mean, real* moving_var, real* saved_mean, real* saved_var, int B, int C,
                                   int H, int W, real eps, real momentum, int training) {
    const int hw = H * W;
    const int L = B * hw;
    for (int o = 0; o < C; ++o) {
        real u, var;
        if (training) {
            u = 0;
            for (int b = 0; b < B; ++b) {
                const real* src = x + ((size_t)b * C + o) * hw;
                for (int i = 0; i < hw; ++i) u += src[i];
            }
            u = u / L;
            var = 0;
            for (int b = 0; b < B; ++b) {
                const real* src = x + ((size_t)b * C + o) * hw;
                for (int i = 0; i < hw; ++i) var += (src[i] - u) * (src[i] - u);
            }
            var = var / L;
            if (saved_mean) saved_mean[o] = u;
            if (saved_var) saved_var[o] = var;
        } else {
            u = moving_mean[o];
            var = moving_var[o];
        }
        const real var_inv = (real)(1. / REAL_SQRT((real)(var + eps))) /* :69 */;
        for (int b = 0; b < B; ++b) {
            const real* src = x + ((size_t)b * C + o) * hw;
            real* dst = y + ((size_t)b * C + o) * hw;
            real* nrm = norm_out ? norm_out + ((size_t)b * C + o) * hw : NULL;
            for (int i = 0; i < hw; ++i) {
                const real n = (src[i] - u) * var_inv;
                if (nrm) nrm[i] = n;
                dst[i] = gamma[o] * n + beta[o];
            }
        }
        if (training) {
            moving_mean[o] = (1 - momentum) * moving_mean[o] + momentum * u;
            moving_var[o] = (1 - momentum) * moving_var[o] + momentum * var;
        }
    }
}

/*
 * batchnorm2d.cpp:98-158, in place on dy like the reference (:149-155).  Per channel:
 *   ggamma = sum dy*norm, gbeta = sum dy (NOT divided by the batch, :123-124), norm_g = dy*gamma (:125),
 *   var_g = sum norm_g*(x-u)*(-0.5)*var_inv^3 (:129-137), inv = var_g/L (:140),
 *   u_g = sum [norm_g*(-var_inv) + inv*(-2)*(x-u)] (:139-146),
 *   dx = norm_g*var_inv + inv*2*(x-u) + u_g/L (:148-155).
 */
void SUF(oracle_batchnorm_backward)(const real* x, real* dy, const real* gamma, const real* saved_mean,
                                    const real* saved_var, real* ggamma, real* gbeta, int B, int C, int H, int W,
                                    real eps) {
    const int hw = H * W;
    const int L = B * hw;
    for (int o = 0; o < C; ++o) {
        const real u = saved_mean[o];
        const real var_inv = (real)(1. / REAL_SQRT((real)(saved_var[o] + eps))) /* :110 */;
        const real var_inv_3 = var_inv * var_inv * var_inv;
        real gg = 0, gb = 0, var_g = 0;
        for (int b = 0; b < B; ++b) {
            const real* d = dy + ((size_t)b * C + o) * hw;
            const real* src = x + ((size_t)b * C + o) * hw;
            for (int i = 0; i < hw; ++i) {
                const real n = (src[i] - u) * var_inv;
                gg += d[i] * n;
                gb += d[i];
            }
        }
        for (int b = 0; b < B; ++b) {
            const real* d = dy + ((size_t)b * C + o) * hw;
            const real* src = x + ((size_t)b * C + o) * hw;
            for (int i = 0; i < hw; ++i) var_g += (d[i] * gamma[o]) * (src[i] - u) * (real)(-0.5) * var_inv_3;
        }
        const real inv = var_g / L;
        real u_g = 0;
        for (int b = 0; b < B; ++b) {
            const real* d = dy + ((size_t)b * C + o) * hw;
            const real* src = x + ((size_t)b * C + o) * hw;
            for (int i = 0; i < hw; ++i) u_g += (d[i] * gamma[o]) * (-var_inv) + inv * (-2) * (src[i] - u);
        }
        for (int b = 0; b < B; ++b) {
            real* d = dy + ((size_t)b * C + o) * hw;
            const real* src = x + ((size_t)b * C + o) * hw;
            for (int i = 0; i < hw; ++i) d[i] = (d[i] * gamma[o]) * var_inv + inv * 2 * (src[i] - u) + u_g / L;
        }
        ggamma[o] = gg;
        gbeta[o] = gb;
    }
}

/* ------------------------------------------------------------------------------------------
 * Loss glue (cpu/src/func.cpp) -- caller side of the path, needed for whole-step parity
 * ---------------------------------------------------------------------------------------- */

/* func.cpp:6-12: clamped exp */
static real clamped_exp(real v) {
    if (v >= 88) return (real)FLT_MAX;
    if (v <= -50) return 0;
    return REAL_EXP(v);
}

/* func.cpp:16-37: max-subtracted softmax (max = first maximum, data_format.cpp:37-48), NaN -> 0 */
void SUF(oracle_softmax)(const real* logits, real* probs, int B, int n) {
    for (int b = 0; b < B; ++b) {
        const real* in = logits + (size_t)b * n;
        real* p = probs + (size_t)b * n;
        real mx = in[0];
        for (int i = 1; i < n; ++i)
            if (in[i] > mx) mx = in[i];
        real sum = 0;
        for (int i = 0; i < n; ++i) {
            p[i] = clamped_exp(in[i] - mx);
            sum += p[i];
        }
        for (int i = 0; i < n; ++i) p[i] /= sum;
        for (int i = 0; i < n; ++i)
            if (isnan(p[i])) p[i] = 0;
    }
}

/* func.cpp:56-73: delta = p - onehot (no 1/B); loss = -(sum_b sum_i log(p)*y) / B */
real SUF(oracle_cross_entropy_backward)(const real* probs, const int* labels, real* delta, int B, int n) {
    real loss = 0;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < n; ++i) {
            const real yv = (labels[b] == i) ? 1 : 0;
            delta[(size_t)b * n + i] = probs[(size_t)b * n + i] - yv;
            loss += REAL_LOG(probs[(size_t)b * n + i]) * yv;
        }
    return (real)(loss * (-1.0) / B);
}

/* ------------------------------------------------------------------------------------------
 * The reference network (cpu/src/alexnet.cpp:10-33, batch_norm=false) as one object:
 *   Conv(3->16,k3,s2) ReLU MaxPool(2,2) Conv(16->32) ReLU Conv(32->64) ReLU Conv(64->128) ReLU
 *   Linear(128*6*6 -> classes)
 * Parameters live in ONE flat buffer in checkpoint order (alexnet.cpp:69-77; conv2d.cpp:220-226:
 * Co filters then Co biases; linear.cpp:105-108: in*out then out), so a .model file memcpy's in.
 * ---------------------------------------------------------------------------------------- */

#define NCONV 4
typedef struct {
    int B, classes, H, W;
    int ci[NCONV], co[NCONV], hin[NCONV], win[NCONV], hout[NCONV], wout[NCONV];
    int lin_in;
    size_t n_params;
    size_t w_off[NCONV], b_off[NCONV], lw_off, lb_off;
    real *params, *grads;
    /* activations */
    real *conv_out[NCONV], *relu_out[NCONV], *pool_out;
    int* pool_mask;
    real* logits;
    /* deltas (what each layer's backward returns) */
    real *d_lin, *d_conv[NCONV], *d_pool;
    const real* input;
} SUF(oracle_net);

static real* ralloc(size_t n) { return (real*)calloc(n ? n : 1, sizeof(real)); }

SUF(oracle_net)* SUF(oracle_net_create)(int B, int classes, int H, int W) {
    SUF(oracle_net)* n = (SUF(oracle_net)*)calloc(1, sizeof(*n));
    static const int chans[NCONV + 1] = {3, 16, 32, 64, 128};
    n->B = B; n->classes = classes; n->H = H; n->W = W;
    int h = H, w = W;
    size_t off = 0;
    for (int l = 0; l < NCONV; ++l) {
        n->ci[l] = chans[l]; n->co[l] = chans[l + 1];
        n->hin[l] = h; n->win[l] = w;
        n->hout[l] = conv_out_dim(h, 3, 2); n->wout[l] = conv_out_dim(w, 3, 2);
        n->w_off[l] = off; off += (size_t)n->co[l] * n->ci[l] * 9;
        n->b_off[l] = off; off += (size_t)n->co[l];
        h = n->hout[l]; w = n->wout[l];
        if (l == 0) { h = (h - 2) / 2 + 1; w = (w - 2) / 2 + 1; } /* MaxPool2D(2,2), alexnet.cpp:16 */
    }
    n->lin_in = 128 * h * w;
    n->lw_off = off; off += (size_t)n->lin_in * classes;
    n->lb_off = off; off += (size_t)classes;
    n->n_params = off;
    n->params = ralloc(off); n->grads = ralloc(off);
    for (int l = 0; l < NCONV; ++l) {
        size_t osz = (size_t)B * n->co[l] * n->hout[l] * n->wout[l];
        n->conv_out[l] = ralloc(osz); n->relu_out[l] = ralloc(osz);
        n->d_conv[l] = ralloc((size_t)B * n->ci[l] * n->hin[l] * n->win[l]);
    }
    size_t psz = (size_t)B * 16 * n->hin[1] * n->win[1];
    n->pool_out = ralloc(psz);
    n->pool_mask = (int*)calloc(psz, sizeof(int));
    n->d_pool = ralloc((size_t)B * 16 * n->hout[0] * n->wout[0]);
    n->logits = ralloc((size_t)B * classes);
    n->d_lin = ralloc((size_t)B * n->lin_in);
    return n;
}

void SUF(oracle_net_destroy)(SUF(oracle_net)* n) {
    if (!n) return;
    for (int l = 0; l < NCONV; ++l) { free(n->conv_out[l]); free(n->relu_out[l]); free(n->d_conv[l]); }
    free(n->pool_out); free(n->pool_mask); free(n->d_pool); free(n->logits); free(n->d_lin);
    free(n->params); free(n->grads); free(n);
}

size_t SUF(oracle_net_num_params)(const SUF(oracle_net)* n) { return n->n_params; }
real* SUF(oracle_net_params)(SUF(oracle_net)* n) { return n->params; }
real* SUF(oracle_net_grads)(SUF(oracle_net)* n) { return n->grads; }
int SUF(oracle_net_linear_in)(const SUF(oracle_net)* n) { return n->lin_in; }

/* alexnet.cpp:35-46; returns the logits buffer [B][classes] */
const real* SUF(oracle_net_forward)(SUF(oracle_net)* n, const real* x) {
    const int B = n->B;
    n->input = x;
    const real* cur = x;
    for (int l = 0; l < NCONV; ++l) {
        SUF(oracle_conv2d_forward)(cur, n->params + n->w_off[l], n->params + n->b_off[l], n->conv_out[l], B,
                                   n->ci[l], n->hin[l], n->win[l], n->co[l], 3, 2);
        size_t osz = (size_t)B * n->co[l] * n->hout[l] * n->wout[l];
        SUF(oracle_relu_forward)(n->conv_out[l], n->relu_out[l], osz);
        cur = n->relu_out[l];
        if (l == 0) {
            SUF(oracle_maxpool_forward)(cur, n->pool_out, n->pool_mask, B, 16, n->hout[0], n->wout[0], 2, 2);
            cur = n->pool_out;
        }
    }
    SUF(oracle_linear_forward)(cur, n->params + n->lw_off, n->params + n->lb_off, n->logits, B, n->lin_in,
                               n->classes);
    return n->logits;
}

/* alexnet.cpp:49-59 (reverse list); delta [B][classes] is consumed (ReLU masks in place, relu.cpp:37-39) */
void SUF(oracle_net_backward)(SUF(oracle_net)* n, const real* delta) {
    const int B = n->B;
    SUF(oracle_linear_backward)(n->relu_out[NCONV - 1], delta, n->params + n->lw_off, n->grads + n->lw_off,
                                n->grads + n->lb_off, n->d_lin, B, n->lin_in, n->classes);
    real* cur = n->d_lin;
    for (int l = NCONV - 1; l >= 0; --l) {
        size_t osz = (size_t)B * n->co[l] * n->hout[l] * n->wout[l];
        if (l == 0) {
            SUF(oracle_maxpool_backward)(cur, n->pool_mask, n->d_pool, B, 16, n->hout[0], n->wout[0], 2, 2);
            cur = n->d_pool;
        }
        SUF(oracle_relu_backward)(n->relu_out[l], cur, osz);
        const real* lin = (l == 0) ? n->input : (l == 1 ? n->pool_out : n->relu_out[l - 1]);
        SUF(oracle_conv2d_backward)(lin, cur, n->params + n->w_off[l], n->grads + n->w_off[l],
                                    n->grads + n->b_off[l], n->d_conv[l], B, n->ci[l], n->hin[l], n->win[l],
                                    n->co[l], 3, 2);
        cur = n->d_conv[l];
    }
}

/* alexnet.cpp:62-65 */
void SUF(oracle_net_update)(SUF(oracle_net)* n, real lr) {
    SUF(oracle_sgd_update)(n->params, n->grads, n->n_params, lr);
}

/* One iteration of cnn.cpp:79-90: forward, softmax, CE, backward, SGD.  Returns the loss. */
real SUF(oracle_net_train_step)(SUF(oracle_net)* n, const real* x, const int* labels, real lr, real* probs_out) {
    const int B = n->B, C = n->classes;
    const real* logits = SUF(oracle_net_forward)(n, x);
    real* probs = ralloc((size_t)B * C);
    real* delta = ralloc((size_t)B * C);
    SUF(oracle_softmax)(logits, probs, B, C);
    real loss = SUF(oracle_cross_entropy_backward)(probs, labels, delta, B, C);
    SUF(oracle_net_backward)(n, delta);
    SUF(oracle_net_update)(n, lr);
    if (probs_out) memcpy(probs_out, probs, sizeof(real) * (size_t)B * C);
    free(probs); free(delta);
    return loss;
}

/* accessors used by the whole-net parity tests */
const real* SUF(oracle_net_conv_out)(const SUF(oracle_net)* n, int l) { return n->conv_out[l]; }
const real* SUF(oracle_net_relu_out)(const SUF(oracle_net)* n, int l) { return n->relu_out[l]; }
const real* SUF(oracle_net_pool_out)(const SUF(oracle_net)* n) { return n->pool_out; }
const int* SUF(oracle_net_pool_mask)(const SUF(oracle_net)* n) { return n->pool_mask; }
const real* SUF(oracle_net_d_conv)(const SUF(oracle_net)* n, int l) { return n->d_conv[l]; }
