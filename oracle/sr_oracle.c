/* Plain-C float32 restatement of the SR network (encoder_10 + decoder_400).
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg -- never by the product path.
 * PARITY UNPINNED (see oracle/sr_oracle.py header): restates published Keras
 * layer semantics; the reference has no golden vectors for this path.
 *
 * Follows, in the reference checkout:
 *   encoder_10   sr-ae-conv.ipynb:c162-169     decoder_400  sr-ae-conv.ipynb:c277-287
 *   composition  PyCFD_ML_accelerated.py:686-689
 * Straight loops in the textbook order (bias first, then taps in (ky,kx,ci)
 * order), one sample per OpenMP iteration.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float silu_f(float x) { return x / (1.0f + expf(-x)); }

/* Conv2D, NHWC, TF-SAME (pad_before = total/2), w: (kh,kw,Cin,Cout). c164-165,c286 */
static void conv2d_same(const float* x, int h, int w, int cin, const float* k, const float* b,
                        int kh, int kw, int cout, int stride, int act, float* y) {
  int oh = (h + stride - 1) / stride, ow = (w + stride - 1) / stride;
  int pth = (oh - 1) * stride + kh - h; if (pth < 0) pth = 0;
  int ptw = (ow - 1) * stride + kw - w; if (ptw < 0) ptw = 0;
  int pt = pth / 2, pl = ptw / 2;
  for (int oy = 0; oy < oh; ++oy)
    for (int ox = 0; ox < ow; ++ox) {
      float* out = y + ((size_t)oy * ow + ox) * cout;
      for (int co = 0; co < cout; ++co) out[co] = b[co];
      for (int ky = 0; ky < kh; ++ky) {
        int iy = oy * stride - pt + ky;
        if (iy < 0 || iy >= h) continue;
        for (int kx = 0; kx < kw; ++kx) {
          int ix = ox * stride - pl + kx;
          if (ix < 0 || ix >= w) continue;
          const float* in = x + ((size_t)iy * w + ix) * cin;
          const float* kk = k + ((size_t)ky * kw + kx) * cin * cout;
          for (int ci = 0; ci < cin; ++ci) {
            float v = in[ci];
            const float* kr = kk + (size_t)ci * cout;
            for (int co = 0; co < cout; ++co) out[co] += v * kr[co];
          }
        }
      }
      if (act) for (int co = 0; co < cout; ++co) out[co] = silu_f(out[co]);
    }
}

/* Conv2DTranspose VALID, w: (kh,kw,Cout,Cin), gather form of the scatter
 * out[s*i+a, s*j+b, co] += x[i,j,ci]*w[a,b,co,ci].  c281-285 */
static void conv2d_transpose_valid(const float* x, int h, int w, int cin, const float* k,
                                   const float* b, int kh, int kw, int cout, int stride,
                                   float* y) {
  int oh = (h - 1) * stride + kh, ow = (w - 1) * stride + kw;
  for (int oy = 0; oy < oh; ++oy)
    for (int ox = 0; ox < ow; ++ox) {
      float* out = y + ((size_t)oy * ow + ox) * cout;
      for (int co = 0; co < cout; ++co) out[co] = b[co];
      for (int a = 0; a < kh; ++a) {
        int ty = oy - a;
        if (ty < 0 || ty % stride) continue;
        int i = ty / stride; if (i >= h) continue;
        for (int bb = 0; bb < kw; ++bb) {
          int tx = ox - bb;
          if (tx < 0 || tx % stride) continue;
          int j = tx / stride; if (j >= w) continue;
          const float* in = x + ((size_t)i * w + j) * cin;
          const float* kk = k + ((size_t)a * kw + bb) * cout * cin;
          for (int co = 0; co < cout; ++co) {
            const float* kr = kk + (size_t)co * cin;
            float s = 0.f;
            for (int ci = 0; ci < cin; ++ci) s += in[ci] * kr[ci];
            out[co] += s;
          }
        }
      }
      for (int co = 0; co < cout; ++co) out[co] = silu_f(out[co]);
    }
}

static void dense_f(const float* x, int in, const float* k, const float* b, int out, int act,
                    float* y) {
  for (int o = 0; o < out; ++o) y[o] = b[o];
  for (int i = 0; i < in; ++i) {
    float v = x[i];
    const float* kr = k + (size_t)i * out;
    for (int o = 0; o < out; ++o) y[o] += v * kr[o];
  }
  if (act) for (int o = 0; o < out; ++o) y[o] = silu_f(y[o]);
}

/* enc_w: conv2d k,b, conv2d_1 k,b, dense k,b, latent_vector k,b  (8 pointers)
 * dec_w: dense_1 k,b, conv2d_transpose{,_1.._4} k,b, output_image_400 k,b (14 pointers)
 * x: (n,10,10,1)  y: (n,400,400,1).  Returns 0, or -1 on allocation failure. */
int sr_oracle_forward_f32(const float* x, int n, const float* const* enc_w,
                          const float* const* dec_w, float* y, float* latent_out) {
  int fail = 0;
#pragma omp parallel for schedule(dynamic)
  for (int s = 0; s < n; ++s) {
    float* a1 = (float*)malloc(sizeof(float) * 5 * 5 * 64);
    float* a2 = (float*)malloc(sizeof(float) * 5 * 5 * 128);
    float a3[128], z[50];
    float* bufA = (float*)malloc(sizeof(float) * 400 * 400 * 8);
    float* bufB = (float*)malloc(sizeof(float) * 200 * 200 * 16);
    if (!a1 || !a2 || !bufA || !bufB) { fail = 1; free(a1); free(a2); free(bufA); free(bufB); continue; }
    conv2d_same(x + (size_t)s * 100, 10, 10, 1, enc_w[0], enc_w[1], 3, 3, 64, 2, 1, a1);
    conv2d_same(a1, 5, 5, 64, enc_w[2], enc_w[3], 3, 3, 128, 1, 1, a2);
    dense_f(a2, 3200, enc_w[4], enc_w[5], 128, 1, a3);
    dense_f(a3, 128, enc_w[6], enc_w[7], 50, 0, z);
    if (latent_out) memcpy(latent_out + (size_t)s * 50, z, sizeof(z));
    dense_f(z, 50, dec_w[0], dec_w[1], 36864, 1, bufB);                       /* (12,12,256) */
    conv2d_transpose_valid(bufB, 12, 12, 256, dec_w[2], dec_w[3], 3, 3, 128, 2, bufA); /* 25 */
    conv2d_transpose_valid(bufA, 25, 25, 128, dec_w[4], dec_w[5], 2, 2, 64, 2, bufB);  /* 50 */
    conv2d_transpose_valid(bufB, 50, 50, 64, dec_w[6], dec_w[7], 2, 2, 32, 2, bufA);   /* 100 */
    conv2d_transpose_valid(bufA, 100, 100, 32, dec_w[8], dec_w[9], 2, 2, 16, 2, bufB); /* 200 */
    conv2d_transpose_valid(bufB, 200, 200, 16, dec_w[10], dec_w[11], 2, 2, 8, 2, bufA);/* 400 */
    conv2d_same(bufA, 400, 400, 8, dec_w[12], dec_w[13], 3, 3, 1, 1, 0, y + (size_t)s * 160000);
    free(a1); free(a2); free(bufA); free(bufB);
  }
  return fail ? -1 : 0;
}
