/*
 * oracle/msda_oracle.c -- CPU restatement of Snipper's multi-scale deformable
 * attention core op (forward + backward).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke() of
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The
 * product path (snipper_amd/) never links, imports or falls back to it.
 *
 * What it restates (reference = /root/reference, JimmyZou/Snipper):
 *   forward  : models/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (sampling
 *              kernel) + :33-84 (4-tap bilinear read), which is the same
 *              function as ms_deform_attn_core_pytorch,
 *              models/ops/functions/ms_deform_attn_func.py:45-65
 *              (grid_sample, bilinear, zeros padding, align_corners=False).
 *   backward : ms_deform_im2col_cuda.cuh:87-159 (tap gradients) as driven by
 *              :513-616; grad_loc is w.r.t. the NORMALISED location, hence the
 *              W / H factors (:157-158).
 *
 * Layouts (all row-major, contiguous):
 *   value [N,S,M,D]   loc [N,Lq,M,L,P,2] (x,y in [0,1])   attn [N,Lq,M,L,P]
 *   shapes [L,2] int64 (H,W)   level_start [L] int64   out [N,Lq,M*D]
 *
 * Pinned against the reference by tests/test_oracle.py: golden vectors made by
 * importing the reference's own Python path (tests/golden/gen_golden.py).
 *
 * Parallelism: `threads` > 1 splits the work over OpenMP threads with a
 * decomposition that keeps every output element owned by one thread
 * (forward: over (n,q); backward: over (n,m)), so results do not depend on it.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* One bilinear footprint: up to four taps, each a pixel index inside the level
 * (or -1 when the tap falls outside the map) and its interpolation weight. */
#define DEFINE_ORACLE(T, SUFFIX)                                                          \
  typedef struct {                                                                        \
    int pix[4];                                                                           \
    T wt[4];                                                                              \
    T lh, lw;                                                                             \
    int inside;                                                                           \
  } footprint_##SUFFIX;                                                                   \
                                                                                          \
  static footprint_##SUFFIX make_footprint_##SUFFIX(T lx, T ly, int H, int W) {           \
    footprint_##SUFFIX f;                                                                 \
    const T y = ly * (T)H - (T)0.5; /* cuh:285 */                                         \
    const T x = lx * (T)W - (T)0.5; /* cuh:286 */                                         \
    f.inside = (y > (T)-1 && x > (T)-1 && y < (T)H && x < (T)W); /* cuh:288 */            \
    const int y0 = (int)floor((double)y), x0 = (int)floor((double)x);                     \
    f.lh = y - (T)y0;                                                                     \
    f.lw = x - (T)x0;                                                                     \
    const T hh = (T)1 - f.lh, hw = (T)1 - f.lw;                                           \
    f.wt[0] = hh * hw; f.wt[1] = hh * f.lw; f.wt[2] = f.lh * hw; f.wt[3] = f.lh * f.lw;   \
    const int yok0 = (y0 >= 0), yok1 = (y0 + 1 <= H - 1);                                 \
    const int xok0 = (x0 >= 0), xok1 = (x0 + 1 <= W - 1);                                 \
    f.pix[0] = (yok0 && xok0) ? y0 * W + x0 : -1;                                         \
    f.pix[1] = (yok0 && xok1) ? y0 * W + x0 + 1 : -1;                                     \
    f.pix[2] = (yok1 && xok0) ? (y0 + 1) * W + x0 : -1;                                   \
    f.pix[3] = (yok1 && xok1) ? (y0 + 1) * W + x0 + 1 : -1;                               \
    return f;                                                                             \
  }                                                                                       \
                                                                                          \
  void msda_oracle_forward_##SUFFIX(const T *value, const int64_t *shapes,                \
                                    const int64_t *level_start, const T *loc,             \
                                    const T *attn, int N, int S, int M, int D, int L,     \
                                    int Lq, int P, T *out, int threads) {                 \
    const int64_t NQ = (int64_t)N * Lq;                                                   \
    (void)threads;                                                                        \
    _Pragma("omp parallel for schedule(static) num_threads(threads)")                    \
    for (int64_t nq = 0; nq < NQ; ++nq) {                                                 \
      const int n = (int)(nq / Lq);                                                       \
      for (int m = 0; m < M; ++m) {                                                       \
        T *o = out + (nq * M + m) * D;                                                    \
        for (int c = 0; c < D; ++c) o[c] = (T)0;                                          \
        const T *lp = loc + (nq * M + m) * (int64_t)L * P * 2;                            \
        const T *ap = attn + (nq * M + m) * (int64_t)L * P;                               \
        for (int l = 0; l < L; ++l) {                                                     \
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                   \
          const T *vbase = value + ((int64_t)n * S + level_start[l]) * M * D + m * D;     \
          for (int p = 0; p < P; ++p) {                                                   \
            const footprint_##SUFFIX f =                                                  \
                make_footprint_##SUFFIX(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W); \
            if (!f.inside) continue;                                                      \
            const T a = ap[l * P + p];                                                    \
            for (int c = 0; c < D; ++c) {                                                 \
              T v[4];                                                                     \
              for (int k = 0; k < 4; ++k)                                                 \
                v[k] = f.pix[k] >= 0 ? vbase[(int64_t)f.pix[k] * M * D + c] : (T)0;       \
              const T s = f.wt[0] * v[0] + f.wt[1] * v[1] + f.wt[2] * v[2] + f.wt[3] * v[3]; \
              o[c] += s * a;                                                              \
            }                                                                             \
          }                                                                               \
        }                                                                                 \
      }                                                                                   \
    }                                                                                     \
  }                                                                                       \
                                                                                          \
  void msda_oracle_backward_##SUFFIX(const T *value, const int64_t *shapes,               \
                                     const int64_t *level_start, const T *loc,            \
                                     const T *attn, const T *grad_out, int N, int S,      \
                                     int M, int D, int L, int Lq, int P, T *grad_value,   \
                                     T *grad_loc, T *grad_attn, int threads) {            \
    memset(grad_value, 0, sizeof(T) * (size_t)N * S * M * D);                             \
    const int NM = N * M;                                                                 \
    (void)threads;                                                                        \
    _Pragma("omp parallel for schedule(static) num_threads(threads)")                    \
    for (int nm = 0; nm < NM; ++nm) {                                                     \
      const int n = nm / M, m = nm % M;                                                   \
      for (int q = 0; q < Lq; ++q) {                                                      \
        const int64_t row = ((int64_t)n * Lq + q) * M + m;                                \
        const T *g = grad_out + row * D;                                                  \
        const T *lp = loc + row * (int64_t)L * P * 2;                                     \
        const T *ap = attn + row * (int64_t)L * P;                                        \
        T *gl = grad_loc + row * (int64_t)L * P * 2;                                      \
        T *ga = grad_attn + row * (int64_t)L * P;                                         \
        for (int l = 0; l < L; ++l) {                                                     \
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                   \
          const int64_t off = ((int64_t)n * S + level_start[l]) * M * D + m * D;          \
          const T *vbase = value + off;                                                   \
          T *gvbase = grad_value + off;                                                   \
          for (int p = 0; p < P; ++p) {                                                   \
            const int sp = l * P + p;                                                     \
            const footprint_##SUFFIX f =                                                  \
                make_footprint_##SUFFIX(lp[sp * 2], lp[sp * 2 + 1], H, W);                \
            T acc_a = (T)0, acc_x = (T)0, acc_y = (T)0;                                   \
            if (f.inside) {                                                               \
              const T a = ap[sp];                                                         \
              const T hh = (T)1 - f.lh, hw = (T)1 - f.lw;                                 \
              for (int c = 0; c < D; ++c) {                                               \
                const T ga_c = g[c] * a; /* cuh:110 */                                    \
                T v[4];                                                                   \
                for (int k = 0; k < 4; ++k) {                                             \
                  if (f.pix[k] >= 0) {                                                    \
                    v[k] = vbase[(int64_t)f.pix[k] * M * D + c];                          \
                    gvbase[(int64_t)f.pix[k] * M * D + c] += f.wt[k] * ga_c;              \
                  } else {                                                                \
                    v[k] = (T)0;                                                          \
                  }                                                                       \
                }                                                                         \
                /* d(sample)/dy and d(sample)/dx in pixel units, cuh:113-153 */           \
                const T dy = -hw * v[0] - f.lw * v[1] + hw * v[2] + f.lw * v[3];          \
                const T dx = -hh * v[0] + hh * v[1] - f.lh * v[2] + f.lh * v[3];          \
                const T s = f.wt[0] * v[0] + f.wt[1] * v[1] + f.wt[2] * v[2] + f.wt[3] * v[3]; \
                acc_a += g[c] * s;            /* cuh:156 */                               \
                acc_x += (T)W * dx * ga_c;    /* cuh:157 */                               \
                acc_y += (T)H * dy * ga_c;    /* cuh:158 */                               \
              }                                                                           \
            }                                                                             \
            ga[sp] = acc_a;                                                               \
            gl[sp * 2] = acc_x;                                                           \
            gl[sp * 2 + 1] = acc_y;                                                       \
          }                                                                               \
        }                                                                                 \
      }                                                                                   \
    }                                                                                     \
  }

DEFINE_ORACLE(double, f64)
DEFINE_ORACLE(float, f32)

int msda_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
