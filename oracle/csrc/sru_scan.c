/* ORACLE (test infrastructure, not product code): the SRU recurrence of oracle/sru_ref.py as a plain C loop, so that the CPU baseline of bench.py
 * does not time a Python loop over time steps (the reference's CPU path runs the sru package's compiled loop; published algorithm: sru 2.6.0
 * elementwise_recurrence_naive, call site /root/reference/src/models/layers/rnn_layers.py:100-105,150).  Same arithmetic, same order as
 * sru_ref.sru_cell_forward:
 *     f = sigmoid(u1 + bf + wf c);  r = sigmoid(u2 + br + wr c);  c = u0 + (c - u0) f;  h = x' + (c - x') r
 * U: [L][B][D][d][k]; xp: [L][B][D][d] (k == 3: x scale_x) or NULL (k == 4: x' = U[..., 3]); wc, bias: [2][D][d]; h: [L][B][D][d]; c_last: [B][D][d].
 * Direction 0 scans t = 0 .. L-1, direction 1 scans t = L-1 .. 0 over the same (unflipped) input.  nthreads: the caller's thread count (torch.get_num_threads(): torch may bundle its own OpenMP runtime, whose setting this library would not see).
 * Built by oracle/build_c.py (gcc -O3 -fopenmp). */
#include <math.h>
#include <stddef.h>

#define SRU_SCAN(NAME, T, EXP)                                                                                                         \
    void NAME(const T* U, const T* xp, const T* wc, const T* bias, int L, int B, int D, int d, int k, T* h, T* c_last, int nthreads) {   \
        const T *wf = wc, *wr = wc + (size_t)D * d, *bf = bias, *br = bias + (size_t)D * d;                                               \
        _Pragma("omp parallel for collapse(2) schedule(static) num_threads(nthreads)") for (int b = 0; b < B; ++b) for (int di = 0; di < D; ++di) {             \
            T c[256];                                                                                                                     \
            for (int j = 0; j < d; ++j) c[j] = 0;                                                                                         \
            for (int s = 0; s < L; ++s) {                                                                                                 \
                const int t = di == 0 ? s : L - 1 - s;                                                                                    \
                const size_t o = (((size_t)t * B + b) * D + di) * d;                                                                      \
                const T* u = U + o * k;                                                                                                   \
                for (int j = 0; j < d; ++j) {                                                                                             \
                    const T u0 = u[j * k], u1 = u[j * k + 1] + bf[di * d + j], u2 = u[j * k + 2] + br[di * d + j];                        \
                    const T x = xp ? xp[o + j] : u[j * k + 3];                                                                            \
                    const T f = (T)1 / ((T)1 + EXP(-(u1 + c[j] * wf[di * d + j])));                                                       \
                    const T r = (T)1 / ((T)1 + EXP(-(u2 + c[j] * wr[di * d + j])));                                                       \
                    c[j] = u0 + (c[j] - u0) * f;                                                                                          \
                    h[o + j] = x + (c[j] - x) * r;                                                                                        \
                }                                                                                                                         \
            }                                                                                                                             \
            for (int j = 0; j < d; ++j) c_last[((size_t)b * D + di) * d + j] = c[j];                                                      \
        }                                                                                                                                 \
    }

SRU_SCAN(sru_scan_f32, float, expf)
SRU_SCAN(sru_scan_f64, double, exp)
