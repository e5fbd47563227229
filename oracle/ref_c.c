/*
 * C restatement of the integer and double-precision parts of the policy-update
 * hot path.  TEST INFRASTRUCTURE ONLY: it is the checker, never the product and
 * never the thing measured as the product.  Parity pin: tests/test_oracle_c.py
 * checks every function here against the golden vectors in tests/golden/, which
 * oracle/make_golden.py produced from the reference's own functions.
 *
 * Reference files followed (relative to the reference root):
 *   src/prime_rl/trainer/utils.py:36-69        sample lengths of a packed row
 *   src/prime_rl/orchestrator/batch.py:76-102  first-fit-decreasing packing
 *   src/prime_rl/trainer/rl/loss.py:13-58      shift, log-softmax gather, entropy
 *   src/prime_rl/orchestrator/advantage.py:9-57 group advantages
 *
 * The floating-point functions accumulate in double, so they serve as a
 * higher-precision arbiter when the fp32 oracle and the fp32 kernels differ
 * in the last digits.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* src/prime_rl/trainer/utils.py:50-69.  lengths has room for max(n, 1) entries; returns how many were written
 * (an empty row is one sample of length 0: boundaries = [0] at :52). */
int64_t prl_oracle_response_lengths(const int64_t* pos, int64_t n, int64_t* lengths) {
  int64_t count = 0, start = 0;
  for (int64_t i = 1; i < n; ++i) {
    if (pos[i] == 0 && pos[i - 1] != 0 && i + 1 < n && pos[i + 1] == 1) {
      lengths[count++] = i - start;
      start = i;
    }
  }
  lengths[count++] = n - start;
  return count;
}

/* src/prime_rl/orchestrator/batch.py:81-102.  bin_of[i] receives the bin of sample i; returns the bin count.
 * Samples are taken in stable descending order of length; each goes to the first bin with room. */
int64_t prl_oracle_first_fit_decreasing(const int64_t* len, int64_t n, int64_t capacity, int64_t* bin_of) {
  int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  int64_t bins = 0;
  for (int64_t i = 0; i < n; ++i) order[i] = i;
  for (int64_t i = 1; i < n; ++i) { /* stable insertion sort, descending */
    int64_t k = order[i], j = i - 1;
    while (j >= 0 && len[order[j]] < len[k]) {
      order[j + 1] = order[j];
      --j;
    }
    order[j + 1] = k;
  }
  for (int64_t q = 0; q < n; ++q) {
    int64_t i = order[q], b = 0;
    for (; b < bins; ++b)
      if (fill[b] + len[i] <= capacity) break;
    if (b == bins) fill[bins++] = 0;
    fill[b] += len[i];
    bin_of[i] = b;
  }
  free(order);
  free(fill);
  return bins;
}

/*
 * Per-token log-probability and entropy of row-major fp32 logits [T, V] with
 * the shift by one position and the temperature of the trainer's chain
 * (src/prime_rl/trainer/rl/train.py:286-288, loss.py:27-30, 43-47), in double.
 */
void prl_oracle_logprob_entropy_f64(const float* logits, int64_t T, int64_t V, const int64_t* ids,
                                    double temperature, int shift, double* logprob, double* entropy, double* lse) {
  for (int64_t t = 0; t < T; ++t) {
    if (shift && t == 0) {
      double l = log((double)V);
      logprob[t] = -l;
      entropy[t] = l;
      lse[t] = l;
      continue;
    }
    const float* x = logits + (t - shift) * V;
    double m = -INFINITY;
    for (int64_t v = 0; v < V; ++v) {
      double y = (double)x[v] / temperature;
      if (y > m) m = y;
    }
    double s = 0.0, w = 0.0;
    for (int64_t v = 0; v < V; ++v) {
      double y = (double)x[v] / temperature;
      double e = exp(y - m);
      s += e;
      w += y * e;
    }
    double l = m + log(s);
    lse[t] = l;
    entropy[t] = l - w / s;
    logprob[t] = (ids[t] >= 0 && ids[t] < V) ? (double)x[ids[t]] / temperature - l : NAN;
  }
}

/*
 * dlogits[r, v] = g[t] / temperature * (1[v == ids[t]] - exp(x[r, v] / temperature - lse[t])), t = r + shift,
 * the closed form of autograd through the same chain; the row that shifting drops is zero.
 */
void prl_oracle_logprob_bwd_f64(const float* logits, int64_t T, int64_t V, const int64_t* ids, const double* lse,
                                const double* g, double temperature, int shift, double* dlogits) {
  for (int64_t r = 0; r < T; ++r) {
    int64_t t = r + shift;
    double* d = dlogits + r * V;
    if (t >= T || g[t] == 0.0) {
      for (int64_t v = 0; v < V; ++v) d[v] = 0.0;
      continue;
    }
    const float* x = logits + r * V;
    double gi = g[t] / temperature;
    for (int64_t v = 0; v < V; ++v) d[v] = -gi * exp((double)x[v] / temperature - lse[t]);
    if (ids[t] >= 0 && ids[t] < V) d[ids[t]] += gi;
  }
}

/* src/prime_rl/orchestrator/advantage.py:18, 32, 41-43, 55-57; kind 0..3 in registry order. */
void prl_oracle_group_advantage(const double* rewards, const int64_t* lengths, int64_t B, int64_t G, int kind,
                                double* out) {
  for (int64_t lo = 0; lo + G <= B; lo += G) {
    double num = 0.0, den = 0.0;
    for (int64_t i = 0; i < G; ++i) {
      double w = kind == 3 ? (double)lengths[lo + i] : 1.0;
      num += rewards[lo + i] * w;
      den += w;
    }
    double base = num / den;
    for (int64_t i = 0; i < G; ++i) {
      double a = rewards[lo + i] - base;
      if (kind == 1 && a < 0.0) a = 0.0;
      if (kind == 2) a = a * (double)G / (double)(G - 1);
      out[lo + i] = a;
    }
  }
}
